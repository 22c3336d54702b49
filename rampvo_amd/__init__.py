"""rampvo_amd -- MI355X-native RAMP-VO tracking hot path.

Mirrors the reference's operator surface for that path (module names follow
``ramp/``): ``altcorr`` (patchify, corr), ``fastba`` (BA, neighbors, reproject),
``lietorch`` (SE3), ``projective_ops``, ``blocks``, ``extractor``, ``net`` (VONet
with ``patchify`` / ``update`` / ``DIM, RES, P``) and ``Ramp_vo``.  Compute goes
through hand-written HIP kernels in ``csrc/`` behind the C ABI of
``include/ramp_hip.h``; there is no CPU fallback.
"""
__version__ = "0.1.0"
