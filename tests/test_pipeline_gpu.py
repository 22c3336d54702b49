"""GPU pipeline tests: rampvo_amd on the MI355X (HIP kernels through the C ABI) against the
golden vectors captured from the reference's python (tests/golden, oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

import pipeline_checks as pc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["SingleScale", "MultiScale"])
def test_patchify_against_reference_golden(mode):
    # encoder runs through MIOpen/own conv kernels: fp32, different summation order than CPU
    print(pc.check_patchify(mode, "cuda", tol=2e-3))


def test_update_operator_against_reference_golden():
    print(pc.check_update("cuda", tol=1e-3))


def test_update_step_teacher_forced():
    """ONE update() (reproject -> corr -> update operator -> BA x2 -> point cloud) and keyframe()
    from the reference's captured state, all HIP kernels, vs the reference's own result"""
    e = pc.check_update_step("cuda")
    print(e)
    scale = max(1.0, e["step"])
    assert e["weight"] <= 1e-4 and e["net"] <= 1e-4
    # north-star tolerance 1e-4 relative (relative to the size of the GN step on this
    # ill-conditioned random-weight problem, like the CPU twin of this test)
    assert e["poses"] <= 1e-4 * scale and e["depths"] <= 1e-4 * scale, e
    for k, v in e.items():
        if k.startswith("kf"):
            assert v <= 1e-5, (k, v)


def test_update_step_teacher_forced_mixed_precision():
    """the same step with MIXED_PRECISION (fp16 features / GEMM I/O, what default.yaml runs) against the
    fp32 reference result: the network outputs that feed BA must agree to fp16 accuracy.  (The
    reference's own fp16 run is not reproducible here; fp16 inputs are exact in the fixture, so the
    differences are the half GEMM I/O roundings.)  Poses are not compared: BA on this random-weight
    problem amplifies input noise ~4000x (see the fp32 twin)."""
    e = pc.check_update_step("cuda", mixed=True)
    print(e)
    assert e["weight"] <= 2e-2 and e["net"] <= 2e-2
    for k, v in e.items():
        if k.startswith("kf") and not k.endswith("poses"):
            assert v <= 2e-3, (k, v)


def test_ramp_vo_free_running_structure():
    print(pc.check_ramp_vo("cuda"))


def test_hip_library_is_the_code_that_ran():
    """the native library is loaded in this process (no silent fallback)"""
    maps = open("/proc/self/maps").read()
    assert "libramp_hip.so" in maps
