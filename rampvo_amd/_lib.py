"""ctypes binding of libramp_hip.so (C ABI declared in include/ramp_hip.h).

There is no CPU fallback: every operator of this package runs its HIP kernel
or raises.  ``lib()`` raises ``RuntimeError`` if the shared library has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C rampvo_amd/csrc``).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("RAMP_HIP_LIB") or os.path.join(CSRC, "libramp_hip.so")   # env: kernel A/B builds

RAMP_F32, RAMP_F16 = 0, 1
RAMP_EUNSUPPORTED = -4
RAMP_CONV_FP8 = 0x40
RAMP_IN_F32, RAMP_CONV_DIRECT, RAMP_CORR_MFMA32, RAMP_CORR_X2 = 0x10, 0x20, 0x40, 0x80
RAMP_CONV_X3 = 0x80
RAMP_NCHW, RAMP_NHWC, RAMP_NHWC32 = 0, 1, 2
KPLANE = 32            # channels per plane of the packed correlation target maps: [h][128 / KPLANE][w][KPLANE]


def kplane(dtype):
    """channels per plane of the packed correlation target maps for a feature dtype: 64 bytes per pixel and plane
    ([h][4][w][32] fp16, [h][8][w][16] fp32).  fp32 features in the split form (corr_f32_mode() == 2) keep the fp32 CONTAINER
    [h][8][w][16] float32 -- the same 512 bytes per pixel -- whose bytes are [h][4][2][w][32] fp16 parts (RAMP_CORR_X2)"""
    import torch
    return KPLANE if dtype == torch.float16 else KPLANE // 2


def corr_f32_mode():
    """how the tracker computes the correlation volume of fp32 features (RAMP_CORR_F32_MFMA): 2 (default) split fp16 pairs on
    the f16 matrix cores (corr_mfma_kernel<CorrX2>, chunked planes of pairs); 1 the fp32 matrix cores
    (corr_mfma_kernel<float>, chunked fp32 planes); 0 corr_kernel<float>, the reference kernel's summation order, plain planes"""
    import os
    try:
        m = int(os.environ.get("RAMP_CORR_F32_MFMA", "2"))
    except ValueError:
        m = 2
    return m if m in (0, 1, 2) else 2

_ERR = {-1: "RAMP_EINVAL (bad argument)", -2: "RAMP_ELAUNCH (HIP launch/runtime error)",
        -3: "RAMP_EWORKSPACE (workspace too small)", -4: "RAMP_EUNSUPPORTED (size/shape not supported)"}

c_p, c_i, c_i64, c_sz, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float


class CorrLevel(ctypes.Structure):
    _fields_ = [("fmap", c_p), ("H2", c_i), ("W2", c_i), ("coord_div", c_f)]


# name -> (restype, argtypes); also the list the CPU test checks for export
SIGNATURES = {
    "ramp_version": (ctypes.c_char_p, []),
    "ramp_corr_kplane": (c_i, []),
    "ramp_patchify_fwd": (c_i, [c_p, c_p, c_p] + [c_i] * 10 + [c_p]),
    "ramp_frame_gather": (c_i, [c_p] * 9 + [c_i] * 8 + [c_p]),
    "ramp_corr_fwd": (c_i, [c_p, ctypes.POINTER(CorrLevel), c_i, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p]),
    "ramp_ms_lstm_superstate": (c_i, [c_p, c_p, ctypes.POINTER(c_p), c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "ramp_ms_lstm_superstate_mfma": (c_i, [c_p] * 6 + [c_i] * 5 + [c_p]),
    "ramp_conv2d_stats_blocks": (c_i, [c_i] * 7),
    "ramp_graph_edit_host": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "ramp_event_stack_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "ramp_event_stack": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "ramp_depth_median_fill": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "ramp_depth_median": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "ramp_event_topk_workspace_bytes": (c_sz, [c_i, c_i]),
    "ramp_event_topk": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "ramp_pyramid_pack": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "ramp_corr_fwd_ordered": (c_i, [c_p, ctypes.POINTER(CorrLevel), c_i, c_p, c_p, c_p, c_p, c_p, c_i, ctypes.c_long,
                                    ctypes.c_long] + [c_i] * 8 + [c_p]),
    "ramp_se3_exp": (c_i, [c_p, c_p, c_i, c_p]),
    "ramp_se3_log": (c_i, [c_p, c_p, c_i, c_p]),
    "ramp_se3_inv": (c_i, [c_p, c_p, c_i, c_p]),
    "ramp_se3_mul": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "ramp_se3_act4": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "ramp_se3_adj": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "ramp_se3_adjT": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "ramp_transform": (c_i, [c_p] * 7 + [c_i, c_i, c_i, c_p]),
    "ramp_reproject": (c_i, [c_p] * 7 + [c_i, c_i, c_p]),
    "ramp_point_cloud": (c_i, [c_p] * 5 + [c_i, c_i, c_p]),
    "ramp_multi_copy": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "ramp_shift_rows": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "ramp_motionmag": (c_i, [c_p] * 10 + [c_i64, c_i64, c_f, c_p, c_i, c_p]),
    "ramp_motion_model": (c_i, [c_p, c_i, c_f, c_p]),
    "ramp_frame_commit": (c_i, [c_p, c_i, c_i, c_f, c_p, c_i64, c_p, c_i64, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i,
                                ctypes.POINTER(c_p), ctypes.POINTER(c_p), ctypes.POINTER(ctypes.c_long), c_p, c_p]),
    "ramp_frame_begin": (c_i, [c_p, c_i, c_i, c_f, c_p, c_i64, c_p, c_i64, c_p, c_i, c_p]),
    "ramp_group_by_workspace_bytes": (c_sz, [c_i]),
    "ramp_group_by": (c_i, [c_p, c_i, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "ramp_neighbors_workspace_bytes": (c_sz, [c_i]),
    "ramp_neighbors": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i64, c_i64, c_p, c_sz, c_p]),
    "ramp_segment_softmax_sum": (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_i, c_p]),
    "ramp_ba_workspace_bytes": (c_sz, [c_i] * 5),
    "ramp_ba_forward": (c_i, [c_p] * 9 + [c_i] * 7 + [c_p, c_sz, c_p, c_p]),
    "ramp_ba_planned_workspace_bytes": (c_sz, [c_i] * 7),
    "ramp_ba_forward_planned": (c_i, [c_p] * 9 + [c_i] * 7 + [c_p] * 4 + [c_i] + [c_p] * 3 + [c_i, c_p, c_sz, c_p, c_p]),
    "ramp_group_by_small_workspace_bytes": (c_sz, [c_i, c_i]),
    "ramp_group_by_small": (c_i, [c_p, c_p, c_i64, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_sz, c_p]),
    "ramp_neighbors_from_groups": (c_i, [c_p] * 7 + [c_i, c_i, c_p]),
    "ramp_any_nonzero": (c_i, [c_p, ctypes.c_long, c_p, ctypes.c_long, c_p, c_p]),
    "ramp_lstm_superstate_tiled": (c_i, [c_p] * 9 + [c_i, c_i, c_i, c_p]),
    "ramp_lstm_superstate_blocks": (c_i, [c_p] * 9 + [c_i, c_i, c_i, c_i, c_p]),
    "ramp_any_nonzero_blocks": (c_i, [c_p, ctypes.c_long, c_p, ctypes.c_long, c_p, c_p]),
    "ramp_any_nonzero_blocks_clear": (c_i, [c_p, ctypes.c_long, c_p, ctypes.c_long, c_p, c_p, ctypes.c_long, c_p]),
    "ramp_conv2d_nhwc": (c_i, [c_p] * 8 + [c_i] * 8 + [c_f, c_i, c_p]),
    "ramp_in_stats_finalize": (c_i, [c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p]),
    "ramp_affine_relu": (c_i, [c_p, c_p, c_p, c_p, ctypes.c_long, c_i, c_p]),
    "ramp_norm_add_relu": (c_i, [c_p] * 7 + [ctypes.c_long, c_i, c_p]),
    "ramp_affine_relu_f16": (c_i, [c_p, c_p, c_p, c_p, ctypes.c_long, c_i, c_p]),
    "ramp_norm_add_relu_f16": (c_i, [c_p] * 7 + [ctypes.c_long, c_i, c_i, c_p]),
    "ramp_norm_add_relu_f16_acc": (c_i, [c_p, c_p, c_f, c_f, c_p, c_p, c_f, c_f, c_p, ctypes.c_long, c_i, c_i, c_p]),
    "ramp_in_acc_finalize": (c_i, [c_p, c_i, c_f, c_f, c_p, c_p, c_p]),
    "ramp_conv2d_nhwc_multi": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "ramp_upd_row_fuse": (c_i, [c_p] * 6 + [ctypes.c_long, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_i, c_i, c_p]),
    "ramp_upd_gather_mask": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "ramp_upd_gated": (c_i, [c_p] * 5 + [c_f, c_p, c_p, c_p, c_i, c_i, c_p]),
    "ramp_upd_heads": (c_i, [c_p] * 5 + [c_i, c_i, c_f, c_f, c_i, c_p]),
    "ramp_upd_segment_softmax": (c_i, [c_p] * 5 + [c_i, c_i, c_p]),
    "ramp_upd_corr_mlp": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, ctypes.c_long,
                                c_p, c_p, c_f, c_p, c_i, c_p]),
    "ramp_upd_heads_linear": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p]),
    "ramp_upd_fg": (c_i, [c_p] * 9 + [c_i, c_p]),
    "ramp_upd_gru": (c_i, [c_p, c_p, c_p, c_p, c_p, c_f, ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_p, c_p, c_f, c_p,
                           c_p, c_i, c_p]),
    "ramp_upd_gru_heads": (c_i, [c_p, c_p, c_p, c_p, c_p, c_f, ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_p, c_p, c_f, c_p,
                                 c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p]),
    "ramp_upd_mlp_lds_bytes": (c_sz, []),
    "ramp_upd_nbr": (c_i, [c_p] * 8 + [c_i, c_p]),
    "ramp_upd_linear": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "ramp_upd_softagg_frag_rows": (c_sz, [c_i, c_i]),
    "ramp_upd_softagg": (c_i, [c_p] * 10 + [c_i, c_p]),
    "ramp_upd_softagg_finish": (c_i, [c_p] * 6 + [c_i, c_p]),
    # the update operator at fp32 accuracy on the f16 matrix cores (csrc/update_x3.hip)
    "ramp_x3_corr_mlp": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, ctypes.c_long,
                               c_p, c_p, c_f, c_p, c_i, c_p]),
    "ramp_x3_nbr": (c_i, [c_p] * 7 + [c_i, c_p]),
    "ramp_x3_fg": (c_i, [c_p] * 9 + [c_i, c_p]),
    "ramp_x3_segment_softmax": (c_i, [c_p] * 5 + [c_i, c_p]),
    "ramp_x3_linear": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "ramp_x3_gru": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_p, c_p, c_f,
                          c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_p]),
    # device-resident tracking step (csrc/track.hip); the ramp_track descriptor is mirrored in track_dev.py
    "ramp_track_sizeof": (c_sz, []),
    "ramp_track_plan_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "ramp_track_ba_workspace_bytes": (c_sz, [c_i] * 6),
    "ramp_track_plan": (c_i, [c_p, c_i, c_p]),
    "ramp_track_step": (c_i, [c_p, c_i, c_i64, c_i, c_i, c_p, c_p, c_p]),
    "ramp_track_warm": (c_i, [c_p, c_p, c_p]),
    "ramp_stream_delay": (c_i, [c_i, c_p]),
    "ramp_signal_alloc": (c_i, [c_p]),
    "ramp_signal_free": (c_i, [c_p]),
    "ramp_stream_signal": (c_i, [c_p, c_p, ctypes.c_uint32]),
    "ramp_stream_wait_flag": (c_i, [c_p, c_p, ctypes.c_uint32, c_i, c_i, c_p]),
    "ramp_host_device_pointer": (c_i, [c_p, c_p]),
}

_lib = None


def build(force=False, jobs=8):
    """compile every HIP source for gfx950 into csrc/libramp_hip.so"""
    cmd = ["make", "-s", "-C", CSRC, "-j%d" % jobs]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd + ["libramp_hip.so"])
    return LIB_PATH


def register_optional(extra):
    """late registration of signatures for optional translation units"""
    SIGNATURES.update(extra)
    if _lib is not None:
        for name, (res, args) in extra.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "rampvo_amd: %s is missing -- the HIP extension is not built and there is no "
                "CPU fallback (run __graft_entry__.build())" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if l.ramp_corr_kplane() != KPLANE:      # (an A/B build with -DCORR_KPLANE=8: this binding would pack the planes wrong)
            raise RuntimeError("rampvo_amd: %s packs %d-channel correlation planes, this binding is written for %d "
                               "(rebuild without -DCORR_KPLANE, or set rampvo_amd._lib.KPLANE)" % (LIB_PATH, l.ramp_corr_kplane(), KPLANE))
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "status %d" % rc)))


def stream():
    """the current HIP stream of the current device as a void* (raw getter: this is called ~30x per frame)"""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("rampvo_amd operators run on the GPU only (got a %s tensor); "
                               "there is no CPU fallback" % t.device)


def dtype_code(t):
    if t.dtype == torch.float32:
        return RAMP_F32
    if t.dtype == torch.float16:
        return RAMP_F16
    raise RuntimeError("unsupported dtype %s (float32 / float16 only)" % t.dtype)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_ws_cache = {}
_ws_retired = []
_ws_owner = [None]


class scratch_owner:
    """``with scratch_owner(obj):`` -- scratch taken inside belongs to ``obj`` as well as to the stream.  hipGraph
    captures all run on torch's one capture stream, so the stream alone would hand two captured graphs (two
    Patchifiers, a re-capture) the same scratch buffer although they replay on different streams."""

    def __init__(self, owner):
        self.owner = owner

    def __enter__(self):
        self.prev, _ws_owner[0] = _ws_owner[0], id(self.owner)

    def __exit__(self, *exc):
        _ws_owner[0] = self.prev


def workspace(nbytes, device, tag="ws"):
    """grow-only scratch buffer per (device, stream, owner, tag): two trackers on different streams of one GPU never
    share scratch, and neither do two captured graphs (see ``scratch_owner``).  Contents are never reused across
    calls.  A buffer that is outgrown is retired, not freed: launches already queued on its stream (or captured in
    a hipGraph) may still reference it."""
    key = (device, torch._C._cuda_getCurrentRawStream(device.index if device.index is not None
                                                      else torch.cuda.current_device()), _ws_owner[0], tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
