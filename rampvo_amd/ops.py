"""Tensor-level wrappers over the C ABI (include/ramp_hip.h).

Each function takes/returns torch CUDA tensors and enqueues one HIP kernel (or a
short fixed pipeline) on torch's current stream.  No CPU fallbacks.
"""
import ctypes

import os

import torch

from . import _lib
from ._lib import RAMP_CORR_MFMA32 as _LIB_CORR_MFMA32
from ._lib import RAMP_CORR_X2 as _LIB_CORR_X2
from ._lib import workspace as _lib_workspace
from ._lib import KPLANE, RAMP_NHWC32, RAMP_NCHW, RAMP_NHWC, CorrLevel, check, dtype_code, kplane, lib, ptr, require_cuda, stream


# ------------------------------------------------------------------- altcorr
def patchify(net, coords, radius, bilinear=True, layout=RAMP_NCHW, out_layout=RAMP_NCHW):
    """net [n,C,H,W] (NCHW) or [n,H,W,C] (NHWC); coords [n,M,2] -> [n,M,C,d,d] (or [n,M,d,d,C])"""
    require_cuda(net, coords)
    net = net.contiguous()
    coords = coords.contiguous().float()
    if layout == RAMP_NCHW:
        n, C, H, W = net.shape
    else:
        n, H, W, C = net.shape
    M = coords.shape[1]
    d = 2 * radius + 1 if bilinear else 2 * radius + 2
    shape = (n, M, C, d, d) if out_layout == RAMP_NCHW else (n, M, d, d, C)
    out = torch.empty(shape, dtype=net.dtype, device=net.device)
    check(lib().ramp_patchify_fwd(ptr(net), ptr(coords), ptr(out), n, C, H, W, M, radius,
                                  int(bilinear), dtype_code(net), layout, out_layout, stream()),
          "ramp_patchify_fwd")
    return out


def frame_gather(f_nhwc, i_nhwc, image, coords):
    """all gathers of one frame at its patch centres, one launch (csrc/altcorr.hip::frame_gather_kernel):
    f_nhwc [h,w,CF], i_nhwc [h,w,CI] (same dtype), image [3,H,W] fp32, coords [M,2] ->
    gmap [M,3,3,CF], imap [M,CI], patches [M,3,3,3] fp32, clr [M,3] fp32, colors [M,3] uint8 BGR"""
    require_cuda(f_nhwc, i_nhwc, image, coords)
    assert f_nhwc.is_contiguous() and i_nhwc.is_contiguous() and f_nhwc.dtype == i_nhwc.dtype
    image, coords = image.contiguous(), coords.contiguous()
    assert image.dtype == torch.float32 and coords.dtype == torch.float32
    (h, w, CF), CI, M = f_nhwc.shape, i_nhwc.shape[-1], coords.shape[0]
    dev = f_nhwc.device
    gmap = torch.empty(M, 3, 3, CF, dtype=f_nhwc.dtype, device=dev)
    imap = torch.empty(M, CI, dtype=f_nhwc.dtype, device=dev)
    patches = torch.empty(M, 3, 3, 3, dtype=torch.float32, device=dev)
    clr = torch.empty(M, 3, dtype=torch.float32, device=dev)
    colors = torch.empty(M, 3, dtype=torch.uint8, device=dev)
    check(lib().ramp_frame_gather(ptr(f_nhwc), ptr(i_nhwc), ptr(image), ptr(coords), ptr(gmap), ptr(imap),
                                  ptr(patches), ptr(clr), ptr(colors), M, h, w, image.shape[1], image.shape[2], CF,
                                  CI, dtype_code(f_nhwc), stream()), "ramp_frame_gather")
    return gmap, imap, patches, clr, colors


def corr(fmap1, fmaps2, coords, ii, jj, radius=3, coord_divs=(1.0,), layout=RAMP_NCHW, order=None, row_elems=0,
         fast_f32=None, mod_ii=0, mod_jj=0):
    """fused multi-level patch correlation.  order: optional int32 [E] schedule (a permutation of
    the edges, e.g. target-frame-major) -- affects which XCD computes an edge, never a value.

    fmap1  [N1,C,P,P] (NCHW) / [N1,P,P,C] (NHWC) patch features
    fmaps2 list of per-level target maps [N2,C,H,W] / [N2,H,W,C]
    coords [E,2,P,P] float32;  ii,jj [E] int64
    -> [E, 2r+1, 2r+1, P, P, nlevels]
    """
    require_cuda(fmap1, coords, ii, jj, *fmaps2)
    fmap1 = fmap1.contiguous()
    fmaps2 = [f.contiguous() for f in fmaps2]
    coords = coords.contiguous().float()
    ii = ii.contiguous()
    jj = jj.contiguous()
    assert ii.dtype == torch.int64 and jj.dtype == torch.int64
    if layout == RAMP_NCHW:
        N1, C, P, _ = fmap1.shape
    else:
        N1, P, _, C = fmap1.shape
    if layout == RAMP_NHWC32:     # target maps [N2][H][C/32][W][32] (fp16) / [N2][H][C/16][W][16] (fp32) (pyramid_pack); fmap1 stays NHWC
        kp = kplane(fmap1.dtype)
        assert fmap1.dtype in (torch.float16, torch.float32) and all(f.dim() == 5 and f.shape[2] * kp == C and f.shape[4] == kp
                                                                     for f in fmaps2)
    E = coords.shape[0]
    L = len(fmaps2)
    levels = (CorrLevel * L)()
    N2 = fmaps2[0].shape[0]
    for l, f in enumerate(fmaps2):
        assert f.dtype == fmap1.dtype and f.shape[0] == N2
        H2, W2 = ((f.shape[2], f.shape[3]) if layout == RAMP_NCHW else
                  (f.shape[1], f.shape[3]) if layout == RAMP_NHWC32 else (f.shape[1], f.shape[2]))
        levels[l] = CorrLevel(f.data_ptr(), H2, W2, float(coord_divs[l]))
    d = 2 * radius + 1
    dense = d * d * P * P * L
    if row_elems and row_elems != dense:      # padded rows [E, row_elems] (tail zero filled by the kernel)
        assert row_elems > dense
        out = torch.empty((E, row_elems), dtype=fmap1.dtype, device=fmap1.device)
    else:
        row_elems = 0
        out = torch.empty((E, d, d, P, P, L), dtype=fmap1.dtype, device=fmap1.device)
    if order is not None:
        assert order.dtype == torch.int32 and order.is_contiguous() and order.shape[0] == E
    code = dtype_code(fmap1)
    if fast_f32 is None:
        fast_f32 = os.environ.get("RAMP_CORR_F32_MFMA", "0") == "1"
    if fast_f32 and fmap1.dtype == torch.float32 and layout in (RAMP_NHWC, RAMP_NHWC32):
        # opt-in: MFMA accumulation order instead of the reference's fmaf chain.  fast_f32 = 2 with RAMP_NHWC32: the target
        # maps are planes of split fp16 pairs (pyramid_pack(split=True)) -> corr_mfma_kernel<CorrX2>
        code |= _LIB_CORR_X2 if (int(fast_f32) == 2 and layout == RAMP_NHWC32) else _LIB_CORR_MFMA32
    assert not (fmap1.dtype == torch.float32 and layout == RAMP_NHWC32 and not fast_f32), \
        "chunked fp32 target maps are read by the MFMA kernels only (fast_f32)"
    check(lib().ramp_corr_fwd_ordered(ptr(fmap1), levels, L, ptr(coords), ptr(ii), ptr(jj),
                                      ptr(order) if order is not None else None, ptr(out), int(row_elems), int(mod_ii),
                                      int(mod_jj), E,
                                      N1, N2, C, P, radius, code, layout, stream()),
          "ramp_corr_fwd_ordered")
    return out


def event_stack(x, y, p, height, width, num_bins=5, as_float=True):
    """EventToStack_Numpy on the device (reference utils/transformers.py:128-161): integer pixel
    coordinates x, y [N] and polarities p [N] (int8, +-1; 0 is read as -1 like data/events.py:29) ->
    the [num_bins, height, width] stack (float32 values of the int8 stack, or int8 with as_float=False)"""
    require_cuda(x, y, p)
    N = x.shape[0]
    xi, yi = x.to(torch.int32).contiguous(), y.to(torch.int32).contiguous()
    pi = p.to(torch.int8)
    pi = torch.where(pi == 0, torch.full_like(pi, -1), pi).contiguous()
    out = torch.empty((num_bins, height, width), dtype=torch.float32 if as_float else torch.int8, device=x.device)
    nbytes = lib().ramp_event_stack_workspace_bytes(num_bins, height, width)
    ws = _lib_workspace(nbytes, x.device, "evstack")
    check(lib().ramp_event_stack(ptr(xi), ptr(yi), ptr(pi), N, num_bins, height, width,
                                 None if as_float else ptr(out), ptr(out) if as_float else None, ptr(ws), nbytes,
                                 stream()), "ramp_event_stack")
    return out


def depth_median_fill(patches_state, n, F, patches_new):
    """patches_new[:, 2] = median of patches_state[n-F:n, :, 2] (reference Ramp_vo.py:370-371), one launch.
    patches_state [N,M,3,P,P], patches_new [M,3,P,P] (both contiguous fp32)"""
    require_cuda(patches_state, patches_new)
    _, M, _, P, _ = patches_state.shape
    assert patches_state.is_contiguous() and patches_new.is_contiguous() and n - F >= 0
    check(lib().ramp_depth_median_fill(ptr(patches_state[n - F]), F, M, P, ptr(patches_new), stream()),
          "ramp_depth_median_fill")


def depth_median_supported(F, M, P):
    return F * M * P * P <= 8192


def event_topk(events, k, nms_kernel_size=11, want_indices=False, out=None):
    """patch centres of one frame: events [bins,H,W] float32 -> coords [k,2] float32 (x + y/h, y) at the
    top-k cells of the NMS'ed mean |event| map (reference utils.py:186-226), one score kernel + one NMS
    kernel + a one-workgroup radix select"""
    require_cuda(events)
    bins, H, W = events.shape
    events = events.contiguous().float()
    coords = out if out is not None else torch.empty((k, 2), dtype=torch.float32, device=events.device)
    assert coords.shape == (k, 2) and coords.dtype == torch.float32 and coords.is_contiguous()
    idx = torch.empty(k, dtype=torch.int64, device=events.device) if want_indices else None
    nbytes = lib().ramp_event_topk_workspace_bytes(H, W)
    ws = _lib_workspace(nbytes, events.device, "topk")
    check(lib().ramp_event_topk(ptr(events), bins, H, W, int(k), int(nms_kernel_size), ptr(coords),
                                ptr(idx) if idx is not None else None, ptr(ws), nbytes, stream()),
          "ramp_event_topk")
    return (coords, idx) if want_indices else coords


def event_topk_supported(events, k, nms_kernel_size):
    return (events.dim() == 3 and events.shape[2] % 4 == 0 and events.shape[1] >= 4 and k <= 512
            and k <= (events.shape[1] // 4) * (events.shape[2] // 4) and (nms_kernel_size == 0 or (nms_kernel_size % 2 == 1 and nms_kernel_size <= 17)))


def pyramid_pack(fmap, out1=None, out4=None, split=None):
    """NHWC map [H,W,128] -> (level1 [H,4,W,32], level4 [H/4,4,W/4,32]) (fp16) or ([H,8,W,16], [H/4,8,W/4,16]) (fp32) in the
    correlation kernel's packed target layout (RAMP_NHWC32: 64 bytes per pixel and plane); level4 is the 4x4 mean
    (Ramp_vo.py:378-381; fp32: torch's avg_pool2d to the bit).  fp32 with ``split`` (default: corr_f32_mode() == 2): the two
    float32 containers hold split fp16 parts [H,4,2,W,32] (x = hi + lo 2^-11; unpack_split() decodes them) for
    corr(..., fast_f32=2)"""
    require_cuda(fmap)
    H, W, C = fmap.shape
    assert fmap.dtype in (torch.float16, torch.float32) and fmap.is_contiguous()
    if split is None:
        split = _lib.corr_f32_mode() == 2
    split = bool(split) and fmap.dtype == torch.float32
    kp = kplane(fmap.dtype)
    if out1 is None:
        out1 = torch.empty((H, C // kp, W, kp), dtype=fmap.dtype, device=fmap.device)
    if out4 is None:
        out4 = torch.empty((H // 4, C // kp, W // 4, kp), dtype=fmap.dtype, device=fmap.device)
    check(lib().ramp_pyramid_pack(ptr(fmap), ptr(out1), ptr(out4), H, W, C,
                                  dtype_code(fmap) | (_LIB_CORR_X2 if split else 0), stream()), "ramp_pyramid_pack")
    return out1, out4


def pack_split(rows):
    """float32 rows [..., H, W, 128] -> the float32 container [..., H, 8, W, 16] of split fp16 pairs, in torch (the
    arithmetic of csrc/altcorr.hip::corr_split2: load_state_dict, tests); pyramid_pack(split=True) is the kernel"""
    assert rows.dtype == torch.float32 and rows.shape[-1] == 128
    lead, (H, W) = rows.shape[:-3], rows.shape[-3:-1]
    hi = torch.where(rows.abs() < 6.103515625e-5, torch.zeros_like(rows), rows.half().float())
    lo = ((rows - hi) * 2048.0).half()
    v = torch.stack((hi.half(), lo), -2)                                  # [.., H, W, 2, 128]
    n = len(lead)
    v = v.view(*lead, H, W, 2, 4, 32).permute(*range(n), n, n + 3, n + 2, n + 1, n + 4).contiguous()   # [.., H, 4, 2, W, 32]
    return v.view(torch.float32).view(*lead, H, 8, W, 16)


def unpack_split(planes):
    """float32 container [..., H, 8, W, 16] of split fp16 pairs (pyramid_pack(split=True)) -> (hi, lo) fp16 tensors
    [..., H, W, 128]: the value is hi + lo * 2**-11 (tests, debugging)"""
    lead, (H, _, W, _) = planes.shape[:-4], planes.shape[-4:]
    v = planes.contiguous().view(torch.float16).view(*lead, H, 4, 2, W, 32)
    n = len(lead)
    perm = tuple(range(n)) + (n + 2, n, n + 3, n + 1, n + 4)          # [.., 2, H, W, 4, 32]
    v = v.permute(*perm).reshape(*lead, 2, H, W, 128)
    return v.select(n, 0), v.select(n, 1)


def pyramid_pack_supported(H, W, C=128):
    return C == 128 and W % 16 == 0 and H % 4 == 0


# ------------------------------------------------------------------ lietorch
def _flat(x, dim):
    return x.reshape(-1, dim).contiguous().float()


def se3_unary(name, x, din, dout):
    require_cuda(x)
    shp = x.shape[:-1]
    x2 = _flat(x, din)
    out = torch.empty((x2.shape[0], dout), dtype=torch.float32, device=x.device)
    check(getattr(lib(), name)(ptr(x2), ptr(out), x2.shape[0], stream()), name)
    return out.view(shp + (dout,))


def se3_binary(name, x, y, dx, dy, dout):
    require_cuda(x, y)
    bs = torch.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    x2 = x.float().expand(bs + (dx,)).reshape(-1, dx).contiguous()
    y2 = y.float().expand(bs + (dy,)).reshape(-1, dy).contiguous()
    out = torch.empty((x2.shape[0], dout), dtype=torch.float32, device=x.device)
    check(getattr(lib(), name)(ptr(x2), ptr(y2), ptr(out), x2.shape[0], stream()), name)
    return out.view(bs + (dout,))


# ----------------------------------------------------------- projective ops
def _idx(t):
    assert t.dtype == torch.int64
    return t.contiguous()


def transform(poses, patches, intrinsics, ii, jj, kk, tonly=False):
    """Ramp_vo.reproject: poses [..,7], patches [..,3,P,P], intrinsics [..,4] -> [1,E,2,P,P]"""
    require_cuda(poses, patches, intrinsics, ii, jj, kk)
    P = patches.shape[-1]
    f32 = torch.float32
    if not (poses.dtype == f32 and patches.dtype == f32 and intrinsics.dtype == f32 and poses.is_contiguous()
            and patches.is_contiguous() and intrinsics.is_contiguous()):        # (the tracker's buffers are)
        poses = poses.reshape(-1, 7).contiguous().float()
        patches = patches.reshape(-1, 3, P, P).contiguous().float()
        intrinsics = intrinsics.reshape(-1, 4).contiguous().float()
    E = ii.shape[0]
    out = torch.empty((1, E, 2, P, P), dtype=torch.float32, device=poses.device)
    check(lib().ramp_transform(ptr(poses), ptr(patches), ptr(intrinsics), ptr(_idx(ii)),
                               ptr(_idx(jj)), ptr(_idx(kk)), ptr(out), E, P, int(bool(tonly)),
                               stream()), "ramp_transform")
    return out


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """cuda_ba.reproject -> [1,E,2,P,P]"""
    require_cuda(poses, patches, intrinsics, ii, jj, kk)
    P = patches.shape[-1]
    poses = poses.reshape(-1, 7).contiguous().float()
    patches = patches.reshape(-1, 3, P, P).contiguous().float()
    intrinsics = intrinsics.reshape(-1, 4).contiguous().float()
    E = ii.shape[0]
    out = torch.empty((1, E, 2, P, P), dtype=torch.float32, device=poses.device)
    check(lib().ramp_reproject(ptr(poses), ptr(patches), ptr(intrinsics), ptr(_idx(ii)),
                               ptr(_idx(jj)), ptr(_idx(kk)), ptr(out), E, P, stream()),
          "ramp_reproject")
    return out


def point_cloud(poses, patches, intrinsics, ix):
    """3-D point of every patch centre: patches [m,3,P,P] (or [1,m,..]), ix [m] -> [m,3]"""
    require_cuda(poses, patches, intrinsics, ix)
    P = patches.shape[-1]
    poses = poses.reshape(-1, 7).contiguous().float()
    patches = patches.reshape(-1, 3, P, P).contiguous().float()
    intrinsics = intrinsics.reshape(-1, 4).contiguous().float()
    m = ix.shape[0]
    assert patches.shape[0] >= m
    out = torch.empty((m, 3), dtype=torch.float32, device=poses.device)
    check(lib().ramp_point_cloud(ptr(poses), ptr(patches), ptr(intrinsics), ptr(_idx(ix)),
                                 ptr(out), m, P, stream()), "ramp_point_cloud")
    return out


def motionmag(poses, patches, intrinsics, ii, jj, kk, pair_groups, key_ij, key_ji, beta=0.5):
    """[mean flow i->j, mean flow j->i] (device tensor [2]); pair_groups: the (ii, jj) grouping"""
    require_cuda(poses, patches, intrinsics, ii, jj, kk)
    P = patches.shape[-1]
    poses = poses.reshape(-1, 7).contiguous().float()
    patches = patches.reshape(-1, 3, P, P).contiguous().float()
    intrinsics = intrinsics.reshape(-1, 4).contiguous().float()
    out = torch.empty(2, dtype=torch.float32, device=poses.device)
    g = pair_groups
    check(lib().ramp_motionmag(ptr(poses), ptr(patches), ptr(intrinsics), ptr(_idx(ii)), ptr(_idx(jj)),
                               ptr(_idx(kk)), ptr(g.order), ptr(g.seg_start), ptr(g.ukeys), ptr(g.ngroups),
                               int(key_ij), int(key_ji), float(beta), ptr(out), P, stream()), "ramp_motionmag")
    return out


def multi_copy(pairs):
    """pairs: list of (src_tensor, dst_tensor) with equal byte sizes, both contiguous; one launch"""
    n = len(pairs)
    src = (ctypes.c_void_p * n)(*[s.data_ptr() for s, _ in pairs])
    dst = (ctypes.c_void_p * n)(*[d.data_ptr() for _, d in pairs])
    nbytes = (ctypes.c_long * n)(*[s.numel() * s.element_size() for s, _ in pairs])
    for s, d in pairs:
        assert s.is_contiguous() and d.is_contiguous() and s.numel() * s.element_size() == d.numel() * d.element_size()
    check(lib().ramp_multi_copy(src, dst, nbytes, n, stream()), "ramp_multi_copy")


def store_rows(srcs, rows):
    """one launch: contiguous tensor srcs[i] -> row rows[i][1] of the contiguous buffer rows[i][0] (no view tensors)"""
    n = len(srcs)
    rb = [b.stride(0) * b.element_size() for b, _ in rows]
    for s_, (b, _), nb in zip(srcs, rows, rb):
        assert s_.numel() * s_.element_size() == nb and s_.is_contiguous() and b.is_contiguous()
    src = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in srcs])
    dst = (ctypes.c_void_p * n)(*[b.data_ptr() + int(r) * nb for (b, r), nb in zip(rows, rb)])
    check(lib().ramp_multi_copy(src, dst, (ctypes.c_long * n)(*rb), n, stream()), "ramp_multi_copy")


def depth_median(patches_state, n, frames, out):
    """median depth of patches_state[n - frames : n] into the device scalar ``out`` (ramp_depth_median)"""
    _, M, _, P, _ = patches_state.shape
    check(lib().ramp_depth_median(ptr(patches_state[n - frames]), int(frames), M, P, ptr(out), stream()),
          "ramp_depth_median")


def frame_commit(poses, n, motion, damping, tstamps, counter, index_map, index_val, intrinsics, copy_k, patches_state,
                 median_frames, patches_new, srcs, rows, median_dev=None):
    """ramp_frame_commit: frame_begin + depth_median_fill + the state stores of one frame in one launch;
    srcs[i] -> row rows[i][1] of buffer rows[i][0] (like store_rows), patches_new -> patches_state[n];
    median_dev: the median of the last ``median_frames`` frames if the caller computed it ahead (depth_median)"""
    n_copy = len(srcs)
    rb = [b.stride(0) * b.element_size() for b, _ in rows]
    for s_, (b, _), nb in zip(srcs, rows, rb):
        assert s_.numel() * s_.element_size() == nb and s_.is_contiguous() and b.is_contiguous()
    src = (ctypes.c_void_p * n_copy)(*[s_.data_ptr() for s_ in srcs])
    dst = (ctypes.c_void_p * n_copy)(*[b.data_ptr() + int(r) * nb for (b, r), nb in zip(rows, rb)])
    _, M, _, P, _ = patches_state.shape
    check(lib().ramp_frame_commit(ptr(poses), int(n), int(motion), float(damping), ptr(tstamps), int(counter),
                                  ptr(index_map), int(index_val), ptr(intrinsics), int(bool(copy_k)), ptr(patches_state),
                                  int(median_frames), M, P, ptr(patches_new), n_copy, src, dst,
                                  (ctypes.c_long * n_copy)(*rb), ptr(median_dev), stream()), "ramp_frame_commit")


class FrameCommitPlan:
    """ramp_frame_commit for a fixed set of sources / destination buffers: the descriptor arrays are built once, a
    call only fills in the destination rows (this launch sits between the keyframe read-back and the correlation
    kernel, where the host is the limiter)"""

    def __init__(self, srcs, bufs, patches_state):
        n = self.n_copy = len(srcs)
        self.keep = (list(srcs), list(bufs), patches_state)
        self.rb = [b.stride(0) * b.element_size() for b in bufs]
        for s_, b, nb in zip(srcs, bufs, self.rb):
            assert s_.numel() * s_.element_size() == nb and s_.is_contiguous() and b.is_contiguous()
        self.src = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in srcs])
        self.dst = (ctypes.c_void_p * n)()
        self.base = [b.data_ptr() for b in bufs]
        self.rbc = (ctypes.c_long * n)(*self.rb)
        _, self.M, _, self.P, _ = patches_state.shape
        self.src_ptrs = tuple(s_.data_ptr() for s_ in srcs)

    def run(self, poses, n, motion, damping, tstamps, counter, index_map, index_val, intrinsics, copy_k, patches_state,
            median_frames, patches_new, rows, median_dev=None):
        for i in range(self.n_copy):
            self.dst[i] = self.base[i] + int(rows[i]) * self.rb[i]
        check(lib().ramp_frame_commit(ptr(poses), int(n), int(motion), float(damping), ptr(tstamps), int(counter),
                                      ptr(index_map), int(index_val), ptr(intrinsics), int(bool(copy_k)),
                                      ptr(patches_state), int(median_frames), self.M, self.P, ptr(patches_new),
                                      self.n_copy, self.src, self.dst, self.rbc, ptr(median_dev), stream()),
              "ramp_frame_commit")


class ShiftPlan:
    """descriptor arrays of shift_rows for a fixed set of buffers (built once)"""

    def __init__(self, bufs):
        n = self.n = len(bufs)
        for t, _ in bufs:
            assert t.is_contiguous()
        self.keep = [t for t, _ in bufs]
        self.base = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in bufs])
        self.rb = (ctypes.c_long * n)(*[t[0].numel() * t.element_size() for t, _ in bufs])
        self.mod = (ctypes.c_int * n)(*[int(m) for _, m in bufs])

    def run(self, k, nrows):
        check(lib().ramp_shift_rows(self.base, self.rb, self.mod, self.n, int(k), int(nrows), stream()),
              "ramp_shift_rows")


def shift_rows(bufs, k, nrows):
    """bufs: list of (tensor [rows, ...], ring_modulus or 0); rows k+1..nrows-1 move down by one"""
    n = len(bufs)
    base = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in bufs])
    rb = (ctypes.c_long * n)(*[t[0].numel() * t.element_size() for t, _ in bufs])
    mod = (ctypes.c_int * n)(*[int(m) for _, m in bufs])
    for t, _ in bufs:
        assert t.is_contiguous()
    check(lib().ramp_shift_rows(base, rb, mod, n, int(k), int(nrows), stream()), "ramp_shift_rows")


def motion_model(poses, n, damping):
    """in place: poses[n] = Exp(damping * Log(poses[n-1] * poses[n-2]^-1)) * poses[n-1]"""
    require_cuda(poses)
    assert poses.dtype == torch.float32 and poses.is_contiguous()
    check(lib().ramp_motion_model(ptr(poses), int(n), float(damping), stream()), "ramp_motion_model")


def frame_begin(poses, n, motion, damping, tstamps, counter, index_map, index_val, intrinsics, copy_k):
    """one launch for the per-frame bookkeeping (reference Ramp_vo.py:345-363); see include/ramp_hip.h"""
    require_cuda(poses, tstamps, index_map, intrinsics)
    assert poses.dtype == torch.float32 and poses.is_contiguous() and intrinsics.is_contiguous()
    assert tstamps.dtype == torch.long and index_map.dtype == torch.long
    check(lib().ramp_frame_begin(ptr(poses), int(n), int(motion), float(damping), ptr(tstamps), int(counter),
                                 ptr(index_map), int(index_val), ptr(intrinsics), int(bool(copy_k)), stream()),
          "ramp_frame_begin")


# --------------------------------------------------------------------- graph
class Groups:
    """result of group_by: order/gid/seg_start/ukeys/ngroups (device tensors)"""
    __slots__ = ("order", "gid", "seg_start", "ukeys", "ngroups", "E")


def group_by(keys, key_bound=0):
    require_cuda(keys)
    keys = _idx(keys)
    E = keys.shape[0]
    dev = keys.device
    g = Groups()
    g.E = E
    g.order = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    g.gid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    g.seg_start = torch.empty(E + 1, dtype=torch.int32, device=dev)
    g.ukeys = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    g.ngroups = torch.empty(1, dtype=torch.int32, device=dev)
    nbytes = lib().ramp_group_by_workspace_bytes(E)
    ws = _lib.workspace(nbytes, dev, "graph")
    check(lib().ramp_group_by(ptr(keys), E, int(key_bound), ptr(g.order), ptr(g.gid),
                              ptr(g.seg_start), ptr(g.ukeys), ptr(g.ngroups), ptr(ws),
                              ws.numel(), stream()), "ramp_group_by")
    return g


def group_by_small(a, b, mul, sub, K, max_groups=0):
    """group_by for keys a*mul + b - sub known to lie in [0, K) (counting sort, 5 short kernels)"""
    require_cuda(a)
    a = _idx(a)
    b = _idx(b) if b is not None else None
    E = a.shape[0]
    dev = a.device
    g = Groups()
    g.E = E
    g.order = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    g.gid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    g.seg_start = torch.empty(E + 1, dtype=torch.int32, device=dev)
    g.ukeys = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    g.ngroups = torch.empty(1, dtype=torch.int32, device=dev)
    nbytes = lib().ramp_group_by_small_workspace_bytes(E, int(K))
    ws = _lib.workspace(nbytes, dev, "graph")
    check(lib().ramp_group_by_small(ptr(a), ptr(b), int(mul), int(sub), int(K), E, ptr(g.order), ptr(g.gid),
                                    ptr(g.seg_start), ptr(g.ukeys), ptr(g.ngroups), int(max_groups), ptr(ws),
                                    ws.numel(), stream()), "ramp_group_by_small")
    return g


def neighbors_from_groups(groups, jj, max_groups, want_kj=False):
    """cuda_ba.neighbors(kk, jj) from the per-kk groups (no extra sort); want_kj: also the factors in (kk, jj) order"""
    require_cuda(jj)
    jj = _idx(jj)
    E = jj.shape[0]
    ix = torch.empty(E, dtype=torch.int64, device=jj.device)
    jx = torch.empty(E, dtype=torch.int64, device=jj.device)
    kj = torch.empty(E, dtype=torch.int32, device=jj.device) if want_kj else None
    check(lib().ramp_neighbors_from_groups(ptr(groups.order), ptr(groups.seg_start), ptr(groups.ngroups), ptr(jj),
                                           ptr(ix), ptr(jx), ptr(kj), E, int(max_groups), stream()),
          "ramp_neighbors_from_groups")
    return (ix, jx, kj) if want_kj else (ix, jx)


def neighbors(kk, jj, kk_bound=0, jj_bound=0):
    require_cuda(kk, jj)
    kk, jj = _idx(kk), _idx(jj)
    E = kk.shape[0]
    ix = torch.empty(E, dtype=torch.int64, device=kk.device)
    jx = torch.empty(E, dtype=torch.int64, device=kk.device)
    if E == 0:
        return ix, jx
    nbytes = lib().ramp_neighbors_workspace_bytes(E)
    ws = _lib.workspace(nbytes, kk.device, "graph")
    check(lib().ramp_neighbors(ptr(kk), ptr(jj), ptr(ix), ptr(jx), E, int(kk_bound),
                               int(jj_bound), ptr(ws), ws.numel(), stream()), "ramp_neighbors")
    return ix, jx


def segment_softmax_sum(fx, gx, groups, max_groups):
    """y[g] = sum_{e in g} softmax_e(gx[e]) * fx[e];  fx,gx [E,C] -> y [max_groups,C]"""
    require_cuda(fx, gx)
    fx = fx.contiguous()
    gx = gx.contiguous()
    assert fx.dtype == gx.dtype and fx.shape == gx.shape
    E, C = fx.shape
    y = torch.zeros((max_groups, C), dtype=fx.dtype, device=fx.device)
    check(lib().ramp_segment_softmax_sum(ptr(fx), ptr(gx), ptr(groups.order),
                                         ptr(groups.seg_start), ptr(groups.ngroups), ptr(y), E, C,
                                         int(max_groups), dtype_code(fx), stream()),
          "ramp_segment_softmax_sum")
    return y


# -------------------------------------------------------------------- fastba
def ba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2,
       info=None, plan=None):
    """in-place bundle adjustment (cuda_ba.forward).  poses [..,7] and patches
    [..,3,P,P] must be contiguous float32 views of the caller's storage."""
    require_cuda(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk)
    for t in (poses, patches):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("BA mutates poses/patches in place: contiguous float32 required")
    P = patches.shape[-1]
    n_poses = poses.numel() // 7
    n_patches = patches.numel() // (3 * P * P)
    intrinsics = intrinsics.reshape(-1, 4).contiguous().float()
    target = target.reshape(-1, 2).contiguous().float()
    weight = weight.reshape(-1, 2).contiguous().float()
    lmbda = lmbda.reshape(-1).contiguous().float()
    E = ii.shape[0]
    assert target.shape[0] == E and weight.shape[0] == E
    if plan is not None:
        # groupings shared with the update operator (GraphPlan): no sort inside BA
        gk, gp = plan.g_kk, plan.g_ij
        mk, mp = max(int(plan.max_kk), 1), max(int(plan.max_ij), 1)
        nbytes = lib().ramp_ba_planned_workspace_bytes(E, n_poses, n_patches, int(t0), int(t1), mk, mp)
        ws = _lib.workspace(nbytes, poses.device, "ba")
        check(lib().ramp_ba_forward_planned(ptr(poses), ptr(patches), ptr(intrinsics), ptr(target), ptr(weight),
                                            ptr(lmbda), ptr(_idx(ii)), ptr(_idx(jj)), ptr(_idx(kk)), E, P, n_poses,
                                            n_patches, int(t0), int(t1), int(iterations), ptr(gk.order),
                                            ptr(gk.seg_start), ptr(gk.ngroups), ptr(gk.ukeys), mk, ptr(gp.order),
                                            ptr(gp.seg_start), ptr(gp.ngroups), mp, ptr(ws), ws.numel(), ptr(info),
                                            stream()), "ramp_ba_forward_planned")
        return
    nbytes = lib().ramp_ba_workspace_bytes(E, n_poses, n_patches, int(t0), int(t1))
    ws = _lib.workspace(nbytes, poses.device, "ba")
    check(lib().ramp_ba_forward(ptr(poses), ptr(patches), ptr(intrinsics), ptr(target),
                                ptr(weight), ptr(lmbda), ptr(_idx(ii)), ptr(_idx(jj)),
                                ptr(_idx(kk)), E, P, n_poses, n_patches, int(t0), int(t1),
                                int(iterations), ptr(ws), ws.numel(), ptr(info), stream()),
          "ramp_ba_forward")
