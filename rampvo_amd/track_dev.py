"""Device-resident steady state of the tracker (csrc/track.hip, include/ramp_hip.h ``ramp_track_step``).

``DeviceTrack`` owns the capacity-sized buffers of one ``Ramp_vo`` instance and the ``ramp_track`` descriptor that
points at them and at the tracker's own state tensors.  Between ``enter()`` and ``leave()`` the factor graph, the
keyframe count and the hidden state live on the GPU only: ``step()`` enqueues a whole frame (reference
ramp/Ramp_vo.py:327-410: frame stores, update(), keyframe(), the next frame's append_factors) in one C call and never
reads the device; ``leave()`` is the one synchronisation that hands the state back to the host mirror.
"""
import collections
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import c_i, c_p, c_sz

DYN_WORDS = 32
(DYN_N, DYN_NROW, DYN_E, DYN_KLO, DYN_FLO, DYN_W, DYN_REMOVED, DYN_K, DYN_NPREV, DYN_EPREV, DYN_EKEPT, DYN_STATUS,
 DYN_NLOG, DYN_FRAME) = range(14)
DYN_FRAME2 = 31          # repeats DYN_FRAME in the other 64-byte half of the block (a torn lazy copy shows)
LOG_WORDS = 12
MAX_AHEAD = 6            # frames the host may enqueue ahead of the newest lazy copy of the sizes (Ramp_vo._track_device)
COMMIT, UPDATE, KEYFRAME, MM_GIVEN, WRAP_COORDS, COMPACT_COORDS, UPDATE_PRE, UPDATE_POST = 1, 2, 4, 8, 16, 32, 64, 128
CORR_ROW = 896


def _ptr_fields(names):
    return [(n, c_p) for n in names]


class TrackWeights(ctypes.Structure):
    _fields_ = (_ptr_fields(["corr_w1", "corr_w2", "corr_w3", "corr_b1", "corr_b2", "corr_b3", "corr_ln_w", "corr_ln_b",
                             "norm_w", "norm_b", "c1_wa", "c1_wb", "c2_wa", "c2_wb", "c1_ba", "c1_bb", "c2_ba", "c2_bb",
                             "kk_wf", "kk_wg", "kk_wh", "ij_wf", "ij_wg", "ij_wh", "kk_bf", "kk_bg", "kk_bh", "ij_bf",
                             "ij_bg", "ij_bh", "ln1_w", "ln1_b", "ln2_w", "ln2_b"])
                + [("gru_w", c_p * 6), ("gru_b", c_p * 6), ("heads_w", c_p), ("heads_b", c_p)]
                + [(n, ctypes.c_float) for n in ("corr_ln_eps", "norm_eps", "ln1_eps", "ln2_eps")])


class Track(ctypes.Structure):
    _fields_ = ([(n, c_i) for n in ("M", "P", "mem", "n_rows", "patch_lifetime", "removal_window", "opt_window",
                                    "keyframe_index", "motion_model", "feat_h", "feat_w", "E_cap", "kk_cap", "ij_cap",
                                    "kkey_cap", "pkey_cap", "log_cap", "m_cap")]
                + [("motion_damping", ctypes.c_float), ("pad0", ctypes.c_float), ("keyframe_thresh", ctypes.c_double)]
                + _ptr_fields(["dyn", "poses", "patches", "intrinsics", "points", "tstamps", "index_map", "ixm", "colors",
                               "imap", "gmap", "fmap1", "fmap2", "lmbda", "fe_colors", "fe_imap", "fe_gmap", "fe_fmap1",
                               "fe_fmap2", "fe_patches"])
                + [("graph", c_p * 2)]
                + _ptr_fields(["kk_order", "kk_gid", "kk_seg", "kk_ngroups", "ij_order", "ij_gid", "ij_seg", "ij_ngroups",
                               "kk_ukeys", "ij_ukeys", "ix", "jx", "kj", "plan_ws"])
                + [("plan_ws_bytes", c_sz), ("w", TrackWeights)]
                + _ptr_fields(["coords", "corr"]) + [("net", c_p * 3)]
                + _ptr_fields(["fg", "ykk", "hkk", "yij", "hij", "relu_t", "sagg_frag", "target", "weight", "ba_ws"])
                + [("ba_ws_bytes", c_sz)]
                + _ptr_fields(["mm", "median", "dlog", "edit_ws", "dyn_host", "dyn_host_dev"]) + [("probe", c_p * 5), ("E_hint", c_i),
                   ("gate_seq", ctypes.c_uint32), ("feat_fp32", c_i), ("feat_plain", c_i), ("gate_flag", c_p), ("fmap1_slot", c_p)])


_E_EST_LAST = 1          # (0: the largest of the last 64 copies, round 3's first rule)
_E_EST_MARGIN = 0


class Signal:
    """a 32-bit word in signal memory (csrc/track.hip::ramp_signal_alloc): a kernel stores a sequence number, another
    stream waits for it with one sleeping wave -- a cross-stream "go" without a packet on the producer's stream.  ptr is
    None where the allocation is not supported (the caller keeps its event then)."""

    def __init__(self):
        p = ctypes.c_void_p()
        rc = _lib.lib().ramp_signal_alloc(ctypes.byref(p))
        self.ptr = p if rc == 0 and p.value else None

    def wait(self, stream, value, timeout_us=50000, then_delay_us=0, status=None):
        """status: optional device int32 word; bit 128 is ORed in if the wait gives up (the consumer then ran without the
        order it asked for -- the tracker raises when it reads the bit)"""
        _lib.check(_lib.lib().ramp_stream_wait_flag(ctypes.c_void_p(stream.cuda_stream), self.ptr, int(value) & 0xFFFFFFFF,
                                                    int(timeout_us), int(then_delay_us),
                                                    ctypes.c_void_p(status) if status else None), "ramp_stream_wait_flag")

    def __del__(self):
        try:
            if self.ptr is not None:
                _lib.lib().ramp_signal_free(self.ptr)
        except Exception:
            pass


def supported(slam):
    """the configurations the device-resident step takes: fp16 or fp32 features (update operator: csrc/update_mlp.hip /
    csrc/update_x3.hip's fused chains), P = 3, DIM = 384, any PATCHES_PER_FRAME up to 303, DAMPED_LINEAR motion model, an
    optimisation window of at most 32 poses"""
    cfg = slam.cfg
    # (the chunked pyramid layout, or plain NHWC planes where the feature plane's shape does not fit it)
    return (slam.dtype in (torch.half, torch.float) and slam._lazy_net and slam.P == 3 and slam.DIM == 384
            and 3 * slam.M * 9 <= 8192 and cfg.MOTION_MODEL in ("DAMPED_LINEAR",)
            and cfg.PATCH_LIFETIME <= cfg.REMOVAL_WINDOW + 1 and cfg.KEYFRAME_INDEX >= 2
            and 6 * cfg.OPTIMIZATION_WINDOW <= 192)


def unsupported_reason(slam):
    """which of supported()'s conditions fails (for the one-time warning of Ramp_vo._enter_device)"""
    cfg = slam.cfg
    checks = [(slam.P == 3 and slam.DIM == 384, "patch size / feature width other than 3 / 384"),
              (3 * slam.M * 9 <= 8192, "PATCHES_PER_FRAME above 303 (the depth median of three frames is one workgroup's)"),
              (cfg.MOTION_MODEL in ("DAMPED_LINEAR",), "MOTION_MODEL other than DAMPED_LINEAR"),
              (cfg.PATCH_LIFETIME <= cfg.REMOVAL_WINDOW + 1, "PATCH_LIFETIME exceeds REMOVAL_WINDOW + 1"),
              (cfg.KEYFRAME_INDEX >= 2, "KEYFRAME_INDEX below 2"),
              (6 * cfg.OPTIMIZATION_WINDOW <= 192, "OPTIMIZATION_WINDOW above 32")]
    return "; ".join(msg for ok, msg in checks if not ok) or "unknown"


class DeviceTrack:
    def __init__(self, slam):
        assert supported(slam)
        lib = _lib.lib()
        cfg, dev = slam.cfg, slam.device
        M, r, R = slam.M, cfg.PATCH_LIFETIME, cfg.REMOVAL_WINDOW
        self.slam = slam
        self.E_cap = E_cap = (R + 2) * (2 * r - 1) * M
        self.kk_cap = kk_cap = (R + 2) * M
        self.ij_cap = ij_cap = min((R + 2) * (2 * r), (R + r + 1) ** 2)
        self.kkey_cap = (R + 2) * M
        self.pkey_cap = (R + r + 1) ** 2
        self.log_cap = 4096
        self.new_cap = (2 * r - 1) * M
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        i32, i64, f32, f16 = torch.int32, torch.int64, torch.float32, torch.float16
        self.dyn = z(DYN_WORDS, i32)
        self.dyn_host = torch.zeros(DYN_WORDS, dtype=i32).pin_memory()
        self.graph = [z((4, E_cap), i64), z((4, E_cap), i64)]
        self.kk = dict(order=z(E_cap, i32), gid=z(E_cap, i32), seg=z(kk_cap + 2, i32), ngroups=z(1, i32),
                       ukeys=z(kk_cap + 2, i64))
        self.ij = dict(order=z(E_cap, i32), gid=z(E_cap, i32), seg=z(ij_cap + 2, i32), ngroups=z(1, i32),
                       ukeys=z(ij_cap + 2, i64))
        self.ix, self.jx, self.kj = z(E_cap, i64), z(E_cap, i64), z(E_cap, i32)
        # (zeroed: the plan's histograms are cleared by each plan at its end instead of by a memset launch at its start)
        self.plan_ws = z(lib.ramp_track_plan_workspace_bytes(E_cap, self.kkey_cap, self.pkey_cap), torch.uint8)
        self.fp32 = slam.dtype == torch.float
        # fp32 features: the operator's chains are csrc/update_x3.hip's (one C call per frame, as on the fp16 path) unless
        # RAMP_X3=0 (library GEMMs issued by the host between the two halves of the step)
        self.x3 = self.fp32 and slam.network.update.fused(torch.float32).use_x3
        self.coords = e((E_cap, 2, 3, 3), f32)
        self.corr = e((E_cap, CORR_ROW), f32 if self.fp32 else f16)
        self.net = [z((E_cap, 384), f32) for _ in range(3)]
        tt = f32 if self.fp32 else f16                 # SoftAgg rows / tables
        self.fg = e((16 if (self.fp32 and not self.x3) else E_cap, 768), tt)
        self.ykk, self.hkk = z((kk_cap, 384), tt), z((kk_cap, 384), tt)
        self.yij, self.hij = z((ij_cap, 384), tt), z((ij_cap, 384), tt)
        self.relu_t = e((16 if self.fp32 else E_cap, 384), f16)
        # fragment table of the fused SoftAgg (csrc/update_mlp.hip::upd_softagg_kernel): (m, z, a)[384] per run
        self.sagg_frag = e((16 if self.fp32 else lib.ramp_upd_softagg_frag_rows(E_cap, max(kk_cap, ij_cap)), 3, 384), f32)
        self.target, self.weight = z((E_cap, 2), f32), z((E_cap, 2), f32)
        self.ba_ws = e(lib.ramp_track_ba_workspace_bytes(E_cap, slam.N, M, cfg.OPTIMIZATION_WINDOW, kk_cap, ij_cap),
                       torch.uint8)
        self.mm = z(2, f32)
        self.median = z(1, f32)
        self.sink = z(1, i32)
        self.dlog = z((self.log_cap, LOG_WORDS), f32)
        self.edit_ws = z(3 * ((E_cap + 255) // 256) + 8, i32)          # (sized for edit workgroups of 256 factors; the build uses 1024)
        self.ixm = (torch.arange(slam.N * M, device=dev) // M).contiguous()
        self.k_new = z(4, f32)
        self.cur = 0
        self.active = False
        self._e_seen = collections.deque(maxlen=64)
        self._wkey = None
        self._fe_key = None
        self._keep = []
        t = self.t = Track()
        assert ctypes.sizeof(Track) == lib.ramp_track_sizeof(), "ramp_track mirror out of date"
        # ring slots: [mem, h, 4, w, 32] (fp16) / [mem, h, 8, w, 16] (fp32) chunked, or [mem, h, w, 128]
        h, w = slam.fmap1_.shape[1], (slam.fmap1_.shape[3] if slam._chunked else slam.fmap1_.shape[2])
        for name, val in dict(M=M, P=slam.P, mem=slam.mem, n_rows=slam.N, patch_lifetime=r, removal_window=R,
                              opt_window=cfg.OPTIMIZATION_WINDOW, keyframe_index=cfg.KEYFRAME_INDEX, motion_model=1,
                              feat_h=h, feat_w=w, E_cap=E_cap, kk_cap=kk_cap, ij_cap=ij_cap, kkey_cap=self.kkey_cap,
                              pkey_cap=self.pkey_cap, log_cap=self.log_cap, m_cap=slam.N * M).items():
            setattr(t, name, int(val))
        t.motion_damping = float(cfg.MOTION_DAMPING)
        t.keyframe_thresh = float(cfg.KEYFRAME_THRESH)
        t.feat_fp32 = (2 if slam._split else 1) if self.fp32 else 0      # (2: planes of split fp16 pairs, RAMP_CORR_X2)
        t.feat_plain = 0 if slam._chunked else 1
        P = lambda x: x.data_ptr()
        for name, ten in dict(dyn=self.dyn, poses=slam.poses_, patches=slam.patches_, intrinsics=slam.intrinsics_,
                              points=slam.points_, tstamps=slam.tstamps_, index_map=slam.index_map_, ixm=self.ixm,
                              colors=slam.colors_, imap=slam.imap_, gmap=slam.gmap_, fmap1=slam.fmap1_, fmap2=slam.fmap2_,
                              lmbda=slam.lmbda, kk_order=self.kk["order"], kk_gid=self.kk["gid"], kk_seg=self.kk["seg"],
                              kk_ngroups=self.kk["ngroups"], kk_ukeys=self.kk["ukeys"], ij_order=self.ij["order"],
                              ij_gid=self.ij["gid"], ij_seg=self.ij["seg"], ij_ngroups=self.ij["ngroups"],
                              ij_ukeys=self.ij["ukeys"], ix=self.ix, jx=self.jx, kj=self.kj, plan_ws=self.plan_ws,
                              coords=self.coords, corr=self.corr, fg=self.fg, ykk=self.ykk, hkk=self.hkk, yij=self.yij,
                              hij=self.hij, relu_t=self.relu_t, sagg_frag=self.sagg_frag, target=self.target, weight=self.weight,
                              ba_ws=self.ba_ws, mm=self.mm, dlog=self.dlog, edit_ws=self.edit_ws,
                              dyn_host=self.dyn_host).items():
            setattr(t, name, P(ten))
        t.plan_ws_bytes, t.ba_ws_bytes = self.plan_ws.numel(), self.ba_ws.numel()
        # level-0 correlation planes by slot table (csrc/track.hip: a dropped keyframe rotates table entries instead of moving
        # three planes; leave() undoes the permutation).  RAMP_SLOT_TABLE=0: rows are slots, physical shifts
        self.fmap1_slot = None
        if os.environ.get("RAMP_SLOT_TABLE", "1") != "0":
            self.fmap1_slot = torch.arange(slam.mem, dtype=torch.int32, device=dev)
            t.fmap1_slot = self.fmap1_slot.data_ptr()
        dp = ctypes.c_void_p()                    # device address of the pinned copy: resolved once, not per frame
        if lib.ramp_host_device_pointer(ctypes.c_void_p(self.dyn_host.data_ptr()), ctypes.byref(dp)) == 0 and dp.value:
            t.dyn_host_dev = dp.value
        self.status_ptr = self.dyn.data_ptr() + 4 * DYN_STATUS
        t.median = P(self.median)                 # (the depth median of the next frame: computed beside the motion test)
        t.graph[0], t.graph[1] = P(self.graph[0]), P(self.graph[1])
        for i in range(3):
            t.net[i] = P(self.net[i])

    # ------------------------------------------------------------------ weights / front-end outputs
    def bind_weights(self, fu):
        w = fu.weights()
        if self._wkey == fu._key:
            return
        tw = self.t.w
        P = lambda x: x.data_ptr()
        tw.corr_w1, tw.corr_b1 = map(P, w["corr1_pack"])
        tw.corr_w2, tw.corr_b2, tw.corr_w3, tw.corr_b3 = map(P, w["tail_pack"])
        tw.corr_ln_w, tw.corr_ln_b, tw.corr_ln_eps = P(w["corr_ln"][0]), P(w["corr_ln"][1]), float(w["corr_ln"][2])
        tw.norm_w, tw.norm_b, tw.norm_eps = P(w["norm"][0]), P(w["norm"][1]), float(w["norm"][2])
        tw.c1_wa, tw.c1_ba, tw.c1_wb, tw.c1_bb = map(P, w["c1_pack"])
        tw.c2_wa, tw.c2_ba, tw.c2_wb, tw.c2_bb = map(P, w["c2_pack"])
        tw.kk_wf, tw.kk_bf, tw.kk_wg, tw.kk_bg = map(P, w["kk_fg_pack"])
        tw.ij_wf, tw.ij_bf, tw.ij_wg, tw.ij_bg = map(P, w["ij_fg_pack"])
        tw.kk_wh, tw.kk_bh = map(P, w["kk_h_pack"])
        tw.ij_wh, tw.ij_bh = map(P, w["ij_h_pack"])
        tw.ln1_w, tw.ln1_b, tw.ln1_eps = P(w["ln1"][0]), P(w["ln1"][1]), float(w["ln1"][2])
        tw.ln2_w, tw.ln2_b, tw.ln2_eps = P(w["ln2"][0]), P(w["ln2"][1]), float(w["ln2"][2])
        wp, bs = w["gru_pack"][0], w["gru_pack"][1]
        for i in range(6):
            tw.gru_w[i], tw.gru_b[i] = P(wp[i]), P(bs[i])
        tw.heads_w, tw.heads_b = map(P, w["heads_pack"])
        self._wkey = fu._key

    def bind_front_end(self, ex, patches):
        """the static output buffers of the front-end graph (the same objects every frame)"""
        key = (id(ex), patches.data_ptr())
        if self._fe_key == key:
            return True
        ok = (ex is not None and ex["fmap"].dtype == (torch.float32 if self.fp32 else torch.float16)
              and bool(ex["chunked"]) == bool(self.slam._chunked) and patches.is_contiguous()
              and patches.dtype == torch.float32
              and all(ex[k].data_ptr() % 16 == 0 and ex[k].is_contiguous()
                      for k in ("colors", "imap", "gmap", "fmap", "fmap2")))
        if not ok:
            return False
        t = self.t
        t.fe_colors, t.fe_imap, t.fe_gmap = ex["colors"].data_ptr(), ex["imap"].data_ptr(), ex["gmap"].data_ptr()
        t.fe_fmap1, t.fe_fmap2, t.fe_patches = ex["fmap"].data_ptr(), ex["fmap2"].data_ptr(), patches.data_ptr()
        self._fe_key = key
        self._keep = [ex, patches]
        return True

    # ------------------------------------------------------------------ host <-> device hand-over
    def enter(self, ii, jj, kk, rows, new_edges, net_buf, n):
        """host state after keyframe() -> device: kept factors (host arrays, ``rows`` = their hidden-state rows in
        ``net_buf``) followed by the factors the next frame adds; n = Ramp_vo.n"""
        slam = self.slam
        e_ii, e_jj, e_kk = new_edges
        Ek, ne = len(ii), len(e_kk)
        E = Ek + ne
        if E > self.E_cap or net_buf.shape[0] > self.E_cap:
            return False
        g = self.pinned_graph()
        gn = g.numpy()
        for row, a, b, fill in ((0, ii, e_ii, None), (1, jj, e_jj, None), (2, kk, e_kk, None), (3, rows, None, -1)):
            gn[row, :Ek] = a
            gn[row, Ek:E] = b if b is not None else fill
        self.cur = 0
        self._lazy_ok = None                    # (a consistent copy of the sizes from a previous residency is not this one's)
        self.graph[0][:, :E].copy_(g[:, :E], non_blocking=True)
        if net_buf.data_ptr() != self.net[0].data_ptr():
            self.net[0][:net_buf.shape[0]].copy_(net_buf)
        M = slam.M
        k_lo = int(gn[2, :E].min())
        f_lo = int(min(gn[0, :E].min(), gn[1, :E].min()))
        d = self.dyn_host.numpy()
        d[:] = 0
        d[DYN_N], d[DYN_NROW], d[DYN_E] = n + 1, n, E
        d[DYN_KLO], d[DYN_FLO], d[DYN_W] = (k_lo // M) * M, f_lo, n + 1 - f_lo
        d[DYN_NPREV], d[DYN_EPREV], d[DYN_EKEPT] = n, net_buf.shape[0], Ek
        d[DYN_FRAME] = d[DYN_FRAME2] = slam.counter - 1
        if (n + 1) * M - d[DYN_KLO] > self.kkey_cap or d[DYN_W] ** 2 > self.pkey_cap:
            return False
        self.dyn.copy_(self.dyn_host, non_blocking=True)
        if self.fmap1_slot is not None:
            self.fmap1_slot.copy_(torch.arange(slam.mem, dtype=torch.int32, device=self.fmap1_slot.device))   # (rows are slots on the host side)
        _lib.check(_lib.lib().ramp_track_plan(ctypes.byref(self.t), self.cur, _lib.stream()), "ramp_track_plan")
        self.active = True
        self._frames = 0
        return True

    def pinned_graph(self):
        """the [4][E_cap] int64 staging buffer of the hand-overs, in pinned host memory: the factor list crosses the bus through
        it in both directions, so the ~1.2 MB copies of a hand-back are plain DMA with no pin / unpin of user pages by the
        runtime"""
        if getattr(self, "_g_host", None) is None:
            self._g_host = torch.empty((4, self.E_cap), dtype=torch.int64).pin_memory()
        return self._g_host

    def factor_bound(self, counter):
        """upper bound of the live factor count from the lazy host copy of dyn: a frame adds at most (2r - 1) M
        factors, the copy is `lag` frames old (one more frame of slack in case it was read mid-update); the step
        flags a bound that turns out too small (status bit 32)"""
        d = self.lazy_state()
        lag = max(int(counter) - int(d[DYN_FRAME]), 1) + 1
        return min(int(d[DYN_E]) + lag * self.new_cap, self.E_cap)

    def factor_estimate(self):
        """the live factor count of the newest lazy copy of the sizes (a frame or two old; tools/e_trace.py: the count
        drifts by a few hundred per frame over 39k .. 46k at the bench size, so a maximum over 64 frames sits above the
        80-row gru tile's limit of 40960 most of the time although half of the frames are below it): picks tile sizes,
        bounds nothing.  Sequential rate 834 -> 853 kf/s, pipelined unchanged."""
        e = int(self.lazy_state()[DYN_E])
        if not self._e_seen or self._e_seen[-1] != e:
            self._e_seen.append(e)
        k = _E_EST_LAST
        if k > 0:                                     # the last k copies + a frame's growth (tools/e_trace.py: the count moves
            recent = list(self._e_seen)[-k:]          # by a few hundred per frame, over a range of 39k .. 46k)
            return int(min(max(recent) + _E_EST_MARGIN, self.E_cap))
        return int(min(max(self._e_seen), self.E_cap))

    def step(self, counter, flags, k_new=None, gate_event=None, gate_flag=None, gate_seq=0, E_bound=None):
        """gate_flag / gate_seq: a signal word (Signal) the update operator's last launch stores gate_seq into -- the
        cheaper form of gate_event (no packet on this stream); the waiting stream uses Signal.wait.  E_bound: the launch
        bound of this call (default: factor_bound(counter)); the two halves of an fp32 step share one."""
        ev = ctypes.c_void_p(gate_event) if gate_event else None
        self.t.E_hint = self.factor_estimate()
        self.t.gate_flag, self.t.gate_seq = (gate_flag, int(gate_seq) & 0xFFFFFFFF) if gate_flag else (None, 0)
        _lib.check(_lib.lib().ramp_track_step(ctypes.byref(self.t), self.cur, int(counter), int(flags),
                                              self.factor_bound(counter) if E_bound is None else int(E_bound),
                                              _lib.ptr(k_new), ev, _lib.stream()),
                   "ramp_track_step")
        if flags & KEYFRAME:
            self.cur ^= 1
        if flags & (UPDATE | UPDATE_POST | COMMIT) and not (flags & UPDATE_PRE and not flags & UPDATE_POST):
            self._frames += 1

    class _Groups:
        __slots__ = ("order", "gid", "seg_start", "ngroups", "ukeys", "E")

    class _Plan:
        __slots__ = ("ix_raw", "jx_raw", "ix", "jx", "g_kk", "g_ij", "max_kk", "max_ij", "kj", "E", "mask_ix", "mask_jx", "pair_mul")

    def plan_view(self, Eb):
        """the device-side plan of the current graph as the object FusedUpdate.hidden() takes (views of the capacity-sized
        buffers, Eb rows: entries between the live factor count and Eb are defined -- no neighbour, group 0, zero state row)"""
        p = DeviceTrack._Plan()
        p.ix_raw = p.ix = self.ix[:Eb]
        p.jx_raw = p.jx = self.jx[:Eb]
        p.kj, p.E, p.mask_ix, p.mask_jx, p.pair_mul = None, Eb, None, None, None
        for name, src, cap in (("g_kk", self.kk, self.kk_cap), ("g_ij", self.ij, self.ij_cap)):
            g = DeviceTrack._Groups()
            g.order, g.gid, g.seg_start, g.ngroups, g.ukeys, g.E = src["order"], src["gid"], src["seg"], src["ngroups"], src["ukeys"], Eb
            setattr(p, name, g)
        p.max_kk, p.max_ij = self.kk_cap, self.ij_cap
        return p

    def warm(self):
        """(on the current stream) read the correlation planes of the window once: csrc/track.hip::trk_warm_kernel"""
        _lib.check(_lib.lib().ramp_track_warm(ctypes.byref(self.t), _lib.ptr(self.sink), _lib.stream()), "ramp_track_warm")

    def lazy_state(self):
        """the host's (possibly a few frames old) copy of dyn -- never waited for.  The device writes the block while
        the host may be reading it: the frame tag sits in both 64-byte halves, a copy whose tags differ is re-read."""
        d = self.dyn_host.numpy()
        for _ in range(64):
            c = d.copy()
            if c[DYN_FRAME] == c[DYN_FRAME2]:
                self._lazy_ok = c
                return c
        # (64 torn reads in a row: the last CONSISTENT copy rather than a torn one -- its sizes are older, the launch bounds
        # derived from them carry a per-frame margin and status bit 32 flags a bound that was too small)
        return self._lazy_ok if getattr(self, "_lazy_ok", None) is not None else c

    def wait_frame(self, counter):
        """(fp32 steps) poll the pinned copy of the sizes until it is the one the plan of frame `counter` wrote, and return
        it: the host stays at most one frame ahead of the device there -- the correlation launch it has just enqueued keeps
        the GPU busy meanwhile -- and the update operator's library GEMMs run on exactly the live factor count"""
        import time
        t0 = time.perf_counter()
        while True:
            d = self.lazy_state()
            if int(d[DYN_FRAME]) >= int(counter) and int(d[DYN_FRAME]) == int(d[DYN_FRAME2]):
                return d
            if time.perf_counter() - t0 > 5.0:
                raise RuntimeError("device-resident tracker: the GPU has not finished a frame for 5 s")
            time.sleep(1e-5)

    def throttle(self, counter):
        """bound the host's run-ahead: wait (sleeping) until the lazy copy is at most MAX_AHEAD frames old, so that the
        capacity margins the host checks against it (frame buffers, delta log: + 8) hold -- a caller that feeds
        preloaded inputs enqueues frames ~5x faster than the GPU tracks them.  The GPU stays MAX_AHEAD frames deep in
        work, so this never idles it."""
        import time
        d = self.dyn_host.numpy()
        t0 = None
        while int(counter) - int(d[DYN_FRAME]) > MAX_AHEAD:
            if t0 is None:
                t0 = time.perf_counter()
            elif time.perf_counter() - t0 > 5.0:
                raise RuntimeError("device-resident tracker: the GPU has not finished a frame for 5 s")
            time.sleep(2e-5)
        if t0 is not None:                      # (Ramp_vo.stats: how often and how long the host waited for the GPU here)
            st = self.slam.stats
            st["throttle_waits"] += 1
            st["throttle_s"] += time.perf_counter() - t0

    def leave(self):
        """synchronise and return the host-side view of the state: dict(n, ii, jj, kk, rows (host arrays of the kept
        factors), net (device [EPREV, 384] view), log (list of (t1, t0, dP[7] device tensor)), status)"""
        torch.cuda.current_stream().synchronize()
        if self.fmap1_slot is not None:
            # the host-driven path addresses ring row r as slot r: undo the table's permutation of the level-0 planes
            # (cycle by cycle through ONE scratch plane: a gather of the whole ring would materialise a second copy of
            # it -- mem x 4.9 MB at 640 x 480 -- on every hand-back behind a dropped keyframe)
            perm = [int(v) for v in self.fmap1_slot.cpu().tolist()]          # row r lives in slot perm[r]
            if perm != list(range(len(perm))):
                buf, done = self.slam.fmap1_, [False] * len(perm)
                for r0 in range(len(perm)):
                    if done[r0] or perm[r0] == r0:
                        done[r0] = True
                        continue
                    tmp, r = buf[r0].clone(), r0
                    while perm[r] != r0:
                        buf[r].copy_(buf[perm[r]])
                        done[r], r = True, perm[r]
                    buf[r].copy_(tmp)
                    done[r] = True
                self.fmap1_slot.copy_(torch.arange(len(perm), dtype=torch.int32, device=self.fmap1_slot.device))
        d = self.dyn.cpu().numpy()
        Ek, n = int(d[DYN_EKEPT]), int(d[DYN_NROW])
        pg = self.pinned_graph()
        pg[:, :Ek].copy_(self.graph[self.cur][:, :Ek])           # (device -> pinned staging; pinned_graph's note)
        g = pg[:, :Ek].numpy().copy()
        nlog = int(d[DYN_NLOG])
        log = []
        if nlog:
            raw = self.dlog[:nlog].clone()
            ints = raw[:, :2].contiguous().view(torch.int32).cpu().numpy()
            log = [(int(ints[i, 0]), int(ints[i, 1]), raw[i, 2:9].clone()) for i in range(nlog)]
        self.active = False
        return dict(n=n, ii=np.ascontiguousarray(g[0]), jj=np.ascontiguousarray(g[1]), kk=np.ascontiguousarray(g[2]),
                    rows=np.ascontiguousarray(g[3]), net=self.net[0][:int(d[DYN_EPREV])], log=log,
                    status=int(d[DYN_STATUS]), weight=self.weight[:int(d[DYN_EPREV])].clone())
