"""Sliding-window visual-odometry front-end (reference: ramp/Ramp_vo.py:27-410).

Same plugin surface as the reference class -- ``Ramp_vo(cfg, network, train_cfg,
ht, wd)``, ``slam(t, input_tensor=(events, image, mask), intrinsics=K)``,
``slam.update()``, ``slam.terminate()``, attributes ``points_ / colors_ / m / n /
poses_ / patches_`` -- so ``evaluate.py::run`` drives it unchanged; the
pose-prediction mode (reference :414-534, ``evaluate.py::run_pose_pred``) is at the
end of the class.

MI355X-first differences that do not change results:
  * feature ring buffers are channels-last (one pixel's 128 channels contiguous);
  * reproject / corr(+pyramid stack) / BA / point cloud are one fused HIP launch
    each (reference: ~13 / ~26 / ~50 / 4 launches);
  * neighbour / group index structures are built once per graph change and
    shared by the update operator and BA;
  * STEADY STATE IS DEVICE RESIDENT (``track_dev.DeviceTrack``, csrc/track.hip): once the
    optimisation window is full, a tracked frame is ONE host call that never reads the
    device -- the keyframe decision (the reference's ``.item()``), the edit of the factor
    graph, the next frame's new factors and the graph plan are kernels, and every size that
    depends on a decision is read by the kernels from device memory.  ``settle()`` is the
    one synchronisation that brings the host mirror (``n``, ``ii/jj/kk``, ``net``, ``delta``)
    up to date; public attributes call it.  The first frames (initialisation, a window
    that is still filling), fp32 mode and anything unusual run the host-driven path below,
    which makes the same calls in the reference's order.
"""
from collections import OrderedDict

import ctypes
import os
import warnings

import numpy as np
import torch
import torch.nn.functional as F

from . import altcorr, fastba, hostenv, lietorch, ops, track_dev
from . import projective_ops as pops
from . import _lib
from ._lib import RAMP_NHWC, RAMP_NHWC32, kplane
from .update_fused import CORR_ROW
from .lietorch import SE3
from .net import GraphPlan, VONet
from .utils import Timer, preprocess_input


_CUS = {}


def _gru_tile_rows(E, device):
    """rows per workgroup ramp_upd_gru picks for E factors (csrc/update_mlp.hip: the tile that needs fewer rounds of one
    workgroup per CU)"""
    key = str(device)
    if key not in _CUS:
        _CUS[key] = torch.cuda.get_device_properties(device).multi_processor_count
    cus = _CUS[key]
    r4 = -(-(-(-E // 64)) // cus) * 4
    r5 = -(-(-(-E // 80)) // cus) * 5
    return 80 if r5 < r4 else 64


_FE_DELAY_US = int(os.environ.get("RAMP_FE_DELAY_US", "35"))     # A/B switch: 0 = the encoder graph right behind the selection
_GATE_FLAG_DELAY_US = 0
_FE_WAIT_PROBE = False      # (diagnostic, set by tools/fe_wait.py before the tracker is built: events around the main queue's wait for the front end)
# The front end's "done" is a data dependency (the frame commit reads fe_fmap1 / imap / gmap / patches): the wave that waits
# for its signal word must not give up while the producer can still arrive.  The time-out is a hang guard only -- 20 s, far
# beyond any profiler slow-down -- and a wait that does give up still raises sticky status bit 128, on which settle() /
# lazy_state() raise.  Runtimes that serialise kernels never get here (_kernels_are_serialised: events instead).  The gate
# wait (_gate_wait) keeps its 50 ms: it orders nothing but timing (a front end that starts early is slower, not wrong).
_FE_DONE_TIMEOUT_US = 20000000


def _kernels_are_serialised():
    """a cross-stream wait by a spinning wave needs the two streams to run CONCURRENTLY.  Under counter collection
    (rocprofv3 --pmc: ROCPROF_COUNTER_COLLECTION), AMD_SERIALIZE_KERNEL or HIP_LAUNCH_BLOCKING the runtime runs one
    kernel at a time -- the waiting wave would keep its producer off the GPU until it times out -- so the signal words
    are not used there: events order the streams"""
    def on(k):
        return os.environ.get(k, "0") not in ("", "0")
    return on("ROCPROF_COUNTER_COLLECTION") or on("AMD_SERIALIZE_KERNEL") or on("HIP_LAUNCH_BLOCKING") or on("RAMP_NO_FLAG_WAITS")
_LIVE = {}        # device index -> trackers alive on it (Ramp_vo._flag_wait_is_safe)


def _live_dec(idx):
    _LIVE[idx] = _LIVE.get(idx, 1) - 1


_WARM = os.environ.get("RAMP_WARM", "1") != "0"     # A/B switch: the cache warm-up behind the front end


class Ramp_vo:
    def __init__(self, cfg, network, train_cfg, ht=480, wd=640, device="cuda"):
        self.cfg = cfg
        self.event_bias = train_cfg["event_bias"]
        self.train_cfg = train_cfg
        self.device = torch.device(device)
        dev = self.device
        if dev.type != "cuda" and not getattr(self, "_allow_cpu", False):
            raise RuntimeError("rampvo_amd.Ramp_vo runs on the GPU only (HIP kernels, no CPU fallback); got device %s" % dev)
        # counters a caller (bench.py's legs) can read without touching the device: frames tracked device resident / host
        # driven, hand-backs to the host (settle), waits of the host for the GPU (track_dev.DeviceTrack.throttle)
        self.stats = dict(device_frames=0, host_frames=0, settles=0, throttle_waits=0, throttle_s=0.0)
        hostenv.fit_host_threads()          # (once per process) a host thread pool larger than the container's CPU quota: see there
        import weakref
        idx = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else -1)
        self._live_idx = idx
        _LIVE[idx] = _LIVE.get(idx, 0) + 1
        weakref.finalize(self, _live_dec, idx)

        self._dev = None                # DeviceTrack while the steady state is device resident
        self.lmbda = torch.as_tensor([1e-4], device=dev)
        self.load_weights(network)
        self.is_initialized = False
        self.enable_timing = False

        self._n = 0     # number of keyframes
        self._m = 0     # number of patches
        self.M = self.cfg.PATCHES_PER_FRAME
        self.N = self.cfg.BUFFER_SIZE
        self.ht, self.wd = ht, wd
        DIM, RES = self.DIM, self.RES

        self.tlist = []
        self.counter = 0
        self._tstamps = []          # host mirror of tstamps_[:n]
        self.tstamps_ = torch.zeros(self.N, dtype=torch.long, device=dev)
        self.poses_ = torch.zeros(self.N, 7, dtype=torch.float, device=dev)
        self.patches_ = torch.zeros(self.N, self.M, 3, self.P, self.P, dtype=torch.float, device=dev)
        self.intrinsics_ = torch.zeros(self.N, 4, dtype=torch.float, device=dev)
        self.points_ = torch.zeros(self.N * self.M, 3, dtype=torch.float, device=dev)
        self.colors_ = torch.zeros(self.N, self.M, 3, dtype=torch.uint8, device=dev)
        self.index_ = torch.arange(self.N, device=dev).view(-1, 1).repeat(1, self.M)
        self.index_[0] = 0
        self.index_map_ = torch.zeros(self.N, dtype=torch.long, device=dev)

        self.mem = 32
        self.dtype = torch.half if self.cfg.MIXED_PRECISION else torch.float
        self.kwargs = kwargs = {"device": dev, "dtype": self.dtype}
        h, w = ht // RES, wd // RES
        # channels-last ring buffers (reference: [mem,M,DIM], [mem,M,128,P,P], [1,mem,128,h,w])
        self.imap_ = torch.zeros(self.mem, self.M, DIM, **kwargs)
        self.gmap_ = torch.zeros(self.mem, self.M, self.P, self.P, 128, **kwargs)
        # pyramid on the GPU: [h][C/32][w][32] (fp16) / [h][C/16][w][16] (fp32) slots (csrc/altcorr.hip, the MFMA kernels' target
        # layout: 64 bytes per pixel and plane); otherwise plain channels-last.  fp32 features, default: the float32 slots
        # hold split fp16 parts [h][4][2][w][32] (self._split; corr_mfma_kernel<CorrX2>, include/ramp_hip.h RAMP_CORR_X2)
        self._chunked, self._lazy_net = self._layout_flags(h, w)
        self._split = bool(self._chunked and self.dtype == torch.float and self._f32_mode == 2)
        self._kp = KP = kplane(self.dtype)
        if self._chunked:
            self.fmap1_ = torch.zeros(self.mem, h, 128 // KP, w, KP, **kwargs)
            self.fmap2_ = torch.zeros(self.mem, h // 4, 128 // KP, w // 4, KP, **kwargs)
        else:
            self.fmap1_ = torch.zeros(self.mem, h, w, 128, **kwargs)
            self.fmap2_ = torch.zeros(self.mem, h // 4, w // 4, 128, **kwargs)
        self.pyramid = (self.fmap1_, self.fmap2_)

        self._net_map = None                     # host int64 [E]: row of _net_buf per current edge (-1: zeros)
        self._net_map_dev = None
        self._ixm = None
        # Frame pipelining (opt-in): the caller guarantees that the tensors it hands to __call__ are complete (not
        # still being produced on the current stream).  In the device-resident steady state the next frame's front
        # end is then launched on a side stream, gated by an event recorded before the last kernel of the previous
        # frame's update operator: it runs next to the gru chain and BA (a few small blocks on a 256-CU chip).
        # inputs_ready = "stream" (round 5): the same pipelining for callers whose tensors ARE still being produced on the
        # current stream when they call (the reference's evaluate.py resizes each frame right before slam(...)): the tracker
        # records an event on the caller's stream, its front end waits for THAT, and everything else of the tracker runs on
        # the tracker's own stream (_main_stream) -- so the caller's stream carries the caller's work only and the next
        # frame's inputs are not queued behind this frame's bundle adjustment.  The caller's stream in turn waits until the
        # front end has taken its copy of the inputs (they may be overwritten or freed right after the call, as on one
        # stream).  Same contract for reading state as with True: through settle() / update() / terminate() / peek() /
        # state_dict() / the n, m, ii ... properties, which join the tracker's stream first.  rampvo_amd.evaluate.run sets it.
        # RAMP_INPUTS_READY=stream|1 sets the default without touching the caller's code (the reference's evaluate.py with
        # its imports redirected: it reads points_ / colors_ / m only behind its closing update() calls, which join).
        self.inputs_ready = {"stream": "stream", "1": True}.get(os.environ.get("RAMP_INPUTS_READY", ""), False)
        self.device_steps = os.environ.get("RAMP_DEVICE_STEP", "1") == "1"     # A/B and test switch
        self._edge_tmpl = None
        self._shift_plan = None
        self._cur_stream = None
        self._corr_levels = None
        self._fc_plan = None           # (front-end outputs, patches, FrameCommitPlan or None)
        self._warned_slow = False
        self._extra_step_flags = 0     # (measurement: track_dev.WRAP_COORDS, bench.py's live-factor leg)
        self._ba_flags = 0             # bits of fastba's info seen so far (1: a pose step was dropped, 2: pair list overflow)
        self._init_streams(dev)
        self._net_buf = torch.zeros(1, 0, DIM, dtype=torch.float, device=dev)   # hidden state is fp32 (as under autocast)
        self._dii = torch.zeros(0, dtype=torch.long, device=dev)
        self._djj = torch.zeros(0, dtype=torch.long, device=dev)
        self._dkk = torch.zeros(0, dtype=torch.long, device=dev)
        # host mirror of the factor graph
        self._hii = np.zeros(0, np.int64)
        self._hjj = np.zeros(0, np.int64)
        self._hkk = np.zeros(0, np.int64)
        self._plan = None

        self.poses_[:, 6] = 1.0
        self._delta = {}
        self.patch_dict_ = None          # pose-prediction mode (reference :34-35)
        self.patches_models = None
        self._ba_info = torch.zeros(1, dtype=torch.int32, device=dev)
        self._last_K = self._last_K_raw = None     # last intrinsics written to a row of intrinsics_, and which row
        self._k_cuda, self._k_cuda_for = None, -1  # device-side intrinsics rows of CUDA `intrinsics` arguments (_cuda_intrinsics)
        self._last_K_row = -1

    def close(self):
        """kept for callers of earlier versions (the pipelined mode no longer owns a helper thread)"""

    # ------------------------------------------------------- state the device may own
    # In the device-resident steady state the keyframe count, the factor graph, the hidden state and the delta chain
    # live on the GPU; reading any of them through its public name first hands the state back (one synchronisation).
    def _mirror(name):
        priv = "_" + name if not name.startswith("_") else "_h" + name[1:]

        def get(self):
            if self._dev is not None and self._dev.active:
                self.settle()
            return getattr(self, priv)

        def put(self, v):
            if self._dev is not None and self._dev.active:
                self.settle()
            setattr(self, priv, v)
        return property(get, put)

    n, m, delta = _mirror("n"), _mirror("m"), _mirror("delta")
    ii, jj, kk = _mirror("dii"), _mirror("djj"), _mirror("dkk")
    _ii, _jj, _kk = _mirror("_ii"), _mirror("_jj"), _mirror("_kk")
    del _mirror

    def _layout_flags(self, h, w):
        """(fp16 pyramid in the MFMA correlation kernel's [h][C/32][w][32] slots, lazy hidden-state row map: the [E,384]
        state is re-indexed, not copied, when the graph changes)"""
        # (fp32 features: chunked planes are read by corr_mfma_kernel<float> only -- RAMP_CORR_F32_MFMA=0 keeps plain planes for
        # corr_kernel<float>, the reference kernel's summation order)
        # (the patchifier packs the planes inside the front end's graph: its mode -- _lib.corr_f32_mode() when it was built --
        # is the one that counts)
        self._f32_mode = int(getattr(self.network.patchify, "pack_f32", 0)) if self.dtype == torch.float else 0
        can = self.dtype == torch.half or (self.dtype == torch.float and self._f32_mode != 0)
        chunked = can and ops.pyramid_pack_supported(h, w) and (h // 4) > 0 and (w // 4) > 0
        if self.dtype == torch.half and not chunked:
            # a performance cliff, not an error: say so once
            warnings.warn("rampvo_amd: feature plane %dx%d does not fit the chunked pyramid layout (width %% 16, height %% 4 at "
                          "1/4 resolution): plain NHWC slots and the slower correlation kernel" % (w, h))
        return chunked, True

    def _init_streams(self, dev):
        self._fe_stream = torch.cuda.Stream(device=dev)
        self._main_stream = torch.cuda.Stream(device=dev)     # inputs_ready = "stream": the tracker's own main stream
        self._main_used, self._user_dirty = False, True
        self._ev_taken = torch.cuda.Event()
        # events are re-recorded every frame (creating one costs a hipEventCreate/Destroy pair per use)
        self._fe_done_sig, self._fe_done_seq = None, 0
        self._ev_fe_done, self._ev_gate, self._ev_in = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        for ev in (self._ev_fe_done, self._ev_gate, self._ev_in):
            ev.record()                      # torch creates the hipEvent lazily; csrc/track.hip records the raw handle
        self._gate_armed = False
        # the gate as a signal word the gru launch stores a sequence number into (no event packet between the update
        # operator's launches: tools/mb/stream_signal.hip); RAMP_GATE_FLAG=0: the event (A/B runs)
        self._gate_sig, self._gate_seq, self._gate_by_flag = None, 0, False

    # ------------------------------------------------------------------ weights
    def load_weights(self, network):
        if isinstance(network, str):
            checkpoint = torch.load(network, map_location="cpu")
            state_dict = checkpoint['model_state_dict'] if checkpoint.get('model_state_dict') else checkpoint
            new_state_dict = OrderedDict()
            for k, v in state_dict.items():
                if "update.lmbda" not in k:
                    new_state_dict[k.replace('module.', '')] = v
            self.network = VONet(cfg=self.train_cfg)
            self.network.load_state_dict(new_state_dict)
        else:
            self.network = network
        self.DIM, self.RES, self.P = self.network.DIM, self.network.RES, self.network.P
        self.network.to(self.device)
        self.network.eval()
        enc = getattr(self.network.patchify, "encoder", None)
        if hasattr(enc, "mixed_precision"):
            enc.mixed_precision = bool(self.cfg.MIXED_PRECISION)
            enc.fp8_mfma = bool(self.cfg.get("ENCODER_FP8", False))

    # -------------------------------------------------------------------- views
    @property
    def poses(self):
        return self.poses_.view(1, self.N, 7)

    @property
    def patches(self):
        return self.patches_.view(1, self.N * self.M, 3, 3, 3)

    @property
    def intrinsics(self):
        return self.intrinsics_.view(1, self.N, 4)

    @property
    def ix(self):
        return self.index_.view(-1)

    @property
    def imap(self):
        return self.imap_.view(1, self.mem * self.M, self.DIM)

    @property
    def gmap(self):
        """reference shape [1, mem*M, 128, 3, 3] (a channels-last view)"""
        return self.gmap_.view(1, self.mem * self.M, 3, 3, 128).permute(0, 1, 4, 2, 3)

    # --------------------------------------------------------------- hidden state
    # Reference: ``self.net`` [1,E,384] is compacted on every factor removal (:203-208) and grown with zero
    # rows on every append (:194-201) -- two full copies of a 60 MB tensor per frame.  On the GPU the tensor
    # of the PREVIOUS update is kept as is and a row map follows the graph edits; the update
    # operator's first row kernel gathers through it.  Reading ``self.net`` materialises it.
    @property
    def net(self):
        self.settle()
        if self._net_map is not None:
            m = torch.from_numpy(self._net_map).to(self._net_buf.device)
            rows = self._net_buf[:, m.clamp(min=0)] * (m >= 0).to(self._net_buf.dtype)[None, :, None]
            self._net_buf, self._net_map, self._net_map_dev = rows, None, None
        return self._net_buf

    @net.setter
    def net(self, value):
        self.settle()
        self._net_buf, self._net_map, self._net_map_dev = value, None, None

    def _net_rows(self):
        return self._net_map if self._net_map is not None else np.arange(self._net_buf.shape[1], dtype=np.int64)

    # ----------------------------------------------------------------- snapshot
    def settle(self):
        """bring the host mirror up to date: if the steady state is device resident, synchronise and take the
        keyframe count, the factor graph, the hidden-state row map and the delta chain back (the next tracked frame
        runs host-driven and hands the state over again)"""
        self._join_main()
        dv = self._dev
        if dv is None or not dv.active:
            return
        st = dv.leave()
        self.stats["settles"] += 1
        self._n, self._m = st["n"], st["n"] * self.M
        self._hii, self._hjj, self._hkk = st["ii"], st["jj"], st["kk"]
        # (host -> device through the tracker's pinned staging buffer: plain DMA, no pin / unpin of a pageable array)
        pg, Ek = dv.pinned_graph(), len(st["ii"])
        for r_, a_ in enumerate((st["ii"], st["jj"], st["kk"], st["rows"])):
            pg[r_, :Ek].numpy()[:] = a_
        g = torch.empty((4, Ek), dtype=torch.int64, device=self.device)
        g.copy_(pg[:, :Ek])
        self._dii, self._djj, self._dkk = g[0], g[1], g[2]
        self._net_buf, self._net_map, self._net_map_dev = st["net"][None], st["rows"], g[3]
        self._plan = None
        for t1, t0, dP in st["log"]:
            self._delta[t1] = (t0, SE3(dP))
        self._tstamps = [int(v) for v in self.tstamps_[:self._n].tolist()]
        self._last_K_row = self._n - 1           # every committed frame copied (or wrote) its intrinsics row
        if self._dev._frames:                    # the last update()'s confidence weights (reference :296; the host-driven
            self.last_weight = st["weight"][None]    # update() sets it; the pose-prediction mode reads it)
        self._note_ba_flags(st["status"] & 3)
        if st["status"] & ~3:
            raise RuntimeError("device-resident tracker: status bits %d (4 = factor list full, 8 = group-by key / group "
                               "capacity, 16 = delta log full, 32 = launch bound below the factor count, 64 = frame "
                               "buffers full (BUFFER_SIZE), 128 = a gate wait timed out: a front end ran unordered)"
                               % st["status"])

    def peek(self):
        """(keyframes, factors) right now -- synchronises, but leaves a device-resident state where it is (tests and
        benchmarks read these per frame; ``n`` / ``_ii`` would hand the state back to the host every time)"""
        self._join_main()
        dv = self._dev
        if dv is not None and dv.active:
            torch.cuda.current_stream().synchronize()
            d = dv.dyn.cpu().numpy()
            return dict(n=int(d[track_dev.DYN_NROW]), E=int(d[track_dev.DYN_EKEPT]), resident=True)
        return dict(n=self._n, E=len(self._hii), resident=False)

    def _note_ba_flags(self, bits):
        """bit 0: bundle adjustment dropped a pose step (Cholesky failed / not finite), bit 1: pair list overflow --
        the reference would have propagated NaN / exited; here the tracker goes on and says so once"""
        new = bits & ~self._ba_flags
        self._ba_flags |= bits
        if new & 1:
            warnings.warn("bundle adjustment dropped a pose step (normal equations not positive definite)")
        if new & 2:
            raise RuntimeError("bundle adjustment: more pair records on one pose than ba_assemble's list holds")

    def state_dict(self):
        """VO state as CPU tensors in the REFERENCE's layouts (NCHW feature buffers), so a
        snapshot can move between devices / implementations.  (The reference's VO state
        is not checkpointable; this is what bench.py uses to hand the steady state to
        the CPU baseline and what the teacher-forced parity tests inject.)"""
        self.settle()
        n = self.n
        c = lambda t: t.detach().cpu().clone()
        return dict(
            n=n, m=self.m, counter=self.counter, is_initialized=self.is_initialized, tlist=list(self.tlist),
            tstamps=c(self.tstamps_[:n + 1]), poses=c(self.poses_[:n + 1]), patches=c(self.patches_[:n + 1]),
            intrinsics=c(self.intrinsics_[:n + 1]), colors=c(self.colors_[:n + 1]),
            imap=c(self.imap_), gmap=c(self.gmap_.permute(0, 1, 4, 2, 3)),
            fmap1=c(self._fmap_nchw(self.fmap1_)), fmap2=c(self._fmap_nchw(self.fmap2_)),
            net=c(self.net), ii=c(self.ii), jj=c(self.jj), kk=c(self.kk),
            delta={k: (v[0], c(v[1].data)) for k, v in self.delta.items()})

    def _fmap_nchw(self, buf):
        if self._split:                                                # pairs -> values (22 significant bits), [mem, 128, h, w]
            hi, lo = ops.unpack_split(buf)
            return (hi.float() + lo.float() * 2.0 ** -11).permute(0, 3, 1, 2)
        if self._chunked:                                              # [mem, h, 128 / kp, w, kp] -> [mem, 128, h, w]
            return buf.permute(0, 2, 4, 1, 3).reshape(buf.shape[0], 128, buf.shape[1], buf.shape[3])
        return buf.permute(0, 3, 1, 2)

    def load_state_dict(self, sd):
        self.settle()
        dev = self.device
        n = int(sd["n"])
        self.n, self.m, self.counter = n, int(sd["m"]), int(sd["counter"])
        self.is_initialized = bool(sd["is_initialized"])
        self.tlist = list(sd.get("tlist", []))
        k = sd["poses"].shape[0]
        self.tstamps_[:k] = sd["tstamps"].to(dev)
        self._tstamps = [int(v) for v in sd["tstamps"][:n].tolist()]
        self.poses_[:k] = sd["poses"].to(dev)
        self.patches_[:k] = sd["patches"].to(dev)
        self.intrinsics_[:k] = sd["intrinsics"].to(dev)
        if "colors" in sd:
            self.colors_[:k] = sd["colors"].to(dev)
        self.imap_.copy_(sd["imap"].to(dev))
        self.gmap_.copy_(sd["gmap"].to(dev).permute(0, 1, 3, 4, 2))
        for buf, key in ((self.fmap1_, "fmap1"), (self.fmap2_, "fmap2")):
            src = sd[key].to(dev)                                      # [mem, 128, h, w]
            if self._split:
                buf.copy_(ops.pack_split(src.to(self.dtype).permute(0, 2, 3, 1)))
                continue
            if self._chunked:
                src = src.reshape(src.shape[0], 128 // self._kp, self._kp, src.shape[2], src.shape[3]).permute(0, 3, 1, 4, 2)
            else:
                src = src.permute(0, 2, 3, 1)
            buf.copy_(src)
        self.net = sd["net"].to(dev)
        self.ii, self.jj, self.kk = (sd[x].to(dev).long() for x in ("ii", "jj", "kk"))
        self._ii, self._jj, self._kk = (sd[x].cpu().numpy().astype(np.int64) for x in ("ii", "jj", "kk"))
        self.delta = {k: (v[0], SE3(v[1].to(dev))) for k, v in sd.get("delta", {}).items()}
        self._plan = None
        self._net_map_dev = None
        self._last_K = self._last_K_raw = None
        self._last_K_row = -1

    # --------------------------------------------------------------- trajectory
    def get_pose(self, t):
        if t in self.traj:
            return SE3(self.traj[t])
        t0, dP = self.delta[t]
        return dP * self.get_pose(t0)

    def terminate(self):
        """interpolate the poses of dropped frames; returns (inverse poses [T,7], tstamps)"""
        self.settle()
        self.traj = {}
        ts = self._tstamps[:self.n] if len(self._tstamps) >= self.n else self.tstamps_[:self.n].tolist()
        for i in range(self.n):
            self.traj[ts[i]] = self.poses_[i]
        poses = [self.get_pose(t) for t in range(self.counter)]
        poses = lietorch.stack(poses, dim=0)
        poses = poses.inv().data.cpu().numpy()
        return poses, np.array(self.tlist, dtype=float)

    # ------------------------------------------------------------------ kernels
    def corr(self, coords, indicies=None, order=None):
        """local correlation volume, both pyramid levels fused: [1, E, 882].  order: the graph
        plan's target-frame-major edge permutation (scheduling only)"""
        ii, jj = indicies if indicies is not None else (self.kk, self.jj)
        if (self._chunked and self.dtype == torch.half and order is not None and coords.shape[1] > 0 and coords.dtype == torch.float32 and coords.is_contiguous()
                and ii.is_contiguous() and jj.is_contiguous()):
            # the tracker's own per-frame call: same launch as below without the generic wrapper's checks (the
            # level descriptors of the fixed pyramid buffers are built once)
            return self._corr_launch(coords, ii, jj, order)
        # ring-buffer slots (kk % (M*mem), jj % mem) are taken inside the kernel; rows padded 882 -> 896
        # (16-byte aligned rows for the first Linear layer, update_fused.py).  fp32 features: the tracker's own volume comes
        # from corr_mfma_kernel<float> (the fp32 matrix cores' summation order, <= 1e-5 of the reference kernel's fmaf chain --
        # 2.1x faster; altcorr.corr, the reference entry point, keeps the chain's order; RAMP_CORR_F32_MFMA=0: here too)
        return altcorr.corr_pyramid(self.gmap_.view(-1, 3, 3, 128), self.pyramid, coords[0], ii, jj, 3, (1, 4),
                                    RAMP_NHWC32 if self._chunked else RAMP_NHWC, order=order, row_elems=CORR_ROW,
                                    mod_ii=self.M * self.mem, mod_jj=self.mem,
                                    fast_f32=(2 if self._split else 1) if self._f32_mode != 0 else 0)

    def _corr_launch(self, coords, ii, jj, order):
        """ramp_corr_fwd_ordered on the tracker's own buffers (fp16 chunked pyramid, padded rows)"""
        if self._corr_levels is None:
            lv = (_lib.CorrLevel * 2)()
            lv[0] = _lib.CorrLevel(self.fmap1_.data_ptr(), self.fmap1_.shape[1], self.fmap1_.shape[3], 1.0)
            lv[1] = _lib.CorrLevel(self.fmap2_.data_ptr(), self.fmap2_.shape[1], self.fmap2_.shape[3], 4.0)
            self._corr_levels = lv
        E = coords.shape[1]
        out = torch.empty((E, CORR_ROW), dtype=torch.half, device=self.device)
        _lib.check(_lib.lib().ramp_corr_fwd_ordered(
            _lib.ptr(self.gmap_), self._corr_levels, 2, _lib.ptr(coords), _lib.ptr(ii), _lib.ptr(jj),
            _lib.ptr(order), _lib.ptr(out), CORR_ROW, self.M * self.mem, self.mem, E, self.mem * self.M,
            self.mem, 128, 3, 3, _lib.RAMP_F16, RAMP_NHWC32, _lib.stream()), "ramp_corr_fwd_ordered")
        return out.view(1, E, CORR_ROW)

    def reproject(self, indicies=None, poses=None, patches=None, intrinsics=None):
        if indicies is None and poses is None and patches is None and intrinsics is None:
            # the tracker's own per-frame call on its own (contiguous fp32) buffers and graph
            E = self.ii.shape[0]
            out = torch.empty((1, E, 2, self.P, self.P), dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().ramp_transform(_lib.ptr(self.poses_), _lib.ptr(self.patches_), _lib.ptr(self.intrinsics_),
                                                 _lib.ptr(self.ii), _lib.ptr(self.jj), _lib.ptr(self.kk), _lib.ptr(out),
                                                 E, self.P, 0, _lib.stream()), "ramp_transform")
            return out
        (ii, jj, kk) = indicies if indicies is not None else (self.ii, self.jj, self.kk)
        poses = poses if poses is not None else self.poses
        patches = patches if patches is not None else self.patches
        intrinsics = intrinsics if intrinsics is not None else self.intrinsics
        return pops.reproject(poses, patches, intrinsics, ii, jj, kk)      # [1,E,2,3,3]


    # -------------------------------------------------------------------- graph
    def _upload(self, a):
        """host array -> device"""
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def _new_edges(self, n1):
        """the factors frame n1-1 adds when it is accepted (reference :312-325, :394-395); host arrays"""
        r, M = self.cfg.PATCH_LIFETIME, self.M
        if n1 >= r:
            # steady state: the pattern of frame n1 is the pattern of frame r shifted by n1 - r frames
            if self._edge_tmpl is None:
                self._edge_tmpl = self._new_edges_at(r)
            d = n1 - r
            t_ii, t_jj, t_kk = self._edge_tmpl
            return t_ii + d, t_jj + d, t_kk + M * d
        return self._new_edges_at(n1)

    def _new_edges_at(self, n1):
        r, M = self.cfg.PATCH_LIFETIME, self.M
        kf, jf = np.meshgrid(np.arange(M * max(n1 - r, 0), M * max(n1 - 1, 0)), np.arange(n1 - 1, n1), indexing='ij')
        kb, jb = np.meshgrid(np.arange(M * max(n1 - 1, 0), M * n1), np.arange(max(n1 - r, 0), n1), indexing='ij')
        kk = np.concatenate([kf.reshape(-1), kb.reshape(-1)]).astype(np.int64)
        jj = np.concatenate([jf.reshape(-1), jb.reshape(-1)]).astype(np.int64)
        return kk // M, jj, kk

    def append_factors(self, ii, jj):
        """ii: patch indices, jj: frame indices (host arrays) -- reference :194-201"""
        ii = np.asarray(ii, np.int64)
        jj = np.asarray(jj, np.int64)
        src = ii // self.M                      # == self.ix[ii]: index_[r] = r for every frame row
        dev = self._upload(np.stack([src, jj, ii]))                # one copy for the three arrays
        self._jj = np.concatenate([self._jj, jj])
        self._kk = np.concatenate([self._kk, ii])
        self._ii = np.concatenate([self._ii, src])
        self.ii = torch.cat([self.ii, dev[0]])
        self.jj = torch.cat([self.jj, dev[1]])
        self.kk = torch.cat([self.kk, dev[2]])
        if self._lazy_net:
            self._net_map = np.concatenate([self._net_rows(), np.full(len(ii), -1, np.int64)])
            self._net_map_dev = None
        else:
            net = torch.zeros(1, len(ii), self.DIM, dtype=torch.float, device=self.device)
            self.net = torch.cat([self.net, net], dim=1)
        self._plan = None

    def remove_factors(self, m):
        """m: host boolean mask of factors to drop -- reference :203-208"""
        m = np.asarray(m, bool)
        if not m.any():
            return
        keep = np.nonzero(~m)[0]
        self._ii, self._jj, self._kk = self._ii[keep], self._jj[keep], self._kk[keep]
        kd = self._upload(keep)
        self.ii, self.jj, self.kk = self.ii[kd], self.jj[kd], self.kk[kd]
        if self._lazy_net:
            self._net_map, self._net_map_dev = self._net_rows()[keep], None
        else:
            self.net = self.net[:, kd]
        self._plan = None

    def _build_plan(self, h_ii, h_jj, h_kk, d_ii, d_jj, d_kk, ranges=None):
        if ranges is not None:
            k_lo, k_hi, f_lo, f_hi = ranges
        else:
            k_lo, k_hi = int(h_kk.min()), int(h_kk.max()) + 1
            f_lo = int(min(h_ii.min(), h_jj.min()))
            f_hi = int(max(h_ii.max(), h_jj.max())) + 1
        # group-count upper bounds from the ranges (no host-side unique, no device read-back)
        max_kk = (k_hi // self.M - k_lo // self.M + 1) * self.M
        max_ij = min((f_hi - f_lo) ** 2, len(h_ii))
        return GraphPlan.build(d_ii, d_jj, d_kk, kk_bound=self.N * self.M, jj_bound=self.N,
                               max_kk=max_kk, max_ij=max_ij, kk_range=(k_lo, k_hi), frame_range=(f_lo, f_hi))

    def _graph_plan(self):
        if self._plan is None or self._plan.E != len(self._ii):
            self._plan = self._build_plan(self._ii, self._jj, self._kk, self.ii, self.jj, self.kk)
        return self._plan

    def __edges_forw(self):
        r = self.cfg.PATCH_LIFETIME
        t0 = self.M * max((self.n - r), 0)
        t1 = self.M * max((self.n - 1), 0)
        kk, jj = np.meshgrid(np.arange(t0, t1), np.arange(self.n - 1, self.n), indexing='ij')
        return kk.reshape(-1), jj.reshape(-1)

    def __edges_back(self):
        r = self.cfg.PATCH_LIFETIME
        t0 = self.M * max((self.n - 1), 0)
        t1 = self.M * max((self.n - 0), 0)
        kk, jj = np.meshgrid(np.arange(t0, t1), np.arange(max(self.n - r, 0), self.n), indexing='ij')
        return kk.reshape(-1), jj.reshape(-1)
    # ------------------------------------------------------------------- motion
    def motion_probe(self):
        """median update magnitude of the newest patches against the newest frame (reference :210-225)"""
        kk = torch.arange(self.m - self.M, self.m, device=self.device)
        jj = self.n * torch.ones_like(kk)
        ii = kk // self.M
        coords = self.reproject(indicies=(ii, jj, kk))
        corr = self.corr(coords, indicies=(kk, jj)).to(self.dtype)
        fu = self.network.update.fused(self.dtype)
        plan = GraphPlan.build(ii, jj, kk, kk_bound=self.N * self.M, jj_bound=self.N, max_kk=self.M, max_ij=1)
        _, relu_t = fu.hidden(None, self.imap_.view(-1, self.DIM), kk, self.M * self.mem, corr[0], plan)
        delta = fu.heads(relu_t)[None, :, :2]
        return torch.quantile(delta.norm(dim=-1).float(), 0.5)

    def _apply_removal(self, k):
        """device side of dropping keyframe k: shift the per-frame state down by one row"""
        n = self.n
        del self._tstamps[k]
        if self._last_K_row > k:
            self._last_K_row -= 1
        if (self.M * 3) % 4 == 0:
            if self._shift_plan is None:
                self._shift_plan = ops.ShiftPlan([(self.tstamps_, 0), (self.colors_, 0), (self.poses_, 0),
                                                  (self.patches_, 0), (self.intrinsics_, 0), (self.imap_, self.mem),
                                                  (self.gmap_, self.mem), (self.fmap1_, self.mem),
                                                  (self.fmap2_, self.mem)])
            self._shift_plan.run(k, n)                                                # one launch
        else:
            for buf in (self.tstamps_, self.colors_, self.poses_, self.patches_, self.intrinsics_):
                buf[k:n - 1] = buf[k + 1:n].clone()
            dst = torch.arange(k, n - 1, device=self.device) % self.mem
            src = torch.arange(k + 1, n, device=self.device) % self.mem
            for buf in (self.imap_, self.gmap_, self.fmap1_, self.fmap2_):
                buf[dst] = buf[src]
        self.n -= 1
        self.m -= self.M
    def keyframe(self):
        """drop keyframe n-KEYFRAME_INDEX if the motion around it is small, then cull factors older
        than REMOVAL_WINDOW (reference :237-274) -- the host-driven form: the motion test is read back (the
        reference's two ``.item()``), both removals are decided on the host mirror in one pass
        (``ramp_graph_edit_host``) and applied to the device state as ONE compaction.  The device-resident steady
        state takes the same decision and makes the same edit with kernels (csrc/track.hip)."""
        cfg = self.cfg
        i, j = self.n - cfg.KEYFRAME_INDEX - 1, self.n - cfg.KEYFRAME_INDEX + 1
        k = self.n - cfg.KEYFRAME_INDEX
        plan = self._graph_plan()
        mm = ops.motionmag(self.poses, self.patches, self.intrinsics, self.ii, self.jj, self.kk, plan.g_ij,
                           j * plan.pair_mul + i, i * plan.pair_mul + j, beta=0.5)   # keys are jj*mul+ii
        vals = torch.cat([mm.reshape(2), self._ba_info.to(torch.float32)]).cpu().numpy()   # the read-back
        self._note_ba_flags(int(vals[2]))
        remove = (float(vals[0]) + float(vals[1])) / 2 < cfg.KEYFRAME_THRESH      # python floats, as the reference
        if remove:
            t0, t1 = self._tstamps[k - 1], self._tstamps[k]
            dP = SE3(self.poses_[k]) * SE3(self.poses_[k - 1]).inv()
            self.delta[t1] = (t0, dP)
        n_after = self.n - 1 if remove else self.n
        E = len(self._ii)
        buf = np.empty((4, max(E, 1)), np.int64)
        rows = self._net_map
        h_ii, h_jj, h_kk = (np.ascontiguousarray(a) for a in (self._ii, self._jj, self._kk))
        Ek = _lib.lib().ramp_graph_edit_host(
            h_ii.ctypes.data, h_jj.ctypes.data, h_kk.ctypes.data,
            np.ascontiguousarray(rows).ctypes.data if rows is not None else None, E, self.M, k if remove else -1,
            n_after, cfg.REMOVAL_WINDOW, buf.ctypes.data, buf.shape[1], None)
        assert Ek >= 0
        if remove:
            self._apply_removal(k)
        if Ek == E and not remove:
            return
        dev = self._upload(buf[:, :Ek])
        self._ii, self._jj, self._kk = buf[0, :Ek], buf[1, :Ek], buf[2, :Ek]
        self.ii, self.jj, self.kk = dev[0], dev[1], dev[2]
        if self._lazy_net:
            self._net_map, self._net_map_dev = buf[3, :Ek], dev[3]
        else:
            self.net = self.net[:, dev[3]]
        self._plan = None

    # ------------------------------------------------------------------- update
    def update(self):
        """reference :276-310 (host-driven form; the device-resident step makes the same launches from C)"""
        if self._cur_stream is None:                 # (a public call, not the tracked frame's own update)
            self._join_main()
        with Timer("other", enabled=self.enable_timing):
            plan = self._graph_plan()
            coords = self.reproject()
            order = plan.g_ij.order
            corr = self.corr(coords, order=order).to(self.dtype)
            # GEMMs + row-fused glue (csrc/update.hip) / the fused fp16 chains (csrc/update_mlp.hip); the context
            # gather, the heads' activations, `target = centre + delta` and filter_features are folded in
            fu = self.network.update.fused(self.dtype)
            net_map = None
            if self._net_map is not None:
                net_map = self._net_map_dev if self._net_map_dev is not None else self._upload(self._net_map)
            out32, relu_t = fu.hidden(self._net_buf[0], self.imap_.view(-1, self.DIM), self.kk, self.M * self.mem,
                                      corr[0], plan, net_map=net_map,
                                      heads_at=(coords[0].contiguous(), self.wd // 4, self.ht // 4))
            self.net = out32[None]
            tw = fu.last_tw if relu_t is None else fu.heads_target_weight(relu_t, coords[0], self.wd // 4, self.ht // 4)
            if tw is not None:
                target, weight = tw
            else:
                target, weight, _ = fu.target_weight(fu.heads(relu_t), coords[0], self.wd // 4, self.ht // 4)
            self.last_weight = weight
        with Timer("BA", enabled=self.enable_timing):
            t0 = self.n - self.cfg.OPTIMIZATION_WINDOW if self.is_initialized else 1
            t0 = max(t0, 1)
            try:
                fastba.BA(self.poses, self.patches, self.intrinsics, target, weight, self.lmbda, self.ii, self.jj,
                          self.kk, t0, self.n, M=self.M, iterations=2, eff_impl=False, info=self._ba_info, plan=plan)
            except Exception as e:  # same recovery as the reference (:302-306)
                print(f"WARNING: BA failed...{e}")
            if self._ixm is None or self._ixm.shape[0] < self.m:
                self._ixm = torch.arange(self.N * self.M, device=self.device) // self.M     # patch -> source frame
            ixm = self._ixm[:self.m]
            self.points_[:self.m] = pops.point_cloud(self.poses, self.patches_.view(-1, 3, 3, 3)[:self.m],
                                                     self.intrinsics, ixm)

    # --------------------------------------------------------------------- call
    def __call__(self, tstamp, input_tensor, intrinsics):
        """track a new frame"""
        input_ = preprocess_input(input_tensor=input_tensor)
        with torch.no_grad():
            if self.inputs_ready == "stream" and self.device.type == "cuda":
                return self._call_on_own_stream(tstamp, input_, intrinsics)
            self._join_main()
            self._cur_stream = self._current_stream()
            try:
                if self._dev is not None and self._dev.active:
                    self.stats["device_frames"] += 1
                    return self._track_device(tstamp, input_, intrinsics)
                self.stats["host_frames"] += 1
                return self._track(tstamp, input_, intrinsics)
            finally:
                self._cur_stream = None

    def _call_on_own_stream(self, tstamp, input_, intrinsics):
        """inputs_ready = "stream": see __init__.  The caller's stream gets one event record and one event wait per frame."""
        user, main, fe = torch.cuda.current_stream(self.device), self._main_stream, self._fe_stream
        self._ev_in.record(user)                        # the inputs (and whatever else the caller enqueued) up to here
        for t in input_[:2]:
            if torch.is_tensor(t) and t.is_cuda:        # allocated on the caller's stream, read on the tracker's
                t.record_stream(main)
                t.record_stream(fe)
        # the main stream itself waits for the caller's stream only when it has to: after a public call ran tracker work
        # there (update(), settle() ...: _join_main marks it), or when this frame's front end runs on the main stream (no
        # gate armed: host-driven frames, the first device-resident one) -- in the steady state the front-end stream's wait
        # is the only one, and the main stream keeps no event packet between two frames
        gated = self._dev is not None and self._dev.active and self._gate_armed
        if self._user_dirty or not gated:
            main.wait_event(self._ev_in)
            self._user_dirty = False
        self._in_event_pending, self._taken_recorded = True, False
        self._cur_stream = main
        try:
            with torch.cuda.stream(main):
                if self._dev is not None and self._dev.active:
                    self.stats["device_frames"] += 1
                    out = self._track_device(tstamp, input_, intrinsics)
                else:
                    self.stats["host_frames"] += 1
                    out = self._track(tstamp, input_, intrinsics)
        finally:
            self._cur_stream = None
            self._in_event_pending = False
            self._main_used = True                      # (a settle() inside the call may have joined: there is new work now)
        # the caller may overwrite / free its tensors once the front end has run on them (a frame whose front end ran on the
        # tracker's main stream -- host-driven frames -- : once that frame is through)
        if not self._taken_recorded:
            self._ev_taken.record(main)
        user.wait_event(self._ev_taken)
        return out

    def _join_main(self):
        """(public entry points that run on the caller's stream) everything the tracker enqueued on its own stream first"""
        self._user_dirty = True                       # (whatever follows runs tracker work on the caller's stream)
        if getattr(self, "_main_used", False):
            self._main_stream.synchronize()
            self._fe_stream.synchronize()
            self._main_used = False

    def _current_stream(self):
        return torch.cuda.current_stream()

    def _cur(self):
        """the caller's stream; looked up once per __call__ (torch.cuda.current_stream() costs ~4 us)"""
        return self._cur_stream if self._cur_stream is not None else torch.cuda.current_stream()

    def _intrinsics_row(self, intrinsics, accepts):
        """intrinsics at feature resolution.  Returns (kq, k_dev): k_dev is a device [4] tensor when row n needs new
        values, None when it can be copied from row n-1 (the usual case: same values as the previous frame, recognised
        without building new arrays)"""
        k_dev = None
        if torch.is_tensor(intrinsics) and intrinsics.is_cuda:
            # The reference's evaluate.py hands `intrinsics.cuda()` -- a device tensor produced on the caller's stream right
            # before the call.  Reading it back would stall the host behind everything queued (and, with the tracker on its
            # own streams, race the producer): the row is written from a device-side conversion instead, every frame, on a
            # stream that is ordered behind the caller's (_cuda_intrinsics); the host keeps no copy to compare with.
            self._K_raw_now = None
            return None, self._cuda_intrinsics(intrinsics)
        raw = self._last_K_raw
        # the row copy (intrinsics_[n] = intrinsics_[n-1]) is only right while row n-1 is the row that holds
        # _last_K: not after a frame that wrote a new K to row n and was then rejected by the motion probe
        row_ok = self._last_K_row == self._n - 1
        if (raw is not None and row_ok and intrinsics.device.type == "cpu" and intrinsics.dtype == torch.float32
                and torch.equal(intrinsics, raw)):
            kq = self._last_K
        else:
            kq = intrinsics.detach().cpu().float().numpy() / self.RES
            self._K_raw_now = intrinsics.detach().cpu().float().clone()
            if accepts and not (row_ok and self._last_K is not None and np.array_equal(kq, self._last_K)):
                k_dev = self._upload(kq.astype(np.float32))
        return kq, k_dev

    def _cuda_intrinsics(self, intrinsics, stream=None):
        """device [4] fp32 = intrinsics / RES for a CUDA ``intrinsics`` (cached per call: the pipelined device-resident frame
        converts it ahead, on its front-end stream).  Two buffers by frame parity: a frame's commit launch has read its row
        before the front end of the frame after the next one may start."""
        if self._k_cuda_for == self.counter:
            return self._k_cuda[self.counter & 1]
        if self._k_cuda is None:
            self._k_cuda = [torch.zeros(4, dtype=torch.float32, device=self.device) for _ in range(2)]
        cur = self._cur()
        st = stream if stream is not None else cur
        if stream is None and getattr(self, "_in_event_pending", False):
            cur.wait_event(self._ev_in)                 # (a host-driven frame behind a front end that ran on its own stream)
        buf = self._k_cuda[self.counter & 1]
        with torch.cuda.stream(st):
            torch.div(intrinsics.detach().reshape(4).to(torch.float32), float(self.RES), out=buf)
        intrinsics.record_stream(st)
        self._k_cuda_for = self.counter
        return buf

    def _gate_signal(self):
        """the signal word of the gate, or None (RAMP_GATE_FLAG=0, no signal memory, or 2^31 frames behind us: the event
        then).  Where in the step the word is stored is csrc/track.hip's RAMP_GATE_AT (default: by the first SoftAgg launch)"""
        if self._gate_sig is None:
            use = os.environ.get("RAMP_GATE_FLAG", "1") != "0" and not _kernels_are_serialised()
            with torch.cuda.device(self.device):          # (the word must live on the tracker's GPU, not the caller's current one)
                self._gate_sig = track_dev.Signal() if use else False
        sig = self._gate_sig
        return sig if (sig and sig.ptr is not None and self._gate_seq < 0x7FFFFFF0) else None

    def _fe_done_to(self, fe, cur, dv):
        """the front end's "done" for the main stream: a word in signal memory stored by a one-thread launch behind the
        encoder graph and looked at by one sleeping wave in front of the frame commit (RAMP_FE_DONE_FLAG=0, or no signal
        memory, or a runtime that serialises kernels (_kernels_are_serialised): an event record + stream wait, ~10 us of
        packets on the serial chain even when the front end is long done).  The commit READS what the front end wrote, so
        this wait's time-out is a hang guard (20 s), not a scheduling choice"""
        if self._fe_done_sig is None:
            use = os.environ.get("RAMP_FE_DONE_FLAG", "1") != "0" and not _kernels_are_serialised()
            with torch.cuda.device(self.device):
                self._fe_done_sig = track_dev.Signal() if use else False
        sig = self._fe_done_sig
        # A spinning wave needs its producer to run CONCURRENTLY: the runtime maps streams onto a handful of hardware queues,
        # and a waiter that shares one with its producer would sit in front of it until the hang guard fires.  Two streams of
        # one tracker (the caller's + the front end's: inputs_ready = True) have been measured safe on every box; with a third
        # tracker stream (inputs_ready = "stream": _main_stream), or a second tracker in the process, the DATA dependency goes
        # through an event -- the gate (_gate_wait), which orders timing only, keeps its word
        if sig and self._flag_wait_is_safe() and sig.ptr is not None and self._fe_done_seq < 0x7FFFFFF0:
            self._fe_done_seq += 1
            _lib.check(_lib.lib().ramp_stream_signal(ctypes.c_void_p(fe.cuda_stream), sig.ptr, self._fe_done_seq), "ramp_stream_signal")
            sig.wait(cur, self._fe_done_seq, timeout_us=_FE_DONE_TIMEOUT_US, status=dv.status_ptr)
        else:
            self._ev_fe_done.record(fe)
            cur.wait_event(self._ev_fe_done)

    def _flag_wait_is_safe(self):
        return self.inputs_ready is True and _LIVE.get(self._live_idx, 0) <= 1

    def _gate_wait(self, fe):
        """(front-end stream) wait for the previous frame's gate: the signal word if that step stored one, else the event"""
        if self._gate_by_flag:
            self._gate_sig.wait(fe, self._gate_seq, then_delay_us=_GATE_FLAG_DELAY_US,
                                status=self._dev.status_ptr if self._dev is not None else None)
        else:
            fe.wait_event(self._ev_gate)

    def _fe_delay(self):
        """(on the front-end stream, behind the selection) hold the encoder graph back a little: its LSTM launch should not
        arrive while the gru launch is still filling the chip (csrc/track.hip::trk_delay_kernel)"""
        if _FE_DELAY_US > 0:
            _lib.check(_lib.lib().ramp_stream_delay(_FE_DELAY_US, _lib.stream()), "ramp_stream_delay")

    # ----------------------------------------------------- device-resident steady state
    def _track_device(self, tstamp, input_, intrinsics):
        """one tracked frame without a device->host read: front end (hipGraph), then ONE C call that enqueues the frame
        stores, update(), keyframe() and the next frame's append_factors (csrc/track.hip::ramp_track_step)"""
        dv = self._dev
        mask = input_[2]
        accepts = mask is None or bool(mask)
        dv.throttle(self.counter)                   # at most MAX_AHEAD frames ahead of the newest lazy copy (the margins below)
        lazy = dv.lazy_state()                      # whatever copy of the device-side sizes has arrived: never waited for
        if (accepts and (lazy[track_dev.DYN_N] + 8 >= self.N or lazy[track_dev.DYN_NLOG] + 8 >= dv.log_cap
                         or lazy[track_dev.DYN_STATUS] & ~3)):
            self.settle()                           # buffer / log nearly full, or a capacity flag: back to the host
            return self._track(tstamp, input_, intrinsics)
        cur = self._cur()
        if self.inputs_ready and self._gate_armed:
            # the caller's tensors are complete: the front end runs on its own stream, next to what is left of the
            # previous frame from the gru chain on (the event is recorded inside ramp_track_step)
            # (the staging copies and the patch selection go out ahead of the gate: they depend on the input alone)
            fe = self._fe_stream
            # Where the encoder graph starts relative to the gru launch depends on that launch's tile (csrc/update_mlp.hip
            # picks 64 or 80 rows per workgroup by the factor count, tools/corun_gru_lstm.py): with 80-row tiles (two full
            # rounds of one workgroup per CU, 254 VGPRs) the front end's LSTM launch cannot take a CU from it, so the
            # selection goes out AHEAD of the gate and the graph right behind it; with 64-row tiles an LSTM launch that
            # arrives within ~40 us costs gru 45 us, so the selection runs behind the gate and the graph is held back
            # another 35 us (_fe_delay).
            ahead = _gru_tile_rows(dv.factor_estimate(), self.device) == 80
            if getattr(self, "_in_event_pending", False):
                fe.wait_event(self._ev_in)              # inputs_ready = "stream": the caller's stream up to the call
            if accepts and torch.is_tensor(intrinsics) and intrinsics.is_cuda:
                self._cuda_intrinsics(intrinsics, stream=fe)   # (behind the caller's stream, ahead of this frame's "done")
            if not ahead:
                self._gate_wait(fe)
            with torch.cuda.stream(fe):
                out = self.network.patchify(input_=input_, patches_per_image=self.cfg.PATCHES_PER_FRAME,
                                            event_bias=self.event_bias, reinit_hidden=False,
                                            pre_replay=(lambda: self._gate_wait(fe)) if ahead else self._fe_delay)
            if getattr(self, "_in_event_pending", False):
                self._ev_taken.record(fe)
                self._taken_recorded = True
            if _FE_WAIT_PROBE:                      # (diagnostic: how long the main queue waits for the front end)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
                self._fe_done_to(fe, cur, dv)
                b.record(cur)
                self.fe_wait_pairs.append((a, b))
            else:
                self._fe_done_to(fe, cur, dv)
            if _WARM and not dv.fp32:
                # (behind the event: the frame does not wait for it) the window's correlation planes back into the
                # memory-side cache while the previous frame's bundle adjustment is still running
                with torch.cuda.stream(fe):
                    dv.warm()
        else:
            out = self.network.patchify(input_=input_, patches_per_image=self.cfg.PATCHES_PER_FRAME,
                                        event_bias=self.event_bias, reinit_hidden=False)
        if not accepts:
            return      # events only: the encoder state has advanced, the VO has not
        patches = out[3]
        ex = getattr(self.network.patchify, "_extra", None)
        if patches is None or not dv.bind_front_end(ex, patches):
            self.settle()                           # outputs the one-launch commit cannot take: host-driven frame
            return self._track_tail(tstamp, out, intrinsics)
        if not dv.fp32 or dv.x3:
            dv.bind_weights(self.network.update.fused(self.dtype))
        kq, k_dev = self._intrinsics_row(intrinsics, True)
        k_new = None
        if k_dev is not None:
            if kq is None:
                k_new = k_dev                           # a CUDA `intrinsics`: converted on the device (_cuda_intrinsics)
            else:
                dv.k_new.copy_(k_dev)
                k_new = dv.k_new
            self._last_K, self._last_K_raw = kq, self._K_raw_now
        self.tlist.append(tstamp)
        if dv.fp32 and not dv.x3:
            # MIXED_PRECISION off with RAMP_X3=0: the update operator's Linear layers are library GEMMs, issued from here between the two
            # halves of the step -- on the launch bound's rows, sizes never read back (csrc/track.hip RAMP_TRACK_UPDATE_PRE /
            # _POST); everything else as in the fp16 step
            Eb = dv.factor_bound(self.counter)
            dv.step(self.counter, track_dev.COMMIT | track_dev.UPDATE_PRE, k_new=k_new,
                    E_bound=Eb)
            # the GEMMs take their row count from the host: wait (polling pinned memory, no device call) for the sizes the
            # previous frame's plan wrote -- the GPU is busy with the correlation launch enqueued above meanwhile
            E = int(dv.wait_frame(self.counter - 1)[track_dev.DYN_E])
            self._device_operator_fp32(dv, E)
            if self.inputs_ready:
                self._ev_gate.record(cur)                        # the next frame's front end may start (next to BA)
            dv.step(self.counter, track_dev.UPDATE_POST | track_dev.KEYFRAME, E_bound=E)
            self._gate_armed, self._gate_by_flag = self.inputs_ready, False
            self.counter += 1
            return
        sig = self._gate_signal() if self.inputs_ready else None
        if sig is not None:
            self._gate_seq += 1
        dv.step(self.counter, track_dev.COMMIT | track_dev.UPDATE | track_dev.KEYFRAME | self._extra_step_flags,
                k_new=k_new,
                gate_event=self._ev_gate.cuda_event if (self.inputs_ready and sig is None) else None,
                gate_flag=sig.ptr if sig is not None else None, gate_seq=self._gate_seq)
        self._gate_armed, self._gate_by_flag = self.inputs_ready, sig is not None
        self.counter += 1

    def _device_operator_fp32(self, dv, Eb):
        """the update operator of a device-resident fp32 step (reference :280-297) on the first Eb rows of the device-side
        factor list: GEMMs through torch (hipBLASLt) + the row kernels of csrc/update.hip, exactly the host-driven path's
        calls; rows between the live factor count and Eb are defined dummies (zero state row, no neighbour, group 0)"""
        fu = self.network.update.fused(torch.float32)
        g = dv.graph[dv.cur]
        coords = dv.coords[:Eb]
        prev, work, out = dv.net[0], dv.net[1], dv.net[2]
        out32, relu_t = fu.hidden(prev, self.imap_.view(-1, self.DIM), g[2, :Eb], self.M * self.mem, dv.corr[:Eb],
                                  dv.plan_view(Eb), net_map=g[3, :Eb], net32_buf=work, out32_buf=out)
        target, weight, _ = fu.target_weight(fu.heads(relu_t), coords, self.wd // 4, self.ht // 4)
        dv.target[:Eb].copy_(target[0])
        dv.weight[:Eb].copy_(weight[0])
        # the new hidden state is dv.net[0] from here on (the edit kernel's row map indexes it)
        dv.net[0], dv.net[2] = dv.net[2], dv.net[0]
        dv.t.net[0], dv.t.net[2] = dv.net[0].data_ptr(), dv.net[2].data_ptr()

    def _enter_device(self):
        """hand the state over to the device-resident step if this tracker / configuration supports it and the
        optimisation window is full: BA's system then has a fixed size (OPTIMIZATION_WINDOW poses) and the two paths
        are bit-identical.  A configuration whose optimisation window is longer than its removal window (BASELINE
        configs[4]: 32 vs 22) may never fill it; there the hand-over happens once the removal window is full and BA runs
        the OPTIMIZATION_WINDOW-pose system with the unused pose slots identity-damped (dX = 0 for them: the same
        Gauss-Newton step up to fp32 rounding, csrc/ba.hip::ba_edge_kernel)"""
        cfg = self.cfg
        if not (self.device_steps and self.is_initialized
                and self._n >= min(cfg.OPTIMIZATION_WINDOW, cfg.REMOVAL_WINDOW + 1)):
            return                                   # (not yet: the window is still filling)
        if self.device.type != "cuda":
            return                                   # (oracle.host_cpu.RampVoCPU, the tests' CPU twin)
        why = None
        if self.enable_timing:
            why = "enable_timing is set (its per-stage timers synchronise every frame)"
        elif not track_dev.supported(self):
            why = track_dev.unsupported_reason(self)
        elif not getattr(self.network.patchify, "_graphs", None):
            why = "the front end is not running from its captured graph"
        if why is not None:
            if not self._warned_slow:
                self._warned_slow = True
                warnings.warn("rampvo_amd: the device-resident tracking step is not available -- %s; frames run host driven "
                              "(same results, several times slower)" % why)
            return
        if self._dev is None:
            self._dev = track_dev.DeviceTrack(self)
        n1 = self._n + 1
        ok = self._dev.enter(self._hii, self._hjj, self._hkk, self._net_rows(), self._new_edges(n1),
                             self._net_buf[0], self._n)
        if ok:
            self._gate_armed = False
            # the device owns these now (their public names settle() first)
            self._plan = None

    # ----------------------------------------------------------- host-driven frame
    def _track(self, tstamp, input_, intrinsics):
        if getattr(self, "_in_event_pending", False):
            self._cur().wait_event(self._ev_in)     # inputs_ready = "stream": this frame's front end runs on the main stream
        out = self.network.patchify(
            input_=input_, patches_per_image=self.cfg.PATCHES_PER_FRAME, event_bias=self.event_bias,
            reinit_hidden=True if tstamp == 0 else False)
        mask = input_[2]
        if mask is not None and not mask:
            return      # events only: the encoder state has advanced, the VO has not
        return self._track_tail(tstamp, out, intrinsics)

    def _track_tail(self, tstamp, out, intrinsics):
        fmap, gmap, imap, patches, _, clr = out
        kq, k_dev = self._intrinsics_row(intrinsics, True)
        n = self.n
        self.tlist.append(tstamp)
        del self._tstamps[n:]
        self._tstamps.append(self.counter)
        ex = getattr(self.network.patchify, "_extra", None)
        slot = n % self.mem
        # (the eligibility of the front end's outputs for the one-launch commit is checked once per output set: the
        # hipGraph's static buffers are the same objects every frame)
        fc = self._fc_plan
        if ex is None or fc is None or fc[0] is not ex or fc[1] is not patches:
            ok = (ex is not None and ex["fmap"].dtype == self.dtype
                  and ex["chunked"] == self._chunked and patches.is_contiguous()
                  and patches.dtype == torch.float32 and ops.depth_median_supported(3, self.M, self.P)
                  and all(ex[k].data_ptr() % 16 == 0 for k in ("colors", "imap", "gmap", "fmap", "fmap2")))
            plan_fc = None
            if ok:
                plan_fc = ops.FrameCommitPlan([ex["colors"], ex["imap"], ex["gmap"], ex["fmap"], ex["fmap2"]],
                                              [self.colors_, self.imap_, self.gmap_, self.fmap1_, self.fmap2_],
                                              self.patches_)
            fc = self._fc_plan = (ex, patches, plan_fc)
        if fc[2] is not None and (not self.is_initialized or n >= 3):
            # bookkeeping writes, depth initialisation and all state stores of the frame: ONE launch
            copy_k = k_dev is None and n > 0
            if not copy_k:
                self.intrinsics_[n] = k_dev if k_dev is not None else self._upload(kq.astype(np.float32))
                self._last_K, self._last_K_raw = kq, self._K_raw_now
            motion = 0 if n <= 1 else (1 if self.cfg.MOTION_MODEL == 'DAMPED_LINEAR' else 2)
            if not self.is_initialized:
                patches[:, :, 2] = self._initial_depth(patches)       # reference :369; replaced by the median later
            fc[2].run(self.poses_, n, motion, self.cfg.MOTION_DAMPING, self.tstamps_, self.counter,
                      self.index_map_, self.m + self.M, self.intrinsics_, copy_k, self.patches_,
                      3 if self.is_initialized else 0, patches, (n, slot, slot, slot, slot))
        else:
            self._frame_stores_stepwise(n, slot, k_dev, kq, patches, imap, gmap, fmap, clr, ex)
        self._last_K_row = n                         # row n now holds _last_K (written or copied from row n-1)
        self.counter += 1
        if n > 0 and not self.is_initialized:
            if self.motion_probe() < 2.0:
                self.delta[self.counter - 1] = (self.counter - 2, SE3.Identity(1, device=self.device)[0])
                return

        self.n += 1
        self.m += self.M
        kf, jf = self.__edges_forw()
        kb, jb = self.__edges_back()
        self.append_factors(np.concatenate([kf, kb]), np.concatenate([jf, jb]))

        if self.n == 8 and not self.is_initialized:
            self.is_initialized = True
            for _ in range(12):
                self.update()
        elif self.is_initialized:
            self.update()
            self.keyframe()
            self._enter_device()

    # -------------------------------------------------------- pose prediction
    def _virtual_frame(self, last_keyframe_number):
        """graph, poses and intrinsics extended by a virtual keyframe whose pose is the motion model's
        (reference :416-444 / :452-479)"""
        from .pose_prediction.pose_pred_utils import add_forward_elements, motion_bootstrap
        self.settle()
        next_frame_number = last_keyframe_number + 1       # number starting from 1
        next_frame_index = next_frame_number - 1           # index starting from 0
        poses = self.poses.clone()
        poses[:, next_frame_index] = motion_bootstrap(poses=poses[0, ...], n=self.n,
                                                      MOTION_MODEL=self.cfg.MOTION_MODEL,
                                                      MOTION_DAMPING=self.cfg.MOTION_DAMPING)
        intrinsics = self.intrinsics.clone()
        intrinsics[:, next_frame_index] = intrinsics[:, next_frame_index - 1]
        patches = self.patches.clone()
        ii, jj, kk, weights_up = add_forward_elements(
            frame_num=next_frame_number, patch_extracted_num=self.M, ii=self.ii, jj=self.jj, kk=self.kk, ix=self.ix,
            r=self.cfg.PATCH_LIFETIME, weights=self.last_weight.clone())
        coords = self.reproject(indicies=(ii, jj, kk), poses=poses, patches=patches, intrinsics=intrinsics)
        return next_frame_number, next_frame_index, poses, patches, intrinsics, (ii, jj, kk), weights_up, coords

    def efficient_pose_prediction(self, sec_to_pred_future, abs_time, last_keyframe_number, deg=3, frequency=30):
        """reference :416-444: builds the virtual frame and its reprojections and stops there (returns None)"""
        self._virtual_frame(last_keyframe_number)

    def predict_future_pose(self, sec_to_pred_future, abs_time, last_keyframe_number, deg=3, frequency=30, corrected=False):
        """Extrapolate a virtual keyframe ``sec_to_pred_future`` frames past the last real one (reference
        :447-507, driven by evaluate.py::run_pose_pred): motion-model pose, one extra factor per live patch,
        per-patch spline models of the past reprojections (fitted once, on the first call), two BA iterations
        on the predicted targets, then the pose is appended so that terminate() interpolates through it.

        Default: WHAT UPSTREAM COMPUTES (pinned by tests/golden/pose_pred_e2e.npz, upstream's own method run over the
        oracle): the predicted 3x3 grids are written with x and y exchanged (pose_pred_utils.py:342) and the whole
        ``coords`` tensor [1,E,2,3,3] is handed to BA as ``target``, which cuda_ba views as [-1,2] and reads the first
        E rows of (fastba/ba_cuda.cu:462) -- pairs of neighbouring grid values of the first E/9 factors.
        ``corrected=True``: BA gets what Ramp_vo.update() gives it -- the patch centres [1,E,2], channel 0 = x --
        for the same factors and weights."""
        from .pose_prediction.pose_pred_utils import (compute_patch_track__, fit_model_patch_track,
                                                      predict_patch_on_model)
        (next_frame_number, next_frame_index, poses, patches, intrinsics, (ii, jj, kk), weights_up,
         coords) = self._virtual_frame(last_keyframe_number)
        if self.patch_dict_ is None:
            self.patch_dict_ = compute_patch_track__(coords=coords, ii=ii, jj=jj, kk=kk,
                                                     image_to_proj=next_frame_index)
        if self.patches_models is None:
            self.patches_models = fit_model_patch_track(
                next_frame_index=next_frame_index, patch_dict=self.patch_dict_, img_to_keyframe_map=self.tstamps_,
                ii=ii, jj=jj, data_shape=(self.ht, self.wd), frequency=frequency, deg=deg)
        coords, updated_weight = predict_patch_on_model(
            patch_models=self.patches_models, step_to_pred_future=sec_to_pred_future, frequency=frequency,
            next_frame_index=next_frame_index, coords=coords, weights=weights_up, ii=ii, jj=jj, kk=kk,
            reference_layout=not corrected)
        E = ii.shape[0]
        if corrected:
            target = coords[..., self.P // 2, self.P // 2].contiguous()
        else:
            target = coords.contiguous().view(-1, 2)[:E].view(1, E, 2).contiguous()     # cuda_ba's reading of the tensor
        t0 = max(next_frame_number - self.cfg.OPTIMIZATION_WINDOW if self.is_initialized else 1, 1)
        t1 = next_frame_number
        try:
            fastba.BA(poses, patches, intrinsics, target, updated_weight.contiguous(), self.lmbda, ii, jj, kk, t0, t1,
                      M=self.M, iterations=2, eff_impl=False, info=self._ba_info)
        except Exception as e:
            print(f"WARNING: BA failed...{e}")
        self.update_attributes(abs_time=abs_time, next_frame_index=next_frame_index, poses=poses)

    def update_attributes(self, abs_time, next_frame_index, poses):
        """expose the virtual pose to terminate() (reference :510-519)"""
        assert self._tstamps[self.n - 1] != 0 if self._tstamps else int(self.tstamps_[self.n - 1]) != 0
        self.tstamps_[self.n] = abs_time
        del self._tstamps[self.n:]
        self._tstamps.append(int(abs_time))
        self.poses_[self.n] = poses[0, next_frame_index]
        self.tlist.append(abs_time)
        self.counter += 1
        self.n += 1

    def remove_attributes(self, corrected=False):
        """undo update_attributes (reference :521-528).  Default: as upstream, whose ``poses_[:,6] = 1.0`` rewrites the qw
        of EVERY row of the pose buffer (the real keyframes' too); corrected=True resets only the removed row."""
        self.n -= 1
        self.counter -= 1
        self.tlist.pop()
        del self._tstamps[self.n:]
        self.poses_[self.n] = torch.zeros(7, dtype=torch.float, device=self.device)
        if corrected:
            self.poses_[self.n, 6] = 1.0
        else:
            self.poses_[:, 6] = 1.0
        self.tstamps_[self.n] = 0

    def _frame_stores_stepwise(self, n, slot, k_dev, kq, patches, imap, gmap, fmap, clr, ex):
        """the same writes as ops.frame_commit, a few launches instead of one (first frames, odd shapes)"""
        # time stamp, index map, intrinsics row and motion-model pose: one launch
        copy_k = k_dev is None and n > 0
        if not copy_k:
            self.intrinsics_[n] = k_dev if k_dev is not None else self._upload(kq.astype(np.float32))
            self._last_K, self._last_K_raw = kq, self._K_raw_now
        motion = 0 if n <= 1 else (1 if self.cfg.MOTION_MODEL == 'DAMPED_LINEAR' else 2)
        ops.frame_begin(self.poses_, n, motion, self.cfg.MOTION_DAMPING, self.tstamps_, self.counter,
                        self.index_map_, self.m + self.M, self.intrinsics_, copy_k)
        # reference :369-372: random inverse depths, replaced by the median of the last three keyframes once the
        # tracker is initialised -- the draw is dead then and is not made (nothing else consumes the generator)
        if self.is_initialized:
            if patches.is_contiguous() and patches.dtype == torch.float32 and ops.depth_median_supported(3, self.M, self.P):
                ops.depth_median_fill(self.patches_, n, 3, patches[0])       # radix select + fill, one launch
            else:
                patches[:, :, 2] = torch.median(self.patches_[n - 3:n, :, 2])
        else:
            patches[:, :, 2] = self._initial_depth(patches)
        if (ex is not None and ex["fmap"].dtype == self.dtype and (self.M * 3) % 4 == 0 and patches.is_contiguous()
                and ex["chunked"] == self._chunked):
            # one launch: patches, colours and the four feature tensors into their state rows / ring slots
            ops.store_rows([patches, ex["colors"], ex["imap"], ex["gmap"], ex["fmap"], ex["fmap2"]],
                           [(self.patches_, n), (self.colors_, n), (self.imap_, slot), (self.gmap_, slot),
                            (self.fmap1_, slot), (self.fmap2_, slot)])
        else:
            clr = (clr[0][:, [2, 1, 0]] + 0.5) * (255.0 / 2)
            self.colors_[n] = clr.to(torch.uint8)
            self.patches_[n] = patches
            self.imap_[slot] = imap.reshape(self.M, self.DIM).to(self.dtype)
            self.gmap_[slot] = gmap[0].permute(0, 2, 3, 1).to(self.dtype)
            f = fmap[0]                                                  # [1,128,h,w], channels-last storage
            if self._chunked:
                ops.pyramid_pack(f[0].permute(1, 2, 0).to(self.dtype).contiguous(), self.fmap1_[slot],
                                 self.fmap2_[slot], split=self._split)
            else:
                self.fmap1_[slot] = f[0].permute(1, 2, 0).to(self.dtype)
                self.fmap2_[slot] = F.avg_pool2d(f, 4, 4)[0].permute(1, 2, 0).to(self.dtype)

    def _initial_depth(self, patches):
        """reference :369 -- torch.rand_like; overridable so parity tests can inject the
        oracle's draw"""
        return torch.rand_like(patches[:, :, 2, 0, 0, None, None])
