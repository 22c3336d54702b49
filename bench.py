#!/usr/bin/env python
"""bench.py -- keyframes/sec of the RAMP-VO tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): SingleScale encoder, 640x480 synthetic
event+frame stream, 96 patches/frame, default.yaml windows (lifetime 13, removal
22, optimisation 10), 2 BA iterations per keyframe.  One *step* = one
``Ramp_vo.__call__`` on an initialised tracker = encoder + patchify + reproject +
corr + update operator + 2 GN iterations + keyframe management.  The tracker is
first primed (untimed) until its sliding window is full, so the timed steps run at
the steady-state graph size (E ~ 40k edges), then stepped for >= 0.6 s (untimed,
``--clock-warm-s``) so that even a 20-step timed region runs at settled clocks.
Inputs for all steps are generated and resident in HBM before the timed region.

N > 1: one process per GPU (torchrun), every rank tracks its own independent
sequence (different seed) -- the path shards by sequence only; the single
collective is an all_gather of per-rank metrics.  value = total steps of all ranks
/ max-over-ranks time.

Prints ONE JSON line on rank 0, including
  roofline     the fused correlation kernel against the HBM roof: COMPULSORY bytes per launch
               (every referenced feature plane / patch read once, every output written once:
               frac <= 1 by construction) / mean launch time (HIP events on the launch stream,
               inside the timed region); the SURVEY 8(d) per-edge model and the PMC-measured
               fabric traffic ride along as model_bytes / traffic, its MFMA rate as mfma_*
  roofline_update / roofline_encoder / roofline_ba   the other three stages, same clock
  parity       measured in this process after the timed region: ONE teacher-forced update()
               from the run's own state snapshot -- HIP fp16 (the benchmarked path), HIP fp32
               and the CPU oracle backend -- and a free-running trajectory (damped weights)
               against the reference's own run (tests/golden/ramp_vo_traj_ss.npz): ATE + rel
  cpu_baseline the same steady-state step on the host cores through the CPU oracle
               ("port"), a bounded sample from the identical state
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16 MFMA, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.0    # fp32 MFMA (the --mixed 0 towers)
UPDATE_FLOP_PER_EDGE = 5.40e6   # 2.70 MMAC / edge in the update operator's 18 Linear layers
CORR_FLOP_PER_EDGE_LEVEL = 2 * 128 * 576    # 9 px x 64 window positions, 128-long dots


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prime", type=int, default=70, help="untimed frames to fill the sliding window")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--patches", type=int, default=96)
    ap.add_argument("--mode", default="SingleScale")
    ap.add_argument("--preset", default="default", choices=("default", "precise", "fast"),
                    help="config_vo preset for the BA / lifetime windows (configs[2] uses precise)")
    ap.add_argument("--opt-window", type=int, default=0, help="override OPTIMIZATION_WINDOW (configs[4]: 32)")
    ap.add_argument("--encoder-fp8", type=int, default=0,
                    help="1: the conv towers' products on the fp8 MFMA (configs[4]: 'fp16 encoder on fp8 MFMA'); needs --mixed 1")
    ap.add_argument("--mixed", type=int, default=1,
                    help="1 (default.yaml's MIXED_PRECISION: True): fp16 features / conv + GEMM I/O with fp32 "
                         "accumulation, fp32 hidden state, BA and geometry; 0: fp32 everywhere")
    ap.add_argument("--cpu-steps", type=int, default=6, help="steps of the CPU baseline sample, ~2 s each (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: the synthetic frames are resident in HBM before their step (the bench contract), so the "
                         "tracker may launch frame t+1's front end while frame t's bundle adjustment drains "
                         "(Ramp_vo.inputs_ready); 0: strictly one frame at a time")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--probe-every", type=int, default=4,
                    help="the correlation launch is bracketed by HIP events on every N-th timed step (an event record "
                         "is a ~6 us bubble on the stream it is recorded on)")
    ap.add_argument("--inst-steps", type=int, default=100,
                    help="instrumented steps behind the timed region: events around the update operator, bundle adjustment "
                         "and the front end (roofline_update / roofline_ba / roofline_encoder)")
    ap.add_argument("--clock-warm-s", type=float, default=0.6,
                    help="untimed steady-state steps for at least this long right before the warm-up steps, so that a "
                         "short timed region (the driver's --steps 20 is ~30 ms) runs at settled clocks")
    ap.add_argument("--clock-warm-max", type=int, default=480, help="cap of the clock-warm phase in frames")
    ap.add_argument("--handback-frames", type=int, default=24,
                    help="frames between the state_dict() hand-back and the converged / live legs (reported as "
                         "legs.after_handback)")
    ap.add_argument("--fp32-leg", type=int, default=60,
                    help="(with --mixed 1, rank 0, N = 1) timed steps of the same workload with MIXED_PRECISION off, run as "
                         "`bench.py --mixed 0` in a process of its own behind everything else -> config.fp32_kfps (0 = skip)")
    ap.add_argument("--parity", type=int, default=1,
                    help="1: after the timed region, one teacher-forced update() from the run's snapshot on HIP fp16 / "
                         "HIP fp32 / the CPU oracle, and the free-running trajectory check (rank 0, N = 1)")
    ap.add_argument("--config", type=int, default=0, choices=(0, 1, 2, 3, 4),
                    help="BASELINE.json configs[i] as written (sets mode / preset / size / patches / windows; explicit flags "
                         "still override): 1 = SingleScale 640x480 M=96 default.yaml (the default workload), 2 and 3 = "
                         "MultiScale M=96 precise.yaml windows (3: one such sequence per rank), 4 = MultiScale 1280x720 M=256, "
                         "32-keyframe optimisation window, precise lifetimes, fp8-MFMA encoder, every frame kept a keyframe")
    ap.add_argument("--keyframe-thresh", type=float, default=None,
                    help="override KEYFRAME_THRESH (0: every frame stays a keyframe, so the sliding window fills to its "
                         "bound whatever the random-init weights' motion test says: configs[2..4] as written)")
    ap.add_argument("--live-steps", type=int, default=40,
                    help="steps of the live-factor leg behind everything else (roofline_live): every reprojection is moved "
                         "into the target plane before the correlation launch, so that every factor gathers (0 = skip)")
    ap.add_argument("--np-steps", type=int, default=40,
                    help="extra steps with frame pipelining OFF after the timed region, reported as "
                         "config.non_pipelined_kfps (what evaluate.run's strictly sequential loop gets); 0 = skip")
    args = ap.parse_args()
    given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
    preset_of = {2: dict(mode="MultiScale", preset="precise", prime=160),
                 3: dict(mode="MultiScale", preset="precise", prime=160),
                 4: dict(mode="MultiScale", preset="precise", height=720, width=1280, patches=256, opt_window=32,
                         encoder_fp8=1, keyframe_thresh=0.0, prime=90, clock_warm_max=60, inst_steps=40, np_steps=20)}
    for k, v in preset_of.get(args.config, {}).items():
        if "--" + k.replace("_", "-") not in given:
            setattr(args, k, v)
    return args


def _event_pair():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


class CorrTimer:
    """HIP event pairs around every fused-correlation launch (torch's current stream is the
    stream the C ABI launches on)"""

    def __init__(self):
        self.pairs = []
        self.edges = []
        self.enabled = False

    def install(self):
        from rampvo_amd import altcorr
        inner = altcorr.corr_pyramid
        timer = self

        def timed(gmap, pyramid, coords, ii, jj, *a, **k):
            if not timer.enabled:
                return inner(gmap, pyramid, coords, ii, jj, *a, **k)
            s, e = _event_pair()
            s.record()
            out = inner(gmap, pyramid, coords, ii, jj, *a, **k)
            e.record()
            timer.pairs.append((s, e))
            timer.edges.append(int(ii.shape[0]))
            return out

        altcorr.corr_pyramid = timed
        # the tracker's own per-frame launch goes straight to the C ABI (Ramp_vo._corr_launch)
        from rampvo_amd.Ramp_vo import Ramp_vo
        direct = Ramp_vo._corr_launch

        def timed_direct(slam, coords, ii, jj, order):
            if not timer.enabled:
                return direct(slam, coords, ii, jj, order)
            s, e = _event_pair()
            s.record()
            out = direct(slam, coords, ii, jj, order)
            e.record()
            timer.pairs.append((s, e))
            timer.edges.append(int(ii.shape[0]))
            return out

        Ramp_vo._corr_launch = timed_direct

    @staticmethod
    def pmc_traffic_per_edge(elem_bytes, segment="timed"):
        """fabric-side bytes per edge of the fp16 kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2
        per the guide's gfx950 correction + WRITE_SIZE, separate --pmc passes; the newest
        profiles/pmc/r*_corr_traffic.json, which names its command).  Counters cannot be read inside this process,
        so the per-edge figure of the PMC run of the same workload is scaled by this run's edges per launch.  Since
        round 5 the passes run the bench's own command line to its steady state (profiles/pmc/r05_corr_traffic.json,
        tools/r05_pmc_corr.sh): ``segment`` picks the launches of the timed region or of the all-live legs, and the
        source's own edges per launch ride along (the scaling is then a few per cent, not 23k -> 42k factors)."""
        import glob
        # (fp32 features: the passes of `MIXED=0 tools/r05_pmc_corr.sh` over corr_mfma_kernel<CorrX2>, round 6)
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc", "r*_corr_traffic.json" if elem_bytes == 2 else
                                              "r*_corr_traffic_fp32.json")))
        for f in reversed(files):
            try:
                with open(f) as fh:
                    d = json.load(fh)
                d = d.get(segment, d if segment == "timed" else None)
                if d is None:
                    return None, None, None
                return float(d["bytes_per_edge"]), os.path.relpath(f, ROOT), int(d.get("edges_per_launch", 0))
            except Exception:
                continue
        return None, None, None

    def summary(self, elem_bytes, slam, segment="timed"):
        if not self.pairs:
            return None
        ms = np.array([s.elapsed_time(e) for s, e in self.pairs])
        edges = np.array(self.edges, dtype=np.float64)
        big = edges > 0.5 * edges.max()          # the per-update launches (motion-probe launches are tiny)
        E = float(edges[big].mean())
        mean_ms = float(ms[big].mean())
        sec = mean_ms * 1e-3
        # COMPULSORY bytes of one launch: each referenced target plane (both pyramid levels) and each referenced
        # 3x3x128 patch once, per-edge coordinates + indices + schedule entry, every output row once.  Distinct
        # frames / patches are those of the graph at the end of the timed region (the window is in steady state).
        h, w = slam.ht // slam.RES, slam.wd // slam.RES
        frames = int(len(np.unique(slam._jj % slam.mem)))
        patches = int(len(np.unique(slam._kk % (slam.M * slam.mem))))
        out_row = 896 * elem_bytes
        compulsory = (frames * 128 * (h * w + (h // 4) * (w // 4)) * elem_bytes + patches * 128 * 9 * elem_bytes
                      + E * (72 + 16 + 4) + E * out_row)
        # SURVEY 8(d)'s per-edge model (every edge reads its 10x10 window privately; two levels per launch): counts
        # re-reads that the (jj, ii)-major XCD schedule serves from L2, which is why it can exceed the HBM peak
        per_edge_level = 128 * 9 * elem_bytes + 128 * 100 * elem_bytes + 72 + 16 + 441 * elem_bytes
        model = E * 2 * per_edge_level
        achieved = compulsory / sec / 1e9
        per_edge, src, src_edges = self.pmc_traffic_per_edge(elem_bytes, segment)
        traffic = int(per_edge * E) if per_edge else None
        flops = E * 2 * CORR_FLOP_PER_EDGE_LEVEL
        from rampvo_amd._lib import corr_f32_mode
        f32_mode = corr_f32_mode()            # 2: split fp16 parts, three f16 MFMA products per dot product (3x the flops below)
        peak_tf = MFMA_F16_PEAK_TFLOPS if (elem_bytes == 2 or f32_mode == 2) else MFMA_F32_PEAK_TFLOPS
        if elem_bytes != 2 and f32_mode == 2:
            flops *= 3
        f32_kernel = ("corr_kernel<float>", "corr_mfma_kernel<float>", "corr_mfma_kernel<CorrX2> (fp32 features as split fp16 parts)")[f32_mode]
        out = dict(kernel="corr_mfma_kernel<half>" if elem_bytes == 2 else f32_kernel, bound="hbm",
                   achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                   frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, launches=int(big.sum()),
                   mean_launch_us=round(mean_ms * 1e3, 1), bytes_per_launch=int(compulsory),
                   edges_per_launch=int(E), target_frames=frames, patches=patches,
                   model_bytes=int(model), model_gbps=round(model / sec / 1e9, 1),
                   mfma_tflops=round(flops / sec / 1e12, 1), mfma_frac=round(flops / sec / 1e12 / peak_tf, 4),
                   limiter="vector L1 / texture addresser (window gathers: TCP_GATE_EN1 ~80 %, profiles/pmc), neither "
                           "HBM nor MFMA: achieved = compulsory bytes, model_bytes = SURVEY 8(d) per-edge bytes "
                           "(L2-served re-reads included), traffic = PMC fabric bytes")
        if traffic:
            out["traffic_gbps"] = round(traffic / sec / 1e9, 1)
            out["traffic_frac_of_peak"] = round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4)
            out["traffic_source"] = src
            out["traffic_source_edges_per_launch"] = src_edges     # the PMC pass's own launch size (scaled to this run's)
        return out


class DeviceProbe:
    """The device-resident step is one C call (csrc/track.hip::ramp_track_step): its kernels are not launched from
    Python, so the HIP events of the roofline legs are recorded by the C function itself, on the stream it launches
    on, through the optional ``ramp_track.probe`` handles -- one pre-created event set per timed step."""

    def __init__(self, steps):
        self.sets = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps)]
        for evs in self.sets:
            for ev in evs:
                ev.record()                    # torch creates the hipEvent on the first record
        self.used, self.edges, self.enabled = 0, [], False
        self.mode, self.modes = "all", []              # "corr": only the correlation launch is bracketed (the timed region)
        self.corr_inst_ms = []
        self.live_pairs, self.live_edges = {"live": [], "compact": []}, {"live": [], "compact": []}
        self.fp32, self.cur = False, None       # fp32: the operator is launched from Python (fp32 features with RAMP_X3=0)

    def install(self):
        from rampvo_amd import track_dev
        inner, probe = track_dev.DeviceTrack.step, self

        def step(dv, counter, flags, **k):
            # (an fp32 step is two calls -- PRE: reprojection + correlation, POST: BA + keyframe(); they share one event set)
            starts = flags & (track_dev.UPDATE | track_dev.UPDATE_PRE)
            if starts:
                on = bool(probe.enabled and probe.mode and probe.used < len(probe.sets))
                probe.cur = probe.used if on else None
                if on:
                    probe.edges.append(int(dv.lazy_state()[track_dev.DYN_E]))      # a frame or two old: fine for a mean
                    probe.modes.append(probe.mode)
                    probe.fp32 = bool(getattr(dv, "fp32", False)) and not bool(getattr(dv, "x3", False))   # (RAMP_X3=0)
                    probe.used += 1
            elif not (flags & track_dev.UPDATE_POST):
                probe.cur = None
            idx = getattr(probe, "cur", None)
            for i in range(5):
                dv.t.probe[i] = (probe.sets[idx][i].cuda_event
                                 if idx is not None and (probe.mode not in ("corr", "live", "compact") or i < 2) else None)
            return inner(dv, counter, flags, **k)

        track_dev.DeviceTrack.step = step

    def feed(self, ctimer, utimer, btimer, opt_window):
        for evs, E, mode in zip(self.sets[:self.used], self.edges, self.modes):
            if mode == "corr":                 # the timed region's samples: what roofline.achieved is computed from
                ctimer.pairs.append((evs[0], evs[1])); ctimer.edges.append(E)
            elif mode in ("live", "compact"):  # the live-factor legs (every reprojection inside the plane)
                self.live_pairs[mode].append((evs[0], evs[1])); self.live_edges[mode].append(E)
            elif mode == "alone":              # the sequential pass: nothing else on the GPU
                if not self.fp32:              # (fp32 steps: the operator is launched from Python, timed by UpdateTimer's hook)
                    utimer.alone_ms.append(evs[1].elapsed_time(evs[2]))
                btimer.alone_ms.append(evs[3].elapsed_time(evs[4]))
            else:                              # the instrumented pass behind it
                self.corr_inst_ms.append(evs[0].elapsed_time(evs[1]))
                if not self.fp32:
                    utimer.pairs.append((evs[1], evs[2])); utimer.edges.append(E)
                btimer.pairs.append((evs[3], evs[4])); btimer.meta.append((E, opt_window, 2))


class UpdateTimer:
    """HIP events around the update operator (FusedUpdate.hidden: correlation MLP, neighbour MLPs, SoftAgg x2, gru -- the gru
    launch carries the two heads in its epilogue)"""

    def __init__(self):
        self.pairs, self.edges, self.enabled, self.alone_ms = [], [], False, []

    def install(self):
        from rampvo_amd.update_fused import FusedUpdate
        inner, timer = FusedUpdate.hidden, self

        def timed(fu, net, inp_table, inp_idx, inp_mod, corr, plan, **kw):
            if not timer.enabled:
                return inner(fu, net, inp_table, inp_idx, inp_mod, corr, plan, **kw)
            s, e = _event_pair()
            s.record()
            out = inner(fu, net, inp_table, inp_idx, inp_mod, corr, plan, **kw)
            e.record()
            timer.pairs.append((s, e))
            timer.edges.append(int(corr.shape[0]))
            return out

        FusedUpdate.hidden = timed

    def summary(self, mixed):
        if not self.pairs:
            return None
        ms = np.array([s.elapsed_time(e) for s, e in self.pairs])
        edges = np.array(self.edges, dtype=np.float64)
        big = edges > 0.5 * edges.max()
        E, mean_ms = float(edges[big].mean()), float(ms[big].mean())
        x3 = (not mixed) and os.environ.get("RAMP_X3", "1") == "1"
        # fp32 features: every product is three f16 MFMA products (split operands, csrc/update_x3.hip): the bound of the
        # ALGORITHMIC flops is a third of the f16 peak (RAMP_X3=0: library GEMMs on the f32 MFMA)
        peak = MFMA_F16_PEAK_TFLOPS if mixed else (round(MFMA_F16_PEAK_TFLOPS / 3.0, 1) if x3 else MFMA_F32_PEAK_TFLOPS)
        ach = E * UPDATE_FLOP_PER_EDGE / (mean_ms * 1e-3) / 1e12
        out = dict(kernel=("update operator (fp32 accuracy, f16x3): x3_corr_mlp + x3_nbr x2 + (x3_fg + x3_segment_softmax + x3_linear) x2 + x3_gru"
                           if x3 else "update operator: upd_corr_mlp + upd_nbr x2 + (upd_fg + segment softmax + h GEMM) x2 + upd_gru"),
                   bound="mfma", achieved=round(ach, 1), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                   mean_call_us=round(mean_ms * 1e3, 1), edges=int(E), flop_per_edge=UPDATE_FLOP_PER_EDGE,
                   note="18 Linear layers of 384 (one of K = 882) per edge, SURVEY 8(d)'s 5.40 MFLOP/edge; the "
                        "time is the whole operator incl. its row-wise LayerNorm / gate / softmax passes (pipelined: "
                        "its gru launch shares the chip with the next frame's front end)")
        if self.alone_ms:
            a = float(np.mean(self.alone_ms))
            out["mean_call_us_alone"] = round(a * 1e3, 1)
            out["frac_alone"] = round(E * UPDATE_FLOP_PER_EDGE / (a * 1e-3) / 1e12 / peak, 4)
        return out


class BaTimer:
    """HIP events around fastba.BA (two Gauss-Newton iterations)"""

    def __init__(self):
        self.pairs, self.meta, self.enabled, self.alone_ms = [], [], False, []

    def install(self):
        from rampvo_amd import fastba
        inner, timer = fastba.BA, self

        def timed(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k):
            if not timer.enabled:
                return inner(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k)
            s, e = _event_pair()
            s.record()
            out = inner(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k)
            e.record()
            timer.pairs.append((s, e))
            timer.meta.append((int(ii.shape[0]), int(t1 - t0), int(k.get("iterations", 2))))
            return out

        fastba.BA = timed

    def summary(self, patches_per_frame, removal_window):
        if not self.pairs:
            return None
        ms = float(np.mean([s.elapsed_time(e) for s, e in self.pairs]))
        E, N, it = (float(np.mean([m[i] for m in self.meta])) for i in range(3))
        mu = patches_per_frame * removal_window
        # 116 B per edge per iteration in, + per-iteration outputs (6N)^2 + 6N*Mu + 2*Mu + 6N floats
        nbytes = it * (116.0 * E + 4.0 * ((6 * N) ** 2 + 6 * N * mu + 2 * mu + 6 * N))
        ach = nbytes / (ms * 1e-3) / 1e9
        out = dict(kernel="fastba.BA (2 Gauss-Newton iterations)",
                   bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                   traffic=None, mean_call_us=round(ms * 1e3, 1), bytes_per_call=int(nbytes), edges=int(E),
                   free_poses=int(N), note="a short dependent chain on ~10 MB: latency, not bandwidth, bounds it "
                                           "(time includes the overlap with the next frame's front end when pipelining)")
        if self.alone_ms:
            out["mean_call_us_alone"] = round(float(np.mean(self.alone_ms)) * 1e3, 1)
        return out


class EncoderTimer:
    """HIP events around the encoder front end (one hipGraph replay per frame: fused LSTM + conv towers
    + patch selection + gathers) and the conv FLOPs of one frame, counted from the shapes of the
    conv launches of an eager (un-captured) frame."""

    def __init__(self):
        self.pairs, self.alone, self.enabled = [], [], False
        self.flops_frame, self._acc = 0, 0

    def install(self, net):
        from rampvo_amd import conv_hip
        timer, towers_inner = self, conv_hip.conv2d_towers

        def counted(jobs, half, *a, **k):             # every conv of the towers goes through conv2d_towers
            ys = towers_inner(jobs, half, *a, **k)
            for j, y in zip(jobs, ys):
                o = y.raw if isinstance(y, conv_hip.Pending) else y
                cout, cin, kh, kw = j["conv"].weight.shape
                timer._acc += 2 * o.shape[0] * o.shape[1] * cout * cin * kh * kw
            return ys

        conv_hip.conv2d_towers = counted
        pat = net.patchify
        impl_inner = pat._forward_impl

        def impl(*a, **k):
            timer._acc = 0
            r = impl_inner(*a, **k)
            timer.flops_frame = max(timer.flops_frame, timer._acc)
            return r

        replay_inner = pat._run_graph

        def replay(graph):                 # the captured front end (the staging copies and the patch selection run ahead of it)
            if not timer.enabled:
                return replay_inner(graph)
            s, e = _event_pair()
            s.record()
            replay_inner(graph)
            e.record()
            timer.pairs.append((s, e))

        pat._forward_impl, pat._run_graph = impl, replay

    def summary(self, mixed):
        if not self.pairs or not self.flops_frame:
            return None
        # the replay ALONE (the strictly sequential pass: nothing else on the GPU) when that pass ran, and next to the
        # previous frame's gru chain + bundle adjustment (the pipelined instrumented pass), where it shares the chip
        ms_frame = float(np.mean([s.elapsed_time(e) for s, e in self.pairs]))
        ms = float(np.mean([s.elapsed_time(e) for s, e in self.alone])) if self.alone else ms_frame
        peak = MFMA_F16_PEAK_TFLOPS if mixed else MFMA_F32_PEAK_TFLOPS
        ach = self.flops_frame / (ms * 1e-3) / 1e12
        out = dict(kernel="encoder front end (fused LSTM + conv towers + gathers, 1 hipGraph; the patch selection runs ahead of it)",
                   bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 5),
                   conv_gflop_per_frame=round(self.flops_frame / 1e9, 2), mean_front_end_us=round(ms * 1e3, 1),
                   note="32/64-channel layers: arithmetic intensity is below the machine balance, the towers are "
                        "latency/HBM bound; when pipelining the front end overlaps the previous frame's gru chain and BA")
        if self.alone:
            out["mean_front_end_us_next_to_the_tail"] = round(ms_frame * 1e3, 1)
            out["mean_front_end_us_is"] = "the graph replay alone (the %d-step sequential pass)" % len(self.alone)
        return out


PARITY_W_BIAS = -14.0   # see parity_block()


def _cpu_tracker(state, args, cfg_kwargs, **net_kw):
    """the CPU oracle backend's tracker loaded with a state snapshot (inside cpu_oracle_ops())"""
    from oracle.backend_cpu import Ramp_vo
    from rampvo_amd.config import make_cfg
    from rampvo_amd.synthetic import make_network
    cfg_kwargs = dict(cfg_kwargs, MIXED_PRECISION=False)       # the host path is fp32 throughout
    state = dict(state)
    for k in ("net", "imap", "gmap", "fmap1", "fmap2"):
        state[k] = state[k].float()
    net = make_network(args.mode, device="cpu", **net_kw)
    slam = Ramp_vo(make_cfg(args.preset, **cfg_kwargs), net, {"event_bias": True}, ht=args.height, wd=args.width,
                   device="cpu")
    slam.load_state_dict(state)
    return slam


def cpu_baseline(state, args, cfg_kwargs, frames, steps):
    """time `steps` tracker steps on the host cores from the same steady-state snapshot, through
    the CPU oracle backend (torch CPU for the encoder / update GEMMs, oracle C for the natives)"""
    from oracle.backend_cpu import cpu_oracle_ops
    # the CPUs this process may really use (affinity and the container's bandwidth quota, not the machine's count: a team
    # larger than the quota is frozen by the kernel for most of every period and measures that)
    from rampvo_amd import hostenv
    cores = min(hostenv.cpu_quota(), int(os.environ.get("RAMP_CPU_THREADS", "64")))
    torch.set_num_threads(cores)      # also the OpenMP team of the oracle's edge-parallel loops
    with cpu_oracle_ops():
        slam = _cpu_tracker(state, args, cfg_kwargs)
        # encoder recurrent state is not part of the VO snapshot: one untimed step re-seeds it
        t0 = state["counter"]
        im, ev, K, mask = frames[0]
        slam(t0, input_tensor=(ev, im, mask), intrinsics=K)
        tic = time.perf_counter()
        for i in range(steps):
            im, ev, K, mask = frames[1 + i]
            slam(t0 + 1 + i, input_tensor=(ev, im, mask), intrinsics=K)
        dt = time.perf_counter() - tic
    return dict(value=round(steps / dt, 4), unit="keyframes/s", cores=cores, kind="port",
                sample="%d steady-state steps (E~%d edges) from the GPU run's state snapshot; torch-CPU fp32 "
                       "encoder/update GEMMs + OpenMP oracle C natives on %d threads (host has %d logical cores, this "
                       "process may use %d)" % (steps, len(slam._ii), cores, os.cpu_count() or 1, hostenv.cpu_quota()),
                s_per_step=round(dt / steps, 3))


@torch.no_grad()
def parity_block(state, args, cfg_kwargs, net, dev):
    """ONE teacher-forced update() (reproject -> corr -> update operator -> BA x2) from the run's own steady-state
    snapshot: HIP fp16 (the benchmarked path) and HIP fp32 against the CPU oracle backend (fp32), plus the
    free-running trajectory of the damped-weight tracker against the reference's own run (committed fixture).

    The compared step runs on all three legs with the run's weights except for the confidence head's bias, shifted by
    PARITY_W_BIAS: with the throughput profile's confidences (~0.5) on 2-40 px residuals Gauss-Newton solves depths
    with Q = 1/(C + 1e-4) up to 1e4 on patches seen over almost no baseline, and a 1e-7 input difference moves such
    a depth by 1e-3 -- the conditioning of the random-weight problem, not kernel accuracy."""
    from oracle.backend_cpu import cpu_oracle_ops
    from rampvo_amd.config import make_cfg
    from rampvo_amd.Ramp_vo import Ramp_vo
    from rampvo_amd.synthetic import make_network
    del net
    net = make_network(args.mode, device=dev, w_bias=PARITY_W_BIAS)
    n = int(state["n"])
    before = state["poses"][:n].numpy().copy()
    from rampvo_amd import hostenv
    host_pool = torch.get_num_threads()
    torch.set_num_threads(min(hostenv.cpu_quota(), int(os.environ.get("RAMP_CPU_THREADS", "64"))))
    with cpu_oracle_ops():
        ref = _cpu_tracker(state, args, cfg_kwargs, w_bias=PARITY_W_BIAS)
        ref.update()
        r = dict(poses=ref.poses_[:n].numpy().copy(), depth=ref.patches_[:n, :, 2, 1, 1].numpy().copy(),
                 net=ref.net[0].float().numpy().copy(), w=ref.last_weight.numpy().copy())
    # the same step on the oracle WITH the shipped precision policy's roundings (oracle/host_cpu.py): the fp16 leg's
    # distance to it is kernel error, its distance to the fp32 oracle the policy itself
    with cpu_oracle_ops(fp16_policy=True):
        refp = _cpu_tracker(state, args, cfg_kwargs, w_bias=PARITY_W_BIAS)
        refp.update()
        rp = dict(poses=refp.poses_[:n].numpy().copy(), depth=refp.patches_[:n, :, 2, 1, 1].numpy().copy(),
                  net=refp.net[0].float().numpy().copy(), w=refp.last_weight.numpy().copy())
        del refp
    torch.set_num_threads(host_pool)       # (the HIP legs below: the tracker's own pool again)
    step = float(np.abs(r["poses"] - before).max())
    scale = max(1.0, step)
    tf = dict(edges=int(state["ii"].shape[0]), keyframes=n, gn_step=round(step, 6), w_head_bias_shift=PARITY_W_BIAS,
              note="max abs error of one update() vs the CPU oracle (fp32) from the same snapshot; net relative to its "
                   "largest entry, poses / depths relative to max(1, |GN step|) (depths also to max(1, |depth|)); depths = "
                   "99.5th percentile over the ~1e4 patches (what tests/test_pipeline_gpu.py bounds), depths_max = the "
                   "worst patch (a low-confidence patch on an ill-conditioned depth can sit near 1 in the fp16 leg); "
                   "fp16_vs_fp16_policy_oracle = the fp16 leg against the CPU oracle with the same rounding points "
                   "(fp16 Linear operands / outputs, fp32 accumulate): kernel error without the precision policy's")
    if True:
        for name, mixed in (("fp16", True), ("fp32", False)):
            slam = Ramp_vo(make_cfg(args.preset, **dict(cfg_kwargs, MIXED_PRECISION=mixed)), net, {"event_bias": True},
                           ht=args.height, wd=args.width, device=dev)
            slam.load_state_dict(state)
            slam.update()
            torch.cuda.synchronize()
            g_depth = slam.patches_[:n, :, 2, 1, 1].cpu().numpy()
            # the d > 20 -> 1 reset (ba_cuda.cu:220) is a step function: depths ending within 0.1 of it are counted, not compared
            at_reset = (np.abs(r["depth"] - 20.0) < 0.1) | (np.abs(g_depth - 20.0) < 0.1)
            derr = (np.abs(g_depth - r["depth"]) / np.maximum(np.abs(r["depth"]), 1.0))[~at_reset]
            leg = dict(
                net=float(np.abs(slam.net[0].float().cpu().numpy() - r["net"]).max() / np.abs(r["net"]).max()),
                weight=float(np.abs(slam.last_weight.cpu().numpy() - r["w"]).max()),
                poses=float(np.abs(slam.poses_[:n].cpu().numpy() - r["poses"]).max() / scale),
                poses_over_gn_step=float(np.abs(slam.poses_[:n].cpu().numpy() - r["poses"]).max() / max(step, 1e-12)),
                depths=float(np.percentile(derr, 99.5) / scale), depths_p999=float(np.percentile(derr, 99.9) / scale),
                depths_max=float(derr.max() / scale),
                depths_at_reset_threshold=float(at_reset.sum()))
            tf[name] = {k: float("%.3g" % v) for k, v in leg.items()}
            if mixed:
                at_p = (np.abs(rp["depth"] - 20.0) < 0.1) | (np.abs(g_depth - 20.0) < 0.1)
                dp = (np.abs(g_depth - rp["depth"]) / np.maximum(np.abs(rp["depth"]), 1.0))[~at_p]
                tf["fp16_vs_fp16_policy_oracle"] = {k: float("%.3g" % v) for k, v in dict(
                    net=float(np.abs(slam.net[0].float().cpu().numpy() - rp["net"]).max() / np.abs(rp["net"]).max()),
                    weight=float(np.abs(slam.last_weight.cpu().numpy() - rp["w"]).max()),
                    poses_over_gn_step=float(np.abs(slam.poses_[:n].cpu().numpy() - rp["poses"]).max() / max(step, 1e-12)),
                    depths=float(np.percentile(dp, 99.5) / scale), depths_p999=float(np.percentile(dp, 99.9) / scale),
                    depths_max=float(dp.max() / scale)).items()}
            del slam
    # the same step with the weights exactly as the benchmark tracks with them (no bias shift: confidences ~0.5, the
    # ill-conditioned regime), fp32 leg, next to the problem's own fp32 rounding envelope: the oracle step's
    # bundle-adjustment inputs solved by the fp64 build of the same oracle source (as tests/test_pipeline_gpu.py::
    # test_full_size_update_step_in_the_wide_regime asserts it: HIP-vs-oracle <= 4 x oracle-fp32-vs-fp64)
    net0 = make_network(args.mode, device=dev)
    ba_in = []
    with cpu_oracle_ops():
        import rampvo_amd.ops as _ops
        _ba = _ops.ba

        def spy(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2, info=None, **kw):
            P = patches.shape[-1]
            ba_in.append((poses.view(-1, 7).numpy().copy(), patches.view(-1, 3, P, P).numpy().copy(), intrinsics.numpy().copy(),
                          target.numpy().copy(), weight.numpy().copy(), lmbda.numpy().copy(), ii.numpy().copy(),
                          jj.numpy().copy(), kk.numpy().copy(), int(t0), int(t1), int(iterations)))
            return _ba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations, info, **kw)
        _ops.ba = spy
        ref = _cpu_tracker(state, args, cfg_kwargs)
        ref.update()
        r0 = dict(poses=ref.poses_[:n].numpy().copy(), depth=ref.patches_[:n, :, 2, 1, 1].numpy().copy())
    import oracle as _orc
    p64, pt64 = _orc.ba_f64(*ba_in[-1])
    d64 = pt64.reshape(-1, r0["depth"].shape[1], 3, 3, 3)[:n, :, 2, 1, 1]
    slam = Ramp_vo(make_cfg(args.preset, **dict(cfg_kwargs, MIXED_PRECISION=False)), net0, {"event_bias": True},
                   ht=args.height, wd=args.width, device=dev)
    slam.load_state_dict(state)
    slam.update()
    g_depth = slam.patches_[:n, :, 2, 1, 1].cpu().numpy()
    step0 = float(np.abs(r0["poses"] - before).max())
    at_reset = (np.abs(r0["depth"] - 20.0) < 0.1) | (np.abs(g_depth - 20.0) < 0.1)
    derr = (np.abs(g_depth - r0["depth"]) / np.maximum(np.abs(r0["depth"]), 1.0))[~at_reset]
    at64 = (np.abs(r0["depth"] - 20.0) < 0.1) | (np.abs(d64 - 20.0) < 0.1)
    e64 = (np.abs(r0["depth"] - d64) / np.maximum(np.abs(d64), 1.0))[~at64]
    tf["fp32_wide_regime"] = {k: float("%.3g" % v) for k, v in dict(
        gn_step=step0, poses_over_gn_step=float(np.abs(slam.poses_[:n].cpu().numpy() - r0["poses"]).max() / max(step0, 1e-12)),
        depths_p999=float(np.percentile(derr, 99.9)), depths_max=float(derr.max()),
        oracle_fp32_vs_fp64_poses_over_gn_step=float(np.abs(r0["poses"] - p64[:n]).max() / max(step0, 1e-12)),
        oracle_fp32_vs_fp64_depths_p999=float(np.percentile(e64, 99.9)), oracle_fp32_vs_fp64_depths_max=float(e64.max())).items()}
    del slam
    out = dict(teacher_forced=tf)
    # trajectory level: tests/pipeline_checks.py::check_trajectory against tests/golden/ramp_vo_traj_ss.npz
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pipeline_checks as pc
    tr = {}
    for name, mixed in (("fp32", False), ("fp16", True)):
        e = pc.check_trajectory("ss", "cuda", mixed=mixed)
        tr[name] = dict(ate_vs_reference=float("%.3g" % e["ate_rmse"]), rel=float("%.3g" % e["rel"]),
                        depths_rel=float("%.3g" % e["depths_rel"]))
        tr.update(frames=e["frames"], path_length=round(e["path_length"], 4))
    tr["note"] = ("free-running %s 192x256, 16 patches, default.yaml windows, damped weight profile, against the "
                  "reference's own CPU run of the same stream (tests/golden/ramp_vo_traj_ss.npz): identical keyframe "
                  "decisions and graph; ate = Sim(3)-aligned RMSE (evaluate.py:295-304), rel = max abs / "
                  "max(1, largest translation)" % pc.TRAJ["ss"]["mode"])
    out["trajectory"] = tr
    return out


def fp32_leg(args, traj_fp32):
    """The same workload with MIXED_PRECISION off -- fp32 features, correlation volume, hidden state; the update operator's
    Linear layers at fp32 accuracy (csrc/update_x3.hip) -- the precision whose poses / depths agree with the reference to
    north_star's 1e-4 (the fp16 headline is default.yaml's autocast policy, a few 1e-3): a short timed run of this script in a
    process of its own, behind everything else of rank 0 (N = 1).  Returns the keys merged into `config`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--mixed", "0", "--steps", str(args.fp32_leg), "--warmup", "10",
           "--cpu-steps", "0", "--parity", "0", "--fp32-leg", "0", "--np-steps", "0", "--live-steps", "0", "--inst-steps", "40",
           "--height", str(args.height), "--width", str(args.width), "--patches", str(args.patches), "--mode", args.mode,
           "--preset", args.preset, "--prime", str(args.prime), "--pipeline", str(args.pipeline)]
    if args.opt_window:
        cmd += ["--opt-window", str(args.opt_window)]
    if args.keyframe_thresh is not None:
        cmd += ["--keyframe-thresh", str(args.keyframe_thresh)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                            "ROLE_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    tic = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("bench.py: the fp32 leg failed (rc %d): %s" % (r.returncode, r.stderr[-1500:]))
    d = json.loads(lines[-1])
    out = {"fp32_kfps": d["value"], "fp32_ms_per_step": d["ms_per_step"],
           "fp32_leg": {"what": "the same workload and stream with MIXED_PRECISION off (`bench.py --mixed 0`): %d timed steps in a "
                                "process of its own behind this one's measurements (%.0f s incl. start-up and priming)"
                                % (d["steps"], time.perf_counter() - tic),
                        "edges": d["config"].get("edges"), "frame_pipelining": d["config"].get("frame_pipelining"),
                        "roofline": {k: d.get("roofline", {}).get(k) for k in ("kernel", "mean_launch_us", "achieved", "frac",
                                                                               "edges_per_launch", "bytes_per_launch", "traffic",
                                                                               "traffic_source")},
                        "roofline_update": {k: (d.get("roofline_update") or {}).get(k)
                                            for k in ("kernel", "mean_call_us", "achieved", "peak", "frac", "edges")},
                        "roofline_encoder_mean_front_end_us": (d.get("roofline_encoder") or {}).get("mean_front_end_us"),
                        "roofline_ba_mean_call_us": (d.get("roofline_ba") or {}).get("mean_call_us")}}
    if traj_fp32:
        # (this process's parity block: the fp32 tracker -- the same kernels -- free running against the reference's own run)
        out["fp32_traj_rel_vs_reference"] = traj_fp32.get("rel")
        out["fp32_ate_vs_reference"] = traj_fp32.get("ate_vs_reference")
    return out


class LegStats:
    """host-side view of one bench leg: the return time of every step (a stall of the GPU shows within MAX_AHEAD frames as a
    long host step: the tracker throttles its run-ahead) and the tracker's own counters over the leg -- frames that ran
    device resident / host driven, hand-backs to the host, waits of the host for the GPU.  A leg whose frames fell back to
    the host-driven path, or that contains one long stall, reads as a low rate; these numbers say which."""

    def __init__(self, slam):
        self.slam, self.marks = slam, []
        self.s0 = dict(slam.stats)
        self.marks.append(time.perf_counter())

    def tick(self):
        self.marks.append(time.perf_counter())

    def summary(self):
        d = np.diff(self.marks) if len(self.marks) > 1 else np.zeros(1)
        st = {k: self.slam.stats[k] - self.s0[k] for k in self.s0}
        return {"steps": int(len(d)), "host_step_ms_p50_p90_max": [round(1e3 * float(v), 3) for v in
                                                                   (np.percentile(d, 50), np.percentile(d, 90), d.max())],
                "device_frames": int(st["device_frames"]), "host_frames": int(st["host_frames"]), "settles": int(st["settles"]),
                "throttle_waits": int(st["throttle_waits"]), "throttle_ms": round(1e3 * st["throttle_s"], 2)}


def _which_config(args, world):
    """names the BASELINE.json config the flags amount to (every rank of an N > 1 run tracks the same workload on its
    own sequence; configs[3] = --config 3)"""
    ms, big = args.mode == "MultiScale", (args.height, args.width, args.patches) == (720, 1280, 256)
    if not ms and args.preset == "default" and (args.height, args.width, args.patches) == (480, 640, 96) and not args.opt_window:
        name = "BASELINE configs[1]"
        if world > 1:
            name += " on every rank (configs[3], MultiScale / precise windows per rank: --config 3)"
    elif ms and args.preset == "precise" and (args.height, args.width, args.patches) == (480, 640, 96) and not args.opt_window:
        name = "BASELINE configs[3] (configs[2]'s workload, one sequence per rank)" if world > 1 else "BASELINE configs[2]"
    elif ms and big and args.opt_window == 32 and args.preset == "precise":
        name = "BASELINE configs[4]" + (" as written" if (args.encoder_fp8 and args.keyframe_thresh == 0.0) else " (variant)")
    else:
        return ""
    return name + ": "


def graph_size(slam):
    """(factors, keyframes) of the tracker without taking a device-resident state back to the host"""
    dv = getattr(slam, "_dev", None)
    if dv is not None and dv.active:
        from rampvo_amd import track_dev
        torch.cuda.synchronize()
        d = dv.dyn.cpu().numpy()
        return int(d[track_dev.DYN_EKEPT]), int(d[track_dev.DYN_NROW])
    return len(slam._ii), slam.n


def live_factor_fractions(slam, windows=False):
    """share of the current graph's factors with a correlation window inside the target plane, per level (the
    kernel's own test, ramp/altcorr/correlation_kernel.cu:104-110's bounds on the 8 x 8 window), from the
    device-resident step's reprojection.  windows: a dict that also carries the median / 90th percentile area (pixels) of
    the union of a factor's nine 8 x 8 windows per level (one 10 x 10 window = 100; the kernel gathers the union in one
    pass up to 192 pixels and window by window beyond that)"""
    dv = getattr(slam, "_dev", None)
    if dv is None or not dv.active:
        return None
    from rampvo_amd import track_dev
    torch.cuda.synchronize()
    E = int(dv.dyn.cpu().numpy()[track_dev.DYN_E])
    co = dv.coords[:E].reshape(E, 2, 9)
    h, w = slam.ht // slam.RES, slam.wd // slam.RES
    out, areas = [], []
    for div, (H, W) in ((1.0, (h, w)), (4.0, (h // 4, w // 4))):
        fx, fy = torch.floor(co[:, 0] / div), torch.floor(co[:, 1] / div)
        live = (fx - 3 < W) & (fx - 3 + 8 > 0) & (fy - 3 < H) & (fy - 3 + 8 > 0)
        out.append(round(float(live.any(1).float().mean()), 3))
        if windows:
            big = 1e9
            bw = torch.where(live, fx, -big).max(1).values - torch.where(live, fx, big).min(1).values + 8
            bh = torch.where(live, fy, -big).max(1).values - torch.where(live, fy, big).min(1).values + 8
            a = (bw * bh)[live.any(1)].double().clamp(max=1e9)
            areas.append([int(a.median()), int(a.quantile(0.9))] if a.numel() else None)
    if windows:
        return {"live_factor_fraction_fine_coarse": out, "union_window_px_median_p90_fine_coarse": areas}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    # RAMP_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks share
    # devices, RCCL refuses that); the driver's runs use the default, RCCL
    backend = os.environ.get("RAMP_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # launched by torch.distributed.run (RANK / WORLD_SIZE in the environment): the process group is created even for a
    # world of one, so that the RCCL initialisation, the barriers and the two collectives of the N > 1 path run on a
    # one-GPU box too (tests/test_pipeline_gpu.py::test_bench_under_torchrun_initialises_rccl); a plain
    # `python bench.py` (the driver's N = 1 run) creates none
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    host = {}
    if world > 1:
        # N trackers on one host: each rank's Python needs about one core (a tracked frame is one C call in steady
        # state: ~0.2 ms of host time per 1.1 ms step).  Without this every rank brings torch's intra-op pool of
        # all-cores threads and the ranks migrate over each other's caches: one thread per rank, and a private,
        # contiguous slice of the cores this job may use (contiguous = same NUMA node / CCD on the usual enumeration)
        torch.set_num_threads(1)
        try:
            cores = sorted(os.sched_getaffinity(0))
            per = max(len(cores) // world, 1)
            mine = cores[(local_rank * per) % len(cores):][:per] or cores
            os.sched_setaffinity(0, mine)
            host = {"threads": 1, "cores_of_rank0": "%d-%d" % (mine[0], mine[-1]), "cores_per_rank": len(mine)}
        except (AttributeError, OSError):
            host = {"threads": 1}
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from rampvo_amd.config import make_cfg
    from rampvo_amd.Ramp_vo import Ramp_vo
    from rampvo_amd.synthetic import SyntheticStream, make_network

    cfg_kwargs = dict(PATCHES_PER_FRAME=args.patches, MIXED_PRECISION=bool(args.mixed))
    if args.keyframe_thresh is not None:
        cfg_kwargs["KEYFRAME_THRESH"] = float(args.keyframe_thresh)
    if args.opt_window:
        cfg_kwargs["OPTIMIZATION_WINDOW"] = args.opt_window
    if args.encoder_fp8:
        cfg_kwargs["ENCODER_FP8"] = True
    cfg = make_cfg(args.preset, **cfg_kwargs)
    torch.manual_seed(1234 + rank)
    net = make_network(args.mode, device=dev)
    slam = Ramp_vo(cfg, net, {"event_bias": True}, ht=args.height, wd=args.width, device=dev)
    slam.inputs_ready = bool(args.pipeline)     # every frame is resident before its __call__ (see parse())
    solo = rank == 0 and world == 1
    n_warm = args.clock_warm_max if args.clock_warm_s > 0 else 0
    n_np = args.np_steps if args.pipeline else 0
    n_inst = 0 if args.no_kernel_timing else args.inst_steps
    n_alone = 20 if (n_inst and n_np) else 0
    n_live = args.live_steps if (n_inst and solo) else 0
    n_os = n_np                                   # the own-stream leg (inputs produced on the caller's stream: evaluate.run's loop)
    total = args.prime + n_warm + args.warmup + args.steps + n_inst + n_os + n_np + n_alone + (2 * n_live + args.handback_frames if n_live else 0)
    n_cpu = args.cpu_steps + 1 if (solo and args.cpu_steps > 0) else 0
    stream = SyntheticStream(args.height, args.width, total + n_cpu + 1, seed=1234 + rank, device=dev)
    frames = [tuple(x.to(dev) if i < 2 else x for i, x in enumerate(stream.frame(t))) for t in range(total)]

    ctimer, etimer, btimer, utimer = CorrTimer(), EncoderTimer(), BaTimer(), UpdateTimer()
    dprobe = None
    if not args.no_kernel_timing:
        ctimer.install()
        etimer.install(net)
        btimer.install()
        utimer.install()
        dprobe = DeviceProbe(args.steps + n_inst + n_alone + 2 * n_live)
        dprobe.install()

    pos = {"t": 0}

    def step():
        im, ev, K, mask = frames[pos["t"]]
        slam(pos["t"], input_tensor=(ev, im, mask), intrinsics=K)
        pos["t"] += 1

    for _ in range(args.prime):
        step()
    assert slam.is_initialized, "tracker did not initialise during priming"
    # A generation-2 pass of CPython's cyclic collector over this process's ~10^6 long-lived objects (modules,
    # the resident frame list) takes ~57 ms -- 0.3 ms per step if one lands in a 200-step timed region.  The
    # tracker itself creates no reference cycles per frame, so the long-lived objects are moved out of the
    # collector's sight once (what a deployed tracker process does after start-up as well).  Done BEFORE the clock
    # warm phase: a 57 ms idle GPU right in front of a 25 ms timed region (the driver's --steps 20) is a cold start.
    gc.collect()
    gc.freeze()
    # clock warm: the same steady-state steps, untimed, until the GPU has been busy for clock_warm_s (a fresh
    # box idles at low clocks; a 20-step region is ~30 ms)
    warm_steps, warm_tic = 0, time.perf_counter()
    while warm_steps < n_warm and time.perf_counter() - warm_tic < args.clock_warm_s:
        step()
        warm_steps += 1
    torch.cuda.synchronize()
    warm_s = time.perf_counter() - warm_tic
    for _ in range(args.warmup):
        step()
    E0, n0 = graph_size(slam)

    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # Inside the timed region only the dominant kernel is bracketed, and only on every --probe-every-th step: an event
    # record costs a ~6 us bubble on its stream (five per step were 1.4 % of the rate).  The update operator, bundle
    # adjustment and the front end are timed in the instrumented pass BEHIND the timed region (same stream of frames).
    dv = getattr(slam, "_dev", None)              # (the fp32 path and unsupported configurations stay host driven)
    device_step = dprobe is not None and dv is not None and bool(getattr(dv, "active", False))
    ctimer.enabled = not device_step             # (host-driven path: the Python-level hooks, every step, as before)
    etimer.enabled = btimer.enabled = utimer.enabled = ctimer.enabled and dprobe is not None
    if dprobe is not None:
        dprobe.enabled = True
    # (the live factor count of every timed frame from the host's lazy copy of the device-side sizes -- pinned memory, a few
    # frames old, never waited for: a frame's time is proportional to it and it drifts by +-8 % along the stream, which is
    # most of the spread between two short timed regions)
    e_lazy = dv.dyn_host.numpy() if device_step else None
    e_seen = []
    legs = {}
    leg = LegStats(slam)
    tic = time.perf_counter()
    marks = [tic]
    for i in range(args.steps):
        if dprobe is not None:
            # (not the first step behind the synchronisation: the drained pipeline ran no cache warm-up for it)
            dprobe.mode = "corr" if i % max(1, args.probe_every) == max(1, args.probe_every) // 2 else None
        step()
        marks.append(time.perf_counter())         # host-side return times (the GPU may lag by less than a step)
        leg.marks.append(marks[-1])
        if e_lazy is not None:
            e_seen.append(int(e_lazy[2]))
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - tic
    legs["timed"] = leg.summary()
    gate_kind = "signal word stored by the gru launch" if getattr(slam, "_gate_by_flag", False) else "event"
    ctimer.enabled = etimer.enabled = btimer.enabled = utimer.enabled = False
    fp32_dev = device_step and bool(getattr(dv, "fp32", False)) and not bool(getattr(dv, "x3", False))
    if dprobe is not None and device_step:
        dprobe.mode, etimer.enabled, utimer.enabled = "all", True, fp32_dev
        for _ in range(n_inst):
            step()
        torch.cuda.synchronize()
        etimer.enabled = utimer.enabled = False
    if dprobe is not None:
        dprobe.enabled = False

    from rampvo_amd.shard import gather_metrics, max_over_ranks
    dt_all = max_over_ranks(dt, dev)
    per_rank = None
    if use_dist:
        # the path's single collective: per-sequence metrics (kf/s, E, n, pose checksum)
        E_r, n_r = graph_size(slam)
        g = gather_metrics([args.steps / dt, float(E_r), float(n_r), float(slam.poses_[:n_r].double().sum())], dev)
        per_rank = [[round(float(v), 4) for v in row] for row in g.tolist()]
    E1 = graph_size(slam)[0]
    live = live_factor_fractions(slam)

    # inputs_ready = "stream": what a caller gets that produces every frame's tensors on ITS stream right before the call
    # -- the reference's evaluate.py resizes each frame there; rampvo_amd.evaluate.run tracks in this mode -- : the tracker
    # waits for the caller's stream by an event and runs on its own, so the frames still pipeline
    os_kfps = None
    if n_os > 0:
        slam.inputs_ready = "stream"
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        leg = LegStats(slam)
        t_os = time.perf_counter()
        for _ in range(n_os - 2):
            im, ev, K, mask = frames[pos["t"]]
            ev2, im2 = ev.clone(), im.clone()      # (the caller's per-frame preprocessing, on the current stream)
            slam(pos["t"], input_tensor=(ev2, im2, mask), intrinsics=K)
            del ev2, im2                           # (and its tensors die right behind the call)
            pos["t"] += 1
            leg.tick()
        slam.peek()
        torch.cuda.synchronize()
        os_kfps = (n_os - 2) / (time.perf_counter() - t_os)
        legs["own_stream"] = leg.summary()
        slam.inputs_ready = bool(args.pipeline)
    # the strictly sequential rate (no frame pipelining): what a caller on ONE stream with inputs_ready False gets
    np_kfps = None
    if n_np > 0:
        slam.inputs_ready = False
        torch.cuda.synchronize()
        leg = LegStats(slam)
        t_np = time.perf_counter()
        for _ in range(n_np):
            step()
            leg.tick()
        torch.cuda.synchronize()
        np_kfps = n_np / (time.perf_counter() - t_np)
        legs["non_pipelined"] = leg.summary()
        if dprobe is not None and device_step:     # (behind the timed sequential pass) the front end's replay alone
            keep, etimer.pairs, etimer.enabled = etimer.pairs, [], True
            dprobe.enabled, dprobe.mode = True, "alone"
            for _ in range(n_alone):
                step()
            torch.cuda.synchronize()
            dprobe.enabled = False
            etimer.alone, etimer.pairs, etimer.enabled = etimer.pairs, keep, False

    # the state the parity block and the CPU baseline start from: taken BEFORE the live-factor leg changes what the
    # tracker correlates (state_dict() hands a device-resident state back to the host; the next frame re-enters)
    snapshot = slam.state_dict() if (rank == 0 and ((solo and args.cpu_steps > 0) or (solo and args.parity))) else None
    live_leg = None
    if n_live and dprobe is not None and device_step:
        from rampvo_amd import track_dev
        slam.inputs_ready = bool(args.pipeline)
        # back into the device-resident state.  These frames are a leg of their own (config.legs.after_handback): a host
        # pool larger than the container's CPU quota used to get the process frozen for 15 .. 85 ms a few frames behind a
        # hand-back (rampvo_amd/hostenv.py is the cure) and the converged leg, which started
        # 6 frames behind the snapshot, measured that -- if anything of the kind comes back it shows here as `max`
        hb = LegStats(slam)
        for _ in range(args.handback_frames):
            step()
            hb.tick()
        torch.cuda.synchronize()
        legs["after_handback"] = hb.summary()
        dvl = getattr(slam, "_dev", None)
        if dvl is not None and dvl.active:
            live_leg = {}
            for mode, flag in (("compact", track_dev.COMPACT_COORDS), ("live", track_dev.WRAP_COORDS)):
                slam._extra_step_flags = flag
                dprobe.enabled, dprobe.mode = True, mode
                for _ in range(2):                 # (the pipeline refilling behind the flag change: untimed)
                    step()
                torch.cuda.synchronize()
                leg = LegStats(slam)
                t_leg = time.perf_counter()
                for _ in range(n_live - 2):
                    step()
                    leg.tick()
                torch.cuda.synchronize()
                t_leg = time.perf_counter() - t_leg
                legs["converged" if mode == "compact" else "live"] = leg.summary()
                dprobe.enabled = False
                live_leg[mode] = live_factor_fractions(slam, windows=True)
                # the whole step, end to end, with this leg's factors (the correlation launch is bracketed by events on
                # every step here, ~1 % of the rate): what a converged tracker would run at, not just its kernel
                live_leg[mode]["kfps"] = round((n_live - 2) / t_leg, 1)
            slam._extra_step_flags = 0
    if dprobe is not None:
        dprobe.feed(ctimer, utimer, btimer, cfg.OPTIMIZATION_WINDOW)
    if rank == 0:
        value = world * args.steps / dt_all
        out = {
            "metric": "keyframes/sec (BA iters/sec) %s %dx%d; ATE vs reference" % (args.mode, args.width, args.height),
            "value": round(value, 3), "unit": "keyframes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt_all / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (features, MFMA inputs; f32 accumulate, hidden state, BA, geometry: default.yaml "
                     "MIXED_PRECISION)" if args.mixed else "f32",
            "data": "synthetic (seeded %dx%d event+frame stream, seeded random-init weights)" % (args.width, args.height),
            "config": {"workload": "%s%s %dx%d, %d patches/frame, %s.yaml windows%s, 2 BA iters/keyframe, "
                                   "steady-state sliding window%s"
                                   % (_which_config(args, world), args.mode, args.width, args.height, args.patches, args.preset,
                                      " with OPTIMIZATION_WINDOW %d" % args.opt_window if args.opt_window else "",
                                      "".join([", KEYFRAME_THRESH %g" % args.keyframe_thresh if args.keyframe_thresh is not None else "",
                                               ", fp8-MFMA encoder" if args.encoder_fp8 else "",
                                               ", fp32 everywhere" if not args.mixed else "",
                                               ", device-resident steps with the operator's GEMMs launched by the host"
                                               if fp32_dev else "",
                                               ", device-resident steps (the operator's Linear layers: split-fp16 operands on the "
                                               "f16 matrix cores, fp32 accumulate -- csrc/update_x3.hip; correlation: "
                                               "the same split on the target planes, corr_mfma_kernel<CorrX2>)"
                                               if (not args.mixed and device_step and not fp32_dev) else "",
                                               ", frame pipelining off" if not args.pipeline else "",
                                               ", host-driven steps (RAMP_DEVICE_STEP=0)"
                                               if os.environ.get("RAMP_DEVICE_STEP", "1") != "1" else ""])),
                       "ba_iters_per_s": round(2 * value, 2), "edges": E0, "edges_end": E1,
                       "factors_per_timed_frame": ({"mean": round(float(np.mean(e_seen)), 1), "min": int(min(e_seen)),
                                                    "max": int(max(e_seen))} if e_seen else None),
                       "factor_updates_per_s": round(value * float(np.mean(e_seen)), 0) if e_seen else None,
                       "keyframes_in_window": n0, "prime_frames": args.prime,
                       "clock_warm": "%d untimed steady-state steps (%.2f s) before the %d warm-up steps"
                                     % (warm_steps, warm_s, args.warmup),
                       "frame_pipelining": bool(args.pipeline),
                       "pipeline_gate": gate_kind if args.pipeline else None,
                       "non_pipelined_kfps": round(np_kfps, 1) if np_kfps else None,
                       "own_stream_kfps": round(os_kfps, 1) if os_kfps else None,
                       "host_step_ms_p50_p90_max": [round(1e3 * float(v), 3) for v in
                                                    (np.percentile(np.diff(marks), 50), np.percentile(np.diff(marks), 90),
                                                     np.max(np.diff(marks)))],
                       # per leg: host step times and the tracker's counters (LegStats): a leg with host_frames > 0 fell back
                       # to the host-driven path, one with max >> p50 contains a stall
                       "legs": legs,
                       "sharding": "independent sequences, 1 per GPU" if world > 1 else "single sequence"},
        }
        if per_rank is not None:
            out["config"]["per_rank_kfps_E_n_chk"] = per_rank
            out["config"]["process_group"] = {"backend": dist.get_backend(), "world": dist.get_world_size()}
            out["config"]["host_placement"] = host
        from rampvo_amd import hostenv
        # the host side: torch's intra-op pool as the tracker left it and the CPUs this process may use (rampvo_amd/hostenv.py)
        out["config"]["host_threads"] = {"pool": torch.get_num_threads(), "cpu_quota": hostenv.cpu_quota(),
                                         "logical_cpus": os.cpu_count()}
        rl = ctimer.summary(2 if args.mixed else 4, slam)
        assert rl is not None or args.no_kernel_timing, "no correlation launch was timed: the roofline hook is stale"
        if rl is not None:
            if live is not None:
                # factors with at least one patch pixel's window inside the target plane, per pyramid level: the others
                # (the random-init network lets half of the projections leave the image) cost a row of zeros, and
                # model_bytes / mfma_tflops count them as if they gathered
                rl["live_factor_fraction_fine_coarse"] = live
            if dprobe is not None and device_step:
                rl["sampling"] = ("HIP events around the launch on every %d-th step of the timed region (%d launches); "
                                  "mean over the %d-step instrumented pass behind it: %.1f us"
                                  % (max(1, args.probe_every), rl["launches"], n_inst,
                                     1e3 * float(np.mean(dprobe.corr_inst_ms)) if dprobe.corr_inst_ms else float("nan")))
            out["roofline"] = rl
            what = {"live": "the same kernel inside the pipelined frame with EVERY factor's windows inside the target plane: each "
                            "reprojection is moved into the plane by whole plane widths / heights right before the correlation "
                            "launch (RAMP_TRACK_WRAP_COORDS, %d steps behind everything else; factor list, target frames, patches "
                            "and the patches' projected shapes are the random-weight tracker's own: see union_window_px)" % n_live,
                    "compact": "the same, and every patch reprojected with unit pixel spacing around its centre "
                               "(RAMP_TRACK_COMPACT_COORDS): the factors of a converged tracker -- one ~10 x 10 union window "
                               "per level, where the random-weight tracker's patches spread over tens of pixels and take the "
                               "kernel's nine-separate-windows path"}
            for mode, key in (("compact", "roofline_live_compact"), ("live", "roofline_live")):
                if live_leg is None or not dprobe.live_pairs[mode]:
                    continue
                lt = CorrTimer()
                lt.pairs, lt.edges = dprobe.live_pairs[mode][2:], dprobe.live_edges[mode][2:]   # (the first two: the pipeline refilling)
                ll = lt.summary(2 if args.mixed else 4, slam, segment={"compact": "live_compact", "live": "live_wrapped"}[mode])
                if mode == "compact":
                    out["config"]["converged_kfps"] = live_leg[mode]["kfps"]
                    rl["frac_live_compact"] = ll["frac"]
                out[key] = {"what": what[mode], **live_leg[mode],
                            **{k: ll[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches", "mean_launch_us",
                                                  "bytes_per_launch", "edges_per_launch", "target_frames", "patches",
                                                  "model_bytes", "model_gbps", "mfma_tflops", "mfma_frac", "traffic",
                                                  "traffic_gbps", "traffic_frac_of_peak", "traffic_source_edges_per_launch")
                               if k in ll}}
        for key, val in (("roofline_update", utimer.summary(bool(args.mixed))),
                         ("roofline_encoder", etimer.summary(bool(args.mixed))),
                         ("roofline_ba", btimer.summary(args.patches, cfg.REMOVAL_WINDOW))):
            if val is not None:
                if device_step:
                    val["measured_in"] = "the %d-step instrumented pass behind the timed region" % n_inst
                    if key == "roofline_encoder" and etimer.alone:
                        val["measured_in"] = ("%d sequential steps behind the sequential pass (alone); next to the tail: the "
                                              "%d-step instrumented pass" % (len(etimer.alone), n_inst))
                out[key] = val
        if solo and args.parity:
            # a failing checker fails the run (rc != 0): a headline without its parity block is not a result
            out["parity"] = parity_block(snapshot, args, cfg_kwargs, net, dev)
            tr = out["parity"].get("trajectory", {})
            if "fp32" in tr:
                out["ate_vs_oracle"] = tr["fp32"]["ate_vs_reference"]
        if solo and args.mixed and args.fp32_leg > 0 and _which_config(args, world):
            out["config"].update(fp32_leg(args, out.get("parity", {}).get("trajectory", {}).get("fp32")))
        if n_cpu:
            cpu_frames = [stream.frame(total + i) for i in range(n_cpu)]
            cpu_frames = [tuple(x.cpu() for x in f) for f in cpu_frames]
            out["cpu_baseline"] = cpu_baseline(snapshot, args, cfg_kwargs, cpu_frames, args.cpu_steps)
            out["config"]["speedup_vs_cpu_port"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
