#!/usr/bin/env python
"""bench.py -- keyframes/sec of the RAMP-VO tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): SingleScale encoder, 640x480 synthetic
event+frame stream, 96 patches/frame, default.yaml windows (lifetime 13, removal
22, optimisation 10), 2 BA iterations per keyframe.  One *step* = one
``Ramp_vo.__call__`` on an initialised tracker = encoder + patchify + reproject +
corr + update operator + 2 GN iterations + keyframe management.  The tracker is
first primed (untimed) until its sliding window is full, so the timed steps run at
the steady-state graph size (E ~ 45k edges).  Inputs for all steps are generated
and resident in HBM before the timed region.

N > 1: one process per GPU (torchrun), every rank tracks its own independent
sequence (different seed) -- the path shards by sequence only; the single
collective is an all_gather of per-rank metrics.  value = total steps of all ranks
/ max-over-ranks time.

Prints ONE JSON line on rank 0, including
  roofline     the fused correlation kernel: algorithmic bytes per launch / mean launch
               time (HIP events on the launch stream, inside the timed region)
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
    ap.add_argument("--mixed", type=int, default=1,
                    help="1 (default.yaml's MIXED_PRECISION: True): fp16 features / conv + GEMM I/O with fp32 "
                         "accumulation, fp32 hidden state, BA and geometry; 0: fp32 everywhere")
    ap.add_argument("--cpu-steps", type=int, default=6, help="steps of the CPU baseline sample, ~2 s each (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: the synthetic frames are resident in HBM before their step (the bench contract), so the "
                         "tracker may launch frame t+1's front end while frame t's bundle adjustment drains "
                         "(Ramp_vo.inputs_ready); 0: strictly one frame at a time")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


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
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = direct(slam, coords, ii, jj, order)
            e.record()
            timer.pairs.append((s, e))
            timer.edges.append(int(ii.shape[0]))
            return out

        Ramp_vo._corr_launch = timed_direct

    @staticmethod
    def pmc_traffic_per_edge(elem_bytes):
        """HBM-side bytes per edge of the fp16 kernel from the committed rocprofv3 PMC passes
        (FETCH_SIZE x2 per the guide's gfx950 correction + WRITE_SIZE; profiles/pmc/r01_corr_traffic.json).
        Counters cannot be read inside this process, so the per-edge figure of the PMC run of the same
        workload is scaled by this run's edges per launch.  None for the fp32 kernel (not measured)."""
        if elem_bytes != 2:
            return None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc", "r01_corr_traffic.json")) as f:
                return float(json.load(f)["bytes_per_edge"])
        except Exception:
            return None

    def summary(self, elem_bytes):
        if not self.pairs:
            return None
        ms = np.array([s.elapsed_time(e) for s, e in self.pairs])
        edges = np.array(self.edges, dtype=np.float64)
        # per edge per level: 128*9*s (patch) + 128*100*s (union of the 9 8x8 windows)
        # + 72 (coords) + 16 (indices) + 441*s (output); two levels per launch
        per_edge_level = 128 * 9 * elem_bytes + 128 * 100 * elem_bytes + 72 + 16 + 441 * elem_bytes
        big = edges > 0.5 * edges.max()          # the per-update launches (motion-probe launches are tiny)
        bytes_per_launch = float((edges[big] * 2 * per_edge_level).mean())
        mean_ms = float(ms[big].mean())
        achieved = bytes_per_launch / (mean_ms * 1e-3) / 1e9
        per_edge = self.pmc_traffic_per_edge(elem_bytes)
        traffic = int(per_edge * edges[big].mean()) if per_edge else None
        out = dict(kernel="corr_mfma_f16_kernel" if elem_bytes == 2 else "corr_kernel<float>", bound="hbm",
                   achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                   frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, launches=int(big.sum()),
                   mean_launch_us=round(mean_ms * 1e3, 1), bytes_per_launch=int(bytes_per_launch),
                   edges_per_launch=int(edges[big].mean()))
        if traffic:
            # the algorithmic model counts every edge's window privately; overlapping windows are served
            # from L2 / the XCD-local schedule, so measured traffic is well below it and frac can exceed 1
            out["traffic_gbps"] = round(traffic / (mean_ms * 1e-3) / 1e9, 1)
            out["traffic_frac_of_peak"] = round(traffic / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        return out


class BaTimer:
    """HIP events around fastba.BA (two Gauss-Newton iterations, 6 launches each + one memset)"""

    def __init__(self):
        self.pairs, self.meta, self.enabled = [], [], False

    def install(self):
        from rampvo_amd import fastba
        inner, timer = fastba.BA, self

        def timed(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k):
            if not timer.enabled:
                return inner(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
        return dict(kernel="fastba.BA: ba_edge / ba_patch_pair / ba_schur / ba_assemble / ba_chol64 / ba_retract x 2",
                    bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                    traffic=None, mean_call_us=round(ms * 1e3, 1), bytes_per_call=int(nbytes), edges=int(E),
                    free_poses=int(N), note="a dependent chain of 13 small launches: latency, not bandwidth, bounds it "
                                            "(time includes the overlap with the next frame's front end when pipelining)")


MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16 MFMA, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.0    # fp32 MFMA (the --mixed 0 towers)


class EncoderTimer:
    """HIP events around the encoder front end (one hipGraph replay per frame: fused LSTM + conv towers
    + patch selection + gathers) and the conv FLOPs of one frame, counted from the shapes of the
    ramp_conv2d_nhwc calls of an eager (un-captured) frame."""

    def __init__(self):
        self.pairs, self.enabled = [], False
        self.flops_frame, self._acc = 0, 0

    def install(self, net):
        from rampvo_amd import conv_hip
        timer, conv_inner = self, conv_hip.conv2d

        def counted(x, conv, *a, **k):
            y = conv_inner(x, conv, *a, **k)
            o = y.raw if isinstance(y, conv_hip.Pending) else y
            cout, cin, kh, kw = conv.weight.shape
            timer._acc += 2 * o.shape[0] * o.shape[1] * cout * cin * kh * kw
            return y

        conv_hip.conv2d = counted
        pat = net.patchify
        impl_inner, fwd_inner = pat._forward_impl, pat.forward

        def impl(*a, **k):
            timer._acc = 0
            r = impl_inner(*a, **k)
            timer.flops_frame = max(timer.flops_frame, timer._acc)
            return r

        def fwd(*a, **k):
            if not timer.enabled:
                return fwd_inner(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fwd_inner(*a, **k)
            e.record()
            timer.pairs.append((s, e))
            return r

        pat._forward_impl, pat.forward = impl, fwd

    def summary(self, mixed):
        if not self.pairs or not self.flops_frame:
            return None
        ms = float(np.mean([s.elapsed_time(e) for s, e in self.pairs]))
        peak = MFMA_F16_PEAK_TFLOPS if mixed else MFMA_F32_PEAK_TFLOPS
        ach = self.flops_frame / (ms * 1e-3) / 1e12
        return dict(kernel="encoder front end (conv_mfma_* towers + fused LSTM + patch selection, 1 hipGraph)",
                    bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 5),
                    conv_gflop_per_frame=round(self.flops_frame / 1e9, 2), mean_front_end_us=round(ms * 1e3, 1),
                    note="32/64-channel layers: arithmetic intensity is below the machine balance, the towers are "
                         "latency/HBM bound; MFMA busy from PMC: profiles/pmc/r01_g_MFMA_BUSY_per_kernel.txt")


def cpu_baseline(state, args, cfg_kwargs, frames, steps):
    """time `steps` tracker steps on the host cores from the same steady-state snapshot, through
    the CPU oracle backend (torch CPU for the encoder / update GEMMs, oracle C for the natives)"""
    from oracle.backend_cpu import cpu_oracle_ops
    from rampvo_amd.config import make_cfg
    from rampvo_amd.Ramp_vo import Ramp_vo
    from rampvo_amd.synthetic import make_network
    cores = min(os.cpu_count() or 1, int(os.environ.get("RAMP_CPU_THREADS", "64")))
    torch.set_num_threads(cores)      # also the OpenMP team of the oracle's edge-parallel loops
    cfg_kwargs = dict(cfg_kwargs, MIXED_PRECISION=False)       # the host path is fp32 throughout
    state = dict(state)
    for k in ("net", "imap", "gmap", "fmap1", "fmap2"):
        state[k] = state[k].float()
    with cpu_oracle_ops():
        net = make_network(args.mode, device="cpu")
        slam = Ramp_vo(make_cfg(args.preset, **cfg_kwargs), net, {"event_bias": True}, ht=args.height, wd=args.width,
                       device="cpu")
        slam.load_state_dict(state)
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
                       "encoder/update GEMMs + OpenMP oracle C natives on %d threads (host has %d logical cores)"
                       % (steps, len(slam._ii), cores, os.cpu_count() or 1),
                s_per_step=round(dt / steps, 3))


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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from rampvo_amd.config import make_cfg
    from rampvo_amd.Ramp_vo import Ramp_vo
    from rampvo_amd.synthetic import SyntheticStream, make_network

    cfg_kwargs = dict(PATCHES_PER_FRAME=args.patches, MIXED_PRECISION=bool(args.mixed))
    cfg = make_cfg(args.preset, **cfg_kwargs)
    torch.manual_seed(1234 + rank)
    net = make_network(args.mode, device=dev)
    slam = Ramp_vo(cfg, net, {"event_bias": True}, ht=args.height, wd=args.width, device=dev)
    slam.inputs_ready = bool(args.pipeline)     # every frame is resident before its __call__ (see parse())
    total = args.prime + args.warmup + args.steps
    n_cpu = args.cpu_steps + 1 if (rank == 0 and world == 1 and args.cpu_steps > 0) else 0
    stream = SyntheticStream(args.height, args.width, total + n_cpu + 1, seed=1234 + rank, device=dev)
    frames = [tuple(x.to(dev) if i < 2 else x for i, x in enumerate(stream.frame(t))) for t in range(total)]

    ctimer, etimer, btimer = CorrTimer(), EncoderTimer(), BaTimer()
    if not args.no_kernel_timing:
        ctimer.install()
        etimer.install(net)
        btimer.install()

    def step(t):
        im, ev, K, mask = frames[t]
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)

    t = 0
    for _ in range(args.prime):
        step(t); t += 1
    assert slam.is_initialized, "tracker did not initialise during priming"
    for _ in range(args.warmup):
        step(t); t += 1
    E0, n0 = len(slam._ii), slam.n

    # A generation-2 pass of CPython's cyclic collector over this process's ~10^6 long-lived objects (modules,
    # the resident frame list) takes ~57 ms -- 0.3 ms per step if one lands in a 200-step timed region.  The
    # tracker itself creates no reference cycles per frame, so the long-lived objects are moved out of the
    # collector's sight once (what a deployed tracker process does after start-up as well).
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ctimer.enabled = etimer.enabled = btimer.enabled = True
    tic = time.perf_counter()
    marks = [tic]
    for _ in range(args.steps):
        step(t); t += 1
        marks.append(time.perf_counter())         # host-side return times (the GPU may lag by less than a step)
    slam.settle()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - tic
    ctimer.enabled = etimer.enabled = btimer.enabled = False

    from rampvo_amd.shard import gather_metrics, max_over_ranks
    dt_all = max_over_ranks(dt, dev)
    per_rank = None
    if world > 1:
        # the path's single collective: per-sequence metrics (kf/s, E, n, pose checksum)
        g = gather_metrics([args.steps / dt, float(len(slam._ii)), float(slam.n),
                            float(slam.poses_[:slam.n].double().sum())], dev)
        per_rank = [[round(float(v), 4) for v in row] for row in g.tolist()]

    if rank == 0:
        value = world * args.steps / dt_all
        out = {
            "metric": "keyframes/sec (BA iters/sec) %s %dx%d; ATE vs reference" % (args.mode, args.width, args.height),
            "value": round(value, 3), "unit": "keyframes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt_all / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 storage+MFMA inputs / f32 accumulate, state, BA (default.yaml MIXED_PRECISION)" if args.mixed else "f32",
            "data": "synthetic (seeded %dx%d event+frame stream, seeded random-init weights)" % (args.width, args.height),
            "config": {"workload": "%s %dx%d, %d patches/frame, %s.yaml windows, 2 BA iters/keyframe, "
                                   "steady-state sliding window" % (args.mode, args.width, args.height, args.patches,
                                                                    args.preset),
                       "ba_iters_per_s": round(2 * value, 2), "edges": E0, "edges_end": len(slam._ii),
                       "keyframes_in_window": n0, "prime_frames": args.prime,
                       "frame_pipelining": bool(args.pipeline),
                       "host_step_ms_p50_p90_max": [round(1e3 * float(v), 3) for v in
                                                    (np.percentile(np.diff(marks), 50), np.percentile(np.diff(marks), 90),
                                                     np.max(np.diff(marks)))],
                       "sharding": "independent sequences, 1 per GPU" if world > 1 else "single sequence"},
        }
        if per_rank is not None:
            out["config"]["per_rank_kfps_E_n_chk"] = per_rank
        rl = ctimer.summary(2 if args.mixed else 4)
        assert rl is not None or args.no_kernel_timing, "no correlation launch was timed: the roofline hook is stale"
        if rl is not None:
            out["roofline"] = rl
        el = etimer.summary(bool(args.mixed))
        if el is not None:
            out["roofline_encoder"] = el
        bl = btimer.summary(args.patches, cfg.REMOVAL_WINDOW)
        if bl is not None:
            out["roofline_ba"] = bl
        if n_cpu:
            state = slam.state_dict()
            cpu_frames = [stream.frame(total + i) for i in range(n_cpu)]
            cpu_frames = [tuple(x.cpu() for x in f) for f in cpu_frames]
            try:
                out["cpu_baseline"] = cpu_baseline(state, args, cfg_kwargs, cpu_frames, args.cpu_steps)
                out["config"]["speedup_vs_cpu_port"] = round(value / out["cpu_baseline"]["value"], 1)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
