#!/usr/bin/env python
"""Hunt for the rare ~50-70 ms stall of the leg behind a state_dict() hand-back (bench.py's `converged` leg: 426 kf/s in r05,
361 in r06_y): N cycles of [state_dict() -> a few frames to re-enter the device-resident state -> K pipelined frames with
COMPACT_COORDS], host return time of every frame; prints every frame slower than 5 ms with its position in the cycle, the
tracker's counters and the lazy copy of the device-side sizes around it.
  python tools/stall_hunt.py [N=20] [K=40] [mode=compact|plain] [snapshot=1|0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import numpy as np
import torch
from rampvo_amd import track_dev
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = sys.argv[3] if len(sys.argv) > 3 else "compact"
snap = sys.argv[4] if len(sys.argv) > 4 else "1"     # 1: state_dict(); 0: nothing; copy: large pageable device-to-host copies only;
                                                      # settle: the hand-back alone; pinned: the same large copies into pinned memory
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale", device=dev),
               {"event_bias": True}, ht=480, wd=640, device=dev)
slam.inputs_ready = True
total = 80 + N * (K + 8 + (int(snap[6:]) if snap.startswith('settle') and len(snap) > 6 else 0)) + 8
stream = SyntheticStream(480, 640, total + 1, seed=1234, device=dev)
frames = [tuple(x.to(dev) if i < 2 else x for i, x in enumerate(stream.frame(t))) for t in range(total)]
pos = [0]
def step():
    im, ev, Kc, mask = frames[pos[0]]
    slam(pos[0], input_tensor=(ev, im, mask), intrinsics=Kc)
    pos[0] += 1
# where a slow frame spends its host time: wall time inside the calls of Ramp_vo._track_device, per frame
import rampvo_amd.Ramp_vo as rv
acc = {}
def timed(obj, name, key):
    inner = getattr(obj, name)
    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return inner(*a, **k)
        finally:
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
    setattr(obj, name, f)
timed(track_dev.DeviceTrack, "throttle", "throttle")
timed(track_dev.DeviceTrack, "step", "dv.step")
timed(track_dev.DeviceTrack, "lazy_state", "lazy_state")
timed(rv.Ramp_vo, "_fe_done_to", "fe_done_to")
timed(rv.Ramp_vo, "_gate_wait", "gate_wait")
timed(rv.Ramp_vo, "_intrinsics_row", "intrinsics_row")
timed(type(slam.network.patchify), "forward", "patchify")
# STALL_SPY=1: a watchdog thread that, when the main thread has been inside one frame for more than 8 ms, prints for every
# thread of the process what the KERNEL says it is doing (/proc/self/task/*/{comm,wchan,syscall,stack}: root on the GPU box)
# and has the main thread print its own user-space backtrace from a signal handler (tools/stall_spy.c)
hb = [time.perf_counter(), 0, False]
if os.environ.get("STALL_SPY", "0") == "1":
    import ctypes, threading, subprocess, signal
    here = os.path.dirname(os.path.abspath(__file__))
    so = "/tmp/stall_spy.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(here, "stall_spy.c")])
    spy = ctypes.CDLL(so)
    spy.spy_install(signal.SIGUSR2)
    main_tid = threading.get_native_id()
    libc = ctypes.CDLL(None, use_errno=True)
    def rd(path):
        try:
            return open(path).read().strip()
        except Exception as e:
            return "<%s>" % type(e).__name__
    def watchdog():
        dumped = -1
        while True:
            time.sleep(0.001)
            if not hb[2] or hb[1] == dumped or time.perf_counter() - hb[0] < 0.008:
                continue
            dumped = hb[1]
            t_in = (time.perf_counter() - hb[0]) * 1e3
            out = ["#### frame %d has lasted %.1f ms; threads of pid %d:" % (hb[1], t_in, os.getpid())]
            for tid in sorted(int(x) for x in os.listdir("/proc/self/task")):
                b = "/proc/self/task/%d/" % tid
                st = [l for l in rd(b + "status").split("\n") if l.startswith("State")]
                out.append("  tid %d%s comm=%s %s wchan=%s syscall=%s" % (tid, " (MAIN)" if tid == main_tid else "", rd(b + "comm"),
                                                                         st[0] if st else "", rd(b + "wchan"), rd(b + "syscall")[:90]))
                ks = rd(b + "stack")
                if ks and not ks.startswith("<"):
                    out.append("      kernel stack: " + " <- ".join(l.split("] ")[-1] for l in ks.split("\n")[:12]))
            sys.stderr.write("\n".join(out) + "\n"); sys.stderr.flush()
            libc.syscall(200 + 34, os.getpid(), main_tid, signal.SIGUSR2)      # tgkill: the main thread prints its backtrace
    threading.Thread(target=watchdog, daemon=True).start()
for _ in range(80):
    step()
gc.collect(); gc.freeze()
torch.cuda.synchronize()
hits = 0
dev_allocs = lambda: (torch.cuda.memory_stats().get("num_device_alloc", -1), torch.cuda.memory_stats().get("num_device_free", -1))
for cyc in range(N):
    a0 = dev_allocs()
    if snap == "1":
        sd = slam.state_dict()
    elif snap == "copy":
        sd = [slam.fmap1_.cpu(), slam.gmap_.cpu(), slam.imap_.cpu(), slam.fmap2_.cpu()]
    elif snap == "settle":
        slam.settle()
    elif snap.startswith("settle"):         # settleN: the hand-back, and N more frames than usual before the timed leg
        slam.settle()
        for _ in range(int(snap[6:])):
            step()
    elif snap == "idle":                    # the GPU idle for 100 ms, nothing else
        torch.cuda.synchronize(); time.sleep(0.1)
    elif snap == "peek":                    # a synchronise + small device-to-host copies, no hand-back
        slam.peek(); _ = slam._dev.dyn.cpu().numpy().copy(); _ = slam._dev.graph[slam._dev.cur][:, :1000].cpu()
    elif snap == "pinned":
        if cyc == 0:
            pins = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in (slam.fmap1_, slam.gmap_, slam.imap_, slam.fmap2_)]
        for pb, t in zip(pins, (slam.fmap1_, slam.gmap_, slam.imap_, slam.fmap2_)):
            pb.copy_(t)
        torch.cuda.synchronize()
    for _ in range(4):
        step()
    slam._extra_step_flags = track_dev.COMPACT_COORDS if mode == "compact" else 0
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    s0 = dict(slam.stats)
    t = [time.perf_counter()]
    lz = []
    parts = []
    hb[2] = True
    for _ in range(K):
        acc.clear()
        hb[0], hb[1] = time.perf_counter(), pos[0]
        step()
        t.append(time.perf_counter())
        parts.append({k: round(v * 1e3, 2) for k, v in acc.items()})
        lz.append(slam._dev.dyn_host.numpy()[[track_dev.DYN_FRAME, track_dev.DYN_E, track_dev.DYN_STATUS]].copy() if slam._dev is not None else None)
    hb[2] = False
    torch.cuda.synchronize()
    tend = time.perf_counter()
    slam._extra_step_flags = 0
    d = np.diff(t) * 1e3
    st = {k: slam.stats[k] - s0[k] for k in s0}
    slow = np.nonzero(d > 5.0)[0]
    a1 = dev_allocs()
    line = "cycle %2d: [hipMalloc %d hipFree %d] %.1f kf/s p50 %.3f max %.2f ms (frame %d of %d) drain %.2f ms | %s" % (
        cyc, a1[0] - a0[0], a1[1] - a0[1], K / (tend - t[0]) / 1e0, np.percentile(d, 50), d.max(), int(d.argmax()), K, (tend - t[-1]) * 1e3,
        {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()})
    if len(slow):
        hits += 1
        line += "  <-- STALL at frames %s, lazy (frame, E, status) around: %s; host ms inside: %s" % (slow.tolist(), [lz[i].tolist() for i in slow[:3]], [parts[i] for i in slow[:3]])
    print(line, flush=True)
print("cycles with a frame > 5 ms: %d of %d" % (hits, N))
