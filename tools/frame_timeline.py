#!/usr/bin/env python
"""steady-state frame timeline from a rocprofv3 kernel trace.
   run:      rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/frame_timeline.py run [mode]
   analyse:  python tools/frame_timeline.py show DIR/.../*_kernel_trace.csv
mode: pipelined (default) | sequential | stream"""
import os, sys, glob, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(mode):
    import torch
    from rampvo_amd.config import make_cfg
    from rampvo_amd.Ramp_vo import Ramp_vo
    from rampvo_amd.synthetic import SyntheticStream, make_network
    N = 160
    slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
    slam.inputs_ready = {"pipelined": True, "sequential": False, "stream": "stream"}[mode]
    st = SyntheticStream(480, 640, N + 1, seed=1234, device="cuda")
    frames = [st.frame(t) for t in range(N)]
    torch.cuda.synchronize()
    for t in range(N):
        im, ev, K, m = frames[t]
        if mode == "stream":
            ev, im = ev * 1.0, im * 1.0
        slam(t, input_tensor=(ev, im, m), intrinsics=K)
    torch.cuda.synchronize()


def show(path, nframes=3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")[:44]
    # frame marker: the correlation kernel
    marks = [i for i, r in enumerate(rows) if "corr_mfma_kernel" in r["Kernel_Name"] or "corr_l1_kernel" in r["Kernel_Name"]]
    if len(marks) < nframes + 12:
        print("too few frames"); return
    per = [int(rows[marks[i + 1]]["Start_Timestamp"]) - int(rows[marks[i]]["Start_Timestamp"]) for i in range(len(marks) - 41, len(marks) - 1)]
    print("frame period over the last 40 frames: mean %.1f us, min %.1f, max %.1f" % (sum(per) / len(per) / 1e3, min(per) / 1e3, max(per) / 1e3))
    a, b = marks[-nframes - 5], marks[-5]
    t0 = int(rows[a]["Start_Timestamp"])
    qs = {}
    for r in rows[a:b]:
        q = r.get("Queue_Id", "?") + "/" + r.get("Stream_Id", "?")
        qs.setdefault(q, len(qs))
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print("%9.1f %8.1f  q%d %s%s" % (s, e - s, qs[q], "    " * qs[q], name(r)))
    print({v: k for k, v in qs.items()})


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "pipelined")
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
