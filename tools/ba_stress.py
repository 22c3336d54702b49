#!/usr/bin/env python
"""bitwise reproducibility of fastba.BA on fixed inputs while another stream keeps the GPU busy with the front end"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import fastba, _lib
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
net = make_network("SingleScale")
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
slam.device_steps = False
T = 60
stream = SyntheticStream(480, 640, T + 1, seed=100, device="cuda")
frames = [stream.frame(t) for t in range(T + 1)]
cap = {}
inner = fastba.BA
def spy(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k):
    cap.update(poses=poses.clone(), patches=patches.clone(), intr=intrinsics.clone(), target=target.clone(), weight=weight.clone(),
               lmbda=lmbda, ii=ii.clone(), jj=jj.clone(), kk=kk.clone(), t0=t0, t1=t1, kw=dict(k))
    return inner(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k)
fastba.BA = spy
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = frames[t]
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
fastba.BA = inner
torch.cuda.synchronize()
c = cap
c["kw"]["iterations"] = int(os.environ.get("BA_ITERS", "1"))
print("captured BA: E=%d t0=%d t1=%d" % (c["ii"].shape[0], c["t0"], c["t1"]))
def run():
    p, pt = c["poses"].clone(), c["patches"].clone()
    inner(p, pt, c["intr"], c["target"], c["weight"], c["lmbda"], c["ii"], c["jj"], c["kk"], c["t0"], c["t1"], **c["kw"])
    return p, pt
ref_p, ref_pt = run()
torch.cuda.synchronize()
ws = [v for k, v in _lib._ws_cache.items() if k[-1] == "ba"][0]
ref_ws = ws.clone()
def regions():
    """mirror of csrc/ba.hip::ba_carve (own_groups = 0)"""
    E = c["ii"].shape[0]; N = c["t1"] - c["t0"]; n6 = 6 * N
    plan = c["kw"]["plan"]
    n_poses = c["poses"].numel() // 7; n_patches = c["patches"].numel() // 27
    Mu = min(n_patches, E, max(int(plan.max_kk), 1)); Gp = min(n_poses * n_poses, E, max(int(plan.max_ij), 1))
    tiles = max((n6 + 63) // 64, 1); KS = min(max(256 // (tiles * tiles), 4), 64)
    al = lambda x: (x + 255) // 256 * 256
    out, off = [], 0
    for name, nb in (("pair_ij", Gp * 8), ("counters", 64), ("rec", E * 32 * 4), ("Erow", Mu * n6 * 4), ("Cv", Mu * 4),
                     ("uv", Mu * 4), ("Qv", Mu * 4), ("pairs", Gp * 160 * 4), ("S_part", KS * (n6 * n6 + 1) * 4),
                     ("y_part", KS * (n6 + 1) * 4), ("S", (n6 * n6 + 1) * 4), ("yv", (n6 + 1) * 4), ("dX", (n6 + 1) * 4)):
        out.append((name, off, nb)); off += al(nb)
    return out


def trial(load, n=400):
    side = torch.cuda.Stream()
    bad = 0
    first = None
    stats = {}
    for i in range(n):
        if load == "fe":
            with torch.cuda.stream(side), torch.no_grad():
                im, ev, K, mask = frames[i % T]
                net.patchify(input_=(ev, im, mask), patches_per_image=96, event_bias=True, reinit_hidden=False)
        elif isinstance(load, torch.cuda.CUDAGraph):
            with torch.cuda.stream(side):
                load.replay()
        elif load == "fe_same_stream":
            with torch.no_grad():
                im, ev, K, mask = frames[i % T]
                net.patchify(input_=(ev, im, mask), patches_per_image=96, event_bias=True, reinit_hidden=False)
        elif load == "mm_graph":
            with torch.cuda.stream(side):
                MMG.replay()
        elif load == "fill_graph":
            with torch.cuda.stream(side):
                FG.replay()
        elif load == "fork_graph":
            with torch.cuda.stream(side):
                KG.replay()
        elif load == "lstm":
            with torch.cuda.stream(side), torch.no_grad():
                im, ev, K, mask = frames[i % T]
                from rampvo_amd import conv_hip
                enc = net.patchify.encoder
                for _ in range(3):
                    conv_hip.lstm_superstate_step(enc, ev[0, 0].float().contiguous(), im[0, 0].float().contiguous(), enc._hip_state)
        elif load == "towers":
            with torch.cuda.stream(side), torch.no_grad():
                from rampvo_amd import conv_hip
                enc = net.patchify.encoder
                s16 = enc._hip_state.ss.view(480, 640, 16)
                conv_hip.basic_encoder4(enc.fmap_encoder, s16, 0.25, half=True)
                conv_hip.basic_encoder4(enc.imap_encoder, s16, 0.25, half=True)
        elif load == "select":
            with torch.cuda.stream(side), torch.no_grad():
                from rampvo_amd.utils import get_coords_from_topk_events
                im, ev, K, mask = frames[i % T]
                for _ in range(4):
                    get_coords_from_topk_events(events=ev, patches_per_image=96, border_suppression_size=0, non_max_supp_rad=11)
        elif load == "mm":
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(4):
                    torch.mm(MA, MB, out=MC)
        elif load == "copy":
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(8):
                    BUF2.copy_(BUF1)
        p, pt = run()
        if not (torch.equal(p, ref_p) and torch.equal(pt, ref_pt)):
            bad += 1
            d_ = (ws != ref_ws)
            for nm, o, nb in regions():
                if int(d_[o:o + nb].sum()):
                    stats[nm] = stats.get(nm, 0) + 1
                    break
            if first is None and os.environ.get('VERBOSE'):
                torch.cuda.synchronize()
                d = (ws != ref_ws)
                first = (i, float((p - ref_p).abs().max()), float((pt - ref_pt).abs().max()),
                         {nm: int(d[o:o + nb].sum()) for nm, o, nb in regions() if int(d[o:o + nb].sum())})
                for nm, o, nb in regions():
                    if nm in ("rec", "Erow", "pairs"):
                        a_ = ws[o:o + nb // 4 * 4].view(torch.float32); b_ = ref_ws[o:o + nb // 4 * 4].view(torch.float32)
                        idx = (a_.view(torch.int32) != b_.view(torch.int32)).nonzero().flatten().tolist()
                        if idx:
                            W = {"rec": 32, "Erow": 60, "pairs": 160}[nm]
                            print("   ", nm, "n=%d" % len(idx), [(k // W, k % W, float(a_[k]), float(b_[k])) for k in idx[:16]])
                    if nm in ("y_part", "yv", "uv", "Qv", "S_part", "S"):
                        a_ = ws[o:o + nb // 4 * 4].view(torch.float32); b_ = ref_ws[o:o + nb // 4 * 4].view(torch.float32)
                        idx = (a_ != b_).nonzero().flatten()[:6].tolist()
                        if idx:
                            print("   ", nm, [(k, float(a_[k]), float(b_[k])) for k in idx])
    torch.cuda.synchronize()
    if stats:
        print("    first differing buffer per failing call:", stats)
    return bad, first
MA = torch.randn(4096, 4096, device="cuda", dtype=torch.half); MB = torch.randn(4096, 4096, device="cuda", dtype=torch.half)
MC = torch.empty(4096, 4096, device="cuda", dtype=torch.half)
BUF1 = torch.randn(64 << 20, device="cuda"); BUF2 = torch.empty_like(BUF1)
torch.cuda.synchronize()
print("no concurrent load:", trial(False)[0])
net.patchify.use_graph = False
from rampvo_amd import conv_hip
from rampvo_amd.utils import get_coords_from_topk_events
enc = net.patchify.encoder
EV, IM = frames[5][1], frames[5][0]
evc, imc = EV[0, 0].float().contiguous(), IM[0, 0].float().contiguous()
def cap(fn, warm=True):
    if warm:
        fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    g._keep = keep
    return g
s16 = enc._hip_state.ss.view(480, 640, 16)
with torch.no_grad():
    G_lstm = cap(lambda: conv_hip.lstm_superstate_step(enc, evc, imc, enc._hip_state))
    G_f = cap(lambda: conv_hip.basic_encoder4(enc.fmap_encoder, s16, 0.25, half=True))
    G_i = cap(lambda: conv_hip.basic_encoder4(enc.imap_encoder, s16, 0.25, half=True))
    G_sel = cap(lambda: get_coords_from_topk_events(events=EV, patches_per_image=96, border_suppression_size=0, non_max_supp_rad=11))
    G_all1 = cap(lambda: net.patchify._forward_impl((EV, IM, frames[5][3]), 96, False, None, True, False))
for nm, g in (("imap tower graph", G_i), ("fmap tower graph", G_f), ("imap tower graph", G_i), ("imap tower graph", G_i)):
    print(nm + ":", trial(g)[0], flush=True)
