#!/usr/bin/env python
"""microbenchmark of SoftAgg's segment softmax at the bench size: the patch grouping (~2100 groups of ~19 scattered
rows) and the pair grouping (~420 groups of 96 rows); prints us per launch and a checksum of the result (A/B across
RAMP_SEG_X8=0|1 must print the same checksums: the arithmetic is the same)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd._lib import check, lib, ptr, stream
E = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
g = torch.Generator().manual_seed(3)
fg = (torch.randn(E, 768, generator=g) * 2).half().cuda()
F16 = 1
def grouping(ng):
    key = torch.randint(0, ng, (E,), generator=g)
    order = torch.argsort(key, stable=True).int()
    cnt = torch.bincount(key, minlength=ng)
    seg = torch.zeros(ng + 1, dtype=torch.int32); seg[1:] = torch.cumsum(cnt, 0)
    return order.cuda(), seg.cuda(), torch.tensor([ng], dtype=torch.int32).cuda()
from rampvo_amd import _lib
F16 = getattr(_lib, "RAMP_F16", 1)
for name, ng in (("patch grouping", 2112), ("pair grouping", 420)):
    order, seg, n = grouping(ng)
    cap = ng + 96
    y = torch.empty(cap, 384, dtype=torch.half, device="cuda")
    fn = lambda: check(lib().ramp_upd_segment_softmax(ptr(fg), ptr(order), ptr(seg), ptr(n), ptr(y), cap, F16, stream()), "seg")
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        fn()
    e.record(); torch.cuda.synchronize()
    print("%s: %.1f us  crc %08x" % (name, s.elapsed_time(e) / 50 * 1e3, zlib.crc32(y.cpu().numpy().tobytes())), flush=True)
