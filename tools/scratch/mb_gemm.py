import torch, torch.nn.functional as F
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
E = 40000
for K in (882, 896, 1024):
    x = torch.randn(E, K, device="cuda").half(); w = torch.randn(384, K, device="cuda").half(); b = torch.randn(384, device="cuda").half()
    print("K=%d  linear %.1f us   addmm_act(relu) %.1f us" % (K, timeit(lambda: F.linear(x, w, b)), timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))))
xp = torch.randn(E, 896, device="cuda").half()
xv = xp[:, :882]   # strided view (lda 896), K=882
w = torch.randn(384, 882, device="cuda").half(); b = torch.randn(384, device="cuda").half()
print("K=882 lda=896 view: %.1f us" % timeit(lambda: F.linear(xv, w, b)))
x = torch.randn(E, 384, device="cuda").half()
for N in (384, 768):
    w = torch.randn(N, 384, device="cuda").half(); b = torch.randn(N, device="cuda").half()
    print("384->%d: %.1f us" % (N, timeit(lambda: F.linear(x, w, b))))
