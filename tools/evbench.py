import time, ctypes, torch
torch.cuda.init()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ev = torch.cuda.Event()
ev.record(s1); torch.cuda.synchronize()
N = 20000
t = time.perf_counter()
for _ in range(N):
    ev.record(s1); s2.wait_event(ev)
torch.cuda.synchronize()
print("torch record + wait_event: %.2f us per pair" % ((time.perf_counter() - t) / N * 1e6))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
e = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0      # hipEventDisableTiming
a, b = s1.cuda_stream, s2.cuda_stream
t = time.perf_counter()
for _ in range(N):
    hip.hipEventRecord(e, a); hip.hipStreamWaitEvent(b, e, 0)
torch.cuda.synchronize()
print("ctypes hipEventRecord + hipStreamWaitEvent: %.2f us per pair" % ((time.perf_counter() - t) / N * 1e6))
cur = torch.cuda.current_stream
t = time.perf_counter()
for _ in range(N):
    cur()
print("torch.cuda.current_stream(): %.2f us" % ((time.perf_counter() - t) / N * 1e6))
