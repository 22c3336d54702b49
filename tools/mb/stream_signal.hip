// What a cross-stream "go" costs on the producer's stream and how late the consumer starts:
//   (a) hipEventRecord between two kernels of stream A + hipStreamWaitEvent on stream B      (what ramp_track_step does)
//   (b) the second kernel of stream A writes a flag itself + hipStreamWaitValue32 on stream B (no packet on stream A)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_signal tools/mb/stream_signal.hip ; run under `timeout 60`
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(long ticks, uint64_t *stamp, uint32_t *flag, uint32_t seq) {   // stamp[0] = start, stamp[1] = end
  const uint64_t t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    stamp[0] = t0;
    if (flag) { __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
  while ((long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}

// (c) what ONE resident wave on a second stream costs a chain of large launches on the first: a pure sleeper, and one
//     that polls a word between sleeps
__global__ void copy_kernel(const uint4 *__restrict__ a, uint4 *__restrict__ b, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void resident_kernel(long ticks, const uint32_t *flag) {
  const uint64_t t0 = wall_clock64();
  uint32_t v = 0;
  while ((long)(wall_clock64() - t0) < ticks) {
    if (flag) v += __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_s_sleep(8);
  }
  if (v == 0xFFFFFFFFu) __builtin_trap();
}
static int resident_cost(hipStream_t A, hipStream_t B, uint32_t *flag) {
  const long n = (64l << 20) / 16;
  uint4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
      CK(hipDeviceSynchronize());
      if (mode) hipLaunchKernelGGL(resident_kernel, dim3(1), dim3(64), 0, B, 200000l, mode == 2 ? flag : (const uint32_t *)nullptr);   // 2 ms
      CK(hipEventRecord(e0, A));
      for (int k = 0; k < 40; k++) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, A, a, b, n);
      CK(hipEventRecord(e1, A));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2 && ms < best) best = ms;
    }
    printf("40 x 64 MB copies on stream A, %-38s %.1f us per launch\n", mode == 0 ? "alone:" : mode == 1 ? "one sleeping wave resident on B:" : "one polling wave resident on B:", best * 1e3 / 40);
  }
  return 0;
}

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  uint64_t *st; CK(hipMalloc(&st, 6 * sizeof(uint64_t)));
  uint32_t *flag; CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const long T = 5000;        // 50 us at 100 MHz
  uint64_t h[6];
  for (int mode = 0; mode < 3; mode++) {
    if (mode == 2 && !can) break;
    std::vector<double> gapA, lagB;
    for (int it = 0; it < 40; it++) {
      const uint32_t seq = 1000 * mode + it + 1;
      hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, T, st, (uint32_t *)nullptr, 0u);
      if (mode == 1) { CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0)); }
      if (mode == 2) CK(hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
      hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, T, st + 2, mode == 2 ? flag : (uint32_t *)nullptr, seq);
      if (mode != 0) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, B, T, st + 4, (uint32_t *)nullptr, 0u);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
      if (it >= 8) { gapA.push_back((double)(h[2] - h[1]) / 100.0); if (mode) lagB.push_back((double)((int64_t)h[4] - (int64_t)h[2]) / 100.0); }
    }
    std::sort(gapA.begin(), gapA.end()); std::sort(lagB.begin(), lagB.end());
    printf("%-44s gap between the two kernels of stream A: median %.1f us", mode == 0 ? "plain" : mode == 1 ? "hipEventRecord + hipStreamWaitEvent" : "flag store in kernel + hipStreamWaitValue32", gapA[gapA.size() / 2]);
    if (mode) printf("   stream B's kernel starts %.1f us after A's second kernel (median; p90 %.1f)", lagB[lagB.size() / 2], lagB[lagB.size() * 9 / 10]);
    printf("\n");
  }
  // (d) a wait on an event that completed long ago, between two kernels of stream A
  {
    hipEvent_t done; CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    std::vector<double> gap;
    for (int it = 0; it < 40; it++) {
      hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, B, 100l, st + 4, (uint32_t *)nullptr, 0u);
      CK(hipEventRecord(done, B));
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, T, st, (uint32_t *)nullptr, 0u);
      CK(hipStreamWaitEvent(A, done, 0));
      hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, T, st + 2, (uint32_t *)nullptr, 0u);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
      if (it >= 8) gap.push_back((double)(h[2] - h[1]) / 100.0);
    }
    std::sort(gap.begin(), gap.end());
    printf("hipStreamWaitEvent on a COMPLETED event between two kernels of stream A: gap median %.1f us\n", gap[gap.size() / 2]);
  }
  return resident_cost(A, B, flag);
}
