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
  return 0;
}
