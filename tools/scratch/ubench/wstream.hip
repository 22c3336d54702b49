// micro-benchmark: every workgroup streams the same W bytes from L2 (fragment-ordered 1 KB pieces per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void __launch_bounds__(512) stream_reg(const u4 *__restrict__ w, int pieces_per_wave, unsigned *out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u4 *src = w + (size_t)wave * pieces_per_wave * 64 + lane;
  u4 acc = {0, 0, 0, 0};
  for (int p = 0; p < pieces_per_wave; p += DEPTH) {
    u4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) v[d] = src[(size_t)(p + d) * 64];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc ^= v[d];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int DEPTH>
__global__ void __launch_bounds__(512) stream_lds(const u4 *__restrict__ w, int pieces_per_wave, unsigned *out) {
  __shared__ u4 ring[8][DEPTH][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u4 *src = w + (size_t)wave * pieces_per_wave * 64 + lane;
  u4 acc = {0, 0, 0, 0};
  for (int p = 0; p < pieces_per_wave; p += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_global_load_lds(src + (size_t)(p + d) * 64, &ring[wave][d][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      ;
#endif
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc ^= ring[wave][d][lane];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

int main() {
  const int pieces_per_wave = 216;                       // 8 waves x 216 KB = 1.77 MB per workgroup
  const size_t bytes = (size_t)8 * pieces_per_wave * 1024;
  u4 *w; unsigned *out;
  hipMalloc(&w, bytes); hipMalloc(&out, 4);
  hipMemset(w, 1, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char *name, auto kern, int blocks) {
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, w, pieces_per_wave, out);
    hipEventRecord(a);
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, w, pieces_per_wave, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    const double per_cu_rounds = (double)((blocks + 255) / 256);
    printf("%-14s blocks %4d: %8.1f us  -> %6.1f GB/s aggregate, %5.1f B/clk/CU (2.4 GHz, %g rounds)\n", name, blocks,
           ms * 1e3, blocks * (double)bytes / ms / 1e6, bytes * per_cu_rounds / (ms * 1e-3) / 2.4e9, per_cu_rounds);
  };
  for (int blocks : {256, 512}) {
    run("reg depth 1", stream_reg<1>, blocks);
    run("reg depth 3", stream_reg<3>, blocks);
    run("reg depth 6", stream_reg<6>, blocks);
    run("reg depth 12", stream_reg<12>, blocks);
    run("lds depth 3", stream_lds<3>, blocks);
    run("lds depth 6", stream_lds<6>, blocks);
    run("lds depth 12", stream_lds<12>, blocks);
  }
  return 0;
}
