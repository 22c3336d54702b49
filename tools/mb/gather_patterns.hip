// Microbenchmark: how fast does one CU's vector L1 deliver 16-byte-per-lane loads for the address patterns of the
// correlation kernel?  Every wave issues batches of 16 global_load_dwordx4 (1 KB per instruction) at pseudo-random
// window positions of an L2-resident feature plane set and adds the data up (so that nothing is optimised away).
//   P0  one contiguous KB per instruction
//   P1  4 segments of 256 B (one per 8-channel chunk q, 16 x-neighbours): an aligned 16-wide window row
//   P2  the kernel's pattern: chunked layout, 10-wide window -> per chunk q the 16 lanes cover 1.6 window rows
//   P3  NHWC: lane (q, j) = pixel j (256 B apart), 16-byte piece q of a 64-byte step
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_patterns tools/mb/gather_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int H = 120, W = 160, CH = 16;          // plane: [H][16 chunks][W][8 halfs] = 4.9 MB
constexpr size_t PLANE = (size_t)H * CH * W * 16; // bytes
constexpr int NPLANES = 8;                        // one per XCD; a workgroup stays inside a 48-row band of "its" plane (2 MB: L2 resident)

template <int PAT>
__global__ void __launch_bounds__(64) gather_kernel(const uint4 *__restrict__ buf, float *__restrict__ out, int iters) {
  const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
  unsigned h = blockIdx.x * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    const int plane = blockIdx.x % NPLANES;          // workgroups are dealt round-robin to the 8 XCDs
    const int y0 = (h >> 12) % 36, x0 = (h >> 20) % (W - 32);
    const char *base = reinterpret_cast<const char *>(buf) + plane * PLANE;
    uint4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int g = u >> 2, s = u & 3;               // pixel group g (4 per batch), K step s
      size_t off;
      if (PAT == 0) {
        off = ((size_t)(y0 + g) * CH * W + (size_t)s * 4 * W + x0) * 16 + (size_t)lane * 16;
      } else if (PAT == 1) {
        off = (((size_t)(y0 + g) * CH + 4 * s + q) * W + x0 + j) * 16;
      } else if (PAT == 2) {
        const int t = g * 16 + j, ty = t / 10, tx = t - ty * 10;
        off = (((size_t)(y0 + ty) * CH + 4 * s + q) * W + x0 + tx) * 16;
      } else if (PAT == 5) {
        // [H][4 K-steps][W][32 halfs]: a window row of a K step is 640 contiguous bytes
        const int t = g * 16 + j, ty = t / 10, tx = t - ty * 10;
        off = ((((size_t)(y0 + ty) * 4 + s) * W + x0 + tx) * 4 + q) * 16;
      } else if (PAT == 6) {
        // [4 K-steps][H][W][32 halfs]: the same, K-step planes apart
        const int t = g * 16 + j, ty = t / 10, tx = t - ty * 10;
        off = ((((size_t)s * H + y0 + ty) * W + x0 + tx) * 4 + q) * 16;
      } else if (PAT == 7) {
        // [H][2][W][64 halfs] with two K steps per 128-byte pixel slot: lane (q, j) takes 16 bytes at 32 * q + 16 * (s & 1)
        const int t = g * 16 + j, ty = t / 10, tx = t - ty * 10;
        off = ((((size_t)(y0 + ty) * 2 + (s >> 1)) * W + x0 + tx) * 8 + 2 * q + (s & 1)) * 16;
      } else {
        const int t = g * 16 + j, ty = t / 10, tx = t - ty * 10;
        off = ((size_t)(y0 + ty) * W + x0 + tx) * 256 + (size_t)(4 * s + q) * 16;   // NHWC: 256 B per pixel
      }
      v[u] = *reinterpret_cast<const uint4 *>(base + off);
    }
#pragma unroll
    for (int u = 0; u < 16; u++) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;
}

// P4: NHWC planes, a 10 x 10 window = 10 runs of 2560 contiguous bytes, moved with global_load_lds_dwordx4 (1 KB per
// instruction: 4 pixels x 256 B, LDS destination lane-linear, the 16-byte chunk of a lane XOR-swizzled on the source side
// with the pixel index), then read back as MFMA B fragments (lane (q, j): pixel 16 g + j, chunk 4 s + q) with the same
// XOR: conflict-free ds_read_b128.  28 KB of LDS per wave.
__global__ void __launch_bounds__(64) gather_lds_kernel(const uint4 *__restrict__ buf, float *__restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
  unsigned h = blockIdx.x * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    const int plane = blockIdx.x % NPLANES;
    const int y0 = (h >> 12) % 36, x0 = (h >> 20) % (W - 32);
    const char *base = reinterpret_cast<const char *>(buf) + plane * PLANE;
#pragma unroll
    for (int i = 0; i < 25; i++) {
      const int pix = 4 * i + q;                     // lane (q, j) here: pixel slot q of the instruction, chunk j
      const int ty = pix / 10, tx = pix - ty * 10;
      const char *src = base + ((size_t)(y0 + ty) * W + x0 + tx) * 256 + ((j ^ (pix & 15)) * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(lds + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int g = 0; g < 6; g++)
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int P = 16 * g + j, ch = 4 * s + q;
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + P * 256 + ((ch ^ (P & 15)) * 16));
        acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w);
      }
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;
}

static void run_lds(const uint4 *buf, float *out, int waves_per_cu) {
  const int waves = 256 * waves_per_cu * 8, iters = 8;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const size_t lds = (size_t)(160 * 1024) / waves_per_cu / 1024 * 1024;      // forces waves_per_cu workgroups per CU
  (void)hipFuncSetAttribute((const void *)gather_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gather_lds_kernel, dim3(waves), dim3(64), lds, 0, buf, out, iters);
  (void)hipEventRecord(a);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(gather_lds_kernel, dim3(waves), dim3(64), lds, 0, buf, out, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = 5.0 * waves * iters * 25 * 1024;
  printf("P4 NHWC rows -> LDS (glds) -> B fragments, %d waves/CU  %7.1f us / launch   %6.2f TB/s   %5.1f B/clk/CU\n", waves_per_cu,
         ms * 1e3 / 5, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

template <int PAT>
static void run(const uint4 *buf, float *out, const char *name) {
  const int waves = 256 * 16 * 8, iters = 8;       // 16 waves per CU resident, 8 rounds
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(gather_kernel<PAT>, dim3(waves), dim3(64), 0, 0, buf, out, iters);
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(gather_kernel<PAT>, dim3(waves), dim3(64), 0, 0, buf, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = 5.0 * waves * iters * 16 * 1024;
  printf("%-44s %7.1f us / launch   %6.2f TB/s   %5.1f B/clk/CU (2.4 GHz)\n", name, ms * 1e3 / 5, bytes / (ms * 1e-3) / 1e12,
         bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  uint4 *buf; float *out;
  hipMalloc(&buf, PLANE * NPLANES + (1 << 20));
  hipMalloc(&out, 1 << 22);
  hipMemset(buf, 1, PLANE * NPLANES + (1 << 20));
  run<0>(buf, out, "P0 contiguous 1 KB");
  run<1>(buf, out, "P1 4 x 256 B (aligned 16-wide rows)");
  run<2>(buf, out, "P2 chunked, 10-wide window (the kernel)");
  run<3>(buf, out, "P3 NHWC, 10-wide window");
  run<5>(buf, out, "P5 [H][4][W][32], 10-wide window");
  run<6>(buf, out, "P6 [4][H][W][32], 10-wide window");
  run<7>(buf, out, "P7 [H][2][W][64] (half a line per lane pair)");
  run<0>(buf, out, "P0 again");
  run_lds(buf, out, 5);
  run_lds(buf, out, 4);
  run_lds(buf, out, 3);
  return 0;
}
