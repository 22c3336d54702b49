// Lower median of the depth plane of F frames' patches by a four-pass 8-bit radix select in one workgroup
// (reference ramp/Ramp_vo.py:370-371: torch.median(self.patches_[n-3:n, :, 2])).  Shared by csrc/select.hip (frame
// commit) and csrc/lie.hip (the same median computed beside the previous step's motion test, off the next frame's serial
// path).
#pragma once
#include "ramp_device.h"

// lower median of the F*M*PP depth values of src ([F*M][3][PP] rows, channel 2); every thread of the THREADS returns it
// (F*M*PP <= THREADS * PER)
template <int THREADS, int PER>
__device__ __forceinline__ float depth_median_block_t(const float *__restrict__ src, int F, int M, int PP) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_remaining;
  const int tid = threadIdx.x, n = F * M * PP;
  unsigned key[PER];
  bool has[PER];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int i = tid + u * THREADS;
    has[u] = i < n;
    unsigned b = 0;
    if (has[u]) {
      const int fm = i / PP, p = i - fm * PP;               // (frame, patch) pair, pixel
      b = __float_as_uint(src[((size_t)fm * 3 + 2) * PP + p]);
      b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;          // order-preserving map of float to uint
    }
    key[u] = b;
  }
  if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)((n - 1) / 2) + 1; }   // rank of the lower median, 1-based
  unsigned mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
#pragma unroll
    for (int u = 0; u < PER; u++)
      if (has[u] && (key[u] & mask) == prefix) atomicAdd(&hist[(key[u] >> shift) & 255u], 1u);
    __syncthreads();
    if (tid < 64) {
      unsigned c[4];
#pragma unroll
      for (int b = 0; b < 4; b++) c[b] = hist[4 * tid + b];
      const unsigned tot = c[0] + c[1] + c[2] + c[3];
      unsigned pre = tot;                                   // inclusive prefix sum over lanes (ascending bins)
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(pre, o, 64);
        if (tid >= o) pre += v;
      }
      const unsigned rem = s_remaining, below = pre - tot;
      if (below < rem && pre >= rem) {
        unsigned rr = rem - below;
        int bin = 4 * tid + 3;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          if (c[b] >= rr) { bin = 4 * tid + b; break; }
          rr -= c[b];
        }
        s_remaining = rr;
        s_prefix = prefix | ((unsigned)bin << shift);
      }
    }
    mask |= 255u << shift;
    __syncthreads();
  }
  unsigned b = s_prefix;
  b ^= (b >> 31) ? 0x80000000u : 0xffffffffu;             // inverse map
  return __uint_as_float(b);
}

