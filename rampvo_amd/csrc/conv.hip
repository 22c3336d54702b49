// RAMP encoder kernels for gfx950.
//
//  1. lstm_superstate_mfma_kernel: the two per-pixel LSTM cells (events 5->15, image 3->15, state
//     carried) and the shared super-state 1x1 convolution, fused into ONE kernel on fp32 MFMA
//     (reference: cuDNN LSTM over 307,200 length-1 sequences + two conv launches + host syncs on
//     torch.any; ramp/extractor.py:233-259).  Recurrent state is tile-major [HW/16][16][16], the
//     super-state leaves as channels-last [H*W][16] (channel 15 = 0) for the conv towers.
//
//  2. conv_mfma_kernel: implicit-GEMM convolution on the matrix cores.  NHWC activations, one
//     wavefront = 32 output pixels x 32 output channels = 2x2 tiles of v_mfma_f32_16x16x4_f32
//     (exact fp32, k-ordered fmaf chain) or v_mfma_f32_16x16x32_f16.  No LDS: the K axis is
//     permuted so that every lane's A fragment for 4 (fp32) / 1 (f16) MFMA steps is ONE 16-byte
//     channels-last load of its own pixel, and the weights are pre-packed host-side in fragment
//     order so B is one coalesced 16-byte load per lane (L2-resident, shared by all blocks).
//     Prologue (optional): per-channel affine + ReLU applied while loading the input, i.e. the
//     previous layer's InstanceNorm + ReLU are never materialised.  Epilogue: bias, optional ReLU,
//     optional residual add + ReLU, output scale, and per-block partial sums / sums of squares
//     per channel for the next InstanceNorm (deterministic two-stage reduction, no atomics).
//
//  3. in_stats_finalize_kernel, norm_add_relu_kernel: InstanceNorm statistics -> (scale, shift),
//     and the residual-block tail relu(skip' + relu(norm(y))).
#include "ramp_device.h"
#include <stdlib.h>
#include <type_traits>

// ------------------------------------------------------------------ any != 0
#define ANY_MAXB 1024          // workgroups of any_nonzero_kernel (per-workgroup results: flags [2][ANY_MAXB])
__global__ void __launch_bounds__(256)
    any_nonzero_kernel(const float *__restrict__ a, long na, const float *__restrict__ b, long nb,
                       int *__restrict__ flags, int per_block, uint4 *__restrict__ clear, long clear_n) {
  __shared__ int s_any[2];
  if (threadIdx.x < 2) s_any[threadIdx.x] = 0;
  __syncthreads();
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  // (the front end's first launch also zeroes the InstanceNorm accumulators of the frame: no memset launch)
  for (long k = i / 4; k < clear_n; k += stride / 4) clear[k] = make_uint4(0u, 0u, 0u, 0u);
  bool fa = false, fb = false;
  for (long k = i; k < na; k += stride) {
    if (k + 3 < na) {
      const float4 v = *reinterpret_cast<const float4 *>(a + k);
      fa |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    } else {
      for (long t = k; t < na; t++) fa |= (a[t] != 0.0f);
    }
  }
  for (long k = i; k < nb; k += stride) {
    if (k + 3 < nb) {
      const float4 v = *reinterpret_cast<const float4 *>(b + k);
      fb |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    } else {
      for (long t = k; t < nb; t++) fb |= (b[t] != 0.0f);
    }
  }
  // one (benign, idempotent) plain store per workgroup instead of an atomic per wavefront
  if (__any(fa) && (threadIdx.x & 63) == 0) s_any[0] = 1;
  if (__any(fb) && (threadIdx.x & 63) == 0) s_any[1] = 1;
  __syncthreads();
  if (per_block) {                               // every workgroup reports: nothing to clear beforehand
    if (threadIdx.x < 2) flags[threadIdx.x * ANY_MAXB + blockIdx.x] = s_any[threadIdx.x];
  } else if (threadIdx.x < 2 && s_any[threadIdx.x]) {
    flags[threadIdx.x] = 1;
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvParams {
  const void *x;        // NHWC [H][W][Cin]
  const void *wpk;      // packed weights (fragment order)
  const float *bias;    // [Cout] or null
  const float *pre_scale, *pre_shift;  // [Cin] or null: x <- relu(x*scale+shift) on load
  const void *res;      // NHWC [OH][OW][Cout] residual (added before the final ReLU) or null
  void *y;              // NHWC [OH][OW][Cout]
  float *stats;         // [Cout][2][nblk] partial (sum, sumsq) of the raw (bias-added) output or null
  // accumulator mode (LDS-tiled fp16 kernels): the statistics travel as exact fixed-point integers -- every workgroup
  // adds its partial sums to one of IN_ACC_R replicas (order independent, so bit reproducible), every CONSUMER
  // workgroup sums the replicas and forms (scale, shift) itself: no finalize launch between two layers
  unsigned long long *acc_out;         // [IN_ACC_R][Cout][2] of this layer's output or null
  const unsigned long long *acc_in;    // [IN_ACC_R][Cin][2] of the input (instead of pre_scale / pre_shift) or null
  float in_count, in_eps;
  int H, W, Cin, OH, OW, Cout;
  int relu;             // ReLU on the conv output (before the residual add; the add is followed by its own ReLU)
  float out_scale;
  // fp8 MFMA variant of the LDS-tiled kernel: operands are x * act_scale and w * w_scale rounded to OCP e4m3
  // (saturating at +-448), the fp32 accumulator is multiplied by descale = 1 / (act_scale * w_scale) before the bias
  float act_scale, descale;
  // LDS-tiled fp16 kernel only: channels [C0, Cin) of the input come from a SECOND tensor x2 [H][W][Cin - C0] (x is then
  // [H][W][C0]) -- the MultiScale towers' torch.cat((x, x_down2), channel) without the copy (ramp/extractor.py:300, 306)
  const void *x2;
  int C0;
  // LDS-tiled fp16 kernel only: the residual block's tail applied while the NEXT layer loads its input -- the input pixel is
  // relu(relu(x * scale + shift) + skip') with skip' = skip, skip * scale_s + shift_s (acc_skip) or the fp16-rounded
  // relu of that (skip_relu): norm_add_relu_f16_acc_kernel's expression, rounded to fp16 once as that kernel's output is.
  // `mat`: the workgroups of a stride-1 layer also write the pixels they own [H][W][Cin], for the block that needs them as
  // its skip (ramp/extractor.py:49-57; one launch per residual block less)
  const void *skip;
  const unsigned long long *acc_skip;
  float skip_count, skip_eps;
  int skip_relu;
  void *mat;
};
// up to two independent problems of one layer shape in one launch (blockIdx.z): the towers of the encoder
struct ConvMulti { ConvParams t[2]; };

#define IN_ACC_R 8
#define IN_ACC_ONE 1048576.0          // fixed point: 2^20 per unit (a tile's sum of squares stays below 2^43)
__device__ __forceinline__ unsigned long long in_acc_fix(float v) { return (unsigned long long)(long long)rint((double)v * IN_ACC_ONE); }
// (scale, shift) of channels [0, C) from the replicated accumulators into LDS tables; all threads of the workgroup call
// it, a barrier follows inside
__device__ __forceinline__ void in_acc_finalize(const unsigned long long *__restrict__ acc, int C, float count, float eps,
                                                float *s_scale, float *s_shift, long long *s_sum /* [2 * C] */) {
  const int tid = threadIdx.x, nth = blockDim.x;
  for (int i = tid; i < 2 * C; i += nth) {
    long long t = 0;
#pragma unroll
    for (int r = 0; r < IN_ACC_R; r++) t += (long long)acc[(size_t)r * 2 * C + i];
    s_sum[i] = t;
  }
  __syncthreads();
  for (int c = tid; c < C; c += nth) {
    const double mean = (double)s_sum[2 * c] / IN_ACC_ONE / (double)count;
    double var = (double)s_sum[2 * c + 1] / IN_ACC_ONE / (double)count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    s_scale[c] = (float)rstd;
    s_shift[c] = (float)(-mean * rstd);
  }
  __syncthreads();
}

// fp32: wave tile 32 px x 32 ch, K chunk = 16 input channels per tap.
// A fragment (v_mfma_f32_16x16x4_f32): lane l supplies A[i=l&15][k=l>>4]; with the channel
// permutation c = 4*(l>>4) + step one float4 load feeds 4 MFMA steps.
template <int KH, int KW, int STRIDE>
__global__ void __launch_bounds__(256)
    conv_mfma_f32_kernel(const ConvParams p) {
  constexpr int PAD = KH / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int M = p.OH * p.OW;
  const int m_wave = (blockIdx.x * 4 + wave) * 32;
  const int n0 = blockIdx.y * 32;
  const float *x = reinterpret_cast<const float *>(p.x);
  const float *wpk = reinterpret_cast<const float *>(p.wpk);
  const int nchunk = p.Cin / 16;

  // the two pixels (one per m-tile) whose activations this lane loads
  int oy[2], ox[2];
  bool mval[2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    const int m = m_wave + mt * 16 + j;
    mval[mt] = m < M;
    const int mm = mval[mt] ? m : 0;
    oy[mt] = mm / p.OW;
    ox[mt] = mm - oy[mt] * p.OW;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // packed weights: [tap][chunk][ntile(Cout/16)][lane(64)][4]
  const int ntiles = p.Cout / 16;
  for (int ky = 0; ky < KH; ky++) {
    for (int kx = 0; kx < KW; kx++) {
      const int tap = ky * KW + kx;
      size_t aoff[2];
      bool aval[2];
#pragma unroll
      for (int mt = 0; mt < 2; mt++) {
        const int iy = oy[mt] * STRIDE + ky - PAD, ix = ox[mt] * STRIDE + kx - PAD;
        aval[mt] = mval[mt] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        aoff[mt] = ((size_t)(aval[mt] ? iy : 0) * p.W + (aval[mt] ? ix : 0)) * p.Cin + 4 * q;
      }
      for (int ch = 0; ch < nchunk; ch++) {
        float4 a[2], b[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
          a[mt] = aval[mt] ? *reinterpret_cast<const float4 *>(x + aoff[mt] + ch * 16)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (p.pre_scale) {
          const float4 sc = *reinterpret_cast<const float4 *>(p.pre_scale + ch * 16 + 4 * q);
          const float4 sh = *reinterpret_cast<const float4 *>(p.pre_shift + ch * 16 + 4 * q);
#pragma unroll
          for (int mt = 0; mt < 2; mt++) {
            if (aval[mt]) {
              a[mt].x = fmaxf(a[mt].x * sc.x + sh.x, 0.f);
              a[mt].y = fmaxf(a[mt].y * sc.y + sh.y, 0.f);
              a[mt].z = fmaxf(a[mt].z * sc.z + sh.z, 0.f);
              a[mt].w = fmaxf(a[mt].w * sc.w + sh.w, 0.f);
            }
          }
        }
        const float *wb = wpk + (((size_t)(tap * nchunk + ch) * ntiles + (n0 / 16)) * 64 + lane) * 4;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) b[nt] = *reinterpret_cast<const float4 *>(wb + nt * 256);
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b[nt].x, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b[nt].y, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b[nt].z, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b[nt].w, acc[mt][nt], 0, 0, 0);
          }
      }
    }
  }
  // ---- epilogue.  D layout: lane holds rows 4q..4q+3 (pixels) of column j (channel)
  __shared__ float s_stat[4][32][2];
  float *y = reinterpret_cast<float *>(p.y);
  const float *res = reinterpret_cast<const float *>(p.res);
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const int c = n0 + nt * 16 + j;
    const float bv = p.bias ? p.bias[c] : 0.0f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m_wave + mt * 16 + 4 * q + r;
        if (m < M) {
          float v = acc[mt][nt][r] + bv;
          s1 += v;
          s2 += v * v;
          if (p.relu) v = fmaxf(v, 0.f);
          if (res) v = fmaxf(v + res[(size_t)m * p.Cout + c], 0.f);
          y[(size_t)m * p.Cout + c] = v * p.out_scale;
        }
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (q == 0) { s_stat[wave][nt * 16 + j][0] = s1; s_stat[wave][nt * 16 + j][1] = s2; }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (threadIdx.x < 64) {
      const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
      const float v = ((s_stat[0][c][k] + s_stat[1][c][k]) + s_stat[2][c][k]) + s_stat[3][c][k];
      p.stats[((size_t)(n0 + c) * 2 + k) * gridDim.x + blockIdx.x] = v;   // [C][2][nblk]: contiguous per channel
    }
  }
}

// fp16 storage / fp16 MFMA variant (MIXED_PRECISION): activations and weights are half, the
// accumulators, bias, InstanceNorm statistics and the prologue affine are fp32.
//   IN_F32 = true : first layer (Cin = 16, fp32 super-state in): one tap per
//                   v_mfma_f32_16x16x16_f16, lane loads float4 -> 4 halves
//   IN_F32 = false: Cin % 32 == 0, half in: one (tap, 32-channel chunk) per
//                   v_mfma_f32_16x16x32_f16, lane loads 8 halves (16 B)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int KH, int KW, int STRIDE, bool IN_F32>
__global__ void __launch_bounds__(256)
    conv_mfma_f16_kernel(const ConvParams p) {
  constexpr int PAD = KH / 2;
  constexpr int KC = IN_F32 ? 16 : 32;   // channels consumed per MFMA
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int M = p.OH * p.OW;
  const int m_wave = (blockIdx.x * 4 + wave) * 32;
  const int n0 = blockIdx.y * 32;
  const int nchunk = p.Cin / KC;
  const int ntiles = p.Cout / 16;
  int oy[2], ox[2];
  bool mval[2];
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    const int m = m_wave + mt * 16 + j;
    mval[mt] = m < M;
    const int mm = mval[mt] ? m : 0;
    oy[mt] = mm / p.OW;
    ox[mt] = mm - oy[mt] * p.OW;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int CPL = KC / 4;   // channels per lane per MFMA (4 or 8)

  for (int ky = 0; ky < KH; ky++) {
    for (int kx = 0; kx < KW; kx++) {
      const int tap = ky * KW + kx;
      size_t aoff[2];
      bool aval[2];
#pragma unroll
      for (int mt = 0; mt < 2; mt++) {
        const int iy = oy[mt] * STRIDE + ky - PAD, ix = ox[mt] * STRIDE + kx - PAD;
        aval[mt] = mval[mt] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        aoff[mt] = ((size_t)(aval[mt] ? iy : 0) * p.W + (aval[mt] ? ix : 0)) * p.Cin + CPL * q;
      }
      for (int ch = 0; ch < nchunk; ch++) {
        float av[2][CPL];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
          if (!aval[mt]) {
#pragma unroll
            for (int c = 0; c < CPL; c++) av[mt][c] = 0.f;
          } else if (IN_F32) {
            const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.x) + aoff[mt] + ch * KC);
            av[mt][0] = v.x; av[mt][1] = v.y; av[mt][2] = v.z; av[mt][3] = v.w;
          } else {
            const f16x8 v = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(p.x) + aoff[mt] + ch * KC);
#pragma unroll
            for (int c = 0; c < CPL; c++) av[mt][c] = (float)v[c];
          }
        }
        if (p.pre_scale) {
#pragma unroll
          for (int c = 0; c < CPL; c++) {
            const float sc = p.pre_scale[ch * KC + CPL * q + c], sh = p.pre_shift[ch * KC + CPL * q + c];
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
              if (aval[mt]) av[mt][c] = fmaxf(av[mt][c] * sc + sh, 0.f);
          }
        }
        const _Float16 *wb = reinterpret_cast<const _Float16 *>(p.wpk) +
                             (((size_t)(tap * nchunk + ch) * ntiles + (n0 / 16)) * 64 + lane) * CPL;
        if (IN_F32) {
          f16x4 a[2], b[2];
#pragma unroll
          for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int c = 0; c < 4; c++) a[mt][c] = (_Float16)av[mt][c];
#pragma unroll
          for (int nt = 0; nt < 2; nt++) b[nt] = *reinterpret_cast<const f16x4 *>(wb + nt * 64 * CPL);
#pragma unroll
          for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        } else {
          f16x8 a[2], b[2];
#pragma unroll
          for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int c = 0; c < 8; c++) a[mt][c] = (_Float16)av[mt][c < CPL ? c : 0];
#pragma unroll
          for (int nt = 0; nt < 2; nt++) b[nt] = *reinterpret_cast<const f16x8 *>(wb + nt * 64 * CPL);
#pragma unroll
          for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
  __shared__ float s_stat[4][32][2];
  _Float16 *y = reinterpret_cast<_Float16 *>(p.y);
  const _Float16 *res = reinterpret_cast<const _Float16 *>(p.res);
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const int c = n0 + nt * 16 + j;
    const float bv = p.bias ? p.bias[c] : 0.0f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m_wave + mt * 16 + 4 * q + r;
        if (m < M) {
          float v = acc[mt][nt][r] + bv;
          s1 += v;
          s2 += v * v;
          if (p.relu) v = fmaxf(v, 0.f);
          if (res) v = fmaxf(v + (float)res[(size_t)m * p.Cout + c], 0.f);
          y[(size_t)m * p.Cout + c] = (_Float16)(v * p.out_scale);
        }
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (q == 0) { s_stat[wave][nt * 16 + j][0] = s1; s_stat[wave][nt * 16 + j][1] = s2; }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (threadIdx.x < 64) {
      const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
      const float v = ((s_stat[0][c][k] + s_stat[1][c][k]) + s_stat[2][c][k]) + s_stat[3][c][k];
      p.stats[((size_t)(n0 + c) * 2 + k) * gridDim.x + blockIdx.x] = v;   // [C][2][nblk]: contiguous per channel
    }
  }
}

// epilogue of the LDS-tiled kernels: bias, InstanceNorm partial statistics, ReLU -> fp32 tile in LDS (aliases the
// input / weight tiles: the caller has passed a barrier after its last read of them) -> residual, scale, coalesced
// 16-byte stores.  acc[mt][nt]: rows 2 wave + mt of the 8 x 16 tile, 16-channel tile nt of the block starting at n0.
template <int NT, int TH = 8>
__device__ __forceinline__ void conv_tile_epilogue(const ConvParams &p, const f32x4 (&acc)[TH / 4][NT], unsigned char *smem,
                                                   float (&s_stat)[4][NT * 16][2], int oy0, int ox0, int n0) {
  constexpr int TW = 16, MTW = TH / 4;                // (m-tiles per wave: rows MTW wave + mt of the TH x 16 tile)
  constexpr int OSTR = NT * 16 + 4;                   // fp32 staging row (floats), 16-byte aligned
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
  float *s_out = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int c = n0 + nt * 16 + j;
    const float bv = p.bias ? p.bias[c] : 0.0f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
      const int r = MTW * wave + mt;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int xx = 4 * q + rr;
        float v = acc[mt][nt][rr] + bv;
        if (oy0 + r < p.OH && ox0 + xx < p.OW) { s1 += v; s2 += v * v; }
        if (p.relu) v = fmaxf(v, 0.f);
        s_out[(r * TW + xx) * OSTR + nt * 16 + j] = v;
      }
    }
    if (p.stats || p.acc_out) {
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (q == 0) { s_stat[wave][nt * 16 + j][0] = s1; s_stat[wave][nt * 16 + j][1] = s2; }
    }
  }
  __syncthreads();
  if ((p.stats || p.acc_out) && tid < NT * 32) {
    const int c = tid >> 1, k = tid & 1;
    const float v = ((s_stat[0][c][k] + s_stat[1][c][k]) + s_stat[2][c][k]) + s_stat[3][c][k];
    if (p.acc_out)
#ifdef HZ_SKIP_ATOM                                   // (diagnostic: what the statistics' atomics cost)
      { if (v == 123.456f) p.acc_out[0] = 1; }
#else
      atomicAdd(p.acc_out + ((size_t)(blockIdx.x % IN_ACC_R) * p.Cout + n0 + c) * 2 + k, in_acc_fix(v));
#endif
    else
      p.stats[((size_t)(n0 + c) * 2 + k) * gridDim.x + blockIdx.x] = v;   // [C][2][nblk]: contiguous per channel
  }
  // 16-byte pieces: pixel-major, 8 channels each
  constexpr int PPP = NT * 2;                         // pieces per pixel
  for (int i = tid; i < TH * TW * PPP; i += 256) {
    const int pix = i / PPP, piece = i - pix * PPP;
    const int r = pix / TW, xx = pix - r * TW;
    const int oy = oy0 + r, ox = ox0 + xx;
    if (oy >= p.OH || ox >= p.OW) continue;
    const float *sv = s_out + pix * OSTR + piece * 8;
    const size_t go = ((size_t)oy * p.OW + ox) * p.Cout + n0 + piece * 8;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = sv[c];
    if (p.res) {
      const f16x8 rv = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const _Float16 *>(p.res) + go);
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = fmaxf(v[c] + (float)rv[c], 0.f);
    }
    f16x8 h;
#pragma unroll
    for (int c = 0; c < 8; c++) h[c] = (_Float16)(v[c] * p.out_scale);
#ifdef HZ_SKIP_STORE                                  // (diagnostic: what the output stores cost)
    if (v[0] == 123.456f)
#endif
    *reinterpret_cast<f16x8 *>(reinterpret_cast<_Float16 *>(p.y) + go) = h;
  }
}

// dynamic LDS bytes of conv_tile_f16_kernel<K, S, IN_F32, CIN, NT, FP8> (the kernel asserts the same number)
constexpr int conv_tile_lds_bytes(int K, int S, bool IN_F32, int CIN, int NT, bool FP8, int TH = 8) {
  const int TW = 16, SP = K == 1 ? 1 : S;
  const int IH = (TH - 1) * SP + K, IW = (TW - 1) * SP + K;
  const int PSTR = FP8 ? CIN + 8 : CIN * 2 + 16;
  const int KC = IN_F32 ? 16 : 32, NCH = CIN / KC, FRAG = (IN_F32 || FP8) ? 8 : 16;
  const int WBYTES = K * K * NCH * NT * 64 * FRAG, IBYTES = IH * IW * PSTR;
  const int OBYTES = TH * TW * (NT * 16 + 4) * 4;
  return (WBYTES + IBYTES) > OBYTES ? (WBYTES + IBYTES) : OBYTES;
}
template <typename KernelT>
static int conv_tile_attr(KernelT kernel, int lds) {
  if (lds <= 64 * 1024) return RAMP_OK;
  return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess ? RAMP_OK : RAMP_ELAUNCH;
}

// LDS-tiled fp16 variant: the layers of these towers are tiny (32/64 channels, <= 77k pixels), so
// the direct kernel above spends its time on one global round trip per tap and on re-applying
// the InstanceNorm prologue 9 (49) times per input value.  Here a workgroup owns an 8 x 16 output
// tile:
//   1. every thread issues ALL its 16-byte loads of the (halo) input tile and of the weight
//      fragments at once, applies relu(x*scale+shift) once per value and parks them in LDS;
//   2. the K*K*Cin/32 MFMA steps then run from LDS only (wave w = output rows 2w, 2w+1; the A
//      fragment of lane (q, j) is pixel (row, j), channels 8q.. of the chunk: one ds_read_b128;
//      pixel stride Cin*2+16 bytes keeps the 16 lanes of a quarter-wave on distinct banks);
//   3. bias / statistics / ReLU in registers, then the tile goes through LDS once more (fp32) so
//      the residual is read and the result written as contiguous 16-byte pieces.
// Same accumulation order as the direct kernel => identical results.
template <int K, int S, bool IN_F32, int CIN, int NT, bool FP8 = false, int TH = 8, bool TAIL = false>
__global__ void __launch_bounds__(256) conv_tile_f16_kernel(const ConvMulti pm) {
  static_assert(!TAIL || (!IN_F32 && !FP8 && K != 7), "the fused block tail: f16 instances behind the first layer");
  static_assert(!(FP8 && IN_F32), "the fp32-input first layer stays on the f16 MFMA");
  const ConvParams &p = pm.t[blockIdx.z];
  if ((int)blockIdx.y * NT * 16 >= p.Cout) return;    // the towers may differ in Cout (grid.y = the larger one's)
  constexpr int PAD = K / 2, TW = 16, MTW = TH / 4;   // (TH = 16: half the weight staging and 1.27 instead of 1.41 halo reads per
                                                      // output pixel; for the layers whose tile count still fills the chip)
  constexpr int SP = K == 1 ? 1 : S;                  // tile-pixel step between output neighbours
  constexpr int STEP = K == 1 ? S : 1;                // image-pixel step between tile pixels
  constexpr int IH = (TH - 1) * SP + K, IW = (TW - 1) * SP + K;
  constexpr int PSTR = FP8 ? CIN + 8 : CIN * 2 + 16;  // LDS bytes per tile pixel (fp8: 1 byte per channel)
  constexpr int KC = IN_F32 ? 16 : 32, NCH = CIN / KC;
  constexpr int FRAG = (IN_F32 || FP8) ? 8 : 16;      // bytes per lane per B fragment
  constexpr int CPI = IN_F32 ? 4 : 8;                 // channels per 16-byte global item
  constexpr int CH8 = CIN / CPI;                      // items per pixel
  constexpr int NITEM = IH * IW * CH8, NI = (NITEM + 255) / 256;
  constexpr int WBYTES = K * K * NCH * NT * 64 * FRAG, NW = (WBYTES / 16 + 255) / 256;
  constexpr int IBYTES = IH * IW * PSTR;
  constexpr int OSTR = NT * 16 + 4;                   // fp32 staging row (floats), 16-byte aligned
  constexpr int OBYTES = TH * TW * OSTR * 4;
  constexpr int SMB = (WBYTES + IBYTES) > OBYTES ? (WBYTES + IBYTES) : OBYTES;
  static_assert(256 % CH8 == 0, "a thread keeps one channel slot");
  static_assert(SMB == conv_tile_lds_bytes(K, S, IN_F32, CIN, NT, FP8, TH), "launch-side size out of date");
  // (dynamic: the stride-2 64-channel layer of the MultiScale towers needs 118 KB -- above the static limit)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_stat[4][NT * 16][2];
  unsigned char *s_w = smem, *s_in = smem + WBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
  const int tiles_x = (p.OW + TW - 1) / TW;
  const int ty0 = blockIdx.x / tiles_x, tx0 = blockIdx.x - ty0 * tiles_x;
  const int oy0 = ty0 * TH, ox0 = tx0 * TW;
  const int n0 = blockIdx.y * NT * 16, ntiles = p.Cout / 16;

  // ---- 1. all global loads first (native vector types: the arrays must stay in registers)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 wbuf[NW];
#pragma unroll
  for (int n = 0; n < NW; n++) {
    const int i = tid + n * 256;
    constexpr int SEG = NT * 64 * FRAG / 16;        // 16-byte pieces per (tap, chunk) of this n-block
    const int ic = i < WBYTES / 16 ? i : 0;
    const int seg = ic / SEG, within = ic - seg * SEG;
    wbuf[n] = reinterpret_cast<const u32x4 *>(p.wpk)[((size_t)seg * ntiles + n0 / 16) * (64 * FRAG / 16) + within];
  }
  u32x4 ibuf[NI];
  bool iok[NI];
  const int cslot = tid % CH8;
#pragma unroll
  for (int n = 0; n < NI; n++) {
    const int i = tid + n * 256;
    const int pix = i / CH8, ty = pix / IW, tx = pix - ty * IW;
    const int gy = oy0 * S + ty * STEP - PAD, gx = ox0 * S + tx * STEP - PAD;
    iok[n] = i < NITEM && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int cy = iok[n] ? gy : 0, cx = iok[n] ? gx : 0;      // always a valid address; masked below
    if (!IN_F32 && p.x2) {                                    // (uniform) two sources: the channel slot picks one
      const int s0 = p.C0 / CPI;
      const bool second = cslot >= s0;
      const u32x4 *src = reinterpret_cast<const u32x4 *>(second ? p.x2 : p.x);
      ibuf[n] = src[((size_t)cy * p.W + cx) * (second ? CH8 - s0 : s0) + (second ? cslot - s0 : cslot)];
    } else {
      ibuf[n] = reinterpret_cast<const u32x4 *>(p.x)[((size_t)cy * p.W + cx) * CH8 + cslot];
    }
  }
  // (uniform) fused block tail: the skip operand of every tile pixel, in flight with the rest
  // (TAIL instances: a launch in which one of the towers takes a fused block tail; the other instances keep their registers)
  constexpr int NS = TAIL ? NI : 1;
  const bool tail = TAIL && p.skip != nullptr;
  u32x4 kbuf[NS];
  if (tail) {
#pragma unroll
    for (int n = 0; n < NS; n++) {
      const int i = tid + n * 256;
      const int pix = i / CH8, ty = pix / IW, tx = pix - ty * IW;
      const int gy = oy0 * S + ty * STEP - PAD, gx = ox0 * S + tx * STEP - PAD;
      const bool ok = i < NITEM && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      kbuf[n] = reinterpret_cast<const u32x4 *>(p.skip)[((size_t)(ok ? gy : 0) * p.W + (ok ? gx : 0)) * CH8 + cslot];
    }
  }
  float sc[8], sh[8], sc2[8], sh2[8];
  const bool pre = p.pre_scale != nullptr || p.acc_in != nullptr;
  if (p.acc_in) {                                     // (uniform) the input's InstanceNorm from its accumulators
    float *s_sc = reinterpret_cast<float *>(s_in), *s_sh = s_sc + 128;
    long long *s_sum = reinterpret_cast<long long *>(s_sh + 128);
    in_acc_finalize(p.acc_in, p.Cin, p.in_count, p.in_eps, s_sc, s_sh, s_sum);
#pragma unroll
    for (int c = 0; c < CPI; c++) { sc[c] = s_sc[cslot * CPI + c]; sh[c] = s_sh[cslot * CPI + c]; }
    if (tail && p.acc_skip) {                         // (uniform) the skip's own InstanceNorm (conv1's, the downsample path's)
      __syncthreads();
      in_acc_finalize(p.acc_skip, p.Cin, p.skip_count, p.skip_eps, s_sc, s_sh, s_sum);
#pragma unroll
      for (int c = 0; c < CPI; c++) { sc2[c] = s_sc[cslot * CPI + c]; sh2[c] = s_sh[cslot * CPI + c]; }
    }
    __syncthreads();                                  // the tables sit where the input tile goes
  } else if (pre) {
#pragma unroll
    for (int c = 0; c < CPI; c++) { sc[c] = p.pre_scale[cslot * CPI + c]; sh[c] = p.pre_shift[cslot * CPI + c]; }
  }
#pragma unroll
  for (int n = 0; n < NW; n++) {
    const int i = tid + n * 256;
    if (i < WBYTES / 16) reinterpret_cast<u32x4 *>(s_w)[i] = wbuf[n];
  }
#pragma unroll
  for (int n = 0; n < NI; n++) {
    const int i = tid + n * 256;
    const int pix = i / CH8;
    if (IN_F32) {
      const f32x4 f = __builtin_bit_cast(f32x4, ibuf[n]);
      f16x4 h;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float v = pre ? fmaxf(f[c] * sc[c] + sh[c], 0.f) : f[c];
        h[c] = iok[n] ? (_Float16)v : (_Float16)0.f;
      }
      if (i < NITEM) *reinterpret_cast<f16x4 *>(s_in + pix * PSTR + cslot * 8) = h;
    } else if (FP8) {
      const f16x8 h = __builtin_bit_cast(f16x8, ibuf[n]);
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float t = pre ? fmaxf((float)h[c] * sc[c] + sh[c], 0.f) : (float)h[c];
        v[c] = iok[n] ? fminf(fmaxf(t * p.act_scale, -448.f), 448.f) : 0.f;
      }
      int lo = 0, hi = 0;                               // 8 x e4m3, channel c in byte c
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
      if (i < NITEM) *reinterpret_cast<int2 *>(s_in + pix * PSTR + cslot * 8) = make_int2(lo, hi);
    } else {
      f16x8 h = __builtin_bit_cast(f16x8, ibuf[n]);
#pragma unroll
      for (int c = 0; c < 8; c++) {
        float v = pre ? fmaxf((float)h[c] * sc[c] + sh[c], 0.f) : (float)h[c];
        if (tail) {
          float b = (float)__builtin_bit_cast(f16x8, kbuf[n < NS ? n : 0])[c];
          if (p.acc_skip) b = b * sc2[c] + sh2[c];
          if (p.skip_relu) b = (float)(_Float16)fmaxf(b, 0.f);
          v = fmaxf(v + b, 0.f);
        }
        h[c] = iok[n] ? (_Float16)v : (_Float16)0.f;
      }
      if (TAIL && tail && p.mat && S == 1) {
        const int ty = pix / IW, tx = pix - ty * IW;
        if (i < NITEM && iok[n] && blockIdx.y == 0 && ty >= PAD && ty < PAD + TH && tx >= PAD && tx < PAD + TW)
          reinterpret_cast<f16x8 *>(p.mat)[((size_t)(oy0 + ty - PAD) * p.W + ox0 + tx - PAD) * CH8 + cslot] = h;
      }
      if (i < NITEM) *reinterpret_cast<f16x8 *>(s_in + pix * PSTR + cslot * 16) = h;
    }
  }
  __syncthreads();

  // ---- 2. MFMA from LDS   (HZ_*: diagnostic builds for tools/mb/pk_hazard.hip only)
  f32x4 acc[MTW][NT];
#pragma unroll
  for (int a = 0; a < MTW; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned char *a_base = s_in + ((MTW * wave) * SP * IW + j * SP) * PSTR + q * ((IN_F32 || FP8) ? 8 : 16);
  const unsigned char *b_base = s_w + lane * FRAG;
#ifndef HZ_SKIP_MMA
#pragma unroll
  for (int ky = 0; ky < K; ky++) {
#pragma unroll
    for (int kx = 0; kx < K; kx++) {
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) {
        const int tap = ky * K + kx;
        if (IN_F32) {
          f16x4 a[MTW], b[NT];
#pragma unroll
          for (int mt = 0; mt < MTW; mt++)
            a[mt] = *reinterpret_cast<const f16x4 *>(a_base + ((mt * SP + ky) * IW + kx) * PSTR + ch * KC * 2);
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
            b[nt] = *reinterpret_cast<const f16x4 *>(b_base + ((tap * NCH + ch) * NT + nt) * 64 * FRAG);
#pragma unroll
          for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        } else if (FP8) {
          long a[MTW], b[NT];
#pragma unroll
          for (int mt = 0; mt < MTW; mt++)
            a[mt] = *reinterpret_cast<const long *>(a_base + ((mt * SP + ky) * IW + kx) * PSTR + ch * KC);
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
            b[nt] = *reinterpret_cast<const long *>(b_base + ((tap * NCH + ch) * NT + nt) * 64 * FRAG);
#pragma unroll
          for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        } else {
          f16x8 a[MTW], b[NT];
#pragma unroll
          for (int mt = 0; mt < MTW; mt++)
            a[mt] = *reinterpret_cast<const f16x8 *>(a_base + ((mt * SP + ky) * IW + kx) * PSTR + ch * KC * 2);
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
            b[nt] = *reinterpret_cast<const f16x8 *>(b_base + ((tap * NCH + ch) * NT + nt) * 64 * FRAG);
#pragma unroll
          for (int mt = 0; mt < MTW; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
#endif
  __syncthreads();          // every wave is done with the input / weight tiles

  if (FP8) {
#pragma unroll
    for (int a = 0; a < MTW; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[a][b][r] *= p.descale;
  }
#ifdef HZ_SKIP_EPI
  if (acc[0][0][0] + acc[MTW - 1][NT - 1][3] == 123.456f) reinterpret_cast<_Float16 *>(p.y)[tid] = (_Float16)acc[0][0][1];
#else
  conv_tile_epilogue<NT, TH>(p, acc, smem, s_stat, oy0, ox0, n0);
#endif
}

// fp32 towers on the f16 matrix cores with EXACT-class products (round 6): the fp32 towers ran on v_mfma_f32_16x16x4_f32 through
// the direct kernel at the top of this file -- one global round trip per tap, 40 us per 3x3 layer -- and the front end's 1.0 ms
// next to a 1.7 ms step cost the step 25 % by contention.  Here every fp32 operand is split into THREE fp16 numbers,
//     x = x0 + x1 2^-11 + x2 2^-22,   x0 = fp16(x),  x1 = fp16((x - x0) 2^11),  x2 = fp16((x - x0 - x1 2^-11) 2^22)
// (33 significant bits; every part has the magnitude of x, so no subnormal operand arises from the split), and a product is
// the six MFMA products of order <= 2 into three fp32 accumulators, one per order:
//     acc0 += x0 w0;   acc1 += x0 w1 + x1 w0;   acc2 += x0 w2 + x1 w1 + x2 w0;   result = acc0 + 2^-11 acc1 + 2^-22 acc2
// (dropped: order 3 and above, 2^-33).  The two-part split of csrc/update_x3.hip (22-bit operands, three products) measures
// the same against fp64 but is NOT the reference's arithmetic closely enough here: its free-running trajectory sits 7e-6
// instead of 7e-7 from the reference's own run and two of 2,880 depths leave the 1e-4 bound (tools/traj_fp32_conv_cmp.py) --
// the reference multiplies exactly and rounds only sums, and so does this kernel.
// A workgroup owns an 8 x 16 output tile of ALL output channels:
//   * the fp32 halo tile is loaded once, relu(x scale + shift) applied once per value, the three parts parked in LDS as three
//     planes (pixel stride CIN 2 + 16 bytes: conflict-free ds_read_b128 A fragments);
//   * weights are packed per layer as fp16 fragments of the three parts of W 2^s (s: max |W| 2^s in [2^12, 2^13)), plus the
//     exact inverse 2^-s behind the pack (rampvo_amd/conv_hip.py::pack_conv_weight mode "x3");
//   * a wave owns one 16-channel tile of the output and 8 (COUT = 64) or 4 (COUT = 32) of the tile's rows, so every weight
//     fragment is fetched from L2 once per workgroup and the A fragments come from LDS;
//   * bias, InstanceNorm partial sums (sum, sum of squares of the raw output over the tile's valid pixels -> stats[C][2][nblk],
//     the direct kernel's contract), ReLU, residual + ReLU, out_scale on the accumulators.
#ifndef CONV_X3_PF
#define CONV_X3_PF 1      // weight fragments one trip (two steps) ahead
#endif
#ifndef CONV_X3_PAD
#define CONV_X3_PAD 0     // bytes of padding per tile pixel and plane: 16 spreads the A-fragment reads over the banks, 0 lets more
                          // workgroups share a CU (7x7 layer 112 -> 75 KB) -- front end alone 672 -> 643 us (profiles/r06_convpad_ab.txt)
#endif
template <int K, int S, int CIN, int COUT>
__global__ void __launch_bounds__(256) conv_x3_kernel(const ConvParams p) {
  constexpr int PAD = K / 2, TH = 8, TW = 16;
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int KC = CIN >= 32 ? 32 : 16, NCH = CIN / KC, CPL = KC / 4;   // channels per MFMA step, steps per tap, halfs per lane
  constexpr int PSTR = CIN * 2 + CONV_X3_PAD;                           // LDS bytes per tile pixel and plane
  constexpr int PLANE = IH * IW * PSTR;
  constexpr int NT = COUT / 16, WPN = 4 / NT, MTW = TH / WPN;           // waves per channel tile, rows (m-tiles) per wave
  constexpr int CH4 = CIN / 4, NITEM = IH * IW * CH4;
  static_assert(NT == 2 || NT == 4, "32 or 64 output channels");
  static_assert(256 % CH4 == 0, "a thread keeps one channel slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_stat[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
  const int tiles_x = (p.OW + TW - 1) / TW;
  const int ty0 = blockIdx.x / tiles_x, tx0 = blockIdx.x - ty0 * tiles_x;
  const int oy0 = ty0 * TH, ox0 = tx0 * TW;
  const int nt = wave % NT, rg = wave / NT;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));

  // ---- 1. the halo tile: fp32 -> three fp16 planes in LDS, in batches of 8 sixteen-byte loads per thread
  const float *x = reinterpret_cast<const float *>(p.x);
  const int cslot = tid % CH4;
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  const bool pre = p.pre_scale != nullptr;
  if (pre) {
#pragma unroll
    for (int c = 0; c < 4; c++) { sc[c] = p.pre_scale[cslot * 4 + c]; sh[c] = p.pre_shift[cslot * 4 + c]; }
  }
  constexpr int BATCH = 8;
  for (int n0 = 0; n0 < NITEM; n0 += 256 * BATCH) {
    f32x4 v[BATCH];
    bool ok[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; b++) {
      const int i = n0 + b * 256 + tid;
      const int pix = i / CH4, ty = pix / IW, tx = pix - ty * IW;
      const int gy = oy0 * S + ty - PAD, gx = ox0 * S + tx - PAD;
      ok[b] = i < NITEM && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      v[b] = *reinterpret_cast<const f32x4 *>(x + ((size_t)(ok[b] ? gy : 0) * p.W + (ok[b] ? gx : 0)) * CIN + cslot * 4);
    }
#pragma unroll
    for (int b = 0; b < BATCH; b++) {
      const int i = n0 + b * 256 + tid;
      if (i >= NITEM) continue;
      const int pix = i / CH4;
      h4 h0, h1, h2;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float t = pre ? fmaxf(v[b][c] * sc[c] + sh[c], 0.f) : v[b][c];
        t = ok[b] ? t : 0.f;
        _Float16 a0 = (_Float16)t;
        if (fabsf(t) < 6.103515625e-05f) a0 = (_Float16)0.f;            // (no subnormal operand is relied on)
        const float r1 = t - (float)a0;                                  // exact
        const _Float16 a1 = (_Float16)(r1 * 2048.0f);
        const float r2 = r1 - (float)a1 * 0.00048828125f;                // exact
        h0[c] = a0; h1[c] = a1; h2[c] = (_Float16)(r2 * 4194304.0f);
      }
      *reinterpret_cast<h4 *>(smem + pix * PSTR + cslot * 8) = h0;
      *reinterpret_cast<h4 *>(smem + PLANE + pix * PSTR + cslot * 8) = h1;
      *reinterpret_cast<h4 *>(smem + 2 * PLANE + pix * PSTR + cslot * 8) = h2;
    }
  }
  __syncthreads();

  // ---- 2. six MFMAs per product from LDS (A) and L2 (this wave's weight fragments)
  f32x4 acc0[MTW], acc1[MTW], acc2[MTW];
#pragma unroll
  for (int a = 0; a < MTW; a++) { acc0[a] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[a] = acc0[a]; acc2[a] = acc0[a]; }
  const _Float16 *wph = reinterpret_cast<const _Float16 *>(p.wpk);
  constexpr size_t WTOT = (size_t)K * K * NCH * NT * 64 * CPL;         // halfs per plane of the pack
  const float descale = *reinterpret_cast<const float *>(wph + 3 * WTOT);
  const int a_off = ((rg * MTW) * S * IW + j * S) * PSTR + q * CPL * 2;
  // One flat loop over (tap, chunk), two steps per trip (fully unrolled the compiler hoists every weight load of the layer --
  // 256 VGPRs).  The weight fragments of the NEXT trip are requested before this trip's MFMAs are issued (CONV_X3_PF, round 6):
  // with one wave per SIMD nothing else covers the L2 round trip of a trip's fragments, and a workgroup's life was the sum
  // of its trips' latencies -- 25 trips for the 7x7 layer
  typedef typename std::conditional<KC == 32, f16x8, f16x4>::type wv_t;
  struct WStep { wv_t w0, w1, w2; };
  const int nsteps = K * K * NCH + (p.Cin - CIN);      // (= K K NCH; not a compile-time constant: no full unrolling)
  auto ldw = [&](int step) -> WStep {
    const int sc_ = step < nsteps ? step : nsteps - 1;                   // (past the end: a valid address, never used)
    const size_t wo = (((size_t)sc_ * NT + nt) * 64 + lane) * CPL;
    return WStep{*reinterpret_cast<const wv_t *>(wph + wo), *reinterpret_cast<const wv_t *>(wph + WTOT + wo),
                 *reinterpret_cast<const wv_t *>(wph + 2 * WTOT + wo)};
  };
  auto run = [&](int step, const WStep &w) {
    const int tap = step / NCH, ch = step - tap * NCH;
    const int ky = tap / K, kx = tap - ky * K;
    const int o0 = a_off + (ky * IW + kx) * PSTR + ch * KC * 2;
#pragma unroll
    for (int mt = 0; mt < MTW; mt++) {
      const int o = o0 + mt * S * IW * PSTR;
      const wv_t a0 = *reinterpret_cast<const wv_t *>(smem + o), a1 = *reinterpret_cast<const wv_t *>(smem + PLANE + o),
                 a2 = *reinterpret_cast<const wv_t *>(smem + 2 * PLANE + o);
      if constexpr (KC == 32) {
        acc0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, w.w0, acc0[mt], 0, 0, 0);
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, w.w1, acc1[mt], 0, 0, 0);
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, w.w0, acc1[mt], 0, 0, 0);
        acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, w.w2, acc2[mt], 0, 0, 0);
        acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, w.w1, acc2[mt], 0, 0, 0);
        acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, w.w0, acc2[mt], 0, 0, 0);
      } else {
        acc0[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, w.w0, acc0[mt], 0, 0, 0);
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, w.w1, acc1[mt], 0, 0, 0);
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, w.w0, acc1[mt], 0, 0, 0);
        acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, w.w2, acc2[mt], 0, 0, 0);
        acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, w.w1, acc2[mt], 0, 0, 0);
        acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a2, w.w0, acc2[mt], 0, 0, 0);
      }
    }
  };
#if CONV_X3_PF
  WStep wa = ldw(0), wb = ldw(1);
  for (int step = 0; step < nsteps; step += 2) {
    const WStep na = ldw(step + 2), nb = ldw(step + 3);
    run(step, wa);
    if (step + 1 < nsteps) run(step + 1, wb);
    wa = na; wb = nb;
  }
#else
#pragma unroll 2
  for (int step = 0; step < nsteps; step++) run(step, ldw(step));
#endif

  // ---- 3. epilogue on the accumulators: lane (q, j) holds pixels x = 4q .. 4q+3 of row rg MTW + mt, channel 16 nt + j
  const int c = nt * 16 + j;
  const float bv = p.bias ? p.bias[c] : 0.0f;
  float *y = reinterpret_cast<float *>(p.y);
  const float *res = reinterpret_cast<const float *>(p.res);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < MTW; mt++) {
    const int oy = oy0 + rg * MTW + mt;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int ox = ox0 + 4 * q + rr;
      if (oy < p.OH && ox < p.OW) {
        const float lo = acc1[mt][rr] + acc2[mt][rr] * 0.00048828125f;                  // (small terms first)
        float v = (acc0[mt][rr] + lo * 0.00048828125f) * descale + bv;
        s1 += v;
        s2 += v * v;
        if (p.relu) v = fmaxf(v, 0.f);
        const size_t go = ((size_t)oy * p.OW + ox) * COUT + c;
        if (res) v = fmaxf(v + res[go], 0.f);
        y[go] = v * p.out_scale;
      }
    }
  }
  if (p.stats) {
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    if (q == 0) { s_stat[wave][j][0] = s1; s_stat[wave][j][1] = s2; }
    __syncthreads();
    if (tid < COUT * 2) {
      const int cc = tid >> 1, k = tid & 1, t = cc >> 4, jj = cc & 15;
      float v = s_stat[t][jj][k];                         // the waves of channel tile t: t, t + NT, ... in this order
#pragma unroll
      for (int w = 1; w < WPN; w++) v += s_stat[t + w * NT][jj][k];
      p.stats[((size_t)cc * 2 + k) * gridDim.x + blockIdx.x] = v;   // [C][2][nblk]
    }
  }
}
constexpr int conv_x3_lds_bytes(int K, int S, int CIN) { return 3 * ((8 - 1) * S + K) * ((16 - 1) * S + K) * (CIN * 2 + CONV_X3_PAD); }

// First layer of BOTH towers in one workgroup (7x7 stride 2, 16 fp32 input channels -> 32 channels per tower): the two
// towers read the same super-state, so the halo tile is staged once and feeds four 16-channel output tiles.  Unlike
// the generic tiled kernel the weight fragments are not parked in LDS (49 taps x 4 tiles = 100 KB) but streamed from
// L2 through a ring of registers CONV7_PF taps deep -- LDS holds the 37 KB input tile only, four workgroups per CU
// instead of one, and the 600 tiles of a 640x480 frame are one round (the generic kernel ran 2 x 600 workgroups at one
// per CU: 41 us).  Same tap order and accumulation as conv_tile_f16_kernel<7, 2, true, 16, 2>: identical results.
#ifndef CONV7_PF
#define CONV7_PF 4
#endif
__global__ void __launch_bounds__(256) conv7_dual_kernel(const ConvMulti pm) {
  constexpr int K = 7, S = 2, PAD = 3, TH = 8, TW = 16, NT = 2;
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int CIN = 16, PSTR = CIN * 2 + 16;        // LDS bytes per tile pixel
  constexpr int CH8 = CIN / 4;                        // 16-byte (4 x fp32) items per pixel
  constexpr int NITEM = IH * IW * CH8, NI = (NITEM + 255) / 256;
  constexpr int IBYTES = IH * IW * PSTR;
  constexpr int OBYTES = TH * TW * (NT * 16 + 4) * 4;
  constexpr int SMB = IBYTES > OBYTES ? IBYTES : OBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMB];
  __shared__ float s_stat[4][NT * 16][2];
  const ConvParams &p = pm.t[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
  const int tiles_x = (p.OW + TW - 1) / TW;
  const int ty0 = blockIdx.x / tiles_x, tx0 = blockIdx.x - ty0 * tiles_x;
  const int oy0 = ty0 * TH, ox0 = tx0 * TW;

  // ---- 1. the halo tile: all loads first, fp32 -> fp16 into LDS
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ibuf[NI];
  bool iok[NI];
  const int cslot = tid % CH8;
#pragma unroll
  for (int n = 0; n < NI; n++) {
    const int i = tid + n * 256;
    const int pix = i / CH8, ty = pix / IW, tx = pix - ty * IW;
    const int gy = oy0 * S + ty - PAD, gx = ox0 * S + tx - PAD;
    iok[n] = i < NITEM && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int cy = iok[n] ? gy : 0, cx = iok[n] ? gx : 0;
    ibuf[n] = reinterpret_cast<const u32x4 *>(p.x)[((size_t)cy * p.W + cx) * CH8 + cslot];
  }
  // ---- the first taps' weight fragments: lane fragment = 8 bytes at ((tap * 2 + nt) * 64 + lane) * 8 of a tower's pack
  f16x4 ring[CONV7_PF + 1][2][NT];
  const unsigned char *w0 = reinterpret_cast<const unsigned char *>(pm.t[0].wpk) + lane * 8;
  const unsigned char *w1 = reinterpret_cast<const unsigned char *>(pm.t[1].wpk) + lane * 8;
  auto wfrag = [&](int tap, int t, int nt) {
    return *reinterpret_cast<const f16x4 *>((t ? w1 : w0) + (size_t)(tap * NT + nt) * 512);
  };
#pragma unroll
  for (int d = 0; d < CONV7_PF; d++)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) ring[d][t][nt] = wfrag(d, t, nt);
#pragma unroll
  for (int n = 0; n < NI; n++) {
    const int i = tid + n * 256;
    const int pix = i / CH8;
    const f32x4 f = __builtin_bit_cast(f32x4, ibuf[n]);
    f16x4 h;
#pragma unroll
    for (int c = 0; c < 4; c++) h[c] = iok[n] ? (_Float16)f[c] : (_Float16)0.f;
    if (i < NITEM) *reinterpret_cast<f16x4 *>(smem + pix * PSTR + cslot * 8) = h;
  }
  __syncthreads();

  // ---- 2. 49 taps x (2 row tiles x 4 channel tiles) MFMA from LDS / the ring
  f32x4 acc[2][2][NT];                                // [tower][mt][nt]
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < NT; b++) acc[t][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned char *a_base = smem + ((2 * wave) * S * IW + j * S) * PSTR + q * 8;
#pragma unroll
  for (int tap = 0; tap < K * K; tap++) {
    if (tap + CONV7_PF < K * K) {
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) ring[(tap + CONV7_PF) % (CONV7_PF + 1)][t][nt] = wfrag(tap + CONV7_PF, t, nt);
    }
#ifndef CONV7_NOSB
    __builtin_amdgcn_sched_barrier(0);
#endif
    const int ky = tap / K, kx = tap - ky * K;
    f16x4 a[2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
      a[mt] = *reinterpret_cast<const f16x4 *>(a_base + ((mt * S + ky) * IW + kx) * PSTR);
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
          acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[mt], ring[tap % (CONV7_PF + 1)][t][nt], acc[t][mt][nt], 0, 0, 0);
  }
  __syncthreads();          // every wave is done with the input tile
  conv_tile_epilogue<NT>(pm.t[0], acc[0], smem, s_stat, oy0, ox0, 0);
  __syncthreads();          // the staging tile and the statistics table are reused
  conv_tile_epilogue<NT>(pm.t[1], acc[1], smem, s_stat, oy0, ox0, 0);
}

// InstanceNorm statistics: partial[nblk][C][2] -> scale = rstd, shift = -mean*rstd (biased variance)
__global__ void __launch_bounds__(64)
    in_stats_finalize_kernel(const float *__restrict__ partial, int nblk, int C, float count, float eps,
                             float *__restrict__ scale, float *__restrict__ shift) {
  const int c = blockIdx.x, lane = threadIdx.x;
  // partial[C][2][nblk]: a channel's partials are contiguous -> coalesced, 4 loads in flight per lane
  const float *p1 = partial + ((size_t)c * 2 + 0) * nblk, *p2 = partial + ((size_t)c * 2 + 1) * nblk;
  float s1 = 0.f, s2 = 0.f;
  for (int b0 = 0; b0 < nblk; b0 += 256) {
    float a[4], q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int b = b0 + u * 64 + lane;
      a[u] = b < nblk ? p1[b] : 0.f;
      q[u] = b < nblk ? p2[b] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) { s1 += a[u]; s2 += q[u]; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
  if (lane == 0) {
    const float mean = s1 / count;
    float var = s2 / count - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = 1.0f / sqrtf(var + eps);
    scale[c] = rstd;
    shift[c] = -mean * rstd;
  }
}

// out = relu( f(skip) + relu(y*sy + hy) ),  f(skip) = skip*ss + hs when given (norm3 of the
// downsample path, no ReLU) else skip.  NHWC, C multiple of 4.
__global__ void __launch_bounds__(256)
    norm_add_relu_kernel(const float *__restrict__ y, const float *__restrict__ sy,
                         const float *__restrict__ hy, const float *__restrict__ skip,
                         const float *__restrict__ ss, const float *__restrict__ hs,
                         float *__restrict__ out, long n4, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  float4 v = reinterpret_cast<const float4 *>(y)[i];
  const float4 a = *reinterpret_cast<const float4 *>(sy + c), b = *reinterpret_cast<const float4 *>(hy + c);
  v.x = fmaxf(v.x * a.x + b.x, 0.f); v.y = fmaxf(v.y * a.y + b.y, 0.f);
  v.z = fmaxf(v.z * a.z + b.z, 0.f); v.w = fmaxf(v.w * a.w + b.w, 0.f);
  float4 k = reinterpret_cast<const float4 *>(skip)[i];
  if (ss) {
    const float4 e = *reinterpret_cast<const float4 *>(ss + c), f = *reinterpret_cast<const float4 *>(hs + c);
    k.x = k.x * e.x + f.x; k.y = k.y * e.y + f.y; k.z = k.z * e.z + f.z; k.w = k.w * e.w + f.w;
  }
  v.x = fmaxf(v.x + k.x, 0.f); v.y = fmaxf(v.y + k.y, 0.f);
  v.z = fmaxf(v.z + k.z, 0.f); v.w = fmaxf(v.w + k.w, 0.f);
  reinterpret_cast<float4 *>(out)[i] = v;
}

// half-storage variants of the two tail kernels (8 channels / 16 B per lane)
__global__ void __launch_bounds__(256)
    norm_add_relu_f16_kernel(const _Float16 *__restrict__ y, const float *__restrict__ sy,
                             const float *__restrict__ hy, const _Float16 *__restrict__ skip,
                             const float *__restrict__ ss, const float *__restrict__ hs,
                             _Float16 *__restrict__ out, long n8, int C, int skip_relu) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)((i * 8) % C);
  const f16x8 v = reinterpret_cast<const f16x8 *>(y)[i];
  const f16x8 k = reinterpret_cast<const f16x8 *>(skip)[i];
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    float a = fmaxf((float)v[e] * sy[c + e] + hy[c + e], 0.f);
    float b = (float)k[e];
    if (ss) b = b * ss[c + e] + hs[c + e];
    // skip = relu(norm(.)) that was never materialised: rounded to half as the materialised tensor would be
    if (skip_relu) b = (float)(_Float16)fmaxf(b, 0.f);
    o[e] = (_Float16)fmaxf(a + b, 0.f);
  }
  reinterpret_cast<f16x8 *>(out)[i] = o;
}
// the same with the statistics of y (and of a normalised skip) taken from their accumulators: every workgroup forms the
// (scale, shift) tables itself (in_acc_finalize)
__global__ void __launch_bounds__(256)
    norm_add_relu_f16_acc_kernel(const _Float16 *__restrict__ y, const unsigned long long *__restrict__ acc_y, float count_y,
                                 float eps_y, const _Float16 *__restrict__ skip,
                                 const unsigned long long *__restrict__ acc_s, float count_s, float eps_s,
                                 _Float16 *__restrict__ out, long n8, int C, int skip_relu) {
  __shared__ float s_tab[4][128];
  __shared__ long long s_sum[256];
  in_acc_finalize(acc_y, C, count_y, eps_y, s_tab[0], s_tab[1], s_sum);
  if (acc_s) in_acc_finalize(acc_s, C, count_s, eps_s, s_tab[2], s_tab[3], s_sum);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)((i * 8) % C);
  const f16x8 v = reinterpret_cast<const f16x8 *>(y)[i];
  const f16x8 k = reinterpret_cast<const f16x8 *>(skip)[i];
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    float a = fmaxf((float)v[e] * s_tab[0][c + e] + s_tab[1][c + e], 0.f);
    float b = (float)k[e];
    if (acc_s) b = b * s_tab[2][c + e] + s_tab[3][c + e];
    if (skip_relu) b = (float)(_Float16)fmaxf(b, 0.f);
    o[e] = (_Float16)fmaxf(a + b, 0.f);
  }
  reinterpret_cast<f16x8 *>(out)[i] = o;
}
// (scale, shift) arrays from accumulators, for a consumer without the accumulator path
__global__ void __launch_bounds__(128)
    in_acc_to_scale_shift_kernel(const unsigned long long *__restrict__ acc, int C, float count, float eps,
                                 float *__restrict__ scale, float *__restrict__ shift) {
  __shared__ float s_tab[2][128];
  __shared__ long long s_sum[256];
  in_acc_finalize(acc, C, count, eps, s_tab[0], s_tab[1], s_sum);
  if ((int)threadIdx.x < C) { scale[threadIdx.x] = s_tab[0][threadIdx.x]; shift[threadIdx.x] = s_tab[1][threadIdx.x]; }
}
__global__ void __launch_bounds__(256)
    affine_relu_f16_kernel(const _Float16 *__restrict__ x, const float *__restrict__ s,
                           const float *__restrict__ h, _Float16 *__restrict__ out, long n8, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)((i * 8) % C);
  const f16x8 v = reinterpret_cast<const f16x8 *>(x)[i];
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = (_Float16)fmaxf((float)v[e] * s[c + e] + h[c + e], 0.f);
  reinterpret_cast<f16x8 *>(out)[i] = o;
}

// out = relu(x*s + h), NHWC, C multiple of 4
__global__ void __launch_bounds__(256)
    affine_relu_kernel(const float *__restrict__ x, const float *__restrict__ s,
                       const float *__restrict__ h, float *__restrict__ out, long n4, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  float4 v = reinterpret_cast<const float4 *>(x)[i];
  const float4 a = *reinterpret_cast<const float4 *>(s + c), b = *reinterpret_cast<const float4 *>(h + c);
  v.x = fmaxf(v.x * a.x + b.x, 0.f); v.y = fmaxf(v.y * a.y + b.y, 0.f);
  v.z = fmaxf(v.z * a.z + b.z, 0.f); v.w = fmaxf(v.w * a.w + b.w, 0.f);
  reinterpret_cast<float4 *>(out)[i] = v;
}

// ----------------------------------------------- LSTM + super-state on MFMA
// The two per-pixel LSTM cells (events 5->15, image 3->15, state carried) and the shared
// super-state 1x1 convolution of the SingleScale encoder (ramp/extractor.py:233-259; reference:
// cuDNN LSTM over 307,200 length-1 sequences + two conv launches + host syncs on torch.any), fused
// into one kernel, with the three matrix-vector products per pixel batched over 16 pixels on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate):
//   gates[64 x 16 px] = Wg[64 x K] * [h ; x][K x 16 px]      (per modality, K = 24 / 20)
//   s'   [16 x 16 px] = Wss[16 x 32] * [s ; h][32 x 16 px]   (per present modality)
// The freedom in ordering rows and K columns is used so that nothing is ever shuffled:
//   * gate rows are permuted so that output tile t, lane (q, j) holds (i, f, g, o) of unit 4t+q
//     for pixel j in its four accumulator registers -> the cell update is lane-local;
//   * the recurrent state is stored tile-major, [HW/16][16 px][4 q][4 t] (unit 4 t + q; rounds 1-2: [HW/16][16 units][16 px]): for a fixed t the wave
//     reads / writes one contiguous 256-byte run, and the same registers are the B operand of
//     K-step t of the next gate product and of K-step 4+t of the super-state product;
//   * the super-state K order is channel 4q+step, i.e. lane (q, j) feeds component `step` of the
//     float4 it loaded (channels 4q..4q+3 of pixel j) -- and that is also the accumulator layout
//     the product leaves, so the second modality's product takes the first one's output as is.
// Weights arrive pre-arranged as per-lane A fragments (rampvo_amd/conv_hip.py::pack_lstm_mfma).
#define LM_EV 0                         // 4 tiles x 6 K-steps
#define LM_IM (LM_EV + 24)              // 4 tiles x 5 K-steps
#define LM_SS (LM_IM + 20)              // 8 K-steps
#define LM_BEV (LM_SS + 8)              // bias as accumulator init: 4 tiles x 4 regs
#define LM_BIM (LM_BEV + 16)
#define LM_BSS (LM_BIM + 16)            // 4 regs
#define LM_TOTAL (LM_BSS + 4)           // x 64 lanes floats

// v_exp_f32 / v_rcp_f32 based (about 1 ulp each): the IEEE expf + division + tanhf sequences were most
// of this kernel's VALU time; the LSTM gates are insensitive at that level (test tolerance 2e-5)
__device__ __forceinline__ float lm_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float lm_tanh(float x) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * x)) - 1.0f; }
// h = sigmoid(o) tanh(sigmoid(i) tanh(g)) with 4 exponentials and 2 reciprocals (instead of 4 + 4; both run at a quarter
// of the VALU rate): sigmoid(a) tanh(b) = sgn(b) (1 - t) / ((1 + e^-a) (1 + t)), t = e^(-2 |b|) <= 1 -- no overflow in the
// numerator; e^-a = inf gives the right limit 0
__device__ __forceinline__ float ms_sig_tanh(float a, float b) {
  const float t = __expf(-2.0f * fabsf(b)), ea = __expf(-a);
  const float v = (1.0f - t) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + t));
  return b < 0.f ? -v : v;
}
__device__ __forceinline__ float ms_cell(float gi, float gg, float go) { return ms_sig_tanh(go, ms_sig_tanh(gi, gg)); }
__device__ __forceinline__ float lm_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// LDSW: the 104 weight fragments per lane sit in LDS (26 KB per workgroup) instead of registers: <= 64 VGPRs, so a wave of
// this kernel fits on a SIMD next to the two 222-VGPR waves of the update operator's gru launch -- behind the gate the two
// launches otherwise TIME-SHARE the chip -- and eight of them fit when it runs alone
// The recurrent state is read once and written once per frame (120 MB): streamed with the nontemporal hint, so that it
// does not push the update operator's weights -- which the gru launch on the other stream streams from L2 -- out of the L2
#ifdef LM_CACHED
#define LM_LD(p) (*(p))
#define LM_ST(v, p) (*(p) = (v))
#else
#define LM_LD(p) __builtin_nontemporal_load(p)
#define LM_ST(v, p) __builtin_nontemporal_store((v), (p))
#endif
template <bool LDSW>
__global__ void __launch_bounds__(256, LDSW ? 8 : 1)
    lstm_superstate_mfma_kernel(const float *__restrict__ ev, const float *__restrict__ im,
                                float *__restrict__ h_ev, float *__restrict__ c_ev,
                                float *__restrict__ h_im, float *__restrict__ c_im,
                                float *__restrict__ ss, const float *__restrict__ Wf,
                                const int *__restrict__ flags, int HW, int has_state, int has_ss,
                                int tiles_per_wave, int nblk) {
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ntile = (HW + 15) / 16;
  // weights: one float per lane per fragment -- in registers for every tile of this wave, or (LDSW) in LDS
  __shared__ float s_wf[LDSW ? LM_TOTAL * 64 : 1];
  float a_ev[LDSW ? 1 : 24], a_im[LDSW ? 1 : 20], a_ss[LDSW ? 1 : 8], b_ev[LDSW ? 1 : 16], b_im[LDSW ? 1 : 16], b_ss[LDSW ? 1 : 4];
  if constexpr (LDSW) {
    for (int i = threadIdx.x; i < LM_TOTAL * 64 / 4; i += 256)
      reinterpret_cast<float4 *>(s_wf)[i] = reinterpret_cast<const float4 *>(Wf)[i];
    __syncthreads();
  } else {
#pragma unroll
    for (int f = 0; f < 24; f++) a_ev[f] = Wf[(LM_EV + f) * 64 + lane];
#pragma unroll
    for (int f = 0; f < 20; f++) a_im[f] = Wf[(LM_IM + f) * 64 + lane];
#pragma unroll
    for (int f = 0; f < 8; f++) a_ss[f] = Wf[(LM_SS + f) * 64 + lane];
#pragma unroll
    for (int f = 0; f < 16; f++) { b_ev[f] = Wf[(LM_BEV + f) * 64 + lane]; b_im[f] = Wf[(LM_BIM + f) * 64 + lane]; }
#pragma unroll
    for (int f = 0; f < 4; f++) b_ss[f] = Wf[(LM_BSS + f) * 64 + lane];
  }
  auto WA = [&](int base, int f, const float *reg) -> float { if constexpr (LDSW) return s_wf[(base + f) * 64 + lane]; else return reg[f]; };
  int f_ev, f_im;
  if (nblk > 0) {
    // per-workgroup results of any_nonzero_kernel ([2][ANY_MAXB], the first nblk of each row): every wave ORs them
    // itself (4 KB from L2) -- the flags then need no memset launch in front of the front end
    int a = 0, b = 0;
    for (int i = lane; i < nblk; i += 64) { a |= flags[i]; b |= flags[ANY_MAXB + i]; }
    f_ev = __any(a != 0) ? 1 : 0;
    f_im = __any(b != 0) ? 1 : 0;
  } else {
    f_ev = flags[0]; f_im = flags[1];
  }

  for (int it = 0; it < tiles_per_wave; it++) {
    const int tile = gw * tiles_per_wave + it;
    if (tile >= ntile) break;
    const int p = tile * 16 + j;
    const bool pv = p < HW;
    // state layout [tile][pixel j][q][4]: the lane's units 4t+q, t = 0..3, are 16 contiguous bytes -- one load / store per
    // array and lane (the [tile][unit][pixel] layout of rounds 1-2 moved 4 bytes per lane and instruction)
    const size_t sbase = (size_t)tile * 256 + j * 16 + q * 4;
    float hn[2][4];                                       // new h of both modalities, per tile t
    // every load of the tile goes out first (both modalities' h / c / inputs and the super-state): one HBM round trip
    // per tile instead of three dependent ones
    f32x4 hv4_[2], cold_[2];
    float xin_[2][2];
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 *sp = reinterpret_cast<float4 *>(ss + (size_t)p * 16) + q;
#pragma unroll
    for (int mod = 0; mod < 2; mod++) {
      const float *xin = mod == 0 ? ev : im;
      const float *hs = mod == 0 ? h_ev : h_im, *cs = mod == 0 ? c_ev : c_im;
      const int CIN = mod == 0 ? 5 : 3;
      hv4_[mod] = has_state ? LM_LD(reinterpret_cast<const f32x4 *>(hs + sbase)) : (f32x4){0.f, 0.f, 0.f, 0.f};
      cold_[mod] = has_state ? LM_LD(reinterpret_cast<const f32x4 *>(cs + sbase)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 4; s4 < 6; s4++) {
        const int ch = 4 * (s4 - 4) + q;
        xin_[mod][s4 - 4] = (ch < CIN && pv) ? LM_LD(xin + (size_t)ch * HW + p) : 0.0f;
      }
    }
    if (has_ss && pv) sv = *sp;
#pragma unroll
    for (int mod = 0; mod < 2; mod++) {
      float *hs = mod == 0 ? h_ev : h_im, *cs = mod == 0 ? c_ev : c_im;
      // B operand: K-steps 0..3 = h (unit 4s+q), then the input channels
      float bk[6];
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) bk[s4] = hv4_[mod][s4];
      bk[4] = xin_[mod][0]; bk[5] = xin_[mod][1];
      const f32x4 cold = cold_[mod];
      f32x4 cnew, hnew;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = mod == 0 ? WA(LM_BEV, t * 4 + r, b_ev) : WA(LM_BIM, t * 4 + r, b_im);
        if (mod == 0) {
#pragma unroll
          for (int s4 = 0; s4 < 6; s4++)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(WA(LM_EV, t * 6 + s4, a_ev), bk[s4], acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int s4 = 0; s4 < 5; s4++)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(WA(LM_IM, t * 5 + s4, a_im), bk[s4], acc, 0, 0, 0);
        }
        // acc = (i, f, g, o) pre-activations of unit 4t+q, pixel j (torch gate order i, f, g, o)
#ifdef LM_IEEE_CELL                                       // (A/B builds: the round-3 cell, 5 exponentials + 5 IEEE reciprocals)
        const float ig = lm_sigmoid(acc[0]), fg = lm_sigmoid(acc[1]), gg = lm_tanh(acc[2]), og = lm_sigmoid(acc[3]);
        const float cn = has_state ? fg * cold[t] + ig * gg : ig * gg;
        const float hv = og * lm_tanh(cn);
#else
        // sigmoid(i) tanh(g) and sigmoid(o) tanh(c) with one 1-ulp reciprocal each (ms_sig_tanh): 5 exponentials + 3 reciprocals
        const float igg = ms_sig_tanh(acc[0], acc[2]);
        const float cn = has_state ? __builtin_fmaf(lm_sigmoid_fast(acc[1]), cold[t], igg) : igg;
        const float hv = ms_sig_tanh(acc[3], cn);
#endif
        const bool unit_ok = 4 * t + q < 15;
        hn[mod][t] = unit_ok ? hv : 0.0f;
        cnew[t] = unit_ok ? cn : 0.0f;
        hnew[t] = hn[mod][t];
      }
      LM_ST(cnew, reinterpret_cast<f32x4 *>(cs + sbase));
      LM_ST(hnew, reinterpret_cast<f32x4 *>(hs + sbase));
    }
    // super-state: channels 4q..4q+3 of pixel j (loaded above)
    f32x4 sreg = (f32x4){sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int mod = 0; mod < 2; mod++) {
      if (!(mod == 0 ? f_ev : f_im)) continue;       // uniform
      f32x4 acc = (f32x4){WA(LM_BSS, 0, b_ss), WA(LM_BSS, 1, b_ss), WA(LM_BSS, 2, b_ss), WA(LM_BSS, 3, b_ss)};
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(WA(LM_SS, s4, a_ss), sreg[s4], acc, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; t++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(WA(LM_SS, 4 + t, a_ss), hn[mod][t], acc, 0, 0, 0);
      sreg = acc;
    }
    if (pv) *sp = make_float4(sreg[0], sreg[1], sreg[2], q == 3 ? 0.0f : sreg[3]);
  }
}

// ------------------------------------------------- MultiScale LSTM / super-state
// One scale of MultiScaleMergerDoubleNet.forward for one time step (ramp/extractor.py:540-566):
//   LSTMEncoder x2 (:376-385): conv_1 (k = S+1, stride S, pad 1; 1x1 for S = 1) on the 5 event /
//   3 image channels, then ONE per-pixel LSTM step from a zero state (no state carry: c = i*g,
//   h = o*tanh(c); the forget gate never acts), hidden size D = 16 S;
//   SuperStateEncoder x2 (:432-463): s <- Conv1x1(2D -> D)([s ; h_ev]), and, when the frame is
//   present (mask), s <- Conv1x1_im([s ; h_im]).
// Thread (group g, unit u) produces unit u for PB pixels; the mix weights are transposed
// ([2D][D]) so a wave reads one 256-byte row per input channel and re-uses it for its PB pixels.
struct MsLstmParams {
  const float *ev, *im;                 // [5][H][W], [3][H][W]
  const float *wce, *bce, *wci, *bci;   // conv_1 weight [C][C][K][K], bias [C]
  const float *wle, *ble, *wli, *bli;   // LSTM W_ih [4D][C], b_ih + b_hh [4D]
  const float *wme, *bme, *wmi, *bmi;   // mixes, transposed [2D][D], bias [D]
  float *state;                         // [Hs*Ws][D] super-state, in/out
  int H, W, Hs, Ws, has_state, use_im;
};

template <int D, int S>
__global__ void __launch_bounds__(256) ms_lstm_superstate_kernel(const MsLstmParams p) {
  constexpr int K = S > 1 ? S + 1 : 1, PAD = S > 1 ? 1 : 0;
  constexpr int G = 256 / D, PB = 4, NP = G * PB;
  __shared__ float y_e[NP][5], y_i[NP][3];
  __shared__ float vin[NP][2 * D];
  __shared__ float h_im[NP][D];
  const int u = threadIdx.x % D, g = threadIdx.x / D;
  const int pix0 = blockIdx.x * NP, HWs = p.Hs * p.Ws;

  // conv_1: NP pixels x (5 + 3) output channels
  for (int v = threadIdx.x; v < NP * 8; v += 256) {
    const int lp = v >> 3, c = v & 7, pix = pix0 + lp;
    const bool isev = c < 5;
    const int co = isev ? c : c - 5, C = isev ? 5 : 3;
    float acc = 0.f;
    if (pix < HWs) {
      const int oy = pix / p.Ws, ox = pix - oy * p.Ws;
      const float *x = isev ? p.ev : p.im;
      const float *w = (isev ? p.wce : p.wci) + (size_t)co * C * K * K;
      acc = (isev ? p.bce : p.bci)[co];
      for (int ci = 0; ci < C; ci++)
        for (int ky = 0; ky < K; ky++) {
          const int iy = oy * S - PAD + ky;
          if (iy < 0 || iy >= p.H) continue;
          for (int kx = 0; kx < K; kx++) {
            const int ix = ox * S - PAD + kx;
            if (ix < 0 || ix >= p.W) continue;
            acc = __builtin_fmaf(w[(ci * K + ky) * K + kx], x[((size_t)ci * p.H + iy) * p.W + ix], acc);
          }
        }
    }
    if (isev) y_e[lp][co] = acc; else y_i[lp][co] = acc;
  }
  __syncthreads();

  // LSTM step from the zero state, both modalities; stage [s ; h_ev]
#pragma unroll
  for (int b = 0; b < PB; b++) {
    const int lp = g * PB + b, pix = pix0 + lp;
    float gi = p.ble[u], gg = p.ble[2 * D + u], go = p.ble[3 * D + u];
#pragma unroll
    for (int c = 0; c < 5; c++) {
      const float y = y_e[lp][c];
      gi = __builtin_fmaf(p.wle[u * 5 + c], y, gi);
      gg = __builtin_fmaf(p.wle[(2 * D + u) * 5 + c], y, gg);
      go = __builtin_fmaf(p.wle[(3 * D + u) * 5 + c], y, go);
    }
    vin[lp][D + u] = sigmoidf_(go) * tanhf(sigmoidf_(gi) * tanhf(gg));
    gi = p.bli[u]; gg = p.bli[2 * D + u]; go = p.bli[3 * D + u];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float y = y_i[lp][c];
      gi = __builtin_fmaf(p.wli[u * 3 + c], y, gi);
      gg = __builtin_fmaf(p.wli[(2 * D + u) * 3 + c], y, gg);
      go = __builtin_fmaf(p.wli[(3 * D + u) * 3 + c], y, go);
    }
    h_im[lp][u] = sigmoidf_(go) * tanhf(sigmoidf_(gi) * tanhf(gg));
    vin[lp][u] = (p.has_state && pix < HWs) ? p.state[(size_t)pix * D + u] : 0.f;
  }
  __syncthreads();

  float acc[PB];
#pragma unroll
  for (int b = 0; b < PB; b++) acc[b] = p.bme[u];
  for (int k = 0; k < 2 * D; k++) {
    const float w = p.wme[k * D + u];
#pragma unroll
    for (int b = 0; b < PB; b++) acc[b] = __builtin_fmaf(w, vin[g * PB + b][k], acc[b]);
  }
  if (p.use_im) {
    __syncthreads();
#pragma unroll
    for (int b = 0; b < PB; b++) {
      vin[g * PB + b][u] = acc[b];
      vin[g * PB + b][D + u] = h_im[g * PB + b][u];
      acc[b] = p.bmi[u];
    }
    __syncthreads();
    for (int k = 0; k < 2 * D; k++) {
      const float w = p.wmi[k * D + u];
#pragma unroll
      for (int b = 0; b < PB; b++) acc[b] = __builtin_fmaf(w, vin[g * PB + b][k], acc[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < PB; b++) {
    const int pix = pix0 + g * PB + b;
    if (pix < HWs) p.state[(size_t)pix * D + u] = acc[b];
  }
}


// ------------------------------------------------- MultiScale LSTM / super-state on MFMA
// The same step (conv_1 -> zero-state LSTM -> two super-state mixes, per scale) with every matrix-vector product batched
// over 16 pixels on v_mfma_f32_16x16x4_f32 (exact fp32 products), the recipe of lstm_superstate_mfma_kernel generalised to
// D = 16, 32, 64 hidden units:
//   * a wave owns 16-pixel tiles of the scale's grid; lane (q, j) computes conv_1's output channel q of pixel j (events
//     and image; the events' fifth channel comes from the q = 3 lanes over one shuffle) -- which is the B operand of the
//     gate products' K steps as it stands;
//   * gates: one 16-row tile per (16 units, gate) -- i, g, o; the forget gate never acts on a zero state -- so lane (q, j)
//     holds (i, g, o) of units 16 t + 4 q + r, r = 0..3, in matching accumulator registers: the cell is lane-local and
//     h[t][r] is, unchanged, the B operand of K step (t, r) of the mix (its K order is channel 16 t + 4 q + r);
//   * the super-state is read as 16-byte pieces (channels 16 t + 4 q .. + 3 of pixel j) -- the same layout again -- and
//     the first mix's accumulators feed the second mix as they are; stores are 16-byte pieces, plus an fp16 copy of the
//     new state for the conv towers (scales 2, 4: what torch.cat((x, x_down2.half())) used to produce);
//   * the A fragments (one float per lane and MFMA) sit in LDS, packed per lane by conv_hip.pack_ms_scale_mfma.
// fp32 VALU version above: 53 / 61 / 50 us per scale at 640x480 (transcendental-bound at scale 1, LDS-bound mixes at 4).
struct MsMfmaParams {
  const float *ev, *im;                 // [5][H][W], [3][H][W]
  const float *wfrag;                   // per-lane A fragments [nfrag][64] (see MS_* below)
  const float *wsmall;                  // conv_1 ev W [5][5][K][K], b [5], conv_1 im W [3][3][K][K], b [3], gate biases ev
                                        // [3][D] (i, g, o), im [3][D], mix biases ev [D], im [D]
  float *state;                         // [Hs*Ws][D] fp32 super-state, in/out
  _Float16 *state16;                    // optional [Hs*Ws][D] fp16 copy of the new state
  int H, W, Hs, Ws, has_state, use_im, tiles_per_wave;
};
// (1 ulp reciprocal: the IEEE division sequence of __frcp_rn was a third of the cell's VALU work)
__device__ __forceinline__ float ms_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ms_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
template <int D, int S, int NWV>
__global__ void __launch_bounds__(64 * NWV) ms_lstm_superstate_mfma_kernel(const MsMfmaParams p) {
  constexpr int NG = D / 16;                           // 16-unit groups
  constexpr int K = S > 1 ? S + 1 : 1, PAD = S > 1 ? 1 : 0, KK = K * K;
  constexpr int F_GE = 0, F_GI = F_GE + NG * 3 * 2, F_ME = F_GI + NG * 3, F_MI = F_ME + 8 * NG * NG, F_N = F_MI + 8 * NG * NG;
  constexpr int O_WCE = 0, O_BCE = O_WCE + 25 * KK, O_WCI = O_BCE + 5, O_BCI = O_WCI + 9 * KK, O_BGE = O_BCI + 3,
                O_BGI = O_BGE + 3 * D, O_BME = O_BGI + 3 * D, O_BMI = O_BME + D, O_N = O_BMI + D;
  // input window of a 16-pixel tile (S > 1): K rows x (15 S + K) columns x 8 channels, staged per wave with all loads in
  // flight at once (walking the taps from memory was a chain of 25 dependent round trips at scale 4)
  constexpr int WC = 15 * S + K, WIN = S > 1 ? 8 * K * WC : 0;
  extern __shared__ __attribute__((aligned(16))) float ms_smem[];
  float *s_wf = ms_smem, *s_sm = ms_smem + F_N * 64;
  float *s_win = s_sm + ((O_N + 3) & ~3) + (threadIdx.x >> 6) * WIN;
  for (int i = threadIdx.x; i < F_N * 16; i += 64 * NWV)
    reinterpret_cast<float4 *>(s_wf)[i] = reinterpret_cast<const float4 *>(p.wfrag)[i];
  for (int i = threadIdx.x; i < O_N; i += 64 * NWV) s_sm[i] = p.wsmall[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int gw = blockIdx.x * NWV + (threadIdx.x >> 6);
  const int HWs = p.Hs * p.Ws, ntile = (HWs + 15) / 16;
  for (int it = 0; it < p.tiles_per_wave; it++) {
    const int tile = gw * p.tiles_per_wave + it;
    if (tile >= ntile) break;                          // (wave-uniform)
    const int pix = tile * 16 + j;
    const bool pv = pix < HWs;
    const int pc = pv ? pix : HWs - 1;
    const int oy = pc / p.Ws, ox = pc - oy * p.Ws;
    // the old super-state (all loads of the tile go out first)
    f32x4 sreg[NG];
    const float *sp = p.state + (size_t)pc * D + 4 * q;
#pragma unroll
    for (int t = 0; t < NG; t++)
      sreg[t] = p.has_state ? *reinterpret_cast<const f32x4 *>(sp + 16 * t) : (f32x4){0.f, 0.f, 0.f, 0.f};
    // conv_1: lane (q, j) -> events channel q, image channel q (q < 3) or events channel 4 (q == 3)
    float y0 = s_sm[O_BCE + q], y1 = q < 3 ? s_sm[O_BCI + q] : s_sm[O_BCE + 4];
    {
      const float *w0 = s_sm + O_WCE + q * 5 * KK;
      const float *w1 = q < 3 ? s_sm + O_WCI + q * 3 * KK : s_sm + O_WCE + 4 * 5 * KK;
      if constexpr (S == 1) {
        float xe[5], xi[3];
#pragma unroll
        for (int ci = 0; ci < 5; ci++) xe[ci] = p.ev[(size_t)ci * p.H * p.W + pc];
#pragma unroll
        for (int ci = 0; ci < 3; ci++) xi[ci] = p.im[(size_t)ci * p.H * p.W + pc];
#pragma unroll
        for (int ci = 0; ci < 5; ci++) y0 = __builtin_fmaf(w0[ci], xe[ci], y0);
        if (q < 3) {
#pragma unroll
          for (int ci = 0; ci < 3; ci++) y1 = __builtin_fmaf(w1[ci], xi[ci], y1);
        } else {
#pragma unroll
          for (int ci = 0; ci < 5; ci++) y1 = __builtin_fmaf(w1[ci], xe[ci], y1);
        }
      } else {
        // (Ws is a multiple of 16: a tile is 16 neighbours of one row)
        const int toy = (tile * 16) / p.Ws, tox = tile * 16 - toy * p.Ws;
        const int iy0 = toy * S - PAD, ix0 = tox * S - PAD;
        const size_t HW = (size_t)p.H * p.W;
        for (int i = lane; i < WIN; i += 64) {
          const int ch = i / (K * WC), rem = i - ch * (K * WC), ky = rem / WC, cx = rem - ky * WC;
          const int iy = iy0 + ky, ix = ix0 + cx;
          const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
          const float *src = ch < 5 ? p.ev + (size_t)ch * HW : p.im + (size_t)(ch - 5) * HW;
          s_win[i] = ok ? src[(size_t)iy * p.W + ix] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ky = 0; ky < K; ky++)
#pragma unroll
          for (int kx = 0; kx < K; kx++) {
            float xe[5], xi[3];
#pragma unroll
            for (int ci = 0; ci < 5; ci++) xe[ci] = s_win[(ci * K + ky) * WC + j * S + kx];
#pragma unroll
            for (int ci = 0; ci < 3; ci++) xi[ci] = s_win[((5 + ci) * K + ky) * WC + j * S + kx];
#pragma unroll
            for (int ci = 0; ci < 5; ci++) y0 = __builtin_fmaf(w0[(ci * K + ky) * K + kx], xe[ci], y0);
            if (q < 3) {
#pragma unroll
              for (int ci = 0; ci < 3; ci++) y1 = __builtin_fmaf(w1[(ci * K + ky) * K + kx], xi[ci], y1);
            } else {
#pragma unroll
              for (int ci = 0; ci < 5; ci++) y1 = __builtin_fmaf(w1[(ci * K + ky) * K + kx], xe[ci], y1);
            }
          }
        __builtin_amdgcn_wave_barrier();
      }
    }
    const float e4 = __shfl(y1, 48 + j, 64);           // events channel 4 of pixel j (computed by lane (3, j))
    const float be0 = y0, be1 = q == 0 ? e4 : 0.f, bi0 = q < 3 ? y1 : 0.f;
    // LSTM step from the zero state: h = sigmoid(o) tanh(sigmoid(i) tanh(g)), both modalities
    float hn[2][NG][4];
#pragma unroll
    for (int mod = 0; mod < 2; mod++) {
#pragma unroll
      for (int t = 0; t < NG; t++) {
        f32x4 g3[3];
#pragma unroll
        for (int gi = 0; gi < 3; gi++) {
          g3[gi] = *reinterpret_cast<const f32x4 *>(s_sm + (mod == 0 ? O_BGE : O_BGI) + gi * D + 16 * t + 4 * q);
          if (mod == 0) {
            g3[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_wf[(F_GE + (t * 3 + gi) * 2 + 0) * 64 + lane], be0, g3[gi], 0, 0, 0);
            g3[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_wf[(F_GE + (t * 3 + gi) * 2 + 1) * 64 + lane], be1, g3[gi], 0, 0, 0);
          } else {
            g3[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_wf[(F_GI + t * 3 + gi) * 64 + lane], bi0, g3[gi], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) hn[mod][t][r] = ms_cell(g3[0][r], g3[1][r], g3[2][r]);
      }
    }
    // s <- mix_ev([s ; h_ev]);  if the frame is present: s <- mix_im([s ; h_im])
#pragma unroll
    for (int mod = 0; mod < 2; mod++) {
      if (mod == 1 && !p.use_im) break;                // (uniform)
      f32x4 acc[NG];
#pragma unroll
      for (int n = 0; n < NG; n++) acc[n] = *reinterpret_cast<const f32x4 *>(s_sm + (mod == 0 ? O_BME : O_BMI) + 16 * n + 4 * q);
      const float *wf = s_wf + (size_t)(mod == 0 ? F_ME : F_MI) * 64 + lane;
#pragma unroll
      for (int half = 0; half < 2; half++)
#pragma unroll
        for (int t = 0; t < NG; t++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float b = half == 0 ? sreg[t][r] : hn[mod][t][r];
            const int ks = (half * NG + t) * 4 + r;
#pragma unroll
            for (int n = 0; n < NG; n++)
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[(size_t)(n * 8 * NG + ks) * 64], b, acc[n], 0, 0, 0);
          }
#pragma unroll
      for (int n = 0; n < NG; n++) sreg[n] = acc[n];
    }
    if (pv) {
      float *so = p.state + (size_t)pix * D + 4 * q;
#pragma unroll
      for (int t = 0; t < NG; t++) *reinterpret_cast<f32x4 *>(so + 16 * t) = sreg[t];
      if (p.state16) {
        _Float16 *ho = p.state16 + (size_t)pix * D + 4 * q;
#pragma unroll
        for (int t = 0; t < NG; t++)
          *reinterpret_cast<f16x4 *>(ho + 16 * t) = (f16x4){(_Float16)sreg[t][0], (_Float16)sreg[t][1], (_Float16)sreg[t][2], (_Float16)sreg[t][3]};
      }
    }
  }
}

// The same step with ONE 16-pixel tile per workgroup and one wave per 16-unit group (D = 32: 2 waves, 64: 4): wave w
// computes the gates of units 16 w .. 16 w + 15 and output tile w of both mixes; h and the first mix's output cross the
// waves through LDS (two barriers).  The scale-2 / scale-4 grids have only 4800 / 1200 tiles: with a whole tile per wave the
// launch was one wave per SIMD walking a chain of ~300 dependent MFMAs behind an 80 KB weight staging (56 us at scale 4);
// here a wave's chain is 9 + 2 x 8 NG products, its A fragments (9 + 16 NG floats per lane) are prefetched into registers
// straight from L2 while the input window is staged, and the launch is NG x as many waves.
template <int D, int S>
__global__ void __launch_bounds__(4 * D) ms_lstm_superstate_split_kernel(const MsMfmaParams p) {
  constexpr int NG = D / 16;
  constexpr int K = S > 1 ? S + 1 : 1, PAD = S > 1 ? 1 : 0, KK = K * K;
  constexpr int F_GE = 0, F_GI = F_GE + NG * 3 * 2, F_ME = F_GI + NG * 3, F_MI = F_ME + 8 * NG * NG;
  constexpr int O_WCE = 0, O_BCE = O_WCE + 25 * KK, O_WCI = O_BCE + 5, O_BCI = O_WCI + 9 * KK, O_BGE = O_BCI + 3,
                O_BGI = O_BGE + 3 * D, O_BME = O_BGI + 3 * D, O_BMI = O_BME + D, O_N = O_BMI + D;
  constexpr int WC = 15 * S + K, WIN = 8 * K * WC;
  __shared__ float s_sm[(O_N + 3) & ~3];
  __shared__ float s_win[WIN];
  __shared__ __attribute__((aligned(16))) float s_h[2][NG][64][4];
  __shared__ __attribute__((aligned(16))) float s_s[NG][64][4];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int tile = blockIdx.x, HWs = p.Hs * p.Ws;
  const int pix = tile * 16 + j;
  const bool pv = pix < HWs;
  const int pc = pv ? pix : HWs - 1;
  // this wave's A fragments, from L2, ahead of everything else
  float a_ge[6], a_gi[3], a_me[8 * NG], a_mi[8 * NG];
#pragma unroll
  for (int f = 0; f < 6; f++) a_ge[f] = p.wfrag[(size_t)(F_GE + w * 6 + f) * 64 + lane];
#pragma unroll
  for (int f = 0; f < 3; f++) a_gi[f] = p.wfrag[(size_t)(F_GI + w * 3 + f) * 64 + lane];
#pragma unroll
  for (int f = 0; f < 8 * NG; f++) {
    a_me[f] = p.wfrag[(size_t)(F_ME + w * 8 * NG + f) * 64 + lane];
    a_mi[f] = p.wfrag[(size_t)(F_MI + w * 8 * NG + f) * 64 + lane];
  }
  // the old super-state, every 16-channel group of it (B operand of the first mix)
  f32x4 s_all[NG];
  {
    const float *sp = p.state + (size_t)pc * D + 4 * q;
#pragma unroll
    for (int t = 0; t < NG; t++)
      s_all[t] = p.has_state ? *reinterpret_cast<const f32x4 *>(sp + 16 * t) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // the tile's input window and the small weights -> LDS (a tile = 16 neighbours of one row: Ws % 16 == 0)
  {
    const int toy = (tile * 16) / p.Ws, tox = tile * 16 - toy * p.Ws;
    const int iy0 = toy * S - PAD, ix0 = tox * S - PAD;
    const size_t HW = (size_t)p.H * p.W;
#ifndef MS_SKIP_WIN
    // one (channel, tap row) of the window per wave and round: the row's address is wave-uniform, a lane adds its column
    for (int row = w; row < 8 * K; row += NG) {
      const int ch = row / K, ky = row - ch * K, iy = iy0 + ky;
      const bool rok = iy >= 0 && iy < p.H;
      const float *src = (ch < 5 ? p.ev + (size_t)ch * HW : p.im + (size_t)(ch - 5) * HW) + (size_t)(rok ? iy : 0) * p.W;
      for (int cx = lane; cx < WC; cx += 64) {
        const int ix = ix0 + cx;
        s_win[row * WC + cx] = (rok && ix >= 0 && ix < p.W) ? src[ix] : 0.f;
      }
    }
#endif
    for (int i = tid; i < O_N; i += 4 * D) s_sm[i] = p.wsmall[i];
  }
  __syncthreads();
  // conv_1, once per tile: thread (part, c, j) sums its share of the taps of output channel c (0..4 events, 5..7 image) of
  // pixel j -- ~50 independent LDS reads per thread -- and the partial sums meet in LDS.  (Every wave walking all 25 taps of
  // its lanes' two channels was 18 dependent LDS round trips per tap: 35 of the kernel's 58 us at scale 4.)
  constexpr int NPART = (4 * D) / 128;
  __shared__ float s_y[NPART][8][16];
#ifndef MS_SKIP_CONV
  {
    const int cj = tid & 15, cc = (tid >> 4) & 7, part = tid >> 7;     // (part is wave-uniform)
    const bool isev = cc < 5;
    const float *wv = s_sm + (isev ? O_WCE + cc * 5 * KK : O_WCI + (cc - 5) * 3 * KK);
    const float *xw = s_win + (isev ? 0 : 5 * K * WC) + cj * S;
    // tap rows [0, KY0) to part 0, the rest to part 1 (one part: all of them); every offset below is a compile-time constant
    constexpr int KY0 = NPART == 1 ? K : (K + 1) / 2;
    float acc = 0.f;
#define MS_TAPS(CI, KA, KB)                                                                     \
    _Pragma("unroll") for (int ci = 0; ci < CI; ci++)                                             \
    _Pragma("unroll") for (int ky = KA; ky < KB; ky++)                                            \
    _Pragma("unroll") for (int kx = 0; kx < K; kx++)                                              \
      acc = __builtin_fmaf(wv[(ci * K + ky) * K + kx], xw[(ci * K + ky) * WC + kx], acc);
    if (part == 0) {
      if (isev) { MS_TAPS(5, 0, KY0) } else { MS_TAPS(3, 0, KY0) }
    } else {
      if (isev) { MS_TAPS(5, KY0, K) } else { MS_TAPS(3, KY0, K) }
    }
#undef MS_TAPS
    s_y[part][cc][cj] = acc;
  }
  __syncthreads();
#endif
  float y0 = s_sm[O_BCE + q], y1 = q < 3 ? s_sm[O_BCI + q] : s_sm[O_BCE + 4];
#ifndef MS_SKIP_CONV
#pragma unroll
  for (int part = 0; part < NPART; part++) {
    y0 += s_y[part][q][j];
    y1 += s_y[part][q < 3 ? 5 + q : 4][j];
  }
#else
  y0 = s_win[lane]; y1 = s_win[64 + lane];
#endif
  const float e4 = __shfl(y1, 48 + j, 64);
  const float be0 = y0, be1 = q == 0 ? e4 : 0.f, bi0 = q < 3 ? y1 : 0.f;
  // gates of units 16 w + 4 q + r, both modalities -> h, published for the other waves
#pragma unroll
  for (int mod = 0; mod < 2; mod++) {
    f32x4 g3[3];
#pragma unroll
    for (int gi = 0; gi < 3; gi++) {
      g3[gi] = *reinterpret_cast<const f32x4 *>(s_sm + (mod == 0 ? O_BGE : O_BGI) + gi * D + 16 * w + 4 * q);
      if (mod == 0) {
        g3[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ge[gi * 2 + 0], be0, g3[gi], 0, 0, 0);
        g3[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ge[gi * 2 + 1], be1, g3[gi], 0, 0, 0);
      } else {
        g3[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_gi[gi], bi0, g3[gi], 0, 0, 0);
      }
    }
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; r++) h[r] = ms_cell(g3[0][r], g3[1][r], g3[2][r]);
    *reinterpret_cast<f32x4 *>(&s_h[mod][w][lane][0]) = h;
  }
  __syncthreads();
  f32x4 acc = *reinterpret_cast<const f32x4 *>(s_sm + O_BME + 16 * w + 4 * q);
#pragma unroll
  for (int t = 0; t < NG; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_me[t * 4 + r], s_all[t][r], acc, 0, 0, 0);
#pragma unroll
  for (int t = 0; t < NG; t++) {
    const f32x4 h = *reinterpret_cast<const f32x4 *>(&s_h[0][t][lane][0]);
#pragma unroll
    for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_me[(NG + t) * 4 + r], h[r], acc, 0, 0, 0);
  }
  if (p.use_im) {                                    // (uniform)
    *reinterpret_cast<f32x4 *>(&s_s[w][lane][0]) = acc;
    __syncthreads();
    f32x4 acc2 = *reinterpret_cast<const f32x4 *>(s_sm + O_BMI + 16 * w + 4 * q);
#pragma unroll
    for (int t = 0; t < NG; t++) {
      const f32x4 sv = *reinterpret_cast<const f32x4 *>(&s_s[t][lane][0]);
#pragma unroll
      for (int r = 0; r < 4; r++) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_mi[t * 4 + r], sv[r], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NG; t++) {
      const f32x4 h = *reinterpret_cast<const f32x4 *>(&s_h[1][t][lane][0]);
#pragma unroll
      for (int r = 0; r < 4; r++) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_mi[(NG + t) * 4 + r], h[r], acc2, 0, 0, 0);
    }
    acc = acc2;
  }
  if (pv) {
    *reinterpret_cast<f32x4 *>(p.state + (size_t)pix * D + 16 * w + 4 * q) = acc;
    if (p.state16)
      *reinterpret_cast<f16x4 *>(p.state16 + (size_t)pix * D + 16 * w + 4 * q) =
          (f16x4){(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
  }
}

template <int D, int S, int NWV>
static int ms_mfma_launch(const MsMfmaParams &p0, hipStream_t st) {
  constexpr int NG = D / 16, K = S > 1 ? S + 1 : 1, KK = K * K;
  constexpr int F_N = NG * 3 * 2 + NG * 3 + 16 * NG * NG;
  constexpr int O_N = 25 * KK + 5 + 9 * KK + 3 + 6 * D + 2 * D;
  constexpr int WIN = S > 1 ? 8 * K * (15 * S + K) : 0;
  const size_t lds = (size_t)(F_N * 64 + ((O_N + 3) & ~3) + NWV * WIN) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)ms_lstm_superstate_mfma_kernel<D, S, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return RAMP_ELAUNCH;
    attr_set = true;
  }
  MsMfmaParams p = p0;
  const int ntile = ramp_cdiv(p.Hs * p.Ws, 16);
  // one tile per wave while that keeps the launch within ~8 waves per SIMD; more tiles per wave beyond (a workgroup's
  // weight staging is then amortised over them)
  int tpw = ramp_cdiv(ntile, 8 * 1024);
  if (tpw < 1) tpw = 1;
  p.tiles_per_wave = tpw;
  hipLaunchKernelGGL((ms_lstm_superstate_mfma_kernel<D, S, NWV>), dim3(ramp_cdiv(ntile, NWV * tpw)), dim3(64 * NWV), lds, st, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

extern "C" {

int ramp_affine_relu(const float *x, const float *s, const float *h, float *out, long n, int C,
                     void *stream) {
  if (!x || !s || !h || !out || n <= 0 || C % 4 || n % 4) return RAMP_EINVAL;
  const long n4 = n / 4;
  hipLaunchKernelGGL(affine_relu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, s, h, out, n4, C);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_affine_relu_f16(const void *x, const float *s, const float *h, void *out, long n, int C,
                         void *stream) {
  if (!x || !s || !h || !out || n <= 0 || C % 8 || n % 8) return RAMP_EINVAL;
  const long n8 = n / 8;
  hipLaunchKernelGGL(affine_relu_f16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const _Float16 *)x, s, h, (_Float16 *)out, n8, C);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_norm_add_relu_f16(const void *y, const float *sy, const float *hy, const void *skip,
                           const float *ss, const float *hs, void *out, long n, int C, int skip_relu,
                           void *stream) {
  if (!y || !sy || !hy || !skip || !out || n <= 0 || C % 8 || n % 8 || (skip_relu && !ss)) return RAMP_EINVAL;
  const long n8 = n / 8;
  hipLaunchKernelGGL(norm_add_relu_f16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const _Float16 *)y, sy, hy, (const _Float16 *)skip, ss, hs,
                     (_Float16 *)out, n8, C, skip_relu);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_norm_add_relu_f16_acc(const void *y, const void *acc_y, float count_y, float eps_y, const void *skip,
                               const void *acc_skip, float count_skip, float eps_skip, void *out, long n, int C,
                               int skip_relu, void *stream) {
  if (!y || !acc_y || !skip || !out || n <= 0 || C % 8 || C > 128 || n % 8 || count_y <= 0.f || (skip_relu && !acc_skip) ||
      (acc_skip && count_skip <= 0.f))
    return RAMP_EINVAL;
  const long n8 = n / 8;
  hipLaunchKernelGGL(norm_add_relu_f16_acc_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16 *)y, (const unsigned long long *)acc_y, count_y, eps_y, (const _Float16 *)skip,
                     (const unsigned long long *)acc_skip, count_skip, eps_skip, (_Float16 *)out, n8, C, skip_relu);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_in_acc_finalize(const void *acc, int C, float count, float eps, float *scale, float *shift, void *stream) {
  if (!acc || !scale || !shift || C <= 0 || C > 128 || count <= 0.f) return RAMP_EINVAL;
  hipLaunchKernelGGL(in_acc_to_scale_shift_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream,
                     (const unsigned long long *)acc, C, count, eps, scale, shift);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_any_nonzero(const float *a, long na, const float *b, long nb, int32_t *flags, void *stream) {
  if (!flags) return RAMP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), st) != hipSuccess) return RAMP_ELAUNCH;
  const long n = na > nb ? na : nb;
  if (n <= 0) return RAMP_OK;
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(any_nonzero_kernel, dim3(blocks), dim3(256), 0, st, a, na, b, nb, flags, 0, (uint4 *)nullptr, 0l);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_any_nonzero_blocks(const float *a, long na, const float *b, long nb, int32_t *blockflags, void *stream) {
  return ramp_any_nonzero_blocks_clear(a, na, b, nb, blockflags, nullptr, 0, stream);
}

int ramp_any_nonzero_blocks_clear(const float *a, long na, const float *b, long nb, int32_t *blockflags, void *clear,
                                  long clear_bytes, void *stream) {
  if (!blockflags || clear_bytes < 0 || (clear_bytes & 15) || (clear_bytes && (!clear || ((size_t)clear & 15)))) return RAMP_EINVAL;
  const long n = na > nb ? na : nb;
  if (n <= 0) return RAMP_EINVAL;
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > ANY_MAXB) blocks = ANY_MAXB;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(any_nonzero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, blockflags, 1,
                     (uint4 *)clear, clear_bytes / 16);
  if (hipGetLastError() != hipSuccess) return RAMP_ELAUNCH;
  return blocks;
}

int ramp_lstm_superstate_tiled(const float *ev, const float *im, float *h_ev, float *c_ev, float *h_im,
                               float *c_im, float *ss, const float *wfrag, const int32_t *flags, int HW,
                               int has_state, int has_ss, void *stream) {
  return ramp_lstm_superstate_blocks(ev, im, h_ev, c_ev, h_im, c_im, ss, wfrag, flags, 0, HW, has_state, has_ss, stream);
}

int ramp_lstm_superstate_blocks(const float *ev, const float *im, float *h_ev, float *c_ev, float *h_im,
                                float *c_im, float *ss, const float *wfrag, const int32_t *flags, int nblk, int HW,
                                int has_state, int has_ss, void *stream) {
  if (HW <= 0 || !ev || !im || !h_ev || !c_ev || !h_im || !c_im || !ss || !wfrag || !flags || nblk < 0 || nblk > ANY_MAXB)
    return RAMP_EINVAL;
  const int ntile = ramp_cdiv(HW, 16);
  const int tpw = ntile >= 8192 ? 4 : 1;           // tiles per wave: amortise the 104 weight fragments
#ifndef LSTM_LDSW
#define LSTM_LDSW 1                                // (build-time A/B, tools/ab_build.sh: 0 = weight fragments in registers)
#endif
  if (LSTM_LDSW)
    hipLaunchKernelGGL(lstm_superstate_mfma_kernel<true>, dim3(ramp_cdiv(ntile, 4 * tpw)), dim3(256), 0,
                       (hipStream_t)stream, ev, im, h_ev, c_ev, h_im, c_im, ss, wfrag, flags, HW, has_state,
                       has_ss, tpw, nblk);
  else
    hipLaunchKernelGGL(lstm_superstate_mfma_kernel<false>, dim3(ramp_cdiv(ntile, 4 * tpw)), dim3(256), 0,
                       (hipStream_t)stream, ev, im, h_ev, c_ev, h_im, c_im, ss, wfrag, flags, HW, has_state,
                       has_ss, tpw, nblk);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_conv2d_nhwc(const void *x, const void *wpk, const float *bias, const float *pre_scale,
                     const float *pre_shift, const void *res, void *y, float *stats, int H, int W,
                     int Cin, int Cout, int KH, int KW, int stride, int relu, float out_scale,
                     int dtype, void *stream) {
  if (!x || !wpk || !y || H <= 0 || W <= 0) return RAMP_EINVAL;
  if (Cin % 16 || Cout % 32 || KH != KW) return RAMP_EUNSUPPORTED;
  // dtype: RAMP_F32 = fp32 in/out (exact fp32 MFMA); RAMP_F16 = half in/out;
  //        RAMP_F16 | 0x10 = fp32 in, half out (first layer of the mixed-precision tower)
  const bool f16 = (dtype & 0xf) == RAMP_F16, in_f32 = f16 && (dtype & 0x10);
  const bool x3 = dtype == (RAMP_F32 | RAMP_CONV_X3);
  if (!f16 && dtype != RAMP_F32 && !x3) return RAMP_EINVAL;
  if (f16 && !in_f32 && Cin % 32) return RAMP_EUNSUPPORTED;
  if (in_f32 && Cin != 16) return RAMP_EUNSUPPORTED;
  ConvParams p;
  p.x = x; p.wpk = wpk; p.bias = bias; p.pre_scale = pre_scale; p.pre_shift = pre_shift;
  p.res = res; p.y = y; p.stats = stats; p.x2 = nullptr; p.C0 = 0;
  p.skip = nullptr; p.acc_skip = nullptr; p.skip_count = 0.f; p.skip_eps = 0.f; p.skip_relu = 0; p.mat = nullptr;
  p.acc_out = nullptr; p.acc_in = nullptr; p.in_count = 0.f; p.in_eps = 0.f;
  p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int pad = KH / 2;
  p.OH = (H + 2 * pad - KH) / stride + 1;
  p.OW = (W + 2 * pad - KW) / stride + 1;
  p.relu = relu; p.out_scale = out_scale;
  p.act_scale = 1.0f; p.descale = 1.0f;
  const int M = p.OH * p.OW;
  dim3 grid(ramp_cdiv(M, 128), Cout / 32), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (x3) {
    // fp32 in / out at fp32 accuracy on the f16 matrix cores (conv_x3_kernel; wpk = pack_conv_weight mode "x3")
    const dim3 tg(ramp_cdiv(p.OH, 8) * ramp_cdiv(p.OW, 16), 1);
#define X3_CASE(K, S, CIN, COUT)                                                                      \
  if (KH == K && stride == S && Cin == CIN && Cout == COUT) {                                         \
    constexpr int lds_ = conv_x3_lds_bytes(K, S, CIN);                                                \
    if (conv_tile_attr(conv_x3_kernel<K, S, CIN, COUT>, lds_) != RAMP_OK) return RAMP_ELAUNCH;        \
    hipLaunchKernelGGL((conv_x3_kernel<K, S, CIN, COUT>), tg, block, lds_, st, p);                    \
    RAMP_CHECK_LAUNCH();                                                                              \
    return RAMP_OK;                                                                                   \
  }
    X3_CASE(3, 1, 32, 32)
    X3_CASE(3, 1, 64, 64)
    X3_CASE(3, 1, 32, 64)
    X3_CASE(3, 2, 32, 64)
    X3_CASE(3, 1, 64, 32)
    X3_CASE(7, 2, 16, 32)
#undef X3_CASE
    return RAMP_EUNSUPPORTED;
  }
  // fp16: LDS-tiled kernel for the layer shapes of the towers (RAMP_CONV_DIRECT forces the direct one)
  if (f16 && !(dtype & RAMP_CONV_DIRECT)) {
    const dim3 tg(ramp_cdiv(p.OH, 8) * ramp_cdiv(p.OW, 16), 1);
#define TILE_CASE(K, S, INF32, CIN, NT)                                                              \
  if (KH == K && stride == S && in_f32 == INF32 && Cin == CIN && Cout % (NT * 16) == 0) {             \
    ConvMulti pm;                                                                                     \
    pm.t[0] = p; pm.t[1] = p;                                                                         \
    constexpr int lds_ = conv_tile_lds_bytes(K, S, INF32, CIN, NT, false);                            \
    if (conv_tile_attr(conv_tile_f16_kernel<K, S, INF32, CIN, NT>, lds_) != RAMP_OK) return RAMP_ELAUNCH; \
    hipLaunchKernelGGL((conv_tile_f16_kernel<K, S, INF32, CIN, NT>), dim3(tg.x, Cout / (NT * 16), 1),  \
                       block, lds_, st, pm);                                                          \
    RAMP_CHECK_LAUNCH();                                                                              \
    return RAMP_OK;                                                                                   \
  }
    TILE_CASE(7, 2, true, 16, 2)
    TILE_CASE(3, 1, false, 32, 2)
    TILE_CASE(3, 2, false, 32, 2)
    TILE_CASE(3, 2, false, 64, 2)
    TILE_CASE(3, 1, false, 64, 2)
    TILE_CASE(1, 2, false, 32, 4)
    TILE_CASE(1, 2, false, 64, 4)
    TILE_CASE(1, 1, false, 64, 4)
    TILE_CASE(1, 1, false, 128, 4)
#undef TILE_CASE
  }
#define CONV_CASE(K, S)                                                                              \
  if (KH == K && stride == S) {                                                                      \
    if (!f16) hipLaunchKernelGGL((conv_mfma_f32_kernel<K, K, S>), grid, block, 0, st, p);            \
    else if (in_f32) hipLaunchKernelGGL((conv_mfma_f16_kernel<K, K, S, true>), grid, block, 0, st, p); \
    else hipLaunchKernelGGL((conv_mfma_f16_kernel<K, K, S, false>), grid, block, 0, st, p);          \
    RAMP_CHECK_LAUNCH();                                                                             \
    return RAMP_OK;                                                                                  \
  }
  CONV_CASE(7, 2)
  CONV_CASE(3, 1)
  CONV_CASE(3, 2)
  CONV_CASE(1, 1)
  CONV_CASE(1, 2)
#undef CONV_CASE
  return RAMP_EUNSUPPORTED;
}

int ramp_conv2d_nhwc_multi(const ramp_conv_job *jobs, int njobs, int H, int W, int Cin, int KH, int stride,
                           int dtype, void *stream) {
  if (!jobs || njobs < 1 || njobs > 2 || H <= 0 || W <= 0) return RAMP_EINVAL;
  const bool f16 = (dtype & 0xf) == RAMP_F16, in_f32 = f16 && (dtype & RAMP_IN_F32);
  if (!f16 || (dtype & RAMP_CONV_DIRECT)) return RAMP_EUNSUPPORTED;
  const bool fp8 = (dtype & RAMP_CONV_FP8) != 0;
  if (fp8 && in_f32) return RAMP_EUNSUPPORTED;
  const int pad = KH / 2;
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
  if (OH <= 0 || OW <= 0) return RAMP_EINVAL;
  ConvMulti pm;
  int cmax = 0, cgcd_ok64 = 1;
  for (int t = 0; t < 2; t++) {
    const ramp_conv_job &j = jobs[t < njobs ? t : 0];
    if (!j.x || !j.wpk || !j.y || j.Cout <= 0 || j.Cout % 32) return RAMP_EINVAL;
    ConvParams &p = pm.t[t];
    p.x = j.x; p.wpk = j.wpk; p.bias = j.bias; p.pre_scale = j.pre_scale; p.pre_shift = j.pre_shift;
    p.res = j.res; p.y = j.y; p.stats = j.stats;
    p.x2 = j.x2; p.C0 = j.c0;
    p.skip = j.skip; p.acc_skip = (const unsigned long long *)j.acc_skip; p.skip_count = j.skip_count; p.skip_eps = j.skip_eps;
    p.skip_relu = j.skip_relu; p.mat = j.mat;
    if (p.skip && (in_f32 || !j.acc_in || p.x2 || Cin > 128 || (p.acc_skip && !(p.skip_count > 0.f)) || (p.skip_relu && !p.acc_skip) ||
                   (p.mat && (stride != 1 || (KH != 1 && KH != 3)))))
      return RAMP_EINVAL;
    if (!p.skip && (p.mat || p.acc_skip)) return RAMP_EINVAL;
    if (p.skip && (fp8 || j.stats)) return RAMP_EUNSUPPORTED;      // (f16 MFMA instances in accumulator mode only)
    if (p.x2 && (in_f32 || p.C0 <= 0 || p.C0 >= Cin || (p.C0 & 7) || ((Cin - p.C0) & 7) || p.pre_scale || j.acc_in)) return RAMP_EINVAL;
    p.acc_out = (unsigned long long *)j.acc_out; p.acc_in = (const unsigned long long *)j.acc_in;
    p.in_count = j.in_count; p.in_eps = j.in_eps;
    if (p.acc_in && (Cin > 128 || p.in_count <= 0.f)) return RAMP_EINVAL;
    p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = j.Cout;
    p.relu = j.relu; p.out_scale = j.out_scale;
    p.act_scale = 1.0f; p.descale = 1.0f;
    if (fp8) {
      if (!(j.act_scale > 0.0f) || !(j.w_scale > 0.0f)) return RAMP_EINVAL;
      p.act_scale = j.act_scale; p.descale = 1.0f / (j.act_scale * j.w_scale);
    }
    cmax = j.Cout > cmax ? j.Cout : cmax;
    if (j.Cout % 64) cgcd_ok64 = 0;
  }
  const dim3 block(256);
  const int tiles = ramp_cdiv(OH, 8) * ramp_cdiv(OW, 16);
  hipStream_t st = (hipStream_t)stream;
  // the first layer of the two towers: one workgroup per tile computes both (shared input, no prologue)
  if (KH == 7 && stride == 2 && in_f32 && Cin == 16 && njobs == 2 && jobs[0].x == jobs[1].x && jobs[0].Cout == 32 &&
      jobs[1].Cout == 32 && !jobs[0].pre_scale && !jobs[1].pre_scale) {
    hipLaunchKernelGGL(conv7_dual_kernel, dim3(tiles, 1, 1), block, 0, st, pm);
    RAMP_CHECK_LAUNCH();
    return RAMP_OK;
  }
#define TILE_CASE8(K, S, CIN, NT)                                                                    \
  if (fp8 && KH == K && stride == S && Cin == CIN && (NT == 2 || cgcd_ok64)) {                        \
    constexpr int lds_ = conv_tile_lds_bytes(K, S, false, CIN, NT, true);                             \
    if (conv_tile_attr(conv_tile_f16_kernel<K, S, false, CIN, NT, true>, lds_) != RAMP_OK) return RAMP_ELAUNCH; \
    hipLaunchKernelGGL((conv_tile_f16_kernel<K, S, false, CIN, NT, true>), dim3(tiles, cmax / (NT * 16), njobs), \
                       block, lds_, st, pm);                                                          \
    RAMP_CHECK_LAUNCH();                                                                              \
    return RAMP_OK;                                                                                   \
  }
  TILE_CASE8(3, 1, 32, 2)
  TILE_CASE8(3, 2, 32, 2)
  TILE_CASE8(3, 2, 64, 2)
  TILE_CASE8(3, 1, 64, 2)
  TILE_CASE8(1, 2, 32, 4)
  TILE_CASE8(1, 2, 64, 4)
  TILE_CASE8(1, 1, 64, 4)
  TILE_CASE8(1, 1, 128, 4)
#undef TILE_CASE8
  if (fp8) return RAMP_EUNSUPPORTED;
#define TILE_CASE(K, S, INF32, CIN, NT)                                                              \
  if (KH == K && stride == S && in_f32 == INF32 && Cin == CIN && (NT == 2 || cgcd_ok64)) {            \
    constexpr int lds_ = conv_tile_lds_bytes(K, S, INF32, CIN, NT, false);                            \
    if (conv_tile_attr(conv_tile_f16_kernel<K, S, INF32, CIN, NT>, lds_) != RAMP_OK) return RAMP_ELAUNCH; \
    hipLaunchKernelGGL((conv_tile_f16_kernel<K, S, INF32, CIN, NT>), dim3(tiles, cmax / (NT * 16), njobs), \
                       block, lds_, st, pm);                                                          \
    RAMP_CHECK_LAUNCH();                                                                              \
    return RAMP_OK;                                                                                   \
  }
  TILE_CASE(7, 2, true, 16, 2)
  // 16 x 16 output tiles for the 32-channel 3x3 layers at half resolution (600 workgroups of two towers, three per CU: still
  // one round): weight fragments staged once per 256 pixels and a 1.27x instead of 1.41x halo.  Per-block statistics
  // (`stats`, the path without accumulators) keep the 8 x 16 grid their buffers are sized for.  (8 x 16 tiles: the A/B of round 4)
  constexpr bool th16 = true;
  bool block_stats = false, any_skip = false;
  for (int t = 0; t < njobs; t++) { block_stats |= jobs[t].stats != nullptr; any_skip |= jobs[t].skip != nullptr; }
  if (any_skip) {                                     // a tower takes a fused residual-block tail: the TAIL instances
#define TAIL_CASE(K, S, CIN, NT, TH_)                                                               \
    if (KH == K && stride == S && !in_f32 && Cin == CIN && (NT == 2 || cgcd_ok64)) {                  \
      constexpr int lds_ = conv_tile_lds_bytes(K, S, false, CIN, NT, false, TH_);                     \
      if (conv_tile_attr(conv_tile_f16_kernel<K, S, false, CIN, NT, false, TH_, true>, lds_) != RAMP_OK) return RAMP_ELAUNCH; \
      hipLaunchKernelGGL((conv_tile_f16_kernel<K, S, false, CIN, NT, false, TH_, true>),              \
                         dim3(ramp_cdiv(OH, TH_) * ramp_cdiv(OW, 16), cmax / (NT * 16), njobs), block, lds_, st, pm); \
      RAMP_CHECK_LAUNCH();                                                                          \
      return RAMP_OK;                                                                               \
    }
    if (fp8 || block_stats) return RAMP_EUNSUPPORTED;
    if (th16 && (long)ramp_cdiv(OH, 16) * ramp_cdiv(OW, 16) * njobs * (cmax / 32) >= 512) { TAIL_CASE(3, 1, 32, 2, 16) }
    TAIL_CASE(3, 1, 32, 2, 8)
    TAIL_CASE(3, 2, 32, 2, 8)
    TAIL_CASE(1, 2, 32, 4, 8)
    TAIL_CASE(3, 1, 64, 2, 8)
    TAIL_CASE(1, 1, 64, 4, 8)
#undef TAIL_CASE
    return RAMP_EUNSUPPORTED;
  }
  if (th16 && !block_stats && KH == 3 && stride == 1 && !in_f32 && Cin == 32 &&
      (long)ramp_cdiv(OH, 16) * ramp_cdiv(OW, 16) * njobs * (cmax / 32) >= 512) {
    constexpr int lds_ = conv_tile_lds_bytes(3, 1, false, 32, 2, false, 16);
    if (conv_tile_attr(conv_tile_f16_kernel<3, 1, false, 32, 2, false, 16>, lds_) != RAMP_OK) return RAMP_ELAUNCH;
    hipLaunchKernelGGL((conv_tile_f16_kernel<3, 1, false, 32, 2, false, 16>),
                       dim3(ramp_cdiv(OH, 16) * ramp_cdiv(OW, 16), cmax / 32, njobs), block, lds_, st, pm);
    RAMP_CHECK_LAUNCH();
    return RAMP_OK;
  }
  TILE_CASE(3, 1, false, 32, 2)
  TILE_CASE(3, 2, false, 32, 2)
  // (stride 2 at 64 channels, the MultiScale towers' layer3: the 80 KB halo tile leaves one workgroup per CU either way;
  // with all 64 output channels in it the tile is staged once instead of twice -- round 4's A/B)
  constexpr bool s2nt4 = true;
  if (s2nt4) { TILE_CASE(3, 2, false, 64, 4) }
  TILE_CASE(3, 2, false, 64, 2)
  TILE_CASE(3, 1, false, 64, 2)
  TILE_CASE(1, 2, false, 32, 4)
  TILE_CASE(1, 2, false, 64, 4)
  TILE_CASE(1, 1, false, 64, 4)
  TILE_CASE(1, 1, false, 128, 4)
#undef TILE_CASE
  return RAMP_EUNSUPPORTED;
}

int ramp_conv2d_stats_blocks(int H, int W, int Cin, int Cout, int KH, int stride, int dtype) {
  const int pad = KH / 2;
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
  if (OH <= 0 || OW <= 0) return RAMP_EINVAL;
  const bool f16 = (dtype & 0xf) == RAMP_F16, in_f32 = f16 && (dtype & RAMP_IN_F32);
  if (dtype == (RAMP_F32 | RAMP_CONV_X3)) {
    const bool ok = (KH == 3 && stride == 1 && ((Cin == 32 && (Cout == 32 || Cout == 64)) || (Cin == 64 && (Cout == 64 || Cout == 32)))) ||
                    (KH == 3 && stride == 2 && Cin == 32 && Cout == 64) || (KH == 7 && stride == 2 && Cin == 16 && Cout == 32);
    return ok ? ramp_cdiv(OH, 8) * ramp_cdiv(OW, 16) : RAMP_EUNSUPPORTED;
  }
  bool tiled = false;
  if (f16 && !(dtype & RAMP_CONV_DIRECT)) {
    tiled = (KH == 7 && stride == 2 && in_f32 && Cin == 16 && Cout % 32 == 0) ||
            (!in_f32 && KH == 3 && stride == 1 && (Cin == 32 || Cin == 64) && Cout % 32 == 0) ||
            (!in_f32 && KH == 3 && stride == 2 && (Cin == 32 || Cin == 64) && Cout % 32 == 0) ||
            (!in_f32 && KH == 1 && stride == 2 && (Cin == 32 || Cin == 64) && Cout % 64 == 0) ||
            (!in_f32 && KH == 1 && stride == 1 && (Cin == 64 || Cin == 128) && Cout % 64 == 0);
  }
  return tiled ? ramp_cdiv(OH, 8) * ramp_cdiv(OW, 16) : ramp_cdiv(OH * OW, 128);
}

int ramp_in_stats_finalize(const float *partial, int nblk, int C, float count, float eps, float *scale,
                           float *shift, void *stream) {
  if (!partial || !scale || !shift || nblk <= 0 || C <= 0) return RAMP_EINVAL;
  hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partial, nblk,
                     C, count, eps, scale, shift);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_norm_add_relu(const float *y, const float *sy, const float *hy, const float *skip,
                       const float *ss, const float *hs, float *out, long n, int C, void *stream) {
  if (!y || !sy || !hy || !skip || !out || n <= 0 || C % 4 || n % 4) return RAMP_EINVAL;
  const long n4 = n / 4;
  hipLaunchKernelGGL(norm_add_relu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, y, sy, hy, skip, ss, hs, out, n4, C);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_ms_lstm_superstate_mfma(const float *ev, const float *im, const float *wfrag, const float *wsmall, float *state,
                                 void *state16, int H, int W, int scale, int has_state, int use_im, void *stream) {
  if (!ev || !im || !wfrag || !wsmall || !state || H <= 0 || W <= 0) return RAMP_EINVAL;
  if (scale != 1 && scale != 2 && scale != 4) return RAMP_EUNSUPPORTED;
  MsMfmaParams p;
  p.ev = ev; p.im = im; p.wfrag = wfrag; p.wsmall = wsmall; p.state = state; p.state16 = (_Float16 *)state16;
  p.H = H; p.W = W;
  const int k = scale > 1 ? scale + 1 : 1, pad = scale > 1 ? 1 : 0;
  p.Hs = (H + 2 * pad - k) / scale + 1;
  p.Ws = (W + 2 * pad - k) / scale + 1;
  if (p.Hs <= 0 || p.Ws <= 0) return RAMP_EINVAL;
  p.has_state = has_state; p.use_im = use_im; p.tiles_per_wave = 1;
  hipStream_t st = (hipStream_t)stream;
  if (scale > 1 && (p.Ws % 16)) return RAMP_EUNSUPPORTED;      // (a tile = 16 neighbours of one row)
  if (scale == 1) return ms_mfma_launch<16, 1, 4>(p, st);
  constexpr int split = 1;                            // (0: a whole tile per wave at scales 2 / 4 -- measured slower)
  const int ntile = ramp_cdiv(p.Hs * p.Ws, 16);
  if (split) {
    if (scale == 2) hipLaunchKernelGGL((ms_lstm_superstate_split_kernel<32, 2>), dim3(ntile), dim3(128), 0, st, p);
    else hipLaunchKernelGGL((ms_lstm_superstate_split_kernel<64, 4>), dim3(ntile), dim3(256), 0, st, p);
    RAMP_CHECK_LAUNCH();
    return RAMP_OK;
  }
  if (scale == 2) return ms_mfma_launch<32, 2, 4>(p, st);
  return ms_mfma_launch<64, 4, 6>(p, st);
}

int ramp_ms_lstm_superstate(const float *ev, const float *im, const float *const *weights_host,
                            float *state, int H, int W, int scale, int has_state, int use_im,
                            void *stream) {
  if (!ev || !im || !weights_host || !state || H <= 0 || W <= 0) return RAMP_EINVAL;
  for (int i = 0; i < 12; i++)
    if (!weights_host[i]) return RAMP_EINVAL;
  if (scale != 1 && scale != 2 && scale != 4) return RAMP_EUNSUPPORTED;
  MsLstmParams p;
  p.ev = ev; p.im = im;
  p.wce = weights_host[0]; p.bce = weights_host[1]; p.wci = weights_host[2]; p.bci = weights_host[3];
  p.wle = weights_host[4]; p.ble = weights_host[5]; p.wli = weights_host[6]; p.bli = weights_host[7];
  p.wme = weights_host[8]; p.bme = weights_host[9]; p.wmi = weights_host[10]; p.bmi = weights_host[11];
  p.state = state;
  p.H = H; p.W = W;
  const int k = scale > 1 ? scale + 1 : 1, pad = scale > 1 ? 1 : 0;
  p.Hs = (H + 2 * pad - k) / scale + 1;
  p.Ws = (W + 2 * pad - k) / scale + 1;
  if (p.Hs <= 0 || p.Ws <= 0) return RAMP_EINVAL;
  p.has_state = has_state; p.use_im = use_im;
  const int D = 16 * scale, np = (256 / D) * 4;
  const dim3 grid(ramp_cdiv(p.Hs * p.Ws, np));
  hipStream_t st = (hipStream_t)stream;
  if (scale == 1) hipLaunchKernelGGL((ms_lstm_superstate_kernel<16, 1>), grid, dim3(256), 0, st, p);
  else if (scale == 2) hipLaunchKernelGGL((ms_lstm_superstate_kernel<32, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((ms_lstm_superstate_kernel<64, 4>), grid, dim3(256), 0, st, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

}  // extern "C"
