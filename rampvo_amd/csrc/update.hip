// Row-fused kernels of the update operator (reference: ramp/net.py:69-90, ramp/blocks.py:15-50).
//
// The operator is ~19 GEMMs over [E, 384] activations glued by gathers, residual adds,
// LayerNorms, gates and dtype casts; the reference (and a naive port) spends more launches and
// HBM passes on the glue than on the GEMMs.  Here the glue is four kernels, each ONE pass over
// the rows with one wavefront per row (64 lanes x 6 channels, wave-shuffle LayerNorm):
//
//   upd_row_fuse      t = A[e] + B[idxB(e)] + C[idxC(e)]  [-> LayerNorm(w,b)] [-> ReLU]
//                     -> fp32 state and/or GEMM-dtype copy          (adds, gathers, norms, casts)
//   upd_gather_mask   out[e] = idx[e] >= 0 ? X[idx[e]] : 0                    (temporal neighbours)
//   upd_gated         t = X + sigmoid(G) * R  [-> LayerNorm] -> fp32, copy, ReLU copy (GatedResidual)
//   upd_heads         target = centre(coords) + delta, weight = sigmoid(w) * in_bounds(target)
//                     (heads' activation + Ramp_vo.update's target/filter_features, ramp/utils.py:557-570)
//
// plus the SoftAgg segment kernel in its single-pass (online softmax) form reading the stacked
// [f | g] GEMM output.  The hidden state `net` stays fp32 (as under the reference's autocast);
// T is the GEMM I/O dtype (half under MIXED_PRECISION, float otherwise).
#include "ramp_device.h"

#define UD 384

template <typename T> __device__ __forceinline__ float2 ld2(const T *p);
template <> __device__ __forceinline__ float2 ld2<float>(const float *p) {
  return *reinterpret_cast<const float2 *>(p);
}
template <> __device__ __forceinline__ float2 ld2<_Float16>(const _Float16 *p) {
  const __half2 h = *reinterpret_cast<const __half2 *>(p);
  return __half22float2(h);
}
template <typename T> __device__ __forceinline__ void st2(T *p, float a, float b);
template <> __device__ __forceinline__ void st2<float>(float *p, float a, float b) {
  *reinterpret_cast<float2 *>(p) = make_float2(a, b);
}
template <> __device__ __forceinline__ void st2<_Float16>(_Float16 *p, float a, float b) {
  *reinterpret_cast<__half2 *>(p) = __floats2half2_rn(a, b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LayerNorm over the 384 channels of a row held as v[3][2] per lane (channels 2*lane + 128*k + {0,1})
__device__ __forceinline__ void row_layernorm(float v[3][2], const float *__restrict__ w,
                                              const float *__restrict__ b, float eps, int lane) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) s += v[k][0] + v[k][1];
  const float mean = wave_sum(s) * (1.0f / UD);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float a = v[k][0] - mean, c = v[k][1] - mean;
    q += a * a + c * c;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / UD) + eps);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int c = 2 * lane + 128 * k;
    v[k][0] = (v[k][0] - mean) * rstd * w[c] + b[c];
    v[k][1] = (v[k][1] - mean) * rstd * w[c + 1] + b[c + 1];
  }
}

struct RowFuseParams {
  const float *A;            // [*][384] fp32 or null
  const int64_t *idxA;       // optional row of A per edge; < 0: a zero row (an edge new in this update)
  const void *B, *C;         // [*][384] T or null
  const int64_t *idxB, *idxC;   // optional int64 row index (null: row e)
  const int32_t *idxB32, *idxC32;  // optional int32 row index
  long modB;                 // idxB taken modulo modB when > 0 (ring buffer)
  const float *ln_w, *ln_b;  // LayerNorm affine or null
  float eps;
  int relu;
  float *out_f32;            // optional
  void *out_t;               // optional, dtype T
  int E;
};

template <typename T>
__global__ void __launch_bounds__(256) upd_row_fuse_kernel(const RowFuseParams p) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= p.E) return;
  float v[3][2];
#pragma unroll
  for (int k = 0; k < 3; k++) { v[k][0] = 0.f; v[k][1] = 0.f; }
  if (p.A) {
    const long ra = p.idxA ? p.idxA[e] : (long)e;
    if (ra >= 0) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float2 a = *reinterpret_cast<const float2 *>(p.A + (size_t)ra * UD + 2 * lane + 128 * k);
        v[k][0] = a.x; v[k][1] = a.y;
      }
    }
  }
  if (p.B) {
    long r = p.idxB ? p.idxB[e] : (p.idxB32 ? (long)p.idxB32[e] : (long)e);
    if (p.modB > 0) r %= p.modB;
    const T *B = reinterpret_cast<const T *>(p.B) + (size_t)r * UD;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float2 a = ld2<T>(B + 2 * lane + 128 * k);
      v[k][0] += a.x; v[k][1] += a.y;
    }
  }
  if (p.C) {
    const long r = p.idxC ? p.idxC[e] : (p.idxC32 ? (long)p.idxC32[e] : (long)e);
    const T *C = reinterpret_cast<const T *>(p.C) + (size_t)r * UD;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float2 a = ld2<T>(C + 2 * lane + 128 * k);
      v[k][0] += a.x; v[k][1] += a.y;
    }
  }
  if (p.ln_w) row_layernorm(v, p.ln_w, p.ln_b, p.eps, lane);
  if (p.relu) {
#pragma unroll
    for (int k = 0; k < 3; k++) { v[k][0] = fmaxf(v[k][0], 0.f); v[k][1] = fmaxf(v[k][1], 0.f); }
  }
  if (p.out_f32) {
#pragma unroll
    for (int k = 0; k < 3; k++)
      *reinterpret_cast<float2 *>(p.out_f32 + (size_t)e * UD + 2 * lane + 128 * k) = make_float2(v[k][0], v[k][1]);
  }
  if (p.out_t) {
    T *o = reinterpret_cast<T *>(p.out_t) + (size_t)e * UD;
#pragma unroll
    for (int k = 0; k < 3; k++) st2<T>(o + 2 * lane + 128 * k, v[k][0], v[k][1]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
    upd_gather_mask_kernel(const float *__restrict__ X, const int64_t *__restrict__ idx,
                           T *__restrict__ out, int E) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  const long r = idx[e];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float2 a = make_float2(0.f, 0.f);
    if (r >= 0) a = *reinterpret_cast<const float2 *>(X + (size_t)r * UD + 2 * lane + 128 * k);
    st2<T>(out + (size_t)e * UD + 2 * lane + 128 * k, a.x, a.y);
  }
}

struct GatedParams {
  const float *X;           // [E][384] fp32
  const void *G, *R;        // gate pre-activation, residual branch (T)
  const float *ln_w, *ln_b; // optional LayerNorm applied to the result
  float eps;
  float *out_f32;           // optional
  void *out_t;              // optional copy (T)
  void *out_relu_t;         // optional ReLU copy (T)
  int E;
};

template <typename T>
__global__ void __launch_bounds__(256) upd_gated_kernel(const GatedParams p) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= p.E) return;
  const T *G = reinterpret_cast<const T *>(p.G) + (size_t)e * UD;
  const T *R = reinterpret_cast<const T *>(p.R) + (size_t)e * UD;
  float v[3][2];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int c = 2 * lane + 128 * k;
    const float2 x = *reinterpret_cast<const float2 *>(p.X + (size_t)e * UD + c);
    const float2 g = ld2<T>(G + c), r = ld2<T>(R + c);
    v[k][0] = x.x + (1.0f / (1.0f + expf(-g.x))) * r.x;
    v[k][1] = x.y + (1.0f / (1.0f + expf(-g.y))) * r.y;
  }
  if (p.ln_w) row_layernorm(v, p.ln_w, p.ln_b, p.eps, lane);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int c = 2 * lane + 128 * k;
    if (p.out_f32) *reinterpret_cast<float2 *>(p.out_f32 + (size_t)e * UD + c) = make_float2(v[k][0], v[k][1]);
    if (p.out_t) st2<T>(reinterpret_cast<T *>(p.out_t) + (size_t)e * UD + c, v[k][0], v[k][1]);
    if (p.out_relu_t)
      st2<T>(reinterpret_cast<T *>(p.out_relu_t) + (size_t)e * UD + c, fmaxf(v[k][0], 0.f), fmaxf(v[k][1], 0.f));
  }
}

// heads: hw [E][4] = (delta_x, delta_y, w_x_logit, w_y_logit); coords [E][2][P][P]
template <typename T>
__global__ void __launch_bounds__(256)
    upd_heads_kernel(const T *__restrict__ hw, const float *__restrict__ coords, float *__restrict__ target,
                     float *__restrict__ weight, float *__restrict__ delta, int E, int PP, int ctr,
                     float wd, float ht) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float dx = (float)hw[4 * (size_t)e + 0], dy = (float)hw[4 * (size_t)e + 1];
  // the reference applies Sigmoid in the GEMM dtype, then .float()
  const float wx = (float)(T)(1.0f / (1.0f + expf(-(float)hw[4 * (size_t)e + 2])));
  const float wy = (float)(T)(1.0f / (1.0f + expf(-(float)hw[4 * (size_t)e + 3])));
  const float tx = coords[((size_t)e * 2 + 0) * PP + ctr] + dx;
  const float ty = coords[((size_t)e * 2 + 1) * PP + ctr] + dy;
  const bool outside = (tx < 0) || (tx > wd) || (ty < 0) || (ty > ht);
  target[2 * (size_t)e + 0] = tx;
  target[2 * (size_t)e + 1] = ty;
  weight[2 * (size_t)e + 0] = outside ? 0.0f : wx;
  weight[2 * (size_t)e + 1] = outside ? 0.0f : wy;
  if (delta) { delta[2 * (size_t)e + 0] = dx; delta[2 * (size_t)e + 1] = dy; }
}

// fp16 path: the two heads' Linear layers (ramp/net.py:64-66; 4 outputs per edge) and the epilogue above in one
// launch -- a [E,384]x[384,4] GEMM is a row-wise dot product: one wave per edge row (lane l: channels 2l + 128k +
// {0,1}), DPP wave sums, lane 0 finishes like upd_heads_kernel.  Outputs are rounded to fp16 where the GEMM's are.
__device__ __forceinline__ float heads_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
typedef _Float16 hd_h2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256)
    upd_heads_linear_f16_kernel(const _Float16 *__restrict__ relu_t, const _Float16 *__restrict__ hwt,
                                const float *__restrict__ hb, const float *__restrict__ coords,
                                float *__restrict__ target, float *__restrict__ weight, int E, int PP, int ctr,
                                float wd, float ht, const int32_t *__restrict__ dyn) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (dyn) E = dyn[RAMP_DYN_E];                 // device-side size: the argument is the launch bound
  if (e >= E) return;
  float x[6];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const hd_h2 v = *reinterpret_cast<const hd_h2 *>(relu_t + (size_t)e * UD + 2 * lane + 128 * k);
    x[2 * k] = (float)v[0]; x[2 * k + 1] = (float)v[1];
  }
  float o[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const hd_h2 wv = *reinterpret_cast<const hd_h2 *>(hwt + c * UD + 2 * lane + 128 * k);
      acc += x[2 * k] * (float)wv[0];
      acc += x[2 * k + 1] * (float)wv[1];
    }
    o[c] = (float)(_Float16)(heads_wave_sum(acc) + hb[c]);          // the Linear output is a half tensor
  }
  if (lane != 0) return;
  const float wx = (float)(_Float16)(1.0f / (1.0f + expf(-o[2])));
  const float wy = (float)(_Float16)(1.0f / (1.0f + expf(-o[3])));
  const float tx = coords[((size_t)e * 2 + 0) * PP + ctr] + o[0];
  const float ty = coords[((size_t)e * 2 + 1) * PP + ctr] + o[1];
  const bool outside = (tx < 0) || (tx > wd) || (ty < 0) || (ty > ht);
  target[2 * (size_t)e + 0] = tx;
  target[2 * (size_t)e + 1] = ty;
  weight[2 * (size_t)e + 0] = outside ? 0.0f : wx;
  weight[2 * (size_t)e + 1] = outside ? 0.0f : wy;
}

// SoftAgg core over the stacked [f | g] rows (row stride 768): y[g][c] = sum softmax(g) * f.
// One workgroup per group: SEG_R row lanes x 96 threads x 4 channels.  Row lane w walks rows w, w + R, ... of
// the group (ascending edge order) with a running max (online softmax); the R partial (max, sum, weighted
// sum) triples are merged in lane order.  (A single row lane per group left the 96-row pair groups with a
// 48-step dependent chain on 420 workgroups.)
// exp: the hardware exponential (v_exp_f32, ~1 ulp) -- with libm's expf this kernel was VALU bound (6 x ~15
// instructions per row and lane pair); it only runs on the fp16 path, whose inputs carry 11 bits.
#define SEG_R 8
#ifndef SEG_NT
#define SEG_NT 0     // 1: the [f | g] rows (read once) with the nontemporal hint -- part of the A/B of csrc/update_mlp.hip's UPD_NT
#endif
#if SEG_NT
#define SEG_LD(p) __builtin_nontemporal_load(p)
#else
#define SEG_LD(p) (*(p))
#endif
#define SEG_T 96                // threads per row lane: 4 channels (one 8-byte load) each
typedef _Float16 seg_h4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(SEG_T * SEG_R)
    upd_segment_softmax_f16_kernel(const _Float16 *__restrict__ fg, const int32_t *__restrict__ order,
                                   const int32_t *__restrict__ seg_start, const int32_t *__restrict__ ngroups,
                                   _Float16 *__restrict__ y) {
  __shared__ float part[SEG_R][12][SEG_T];
  const int g = blockIdx.x;
  const int t = threadIdx.x % SEG_T, w = threadIdx.x / SEG_T;
  const int c = 4 * t;
  if (g >= *ngroups) {          // unused tail of the table: defined (zero) rows, no memset launch needed
    if (w == 0) *reinterpret_cast<seg_h4 *>(y + (size_t)g * UD + c) = (seg_h4){0, 0, 0, 0};
    return;
  }
  const int s0 = seg_start[g], s1 = seg_start[g + 1];
  float m[4], z[4], a[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { m[k] = -INFINITY; z[k] = 0.f; a[k] = 0.f; }
  for (int p = s0 + w; p < s1; p += SEG_R) {
    const size_t r0 = (size_t)order[p] * (2 * UD);
    const seg_h4 fv = SEG_LD(reinterpret_cast<const seg_h4 *>(fg + r0 + c));
    const seg_h4 gv = SEG_LD(reinterpret_cast<const seg_h4 *>(fg + r0 + UD + c));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float gk = (float)gv[k], n = fmaxf(m[k], gk);
      const float sc = __expf(m[k] - n), e = __expf(gk - n);
      z[k] = z[k] * sc + e; a[k] = a[k] * sc + (float)fv[k] * e;
      m[k] = n;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { part[w][k][t] = m[k]; part[w][4 + k][t] = z[k]; part[w][8 + k][t] = a[k]; }
  __syncthreads();
  if (w != 0) return;
  for (int q = 1; q < SEG_R; q++) {
    if (part[q][4][t] == 0.f) continue;       // this row lane saw no row (all channels share the rows)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float mk = part[q][k][t], n = fmaxf(m[k], mk);
      const float sc = __expf(m[k] - n), tc = __expf(mk - n);
      z[k] = z[k] * sc + part[q][4 + k][t] * tc; a[k] = a[k] * sc + part[q][8 + k][t] * tc;
      m[k] = n;
    }
  }
  *reinterpret_cast<seg_h4 *>(y + (size_t)g * UD + c) =
      (seg_h4){(_Float16)(a[0] / z[0]), (_Float16)(a[1] / z[1]), (_Float16)(a[2] / z[2]), (_Float16)(a[3] / z[3])};
}

// The same arithmetic (row lane w walks rows w, w + R, ... in order; partials merged in lane order: bit-identical
// results) with more bytes in flight: a thread owns 8 channels (two 16-byte loads per row instead of two 8-byte ones),
// a row lane is 48 threads, and the indices and rows of SEG_D steps are requested before the first one is used.  The
// 8-byte kernel keeps 16 B per thread behind a dependent index load in flight -- 24 KB per CU at its two workgroups per
// CU, i.e. ~2.8 TB/s at ~2 us of latency, which is what it measured (61 MB in 22 us).
#define SEG_T8 48
#define SEG_D 4
typedef _Float16 seg_h8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(SEG_T8 * SEG_R)
    upd_segment_softmax_f16x8_kernel(const _Float16 *__restrict__ fg, const int32_t *__restrict__ order,
                                     const int32_t *__restrict__ seg_start, const int32_t *__restrict__ ngroups,
                                     _Float16 *__restrict__ y) {
  __shared__ float part[SEG_R][24][SEG_T8];
  const int g = blockIdx.x;
  const int t = threadIdx.x % SEG_T8, w = threadIdx.x / SEG_T8;
  const int c = 8 * t;
  if (g >= *ngroups) {          // unused tail of the table: defined (zero) rows
    if (w == 0) *reinterpret_cast<seg_h8 *>(y + (size_t)g * UD + c) = (seg_h8){0, 0, 0, 0, 0, 0, 0, 0};
    return;
  }
  const int s0 = seg_start[g], s1 = seg_start[g + 1];
  float m[8], z[8], a[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { m[k] = -INFINITY; z[k] = 0.f; a[k] = 0.f; }
  for (int p0 = s0 + w; p0 < s1; p0 += SEG_R * SEG_D) {
    int idx[SEG_D];
#pragma unroll
    for (int d = 0; d < SEG_D; d++) { const int p = p0 + d * SEG_R; idx[d] = order[p < s1 ? p : p0]; }
    seg_h8 fv[SEG_D], gv[SEG_D];
#pragma unroll
    for (int d = 0; d < SEG_D; d++) {
      const size_t r0 = (size_t)idx[d] * (2 * UD);
      fv[d] = SEG_LD(reinterpret_cast<const seg_h8 *>(fg + r0 + c));
      gv[d] = SEG_LD(reinterpret_cast<const seg_h8 *>(fg + r0 + UD + c));
    }
#pragma unroll
    for (int d = 0; d < SEG_D; d++) {
      if (p0 + d * SEG_R >= s1) break;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float gk = (float)gv[d][k], n = fmaxf(m[k], gk);
        const float sc = __expf(m[k] - n), e = __expf(gk - n);
        z[k] = z[k] * sc + e; a[k] = a[k] * sc + (float)fv[d][k] * e;
        m[k] = n;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) { part[w][k][t] = m[k]; part[w][8 + k][t] = z[k]; part[w][16 + k][t] = a[k]; }
  __syncthreads();
  if (w != 0) return;
  for (int q = 1; q < SEG_R; q++) {
    if (part[q][8][t] == 0.f) continue;       // this row lane saw no row (all channels share the rows)
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float mk = part[q][k][t], n = fmaxf(m[k], mk);
      const float sc = __expf(m[k] - n), tc = __expf(mk - n);
      z[k] = z[k] * sc + part[q][8 + k][t] * tc; a[k] = a[k] * sc + part[q][16 + k][t] * tc;
      m[k] = n;
    }
  }
  seg_h8 o;
#pragma unroll
  for (int k = 0; k < 8; k++) o[k] = (_Float16)(a[k] / z[k]);
  *reinterpret_cast<seg_h8 *>(y + (size_t)g * UD + c) = o;
}

// Sequential variant (fp32 path): the summation order the fp32 parity fixtures were recorded with.
// y[g][c] = sum softmax(g) * f over the stacked [f | g] rows (row stride 768).
// One workgroup (192 lanes x 2 channels) per group, single pass with a running max (online
// softmax), rows visited in ascending edge order.
template <typename T>
__global__ void __launch_bounds__(192)
    upd_segment_softmax_seq_kernel(const T *__restrict__ fg, const int32_t *__restrict__ order,
                               const int32_t *__restrict__ seg_start, const int32_t *__restrict__ ngroups,
                               T *__restrict__ y) {
  const int g = blockIdx.x;
  const int c = 2 * threadIdx.x;
  if (g >= *ngroups) {          // unused tail of the table: defined (zero) rows, no memset launch needed
    st2<T>(y + (size_t)g * UD + c, 0.f, 0.f);
    return;
  }
  const int s0 = seg_start[g], s1 = seg_start[g + 1];
  float m0 = -INFINITY, m1 = -INFINITY, z0 = 0.f, z1 = 0.f, a0 = 0.f, a1 = 0.f;
  int p = s0;
  for (; p + 1 < s1; p += 2) {   // two rows in flight
    const size_t r0 = (size_t)order[p] * (2 * UD), r1 = (size_t)order[p + 1] * (2 * UD);
    const float2 f0 = ld2<T>(fg + r0 + c), g0 = ld2<T>(fg + r0 + UD + c);
    const float2 f1 = ld2<T>(fg + r1 + c), g1 = ld2<T>(fg + r1 + UD + c);
    {
      const float n0 = fmaxf(m0, fmaxf(g0.x, g1.x)), n1 = fmaxf(m1, fmaxf(g0.y, g1.y));
      const float s0_ = expf(m0 - n0), s1_ = expf(m1 - n1);
      const float e00 = expf(g0.x - n0), e01 = expf(g1.x - n0), e10 = expf(g0.y - n1), e11 = expf(g1.y - n1);
      z0 = z0 * s0_ + e00 + e01; a0 = a0 * s0_ + f0.x * e00 + f1.x * e01;
      z1 = z1 * s1_ + e10 + e11; a1 = a1 * s1_ + f0.y * e10 + f1.y * e11;
      m0 = n0; m1 = n1;
    }
  }
  if (p < s1) {
    const size_t r0 = (size_t)order[p] * (2 * UD);
    const float2 f0 = ld2<T>(fg + r0 + c), g0 = ld2<T>(fg + r0 + UD + c);
    const float n0 = fmaxf(m0, g0.x), n1 = fmaxf(m1, g0.y);
    const float s0_ = expf(m0 - n0), s1_ = expf(m1 - n1);
    const float e0 = expf(g0.x - n0), e1 = expf(g0.y - n1);
    z0 = z0 * s0_ + e0; a0 = a0 * s0_ + f0.x * e0;
    z1 = z1 * s1_ + e1; a1 = a1 * s1_ + f0.y * e1;
  }
  st2<T>(y + (size_t)g * UD + c, a0 / z0, a1 / z1);
}

extern "C" {

int ramp_upd_row_fuse(const float *A, const int64_t *idxA, const void *B, const void *C, const int64_t *idxB,
                      const int32_t *idxB32, long modB, const int64_t *idxC, const int32_t *idxC32,
                      const float *ln_w, const float *ln_b, float eps, int relu, float *out_f32,
                      void *out_t, int E, int dtype, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if ((!A && !B && !C) || (!out_f32 && !out_t) || ((ln_w == nullptr) != (ln_b == nullptr))) return RAMP_EINVAL;
  RowFuseParams p;
  p.A = A; p.idxA = idxA; p.B = B; p.C = C; p.idxB = idxB; p.idxC = idxC; p.idxB32 = idxB32; p.idxC32 = idxC32;
  p.modB = modB; p.ln_w = ln_w; p.ln_b = ln_b; p.eps = eps; p.relu = relu;
  p.out_f32 = out_f32; p.out_t = out_t; p.E = E;
  const dim3 grid(ramp_cdiv(E, 4)), block(256);
  if (dtype == RAMP_F32) hipLaunchKernelGGL(upd_row_fuse_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
  else if (dtype == RAMP_F16) hipLaunchKernelGGL(upd_row_fuse_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, p);
  else return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_upd_gather_mask(const float *X, const int64_t *idx, void *out, int E, int dtype, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!X || !idx || !out) return RAMP_EINVAL;
  const dim3 grid(ramp_cdiv(E, 4)), block(256);
  if (dtype == RAMP_F32)
    hipLaunchKernelGGL(upd_gather_mask_kernel<float>, grid, block, 0, (hipStream_t)stream, X, idx, (float *)out, E);
  else if (dtype == RAMP_F16)
    hipLaunchKernelGGL(upd_gather_mask_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, X, idx,
                       (_Float16 *)out, E);
  else return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_upd_gated(const float *X, const void *G, const void *R, const float *ln_w, const float *ln_b,
                   float eps, float *out_f32, void *out_t, void *out_relu_t, int E, int dtype,
                   void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!X || !G || !R || ((ln_w == nullptr) != (ln_b == nullptr))) return RAMP_EINVAL;
  GatedParams p;
  p.X = X; p.G = G; p.R = R; p.ln_w = ln_w; p.ln_b = ln_b; p.eps = eps;
  p.out_f32 = out_f32; p.out_t = out_t; p.out_relu_t = out_relu_t; p.E = E;
  const dim3 grid(ramp_cdiv(E, 4)), block(256);
  if (dtype == RAMP_F32) hipLaunchKernelGGL(upd_gated_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
  else if (dtype == RAMP_F16) hipLaunchKernelGGL(upd_gated_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, p);
  else return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_upd_heads(const void *hw, const float *coords, float *target, float *weight, float *delta,
                   int E, int P, float wd, float ht, int dtype, void *stream) {
  if (E < 0 || P < 1) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!hw || !coords || !target || !weight) return RAMP_EINVAL;
  const int PP = P * P, ctr = (P / 2) * P + P / 2;
  const dim3 grid(ramp_cdiv(E, 256)), block(256);
  if (dtype == RAMP_F32)
    hipLaunchKernelGGL(upd_heads_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)hw, coords,
                       target, weight, delta, E, PP, ctr, wd, ht);
  else if (dtype == RAMP_F16)
    hipLaunchKernelGGL(upd_heads_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, (const _Float16 *)hw,
                       coords, target, weight, delta, E, PP, ctr, wd, ht);
  else return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_i_upd_heads_linear(const void *relu_t, const void *heads_w, const float *heads_b, const float *coords,
                            float *target, float *weight, int E, int P, float wd, float ht, const int32_t *dyn,
                            void *stream) {
  if (E < 0 || P < 1) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!relu_t || !heads_w || !heads_b || !coords || !target || !weight) return RAMP_EINVAL;
  const int PP = P * P, ctr = (P / 2) * P + P / 2;
  hipLaunchKernelGGL(upd_heads_linear_f16_kernel, dim3(ramp_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16 *)relu_t, (const _Float16 *)heads_w, heads_b, coords, target, weight, E, PP, ctr,
                     wd, ht, dyn);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_upd_heads_linear(const void *relu_t, const void *heads_w, const float *heads_b, const float *coords,
                          float *target, float *weight, int E, int P, float wd, float ht, void *stream) {
  return ramp_i_upd_heads_linear(relu_t, heads_w, heads_b, coords, target, weight, E, P, wd, ht, nullptr, stream);
}

int ramp_upd_segment_softmax(const void *fg, const int32_t *order, const int32_t *seg_start,
                             const int32_t *ngroups, void *y, int max_groups, int dtype, void *stream) {
  if (max_groups < 0) return RAMP_EINVAL;
  if (max_groups == 0) return RAMP_OK;
  if (!fg || !order || !seg_start || !ngroups || !y) return RAMP_EINVAL;
  const dim3 grid(max_groups);
  if (dtype == RAMP_F32)
    hipLaunchKernelGGL(upd_segment_softmax_seq_kernel<float>, grid, dim3(192), 0, (hipStream_t)stream,
                       (const float *)fg, order, seg_start, ngroups, (float *)y);
  else if (dtype == RAMP_F16) {
    const int v8 = 1;
    // many short groups (the patch grouping: ~2100 x ~19 rows) gain from the 16-byte kernel (27.3 -> 23.7 us); the pair
    // grouping's 420 x 96 rows are bound by the work per thread and want the 768-thread kernel (18.1 vs 21.9 us)
    if (v8 && max_groups >= 1024)
      hipLaunchKernelGGL(upd_segment_softmax_f16x8_kernel, grid, dim3(SEG_T8 * SEG_R), 0, (hipStream_t)stream,
                         (const _Float16 *)fg, order, seg_start, ngroups, (_Float16 *)y);
    else
      hipLaunchKernelGGL(upd_segment_softmax_f16_kernel, grid, dim3(SEG_T * SEG_R), 0, (hipStream_t)stream,
                         (const _Float16 *)fg, order, seg_start, ngroups, (_Float16 *)y);
  }
  else return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

}  // extern "C"
