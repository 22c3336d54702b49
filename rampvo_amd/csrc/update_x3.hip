// The update operator's Linear chains at fp32 accuracy on the f16 matrix cores ("f16x3").
//
// MIXED_PRECISION off (the configuration whose poses / depths agree with the reference to 1e-4, north_star's tolerance)
// asks for fp32 Linear layers (ramp/net.py:34-90 without autocast).  gfx950's f32-input MFMA runs at the vector rate
// (157 TFLOP/s, 1/16 of the f16 rate) and has no tf32-like form, so 224 GFLOP per update cost >= 1.4 ms there and
// 2.7 ms as library GEMMs + row kernels (round 5).  Here every fp32 operand is split into two fp16 numbers,
//     x = xh + xl,  xh = fp16(x),  xl = fp16((x - xh) 2^11) 2^-11          (|x - xh - xl| <= 2^-22 |x|)
// and a product x w is formed as three f16 MFMA products accumulated in ONE fp32 accumulator,
//     x w ~= xh wh + xl' (wh 2^-11) + xh wl          (xl' = xl 2^11; the xl wl term, 2^-22 |x w|, is dropped)
// -- 3/16 of the f32 MFMA's issue time at 22 instead of 24 bits per operand; the fp32 accumulation is the matrix
// core's own.  Range handling: a weight matrix is scaled by a power of two so that max |W| lands in [2^12, 2^13)
// (wh 2^-11 and the low parts stay fp16 NORMAL numbers; the epilogue multiplies by the inverse, exactly); an
// activation below the fp16 normal range goes into the scaled low part whole (no subnormal operand is relied on);
// |x| must stay below 65504 (the shipped MIXED_PRECISION path has the same limit).
//
// Structure as csrc/update_mlp.hip (one workgroup owns a row tile for a whole chain, activations in LDS -- here as a
// high and a low plane --, weights pre-packed in fragment order and streamed from L2, operands exchanged so that a lane
// holds four consecutive output columns of one row), arithmetic as the fp32 operator: nothing is rounded to fp16
// between layers, biases / LayerNorm / gate / residual stream / heads in fp32.
//
// Packed weight matrix (rampvo_amd/update_fused.py::pack_linear_x3), W [N = 384][K]:
//     [K/32][24][2][64 lanes][8] fp16 -- plane 0: fp16(W 2^s), plane 1: fp16(W 2^s - plane 0); lane (q, j) of fragment
//     (ks, nt) holds W[16 nt + j][32 ks + 8 q .. + 8]; followed by one float, 2^-s.
#include "ramp_device.h"
#include <stdlib.h>

#define XD 384
#define XS (XD + 8)            // LDS row stride of a plane (halfs)
#define XKS (XD / 32)
#define XNTW 3                 // 16-column tiles per wave
#define XWAVES 8
#define XFRAG 1024             // halfs per packed fragment pair (two planes x 64 lanes x 8)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// (elementwise on purpose: a vector-typed f4 add would select v_pk_add_f32 -- see the Makefile's note on packed fp32)
__device__ __forceinline__ f4 x3_add(const f4 a, const f4 b) { return (f4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }
__device__ __forceinline__ void x3_split(float v, _Float16 &hi, _Float16 &lo) {
  const _Float16 h = fabsf(v) < 6.103515625e-05f ? (_Float16)0.0f : (_Float16)v;
  hi = h;
  lo = (_Float16)((v - (float)h) * 2048.0f);
}
__device__ __forceinline__ void x3_split4(const f4 v, h4 &hi, h4 &lo) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    _Float16 h, l;
    x3_split(v[i], h, l);
    hi[i] = h; lo[i] = l;
  }
}
// 2^-s behind the fragments of a packed matrix with K = 32 nks
__device__ __forceinline__ float x3_inv_scale(const _Float16 *wp, int nks) {
  return *reinterpret_cast<const float *>(wp + (size_t)nks * (XD / 16) * XFRAG);
}

// acc[mt][nt] (+)= X[16 MT x 32 nks] W[:, 32 wks0 ..)^T for this wave's 48 columns, transposed accumulators (lane (q, j):
// row j of row tile mt, columns 4q .. 4q+3 of column tile nt).  One K step = 6 fragment loads (two planes x three column
// tiles), 2 MT LDS reads, 9 MT matrix instructions.
#ifndef X3_GRU_PF
#define X3_GRU_PF 0            // (the gru launch holds the residual stream and the gate in registers: no room for a ring)
#endif
#ifndef X3_PF
#define X3_PF 1                // K steps of weight fragments in flight ahead of the matrix work (0: loaded where used)
#endif
template <int MT>
__device__ __forceinline__ void x3_mma_step(const _Float16 *Xh, const _Float16 *Xl, int ks, int q, int j, const h8 (&wh)[XNTW],
                                            const h8 (&wl)[XNTW], f4 (&acc)[MT][XNTW]) {
  const _Float16 c11 = (_Float16)0.00048828125f;           // 2^-11
  const h8 s11 = (h8){c11, c11, c11, c11, c11, c11, c11, c11};
  h8 ah[MT], al[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int off = (mt * 16 + j) * XS + ks * 32 + 8 * q;
    ah[mt] = *reinterpret_cast<const h8 *>(Xh + off);
    al[mt] = *reinterpret_cast<const h8 *>(Xl + off);
  }
#if defined(X3_DIAG_NOMMA)                                    // (diagnostic build: operand traffic without the matrix work)
#pragma unroll
  for (int nt = 0; nt < XNTW; nt++)
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[mt][nt][i] += (float)wh[nt][i] + (float)wl[nt][i + 4] + (float)ah[mt][i] + (float)al[mt][i + 4];
#else
#pragma unroll
  for (int nt = 0; nt < XNTW; nt++) {
    const h8 ws = wh[nt] * s11;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], ah[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws, al[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], ah[mt], acc[mt][nt], 0, 0, 0);
  }
#endif
}

// NKS K steps, fully unrolled, the weight fragments of steps ks+1 .. ks+PF in flight while the matrix cores work on step ks (a
// ring of PF+1 fragment sets with static indices; the scheduling barrier keeps each load ABOVE the matrix work of the step it is
// issued in -- the compiler sinks it to its first use otherwise, which exposes the L2 latency on every K step: two waves per
// SIMD cannot cover it)
template <int MT, int NKS, int PF>
__device__ __forceinline__ void x3_gemm_static(const _Float16 *Xh, const _Float16 *Xl, const _Float16 *wp, int wks0, int wave,
                                               int lane, f4 (&acc)[MT][XNTW]) {
  const int q = lane >> 4, j = lane & 15;
  const _Float16 *wb0 = wp + ((size_t)wks0 * (XD / 16) + wave * XNTW) * XFRAG + lane * 8;
  h8 rh[PF + 1][XNTW], rl[PF + 1][XNTW];
  auto wload = [&](int ks, h8 (&wh)[XNTW], h8 (&wl)[XNTW]) {
#if defined(X3_DIAG_NOW)                                      // (diagnostic build: every K step reads the first step's fragments)
    const _Float16 *wb = wb0 + (size_t)(ks & 0) * (XD / 16) * XFRAG;
#else
    const _Float16 *wb = wb0 + (size_t)ks * (XD / 16) * XFRAG;
#endif
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) {
      wh[nt] = *reinterpret_cast<const h8 *>(wb + nt * XFRAG);
      wl[nt] = *reinterpret_cast<const h8 *>(wb + nt * XFRAG + 512);
    }
  };
#pragma unroll
  for (int d = 0; d < PF && d < NKS; d++) wload(d, rh[d], rl[d]);
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) {
    if (ks + PF < NKS) wload(ks + PF, rh[(ks + PF) % (PF + 1)], rl[(ks + PF) % (PF + 1)]);
    if (PF > 0) __builtin_amdgcn_sched_barrier(0);
    x3_mma_step<MT>(Xh, Xl, ks, q, j, rh[ks % (PF + 1)], rl[ks % (PF + 1)], acc);
  }
}

template <int MT, bool ZERO, int PF = X3_PF>
__device__ __forceinline__ void x3_gemm(const _Float16 *Xh, const _Float16 *Xl, const _Float16 *wp, int nks, int wks0,
                                        int wave, int lane, f4 (&acc)[MT][XNTW]) {
  if (ZERO) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < XNTW; nt++) acc[mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  if (PF > 0 && nks == XKS) return x3_gemm_static<MT, XKS, PF>(Xh, Xl, wp, wks0, wave, lane, acc);
  if (PF > 0 && nks == 4) return x3_gemm_static<MT, 4, PF>(Xh, Xl, wp, wks0, wave, lane, acc);
  const int q = lane >> 4, j = lane & 15;
#pragma unroll 1
  for (int ks = 0; ks < nks; ks++) {
    const _Float16 *wb = wp + ((size_t)(wks0 + ks) * (XD / 16) + wave * XNTW) * XFRAG + lane * 8;
    h8 wh[XNTW], wl[XNTW];
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) {
      wh[nt] = *reinterpret_cast<const h8 *>(wb + nt * XFRAG);
      wl[nt] = *reinterpret_cast<const h8 *>(wb + nt * XFRAG + 512);
    }
    x3_mma_step<MT>(Xh, Xl, ks, q, j, wh, wl, acc);
  }
}

// v (transposed accumulator layout) -> the two LDS planes
template <int MT>
__device__ __forceinline__ void x3_to_lds(_Float16 *Xh, _Float16 *Xl, const f4 (&v)[MT][XNTW], int cq, int j) {
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) {
      h4 hi, lo;
      x3_split4(v[mt][nt], hi, lo);
      const int off = (mt * 16 + j) * XS + cq + nt * 16;
      *reinterpret_cast<h4 *>(Xh + off) = hi;
      *reinterpret_cast<h4 *>(Xl + off) = lo;
    }
}


// Stage ROWS x (4 NV4) floats into the two LDS planes: ALL global loads of a thread are issued before the first conversion (a loop
// that loads, converts and stores element by element waits for one memory round trip per iteration: twelve in a row per tile)
template <int ROWS, int NV4, typename LoadT>
__device__ __forceinline__ void x3_stage(_Float16 *Xh, _Float16 *Xl, int tid, LoadT load) {
  constexpr int N = (ROWS * NV4 + 64 * XWAVES - 1) / (64 * XWAVES);
  f4 v[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int i = tid + k * (64 * XWAVES);
    const int r = i / NV4, c4 = i - r * NV4;
    v[k] = (ROWS * NV4) % (64 * XWAVES) == 0 || i < ROWS * NV4 ? load(r, c4) : (f4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int i = tid + k * (64 * XWAVES);
    const int r = i / NV4, c4 = i - r * NV4;
    if ((ROWS * NV4) % (64 * XWAVES) == 0 || i < ROWS * NV4) {
      h4 hi, lo;
      x3_split4(v[k], hi, lo);
      *reinterpret_cast<h4 *>(Xh + r * XS + 4 * c4) = hi;
      *reinterpret_cast<h4 *>(Xl + r * XS + 4 * c4) = lo;
    }
  }
}

// v = acc * inv_scale + bias (optionally ReLU)
template <int MT, bool RELU>
__device__ __forceinline__ void x3_bias(f4 (&v)[MT][XNTW], float inv, const float *__restrict__ bias, int cq) {
#pragma unroll
  for (int nt = 0; nt < XNTW; nt++) {
    const f4 b = *reinterpret_cast<const f4 *>(bias + cq + nt * 16);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float t = v[mt][nt][i] * inv + b[i];
        v[mt][nt][i] = RELU ? fmaxf(t, 0.f) : t;
      }
  }
}

// two-pass LayerNorm of the tile in transposed accumulator layout (the arithmetic of csrc/update_mlp.hip::tile_ln: per-wave
// partial sums meet in an LDS table [rows][8 waves], every lane adds the eight partials of its rows in wave order)
template <int MT>
__device__ __forceinline__ void x3_tile_ln(f4 (&v)[MT][XNTW], const float *__restrict__ w, const float *__restrict__ b,
                                           float eps, float *T1, float *T2, int wave, int q, int j, int cq) {
  float mean[MT], rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) s += (v[mt][nt][0] + v[mt][nt][1]) + (v[mt][nt][2] + v[mt][nt][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) T1[(mt * 16 + j) * XWAVES + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const f4 a = *reinterpret_cast<const f4 *>(T1 + (mt * 16 + j) * XWAVES), c = *reinterpret_cast<const f4 *>(T1 + (mt * 16 + j) * XWAVES + 4);
    mean[mt] = ((((((a[0] + a[1]) + a[2]) + a[3]) + c[0]) + c[1]) + c[2] + c[3]) * (1.0f / XD);
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float d = v[mt][nt][i] - mean[mt];
        s += d * d;
      }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) T2[(mt * 16 + j) * XWAVES + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const f4 a = *reinterpret_cast<const f4 *>(T2 + (mt * 16 + j) * XWAVES), c = *reinterpret_cast<const f4 *>(T2 + (mt * 16 + j) * XWAVES + 4);
    rstd[mt] = 1.0f / sqrtf(((((((a[0] + a[1]) + a[2]) + a[3]) + c[0]) + c[1]) + c[2] + c[3]) * (1.0f / XD) + eps);
  }
#pragma unroll
  for (int nt = 0; nt < XNTW; nt++) {
    const f4 wv = *reinterpret_cast<const f4 *>(w + cq + nt * 16), bv = *reinterpret_cast<const f4 *>(b + cq + nt * 16);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int i = 0; i < 4; i++) v[mt][nt][i] = (v[mt][nt][i] - mean[mt]) * rstd[mt] * wv[i] + bv[i];
  }
}

// Workgroups that start together run their phases together: every CU gathers rows (HBM saturated, matrix cores idle), then every
// CU multiplies (HBM idle).  With several workgroups per CU the second one of each CU (dispatch order: 256 workgroups fill the
// first slot of every CU) starts X3_STAGGER x ~4k cycles late, so that one's row traffic runs beside the other's matrix work.
#ifndef X3_STAGGER
#define X3_STAGGER 0
#endif
__device__ __forceinline__ void x3_stagger() {
#if X3_STAGGER > 0
  if ((blockIdx.x >> 8) & 1) {
#pragma unroll 1
    for (int i = 0; i < X3_STAGGER; i++) __builtin_amdgcn_s_sleep(64);
  }
#endif
}
#define X3_COMMON(MT_)                                                                                      \
  constexpr int ROWS = 16 * (MT_);                                                                          \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];                                  \
  _Float16 *Xh = reinterpret_cast<_Float16 *>(smem_raw), *Xl = Xh + ROWS * XS;                              \
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;              \
  const int row0 = blockIdx.x * ROWS;                                                                       \
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;                                                           \
  if (row0 >= pE) return;                                                                                   \
  const int cq = wave * (16 * XNTW) + 4 * q;                                                                \
  x3_stagger();                                                                                             \
  (void)tid; (void)j; (void)cq

// Tiles (measured in the fp32 frame, profiles/r06_f_x3_tiles_ab.txt: 64-row tiles / one workgroup per CU 496 kf/s; 32-row tiles at
// four waves per SIMD for the three lighter chains + 48 rows for gru 519): the chains are bound by their row traffic at one
// workgroup per CU, two smaller ones overlap it with the other's matrix work.  Row-local arithmetic: the tile size changes no value.
#ifndef X3_MT
#define X3_MT 2
#endif
#ifndef X3_OCC
#define X3_OCC 4
#endif
#ifndef X3_GRU_MT
#define X3_GRU_MT 3
#endif
#if X3_OCC > 0
#define X3_ATTR __attribute__((amdgpu_waves_per_eu(X3_OCC, X3_OCC)))
#else
#define X3_ATTR
#endif
static size_t x3_lds_bytes(int mt, bool ln) { return (size_t)2 * 16 * mt * XS * 2 + (ln ? (size_t)2 * 16 * mt * XWAVES * 4 : 0); }

// ------------------------------------------------------------------ c1 / c2 (ramp/net.py:77-82)
struct X3NbrParams {
  const float *net_in;         // [E][384]
  const int64_t *idx;          // [E] neighbour row or -1
  const _Float16 *wa, *wb;     // packed (x3)
  const float *ba, *bb;        // fp32 biases
  float *net_out;              // [E][384], must not alias net_in
  int E;
  const int32_t *dyn;
};

template <int MT>
__global__ void __launch_bounds__(64 * XWAVES) X3_ATTR x3_nbr_kernel(const X3NbrParams p) {
  X3_COMMON(MT);
  x3_stage<ROWS, XD / 4>(Xh, Xl, tid, [&](int r, int c4) {
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    if (row0 + r < pE) {
      const long src = p.idx[row0 + r];
      if (src >= 0) v = *reinterpret_cast<const f4 *>(p.net_in + (size_t)src * XD + 4 * c4);
    }
    return v;
  });
  __syncthreads();
  f4 acc[MT][XNTW];
  x3_gemm<MT, true>(Xh, Xl, p.wa, XKS, 0, wave, lane, acc);
  x3_bias<MT, true>(acc, x3_inv_scale(p.wa, XKS), p.ba, cq);
  __syncthreads();                                       // every wave is past its reads of x
  x3_to_lds<MT>(Xh, Xl, acc, cq, j);
  __syncthreads();
  x3_gemm<MT, true>(Xh, Xl, p.wb, XKS, 0, wave, lane, acc);
  x3_bias<MT, false>(acc, x3_inv_scale(p.wb, XKS), p.bb, cq);
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int row = row0 + mt * 16 + j;
    if (row >= pE) continue;
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) {
      const f4 x = *reinterpret_cast<const f4 *>(p.net_in + (size_t)row * XD + cq + nt * 16);
      *reinterpret_cast<f4 *>(p.net_out + (size_t)row * XD + cq + nt * 16) = x3_add(x, acc[mt][nt]);
    }
  }
}

// ------------------------------------------------------------------ correlation MLP + Update.norm (ramp/net.py:57-62, 71-74)
//   c = Linear3(relu(LayerNorm(Linear2(relu(Linear1(corr))))));  net = LayerNorm((net_prev + inp) + c)
struct X3CorrMlpParams {
  const float *corr;           // [E][corr_k] fp32 (corr_k a multiple of 32: 896 = 882 + zero tail)
  int corr_k;
  const _Float16 *w1, *w2, *w3;
  const float *b1, *b2, *b3;
  const float *ln_w, *ln_b;    // corr[3]
  float ln_eps;
  const float *net;            // [*][384] previous hidden state or NULL (zeros)
  const int64_t *net_map;      // [E] row of `net` per factor (-1: zero row) or NULL (identity)
  const float *inp;            // context table [*][384] fp32
  const int64_t *inp_idx;      // [E] row of `inp` (modulo inp_mod when > 0) or NULL (identity)
  long inp_mod;
  const float *norm_w, *norm_b;
  float norm_eps;
  float *net_out;              // [E][384]
  int E;
  const int32_t *dyn;
};

template <int MT>
__global__ void __launch_bounds__(64 * XWAVES) X3_ATTR x3_corr_mlp_kernel(const X3CorrMlpParams p) {
  X3_COMMON(MT);
  float *T1 = reinterpret_cast<float *>(Xl + ROWS * XS), *T2 = T1 + ROWS * XWAVES;
  f4 acc[MT][XNTW];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) acc[mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
  const int nks_total = p.corr_k / 32;
  for (int ks0 = 0; ks0 < nks_total; ks0 += XKS) {
    const int nks = min(XKS, nks_total - ks0);
    auto ld = [&](int r, int c4) {
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (row0 + r < pE) v = *reinterpret_cast<const f4 *>(p.corr + (size_t)(row0 + r) * p.corr_k + ks0 * 32 + 4 * c4);
      return v;
    };
    if (nks == XKS) x3_stage<ROWS, XKS * 8>(Xh, Xl, tid, ld);
    else if (nks == 4) x3_stage<ROWS, 32>(Xh, Xl, tid, ld);
    else {
      const int v4 = nks * 8;                                // 16-byte vectors per row of this chunk
      for (int i = tid; i < ROWS * v4; i += 64 * XWAVES) {
        const int r = i / v4, c4 = i - r * v4;
        h4 hi, lo;
        x3_split4(ld(r, c4), hi, lo);
        *reinterpret_cast<h4 *>(Xh + r * XS + 4 * c4) = hi;
        *reinterpret_cast<h4 *>(Xl + r * XS + 4 * c4) = lo;
      }
    }
    __syncthreads();
    x3_gemm<MT, false, 0>(Xh, Xl, p.w1, nks, ks0, wave, lane, acc);      // (the staged chunk's registers are live: no ring)
    __syncthreads();                                       // before the tile is overwritten
  }
  x3_bias<MT, true>(acc, x3_inv_scale(p.w1, nks_total), p.b1, cq);
  x3_to_lds<MT>(Xh, Xl, acc, cq, j);
  __syncthreads();
  x3_gemm<MT, true>(Xh, Xl, p.w2, XKS, 0, wave, lane, acc);
  x3_bias<MT, false>(acc, x3_inv_scale(p.w2, XKS), p.b2, cq);
  x3_tile_ln<MT>(acc, p.ln_w, p.ln_b, p.ln_eps, T1, T2, wave, q, j, cq);       // (its barriers: every wave is past its reads)
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[mt][nt][i] = fmaxf(acc[mt][nt][i], 0.f);
  x3_to_lds<MT>(Xh, Xl, acc, cq, j);
  __syncthreads();
  x3_gemm<MT, true>(Xh, Xl, p.w3, XKS, 0, wave, lane, acc);
  x3_bias<MT, false>(acc, x3_inv_scale(p.w3, XKS), p.b3, cq);
  // (net_prev + inp) + c, LayerNorm
  long rrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int row = min(row0 + mt * 16 + j, pE - 1);
    rrow[mt] = row;
    const long ra = p.net ? (p.net_map ? p.net_map[row] : (long)row) : -1;
    long rb = p.inp_idx ? p.inp_idx[row] : (long)row;
    if (p.inp_mod > 0) rb %= p.inp_mod;
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) {
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (ra >= 0) v = *reinterpret_cast<const f4 *>(p.net + (size_t)ra * XD + cq + nt * 16);
      v = x3_add(v, *reinterpret_cast<const f4 *>(p.inp + (size_t)rb * XD + cq + nt * 16));
      acc[mt][nt] = x3_add(v, acc[mt][nt]);
    }
  }
  x3_tile_ln<MT>(acc, p.norm_w, p.norm_b, p.norm_eps, T1, T2, wave, q, j, cq);
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    if (row0 + mt * 16 + j >= pE) continue;
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) *reinterpret_cast<f4 *>(p.net_out + (size_t)rrow[mt] * XD + cq + nt * 16) = acc[mt][nt];
  }
}

// ------------------------------------------------------------------ SoftAgg front half: [f(x) | g(x)] rows (ramp/blocks.py:42-46)
struct X3FgParams {
  const float *x32;            // [E][384]
  const float *add_t;          // optional [groups][384] fp32: x = x32 + add_t[add_idx]
  const int32_t *add_idx;
  float *x32_out;              // optional: x written back (may be x32)
  const _Float16 *wf, *wg;
  const float *bf, *bg;
  float *fg;                   // [E][768] fp32
  int E;
  const int32_t *dyn;
};

template <int MT>
__global__ void __launch_bounds__(64 * XWAVES) X3_ATTR x3_fg_kernel(const X3FgParams p) {
  X3_COMMON(MT);
  x3_stage<ROWS, XD / 4>(Xh, Xl, tid, [&](int r, int c4) {
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    const int row = row0 + r;
    if (row < pE) {
      v = *reinterpret_cast<const f4 *>(p.x32 + (size_t)row * XD + 4 * c4);
      if (p.add_t) v = x3_add(v, *reinterpret_cast<const f4 *>(p.add_t + (size_t)p.add_idx[row] * XD + 4 * c4));
      if (p.x32_out) *reinterpret_cast<f4 *>(p.x32_out + (size_t)row * XD + 4 * c4) = v;
    }
    return v;
  });
  __syncthreads();
#pragma unroll 1
  for (int part = 0; part < 2; part++) {
    f4 acc[MT][XNTW];
    const _Float16 *w = part ? p.wg : p.wf;
    x3_gemm<MT, true>(Xh, Xl, w, XKS, 0, wave, lane, acc);
    x3_bias<MT, false>(acc, x3_inv_scale(w, XKS), part ? p.bg : p.bf, cq);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const int row = row0 + mt * 16 + j;
      if (row >= pE) continue;
#pragma unroll
      for (int nt = 0; nt < XNTW; nt++)
        *reinterpret_cast<f4 *>(p.fg + (size_t)row * (2 * XD) + part * XD + cq + nt * 16) = acc[mt][nt];
    }
  }
}

// segment softmax-sum over the rows of [f | g] (ramp/blocks.py:46-48), fp32: y[g] = sum_e softmax_e(g_e) f_e over the factors of
// group g in `order`.  8 row lanes x 96 threads x 4 channels, online softmax per lane, lanes merged in lane order.
#define X3SEG_R 8
#define X3SEG_T 96
__global__ void __launch_bounds__(X3SEG_R *X3SEG_T) x3_segment_softmax_kernel(const float *__restrict__ fg, const int32_t *__restrict__ order,
                                                                               const int32_t *__restrict__ seg_start,
                                                                               const int32_t *__restrict__ ngroups, float *__restrict__ y) {
  __shared__ float part[X3SEG_R][12][X3SEG_T];
  const int g = blockIdx.x;
  const int t = threadIdx.x % X3SEG_T, w = threadIdx.x / X3SEG_T;
  const int c = 4 * t;
  if (g >= *ngroups) {          // unused tail of the table: defined (zero) rows
    if (w == 0) *reinterpret_cast<f4 *>(y + (size_t)g * XD + c) = (f4){0.f, 0.f, 0.f, 0.f};
    return;
  }
  const int s0 = seg_start[g], s1 = seg_start[g + 1];
  float m[4], z[4], a[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { m[k] = -INFINITY; z[k] = 0.f; a[k] = 0.f; }
  for (int pp = s0 + w; pp < s1; pp += X3SEG_R) {
    const size_t r0 = (size_t)order[pp] * (2 * XD);
    const f4 fv = *reinterpret_cast<const f4 *>(fg + r0 + c);
    const f4 gv = *reinterpret_cast<const f4 *>(fg + r0 + XD + c);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float n = fmaxf(m[k], gv[k]);
      const float sc = expf(m[k] - n), e = expf(gv[k] - n);
      z[k] = z[k] * sc + e; a[k] = a[k] * sc + fv[k] * e;
      m[k] = n;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { part[w][k][t] = m[k]; part[w][4 + k][t] = z[k]; part[w][8 + k][t] = a[k]; }
  __syncthreads();
  if (w != 0) return;
  for (int r = 1; r < X3SEG_R; r++) {
    if (part[r][4][t] == 0.f) continue;       // this row lane saw no row (all channels share the rows)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float mk = part[r][k][t], n = fmaxf(m[k], mk);
      const float sc = expf(m[k] - n), tc = expf(mk - n);
      z[k] = z[k] * sc + part[r][4 + k][t] * tc; a[k] = a[k] * sc + part[r][8 + k][t] * tc;
      m[k] = n;
    }
  }
  *reinterpret_cast<f4 *>(y + (size_t)g * XD + c) = (f4){a[0] / z[0], a[1] / z[1], a[2] / z[2], a[3] / z[3]};
}

// ------------------------------------------------------------------ one Linear 384 -> 384 on a table of rows (SoftAgg's h)
struct X3LinParams {
  const float *x;              // [rows][384]
  const _Float16 *w;
  const float *b;
  float *y;                    // [rows][384]
  int E;                       // rows (launch bound)
  const int32_t *rows_dev;     // optional device-side row count
  const int32_t *dyn;          // unused (X3_COMMON)
};

__global__ void __launch_bounds__(64 * XWAVES) x3_linear_kernel(const X3LinParams p) {
  constexpr int MT = 1;
  constexpr int ROWS = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xh = reinterpret_cast<_Float16 *>(smem_raw), *Xl = Xh + ROWS * XS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * ROWS;
  const int pE = p.rows_dev ? min(*p.rows_dev, p.E) : p.E;
  if (row0 >= pE) return;
  const int cq = wave * (16 * XNTW) + 4 * q;
  for (int i = tid; i < ROWS * (XD / 4); i += 64 * XWAVES) {
    const int r = i / (XD / 4), c4 = i - r * (XD / 4);
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    if (row0 + r < pE) v = *reinterpret_cast<const f4 *>(p.x + (size_t)(row0 + r) * XD + 4 * c4);
    h4 hi, lo;
    x3_split4(v, hi, lo);
    *reinterpret_cast<h4 *>(Xh + r * XS + 4 * c4) = hi;
    *reinterpret_cast<h4 *>(Xl + r * XS + 4 * c4) = lo;
  }
  __syncthreads();
  f4 acc[MT][XNTW];
  x3_gemm<MT, true>(Xh, Xl, p.w, XKS, 0, wave, lane, acc);
  x3_bias<MT, false>(acc, x3_inv_scale(p.w, XKS), p.b, cq);
  const int row = row0 + j;
  if (row < pE) {
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) *reinterpret_cast<f4 *>(p.y + (size_t)row * XD + cq + nt * 16) = acc[0][nt];
  }
}

// ------------------------------------------------------------------ gru (+ heads) (ramp/net.py:49-54, 64-66, 87-90; ramp/blocks.py:15-31)
struct X3GruParams {
  const float *x32;            // [E][384]: the residual stream entering gru[0]
  const float *add0_t;         // optional first expand-and-add table (fp32) -- (x32 + add0_t[add0_idx]) + add_t[add_idx]
  const int32_t *add0_idx;
  const float *add_t;          // optional: with it x = LayerNorm_pre(x32 (+ add0) + add_t[add_idx]) (gru[0]); without: x = x32
  const int32_t *add_idx;
  const float *pre_w, *pre_b;
  float pre_eps;
  const _Float16 *wp[6];       // g1_gate, g1_r1, g1_r2, g2_gate, g2_r1, g2_r2 (x3 packs)
  const float *bias[6];
  const float *ln_w, *ln_b;    // gru[2]
  float eps;
  float *out32;                // [E][384]
  float *relu32;               // optional [E][384]: relu(result)
  const float *heads_w;        // optional [4][384] fp32 (d rows 0..1, w rows 0..1) -> target / weight
  const float *heads_b;        // [4]
  const float *coords;         // [E][2][PP]
  float *target, *weight;      // [E][2]
  int PP, ctr;
  float wd, ht;
  uint32_t *gate_flag;         // optional: workgroup 0 stores gate_seq here when it starts
  uint32_t gate_seq;
  int E;
  const int32_t *dyn;
};

__device__ __forceinline__ float x3_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int MT>
__global__ void __launch_bounds__(64 * XWAVES) x3_gru_kernel(const X3GruParams p) {
  constexpr int ROWS = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xh = reinterpret_cast<_Float16 *>(smem_raw), *Xl = Xh + ROWS * XS;
  float *T1 = reinterpret_cast<float *>(Xl + ROWS * XS), *T2 = T1 + ROWS * XWAVES;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * ROWS;
  if (p.gate_flag && blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(p.gate_flag, p.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;
  const int cq = wave * (16 * XNTW) + 4 * q;

  // ---- the residual stream, fp32, in registers (rows past E: clamped loads, no stores)
  f4 res[MT][XNTW];
  size_t roff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int row = min(row0 + mt * 16 + j, pE - 1);
    roff[mt] = (size_t)row * XD;
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) res[mt][nt] = *reinterpret_cast<const f4 *>(p.x32 + roff[mt] + cq + nt * 16);
  }
  if (p.add0_t) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const float *a = p.add0_t + (size_t)p.add0_idx[roff[mt] / XD] * XD;
#pragma unroll
      for (int nt = 0; nt < XNTW; nt++) res[mt][nt] = x3_add(res[mt][nt], *reinterpret_cast<const f4 *>(a + cq + nt * 16));
    }
  }
  if (p.add_t) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const float *a = p.add_t + (size_t)p.add_idx[roff[mt] / XD] * XD;
#pragma unroll
      for (int nt = 0; nt < XNTW; nt++) res[mt][nt] = x3_add(res[mt][nt], *reinterpret_cast<const f4 *>(a + cq + nt * 16));
    }
    x3_tile_ln<MT>(res, p.pre_w, p.pre_b, p.pre_eps, T1, T2, wave, q, j, cq);
  }
  x3_to_lds<MT>(Xh, Xl, res, cq, j);
  __syncthreads();

#pragma unroll 1
  for (int stage = 0; stage < 2; stage++) {
    const int wb = 3 * stage;
    f4 gate[MT][XNTW], acc[MT][XNTW];
    x3_gemm<MT, true, X3_GRU_PF>(Xh, Xl, p.wp[wb + 0], XKS, 0, wave, lane, gate);
    {
      const float inv = x3_inv_scale(p.wp[wb + 0], XKS);
#pragma unroll
      for (int nt = 0; nt < XNTW; nt++) {
        const f4 b = *reinterpret_cast<const f4 *>(p.bias[wb + 0] + cq + nt * 16);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int i = 0; i < 4; i++) gate[mt][nt][i] = x3_sigmoid(gate[mt][nt][i] * inv + b[i]);
      }
    }
    x3_gemm<MT, true, X3_GRU_PF>(Xh, Xl, p.wp[wb + 1], XKS, 0, wave, lane, acc);
    x3_bias<MT, true>(acc, x3_inv_scale(p.wp[wb + 1], XKS), p.bias[wb + 1], cq);
    __syncthreads();                            // every wave is past its reads of x: h takes its place
    x3_to_lds<MT>(Xh, Xl, acc, cq, j);
    __syncthreads();
    x3_gemm<MT, true, X3_GRU_PF>(Xh, Xl, p.wp[wb + 2], XKS, 0, wave, lane, acc);
    x3_bias<MT, false>(acc, x3_inv_scale(p.wp[wb + 2], XKS), p.bias[wb + 2], cq);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < XNTW; nt++)
#pragma unroll
        for (int i = 0; i < 4; i++) res[mt][nt][i] += gate[mt][nt][i] * acc[mt][nt][i];
    if (stage == 0) {
      x3_tile_ln<MT>(res, p.ln_w, p.ln_b, p.eps, T1, T2, wave, q, j, cq);   // (its barriers: every wave is past its reads of h)
      x3_to_lds<MT>(Xh, Xl, res, cq, j);
      __syncthreads();
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    if (row0 + mt * 16 + j >= pE) continue;
#pragma unroll
    for (int nt = 0; nt < XNTW; nt++) {
      const f4 v = res[mt][nt];
      *reinterpret_cast<f4 *>(p.out32 + roff[mt] + cq + nt * 16) = v;
      if (p.relu32)
        *reinterpret_cast<f4 *>(p.relu32 + roff[mt] + cq + nt * 16) =
            (f4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    }
  }
  if (p.heads_w) {                              // (uniform)
    // d / w heads: 4 dot products of relu(result) with the head rows per row of the tile: a lane sums its 12 columns, the four
    // quarter-lanes of a row meet over two shuffles, the eight waves over an LDS table; fixed order throughout
    float part[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int c = 0; c < 4; c++) part[mt][c] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int nt = 0; nt < XNTW; nt++) {
        const f4 wv = *reinterpret_cast<const f4 *>(p.heads_w + c * XD + cq + nt * 16);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int i = 0; i < 4; i++) part[mt][c] = __builtin_fmaf(fmaxf(res[mt][nt][i], 0.f), wv[i], part[mt][c]);
      }
    float *HT = reinterpret_cast<float *>(Xh);              // [rows][8 waves][4]
    __syncthreads();                                        // every wave is past its reads of h
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float v = part[mt][c];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        part[mt][c] = v;
      }
    if (q == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
        *reinterpret_cast<f4 *>(HT + ((mt * 16 + j) * XWAVES + wave) * 4) = (f4){part[mt][0], part[mt][1], part[mt][2], part[mt][3]};
    }
    __syncthreads();
    const int e = row0 + tid;
    if (tid < ROWS && e < pE) {
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < XWAVES; w++) {
        const f4 v = *reinterpret_cast<const f4 *>(HT + (tid * XWAVES + w) * 4);
#pragma unroll
        for (int c = 0; c < 4; c++) o[c] += v[c];
      }
#pragma unroll
      for (int c = 0; c < 4; c++) o[c] += p.heads_b[c];
      const float wx = x3_sigmoid(o[2]), wy = x3_sigmoid(o[3]);
      const float tx = p.coords[((size_t)e * 2 + 0) * p.PP + p.ctr] + o[0];
      const float ty = p.coords[((size_t)e * 2 + 1) * p.PP + p.ctr] + o[1];
      const bool outside = (tx < 0) || (tx > p.wd) || (ty < 0) || (ty > p.ht);
      p.target[2 * (size_t)e + 0] = tx;
      p.target[2 * (size_t)e + 1] = ty;
      p.weight[2 * (size_t)e + 0] = outside ? 0.0f : wx;
      p.weight[2 * (size_t)e + 1] = outside ? 0.0f : wy;
    }
  }
}

// ------------------------------------------------------------------ launchers

template <typename KernelT, typename ParamsT>
static int x3_launch(KernelT kernel, const ParamsT &p, int rows, int mt, bool ln, hipStream_t st) {
  const size_t lds = x3_lds_bytes(mt, ln);
  // (per call: the attribute is per device, and a process may drive several)
  if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return RAMP_ELAUNCH;
  hipLaunchKernelGGL(kernel, dim3(ramp_cdiv(rows, 16 * mt)), dim3(64 * XWAVES), lds, st, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

extern "C" {

int ramp_i_x3_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb, const float *bb,
                  float *net_out, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!net_in || !idx || !wa || !ba || !wb || !bb || !net_out || net_in == net_out) return RAMP_EINVAL;
  X3NbrParams p;
  p.net_in = net_in; p.idx = idx; p.wa = (const _Float16 *)wa; p.wb = (const _Float16 *)wb; p.ba = ba; p.bb = bb;
  p.net_out = net_out; p.E = E; p.dyn = dyn;
  return x3_launch(x3_nbr_kernel<X3_MT>, p, E, X3_MT, false, (hipStream_t)stream);
}

int ramp_i_x3_corr_mlp(const float *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                       const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps, const float *net,
                       const int64_t *net_map, const float *inp, const int64_t *inp_idx, long inp_mod, const float *norm_w,
                       const float *norm_b, float norm_eps, float *net_out, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!corr || corr_k <= 0 || (corr_k & 31) || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !ln_w || !ln_b || !inp || !norm_w ||
      !norm_b || !net_out || net == net_out)
    return RAMP_EINVAL;
  X3CorrMlpParams p;
  p.corr = corr; p.corr_k = corr_k; p.w1 = (const _Float16 *)w1; p.w2 = (const _Float16 *)w2; p.w3 = (const _Float16 *)w3;
  p.b1 = b1; p.b2 = b2; p.b3 = b3; p.ln_w = ln_w; p.ln_b = ln_b; p.ln_eps = ln_eps; p.net = net; p.net_map = net_map;
  p.inp = inp; p.inp_idx = inp_idx; p.inp_mod = inp_mod; p.norm_w = norm_w; p.norm_b = norm_b; p.norm_eps = norm_eps;
  p.net_out = net_out; p.E = E; p.dyn = dyn;
  return x3_launch(x3_corr_mlp_kernel<X3_MT>, p, E, X3_MT, true, (hipStream_t)stream);
}

int ramp_i_x3_fg(const float *x32, const float *add_t, const int32_t *add_idx, float *x32_out, const void *wf, const float *bf,
                 const void *wg, const float *bg, float *fg, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!x32 || !wf || !bf || !wg || !bg || !fg || (add_t && !add_idx)) return RAMP_EINVAL;
  X3FgParams p;
  p.x32 = x32; p.add_t = add_t; p.add_idx = add_idx; p.x32_out = x32_out; p.wf = (const _Float16 *)wf;
  p.wg = (const _Float16 *)wg; p.bf = bf; p.bg = bg; p.fg = fg; p.E = E; p.dyn = dyn;
  return x3_launch(x3_fg_kernel<X3_MT>, p, E, X3_MT, false, (hipStream_t)stream);
}

int ramp_x3_segment_softmax(const float *fg, const int32_t *order, const int32_t *seg_start, const int32_t *ngroups, float *y,
                            int max_groups, void *stream) {
  if (max_groups < 0) return RAMP_EINVAL;
  if (max_groups == 0) return RAMP_OK;
  if (!fg || !order || !seg_start || !ngroups || !y) return RAMP_EINVAL;
  hipLaunchKernelGGL(x3_segment_softmax_kernel, dim3(max_groups), dim3(X3SEG_R * X3SEG_T), 0, (hipStream_t)stream, fg, order,
                     seg_start, ngroups, y);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_x3_linear(const float *x, const void *w_packed, const float *bias, float *y, int rows, const int32_t *rows_dev,
                   void *stream) {
  if (rows < 0) return RAMP_EINVAL;
  if (rows == 0) return RAMP_OK;
  if (!x || !w_packed || !bias || !y) return RAMP_EINVAL;
  X3LinParams p;
  p.x = x; p.w = (const _Float16 *)w_packed; p.b = bias; p.y = y; p.E = rows; p.rows_dev = rows_dev; p.dyn = nullptr;
  return x3_launch(x3_linear_kernel, p, rows, 1, false, (hipStream_t)stream);
}

int ramp_i_x3_gru(const float *x32, const float *add0_t, const int32_t *add0_idx, const float *add_t, const int32_t *add_idx,
                  const float *pre_w, const float *pre_b, float pre_eps, const void *const *wp_host,
                  const float *const *bias_host, const float *ln_w, const float *ln_b, float eps, float *out32, float *relu32,
                  int E, const int32_t *dyn, const float *heads_w, const float *heads_b, const float *coords, float *target,
                  float *weight, int P, float wd, float ht, uint32_t *gate_flag, uint32_t gate_seq, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!x32 || !wp_host || !bias_host || !ln_w || !ln_b || !out32) return RAMP_EINVAL;
  if (heads_w && (!heads_b || !coords || !target || !weight || P < 1)) return RAMP_EINVAL;
  if (add_t && (!add_idx || !pre_w || !pre_b)) return RAMP_EINVAL;
  if (add0_t && (!add0_idx || !add_t)) return RAMP_EINVAL;
  X3GruParams p;
  p.x32 = x32; p.add0_t = add0_t; p.add0_idx = add0_idx; p.add_t = add_t; p.add_idx = add_idx;
  p.pre_w = pre_w; p.pre_b = pre_b; p.pre_eps = pre_eps;
  for (int i = 0; i < 6; i++) {
    if (!wp_host[i] || !bias_host[i]) return RAMP_EINVAL;
    p.wp[i] = (const _Float16 *)wp_host[i];
    p.bias[i] = bias_host[i];
  }
  p.ln_w = ln_w; p.ln_b = ln_b; p.eps = eps; p.out32 = out32; p.relu32 = relu32; p.heads_w = heads_w; p.heads_b = heads_b;
  p.coords = coords; p.target = target; p.weight = weight; p.PP = P * P; p.ctr = (P / 2) * P + P / 2; p.wd = wd; p.ht = ht;
  p.gate_flag = gate_flag; p.gate_seq = gate_seq; p.E = E; p.dyn = dyn;
  return x3_launch(x3_gru_kernel<X3_GRU_MT>, p, E, X3_GRU_MT, true, (hipStream_t)stream);
}

// ---- public entry points: host-side sizes
int ramp_x3_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb, const float *bb,
                float *net_out, int E, void *stream) {
  return ramp_i_x3_nbr(net_in, idx, wa, ba, wb, bb, net_out, E, nullptr, stream);
}
int ramp_x3_corr_mlp(const float *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                     const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps, const float *net,
                     const int64_t *net_map, const float *inp, const int64_t *inp_idx, long inp_mod, const float *norm_w,
                     const float *norm_b, float norm_eps, float *net_out, int E, void *stream) {
  return ramp_i_x3_corr_mlp(corr, corr_k, w1, b1, w2, b2, w3, b3, ln_w, ln_b, ln_eps, net, net_map, inp, inp_idx, inp_mod, norm_w,
                            norm_b, norm_eps, net_out, E, nullptr, stream);
}
int ramp_x3_fg(const float *x32, const float *add_t, const int32_t *add_idx, float *x32_out, const void *wf, const float *bf,
               const void *wg, const float *bg, float *fg, int E, void *stream) {
  return ramp_i_x3_fg(x32, add_t, add_idx, x32_out, wf, bf, wg, bg, fg, E, nullptr, stream);
}
int ramp_x3_gru(const float *x32, const float *add0_t, const int32_t *add0_idx, const float *add_t, const int32_t *add_idx,
                const float *pre_w, const float *pre_b, float pre_eps, const void *const *wp_host, const float *const *bias_host,
                const float *ln_w, const float *ln_b, float eps, float *out32, float *relu32, int E, const float *heads_w,
                const float *heads_b, const float *coords, float *target, float *weight, int P, float wd, float ht, void *stream) {
  return ramp_i_x3_gru(x32, add0_t, add0_idx, add_t, add_idx, pre_w, pre_b, pre_eps, wp_host, bias_host, ln_w, ln_b, eps, out32,
                       relu32, E, nullptr, heads_w, heads_b, coords, target, weight, P, wd, ht, nullptr, 0u, stream);
}

}  // extern "C"
