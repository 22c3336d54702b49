// Fused GEMM chains of the update operator on the matrix cores (fp16 in, fp32 accumulate).
//
// The per-edge MLPs of the operator (ramp/net.py:49-54 `gru`, :43-46 `c1`/`c2`, ramp/blocks.py:15-31
// GatedResidual) are chains of [E,384]x[384,384] Linear layers with row-local glue between them.  As
// separate library GEMMs every layer reads and writes the full [E,384] activation (2 x 30 MB) and
// every glue step is another pass; at N = K = 384 a GEMM is neither compute nor bandwidth bound but
// tile-quantised and launch-latency bound.  Here a workgroup owns 64 rows for the WHOLE chain:
//   * the activation tile lives in LDS (fp16, [64][384+8]); the A fragment of lane (q, j) is row j,
//     channels 8q..8q+7 of a 32-channel K step: one ds_read_b128, conflict free with the +8 pad;
//   * 8 waves, wave w owns output columns [48w, 48w+48): 4 x 3 accumulator tiles of
//     v_mfma_f32_16x16x32_f16; weights are pre-packed in fragment order (one contiguous KB per load
//     instruction) and streamed from L2 -- a wave never re-reads a fragment;
//   * bias / ReLU / sigmoid gate / residual / LayerNorm run on the accumulator registers; only the
//     next layer's fp16 input goes back to LDS.  Linear outputs are rounded to fp16 where the
//     reference's autocast would (they are half tensors there).
#include "ramp_device.h"
#include <stdlib.h>

#define MD 384                 // feature width
#define MBM 64                 // rows per workgroup
#define MXS (MD + 8)           // LDS row stride (halfs)
#define MKS (MD / 32)          // K steps per 384-wide layer
#define MNTW 3                 // 16-column tiles per wave
#define MWAVES 8
#ifndef MLP_PREFETCH_DEFAULT
#define MLP_PREFETCH_DEFAULT 0
#endif
#ifndef GRU_PF
#define GRU_PF 2
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Nontemporal hints on the operator's state traffic (the [E,384] fp32 residual stream in and out of every chain, the
// [E,896] correlation rows, the [E,768] [f|g] rows: ~1.3 GB per update) -- VERDICT r3 item 1a -- measured on MI355X
// (tools/ab_build.sh, alternating builds on one box): with the hint on EVERY such access the step is 12 % SLOWER
// (899 vs 1020 kf/s; sequential 730 vs 824): a chain's 61 MB output is the next chain's input and is served from the
// memory-side cache -- the hint takes that away.  UPD_NT = 1 is that build; UPD_NT_DEAD = 1 hints only the loads of rows
// nobody reads again (the correlation rows, the previous hidden state), see DESIGN.md section 8.
#ifndef UPD_NT
#define UPD_NT 0
#endif
#ifndef UPD_NT_DEAD
#define UPD_NT_DEAD 0
#endif
template <typename T>
__device__ __forceinline__ T ld_dead(const T *p) {
#if UPD_NT || UPD_NT_DEAD
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T>
__device__ __forceinline__ T ld_st(const T *p) {
#if UPD_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void st_st(T *p, T v) {
#if UPD_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ float2 ld_st2(const float *p) { const f2 v = ld_st(reinterpret_cast<const f2 *>(p)); return make_float2(v[0], v[1]); }
__device__ __forceinline__ void st_st2(float *p, float2 v) { st_st(reinterpret_cast<f2 *>(p), (f2){v.x, v.y}); }

__device__ __forceinline__ float h_round(float v) { return (float)(_Float16)v; }
// the gate's input was rounded to fp16 one line earlier: the fast exp / reciprocal (~1 ulp) are exact
// enough, and the IEEE expf + division sequence was a quarter of this kernel's time
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// acc[mt][nt] += Xs[64 x 384] * W^T for this wave's 48 columns; NB weight matrices share the A reads
// SWAP: operands exchanged -> the accumulators hold the TRANSPOSED tile (lane (q, j): row j of the 16-row tile,
// columns 4q..4q+3 of the 16-column tile), i.e. four consecutive output columns per lane for direct row-major stores
template <int NB, bool SWAP = false, int PF = MLP_PREFETCH_DEFAULT, int MT = 4>
__device__ __forceinline__ void mlp_gemm(const _Float16 *Xs, const _Float16 *const (&wp)[NB], int wave, int lane,
                                         f4 (&acc)[NB][MT][MNTW]) {
  const int q = lane >> 4, j = lane & 15;
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < MNTW; nt++) acc[b][mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
#ifndef MLP_UNROLL
#define MLP_UNROLL 2
#endif
  auto wfrag = [&](int b, int ks, int nt) {
    return *reinterpret_cast<const h8 *>(wp[b] + (((size_t)ks * (MD / 16) + wave * MNTW + nt) * 64 + lane) * 8);
  };
  auto mma = [&](int ks, h8 (&bw)[NB][MNTW]) {
    h8 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) a[mt] = *reinterpret_cast<const h8 *>(Xs + (mt * 16 + j) * MXS + ks * 32 + 8 * q);
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < MNTW; nt++)
          acc[b][mt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[b][nt], a[mt], acc[b][mt][nt], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], bw[b][nt], acc[b][mt][nt], 0, 0, 0);
  };
  if constexpr (PF > 0) {
    // the weight fragments of K steps ks+1 .. ks+PF are in flight while the matrix cores work on step ks (a ring of
    // PF+1 fragment sets, fully unrolled so that the ring index is static); the scheduling barrier keeps each load
    // ABOVE the matrix work of the step it is issued in -- the compiler sinks it to its first use otherwise, which
    // exposes the full L2 latency on every K step
    h8 ring[PF + 1][NB][MNTW];
#pragma unroll
    for (int d = 0; d < PF; d++)
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int nt = 0; nt < MNTW; nt++) ring[d][b][nt] = wfrag(b, d, nt);
#pragma unroll
    for (int ks = 0; ks < MKS; ks++) {
      if (ks + PF < MKS) {
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
          for (int nt = 0; nt < MNTW; nt++) ring[(ks + PF) % (PF + 1)][b][nt] = wfrag(b, ks + PF, nt);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(ks, ring[ks % (PF + 1)]);
    }
  } else {
#pragma unroll MLP_UNROLL
    for (int ks = 0; ks < MKS; ks++) {
      h8 bw[NB][MNTW];
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int nt = 0; nt < MNTW; nt++) bw[b][nt] = wfrag(b, ks, nt);
      mma(ks, bw);
    }
  }
}

// acc += Xs[64 x 32 nks] * W[:, 32 w_ks0 .. 32 (w_ks0 + nks))^T for this wave's 48 columns: one K chunk of a layer whose
// K does not fit the tile (the activation chunk sits at tile columns 0 .. 32 nks)
__device__ __forceinline__ void mlp_gemm_chunk(const _Float16 *Xs, const _Float16 *wp, int wave, int lane, int nks,
                                               int w_ks0, f4 (&acc)[1][4][MNTW]) {
  const int q = lane >> 4, j = lane & 15;
#pragma unroll 2
  for (int ks = 0; ks < nks; ks++) {
    h8 a[4], bw[MNTW];
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++)
      bw[nt] = *reinterpret_cast<const h8 *>(wp + (((size_t)(w_ks0 + ks) * (MD / 16) + wave * MNTW + nt) * 64 + lane) * 8);
#pragma unroll
    for (int mt = 0; mt < 4; mt++) a[mt] = *reinterpret_cast<const h8 *>(Xs + (mt * 16 + j) * MXS + ks * 32 + 8 * q);
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int nt = 0; nt < MNTW; nt++)
        acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], bw[nt], acc[0][mt][nt], 0, 0, 0);
  }
}

// Row-wise epilogues run in a second, ROW-MAJOR pass: the accumulator layout (lane (q, j): rows 4q..4q+3
// of a 16-row tile, one column) would touch global memory in 64-byte pieces and needs cross-wave
// reductions for a LayerNorm.  Instead the per-element product of the stage is parked in LDS as fp32
// (32 rows at a time, in the dead h tile) and each wave then owns whole rows -- lane l holds channels
// 2l + 128k + {0,1}: 512-byte contiguous global accesses and a wave-shuffle LayerNorm.
#define MPR 32                  // rows per parking pass
#define MPS (MD + 4)            // parking row stride (floats)

// sum over the 64 lanes on the DPP network (6 VALU adds + one readlane; a ds_bpermute butterfly is 6 dependent
// trips through the LDS crossbar)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum64(float v) {
  v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);   // row_half_mirror: 8-lane sums
  v = dpp_add<0x140, 0xF>(v);   // row_mirror: every lane holds its row's 16-lane sum
  v = dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1, 3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// two-pass LayerNorm of a row held as v[3][2] per lane (same arithmetic as csrc/update.hip)
__device__ __forceinline__ void row_ln(float (&v)[3][2], const float *__restrict__ w, const float *__restrict__ b,
                                       float eps, int lane) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) s += v[k][0] + v[k][1];
  const float mean = wave_sum64(s) * (1.0f / MD);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float a = v[k][0] - mean, c = v[k][1] - mean;
    q += a * a + c * c;
  }
  const float rstd = 1.0f / sqrtf(wave_sum64(q) * (1.0f / MD) + eps);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int c = 2 * lane + 128 * k;
    v[k][0] = (v[k][0] - mean) * rstd * w[c] + b[c];
    v[k][1] = (v[k][1] - mean) * rstd * w[c + 1] + b[c + 1];
  }
}

// park the two 16-row tiles (2 half, 2 half + 1) of a value in accumulator layout
__device__ __forceinline__ void park_half(float *P, const float (&val)[4][MNTW][4], int half, int col0, int q, int j) {
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) P[(m * 16 + 4 * q + r) * MPS + col0 + nt * 16 + j] = val[2 * half + m][nt][r];
}

struct GruParams {
  const float *x32;            // [E][384] fp32: the LayerNorm'ed residual stream entering gru[1]
  const _Float16 *wp[6];       // packed weights: g1_gate, g1_r1, g1_r2, g2_gate, g2_r1, g2_r2
  const float *bias[6];        // biases (fp16-rounded values as fp32)
  const float *ln_w, *ln_b;    // gru[2] LayerNorm
  float eps;
  float *out32;                // [E][384] fp32 result of gru[3]
  _Float16 *relu_t;            // [E][384] relu(result) in fp16 (input of the heads)
  // optional prologue: x = LayerNorm_pre(x32 + add_t[add_idx]) -- the last SoftAgg's expand-and-add and gru[0]
  const _Float16 *add_t;       // [groups][384] fp16 or NULL
  const int32_t *add_idx;      // [E]
  uint32_t *gate_flag;         // optional: workgroup 0 stores gate_seq here when it starts (ramp_track.gate_flag)
  uint32_t gate_seq;
  const _Float16 *add0_t;      // optional: a FIRST expand-and-add (x32 + add0_t[add0_idx]) + add_t[add_idx] -- the second-last
  const int32_t *add0_idx;     // SoftAgg's, when the launch that consumed it did not write the sum back
  const float *pre_w, *pre_b;  // gru[0] LayerNorm
  float pre_eps;
  int E;
  const int32_t *dyn;          // optional device-side sizes (RAMP_DYN_*): E is then the launch bound
  // optional epilogue: the two heads and target / weight (ramp/net.py:87-90, ramp/Ramp_vo.py:291-297) from the
  // result tile while it is in registers -- relu_t is then not written (may be NULL)
  const _Float16 *heads_w;     // [4][384] fp16: d.weight rows 0..1, w.weight rows 0..1
  const float *heads_b;        // [4]
  const float *coords;         // [E][2][PP]
  float *target, *weight;      // [E][2]
  int PP, ctr;
  float wd, ht;
};

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// Two-pass LayerNorm of the workgroup's 64 x 384 tile held in TRANSPOSED accumulator layout (lane (q, j) of wave w:
// rows 16 mt + j, columns 48 w + 16 nt + 4 q + {0..3}): per-wave partial sums meet in an LDS table [64 rows][8 waves],
// every lane then adds the eight partials of its rows in wave order.  Same arithmetic as row_ln (mean, then the
// variance of the deviations); the order of the additions differs.
template <int MT>
__device__ __forceinline__ void tile_ln(f4 (&v)[MT][MNTW], const float *__restrict__ w, const float *__restrict__ b,
                                        float eps, float *T1, float *T2, int wave, int q, int j, int col0) {
  float mean[MT], rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) s += (v[mt][nt][0] + v[mt][nt][1]) + (v[mt][nt][2] + v[mt][nt][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) T1[(mt * 16 + j) * MWAVES + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const f4 a = *reinterpret_cast<const f4 *>(T1 + (mt * 16 + j) * MWAVES), c = *reinterpret_cast<const f4 *>(T1 + (mt * 16 + j) * MWAVES + 4);
    mean[mt] = ((((((a[0] + a[1]) + a[2]) + a[3]) + c[0]) + c[1]) + c[2] + c[3]) * (1.0f / MD);
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float d = v[mt][nt][i] - mean[mt];
        s += d * d;
      }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) T2[(mt * 16 + j) * MWAVES + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const f4 a = *reinterpret_cast<const f4 *>(T2 + (mt * 16 + j) * MWAVES), c = *reinterpret_cast<const f4 *>(T2 + (mt * 16 + j) * MWAVES + 4);
    rstd[mt] = 1.0f / sqrtf(((((((a[0] + a[1]) + a[2]) + a[3]) + c[0]) + c[1]) + c[2] + c[3]) * (1.0f / MD) + eps);
  }
#pragma unroll
  for (int nt = 0; nt < MNTW; nt++) {
    const f4 wv = *reinterpret_cast<const f4 *>(w + col0 + nt * 16 + 4 * q), bv = *reinterpret_cast<const f4 *>(b + col0 + nt * 16 + 4 * q);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int i = 0; i < 4; i++) v[mt][nt][i] = (v[mt][nt][i] - mean[mt]) * rstd[mt] * wv[i] + bv[i];
  }
}

// The same reduction for kernels on a 128-register budget (two workgroups per CU): branch-free -- the four quarter-lanes of
// a row store the same partial sum -- so that the scheduling barriers bound what is in flight per row tile (the
// straight-line version above has every table read of the five row tiles live at once: 66 spilled registers).
// Identical arithmetic and order.
template <int MT>
__device__ __forceinline__ void tile_ln_lean(f4 (&v)[MT][MNTW], const float *__restrict__ w, const float *__restrict__ b,
                                             float eps, float *T1, float *T2, int wave, int q, int j, int col0) {
  float mean[MT], rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) s += (v[mt][nt][0] + v[mt][nt][1]) + (v[mt][nt][2] + v[mt][nt][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    T1[(mt * 16 + j) * MWAVES + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    __builtin_amdgcn_sched_barrier(0);
    const f4 a = *reinterpret_cast<const f4 *>(T1 + (mt * 16 + j) * MWAVES), c = *reinterpret_cast<const f4 *>(T1 + (mt * 16 + j) * MWAVES + 4);
    mean[mt] = ((((((a[0] + a[1]) + a[2]) + a[3]) + c[0]) + c[1]) + c[2] + c[3]) * (1.0f / MD);
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float d = v[mt][nt][i] - mean[mt];
        s += d * d;
      }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    T2[(mt * 16 + j) * MWAVES + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    __builtin_amdgcn_sched_barrier(0);
    const f4 a = *reinterpret_cast<const f4 *>(T2 + (mt * 16 + j) * MWAVES), c = *reinterpret_cast<const f4 *>(T2 + (mt * 16 + j) * MWAVES + 4);
    rstd[mt] = 1.0f / sqrtf(((((((a[0] + a[1]) + a[2]) + a[3]) + c[0]) + c[1]) + c[2] + c[3]) * (1.0f / MD) + eps);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int nt = 0; nt < MNTW; nt++) {
    const f4 wv = *reinterpret_cast<const f4 *>(w + col0 + nt * 16 + 4 * q), bv = *reinterpret_cast<const f4 *>(b + col0 + nt * 16 + 4 * q);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int i = 0; i < 4; i++) v[mt][nt][i] = (v[mt][nt][i] - mean[mt]) * rstd[mt] * wv[i] + bv[i];
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef GRU_TRACE
__device__ long long g_gru_trace[32 * 4096];
#define GT(k) do { if (threadIdx.x == 0) g_gru_trace[blockIdx.x * 32 + (k)] = wall_clock64(); } while (0)
#else
#define GT(k)
#endif
// gru = LayerNorm, GatedResidual, LayerNorm, GatedResidual (ramp/net.py:49-54, ramp/blocks.py:15-31) for 64 rows.
//   * Everything row-wise happens in the accumulators' own layout.  The MFMA operands are exchanged (weights as A), so
//     lane (q, j) holds row j of a 16-row tile and FOUR CONSECUTIVE columns: the fp32 residual stream lives in 48
//     registers of that layout for the whole kernel, is loaded and stored as 16-byte pieces, and the fp16 copies the
//     next layer multiplies go to LDS as 8-byte pieces.  (An earlier version parked every product in LDS as fp32 and
//     gave whole rows to single waves for the LayerNorm: 2 x 2.5 us per stage of a 46 us workgroup.)
//   * Three K loops of ONE weight matrix each per stage: 48 accumulator registers instead of 96 leave room for a ring
//     of weight fragments GRU_PF K steps deep (the loads of a step are otherwise exposed with their full L2 latency;
//     two waves per SIMD cannot cover it).  The gate does not stay in registers across the other two loops: its
//     sigmoid (a half tensor under the reference's autocast) waits in the third LDS tile, every lane reading back
//     exactly the elements it wrote.
template <int MT>
__global__ void __launch_bounds__(64 * MWAVES) upd_gru_kernel(const GruParams p) {
  constexpr int ROWS = 16 * MT;                 // rows per workgroup: 64, or 80 when that saves a round of workgroups (one per CU:
                                                // 40k factors are 625 tiles of 64 rows = 2.44 rounds of 256, or 500 of 80 = 1.95)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);            // [64][MXS]
  _Float16 *Hs = Xs;                                                // the hidden tile takes x's place (x is dead after the second
                                                                    // K loop): 104 KB instead of 154, so that a workgroup of the
                                                                    // front end's LSTM launch (22 KB) fits on the CU beside this one
  _Float16 *Gs = Xs + ROWS * MXS;                                    // [64][MXS]: sigmoid(gate)
  float *T1 = reinterpret_cast<float *>(Gs + ROWS * MXS), *T2 = T1 + ROWS * MWAVES;   // LayerNorm partials
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * ROWS;
  // "the next frame's front end may start": a plain store the other stream's sleeping wave looks for (timing only, no
  // data rides on it)
  if (p.gate_flag && blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(p.gate_flag, p.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;                      // (workgroup-uniform: only with device-side sizes)
  const int col0 = wave * (16 * MNTW);
  const int cq = col0 + 4 * q;                  // this lane's first column in n-tile 0

  GT(0);
  // ---- the residual stream, fp32, in registers (rows past E: clamped loads, no stores)
  f4 res[MT][MNTW];
  size_t roff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int row = row0 + mt * 16 + j;
    roff[mt] = (size_t)(row < pE ? row : pE - 1) * MD;
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) res[mt][nt] = ld_st(reinterpret_cast<const f4 *>(p.x32 + roff[mt] + cq + nt * 16));
  }
  if (p.add0_t) {                               // uniform
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const _Float16 *a = p.add0_t + (size_t)p.add0_idx[roff[mt] / MD] * MD;
#pragma unroll
      for (int nt = 0; nt < MNTW; nt++) {
        const h4 v = *reinterpret_cast<const h4 *>(a + cq + nt * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) res[mt][nt][i] += (float)v[i];
      }
    }
  }
  if (p.add_t) {                                // uniform
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const _Float16 *a = p.add_t + (size_t)p.add_idx[roff[mt] / MD] * MD;
#pragma unroll
      for (int nt = 0; nt < MNTW; nt++) {
        const h4 v = *reinterpret_cast<const h4 *>(a + cq + nt * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) res[mt][nt][i] += (float)v[i];
      }
    }
    tile_ln<MT>(res, p.pre_w, p.pre_b, p.pre_eps, T1, T2, wave, q, j, col0);
  }
  auto to_lds = [&](_Float16 *tile, const f4 (&v)[MT][MNTW]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < MNTW; nt++)
        *reinterpret_cast<h4 *>(tile + (mt * 16 + j) * MXS + cq + nt * 16) =
            (h4){(_Float16)v[mt][nt][0], (_Float16)v[mt][nt][1], (_Float16)v[mt][nt][2], (_Float16)v[mt][nt][3]};
  };
  to_lds(Xs, res);
  __syncthreads();
  GT(1);

#pragma unroll 1
  for (int stage = 0; stage < 2; stage++) {
    const int wb = 3 * stage;
    f4 acc[1][MT][MNTW];
    {
      const _Float16 *const w1[1] = {p.wp[wb + 0]};
      mlp_gemm<1, true, GRU_PF, MT>(Xs, w1, wave, lane, acc);
    }
    GT(2 + 8 * stage);
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) {
      const f4 bg = *reinterpret_cast<const f4 *>(p.bias[wb + 0] + cq + nt * 16);
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[0][mt][nt][i] = sigm(h_round(acc[0][mt][nt][i] + bg[i]));
    }
    to_lds(Gs, acc[0]);
    GT(3 + 8 * stage);
    {
      const _Float16 *const w1[1] = {p.wp[wb + 1]};
      mlp_gemm<1, true, GRU_PF, MT>(Xs, w1, wave, lane, acc);
    }
    GT(4 + 8 * stage);
    // h = relu(L1 x + b1) -> Hs (fp16)
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) {
      const f4 b1 = *reinterpret_cast<const f4 *>(p.bias[wb + 1] + cq + nt * 16);
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[0][mt][nt][i] = fmaxf(acc[0][mt][nt][i] + b1[i], 0.f);
    }
    __syncthreads();                            // every wave is past its reads of x: h takes its place
    to_lds(Hs, acc[0]);
    __syncthreads();
    GT(5 + 8 * stage);
    {
      const _Float16 *const w1[1] = {p.wp[wb + 2]};
      mlp_gemm<1, true, GRU_PF, MT>(Hs, w1, wave, lane, acc);
    }
    GT(6 + 8 * stage);
    // residual += sigmoid(gate) * r     (gate and r are half tensors under the reference's autocast)
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) {
      const f4 b2 = *reinterpret_cast<const f4 *>(p.bias[wb + 2] + cq + nt * 16);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const h4 g = *reinterpret_cast<const h4 *>(Gs + (mt * 16 + j) * MXS + cq + nt * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) res[mt][nt][i] += (float)g[i] * h_round(acc[0][mt][nt][i] + b2[i]);
      }
    }
    GT(7 + 8 * stage);
    if (stage == 0) {
      // gru[2]: LayerNorm -> the next residual (registers) and its fp16 copy (every wave is past its reads of x:
      // they ended before the barrier that published h)
      tile_ln<MT>(res, p.ln_w, p.ln_b, p.eps, T1, T2, wave, q, j, col0);
      to_lds(Xs, res);
      __syncthreads();
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        if (row0 + mt * 16 + j >= pE) continue;
#pragma unroll
        for (int nt = 0; nt < MNTW; nt++) {
          const f4 v = res[mt][nt];
          st_st(reinterpret_cast<f4 *>(p.out32 + roff[mt] + cq + nt * 16), v);
          if (p.relu_t)
            *reinterpret_cast<h4 *>(p.relu_t + roff[mt] + cq + nt * 16) =
                (h4){(_Float16)fmaxf(v[0], 0.f), (_Float16)fmaxf(v[1], 0.f), (_Float16)fmaxf(v[2], 0.f), (_Float16)fmaxf(v[3], 0.f)};
        }
      }
      if (p.heads_w) {                            // (uniform)
        // d / w heads: 4 dot products of relu(result) (a half tensor) with the head rows, per row of the tile.  A lane
        // sums its 12 columns, the four quarter-lanes of a row meet over two shuffles, the eight waves over an LDS table
        // in the x / h tile (behind a barrier: the last K loop read it); fixed order throughout.
        float part[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int c = 0; c < 4; c++) part[mt][c] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int nt = 0; nt < MNTW; nt++) {
            const h4 wv = *reinterpret_cast<const h4 *>(p.heads_w + c * MD + cq + nt * 16);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
              for (int i = 0; i < 4; i++) part[mt][c] += h_round(fmaxf(res[mt][nt][i], 0.f)) * (float)wv[i];
          }
        float *HT = reinterpret_cast<float *>(Xs);              // [64 rows][8 waves][4]
        __syncthreads();                                        // every wave is past its reads of h (the x / h tile)
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
            *reinterpret_cast<f4 *>(HT + ((mt * 16 + j) * MWAVES + wave) * 4) = (f4){part[mt][0], part[mt][1], part[mt][2], part[mt][3]};
        }
        __syncthreads();
        const int e = row0 + tid;
        if (tid < ROWS && e < pE) {
          float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int w = 0; w < MWAVES; w++) {
            const f4 v = *reinterpret_cast<const f4 *>(HT + (tid * MWAVES + w) * 4);
#pragma unroll
            for (int c = 0; c < 4; c++) o[c] += v[c];
          }
#pragma unroll
          for (int c = 0; c < 4; c++) o[c] = h_round(o[c] + p.heads_b[c]);         // the Linear output is a half tensor
          const float wx = h_round(1.0f / (1.0f + expf(-o[2])));
          const float wy = h_round(1.0f / (1.0f + expf(-o[3])));
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
    GT(8 + 8 * stage);
  }
}

// ------------------------------------------------------------------ c1 / c2
// net[e] += Lb(relu(La(mask * net[idx[e]])))   (ramp/net.py:77-82: the temporal-neighbour MLPs; idx = -1
// marks a missing neighbour).  The gathered rows are staged straight into the LDS tile; the result is
// written to a SECOND state buffer (other workgroups still gather from the input one).
struct NbrParams {
  const float *net_in;         // [E][384] fp32
  const int64_t *idx;          // [E] neighbour row or -1
  const _Float16 *wa, *wb;     // packed weights of the two Linear layers
  const float *ba, *bb;        // biases (fp16-rounded values as fp32)
  float *net_out;              // [E][384] fp32
  _Float16 *out_t;             // optional [E][384] fp16 copy of net_out
  int E;
  const int32_t *dyn;          // optional device-side sizes (RAMP_DYN_*): E is then the launch bound
};

// waves_per_eu 6: 80 VGPRs -> three workgroups per CU (3 x 50 KB LDS): all ~625 workgroups of an update are
// resident at once (768 slots) instead of 1.2 rounds of 512
__global__ void __launch_bounds__(64 * MWAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) upd_nbr_kernel(const NbrParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // ONE 50 KB tile: gathered input, then (after a barrier) the hidden layer, then the fp32 parking tile --
  // several workgroups per CU, so one's gather / row pass overlaps the others' matrix work
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *Hs = Xs;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * MBM;
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;                      // (workgroup-uniform: only with device-side sizes)
  const int col0 = wave * (16 * MNTW);
  for (int i = tid; i < MBM * (MD / 4); i += 64 * MWAVES) {
    const int r = i / (MD / 4), c4 = i - r * (MD / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < pE) {
      const long src = p.idx[row0 + r];
      if (src >= 0) v = *reinterpret_cast<const float4 *>(p.net_in + (size_t)src * MD + 4 * c4);
    }
    _Float16 *d = Xs + r * MXS + 4 * c4;
    d[0] = (_Float16)v.x; d[1] = (_Float16)v.y; d[2] = (_Float16)v.z; d[3] = (_Float16)v.w;
  }
  __syncthreads();
  f4 acc[1][4][MNTW];
  {
    const _Float16 *const w1[1] = {p.wa};
    mlp_gemm<1>(Xs, w1, wave, lane, acc);
  }
  __syncthreads();                                       // every wave is past its reads of x
#pragma unroll
  for (int nt = 0; nt < MNTW; nt++) {
    const float b1 = p.ba[col0 + nt * 16 + j];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        Hs[(mt * 16 + 4 * q + r) * MXS + col0 + nt * 16 + j] = (_Float16)fmaxf(acc[0][mt][nt][r] + b1, 0.f);
  }
  __syncthreads();
  {
    const _Float16 *const w1[1] = {p.wb};
    mlp_gemm<1>(Hs, w1, wave, lane, acc);
  }
  float y[4][MNTW][4];
#pragma unroll
  for (int nt = 0; nt < MNTW; nt++) {
    const float b2 = p.bb[col0 + nt * 16 + j];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) y[mt][nt][r] = h_round(acc[0][mt][nt][r] + b2);
  }
  __syncthreads();                                       // every wave is past its reads of h
  float *P = reinterpret_cast<float *>(Hs);
#pragma unroll
  for (int half = 0; half < 2; half++) {
    park_half(P, y, half, col0, q, j);
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < MPR / MWAVES; rr++) {
      const int rl = wave * (MPR / MWAVES) + rr;
      const int row = row0 + half * MPR + rl;
      if (row >= pE) continue;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int c = 2 * lane + 128 * k;
        const float2 x = *reinterpret_cast<const float2 *>(p.net_in + (size_t)row * MD + c);
        const float2 pv = *reinterpret_cast<const float2 *>(P + rl * MPS + c);
        const float v0 = x.x + pv.x, v1 = x.y + pv.y;
        *reinterpret_cast<float2 *>(p.net_out + (size_t)row * MD + c) = make_float2(v0, v1);
        if (p.out_t) *reinterpret_cast<h2 *>(p.out_t + (size_t)row * MD + c) = (h2){(_Float16)v0, (_Float16)v1};
      }
    }
    __syncthreads();
  }
}

// Tail of the correlation MLP and the operator's first LayerNorm (ramp/net.py:57-62, 71-74):
//   c = Linear3(relu(LayerNorm(Linear2(c1))));  net = LayerNorm(net_prev + inp + c)
// c1 = relu(Linear1(corr)) comes from the library GEMM (K = 896).  One 50 KB LDS tile (input, hidden, fp32
// parking) -> two workgroups per CU.  Linear outputs are rounded to fp16 where autocast makes them half tensors.
struct CorrTailParams {
  // FULL variant: the first Linear (+ReLU) of the MLP on the correlation rows as well
  const _Float16 *corr;        // [E][corr_k] fp16 (corr_k a multiple of 32, e.g. 896 = 882 + zero padding)
  const _Float16 *w1;          // packed [corr_k/32][24][64][8]
  const float *b1;
  int corr_k;
  const _Float16 *c1;          // [E][384] fp16 (tail-only variant: relu(Linear1(corr)) from a library GEMM)
  const _Float16 *w2, *w3;     // packed weights of corr[2], corr[5]
  const float *b2, *b3;        // biases (fp16-rounded values as fp32)
  const float *ln_w, *ln_b;    // corr[3] LayerNorm
  float ln_eps;
  const float *net;            // [*][384] fp32 previous hidden state or NULL (zeros)
  const int64_t *net_map;      // [E] row of `net` per edge (-1: zero row) or NULL (identity)
  const _Float16 *inp;         // context table [*][384] fp16
  const int64_t *inp_idx;      // [E] row of `inp` (taken modulo inp_mod when inp_mod > 0) or NULL (identity)
  long inp_mod;
  const float *norm_w, *norm_b;
  float norm_eps;
  float *net_out;              // [E][384] fp32
  int E;
  const int32_t *dyn;          // optional device-side sizes (RAMP_DYN_*): E is then the launch bound
};

#ifdef GRU_TRACE
#define CT(k) do { if (threadIdx.x == 0) g_gru_trace[blockIdx.x * 32 + (k)] = wall_clock64(); } while (0)
#else
#define CT(k)
#endif
template <bool FULL>
__global__ void __launch_bounds__(64 * MWAVES) __attribute__((amdgpu_waves_per_eu(6, 6)))
    upd_corr_tail_kernel(const CorrTailParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * MBM;
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;                      // (workgroup-uniform: only with device-side sizes)
  const int col0 = wave * (16 * MNTW);
  f4 acc[1][4][MNTW];
  CT(0);
  // the rows of the previous state and of the context table this tile will add in its last pass: their indices are
  // fetched now (the last pass used to start with two dependent round trips per row, eight rows per wave in sequence)
  __shared__ long s_ra[MBM], s_rb[MBM];
  if (tid < MBM) {
    const int row = row0 + tid;
    long ra = -1, rb = 0;
    if (row < pE) {
      ra = p.net ? (p.net_map ? p.net_map[row] : (long)row) : -1;
      rb = p.inp_idx ? p.inp_idx[row] : (long)row;
      if (p.inp_mod > 0) rb %= p.inp_mod;
    }
    s_ra[tid] = ra; s_rb[tid] = rb;
  }
  if (FULL) {
    // Linear1 over K = corr_k in chunks of <= 12 K steps (the tile is 384 wide), then relu(+b1) becomes the tile
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int nt = 0; nt < MNTW; nt++) acc[0][mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
    const int nks_total = p.corr_k / 32;
    for (int ks0 = 0; ks0 < nks_total; ks0 += MKS) {
      const int nks = min(MKS, nks_total - ks0);
      const int v8 = nks * 4;                                  // 16-byte vectors per row of this chunk
#ifdef CTAIL_SKIP_STAGE                                        // (diagnostic: what the staging of Linear1's operand costs)
      if (ks0 == 0)
#endif
      for (int i = tid; i < MBM * v8; i += 64 * MWAVES) {
        const int r = i / v8, c8 = i - r * v8;
        h8 v = (h8){0, 0, 0, 0, 0, 0, 0, 0};
        if (row0 + r < pE) v = ld_dead(reinterpret_cast<const h8 *>(p.corr + (size_t)(row0 + r) * p.corr_k + ks0 * 32 + 8 * c8));
        *reinterpret_cast<h8 *>(Xs + r * MXS + 8 * c8) = v;
      }
      __syncthreads();
#ifndef CTAIL_SKIP_L1                                          // (diagnostic: Linear1's products)
      mlp_gemm_chunk(Xs, p.w1, wave, lane, nks, ks0, acc);
#endif
      __syncthreads();                                         // before the tile is overwritten
    }
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) {
      const float b = p.b1[col0 + nt * 16 + j];
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          Xs[(mt * 16 + 4 * q + r) * MXS + col0 + nt * 16 + j] = (_Float16)fmaxf(acc[0][mt][nt][r] + b, 0.f);
    }
  } else {
    for (int i = tid; i < MBM * (MD / 8); i += 64 * MWAVES) {
      const int r = i / (MD / 8), c8 = i - r * (MD / 8);
      h8 v = (h8){0, 0, 0, 0, 0, 0, 0, 0};
      if (row0 + r < pE) v = *reinterpret_cast<const h8 *>(p.c1 + (size_t)(row0 + r) * MD + 8 * c8);
      *reinterpret_cast<h8 *>(Xs + r * MXS + 8 * c8) = v;
    }
  }
  __syncthreads();
  CT(1);
  {
    const _Float16 *const w1[1] = {p.w2};
    mlp_gemm<1>(Xs, w1, wave, lane, acc);
  }
  CT(2);
  float y[4][MNTW][4];
#pragma unroll
  for (int nt = 0; nt < MNTW; nt++) {
    const float b = p.b2[col0 + nt * 16 + j];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) y[mt][nt][r] = h_round(acc[0][mt][nt][r] + b);
  }
  __syncthreads();                                       // every wave is past its reads of the input tile
  float *P = reinterpret_cast<float *>(Xs);
  // row pass 1: LayerNorm + ReLU; the 8 rows of this wave wait in registers until both halves are through
  // (the parking tile covers the whole LDS tile), then become the hidden tile
  h2 hrow[2][MPR / MWAVES][3];
#pragma unroll
  for (int half = 0; half < 2; half++) {
    park_half(P, y, half, col0, q, j);
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < MPR / MWAVES; rr++) {
      const int rl = wave * (MPR / MWAVES) + rr;
      float v[3][2];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float2 pv = *reinterpret_cast<const float2 *>(P + rl * MPS + 2 * lane + 128 * k);
        v[k][0] = pv.x; v[k][1] = pv.y;
      }
      row_ln(v, p.ln_w, p.ln_b, p.ln_eps, lane);
#pragma unroll
      for (int k = 0; k < 3; k++) hrow[half][rr][k] = (h2){(_Float16)fmaxf(v[k][0], 0.f), (_Float16)fmaxf(v[k][1], 0.f)};
    }
    __syncthreads();
  }
#pragma unroll
  for (int half = 0; half < 2; half++)
#pragma unroll
    for (int rr = 0; rr < MPR / MWAVES; rr++) {
      const int rt = half * MPR + wave * (MPR / MWAVES) + rr;
#pragma unroll
      for (int k = 0; k < 3; k++) *reinterpret_cast<h2 *>(Xs + rt * MXS + 2 * lane + 128 * k) = hrow[half][rr][k];
    }
  __syncthreads();
  CT(3);
  {
    const _Float16 *const w1[1] = {p.w3};
    mlp_gemm<1>(Xs, w1, wave, lane, acc);
  }
#pragma unroll
  for (int nt = 0; nt < MNTW; nt++) {
    const float b = p.b3[col0 + nt * 16 + j];
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) y[mt][nt][r] = h_round(acc[0][mt][nt][r] + b);
  }
  CT(4);
  __syncthreads();
  // row pass 2: net_prev + inp + c (in that order), LayerNorm, fp32 store
#pragma unroll
  for (int half = 0; half < 2; half++) {
    park_half(P, y, half, col0, q, j);
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < MPR / MWAVES; rr++) {
      const int rl = wave * (MPR / MWAVES) + rr;
      const int row = row0 + half * MPR + rl;
      if (row >= pE) continue;                          // wave-uniform
      float v[3][2];
#pragma unroll
      for (int k = 0; k < 3; k++) { v[k][0] = 0.f; v[k][1] = 0.f; }
      const long ra = s_ra[half * MPR + rl], rb = s_rb[half * MPR + rl];
      if (ra >= 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const f2 a_ = ld_dead(reinterpret_cast<const f2 *>(p.net + (size_t)ra * MD + 2 * lane + 128 * k));
          const float2 a = make_float2(a_[0], a_[1]);
          v[k][0] = a.x; v[k][1] = a.y;
        }
      }
      {
        const _Float16 *b = p.inp + (size_t)rb * MD;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const h2 t = *reinterpret_cast<const h2 *>(b + 2 * lane + 128 * k);
          v[k][0] += (float)t[0]; v[k][1] += (float)t[1];
        }
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float2 pv = *reinterpret_cast<const float2 *>(P + rl * MPS + 2 * lane + 128 * k);
        v[k][0] += pv.x; v[k][1] += pv.y;
      }
      row_ln(v, p.norm_w, p.norm_b, p.norm_eps, lane);
#pragma unroll
      for (int k = 0; k < 3; k++)
        st_st2(p.net_out + (size_t)row * MD + 2 * lane + 128 * k, make_float2(v[k][0], v[k][1]));
    }
    __syncthreads();
  }
  CT(5);
}

// SoftAgg front half (ramp/blocks.py:42-46): fg[e] = [f(x_e) | g(x_e)] for x = x32 (+ add_t[add_idx], the previous
// SoftAgg's expand-and-add, written back to x32_out) -- the row pass that formed x and the [E,384]x[384,768] GEMM in
// one launch.  Transposed accumulators: every lane stores four consecutive fp16 outputs, no parking pass, the
// input tile stays valid for the second matrix.  One 50 KB tile, three workgroups per CU.
struct FgParams {
  const float *x32;            // [E][384] fp32
  const _Float16 *add_t;       // optional [groups][384] fp16
  const int32_t *add_idx;      // [E]
  float *x32_out;              // optional [E][384] fp32 (may be x32): x after the add
  const _Float16 *wf, *wg;     // packed weights of f and g
  const float *bf, *bg;        // biases (fp16-rounded values as fp32)
  _Float16 *fg;                // [E][768] fp16
  int E;
  const int32_t *dyn;          // optional device-side sizes (RAMP_DYN_*): E is then the launch bound
};

__global__ void __launch_bounds__(64 * MWAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) upd_fg_kernel(const FgParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * MBM;
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;                      // (workgroup-uniform: only with device-side sizes)
  const int col0 = wave * (16 * MNTW);
#pragma unroll
  for (int rr = 0; rr < MBM / MWAVES; rr++) {            // 8 whole rows per wave
    const int rt = wave * (MBM / MWAVES) + rr;
    const int row = row0 + rt;
    float v[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++) { v[k][0] = 0.f; v[k][1] = 0.f; }
    if (row < pE) {                                     // wave-uniform
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float2 a = *reinterpret_cast<const float2 *>(p.x32 + (size_t)row * MD + 2 * lane + 128 * k);
        v[k][0] = a.x; v[k][1] = a.y;
      }
      if (p.add_t) {
        const _Float16 *b = p.add_t + (size_t)p.add_idx[row] * MD;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const h2 t = *reinterpret_cast<const h2 *>(b + 2 * lane + 128 * k);
          v[k][0] += (float)t[0]; v[k][1] += (float)t[1];
        }
      }
      if (p.x32_out) {
#pragma unroll
        for (int k = 0; k < 3; k++)
          *reinterpret_cast<float2 *>(p.x32_out + (size_t)row * MD + 2 * lane + 128 * k) = make_float2(v[k][0], v[k][1]);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
      *reinterpret_cast<h2 *>(Xs + rt * MXS + 2 * lane + 128 * k) = (h2){(_Float16)v[k][0], (_Float16)v[k][1]};
  }
  __syncthreads();
  typedef _Float16 hh4 __attribute__((ext_vector_type(4)));
#pragma unroll 1
  for (int part = 0; part < 2; part++) {
    f4 acc[1][4][MNTW];
    const _Float16 *const w1[1] = {part ? p.wg : p.wf};
    mlp_gemm<1, true>(Xs, w1, wave, lane, acc);
    const float *bias = part ? p.bg : p.bf;
#pragma unroll
    for (int nt = 0; nt < MNTW; nt++) {
      const float4 b = *reinterpret_cast<const float4 *>(bias + col0 + nt * 16 + 4 * q);
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        const int row = row0 + mt * 16 + j;
        if (row < pE)
          *reinterpret_cast<hh4 *>(p.fg + (size_t)row * (2 * MD) + part * MD + col0 + nt * 16 + 4 * q) =
              (hh4){(_Float16)(acc[0][mt][nt][0] + b.x), (_Float16)(acc[0][mt][nt][1] + b.y),
                    (_Float16)(acc[0][mt][nt][2] + b.z), (_Float16)(acc[0][mt][nt][3] + b.w)};
      }
    }
  }
}

// =====================================================================================================
// Wider-tile variants of the two chains WITHOUT row statistics (c1 / c2 and SoftAgg's [f | g]): a workgroup owns
// 16 NMT rows (80 / 96 at the bench size) instead of 64, two workgroups per CU, so each streamed weight fragment
// feeds 1.25-1.5x the matrix work and one workgroup's gather / store phase overlaps the other's GEMM.
//   * TRANSPOSED accumulators (weights as the A operand): lane (q, j) holds row j of a 16-row tile and four
//     consecutive columns -- the epilogue is a 16-byte global access straight from the accumulator registers, no fp32
//     parking pass through LDS;
//   * global accesses in that layout use ONE 32-bit byte offset per row tile next to a uniform base pointer; rows
//     past E are clamped for loads and masked for stores;
//   * the weight fragments of K step ks + 1 are issued before the MFMAs of step ks (scheduling barrier: the compiler
//     sinks them to their first use otherwise).
// Measured on MI355X at E = 40,000 (tools/mb_update.py): c1 / c2 48.9 -> 43.3 us (80 rows), [f | g] 56.1 -> 46.1 us
// (96 rows).  Both chains move 184 MB of fp32 state per launch (~37-46 us at 4-5 TB/s): they are HBM bound, the matrix
// work is hidden.  The same treatment of the gru block and the correlation MLP (160 rows per workgroup, one round of
// 250 workgroups, LayerNorm statistics through an LDS table) was slower than the 64-row kernels (192 vs 162 us,
// 131 vs 90 us): one workgroup per CU cannot hide its own weight-load latency, and 240+ live accumulator registers
// spill -- the way forward there is an LDS-DMA weight FIFO with counted vmcnt waits (inline asm), not a bigger tile.
typedef _Float16 hh4 __attribute__((ext_vector_type(4)));

template <int NMT, int NTW>
__device__ __forceinline__ void big_zero(f4 (&acc)[NMT][NTW]) {
#pragma unroll
  for (int mt = 0; mt < NMT; mt++)
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) acc[mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
}

// acc[mt][nt] += (Xs[16 NMT x 32 nks] * W[:, 32 w_ks0 ..)^T)^T for this wave's 16 NTW columns (transposed
// accumulators).  The weight fragments of K step ks + 1 are ISSUED before the matrix work of step ks (the scheduling
// barrier keeps the compiler from sinking them to their first use, which exposed their full L2 latency every step);
// the A fragment of row tile mt + 1 is read from LDS while tile mt multiplies.
#ifndef BIG_PF
#define BIG_PF 1   // measured: 2 is 2 % slower on the whole operator (128 / 146 VGPRs), 1 is the round-2 kernel
#endif
template <int NMT, int NTW, bool SWAP = true>
__device__ __forceinline__ void big_gemm(const _Float16 *Xs, const _Float16 *wp, int nks, int w_ks0, int wave, int lane,
                                         f4 (&acc)[NMT][NTW]) {
  const int q = lane >> 4, j = lane & 15;
  const _Float16 *wb = wp + (((size_t)w_ks0 * (MD / 16) + wave * NTW) * 64 + lane) * 8;
  const _Float16 *xb = Xs + j * MXS + 8 * q;
  // the weight fragments of the next BIG_PF K steps are in flight while the matrix cores work on this one
  h8 ring[BIG_PF + 1][NTW];
#pragma unroll
  for (int d = 0; d < BIG_PF; d++)
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
      ring[d][nt] = *reinterpret_cast<const h8 *>(wb + ((size_t)(d < nks ? d : nks - 1) * (MD / 16) + nt) * 512);
#pragma unroll 1
  for (int ks0 = 0; ks0 < nks; ks0 += BIG_PF + 1) {
#pragma unroll
    for (int r = 0; r < BIG_PF + 1; r++) {             // static ring index
      const int ks = ks0 + r;
      if (ks >= nks) break;
      const int kn = ks + BIG_PF < nks ? ks + BIG_PF : nks - 1;
#pragma unroll
      for (int nt = 0; nt < NTW; nt++)
        ring[(r + BIG_PF) % (BIG_PF + 1)][nt] = *reinterpret_cast<const h8 *>(wb + ((size_t)kn * (MD / 16) + nt) * 512);
      h8 a = *reinterpret_cast<const h8 *>(xb + ks * 32);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < NMT; mt++) {
        h8 an = a;
        if (mt + 1 < NMT) an = *reinterpret_cast<const h8 *>(xb + (mt + 1) * 16 * MXS + ks * 32);
#pragma unroll
        for (int nt = 0; nt < NTW; nt++)
          acc[mt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[r][nt], a, acc[mt][nt], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_f16(a, ring[r][nt], acc[mt][nt], 0, 0, 0);
        a = an;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// h = [relu](acc + bias) as fp16 into the LDS tile (8-byte stores, conflict free with the +8 pad)
template <int NMT, int NTW, bool RELU>
__device__ __forceinline__ void big_store_tile(_Float16 *Xs, const f4 (&acc)[NMT][NTW], const float *__restrict__ bias,
                                               int col0, int q, int j) {
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) {
    const int c = col0 + nt * 16 + 4 * q;
    const float4 b = *reinterpret_cast<const float4 *>(bias + c);
#pragma unroll
    for (int mt = 0; mt < NMT; mt++) {
      float v0 = acc[mt][nt][0] + b.x, v1 = acc[mt][nt][1] + b.y, v2 = acc[mt][nt][2] + b.z, v3 = acc[mt][nt][3] + b.w;
      if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
      *reinterpret_cast<hh4 *>(Xs + (mt * 16 + j) * MXS + c) = (hh4){(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
    }
  }
}

__device__ __forceinline__ float4 ldg4(const float *base, unsigned off, int nt) {
  const f4 v = ld_st(reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(base) + off + nt * 64));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stg4(float *base, unsigned off, int nt, float4 v) {
  st_st(reinterpret_cast<f4 *>(reinterpret_cast<char *>(base) + off + nt * 64), (f4){v.x, v.y, v.z, v.w});
}
__device__ __forceinline__ void stg4h(_Float16 *base, unsigned off_f32, int nt, float4 v) {   // fp16 row of the same shape
  *reinterpret_cast<hh4 *>(reinterpret_cast<char *>(base) + (off_f32 >> 1) + nt * 32) =
      (hh4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
}

// byte offsets (fp32 rows of 384) of this lane's first column in every row tile + the mask of live rows
template <int NMT>
__device__ __forceinline__ unsigned big_row_offsets(unsigned (&ro)[NMT], int row0, int col0, int q, int j, int E) {
  unsigned live = 0;
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    const int row = row0 + mt * 16 + j;
    ro[mt] = (unsigned)(min(row, E - 1) * MD + col0 + 4 * q) * 4u;
    if (row < E) live |= 1u << mt;
  }
  return live;
}

// ------------------------------------------------------------------ c1 / c2 (big tile)
template <int NMT, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) upd_nbr_big_kernel(const NbrParams p) {
  constexpr int NTW = 24 / NW, ROWS = 16 * NMT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * ROWS;
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;                      // (workgroup-uniform: only with device-side sizes)
  const int col0 = wave * (16 * NTW);
  // gather: wave w stages rows w, w + NW, ... (lane l: channels 2l + 128k): 512-byte contiguous reads.  All of the
  // wave's neighbour indices first, then all of its rows: one dependent round trip per workgroup instead of one per
  // row (branch-free: a missing neighbour reads row 0 and is masked)
  constexpr int RPW = ROWS / NW;
  {
    long src[RPW];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const int row = row0 + wave + i * NW;
      src[i] = p.idx[row < pE ? row : pE - 1];
      if (row >= pE) src[i] = -1;
    }
    float2 v[RPW][3];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const float *b = p.net_in + (size_t)(src[i] >= 0 ? src[i] : 0) * MD + 2 * lane;
#pragma unroll
      for (int k = 0; k < 3; k++) v[i][k] = ld_st2(b + 128 * k);
    }
#pragma unroll
    for (int i = 0; i < RPW; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float2 a = src[i] >= 0 ? v[i][k] : make_float2(0.f, 0.f);
        *reinterpret_cast<h2 *>(Xs + (wave + i * NW) * MXS + 2 * lane + 128 * k) = (h2){(_Float16)a.x, (_Float16)a.y};
      }
  }
  unsigned ro[NMT];
  const unsigned live = big_row_offsets<NMT>(ro, row0, col0, q, j, pE);
  __syncthreads();
  f4 acc[NMT][NTW];
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.wa, MKS, 0, wave, lane, acc);
  __syncthreads();                                       // every wave is past its reads of x
  big_store_tile<NMT, NTW, true>(Xs, acc, p.ba, col0, q, j);
  __syncthreads();
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.wb, MKS, 0, wave, lane, acc);
  float4 bv[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) bv[nt] = *reinterpret_cast<const float4 *>(p.bb + col0 + nt * 16 + 4 * q);
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    float4 x[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) x[nt] = ldg4(p.net_in, ro[mt], nt);
    if (!((live >> mt) & 1)) continue;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const float4 o = make_float4(x[nt].x + h_round(acc[mt][nt][0] + bv[nt].x), x[nt].y + h_round(acc[mt][nt][1] + bv[nt].y),
                                   x[nt].z + h_round(acc[mt][nt][2] + bv[nt].z), x[nt].w + h_round(acc[mt][nt][3] + bv[nt].w));
      stg4(p.net_out, ro[mt], nt, o);
      if (p.out_t) stg4h(p.out_t, ro[mt], nt, o);
    }
  }
}

// ------------------------------------------------------------------ SoftAgg [f | g] (big tile)
template <int NMT, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) upd_fg_big_kernel(const FgParams p) {
  constexpr int NTW = 24 / NW, ROWS = 16 * NMT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * ROWS;
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;                      // (workgroup-uniform: only with device-side sizes)
  const int col0 = wave * (16 * NTW);
  // stage x = x32 (+ add_t[add_idx]): all of the wave's group indices first, then all of its rows (rows past E read
  // row E - 1 and are not written back)
  constexpr int RPW = ROWS / NW;
  {
    int ai[RPW];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const int row = row0 + wave + i * NW;
      ai[i] = p.add_t ? p.add_idx[row < pE ? row : pE - 1] : 0;
    }
    float2 v[RPW][3];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const int row = row0 + wave + i * NW;
      const float *b = p.x32 + (size_t)(row < pE ? row : pE - 1) * MD + 2 * lane;
#pragma unroll
      for (int k = 0; k < 3; k++) v[i][k] = ld_st2(b + 128 * k);
    }
    if (p.add_t) {                                       // uniform
      h2 t[RPW][3];
#pragma unroll
      for (int i = 0; i < RPW; i++) {
        const _Float16 *b = p.add_t + (size_t)ai[i] * MD + 2 * lane;
#pragma unroll
        for (int k = 0; k < 3; k++) t[i][k] = *reinterpret_cast<const h2 *>(b + 128 * k);
      }
#pragma unroll
      for (int i = 0; i < RPW; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) { v[i][k].x += (float)t[i][k][0]; v[i][k].y += (float)t[i][k][1]; }
    }
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const int r = wave + i * NW, row = row0 + r;
      const bool live_row = row < pE;                   // wave-uniform
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (live_row && p.x32_out)
          st_st2(p.x32_out + (size_t)row * MD + 2 * lane + 128 * k, v[i][k]);
        const float2 a = live_row ? v[i][k] : make_float2(0.f, 0.f);
        *reinterpret_cast<h2 *>(Xs + r * MXS + 2 * lane + 128 * k) = (h2){(_Float16)a.x, (_Float16)a.y};
      }
    }
  }
  unsigned ro[NMT];
  const unsigned live = big_row_offsets<NMT>(ro, row0, col0, q, j, pE);
  __syncthreads();
#pragma unroll 1
  for (int part = 0; part < 2; part++) {
    f4 acc[NMT][NTW];
    big_zero<NMT, NTW>(acc);
    big_gemm<NMT, NTW>(Xs, part ? p.wg : p.wf, MKS, 0, wave, lane, acc);
    const float *bias = part ? p.bg : p.bf;
    _Float16 *dst = p.fg + part * MD;                      // rows of 768 halfs = 1536 B, like an fp32 row of 384
    const unsigned cfix = (unsigned)(col0 + 4 * q) * 2u;   // ro counts the lane's first column in fp32 bytes; here halfs
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const float4 b = *reinterpret_cast<const float4 *>(bias + col0 + nt * 16 + 4 * q);
#pragma unroll
      for (int mt = 0; mt < NMT; mt++)
        if ((live >> mt) & 1)
          st_st(reinterpret_cast<hh4 *>(reinterpret_cast<char *>(dst) + (ro[mt] - cfix) + nt * 32),
                (hh4){(_Float16)(acc[mt][nt][0] + b.x), (_Float16)(acc[mt][nt][1] + b.y),
                      (_Float16)(acc[mt][nt][2] + b.z), (_Float16)(acc[mt][nt][3] + b.w)});
    }
  }
}


// ------------------------------------------------------------------ SoftAgg without the [f | g] rows
// y[group][c] = sum_e softmax_e(g(x_e)[c]) f(x_e)[c] over the factors e of a group, hy = h(y)  (ramp/blocks.py:42-47).
// upd_fg (+ upd_segment_softmax) write the [E, 768] fp16 rows of [f | g] and read them back through the grouping's
// order: 4 x 61 MB per SoftAgg, two launches.  Here a workgroup owns 80 CONSECUTIVE POSITIONS of the grouping's sorted
// factor list (`order`: groups are contiguous runs of it), gathers their state rows (1536 contiguous bytes each), and
// runs the g and the f product one after the other on the same tile:
//   * plain (not exchanged) MFMA operands: lane (q, j) holds column j of an n-tile and rows 4q..4q+3 of each of the five
//     16-row m-tiles -- and the tile rows are PERMUTED so that those 20 rows are 20 consecutive sorted positions
//     (lane block q <-> positions 20 q .. 20 q + 19 of the tile): a group is a contiguous run of a lane's registers;
//   * g (a half tensor under the reference's autocast) waits as 30 packed registers while the f product runs;
//   * one forward sweep per column over the lane's 20 rows: online softmax (running maximum, rescaled sum and weighted
//     sum -- the arithmetic of upd_segment_softmax), restarted where the group changes.  Each run leaves one FRAGMENT
//     (m, z, a)[384] in slot  group + 4 tile + q  of a table: (group, block) pairs are distinct and the sum grows along
//     the list, so slots are unique, and a group's fragments are the consecutive slots
//     group + first block .. group + last block;
//   * upd_softagg_finish merges a group's fragments in slot order (fixed order: deterministic), y = a / z -> fp16, and
//     applies h on 16 groups per workgroup (the matrix part of upd_linear_kernel).
// Per SoftAgg: 61 MB of state in, ~20 MB of fragments out and in; the two 61 MB [f | g] passes are gone.
#define SAGG_NMT 5
#define SAGG_ROWS (16 * SAGG_NMT)   // 80 positions per workgroup
#define SAGG_BLK (4 * SAGG_NMT)     // 20 positions per lane block
struct SoftAggParams {
  const float *x32;            // [E][384] fp32
  const _Float16 *add_t;       // optional [groups'][384] fp16: x = x32 + add_t[add_idx] (the previous SoftAgg's expand-and-add)
  const int32_t *add_idx;      // [E]
  const int32_t *order;        // [E] sorted position -> factor
  const int32_t *gid;          // [E] factor -> group
  const _Float16 *wf, *wg;     // packed weights of f and g
  const float *bf, *bg;        // biases (fp16-rounded values as fp32)
  float *frag;                 // [slots][3][384] fp32: (m, z, a) per run
  int E;
  const int32_t *dyn;          // optional device-side sizes (RAMP_DYN_*): E is then the launch bound
  uint32_t *gate_flag;         // optional: workgroup 0 stores gate_seq here when it starts (ramp_track.gate_flag)
  uint32_t gate_seq;
};

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) upd_softagg_kernel(const SoftAggParams p) {
  constexpr int NMT = SAGG_NMT, NTW = 3, NW = 8, ROWS = SAGG_ROWS, BLK = SAGG_BLK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  int *s_gid = reinterpret_cast<int *>(Xs + ROWS * MXS);            // [ROWS] group of sorted position p0 + t (-1 past E)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int p0 = blockIdx.x * ROWS;
  if (p.gate_flag && blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(p.gate_flag, p.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (p0 >= pE) return;                        // (workgroup-uniform)
  const int col0 = wave * (16 * NTW);
  // tile row rho = 16 mt + 4 qq + r holds sorted position p0 + BLK qq + 4 mt + r
  constexpr int RPW = ROWS / NW;
  {
    int fe[RPW], ai[RPW];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const int rho = wave + i * NW;
      const int pos = p0 + BLK * ((rho & 15) >> 2) + 4 * (rho >> 4) + (rho & 3);
      fe[i] = pos < pE ? p.order[pos] : -1;
    }
#pragma unroll
    for (int i = 0; i < RPW; i++) ai[i] = (p.add_t && fe[i] >= 0) ? p.add_idx[fe[i]] : 0;
    float2 v[RPW][3];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const float *b = p.x32 + (size_t)(fe[i] >= 0 ? fe[i] : 0) * MD + 2 * lane;
#pragma unroll
      for (int k = 0; k < 3; k++) v[i][k] = *reinterpret_cast<const float2 *>(b + 128 * k);
    }
    if (p.add_t) {                                       // uniform
      h2 t[RPW][3];
#pragma unroll
      for (int i = 0; i < RPW; i++) {
        const _Float16 *b = p.add_t + (size_t)ai[i] * MD + 2 * lane;
#pragma unroll
        for (int k = 0; k < 3; k++) t[i][k] = *reinterpret_cast<const h2 *>(b + 128 * k);
      }
#pragma unroll
      for (int i = 0; i < RPW; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) { v[i][k].x += (float)t[i][k][0]; v[i][k].y += (float)t[i][k][1]; }
    }
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const int rho = wave + i * NW;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float2 a = fe[i] >= 0 ? v[i][k] : make_float2(0.f, 0.f);
        *reinterpret_cast<h2 *>(Xs + rho * MXS + 2 * lane + 128 * k) = (h2){(_Float16)a.x, (_Float16)a.y};
      }
    }
  }
  if (tid < ROWS) s_gid[tid] = (p0 + tid < pE) ? p.gid[p.order[p0 + tid]] : -1;
  __syncthreads();
  // ---- g = x Wg' + bg (half), parked as packed pairs
  f4 acc[NMT][NTW];
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW, false>(Xs, p.wg, MKS, 0, wave, lane, acc);
  h2 gp[NMT][NTW][2];
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) {
    const float b = p.bg[col0 + nt * 16 + j];
#pragma unroll
    for (int mt = 0; mt < NMT; mt++) {
      gp[mt][nt][0] = (h2){(_Float16)(acc[mt][nt][0] + b), (_Float16)(acc[mt][nt][1] + b)};
      gp[mt][nt][1] = (h2){(_Float16)(acc[mt][nt][2] + b), (_Float16)(acc[mt][nt][3] + b)};
    }
  }
  // ---- f = x Wf' + bf on the same tile
  big_zero<NMT, NTW>(acc);
#ifndef SAGG_SKIP_F                                  // (diagnostic builds: tools/mb_softagg.py)
  big_gemm<NMT, NTW, false>(Xs, p.wf, MKS, 0, wave, lane, acc);
#endif
#ifdef SAGG_SKIP_SWEEP
  {
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < NMT; mt++)
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) t += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3] + (float)gp[mt][nt][0][0] + (float)gp[mt][nt][0][1] + (float)gp[mt][nt][1][0] + (float)gp[mt][nt][1][1];
    if (t == 123.f) p.frag[tid] = t;
    return;
  }
#endif
  // ---- where this lane block's runs start, and which of its positions exist
  unsigned startm = 0, validm = 0;
  {
    int prev = -2;
#pragma unroll
    for (int i = 0; i < BLK; i++) {
      const int g = s_gid[BLK * q + i];
      if (g >= 0) validm |= 1u << i;
      if (g != prev) startm |= 1u << i;
      prev = g;
    }
  }
  const size_t slot_base = (size_t)4 * blockIdx.x + q;
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) {
    const int col = col0 + nt * 16 + j;
    const float bfv = p.bf[col];
    // online softmax per run: running maximum, rescaled sum and weighted sum (the arithmetic of upd_segment_softmax).
    // (A block-wide shift with one exponential per row + an exact fallback for runs it underflows on was measured: 56 vs 51 us
    // per launch -- twice the code, 72 instead of 40 spilled registers.)
    float m = -INFINITY, z = 0.f, a = 0.f;
    int run_g = -1;
#pragma unroll
    for (int mt = 0; mt < NMT; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = 4 * mt + r;
        if (!((validm >> i) & 1)) continue;
        if ((startm >> i) & 1) {
          if (run_g >= 0) {
            float *o = p.frag + ((size_t)run_g + slot_base) * (3 * MD) + col;
            o[0] = m; o[MD] = z; o[2 * MD] = a;
          }
          m = -INFINITY; z = 0.f; a = 0.f;
          run_g = s_gid[BLK * q + i];
        }
        // (in units of log2: one multiplication per row instead of one per exponential; the fragments' maxima stay in
        // those units, upd_softagg_finish merges them with exp2 as well)
        const float gk = (float)gp[mt][nt][r >> 1][r & 1] * 1.4426950408889634f;
        const float fk = h_round(acc[mt][nt][r] + bfv);
        const float n = fmaxf(m, gk);
        const float sc = __builtin_amdgcn_exp2f(m - n), e = __builtin_amdgcn_exp2f(gk - n);
        z = z * sc + e; a = a * sc + fk * e;
        m = n;
      }
    if (run_g >= 0) {
      float *o = p.frag + ((size_t)run_g + slot_base) * (3 * MD) + col;
      o[0] = m; o[MD] = z; o[2 * MD] = a;
    }
  }
}

// hy[g] = fp16(h(y[g])), y[g][c] = a / z of group g's merged fragments (slots g + seg[g] / BLK .. g + (seg[g+1] - 1) / BLK).
// 16 groups per workgroup; 768 threads: thread (c, half) owns column c of 8 groups.  A round loads fragments k .. k + 2 of
// every one of its groups that has them (up to 72 loads in flight per thread, consecutive threads = consecutive floats),
// then merges them in slot order -- a loop over a group's fragments per element was a chain of dependent loads on a
// launch of 27 .. 132 workgroups.  Waves 0..7 then apply h (the matrix part of upd_linear_kernel).
#define SAGG_FIN_T 768
#define SAGG_FIN_K 3
__global__ void __launch_bounds__(SAGG_FIN_T) upd_softagg_finish_kernel(const float *__restrict__ frag, const int32_t *__restrict__ seg,
                                                                        const int32_t *__restrict__ ngroups,
                                                                        const _Float16 *__restrict__ wp, const float *__restrict__ bias,
                                                                        _Float16 *__restrict__ y, int rows) {
  __shared__ __attribute__((aligned(16))) _Float16 Xs[16 * MXS];
  __shared__ int s_first[16], s_nfr[16], s_maxn;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * 16;
  const int R = min(*ngroups, rows);
  if (row0 >= R) {                              // unused tail of the table: defined (zero) rows
    for (int i = tid; i < 16 * (MD / 8); i += SAGG_FIN_T) {
      const int r = i / (MD / 8), c8 = i - r * (MD / 8);
      if (row0 + r < rows) *reinterpret_cast<h8 *>(y + (size_t)(row0 + r) * MD + 8 * c8) = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    return;
  }
  constexpr int NT = MNTW, PF = 3;
  const bool mm = wave < MWAVES;                // the waves that multiply
  auto wfrag = [&](int ks, int nt) {
    return *reinterpret_cast<const h8 *>(wp + (((size_t)ks * (MD / 16) + (mm ? wave : 0) * NT + nt) * 64 + lane) * 8);
  };
  h8 ring[PF + 1][NT];
  if (mm) {
#pragma unroll
    for (int d = 0; d < PF; d++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) ring[d][nt] = wfrag(d, nt);
  }
  if (tid < 16) {
    const int g = row0 + tid;
    int first = 0, n = 0;
    if (g < R) {
      const int s0 = seg[g], s1 = seg[g + 1];
      first = g + s0 / SAGG_BLK;
      n = s1 > s0 ? (s1 - 1) / SAGG_BLK - s0 / SAGG_BLK + 1 : 0;
    }
    s_first[tid] = first; s_nfr[tid] = n;
  }
  __syncthreads();
  if (tid == 0) {
    int mx = 0;
    for (int r = 0; r < 16; r++) mx = max(mx, s_nfr[r]);
    s_maxn = mx;
  }
  __syncthreads();
  {
    const int c = tid % MD, r0 = 8 * (tid / MD);
    float m[8], z[8], a[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { m[r] = -INFINITY; z[r] = 0.f; a[r] = 0.f; }
    const int maxn = s_maxn;
    for (int k0 = 0; k0 < maxn; k0 += SAGG_FIN_K) {
      float mf[SAGG_FIN_K][8], zf[SAGG_FIN_K][8], af[SAGG_FIN_K][8];
#pragma unroll
      for (int u = 0; u < SAGG_FIN_K; u++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const bool on = k0 + u < s_nfr[r0 + r];
          const float *f = frag + ((size_t)(on ? s_first[r0 + r] + k0 + u : s_first[r0])) * (3 * MD) + c;
          mf[u][r] = f[0]; zf[u][r] = f[MD]; af[u][r] = f[2 * MD];
        }
#pragma unroll
      for (int u = 0; u < SAGG_FIN_K; u++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
          if (k0 + u >= s_nfr[r0 + r]) continue;
          const float n = fmaxf(m[r], mf[u][r]);
          const float sc = __builtin_amdgcn_exp2f(m[r] - n), tc = __builtin_amdgcn_exp2f(mf[u][r] - n);   // (maxima in units of log2)
          z[r] = z[r] * sc + zf[u][r] * tc; a[r] = a[r] * sc + af[u][r] * tc;
          m[r] = n;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) Xs[(r0 + r) * MXS + c] = (_Float16)(s_nfr[r0 + r] > 0 ? a[r] / z[r] : 0.f);
  }
  __syncthreads();
  if (!mm) return;
  f4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) acc[nt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < MKS; ks++) {
    if (ks + PF < MKS) {
#pragma unroll
      for (int nt = 0; nt < NT; nt++) ring[(ks + PF) % (PF + 1)][nt] = wfrag(ks + PF, nt);
    }
    __builtin_amdgcn_sched_barrier(0);
    const h8 a = *reinterpret_cast<const h8 *>(Xs + j * MXS + ks * 32 + 8 * q);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ks % (PF + 1)][nt], a, acc[nt], 0, 0, 0);
  }
  if (row0 + j >= rows) return;
  _Float16 *o = y + (size_t)(row0 + j) * MD + wave * (16 * NT) + 4 * q;
  const bool live = row0 + j < R;
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const f4 b = *reinterpret_cast<const f4 *>(bias + wave * (16 * NT) + nt * 16 + 4 * q);
    *reinterpret_cast<h4 *>(o + nt * 16) = live ? (h4){(_Float16)(acc[nt][0] + b[0]), (_Float16)(acc[nt][1] + b[1]),
                                                       (_Float16)(acc[nt][2] + b[2]), (_Float16)(acc[nt][3] + b[3])}
                                                : (h4){0, 0, 0, 0};
  }
}

// ------------------------------------------------------------------ plain Linear on a small table
// y[r] = fp16(x[r] W^T + b), r < rows: SoftAgg's `h` layer on the group table (ramp/blocks.py:46-47; a few hundred to a
// few thousand rows).  16 rows per workgroup, 8 waves x 48 columns, operands exchanged so that a lane holds four
// consecutive columns of one row (8-byte stores); the weight fragments of three K steps are in flight ahead of the
// matrix cores (the kernel is one L2 round trip per K step otherwise).  rows_dev: optional device-side row count (the grouping's ngroups).
__global__ void __launch_bounds__(512) upd_linear_kernel(const _Float16 *__restrict__ x, const _Float16 *__restrict__ wp,
                                                         const float *__restrict__ bias, _Float16 *__restrict__ y,
                                                         int rows, const int32_t *__restrict__ rows_dev) {
  __shared__ __attribute__((aligned(16))) _Float16 Xs[16 * MXS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * 16;
  int R = rows;
  if (rows_dev) { const int d = *rows_dev; R = d < rows ? d : rows; }
  if (row0 >= R) return;
  constexpr int NT = MNTW, PF = 3;             // 8 waves x 48 columns; weight fragments of PF K steps in flight
  auto wfrag = [&](int ks, int nt) {
    return *reinterpret_cast<const h8 *>(wp + (((size_t)ks * (MD / 16) + wave * NT + nt) * 64 + lane) * 8);
  };
  h8 ring[PF + 1][NT];
#pragma unroll
  for (int d = 0; d < PF; d++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) ring[d][nt] = wfrag(d, nt);
  for (int i = tid; i < 16 * (MD / 8); i += 512) {
    const int r = i / (MD / 8), c8 = i - r * (MD / 8);
    h8 v = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    if (row0 + r < R) v = *reinterpret_cast<const h8 *>(x + (size_t)(row0 + r) * MD + 8 * c8);
    *reinterpret_cast<h8 *>(Xs + r * MXS + 8 * c8) = v;
  }
  __syncthreads();
  f4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) acc[nt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < MKS; ks++) {
    if (ks + PF < MKS) {
#pragma unroll
      for (int nt = 0; nt < NT; nt++) ring[(ks + PF) % (PF + 1)][nt] = wfrag(ks + PF, nt);
    }
    __builtin_amdgcn_sched_barrier(0);
    const h8 a = *reinterpret_cast<const h8 *>(Xs + j * MXS + ks * 32 + 8 * q);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ks % (PF + 1)][nt], a, acc[nt], 0, 0, 0);
  }
  if (row0 + j >= R) return;
  _Float16 *o = y + (size_t)(row0 + j) * MD + wave * (16 * NT) + 4 * q;
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const f4 b = *reinterpret_cast<const f4 *>(bias + wave * (16 * NT) + nt * 16 + 4 * q);
    *reinterpret_cast<h4 *>(o + nt * 16) = (h4){(_Float16)(acc[nt][0] + b[0]), (_Float16)(acc[nt][1] + b[1]),
                                                (_Float16)(acc[nt][2] + b[2]), (_Float16)(acc[nt][3] + b[3])};
  }
}

// row tiles per workgroup for E edges (0: the 64-row kernels -- small problems, or RAMP_UPD_BIG=0; 4..8 forces a tile
// for A/B runs); `best`: the measured optimum of the chain at the bench size
static int big_pick_nmt(int E, int best) {
  static int mode = -1;
  if (mode < 0) {
    const char *e = getenv("RAMP_UPD_BIG");
    mode = e ? atoi(e) : 1;
  }
  if (mode == 0 || E < 16384 || (long)E * MD * 4 >= (1l << 32)) return 0;
  if (mode >= 4 && mode <= 8) return mode;
  return best;
}

template <typename KernelT, typename ParamsT>
static int big_launch(KernelT kernel, const ParamsT &p, int E, int nmt, int nw, bool with_red, hipStream_t st) {
  const size_t lds = (size_t)16 * nmt * MXS * 2 + (with_red ? (size_t)2 * 16 * nmt * nw * 4 : 0);
  if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return RAMP_ELAUNCH;
  hipLaunchKernelGGL(kernel, dim3(ramp_cdiv(E, 16 * nmt)), dim3(64 * nw), lds, st, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

#define BIG_DISPATCH(KERNEL, NW, P, E, NMT, RED, ST)                                       \
  switch (NMT) {                                                                             \
    case 4: return big_launch(KERNEL<4, NW>, P, E, 4, NW, RED, ST);                          \
    case 5: return big_launch(KERNEL<5, NW>, P, E, 5, NW, RED, ST);                          \
    case 6: return big_launch(KERNEL<6, NW>, P, E, 6, NW, RED, ST);                          \
    case 7: return big_launch(KERNEL<7, NW>, P, E, 7, NW, RED, ST);                          \
    default: return big_launch(KERNEL<8, NW>, P, E, 8, NW, RED, ST);                         \
  }

extern "C" {

#ifdef GRU_TRACE
int ramp_debug_gru_trace(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gru_trace), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#endif
size_t ramp_upd_mlp_lds_bytes(void) { return (size_t)2 * MBM * MXS * 2; }

int ramp_i_upd_gru(const float *x32, const void *add0_t, const int32_t *add0_idx, const void *add_t, const int32_t *add_idx,
                 const float *pre_w, const float *pre_b,
                 float pre_eps, const void *const *wp_host, const float *const *bias_host, const float *ln_w,
                 const float *ln_b, float eps, float *out32, void *relu_t, int E, const int32_t *dyn,
                 const void *heads_w, const float *heads_b, const float *coords, float *target, float *weight, int P,
                 float wd, float ht, int E_hint, uint32_t *gate_flag, uint32_t gate_seq, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!x32 || !wp_host || !bias_host || !ln_w || !ln_b || !out32 || (!relu_t && !heads_w)) return RAMP_EINVAL;
  if (heads_w && (!heads_b || !coords || !target || !weight || P < 1)) return RAMP_EINVAL;
  if (add_t && (!add_idx || !pre_w || !pre_b)) return RAMP_EINVAL;
  if (add0_t && (!add0_idx || !add_t)) return RAMP_EINVAL;
  GruParams p;
  p.x32 = x32;
  p.add0_t = (const _Float16 *)add0_t; p.add0_idx = add0_idx; p.gate_flag = gate_flag; p.gate_seq = gate_seq;
  p.add_t = (const _Float16 *)add_t; p.add_idx = add_idx; p.pre_w = pre_w; p.pre_b = pre_b; p.pre_eps = pre_eps;
  for (int i = 0; i < 6; i++) {
    if (!wp_host[i] || !bias_host[i]) return RAMP_EINVAL;
    p.wp[i] = (const _Float16 *)wp_host[i];
    p.bias[i] = bias_host[i];
  }
  p.ln_w = ln_w; p.ln_b = ln_b; p.eps = eps; p.out32 = out32; p.relu_t = (_Float16 *)relu_t; p.E = E; p.dyn = dyn;
  p.heads_w = (const _Float16 *)heads_w; p.heads_b = heads_b; p.coords = coords; p.target = target; p.weight = weight;
  p.PP = P * P; p.ctr = (P / 2) * P + P / 2; p.wd = wd; p.ht = ht;
  // rows per workgroup: whichever of 64 / 80 needs fewer rounds of one-workgroup-per-CU (ties: fewer row-rounds); with
  // device-side sizes E is the launch bound, the live count is a little below it
  static int cus = 0;
  const int mt_force = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  int mt = 4;
  if (mt_force == 4 || mt_force == 5) mt = mt_force;
  else {
    const int El = (E_hint > 0 && E_hint < E) ? E_hint : E;      // (device-side sizes: E is the launch bound, the hint the caller's estimate)
    const long r4 = (long)ramp_cdiv(ramp_cdiv(El, 64), cus) * 4, r5 = (long)ramp_cdiv(ramp_cdiv(El, 80), cus) * 5;
    mt = r5 < r4 ? 5 : 4;
  }
  const int rows = 16 * mt;
  const size_t lds = (size_t)2 * rows * MXS * 2 + 2 * rows * MWAVES * sizeof(float);   // x / h and sigmoid(gate) tiles + the LayerNorm tables
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)upd_gru_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)2 * 64 * MXS * 2 + 2 * 64 * MWAVES * sizeof(float))) != hipSuccess ||
        hipFuncSetAttribute((const void *)upd_gru_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)2 * 80 * MXS * 2 + 2 * 80 * MWAVES * sizeof(float))) != hipSuccess)
      return RAMP_ELAUNCH;
    attr_set = true;
  }
  if (mt == 5)
    hipLaunchKernelGGL(upd_gru_kernel<5>, dim3(ramp_cdiv(E, rows)), dim3(64 * MWAVES), lds, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(upd_gru_kernel<4>, dim3(ramp_cdiv(E, rows)), dim3(64 * MWAVES), lds, (hipStream_t)stream, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_i_upd_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb,
                 const float *bb, float *net_out, void *out_t, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!net_in || !idx || !wa || !ba || !wb || !bb || !net_out || net_in == net_out) return RAMP_EINVAL;
  NbrParams p;
  p.net_in = net_in; p.idx = idx; p.wa = (const _Float16 *)wa; p.wb = (const _Float16 *)wb; p.ba = ba; p.bb = bb;
  p.net_out = net_out; p.out_t = (_Float16 *)out_t; p.E = E; p.dyn = dyn;
  if (const int nmt = big_pick_nmt(E, 5)) { BIG_DISPATCH(upd_nbr_big_kernel, 8, p, E, nmt, false, (hipStream_t)stream) }
  const size_t lds = (size_t)MBM * MXS * 2;          // one tile (see the kernel)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)upd_nbr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return RAMP_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(upd_nbr_kernel, dim3(ramp_cdiv(E, MBM)), dim3(64 * MWAVES), lds, (hipStream_t)stream, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_i_upd_corr_mlp(const void *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                      const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps,
                      const float *net, const int64_t *net_map, const void *inp, const int64_t *inp_idx, long inp_mod,
                      const float *norm_w, const float *norm_b, float norm_eps, float *net_out, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!corr || corr_k <= 0 || (corr_k & 31) || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !ln_w || !ln_b || !inp ||
      !norm_w || !norm_b || !net_out || net == net_out)
    return RAMP_EINVAL;
  CorrTailParams p;
  p.corr = (const _Float16 *)corr; p.w1 = (const _Float16 *)w1; p.b1 = b1; p.corr_k = corr_k; p.c1 = nullptr;
  p.w2 = (const _Float16 *)w2; p.w3 = (const _Float16 *)w3; p.b2 = b2; p.b3 = b3;
  p.ln_w = ln_w; p.ln_b = ln_b; p.ln_eps = ln_eps; p.net = net; p.net_map = net_map; p.inp = (const _Float16 *)inp;
  p.inp_idx = inp_idx; p.inp_mod = inp_mod; p.norm_w = norm_w; p.norm_b = norm_b; p.norm_eps = norm_eps;
  p.net_out = net_out; p.E = E; p.dyn = dyn;
  const size_t lds = (size_t)MBM * MXS * 2;
  hipLaunchKernelGGL(upd_corr_tail_kernel<true>, dim3(ramp_cdiv(E, MBM)), dim3(64 * MWAVES), lds, (hipStream_t)stream, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_i_upd_fg(const float *x32, const void *add_t, const int32_t *add_idx, float *x32_out, const void *wf,
                const float *bf, const void *wg, const float *bg, void *fg, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!x32 || !wf || !bf || !wg || !bg || !fg || (add_t && !add_idx)) return RAMP_EINVAL;
  FgParams p;
  p.x32 = x32; p.add_t = (const _Float16 *)add_t; p.add_idx = add_idx; p.x32_out = x32_out;
  p.wf = (const _Float16 *)wf; p.wg = (const _Float16 *)wg; p.bf = bf; p.bg = bg; p.fg = (_Float16 *)fg; p.E = E; p.dyn = dyn;
  #ifndef FG_BEST
#define FG_BEST 6
#endif
  if (const int nmt = big_pick_nmt(E, FG_BEST)) { BIG_DISPATCH(upd_fg_big_kernel, 8, p, E, nmt, false, (hipStream_t)stream) }
  const size_t lds = (size_t)MBM * MXS * 2;
  hipLaunchKernelGGL(upd_fg_kernel, dim3(ramp_cdiv(E, MBM)), dim3(64 * MWAVES), lds, (hipStream_t)stream, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}


// ---- public entry points: host-side sizes (dyn = NULL)
int ramp_upd_gru(const float *x32, const void *add_t, const int32_t *add_idx, const float *pre_w, const float *pre_b,
                 float pre_eps, const void *const *wp_host, const float *const *bias_host, const float *ln_w,
                 const float *ln_b, float eps, float *out32, void *relu_t, int E, void *stream) {
  return ramp_i_upd_gru(x32, nullptr, nullptr, add_t, add_idx, pre_w, pre_b, pre_eps, wp_host, bias_host, ln_w, ln_b, eps, out32, relu_t, E, nullptr,
                        nullptr, nullptr, nullptr, nullptr, nullptr, 3, 0.f, 0.f, 0, nullptr, 0u, stream);
}

int ramp_upd_gru_heads(const float *x32, const void *add_t, const int32_t *add_idx, const float *pre_w, const float *pre_b,
                       float pre_eps, const void *const *wp_host, const float *const *bias_host, const float *ln_w,
                       const float *ln_b, float eps, float *out32, const void *heads_w, const float *heads_b,
                       const float *coords, float *target, float *weight, int E, int P, float wd, float ht, void *stream) {
  if (!heads_w) return RAMP_EINVAL;
  return ramp_i_upd_gru(x32, nullptr, nullptr, add_t, add_idx, pre_w, pre_b, pre_eps, wp_host, bias_host, ln_w, ln_b, eps, out32, nullptr, E, nullptr,
                        heads_w, heads_b, coords, target, weight, P, wd, ht, 0, nullptr, 0u, stream);
}

int ramp_upd_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb,
                 const float *bb, float *net_out, void *out_t, int E, void *stream) {
  return ramp_i_upd_nbr(net_in, idx, wa, ba, wb, bb, net_out, out_t, E, nullptr, stream);
}

int ramp_upd_corr_mlp(const void *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                      const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps,
                      const float *net, const int64_t *net_map, const void *inp, const int64_t *inp_idx, long inp_mod,
                      const float *norm_w, const float *norm_b, float norm_eps, float *net_out, int E, void *stream) {
  return ramp_i_upd_corr_mlp(corr, corr_k, w1, b1, w2, b2, w3, b3, ln_w, ln_b, ln_eps, net, net_map, inp, inp_idx, inp_mod, norm_w, norm_b, norm_eps, net_out, E, nullptr, stream);
}

int ramp_upd_fg(const float *x32, const void *add_t, const int32_t *add_idx, float *x32_out, const void *wf,
                const float *bf, const void *wg, const float *bg, void *fg, int E, void *stream) {
  return ramp_i_upd_fg(x32, add_t, add_idx, x32_out, wf, bf, wg, bg, fg, E, nullptr, stream);
}

int ramp_upd_linear(const void *x, const void *w_packed, const float *bias, void *y, int rows, const int32_t *rows_dev,
                    void *stream) {
  if (rows < 0) return RAMP_EINVAL;
  if (rows == 0) return RAMP_OK;
  if (!x || !w_packed || !bias || !y) return RAMP_EINVAL;
  hipLaunchKernelGGL(upd_linear_kernel, dim3(ramp_cdiv(rows, 16)), dim3(512), 0, (hipStream_t)stream,
                     (const _Float16 *)x, (const _Float16 *)w_packed, bias, (_Float16 *)y, rows, rows_dev);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}


size_t ramp_upd_softagg_frag_rows(int E, int max_groups) {
  return (size_t)(max_groups > 0 ? max_groups : 0) + (size_t)((E > 0 ? E : 0) + SAGG_BLK - 1) / SAGG_BLK + 1;
}

int ramp_i_upd_softagg(const float *x32, const void *add_t, const int32_t *add_idx, const int32_t *order, const int32_t *gid,
                       const void *wf, const float *bf, const void *wg, const float *bg, float *frag, int E,
                       const int32_t *dyn, void *stream, uint32_t *gate_flag, uint32_t gate_seq) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!x32 || !order || !gid || !wf || !bf || !wg || !bg || !frag || (add_t && !add_idx)) return RAMP_EINVAL;
  SoftAggParams p;
  p.gate_flag = gate_flag; p.gate_seq = gate_seq;
  p.x32 = x32; p.add_t = (const _Float16 *)add_t; p.add_idx = add_idx; p.order = order; p.gid = gid;
  p.wf = (const _Float16 *)wf; p.wg = (const _Float16 *)wg; p.bf = bf; p.bg = bg; p.frag = frag; p.E = E; p.dyn = dyn;
  const size_t lds = (size_t)SAGG_ROWS * MXS * 2 + SAGG_ROWS * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)upd_softagg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return RAMP_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(upd_softagg_kernel, dim3(ramp_cdiv(E, SAGG_ROWS)), dim3(512), lds, (hipStream_t)stream, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_upd_softagg(const float *x32, const void *add_t, const int32_t *add_idx, const int32_t *order, const int32_t *gid,
                     const void *wf, const float *bf, const void *wg, const float *bg, float *frag, int E, void *stream) {
  return ramp_i_upd_softagg(x32, add_t, add_idx, order, gid, wf, bf, wg, bg, frag, E, nullptr, stream, nullptr, 0);
}
int ramp_upd_softagg_finish(const float *frag, const int32_t *seg_start, const int32_t *ngroups, const void *wh,
                            const float *bh, void *hy, int max_groups, void *stream) {
  if (max_groups < 0) return RAMP_EINVAL;
  if (max_groups == 0) return RAMP_OK;
  if (!frag || !seg_start || !ngroups || !wh || !bh || !hy) return RAMP_EINVAL;
  hipLaunchKernelGGL(upd_softagg_finish_kernel, dim3(ramp_cdiv(max_groups, 16)), dim3(SAGG_FIN_T), 0, (hipStream_t)stream, frag,
                     seg_start, ngroups, (const _Float16 *)wh, bh, (_Float16 *)hy, max_groups);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

}  // extern "C"
