// Microbenchmark: which row-tile shape should the update operator's GEMM chains use?
//
// Every fused chain of csrc/update_mlp.hip has the same core: a workgroup owns R rows of the [E,384] activation in LDS
// and runs L layers X <- act(X W_l^T) on them, streaming the packed 384x384 fp16 weights of each layer from L2.  Per K
// step (32 channels) a CU's pipes carry, for a wave tile of r x c outputs: LDS A fragments r x 64 B, weight fragments
// c x 64 B through the vector L1 (64 B/clk/CU nominal, ~49 measured for 1 KB runs), and (r/16)(c/16) MFMAs of 16 cycles.
// With 8 waves x (64..80 rows x 48 columns) -- rounds 1-4 -- the three pipes are about equally loaded (MFMA : LDS : L1 =
// 1 : 0.67 : 0.8-1.0) and the loops reach ~34 % of the f16 peak.  This program times the bare chain (no LayerNorm, no
// gathers: relu between layers) for several (waves, rows, prefetch depth, workgroups per CU) so that the shapes can be
// ranked on the hardware before a real kernel is rewritten:
//   8 x (R x 48):  R = 64 (3 WG/CU, no prefetch: nbr / corr_tail), 80 (1 WG/CU, ring 2: gru), 80 (2 WG/CU, ring 1: *_big)
//   4 x (R x 96):  R = 64, 80, 96 (2 WG/CU), 128, 160 (1 WG/CU, up to 512 registers per wave)
// All variants compute the same values in the same order (bitwise equal outputs are checked).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_tile tools/mb/chain_tile.hip ; run: /tmp/chain_tile [E] [L]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define MD 384
#define MXS (MD + 8)
#define MKS (MD / 32)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// acc[mt][nt] (TRANSPOSED: lane (q, j) = row 16 mt + j, columns 16 nt + 4 q ..) += Xs[16 MT x 384] * W^T for this wave's
// 16 NTW columns.  PF: K steps of weight fragments in flight ahead of the MFMAs.  APF: the A fragment of the NEXT
// (mt, ks) is read from LDS before the MFMAs of the current one are issued.
template <int NW, int MT, int PF, int APF, int MODE = 0>
__device__ __forceinline__ void tile_gemm(const _Float16 *Xs, const _Float16 *wp, int wave, int lane, f4 (&acc)[MT][24 / NW]) {
  constexpr int NTW = 24 / NW;
  const int q = lane >> 4, j = lane & 15;
  const _Float16 *wb = wp + ((size_t)(wave * NTW) * 64 + lane) * 8;
  const _Float16 *xb = Xs + j * MXS + 8 * q;
  // MODE (diagnostic): 1 = every K step re-uses the weight fragments of step 0 (no weight stream), 2 = every tile
  // re-uses A fragment 0 (no LDS reads in the loop), 3 = both: the bare MFMA rate of this register structure
  auto wfrag = [&](int ks, int nt) { return *reinterpret_cast<const h8 *>(wb + ((size_t)((MODE & 1) ? 0 : ks) * (MD / 16) + nt) * 512); };
  auto afrag = [&](int t) { return *reinterpret_cast<const h8 *>(xb + (t % MT) * 16 * MXS + ((MODE & 2) ? 0 : (t / MT) * 32)); };   // t = ks * MT + mt
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) acc[mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
  h8 ring[PF + 1][NTW];
#pragma unroll
  for (int d = 0; d < PF; d++)
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) ring[d][nt] = wfrag(d, nt);
  h8 afix[(MODE & 2) ? MT : 1];                      // (MODE & 2: the A fragments of K step 0, read once per layer)
  if constexpr ((MODE & 2) != 0) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++) afix[mt] = afrag(mt);
  }
  constexpr int AD = APF > 0 ? APF : 1;
  h8 ar[AD + 1];                                   // A fragments APF tiles ahead of the MFMAs (a ring, static indices)
  if constexpr (APF > 0) {
#pragma unroll
    for (int d = 0; d < AD; d++) ar[d] = afrag(d);
  }
#pragma unroll
  for (int ks = 0; ks < MKS; ks++) {
    if ((ks + PF < MKS || PF == 0) && !((MODE & 1) && ks > 0)) {
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) ring[(ks + PF) % (PF + 1)][nt] = wfrag(ks + PF < MKS ? ks + PF : ks, nt);
    }
    if constexpr (APF == 0) {
      h8 av[MT];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) av[mt] = afrag(ks * MT + mt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ks % (PF + 1)][nt], av[mt], acc[mt][nt], 0, 0, 0);
    } else {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const int t = ks * MT + mt;
        const bool rd = (t + AD < MKS * MT) && !(MODE & 2);
        if (rd) ar[(t + AD) % (AD + 1)] = afrag(t + AD);
        h8 acur = ar[t % (AD + 1)];
        if constexpr ((MODE & 2) != 0) acur = afix[mt];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[(MODE & 1) ? 0 : ks % (PF + 1)][nt], acur, acc[mt][nt], 0, 0, 0);
        if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the LDS read of tile t + APF first ...
        __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);                            // ... then tile t's MFMAs
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NW, int MT, int PF, int APF, int WPE, int MODE = 0>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
    chain_kernel(const _Float16 *__restrict__ x, const _Float16 *__restrict__ w, _Float16 *__restrict__ y, int E, int L, long long *clk) {
  constexpr int NTW = 24 / NW, R = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * R;
  const int col0 = wave * 16 * NTW;
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = tid; i < R * (MD / 8); i += 64 * NW) {
    const int r = i / (MD / 8), c8 = i - r * (MD / 8);
    h8 v = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    if (row0 + r < E) v = *reinterpret_cast<const h8 *>(x + (size_t)(row0 + r) * MD + 8 * c8);
    *reinterpret_cast<h8 *>(Xs + r * MXS + 8 * c8) = v;
  }
  __syncthreads();
#pragma unroll 1
  for (int l = 0; l < L; l++) {
    f4 acc[MT][NTW];
    tile_gemm<NW, MT, PF, APF, MODE>(Xs, w + (size_t)l * MD * MD, wave, lane, acc);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < NTW; nt++)
        *reinterpret_cast<h4 *>(Xs + (mt * 16 + j) * MXS + col0 + nt * 16 + 4 * q) =
            (h4){(_Float16)fmaxf(acc[mt][nt][0], 0.f), (_Float16)fmaxf(acc[mt][nt][1], 0.f),
                 (_Float16)fmaxf(acc[mt][nt][2], 0.f), (_Float16)fmaxf(acc[mt][nt][3], 0.f)};
    __syncthreads();
  }
  for (int i = tid; i < R * (MD / 8); i += 64 * NW) {
    const int r = i / (MD / 8), c8 = i - r * (MD / 8);
    if (row0 + r < E) *reinterpret_cast<h8 *>(y + (size_t)(row0 + r) * MD + 8 * c8) = *reinterpret_cast<const h8 *>(Xs + r * MXS + 8 * c8);
  }
  if (tid == 0 && clk) {                           // shader cycles and 100 MHz ticks of this workgroup
    clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
    clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
}

// The same chain on v_mfma_f32_32x32x16_f16 (twice the flops per operand register): NW waves, wave w owns output columns
// [32 w, 32 w + 32) for all 32 MB rows.  Weights as the A operand (lane l: column 32 w + l % 32, k = 8 (l / 32) ..), the
// activation tile as B (lane l: row 32 mb + l % 32, same k), accumulators transposed: lane l holds row 32 mb + l % 32,
// columns 32 w + 8 (i / 4) + 4 (l / 32) + i % 4, i = 0..15.  w32: fragments [l][ks16][nt32][lane][8].
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NW, int MB, int PF, int AD, int WPE>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
    chain32_kernel(const _Float16 *__restrict__ x, const _Float16 *__restrict__ w32, _Float16 *__restrict__ y, int E, int L, long long *clk) {
  static_assert(NW == 12, "384 columns = 12 x 32");
  constexpr int R = 32 * MB, KS16 = MD / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hl = lane >> 5, j = lane & 31;
  const int row0 = blockIdx.x * R;
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = tid; i < R * (MD / 8); i += 64 * NW) {
    const int r = i / (MD / 8), c8 = i - r * (MD / 8);
    h8 v = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    if (row0 + r < E) v = *reinterpret_cast<const h8 *>(x + (size_t)(row0 + r) * MD + 8 * c8);
    *reinterpret_cast<h8 *>(Xs + r * MXS + 8 * c8) = v;
  }
  __syncthreads();
  const _Float16 *xb = Xs + j * MXS + 8 * hl;
#pragma unroll 1
  for (int l = 0; l < L; l++) {
    const _Float16 *wb = w32 + (size_t)l * MD * MD + ((size_t)wave * 64 + lane) * 8;
    auto wfrag = [&](int ks) { return *reinterpret_cast<const h8 *>(wb + (size_t)ks * NW * 512); };
    auto afrag = [&](int t) { return *reinterpret_cast<const h8 *>(xb + (t % MB) * 32 * MXS + (t / MB) * 16); };   // t = ks * MB + mb
    f16v acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[mb][i] = 0.f;
    h8 ring[PF + 1], ar[AD + 1];
#pragma unroll
    for (int d = 0; d < PF; d++) ring[d] = wfrag(d);
#pragma unroll
    for (int d = 0; d < AD; d++) ar[d] = afrag(d);
#pragma unroll
    for (int ks = 0; ks < KS16; ks++) {
      if (ks + PF < KS16) ring[(ks + PF) % (PF + 1)] = wfrag(ks + PF);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < MB; mb++) {
        const int t = ks * MB + mb;
        const bool rd = t + AD < KS16 * MB;
        if (rd) ar[(t + AD) % (AD + 1)] = afrag(t + AD);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[ks % (PF + 1)], ar[t % (AD + 1)], acc[mb], 0, 0, 0);
        if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
      for (int g = 0; g < 4; g++)
        *reinterpret_cast<h4 *>(Xs + (mb * 32 + j) * MXS + 32 * wave + 8 * g + 4 * hl) =
            (h4){(_Float16)fmaxf(acc[mb][4 * g + 0], 0.f), (_Float16)fmaxf(acc[mb][4 * g + 1], 0.f),
                 (_Float16)fmaxf(acc[mb][4 * g + 2], 0.f), (_Float16)fmaxf(acc[mb][4 * g + 3], 0.f)};
    __syncthreads();
  }
  for (int i = tid; i < R * (MD / 8); i += 64 * NW) {
    const int r = i / (MD / 8), c8 = i - r * (MD / 8);
    if (row0 + r < E) *reinterpret_cast<h8 *>(y + (size_t)(row0 + r) * MD + 8 * c8) = *reinterpret_cast<const h8 *>(Xs + r * MXS + 8 * c8);
  }
  if (tid == 0 && clk) {
    clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
    clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
}

struct Variant {
  const char *name;
  void (*fn)(const _Float16 *, const _Float16 *, _Float16 *, int, int, long long *);
  int nw, mt;
  bool k32 = false;
};
template <int NW, int MB, int PF, int AD, int WPE>
Variant make32(const char *name) {
  auto k = chain32_kernel<NW, MB, PF, AD, WPE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * MB * MXS * 2));
  Variant v{name, k, NW, 2 * MB};
  v.k32 = true;
  return v;
}

template <int NW, int MT, int PF, int APF, int WPE, int MODE = 0>
Variant make(const char *name) {
  auto k = chain_kernel<NW, MT, PF, APF, WPE, MODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * MT * MXS * 2));
  return Variant{name, k, NW, MT};
}

int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 40000, L = argc > 2 ? atoi(argv[2]) : 6, reps = 30;
  std::vector<_Float16> hx((size_t)E * MD), hw((size_t)L * MD * MD);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto &v : hx) v = (_Float16)(2.0f * rnd());
  // packed fragments [l][ks][nt][lane][8]: lane (q, j) of fragment (ks, nt) holds W[16 nt + j][32 ks + 8 q ..]
  std::vector<float> W((size_t)L * MD * MD);
  for (auto &v : W) v = 0.25f * rnd();
  for (int l = 0; l < L; l++)
    for (int ks = 0; ks < MKS; ks++)
      for (int nt = 0; nt < MD / 16; nt++)
        for (int lane = 0; lane < 64; lane++)
          for (int e = 0; e < 8; e++)
            hw[(size_t)l * MD * MD + (((size_t)ks * (MD / 16) + nt) * 64 + lane) * 8 + e] =
                (_Float16)W[(size_t)l * MD * MD + (size_t)(16 * nt + (lane & 15)) * MD + 32 * ks + 8 * (lane >> 4) + e];
  std::vector<_Float16> hw32(hw.size());
  for (int l = 0; l < L; l++)
    for (int ks = 0; ks < MD / 16; ks++)
      for (int nt = 0; nt < MD / 32; nt++)
        for (int lane = 0; lane < 64; lane++)
          for (int e = 0; e < 8; e++)
            hw32[(size_t)l * MD * MD + (((size_t)ks * (MD / 32) + nt) * 64 + lane) * 8 + e] =
                (_Float16)W[(size_t)l * MD * MD + (size_t)(32 * nt + (lane & 31)) * MD + 16 * ks + 8 * (lane >> 5) + e];
  _Float16 *dx, *dw, *dy, *dw32;
  CK(hipMalloc(&dw32, hw.size() * 2));
  CK(hipMemcpy(dw32, hw32.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dy, hx.size() * 2));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  // host reference of two rows
  std::vector<float> ref(2 * MD);
  for (int r = 0; r < 2; r++) {
    std::vector<float> cur(MD), nxt(MD);
    const int row = r == 0 ? 0 : E - 1;
    for (int c = 0; c < MD; c++) cur[c] = (float)hx[(size_t)row * MD + c];
    for (int l = 0; l < L; l++) {
      for (int n = 0; n < MD; n++) {
        double a = 0;
        for (int k = 0; k < MD; k++) a += (double)(float)(_Float16)W[(size_t)l * MD * MD + (size_t)n * MD + k] * cur[k];
        nxt[n] = (float)(_Float16)fmaxf((float)a, 0.f);
      }
      cur = nxt;
    }
    for (int c = 0; c < MD; c++) ref[r * MD + c] = cur[c];
  }
  std::vector<Variant> vs;
  vs.push_back(make<8, 4, 0, 0, 6>("8w x 64r  pf0      3WG/CU (nbr/corr_tail)"));
  vs.push_back(make<8, 4, 2, 0, 2>("8w x 64r  pf2      1WG/CU (gru<4>)"));
  vs.push_back(make<8, 5, 2, 0, 2>("8w x 80r  pf2      1WG/CU (gru<5>)"));
  vs.push_back(make<8, 5, 1, 0, 4>("8w x 80r  pf1      2WG/CU (*_big<5>)"));
  vs.push_back(make<8, 5, 1, 1, 4>("8w x 80r  pf1 apf  2WG/CU"));
  vs.push_back(make<8, 5, 2, 1, 2>("8w x 80r  pf2 apf  1WG/CU"));
  vs.push_back(make<8, 6, 1, 1, 4>("8w x 96r  pf1 apf  2WG/CU"));
  vs.push_back(make<4, 4, 1, 1, 3>("4w x 64r  pf1 apf  3WG/CU"));
  vs.push_back(make<4, 5, 1, 1, 2>("4w x 80r  pf1 apf  2WG/CU"));
  vs.push_back(make<4, 6, 1, 1, 2>("4w x 96r  pf1 apf  2WG/CU"));
  vs.push_back(make<4, 6, 2, 1, 2>("4w x 96r  pf2 apf  2WG/CU"));
  vs.push_back(make<4, 6, 1, 0, 2>("4w x 96r  pf1      2WG/CU"));
  vs.push_back(make<4, 8, 1, 1, 1>("4w x 128r pf1 apf  1WG/CU"));
  vs.push_back(make<4, 8, 2, 1, 1>("4w x 128r pf2 apf  1WG/CU"));
  vs.push_back(make<4, 10, 2, 1, 1>("4w x 160r pf2 apf  1WG/CU"));
  vs.push_back(make<8, 8, 1, 0, 2>("8w x 128r pf1      1WG/CU"));
  vs.push_back(make<8, 8, 1, 1, 2>("8w x 128r pf1 apf  1WG/CU"));
  vs.push_back(make<8, 10, 1, 0, 2>("8w x 160r pf1      1WG/CU"));
  vs.push_back(make<8, 10, 1, 1, 2>("8w x 160r pf1 apf  1WG/CU"));
  vs.push_back(make<8, 10, 2, 1, 2>("8w x 160r pf2 apf  1WG/CU"));
  vs.push_back(make<8, 11, 1, 1, 2>("8w x 176r pf1 apf  1WG/CU"));
  vs.push_back(make<8, 12, 1, 1, 2>("8w x 192r pf1 apf  1WG/CU"));
  vs.push_back(make<12, 10, 1, 1, 3>("12w x 160r pf1 apf 1WG/CU"));
  vs.push_back(make<12, 10, 2, 1, 3>("12w x 160r pf2 apf 1WG/CU"));
  vs.push_back(make<12, 12, 1, 1, 3>("12w x 192r pf1 apf 1WG/CU"));
  vs.push_back(make<12, 5, 1, 1, 6>("12w x 80r  pf1 apf 2WG/CU"));
  vs.push_back(make<8, 5, 1, 3, 4>("8w x 80r  pf1 apf3 2WG/CU"));
  vs.push_back(make<8, 10, 1, 2, 2>("8w x 160r pf1 apf2 1WG/CU"));
  vs.push_back(make<8, 10, 1, 3, 2>("8w x 160r pf1 apf3 1WG/CU"));
  vs.push_back(make<8, 12, 1, 3, 2>("8w x 192r pf1 apf3 1WG/CU"));
  vs.push_back(make<12, 10, 1, 4, 3>("12w x 160r pf1 apf4 1WG/CU"));
  vs.push_back(make<4, 10, 2, 2, 1>("4w x 160r pf2 apf2 1WG/CU"));
  vs.push_back(make<8, 10, 1, 1, 2, 1>("8w x 160r pf1 apf  NO WEIGHT STREAM"));
  vs.push_back(make<8, 5, 1, 1, 4, 1>("8w x 80r  pf1 apf  2WG NO WEIGHT STREAM"));
  vs.push_back(make<8, 5, 1, 1, 4, 2>("8w x 80r  pf1 apf  2WG NO LDS READS"));
  vs.push_back(make<8, 5, 1, 1, 4, 3>("8w x 80r  pf1 apf  2WG MFMA ONLY"));
  vs.push_back(make32<12, 5, 3, 2, 3>("12w x 160r 32x32x16 pf3 ad2"));
  vs.push_back(make32<12, 5, 4, 3, 3>("12w x 160r 32x32x16 pf4 ad3"));
  vs.push_back(make32<12, 4, 3, 2, 3>("12w x 128r 32x32x16 pf3 ad2"));
  vs.push_back(make32<12, 6, 3, 2, 3>("12w x 192r 32x32x16 pf3 ad2"));
  vs.push_back(make32<12, 3, 3, 2, 3>("12w x 96r  32x32x16 pf3 ad2 1WG"));
  long long *dclk;
  CK(hipMalloc(&dclk, 2 * 4096 * sizeof(long long)));
  std::vector<long long> hclk(2 * 4096);
  const bool zeros = argc > 3 && atoi(argv[3]) == 1;
  if (zeros) { CK(hipMemset(dx, 0, hx.size() * 2)); CK(hipMemset(dw, 0, hw.size() * 2)); CK(hipMemset(dw32, 0, hw.size() * 2)); printf("ZERO operands (power / clock check)\n"); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<_Float16> out0, out(hx.size());
  const double flop = 2.0 * E * MD * MD * L;
  printf("E = %d rows, %d layers of 384x384 (%.1f GFLOP, weight stream per row tile %.0f KB/layer)\n", E, L, flop * 1e-9, MD * MD * 2 / 1024.0);
  for (auto &v : vs) {
    const int R = 16 * v.mt, grid = (E + R - 1) / R;
    const size_t lds = (size_t)R * MXS * 2;
    CK(hipMemset(dy, 0, hx.size() * 2));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(64 * v.nw), lds, 0, dx, v.k32 ? dw32 : dw, dy, E, L, dclk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(64 * v.nw), lds, 0, dx, v.k32 ? dw32 : dw, dy, E, L, dclk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(out.data(), dy, out.size() * 2, hipMemcpyDeviceToHost));
    double err = 0;
    for (int c = 0; c < MD; c++) {
      err = fmax(err, fabs((double)(float)out[c] - ref[c]));
      err = fmax(err, fabs((double)(float)out[(size_t)(E - 1) * MD + c] - ref[MD + c]));
    }
    bool same = true;
    if (out0.empty()) out0 = out; else same = memcmp(out0.data(), out.data(), out.size() * 2) == 0;
    const double us = ms * 1e3 / reps;
    CK(hipMemcpy(hclk.data(), dclk, 2 * grid * sizeof(long long), hipMemcpyDeviceToHost));
    double cyc = 0, tick = 0;
    for (int b = 0; b < grid; b++) { cyc += hclk[2 * b]; tick += hclk[2 * b + 1]; }
    const double ghz = cyc / (tick * 10.0), wg_us = tick / grid / 100.0;
    printf("%-44s grid %4d (%.2f per CU)  %7.1f us  %6.1f TFLOP/s  %.3f of 2.5 PF  clk %.2f GHz  WG %.1f us  err %.2e %s\n", v.name, grid,
           grid / 256.0, us, flop / us * 1e-6, flop / us * 1e-6 / 2500.0, ghz, wg_us, err, same ? "bit-equal" : "DIFFERS");
  }
  return 0;
}
