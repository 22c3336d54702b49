// Microbenchmark: do the update operator's chain launches lose their memory phases to lock-step, and does pipelining two
// sub-tiles inside a workgroup get them back?
//
// At E ~ 41k every chain launch of csrc/update_mlp.hip is ONE round of co-resident workgroups: they all load their rows of
// the fp32 state (61 MB), then all multiply, then all re-read the residual and store (2 x 61 MB) -- tools/mb/chain_tile.hip
// showed the launches cost about the SUM of their memory and their MFMA phases.  This program times a chain shaped like
// c1 / c2 (fp32 state rows in -> fp16 LDS tile -> L layers -> out = in + result, fp32) two ways, same values:
//   SPLIT 1  the kernels' structure today: one tile per workgroup, phases in sequence
//   SPLIT 2  the workgroup's rows as TWO sub-tiles: sub-tile 1's rows are requested (into registers) before sub-tile 0's
//            layers run and written to LDS behind them; sub-tile 0's stores drain while sub-tile 1 multiplies.  Costs the
//            weight stream twice per workgroup (chain_tile.hip: the stream is ~12 % of the loops).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_overlap tools/mb/chain_overlap.hip ; run: /tmp/chain_overlap [E] [L]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define MD 384
#define MXS (MD + 8)
#define MKS (MD / 32)
#define NW 8
#define NTW 3
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MT>
__device__ __forceinline__ void tile_gemm(const _Float16 *Xs, const _Float16 *wp, int wave, int lane, f4 (&acc)[MT][NTW]) {
  const int q = lane >> 4, j = lane & 15;
  const _Float16 *wb = wp + ((size_t)(wave * NTW) * 64 + lane) * 8;
  const _Float16 *xb = Xs + j * MXS + 8 * q;
  auto wfrag = [&](int ks, int nt) { return *reinterpret_cast<const h8 *>(wb + ((size_t)ks * (MD / 16) + nt) * 512); };
  auto afrag = [&](int t) { return *reinterpret_cast<const h8 *>(xb + (t % MT) * 16 * MXS + (t / MT) * 32); };
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) acc[mt][nt] = (f4){0.f, 0.f, 0.f, 0.f};
  h8 ring[2][NTW], ar[2];
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) ring[0][nt] = wfrag(0, nt);
  ar[0] = afrag(0);
#pragma unroll
  for (int ks = 0; ks < MKS; ks++) {
    if (ks + 1 < MKS) {
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) ring[(ks + 1) & 1][nt] = wfrag(ks + 1, nt);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const int t = ks * MT + mt;
      const bool rd = t + 1 < MKS * MT;
      if (rd) ar[(t + 1) & 1] = afrag(t + 1);
#pragma unroll
      for (int nt = 0; nt < NTW; nt++)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ks & 1][nt], ar[t & 1], acc[mt][nt], 0, 0, 0);
      if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rows [r0, r0 + 16 HM) of the fp32 state as float4 pieces per thread (all loads in flight), and their fp16 copies to LDS
template <int HM>
struct Stage {
  static constexpr int N4 = HM * 16 * (MD / 4) / (64 * NW);      // float4 pieces per thread
  float4 v[N4];
  __device__ __forceinline__ void load(const float *x, int r0, int E, int tid) {
#pragma unroll
    for (int k = 0; k < N4; k++) {
      const int i = tid + k * 64 * NW, r = i / (MD / 4), c4 = i - r * (MD / 4);
      const int row = min(r0 + r, E - 1);
      v[k] = *reinterpret_cast<const float4 *>(x + (size_t)row * MD + 4 * c4);
    }
  }
  __device__ __forceinline__ void to_lds(_Float16 *Xs, int tid) const {
#pragma unroll
    for (int k = 0; k < N4; k++) {
      const int i = tid + k * 64 * NW, r = i / (MD / 4), c4 = i - r * (MD / 4);
      *reinterpret_cast<h4 *>(Xs + r * MXS + 4 * c4) = (h4){(_Float16)v[k].x, (_Float16)v[k].y, (_Float16)v[k].z, (_Float16)v[k].w};
    }
  }
};

template <int HM>
__device__ __forceinline__ void chain_sub(_Float16 *Xs, const _Float16 *w, const float *x, float *y, int r0, int E, int L,
                                          int wave, int lane) {
  const int q = lane >> 4, j = lane & 15, col0 = wave * 16 * NTW;
  f4 acc[HM][NTW];
  for (int l = 0; l < L; l++) {
    tile_gemm<HM>(Xs, w + (size_t)l * MD * MD, wave, lane, acc);
    if (l + 1 < L) {
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < HM; mt++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++)
          *reinterpret_cast<h4 *>(Xs + (mt * 16 + j) * MXS + col0 + nt * 16 + 4 * q) =
              (h4){(_Float16)fmaxf(acc[mt][nt][0], 0.f), (_Float16)fmaxf(acc[mt][nt][1], 0.f),
                   (_Float16)fmaxf(acc[mt][nt][2], 0.f), (_Float16)fmaxf(acc[mt][nt][3], 0.f)};
      __syncthreads();
    }
  }
  // out = in + result (the residual is re-read, as upd_nbr_big_kernel does)
#pragma unroll
  for (int mt = 0; mt < HM; mt++) {
    const int row = r0 + mt * 16 + j;
    f4 xin[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) xin[nt] = *reinterpret_cast<const f4 *>(x + (size_t)min(row, E - 1) * MD + col0 + nt * 16 + 4 * q);
    if (row < E) {
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) *reinterpret_cast<f4 *>(y + (size_t)row * MD + col0 + nt * 16 + 4 * q) = xin[nt] + acc[mt][nt];
    }
  }
}

template <int MT, int SPLIT, int WPE>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
    chain_io_kernel(const float *__restrict__ x, const _Float16 *__restrict__ w, float *__restrict__ y, int E, int L) {
  constexpr int HM = MT / SPLIT;
  static_assert(MT % SPLIT == 0, "sub-tiles of whole m-tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = blockIdx.x * 16 * MT;
  if constexpr (SPLIT == 1) {
    Stage<HM> s0;
    s0.load(x, row0, E, tid);
    s0.to_lds(Xs, tid);
    __syncthreads();
    chain_sub<HM>(Xs, w, x, y, row0, E, L, wave, lane);
  } else {
    _Float16 *X1 = Xs + HM * 16 * MXS;
    Stage<HM> s0, s1;
    s0.load(x, row0, E, tid);
    s1.load(x, row0 + 16 * HM, E, tid);              // in flight while sub-tile 0 multiplies
    s0.to_lds(Xs, tid);
    __syncthreads();
    chain_sub<HM>(Xs, w, x, y, row0, E, L, wave, lane);
    s1.to_lds(X1, tid);
    __syncthreads();
    chain_sub<HM>(X1, w, x, y, row0 + 16 * HM, E, L, wave, lane);   // sub-tile 0's stores drain behind these loops
  }
}

struct Variant { const char *name; void (*fn)(const float *, const _Float16 *, float *, int, int); int mt; };
template <int MT, int SPLIT, int WPE>
Variant make(const char *name) {
  auto k = chain_io_kernel<MT, SPLIT, WPE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * MT * MXS * 2));
  return Variant{name, k, MT};
}

int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 40960, L = argc > 2 ? atoi(argv[2]) : 2, reps = 30;
  std::vector<float> hx((size_t)E * MD);
  std::vector<_Float16> hw((size_t)L * MD * MD);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto &v : hx) v = 2.0f * rnd();
  for (auto &v : hw) v = (_Float16)(0.25f * rnd());
  float *dx, *dy; _Float16 *dw;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dy, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 2));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  std::vector<Variant> vs;
  vs.push_back(make<4, 1, 6>("64 rows, one tile,      3 WG/CU"));
  vs.push_back(make<4, 2, 6>("64 rows, 2 x 32 pipelined, 3 WG/CU"));
  vs.push_back(make<6, 1, 4>("96 rows, one tile,      2 WG/CU"));
  vs.push_back(make<6, 2, 4>("96 rows, 2 x 48 pipelined, 2 WG/CU"));
  vs.push_back(make<8, 1, 2>("128 rows, one tile,     1 WG/CU"));
  vs.push_back(make<8, 2, 2>("128 rows, 2 x 64 pipelined, 1 WG/CU"));
  vs.push_back(make<10, 1, 2>("160 rows, one tile,     1 WG/CU"));
  vs.push_back(make<10, 2, 2>("160 rows, 2 x 80 pipelined, 1 WG/CU"));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> out0, out(hx.size());
  const double flop = 2.0 * E * MD * MD * L, bytes = 3.0 * E * MD * 4;
  printf("E = %d rows, %d layers (%.1f GFLOP), fp32 state: %.0f MB in + %.0f MB residual re-read + %.0f MB out\n", E, L, flop * 1e-9,
         bytes / 3e6, bytes / 3e6, bytes / 3e6);
  for (auto &v : vs) {
    const int R = 16 * v.mt, grid = (E + R - 1) / R;
    const size_t lds = (size_t)R * MXS * 2;
    CK(hipMemset(dy, 0, hx.size() * 4));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(64 * NW), lds, 0, dx, dw, dy, E, L);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(64 * NW), lds, 0, dx, dw, dy, E, L);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(out.data(), dy, out.size() * 4, hipMemcpyDeviceToHost));
    bool same = true;
    if (out0.empty()) out0 = out; else same = memcmp(out0.data(), out.data(), out.size() * 4) == 0;
    const double us = ms * 1e3 / reps;
    printf("%-40s grid %4d (%.2f per CU)  %7.1f us  %6.1f TFLOP/s  %5.2f TB/s of state traffic  %s\n", v.name, grid, grid / 256.0, us,
           flop / us * 1e-6, bytes / us * 1e-6, same ? "bit-equal" : "DIFFERS");
  }
  return 0;
}
