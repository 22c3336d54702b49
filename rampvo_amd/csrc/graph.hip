// Edge-graph primitives: group-by-key (stable radix sort + segment heads),
// temporal neighbours and the SoftAgg segment softmax/sum -- the device-side
// replacements of the reference's torch.unique syncs and its host std::stable_sort.
#include "ramp_device.h"
#include "ramp_internal.h"
#include <cstdint>
#include <hipcub/hipcub.hpp>

#define GR_THREADS 256

static inline int key_bits(int64_t bound) {
  if (bound <= 0) return 63;
  int b = 1;
  while (b < 63 && ((int64_t)1 << b) < bound) b++;
  return b;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__global__ void __launch_bounds__(GR_THREADS)
    gb_init_kernel(const int64_t *__restrict__ keys, unsigned long long *__restrict__ k64,
                   int32_t *__restrict__ iota, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  k64[i] = (unsigned long long)keys[i];
  iota[i] = i;
}
__global__ void __launch_bounds__(GR_THREADS)
    gb_heads_kernel(const unsigned long long *__restrict__ ks, int32_t *__restrict__ heads, int E) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  heads[p] = (p == 0 || ks[p] != ks[p - 1]) ? 1 : 0;
}
__global__ void __launch_bounds__(GR_THREADS)
    gb_finish_kernel(const unsigned long long *__restrict__ ks, const int32_t *__restrict__ order,
                     const int32_t *__restrict__ heads, const int32_t *__restrict__ incl,
                     int32_t *__restrict__ gid, int32_t *__restrict__ seg_start,
                     int64_t *__restrict__ ukeys, int32_t *__restrict__ ngroups, int E) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  const int g = incl[p] - 1;
  if (gid) gid[order[p]] = g;
  if (heads[p]) {
    seg_start[g] = p;
    if (ukeys) ukeys[g] = (int64_t)ks[p];
  }
  if (p == E - 1) {
    seg_start[g + 1] = E;
    *ngroups = g + 1;
  }
}

// workspace carve for group_by
struct GbWs {
  unsigned long long *k_in, *k_out;
  int32_t *iota, *heads, *incl;
  void *cub;
  size_t cub_bytes;
};
static size_t gb_cub_bytes(int E) {
  size_t a = 0, b = 0;
  unsigned long long *k = nullptr;
  int32_t *v = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, k, k, v, v, E > 0 ? E : 1, 0, 64, (hipStream_t)0);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, v, v, E > 0 ? E : 1, (hipStream_t)0);
  return align_up((a > b ? a : b) + 256, 256);
}
static size_t gb_carve(void *ws, int E, GbWs *w) {
  const size_t n = (size_t)(E > 0 ? E : 1);
  size_t off = 0;
  char *base = (char *)ws;
  auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  w->k_in = (unsigned long long *)take(n * 8);
  w->k_out = (unsigned long long *)take(n * 8);
  w->iota = (int32_t *)take(n * 4);
  w->heads = (int32_t *)take(n * 4);
  w->incl = (int32_t *)take(n * 4);
  w->cub_bytes = gb_cub_bytes(E);
  w->cub = take(w->cub_bytes);
  return off;
}

// keys already prepared in w.k_in / w.iota
static int gb_run_sorted(GbWs &w, int E, int bits, int32_t *order, int32_t *gid,
                         int32_t *seg_start, int64_t *ukeys, int32_t *ngroups, hipStream_t st) {
  size_t cb = w.cub_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.k_in, w.k_out, w.iota, order, E, 0, bits,
                                         st) != hipSuccess)
    return RAMP_ELAUNCH;
  const int nb = ramp_cdiv(E, GR_THREADS);
  hipLaunchKernelGGL(gb_heads_kernel, dim3(nb), dim3(GR_THREADS), 0, st, w.k_out, w.heads, E);
  cb = w.cub_bytes;
  if (hipcub::DeviceScan::InclusiveSum(w.cub, cb, w.heads, w.incl, E, st) != hipSuccess)
    return RAMP_ELAUNCH;
  hipLaunchKernelGGL(gb_finish_kernel, dim3(nb), dim3(GR_THREADS), 0, st, w.k_out, order, w.heads,
                     w.incl, gid, seg_start, ukeys, ngroups, E);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

size_t ramp_internal_group_by_ws(int E) {
  GbWs w;
  return gb_carve(nullptr, E, &w);
}

int ramp_internal_group_by(const int64_t *keys, int E, int64_t key_bound, int32_t *order,
                           int32_t *gid, int32_t *seg_start, int64_t *ukeys, int32_t *ngroups,
                           void *ws, size_t ws_bytes, hipStream_t st) {
  if (E < 0 || !ngroups || !seg_start) return RAMP_EINVAL;
  if (E == 0) {
    (void)hipMemsetAsync(ngroups, 0, sizeof(int32_t), st);
    (void)hipMemsetAsync(seg_start, 0, sizeof(int32_t), st);
    return RAMP_OK;
  }
  if (!keys || !order || !ws) return RAMP_EINVAL;
  GbWs w;
  if (gb_carve(ws, E, &w) > ws_bytes) return RAMP_EWORKSPACE;
  hipLaunchKernelGGL(gb_init_kernel, dim3(ramp_cdiv(E, GR_THREADS)), dim3(GR_THREADS), 0, st, keys,
                     w.k_in, w.iota, E);
  return gb_run_sorted(w, E, key_bits(key_bound), order, gid, seg_start, ukeys, ngroups, st);
}

// ---------------------------------------------------------------- neighbors
__global__ void __launch_bounds__(GR_THREADS)
    nb_key_kernel(const int64_t *__restrict__ kk, const int64_t *__restrict__ jj,
                  unsigned long long *__restrict__ k64, int32_t *__restrict__ iota, int E,
                  long long jmul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  k64[i] = (unsigned long long)(kk[i] * jmul + jj[i]);
  iota[i] = i;
}
__global__ void __launch_bounds__(GR_THREADS)
    nb_gather_key_kernel(const int64_t *__restrict__ kk, const int32_t *__restrict__ order_in,
                         unsigned long long *__restrict__ k64, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  k64[i] = (unsigned long long)kk[order_in[i]];
}
__global__ void __launch_bounds__(GR_THREADS)
    nb_link_kernel(const int64_t *__restrict__ kk, const int32_t *__restrict__ order,
                   int64_t *__restrict__ ix, int64_t *__restrict__ jx, int E) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= E) return;
  const int e = order[p];
  const int64_t k = kk[e];
  int64_t prev = -1, next = -1;
  if (p > 0) { const int q = order[p - 1]; if (kk[q] == k) prev = q; }
  if (p + 1 < E) { const int q = order[p + 1]; if (kk[q] == k) next = q; }
  ix[e] = prev;
  jx[e] = next;
}

// ---------------------------------------------------- segment softmax + sum
// one workgroup per group, lanes over channels; three ordered passes over the
// group's rows (max, sum of exp, weighted sum) == torch_scatter's composite
// scatter_softmax followed by scatter_sum, in sorted (= ascending edge) order.
template <typename T>
__global__ void __launch_bounds__(128)
    seg_softmax_sum_kernel(const T *__restrict__ fx, const T *__restrict__ gx,
                           const int32_t *__restrict__ order, const int32_t *__restrict__ seg_start,
                           const int32_t *__restrict__ ngroups, T *__restrict__ y, int C) {
  const int g = blockIdx.x;
  if (g >= *ngroups) return;
  const int s0 = seg_start[g], s1 = seg_start[g + 1];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = -INFINITY;
    for (int p = s0; p < s1; p++) {
      const float v = (float)gx[(size_t)order[p] * C + c];
      m = v > m ? v : m;
    }
    float s = 0.0f;
    for (int p = s0; p < s1; p++) s += expf((float)gx[(size_t)order[p] * C + c] - m);
    float acc = 0.0f;
    for (int p = s0; p < s1; p++) {
      const size_t r = (size_t)order[p] * C + c;
      const float w = expf((float)gx[r] - m) / s;
      acc += (float)fx[r] * w;
    }
    y[(size_t)g * C + c] = (T)acc;
  }
}

// ------------------------------------------------- counting group-by (small key range)
// key = a[e]*mul + (b ? b[e] : 0) - sub in [0, K).  histogram -> single-workgroup scan ->
// scatter -> per-group rank sort by edge index (restores the stable order).  5 short kernels
// instead of a radix sort's ~12; used when the caller knows a tight key range (the tracker does).
__device__ __forceinline__ long gbc_key(const int64_t *a, const int64_t *b, long mul, long sub, int e) {
  return a[e] * mul + (b ? b[e] : 0) - sub;
}
// The tracker's pair grouping has ~90 distinct keys for 40k edges: one global atomic per edge is 40k atomics on 90
// addresses (22 us).  With K <= GBC_LDS_K the workgroup counts in LDS first and issues one global atomic per key it
// saw (a few dozen per workgroup).
#define GBC_LDS_K 4096
template <bool LDS>
__global__ void __launch_bounds__(256)
    gbc_hist_kernel(const int64_t *__restrict__ a, const int64_t *__restrict__ b, long mul, long sub,
                    int32_t *__restrict__ hist, int E, int K, int32_t *__restrict__ bad) {
  __shared__ int s_bin[LDS ? GBC_LDS_K : 1];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  long k = -1;
  if (e < E) {
    k = gbc_key(a, b, mul, sub, e);
    if (k < 0 || k >= K) { *bad = 1; k = -1; }
  }
  if (!LDS) {
    if (k >= 0) atomicAdd(&hist[k], 1);
    return;
  }
  for (int q = threadIdx.x; q < K; q += 256) s_bin[q] = 0;
  __syncthreads();
  if (k >= 0) atomicAdd(&s_bin[k], 1);
  __syncthreads();
  for (int q = threadIdx.x; q < K; q += 256) {
    const int c = s_bin[q];
    if (c) atomicAdd(&hist[q], c);
  }
}
// one workgroup: exclusive scan of the counts (-> cursor/offset) and of (count > 0) (-> group id)
__global__ void __launch_bounds__(1024)
    gbc_scan_kernel(int32_t *__restrict__ hist, int32_t *__restrict__ gidmap, int32_t *__restrict__ seg_start,
                    int64_t *__restrict__ ukeys, int32_t *__restrict__ ngroups, int K, int E, long sub,
                    long mul_is_pair) {
  __shared__ int s_cnt[1024], s_grp[1024];
  const int tid = threadIdx.x;
  const int per = (K + 1023) / 1024;
  const int k0 = tid * per, k1 = min(K, k0 + per);
  int c = 0, g = 0;
  for (int k = k0; k < k1; k++) { const int h = hist[k]; c += h; g += (h > 0); }
  s_cnt[tid] = c; s_grp[tid] = g;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    int vc = 0, vg = 0;
    if (tid >= off) { vc = s_cnt[tid - off]; vg = s_grp[tid - off]; }
    __syncthreads();
    s_cnt[tid] += vc; s_grp[tid] += vg;
    __syncthreads();
  }
  int oc = s_cnt[tid] - c, og = s_grp[tid] - g;   // exclusive prefixes of this thread's chunk
  for (int k = k0; k < k1; k++) {
    const int h = hist[k];
    hist[k] = oc;               // becomes the scatter cursor
    if (h > 0) {
      gidmap[k] = og;
      seg_start[og] = oc;
      if (ukeys) ukeys[og] = (int64_t)k + sub;
      og++;
    } else {
      gidmap[k] = -1;
    }
    oc += h;
  }
  if (tid == 1023) { *ngroups = s_grp[1023]; seg_start[s_grp[1023]] = E; }
}
// (the order inside a segment is arbitrary here either way; gbc_segsort_kernel restores it)
template <bool LDS>
__global__ void __launch_bounds__(256)
    gbc_scatter_kernel(const int64_t *__restrict__ a, const int64_t *__restrict__ b, long mul, long sub,
                       int32_t *__restrict__ cursor, const int32_t *__restrict__ gidmap,
                       int32_t *__restrict__ tmp_order, int32_t *__restrict__ gid, int E, int K) {
  __shared__ int s_bin[LDS ? GBC_LDS_K : 1];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  long k = -1;
  if (e < E) {
    k = gbc_key(a, b, mul, sub, e);
    if (k < 0 || k >= K) k = -1;
  }
  if (!LDS) {
    if (k < 0) return;
    const int pos = atomicAdd(&cursor[k], 1);
    tmp_order[pos] = e;
    if (gid) gid[e] = gidmap[k];
    return;
  }
  for (int q = threadIdx.x; q < K; q += 256) s_bin[q] = 0;
  __syncthreads();
  int r = 0;
  if (k >= 0) r = atomicAdd(&s_bin[k], 1);            // rank inside this workgroup's share of the segment
  __syncthreads();
  for (int q = threadIdx.x; q < K; q += 256) {
    const int c = s_bin[q];
    if (c) s_bin[q] = atomicAdd(&cursor[q], c);       // the share's base
  }
  __syncthreads();
  if (k >= 0) {
    tmp_order[s_bin[k] + r] = e;
    if (gid) gid[e] = gidmap[k];
  }
}
// restore ascending edge order inside every segment (rank by counting; segments are short)
__global__ void __launch_bounds__(64)
    gbc_segsort_kernel(const int32_t *__restrict__ tmp_order, const int32_t *__restrict__ seg_start,
                       const int32_t *__restrict__ ngroups, int32_t *__restrict__ order) {
  const int g = blockIdx.x;
  if (g >= *ngroups) return;
  const int s0 = seg_start[g], n = seg_start[g + 1] - s0;
  for (int p = threadIdx.x; p < n; p += 64) {
    const int v = tmp_order[s0 + p];
    int r = 0;
    for (int q = 0; q < n; q++) r += (tmp_order[s0 + q] < v);
    order[s0 + r] = v;
  }
}
// temporal neighbours from the per-patch groups: rank each edge of a group by (jj, edge index)
__global__ void __launch_bounds__(64)
    nb_from_groups_kernel(const int32_t *__restrict__ order, const int32_t *__restrict__ seg_start,
                          const int32_t *__restrict__ ngroups, const int64_t *__restrict__ jj,
                          int64_t *__restrict__ ix, int64_t *__restrict__ jx, int32_t *__restrict__ kj) {
  __shared__ int s_sorted[1024];
  const int g = blockIdx.x;
  if (g >= *ngroups) return;
  const int s0 = seg_start[g], n = seg_start[g + 1] - s0;
  if (n > 1024) return;   // host falls back to the sort-based path for such graphs
  for (int p = threadIdx.x; p < n; p += 64) {
    const int e = order[s0 + p];
    const long j = jj[e];
    int r = 0;
    for (int q = 0; q < n; q++) {
      const int f = order[s0 + q];
      const long jf = jj[f];
      r += (jf < j) || (jf == j && f < e);
    }
    s_sorted[r] = e;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += 64) {
    const int e = s_sorted[r];
    ix[e] = r > 0 ? s_sorted[r - 1] : -1;
    jx[e] = r + 1 < n ? s_sorted[r + 1] : -1;
    if (kj) kj[s0 + r] = e;                    // the (kk, jj)-sorted factor list (csrc/update_mlp.hip::upd_nbr2_kernel)
  }
}

// ------------------------------------------------- both groupings of the tracker's graph, device-side sizes
// The same counting group-by for the two keys the update operator and BA share (kk; (jj, ii) lexicographic), in ONE
// set of launches (blockIdx.y = grouping) and with the edge count / key offsets read from the tracker's device-side
// size block (RAMP_DYN_*): nothing here waits for the host to learn the outcome of the keyframe test.
//   grouping 0: key = kk - KLO,                     K = N * M - KLO
//   grouping 1: key = (jj - FLO) * W + (ii - FLO),  K = W * W          (ukeys = jj * W + ii, like the host path)
#define PLAN_LDS_K 12288
struct PlanDyn {
  const int64_t *ii, *jj, *kk;
  const int32_t *dyn;
  int M;
  int32_t *hist[2], *gidmap[2], *tmp[2];
  int32_t *order[2], *gid[2], *seg[2], *ngroups[2];
  int64_t *ukeys[2];
  int Kcap[2];
  int E_fill;            // gid / ix / jx get defined entries up to here (the next step's launch bound)
  int Gcap[2];           // capacity of seg / ukeys (groups): more groups than that are flagged (status bit 8), never written
  int32_t *status;
};
// (up to two plans in one set of launches, blockIdx.z: the shipped step launches one; the pair served round 5's speculative
// keyframe edit, tools/shelved/)
struct PlanPair {
  PlanDyn s[2];
  int64_t *ix[2], *jx[2];
  int32_t *kj[2];
};
__device__ __forceinline__ long plan_key(const PlanDyn &p, int g, int e, int &K, long &sub) {
  if (g == 0) {
    sub = p.dyn[RAMP_DYN_KLO];
    K = p.dyn[RAMP_DYN_N] * p.M - (int)sub;
    return p.kk[e] - sub;
  }
  const long flo = p.dyn[RAMP_DYN_FLO], W = p.dyn[RAMP_DYN_W];
  sub = flo * W + flo;
  K = (int)(W * W);
  return p.jj[e] * W + p.ii[e] - sub;
}
__device__ __forceinline__ void plan_K(const PlanDyn &p, int g, int &K, long &sub) {
  if (g == 0) {
    sub = p.dyn[RAMP_DYN_KLO];
    K = p.dyn[RAMP_DYN_N] * p.M - (int)sub;
  } else {
    const long flo = p.dyn[RAMP_DYN_FLO], W = p.dyn[RAMP_DYN_W];
    sub = flo * W + flo;
    K = (int)(W * W);
  }
  if (K > p.Kcap[g] || K < 0) K = 0;     // flagged by plan_hist_kernel; nothing is written out of bounds
}
// factors per workgroup (PLAN_EPB / 256 per thread): the LDS table (K bins to clear and to flush per workgroup) against the
// number of workgroups that share the launch's latency -- at 40-odd thousand factors more, smaller workgroups win
#ifndef PLAN_EPB
#define PLAN_EPB 256     // (1024 / 512 / 256 measured: scatter 6.7 / 5.5 / 5.0 us, histogram 5.0 / 4.8 / 4.6, scan 6.4 / 6.2 / 5.9)
#endif
template <bool LDS>
__global__ void __launch_bounds__(256) plan_hist_kernel(const PlanPair pp) {
  __shared__ int s_bin[LDS ? PLAN_LDS_K : 1];
  const PlanDyn &p = pp.s[blockIdx.z];
  int32_t *status = p.status;
  const int g = blockIdx.y, E = p.dyn[RAMP_DYN_E];
  if ((int)blockIdx.x * PLAN_EPB >= E) return;
  int K; long sub;
  plan_K(p, g, K, sub);
  long k[PLAN_EPB / 256];
#pragma unroll
  for (int u = 0; u < PLAN_EPB / 256; u++) {
    const int e = blockIdx.x * PLAN_EPB + u * 256 + threadIdx.x;
    k[u] = -1;
    if (e < E) {
      int K2; long s2;
      k[u] = plan_key(p, g, e, K2, s2);
      if (k[u] < 0 || k[u] >= K) { atomicOr(status, 8); k[u] = -1; }
    }
  }
  int32_t *hist = p.hist[g];
  if (!LDS) {
#pragma unroll
    for (int u = 0; u < PLAN_EPB / 256; u++)
      if (k[u] >= 0) atomicAdd(&hist[k[u]], 1);
    return;
  }
  for (int q = threadIdx.x; q < K; q += 256) s_bin[q] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PLAN_EPB / 256; u++)
    if (k[u] >= 0) atomicAdd(&s_bin[k[u]], 1);
  __syncthreads();
  for (int q = threadIdx.x; q < K; q += 256) {
    const int c = s_bin[q];
    if (c) atomicAdd(&hist[q], c);
  }
}
__global__ void __launch_bounds__(1024) plan_scan_kernel(const PlanPair pp) {
  __shared__ int s_cnt[1024], s_grp[1024];
  const PlanDyn &p = pp.s[blockIdx.z];
  const int g = blockIdx.x, tid = threadIdx.x, E = p.dyn[RAMP_DYN_E];
  int K; long sub;
  plan_K(p, g, K, sub);
  int32_t *hist = p.hist[g], *gidmap = p.gidmap[g], *seg_start = p.seg[g];
  int64_t *ukeys = p.ukeys[g];
  const int per = (K + 1023) / 1024;
  const int k0 = tid * per, k1 = min(K, k0 + per);
  int c = 0, n = 0;
  for (int k = k0; k < k1; k++) { const int h = hist[k]; c += h; n += (h > 0); }
  s_cnt[tid] = c; s_grp[tid] = n;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int vc = 0, vg = 0;
    if (tid >= off) { vc = s_cnt[tid - off]; vg = s_grp[tid - off]; }
    __syncthreads();
    s_cnt[tid] += vc; s_grp[tid] += vg;
    __syncthreads();
  }
  int oc = s_cnt[tid] - c, og = s_grp[tid] - n;
  for (int k = k0; k < k1; k++) {
    const int h = hist[k];
    hist[k] = oc;
    if (h > 0) {
      if (og < p.Gcap[g]) {
        gidmap[k] = og;
        seg_start[og] = oc;
        if (ukeys) ukeys[og] = (int64_t)k + sub;
      } else {
        gidmap[k] = -1;
        atomicOr(p.status, 8);
      }
      og++;
    } else {
      gidmap[k] = -1;
    }
    oc += h;
  }
  if (tid == 1023) { const int ng = min(s_grp[1023], p.Gcap[g]); *p.ngroups[g] = ng; seg_start[ng] = E; }
}
template <bool LDS>
__global__ void __launch_bounds__(256) plan_scatter_kernel(const PlanPair pp) {
  __shared__ int s_bin[LDS ? PLAN_LDS_K : 1];
  const PlanDyn &p = pp.s[blockIdx.z];
  const int g = blockIdx.y, E = p.dyn[RAMP_DYN_E];
  if ((int)blockIdx.x * PLAN_EPB >= E) return;
  int K; long sub;
  plan_K(p, g, K, sub);
  long k[PLAN_EPB / 256];
#pragma unroll
  for (int u = 0; u < PLAN_EPB / 256; u++) {
    const int e = blockIdx.x * PLAN_EPB + u * 256 + threadIdx.x;
    k[u] = -1;
    if (e < E) {
      int K2; long s2;
      k[u] = plan_key(p, g, e, K2, s2);
      if (k[u] < 0 || k[u] >= K) k[u] = -1;
    }
  }
  int32_t *cursor = p.hist[g], *tmp_order = p.tmp[g], *gid = p.gid[g];
  const int32_t *gidmap = p.gidmap[g];
  if (!LDS) {
#pragma unroll
    for (int u = 0; u < PLAN_EPB / 256; u++) {
      if (k[u] < 0) continue;
      const int e = blockIdx.x * PLAN_EPB + u * 256 + threadIdx.x;
      tmp_order[atomicAdd(&cursor[k[u]], 1)] = e;
      gid[e] = gidmap[k[u]];
    }
    return;
  }
  for (int q = threadIdx.x; q < K; q += 256) s_bin[q] = 0;
  __syncthreads();
  int r[PLAN_EPB / 256];
#pragma unroll
  for (int u = 0; u < PLAN_EPB / 256; u++) r[u] = k[u] >= 0 ? atomicAdd(&s_bin[k[u]], 1) : 0;   // rank in this workgroup's share
  __syncthreads();
  for (int q = threadIdx.x; q < K; q += 256) {
    const int c = s_bin[q];
    if (c) s_bin[q] = atomicAdd(&cursor[q], c);       // the share's base
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PLAN_EPB / 256; u++) {
    if (k[u] < 0) continue;
    const int e = blockIdx.x * PLAN_EPB + u * 256 + threadIdx.x;
    tmp_order[s_bin[k[u]] + r[u]] = e;
    gid[e] = gidmap[k[u]];
  }
}
// Per segment, from LDS: restore ascending factor order (rank by counting), and -- for the kk grouping -- the temporal
// neighbours of every factor of the patch and the patch's run of the (kk, jj)-sorted factor list (rank by (jj, factor)):
// what nb_from_groups_kernel computed in a launch of its own.  The first workgroups also clear the histograms for the
// next plan (nobody reads them after the scatter): no memset launch.
#define PLAN_SEG_LDS 4096
__global__ void __launch_bounds__(256) plan_segsort_kernel(const PlanPair pp, int hist_words, int32_t *__restrict__ mirror) {
  __shared__ int s_v[PLAN_SEG_LDS];
  __shared__ int s_kj[1024];
  const PlanDyn &p = pp.s[blockIdx.z];
  int64_t *__restrict__ ix = pp.ix[blockIdx.z], *__restrict__ jx = pp.jx[blockIdx.z];
  int32_t *__restrict__ kj = pp.kj[blockIdx.z];
  const int g = blockIdx.y, grp = blockIdx.x;
  if (g == 0) {
    const int z = grp * 256 + threadIdx.x;
    if (z < hist_words) p.hist[0][z] = 0;               // hist[0] and hist[1] are one allocation
  }
  // rows between the live factor count and the launch bound: defined entries (no neighbour, group 0) -- a caller that runs
  // its own update operator on E_bound rows gathers through them
  for (int e = p.dyn[RAMP_DYN_E] + grp * 256 + threadIdx.x; e < p.E_fill; e += gridDim.x * 256) {
    p.gid[g][e] = 0;
    if (g == 0 && ix) { ix[e] = -1; jx[e] = -1; }
  }
  // the host's lazy copy of the sizes (mapped pinned memory): nothing changes them after this launch has started
  // (word RAMP_DYN_FRAME2 repeats RAMP_DYN_FRAME in the other 64-byte half: the host re-reads a copy whose tags differ)
  if (mirror && g == 1 && grp == 0 && threadIdx.x < RAMP_DYN_WORDS)
    mirror[threadIdx.x] = p.dyn[threadIdx.x == RAMP_DYN_FRAME2 ? RAMP_DYN_FRAME : threadIdx.x];
  if (grp >= *p.ngroups[g]) return;
  const int32_t *tmp_order = p.tmp[g], *seg_start = p.seg[g];
  int32_t *order = p.order[g];
  const int s0 = seg_start[grp], n = seg_start[grp + 1] - s0;
  const bool lds = n <= PLAN_SEG_LDS;
  if (lds) {
    for (int q = threadIdx.x; q < n; q += 256) s_v[q] = tmp_order[s0 + q];
    __syncthreads();
  }
  for (int q = threadIdx.x; q < n; q += 256) {
    const int v = lds ? s_v[q] : tmp_order[s0 + q];
    int r = 0;
    if (lds) {
      for (int u = 0; u < n; u++) r += (s_v[u] < v);
    } else {
      for (int u = 0; u < n; u++) r += (tmp_order[s0 + u] < v);
    }
    order[s0 + r] = v;
  }
  if (g != 0 || !ix) return;
  if (n > 1024) {                                        // a patch with more than 1024 factors: no neighbours from here --
    if (threadIdx.x == 0) atomicOr(p.status, 8);         // flagged (the host-driven path has ramp_neighbors' sort for such graphs)
    return;
  }
  for (int q = threadIdx.x; q < n; q += 256) {
    const int e = s_v[q];
    const long j = p.jj[e];
    int r = 0;
    for (int u = 0; u < n; u++) {
      const int f = s_v[u];
      const long jf = p.jj[f];
      r += (jf < j) || (jf == j && f < e);
    }
    s_kj[r] = e;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += 256) {
    const int e = s_kj[r];
    ix[e] = r > 0 ? s_kj[r - 1] : -1;
    jx[e] = r + 1 < n ? s_kj[r + 1] : -1;
    if (kj) kj[s0 + r] = e;
  }
}

size_t ramp_i_plan_dyn_ws(int E_cap, int kkey_cap, int pkey_cap) {
  const size_t e = (size_t)(E_cap > 0 ? E_cap : 1);
  return align_up((size_t)(kkey_cap + pkey_cap + 4) * 4, 256) * 2 + 2 * align_up(e * 4, 256) + 256;
}

// the graph plan (two groupings + temporal neighbours) of the factor list g4 = [4][E_cap] int64 (ii, jj, kk, row)
static int plan_fill(PlanPair &pp, int z, const int64_t *g4, int E_cap, int E_grid, const int32_t *dyn, int32_t *status, int M,
                     int kkey_cap, int pkey_cap, int kk_cap, int ij_cap, int32_t *kk_order, int32_t *kk_gid, int32_t *kk_seg,
                     int32_t *kk_ngroups, int64_t *kk_ukeys, int32_t *ij_order, int32_t *ij_gid, int32_t *ij_seg,
                     int32_t *ij_ngroups, int64_t *ij_ukeys, int64_t *ix, int64_t *jx, int32_t *kj, void *ws, size_t ws_bytes) {
  if (!g4 || !dyn || !status || !ws || E_cap <= 0 || kkey_cap <= 0 || pkey_cap <= 0) return RAMP_EINVAL;
  if (ws_bytes < ramp_i_plan_dyn_ws(E_cap, kkey_cap, pkey_cap)) return RAMP_EWORKSPACE;
  PlanDyn &p = pp.s[z];
  p.ii = g4; p.jj = g4 + E_cap; p.kk = g4 + 2 * (size_t)E_cap; p.dyn = dyn; p.M = M;
  char *base = (char *)ws;
  const size_t hb = align_up((size_t)(kkey_cap + pkey_cap + 4) * 4, 256);
  p.hist[0] = (int32_t *)base; p.hist[1] = p.hist[0] + kkey_cap + 2;
  p.gidmap[0] = (int32_t *)(base + hb); p.gidmap[1] = p.gidmap[0] + kkey_cap + 2;
  p.tmp[0] = (int32_t *)(base + 2 * hb); p.tmp[1] = (int32_t *)(base + 2 * hb + align_up((size_t)E_cap * 4, 256));
  p.order[0] = kk_order; p.gid[0] = kk_gid; p.seg[0] = kk_seg; p.ngroups[0] = kk_ngroups; p.ukeys[0] = kk_ukeys;
  p.order[1] = ij_order; p.gid[1] = ij_gid; p.seg[1] = ij_seg; p.ngroups[1] = ij_ngroups; p.ukeys[1] = ij_ukeys;
  p.Kcap[0] = kkey_cap; p.Kcap[1] = pkey_cap;
  p.Gcap[0] = kk_cap; p.Gcap[1] = ij_cap; p.status = status;
  p.E_fill = E_grid > 0 && E_grid < E_cap ? E_grid : E_cap;
  pp.ix[z] = ix; pp.jx[z] = jx; pp.kj[z] = kj;
  return RAMP_OK;
}
// (the histograms are zero on entry: the caller's workspace starts zeroed and every plan clears them at its end)
static int plan_launch(const PlanPair &pp, int nz, int E_cap, int E_grid, int kkey_cap, int pkey_cap, int kk_cap, int ij_cap,
                       int32_t *mirror, hipStream_t st) {
  const int nb = ramp_cdiv(E_grid > 0 && E_grid < E_cap ? E_grid : E_cap, PLAN_EPB);
  const bool lds = kkey_cap <= PLAN_LDS_K && pkey_cap <= PLAN_LDS_K;
  if (lds) hipLaunchKernelGGL(plan_hist_kernel<true>, dim3(nb, 2, nz), dim3(256), 0, st, pp);
  else hipLaunchKernelGGL(plan_hist_kernel<false>, dim3(nb, 2, nz), dim3(256), 0, st, pp);
  hipLaunchKernelGGL(plan_scan_kernel, dim3(2, 1, nz), dim3(1024), 0, st, pp);
  if (lds) hipLaunchKernelGGL(plan_scatter_kernel<true>, dim3(nb, 2, nz), dim3(256), 0, st, pp);
  else hipLaunchKernelGGL(plan_scatter_kernel<false>, dim3(nb, 2, nz), dim3(256), 0, st, pp);
  const int hist_words = kkey_cap + pkey_cap + 4;
  int gx = kk_cap > ij_cap ? kk_cap : ij_cap;
  if (gx < ramp_cdiv(hist_words, 256)) gx = ramp_cdiv(hist_words, 256);
  hipLaunchKernelGGL(plan_segsort_kernel, dim3(gx, 2, nz), dim3(256), 0, st, pp, hist_words, mirror);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_i_plan_dyn(const int64_t *g4, int E_cap, int E_grid, const int32_t *dyn, int32_t *status, int M, int kkey_cap,
                    int pkey_cap, int kk_cap, int ij_cap, int32_t *kk_order, int32_t *kk_gid, int32_t *kk_seg,
                    int32_t *kk_ngroups, int64_t *kk_ukeys, int32_t *ij_order, int32_t *ij_gid, int32_t *ij_seg,
                    int32_t *ij_ngroups, int64_t *ij_ukeys, int64_t *ix, int64_t *jx, int32_t *kj, void *ws,
                    size_t ws_bytes, int32_t *mirror, hipStream_t st) {
  PlanPair pp;
  const int rc = plan_fill(pp, 0, g4, E_cap, E_grid, dyn, status, M, kkey_cap, pkey_cap, kk_cap, ij_cap, kk_order, kk_gid, kk_seg,
                           kk_ngroups, kk_ukeys, ij_order, ij_gid, ij_seg, ij_ngroups, ij_ukeys, ix, jx, kj, ws, ws_bytes);
  if (rc != RAMP_OK) return rc;
  pp.s[1] = pp.s[0]; pp.ix[1] = ix; pp.jx[1] = jx; pp.kj[1] = kj;
  return plan_launch(pp, 1, E_cap, E_grid, kkey_cap, pkey_cap, kk_cap, ij_cap, mirror, st);
}
extern "C" {

size_t ramp_group_by_small_workspace_bytes(int E, int K) {
  return align_up((size_t)(K + 2) * 4, 256) * 2 + align_up((size_t)(E > 0 ? E : 1) * 4, 256) + 256;
}

int ramp_group_by_small(const int64_t *a, const int64_t *b, int64_t mul, int64_t sub, int K, int E,
                        int32_t *order, int32_t *gid, int32_t *seg_start, int64_t *ukeys,
                        int32_t *ngroups, int max_groups, void *ws, size_t ws_bytes, void *stream) {
  if (E < 0 || K <= 0 || !ngroups || !seg_start) return RAMP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (E == 0) {
    (void)hipMemsetAsync(ngroups, 0, sizeof(int32_t), st);
    (void)hipMemsetAsync(seg_start, 0, sizeof(int32_t), st);
    return RAMP_OK;
  }
  if (!a || !order || !ws) return RAMP_EINVAL;
  if (ws_bytes < ramp_group_by_small_workspace_bytes(E, K)) return RAMP_EWORKSPACE;
  char *base = (char *)ws;
  int32_t *hist = (int32_t *)base;                 // [K + 1] counts, then the out-of-range flag: one memset for both
  int32_t *bad = hist + (K + 1);
  int32_t *gidmap = (int32_t *)(base + align_up((size_t)(K + 2) * 4, 256));
  int32_t *tmp = (int32_t *)(base + 2 * align_up((size_t)(K + 2) * 4, 256));
  (void)hipMemsetAsync(hist, 0, (size_t)(K + 2) * 4, st);
  const int nb = ramp_cdiv(E, 256);
  const bool lds = K <= GBC_LDS_K;
  if (lds)
    hipLaunchKernelGGL(gbc_hist_kernel<true>, dim3(nb), dim3(256), 0, st, a, b, (long)mul, (long)sub, hist, E, K, bad);
  else
    hipLaunchKernelGGL(gbc_hist_kernel<false>, dim3(nb), dim3(256), 0, st, a, b, (long)mul, (long)sub, hist, E, K, bad);
  hipLaunchKernelGGL(gbc_scan_kernel, dim3(1), dim3(1024), 0, st, hist, gidmap, seg_start, ukeys, ngroups, K, E,
                     (long)sub, 0L);
  if (lds)
    hipLaunchKernelGGL(gbc_scatter_kernel<true>, dim3(nb), dim3(256), 0, st, a, b, (long)mul, (long)sub, hist, gidmap,
                       tmp, gid, E, K);
  else
    hipLaunchKernelGGL(gbc_scatter_kernel<false>, dim3(nb), dim3(256), 0, st, a, b, (long)mul, (long)sub, hist, gidmap,
                       tmp, gid, E, K);
  const int ng = max_groups > 0 ? (max_groups < E ? max_groups : E) : (K < E ? K : E);
  hipLaunchKernelGGL(gbc_segsort_kernel, dim3(ng), dim3(64), 0, st, tmp, seg_start, ngroups, order);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_neighbors_from_groups(const int32_t *order, const int32_t *seg_start, const int32_t *ngroups,
                               const int64_t *jj, int64_t *ix, int64_t *jx, int32_t *kj_order, int E, int max_groups,
                               void *stream) {
  if (E < 0 || max_groups < 0) return RAMP_EINVAL;
  if (E == 0 || max_groups == 0) return RAMP_OK;
  if (!order || !seg_start || !ngroups || !jj || !ix || !jx) return RAMP_EINVAL;
  hipLaunchKernelGGL(nb_from_groups_kernel, dim3(max_groups), dim3(64), 0, (hipStream_t)stream, order,
                     seg_start, ngroups, jj, ix, jx, kj_order);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

size_t ramp_group_by_workspace_bytes(int E) { return ramp_internal_group_by_ws(E); }

int ramp_group_by(const int64_t *keys, int E, int64_t key_bound, int32_t *order, int32_t *gid,
                  int32_t *seg_start, int64_t *ukeys, int32_t *ngroups, void *ws,
                  size_t ws_bytes, void *stream) {
  return ramp_internal_group_by(keys, E, key_bound, order, gid, seg_start, ukeys, ngroups, ws,
                                ws_bytes, (hipStream_t)stream);
}

size_t ramp_neighbors_workspace_bytes(int E) {
  const size_t n = (size_t)(E > 0 ? E : 1);
  return ramp_internal_group_by_ws(E) + 2 * align_up(n * 4, 256);
}

int ramp_neighbors(const int64_t *kk, const int64_t *jj, int64_t *ix, int64_t *jx, int E,
                   int64_t kk_bound, int64_t jj_bound, void *ws, size_t ws_bytes,
                   void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!kk || !jj || !ix || !jx || !ws) return RAMP_EINVAL;
  if (ws_bytes < ramp_neighbors_workspace_bytes(E)) return RAMP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  GbWs w;
  const size_t used = gb_carve(ws, E, &w);
  int32_t *order1 = (int32_t *)((char *)ws + used);
  int32_t *order2 = (int32_t *)((char *)ws + used + align_up((size_t)E * 4, 256));
  const int nb = ramp_cdiv(E, GR_THREADS);
  size_t cb;
  const int bk = key_bits(kk_bound), bj = key_bits(jj_bound);
  if (kk_bound > 0 && jj_bound > 0 && bk + bj <= 62) {
    // one pass: composite key (kk, jj); radix sort is stable in the edge index
    hipLaunchKernelGGL(nb_key_kernel, dim3(nb), dim3(GR_THREADS), 0, st, kk, jj, w.k_in, w.iota, E,
                       (long long)jj_bound);
    cb = w.cub_bytes;
    // kk*jj_bound + jj < kk_bound*jj_bound <= 2^(bk+bj)
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.k_in, w.k_out, w.iota, order2, E, 0,
                                           bk + bj, st) != hipSuccess)
      return RAMP_ELAUNCH;
  } else {
    // two stable passes: by jj, then by kk
    hipLaunchKernelGGL(gb_init_kernel, dim3(nb), dim3(GR_THREADS), 0, st, jj, w.k_in, w.iota, E);
    cb = w.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.k_in, w.k_out, w.iota, order1, E, 0, bj,
                                           st) != hipSuccess)
      return RAMP_ELAUNCH;
    hipLaunchKernelGGL(nb_gather_key_kernel, dim3(nb), dim3(GR_THREADS), 0, st, kk, order1, w.k_in,
                       E);
    cb = w.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.k_in, w.k_out, order1, order2, E, 0, bk,
                                           st) != hipSuccess)
      return RAMP_ELAUNCH;
  }
  hipLaunchKernelGGL(nb_link_kernel, dim3(nb), dim3(GR_THREADS), 0, st, kk, order2, ix, jx, E);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_segment_softmax_sum(const void *fx, const void *gx, const int32_t *order,
                             const int32_t *seg_start, const int32_t *ngroups, void *y, int E,
                             int C, int max_groups, int dtype, void *stream) {
  if (E < 0 || C <= 0 || max_groups < 0) return RAMP_EINVAL;
  if (E == 0 || max_groups == 0) return RAMP_OK;
  if (!fx || !gx || !order || !seg_start || !ngroups || !y) return RAMP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RAMP_F32)
    hipLaunchKernelGGL(seg_softmax_sum_kernel<float>, dim3(max_groups), dim3(128), 0, st,
                       (const float *)fx, (const float *)gx, order, seg_start, ngroups, (float *)y,
                       C);
  else if (dtype == RAMP_F16)
    hipLaunchKernelGGL(seg_softmax_sum_kernel<_Float16>, dim3(max_groups), dim3(128), 0, st,
                       (const _Float16 *)fx, (const _Float16 *)gx, order, seg_start, ngroups,
                       (_Float16 *)y, C);
  else
    return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

// HOST function (no GPU work): Ramp_vo.keyframe()'s edit of the factor graph for one outcome of the motion
// test (reference ramp/Ramp_vo.py:247-274, 203-208) in one pass over the host mirror: drop the factors of
// keyframe k_remove (k_remove < 0: none) and renumber the later frames / patches, then drop factors whose
// source frame is older than n_after - removal_window.  out = [4][cap] int64 rows (ii, jj, kk, state row),
// rows_in = the hidden-state row of each factor (NULL: identity).  Returns the number of factors kept.
int ramp_graph_edit_host(const int64_t *ii, const int64_t *jj, const int64_t *kk, const int64_t *rows_in, int E,
                         int M, int k_remove, int n_after, int removal_window, int64_t *out, int cap,
                         int64_t *ranges) {
  if (E < 0 || !out || cap < E || (E > 0 && (!ii || !jj || !kk))) return RAMP_EINVAL;
  int64_t kmin = INT64_MAX, kmax = INT64_MIN, fmin = INT64_MAX, fmax = INT64_MIN;
  int64_t *oi = out, *oj = out + cap, *ok = out + 2 * (size_t)cap, *orow = out + 3 * (size_t)cap;
  const int64_t oldest = (int64_t)n_after - removal_window;
  // kk >= 0: kk / M < oldest  <=>  kk < oldest * M (never when oldest <= 0); no division, no branches in the loop
  const int64_t kcut = oldest > 0 ? oldest * (int64_t)M : 0;
  const int64_t kr = k_remove >= 0 ? (int64_t)k_remove : INT64_MAX;
  int m = 0;
  for (int e = 0; e < E; e++) {
    int64_t i = ii[e], j = jj[e], q = kk[e];
    const bool hit = (i == kr) | (j == kr);
    const int64_t gi = i > kr, gj = j > kr;
    i -= gi; q -= gi * M; j -= gj;
    const bool keep = !hit & (q >= kcut);
    oi[m] = i; oj[m] = j; ok[m] = q; orow[m] = rows_in ? rows_in[e] : (int64_t)e;
    const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    kmin = (keep & (q < kmin)) ? q : kmin; kmax = (keep & (q > kmax)) ? q : kmax;
    fmin = (keep & (lo < fmin)) ? lo : fmin; fmax = (keep & (hi > fmax)) ? hi : fmax;
    m += keep;
  }
  if (ranges) { ranges[0] = kmin; ranges[1] = kmax; ranges[2] = fmin; ranges[3] = fmax; }   // of the kept factors
  return m;
}

}  // extern "C"
