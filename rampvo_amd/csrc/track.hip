// The device-resident tracking step: Ramp_vo.__call__ in steady state as one host call (include/ramp_hip.h,
// "device-resident tracking step").
//
// The reference takes the keyframe decision on the host (ramp/Ramp_vo.py:237-246: two `.item()` reads), so every frame
// waits for the GPU to drain and the GPU then waits for the host to edit the factor graph.  Here
//   * trk_flag / trk_decide / trk_apply take the decision and edit the graph on the device: keep mask -> per-workgroup
//     counts -> scan -> stable compaction into the other half of the double-buffered factor list, the next frame's new
//     factors appended in closed form (Ramp_vo.py:312-325), the keyframe's rows shifted out of the per-frame buffers
//     by the same launch, the (t0, dP) pair the trajectory interpolation needs (:250-253) appended to a log;
//   * every size that depends on a decision -- the row a frame is stored to, the factor count, BA's window -- lives in
//     the int32 block `dyn` and is read by the kernels (their grids are capacity bounds);
//   * ramp_track_step enqueues the whole frame (frame stores, reprojection, correlation, update operator, two BA
//     iterations, point cloud, motion test, graph edit, the next graph's plan) without reading anything back.
#include "ramp_device.h"
#include "ramp_internal.h"
#include <string.h>
#include <stdlib.h>

#ifndef TRK_EB
#define TRK_EB 1024        // factors per workgroup of the edit kernels (256 threads x 4 passes)
#endif
#define TRK_MAXBUF 10

struct TrkEdit {
  int32_t *dyn;
  const float *mm;
  const int64_t *gin;      // [4][E_cap]
  int64_t *gout;
  int E_cap, M, r, removal_window, keyframe_index, log_cap, n_rows, pad;
  double thresh;
  int32_t *cnt, *off, *fmin;   // per edit workgroup: kept factors, their exclusive prefix, lowest frame index kept
  const int64_t *tstamps;
  const float *poses;
  float *dlog;
  int64_t counter;
  int nb;                  // edit workgroups = ceil(E_cap / TRK_EB)
  // row shift of a dropped keyframe (ramp/Ramp_vo.py:259-271)
  char *base[TRK_MAXBUF];
  long row_bytes[TRK_MAXBUF];
  int mod[TRK_MAXBUF];
  int nbuf;
  int32_t *slot_tab;       // optional (ramp_track.fmap1_slot): the rows of buffer slot_buf are not moved, the table is rotated
  int slot_buf, slot_mod;
};
// what a launch reads and writes (one indirection kept from the speculative-edit experiment of round 5, tools/shelved/)
struct TrkCand {
  const int32_t *dyn_in;   // sizes before the edit
  int32_t *dyn;            // sizes after
  int64_t *gout;
  int32_t *cnt, *off, *fmin;
};
__device__ __forceinline__ TrkCand trk_cand(const TrkEdit &p) {
  TrkCand c;
  c.dyn_in = p.dyn; c.dyn = p.dyn; c.gout = p.gout; c.cnt = p.cnt; c.off = p.off; c.fmin = p.fmin;
  return c;
}

// Ramp_vo.keyframe(): m = motionmag(i, j) + motionmag(j, i);  m / 2 < KEYFRAME_THRESH  (python floats: doubles)
__device__ __forceinline__ bool trk_remove(const float *mm, double thresh) {
  const double m = (double)mm[0] + (double)mm[1];
  return m / 2 < thresh;
}
// one factor through keyframe()'s edit (Ramp_vo.py:255-274): dropped if it touches keyframe k, renumbered if it lies
// behind it, dropped if its source frame left the removal window.  Same arithmetic as ramp_graph_edit_host.
__device__ __forceinline__ bool trk_edge(bool remove, long k, long kcut, int M, long &i, long &j, long &q) {
  const bool hit = remove && (i == k || j == k);
  const long gi = (remove && i > k) ? 1 : 0, gj = (remove && j > k) ? 1 : 0;
  i -= gi; q -= gi * M; j -= gj;
  return !hit && q >= kcut;
}
struct TrkDecision { bool remove; long k, kcut; int n, n_after; };
__device__ __forceinline__ TrkDecision trk_decision(const int32_t *dyn, const float *mm, double thresh, int ki,
                                                    int removal_window, int M, bool after_decide) {
  TrkDecision d;
  if (after_decide) {              // trk_decide has already rewritten dyn
    d.n = dyn[RAMP_DYN_NPREV];
    d.remove = dyn[RAMP_DYN_REMOVED] != 0;
  } else {
    d.n = dyn[RAMP_DYN_N];
    d.remove = trk_remove(mm, thresh);
  }
  d.k = d.n - ki;
  d.n_after = d.remove ? d.n - 1 : d.n;
  const long oldest = (long)d.n_after - removal_window;
  d.kcut = oldest > 0 ? oldest * M : 0;
  return d;
}

__global__ void __launch_bounds__(256) trk_flag_kernel(const TrkEdit p) {
  __shared__ int s_cnt[4], s_min[4];
  const TrkCand c = trk_cand(p);
  const int E = c.dyn_in[RAMP_DYN_E];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (b * TRK_EB >= E) return;
  const TrkDecision d = trk_decision(c.dyn_in, p.mm, p.thresh, p.keyframe_index, p.removal_window, p.M, false);
  int cnt = 0, fmin = 0x7fffffff;
#pragma unroll
  for (int pass = 0; pass < TRK_EB / 256; pass++) {
    const int e = b * TRK_EB + pass * 256 + tid;
    bool keep = false;
    if (e < E) {
      long i = p.gin[e], j = p.gin[(size_t)p.E_cap + e], q = p.gin[2 * (size_t)p.E_cap + e];
      keep = trk_edge(d.remove, d.k, d.kcut, p.M, i, j, q);
      if (keep) fmin = min(fmin, (int)min(i, j));
    }
    cnt += __popcll(__ballot(keep));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) fmin = min(fmin, __shfl_xor(fmin, o, 64));
  if (lane == 0) { s_cnt[wave] = cnt; s_min[wave] = fmin; }
  __syncthreads();
  if (tid == 0) {
    c.cnt[b] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    c.fmin[b] = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
  }
}

__global__ void __launch_bounds__(256) trk_decide_kernel(const TrkEdit p) {
  __shared__ int s_sum[256], s_min[256];
  const int tid = threadIdx.x;
  const TrkCand c = trk_cand(p);
  const int E = c.dyn_in[RAMP_DYN_E];
  const int nbl = (E + TRK_EB - 1) / TRK_EB;
  const int per = (nbl + 255) / 256;
  const int b0 = tid * per, b1 = min(nbl, b0 + per);
  int kept = 0, fm = 0x7fffffff;
  for (int b = b0; b < b1; b++) { kept += c.cnt[b]; fm = min(fm, c.fmin[b]); }
  s_sum[tid] = kept; s_min[tid] = fm;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int v = 0;
    if (tid >= o) v = s_sum[tid - o];
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  int run = s_sum[tid] - kept;
  for (int b = b0; b < b1; b++) { const int h = c.cnt[b]; c.off[b] = run; run += h; }
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) s_min[tid] = min(s_min[tid], s_min[tid + o]);
    __syncthreads();
  }
  if (tid != 0) return;
  const int Ek = s_sum[255];
  const TrkDecision d = trk_decision(c.dyn_in, p.mm, p.thresh, p.keyframe_index, p.removal_window, p.M, false);
  int status = 0;
  if (d.remove) {
    // Ramp_vo.py:249-253: delta[t1] = (t0, poses[k] * poses[k-1]^-1), read back by terminate()
    const int idx = p.dyn[RAMP_DYN_NLOG];
    if (idx < p.log_cap) {
      float Pk[7], Pm[7], Pi[7], dP[7];
      for (int q = 0; q < 7; q++) { Pk[q] = p.poses[7 * d.k + q]; Pm[q] = p.poses[7 * (d.k - 1) + q]; }
      lt_inv(Pm, Pi);
      lt_mul(Pk, Pi, dP);
      float *o = p.dlog + (size_t)idx * RAMP_TRACK_LOG;
      o[0] = __int_as_float((int)p.tstamps[d.k]);
      o[1] = __int_as_float((int)p.tstamps[d.k - 1]);
      for (int q = 0; q < 7; q++) o[2 + q] = dP[q];
      p.dyn[RAMP_DYN_NLOG] = idx + 1;
    } else {
      status |= 16;
    }
  }
  // the next frame's factors (Ramp_vo.py:312-325 with n = n_after + 1)
  // the frame buffers hold n_rows frames (row n1 - 1 = n_after is the next frame's; index_map is written at n_after + 1):
  // a full buffer is flagged and the counts stay in bounds -- the reference raises IndexError there; the host reads
  // the flag within a few frames (track_dev.py bounds its run-ahead) and raises
  int n_after = d.n_after;
  if (n_after > p.n_rows - 2) { status |= 64; n_after = p.n_rows - 2; }
  const int n1 = n_after + 1;
  const int lo = max(n1 - p.r, 0);
  const int nf = p.M * (max(n1 - 1, 0) - lo), nbk = p.M * (n1 - lo);
  int ne = nf + nbk;
  if (Ek + ne > p.E_cap) { status |= 4; ne = max(p.E_cap - Ek, 0); }
  const int flo = min(s_min[0], lo);
  c.dyn[RAMP_DYN_NPREV] = d.n;
  c.dyn[RAMP_DYN_EPREV] = E;
  c.dyn[RAMP_DYN_EKEPT] = Ek;
  c.dyn[RAMP_DYN_REMOVED] = d.remove ? 1 : 0;
  c.dyn[RAMP_DYN_K] = (int)d.k;
  c.dyn[RAMP_DYN_NROW] = n_after;
  c.dyn[RAMP_DYN_N] = n1;
  c.dyn[RAMP_DYN_E] = Ek + ne;
  c.dyn[RAMP_DYN_KLO] = (int)min(d.kcut, (long)p.M * lo);
  c.dyn[RAMP_DYN_FLO] = flo;
  c.dyn[RAMP_DYN_W] = n1 - flo;
  c.dyn[RAMP_DYN_FRAME] = (int)p.counter;
  c.dyn[RAMP_DYN_FRAME2] = (int)p.counter;       // second tag, in the other half of the block: a torn host copy shows
  if (status) atomicOr(c.dyn + RAMP_DYN_STATUS, status);   // (the front-end stream's gate wait may OR its time-out bit in)
}

// grid (nb + new-factor workgroups, 1 + nbuf).  y = 0: x < nb compacts the kept factors of edit workgroup x (stable),
// the rest appends the next frame's factors; y > 0: row shift of buffer y - 1 if the keyframe was dropped.
// rows k + 1 .. k + R of a frame buffer moved down by one (ring rows taken modulo `mod`): a thread reads its 16-byte column
// of all R rows, then writes them
template <int R>
__device__ __forceinline__ void trk_shift_rows(char *base, long row_bytes, int mod, int k, long n16, int tid) {
  const uint4 *src[R];
  uint4 *dst[R];
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int ss = mod ? (k + u + 1) % mod : k + u + 1, sd = mod ? (k + u) % mod : k + u;
    src[u] = reinterpret_cast<const uint4 *>(base + (size_t)ss * row_bytes);
    dst[u] = reinterpret_cast<uint4 *>(base + (size_t)sd * row_bytes);
  }
  for (long col = (long)blockIdx.x * 256 + tid; col < n16; col += (long)gridDim.x * 256) {
    uint4 v[R];
#pragma unroll
    for (int u = 0; u < R; u++) v[u] = src[u][col];
#pragma unroll
    for (int u = 0; u < R; u++) dst[u][col] = v[u];
  }
}

// the row shift of frame buffer b behind a dropped keyframe k (rows k + 1 .. nrows - 1 move down by one)
__device__ __forceinline__ void trk_shift_buffer(const TrkEdit &p, int b, int k, int nrows, int tid) {
  if (p.slot_tab && b == p.slot_buf) {
    // rows k + 1 .. nrows - 1 become rows k .. nrows - 2: their SLOTS move down the table, the dropped row's slot goes
    // behind them (the next new frame's).  One thread; readers come in later launches.
    if (blockIdx.x == 0 && tid == 0) {
      const int freed = p.slot_tab[k % p.slot_mod];
      for (int r = k; r < nrows - 1; r++) p.slot_tab[r % p.slot_mod] = p.slot_tab[(r + 1) % p.slot_mod];
      p.slot_tab[(nrows - 1) % p.slot_mod] = freed;
    }
    return;
  }
  const long n4 = p.row_bytes[b] / 4;
  // each thread owns columns c, c + stride, ... and moves the rows itself: no cross-thread hazard.  Up to four rows
  // (KEYFRAME_INDEX - 1 = 3 in every shipped config) are all READ before the first is written -- one round trip
  // instead of a chain of load -> store pairs per column; 16-byte pieces where the row allows
  const int nmove = nrows - 1 - k;
  const int mod = p.mod[b];
  if (nmove >= 1 && nmove <= 4 && !(p.row_bytes[b] & 15)) {
    const long n16 = p.row_bytes[b] / 16;
    switch (nmove) {                                         // (compile-time row count: the rows stay in registers)
      case 1: trk_shift_rows<1>(p.base[b], p.row_bytes[b], mod, k, n16, tid); break;
      case 2: trk_shift_rows<2>(p.base[b], p.row_bytes[b], mod, k, n16, tid); break;
      case 3: trk_shift_rows<3>(p.base[b], p.row_bytes[b], mod, k, n16, tid); break;
      default: trk_shift_rows<4>(p.base[b], p.row_bytes[b], mod, k, n16, tid); break;
    }
    return;
  }
  if (p.row_bytes[b] & 3) {                                  // (an odd PATCHES_PER_FRAME's colour rows: byte by byte)
    for (long col = (long)blockIdx.x * 256 + tid; col < p.row_bytes[b]; col += (long)gridDim.x * 256)
      for (int r = k; r < nrows - 1; r++) {
        const int sd = mod ? r % mod : r, ss = mod ? (r + 1) % mod : r + 1;
        p.base[b][(size_t)sd * p.row_bytes[b] + col] = p.base[b][(size_t)ss * p.row_bytes[b] + col];
      }
    return;
  }
  for (long col = (long)blockIdx.x * 256 + tid; col < n4; col += (long)gridDim.x * 256) {
    for (int r = k; r < nrows - 1; r++) {
      const int sd = mod ? r % mod : r, ss = mod ? (r + 1) % mod : r + 1;
      reinterpret_cast<uint32_t *>(p.base[b] + (size_t)sd * p.row_bytes[b])[col] =
          reinterpret_cast<const uint32_t *>(p.base[b] + (size_t)ss * p.row_bytes[b])[col];
    }
  }
}

__global__ void __launch_bounds__(256) trk_apply_kernel(const TrkEdit p) {
  const int tid = threadIdx.x;
  if (blockIdx.y > 0) {
    if (!p.dyn[RAMP_DYN_REMOVED]) return;
    trk_shift_buffer(p, blockIdx.y - 1, p.dyn[RAMP_DYN_K], p.dyn[RAMP_DYN_NPREV], tid);
    return;
  }
  const TrkCand c = trk_cand(p);
  int64_t *oi = c.gout, *oj = c.gout + p.E_cap, *ok = c.gout + 2 * (size_t)p.E_cap, *orow = c.gout + 3 * (size_t)p.E_cap;
  const int b = blockIdx.x;
  if (b < p.nb) {
    __shared__ int s_w[4];
    const int E = c.dyn[RAMP_DYN_EPREV];
    if (b * TRK_EB >= E) return;
    const TrkDecision d = trk_decision(c.dyn, p.mm, p.thresh, p.keyframe_index, p.removal_window, p.M, true);
    const int lane = tid & 63, wave = tid >> 6;
    int base = c.off[b];
    for (int pass = 0; pass < TRK_EB / 256; pass++) {
      const int e = b * TRK_EB + pass * 256 + tid;
      bool keep = false;
      long i = 0, j = 0, q = 0;
      if (e < E) {
        i = p.gin[e]; j = p.gin[(size_t)p.E_cap + e]; q = p.gin[2 * (size_t)p.E_cap + e];
        keep = trk_edge(d.remove, d.k, d.kcut, p.M, i, j, q);
      }
      const unsigned long long m = __ballot(keep);
      __syncthreads();                                   // s_w of the previous pass has been read
      if (lane == 0) s_w[wave] = __popcll(m);
      __syncthreads();
      int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      for (int w = 0; w < wave; w++) pos += s_w[w];
      if (keep) { oi[pos] = i; oj[pos] = j; ok[pos] = q; orow[pos] = e; }
      base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    return;
  }
  // new factors: forward (every live patch of the last r - 1 frames -> the new frame), then backward (the new frame's
  // patches -> the last r frames incl. itself), patch-major like the reference's meshgrid(indexing='ij')
  const int Ek = c.dyn[RAMP_DYN_EKEPT], ne = c.dyn[RAMP_DYN_E] - Ek;
  const int idx = (b - p.nb) * 256 + tid;
  if (idx >= ne) {
    // rows between the live count and the next step's launch bound: defined (harmless) entries -- a caller that runs its
    // own operator on E_bound rows (the fp32 path) gathers through them
    if (idx < ne + p.pad && Ek + idx < p.E_cap) { oi[Ek + idx] = 0; oj[Ek + idx] = 0; ok[Ek + idx] = 0; orow[Ek + idx] = -1; }
    return;
  }
  const int n1 = c.dyn[RAMP_DYN_N];
  const int lo = max(n1 - p.r, 0);
  const int nf = p.M * (max(n1 - 1, 0) - lo);
  long kk, jj;
  if (idx < nf) {
    kk = (long)p.M * lo + idx;
    jj = n1 - 1;
  } else {
    const int q = idx - nf, nt = n1 - lo;
    kk = (long)p.M * (n1 - 1) + q / nt;
    jj = lo + q % nt;
  }
  oi[Ek + idx] = kk / p.M; oj[Ek + idx] = jj; ok[Ek + idx] = kk; orow[Ek + idx] = -1;
}

__global__ void __launch_bounds__(256) trk_iota_kernel(int64_t *__restrict__ row, int32_t *__restrict__ dyn) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e == 0) dyn[RAMP_DYN_MEDOK] = 0;           // bundle adjustment moved the depths and no motion test follows
  if (e < dyn[RAMP_DYN_E]) row[e] = e;
}

// (measurement only, RAMP_TRACK_WRAP_COORDS) every factor's reprojection moved into the target plane by whole plane
// widths / heights (the patch keeps its shape): with random-init weights a third to a half of the projections leave
// the image and cost the correlation kernel a row of zeros and no gathers; bench.py's "live" roofline leg times the
// kernel with every factor gathering
// RAMP_TRACK_COMPACT_COORDS: in addition the patch is given unit pixel spacing around its (wrapped) centre -- the factors
// of a converged tracker (a patch reprojects to about its own 3 x 3 footprint: one 10 x 10 union window per level); the
// random-weight tracker's own patches spread over tens of pixels and take the kernel's nine-separate-windows path.
__global__ void __launch_bounds__(256) trk_wrap_coords_kernel(float *__restrict__ coords, const int32_t *__restrict__ dyn,
                                                              int P, float w, float h, int compact) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= dyn[RAMP_DYN_E]) return;
  const int PP = P * P;
  float *c = coords + (size_t)e * 2 * PP;
  const float cx = c[PP / 2], cy = c[PP + PP / 2];
  const bool ok = fabsf(cx) < 1e8f && fabsf(cy) < 1e8f;       // (false for NaN / inf as well)
  const float sx = ok ? floorf(cx / w) * w : 0.f, sy = ok ? floorf(cy / h) * h : 0.f;
  const float mx = ok ? cx - sx : 0.5f * w, my = ok ? cy - sy : 0.5f * h;
  for (int p = 0; p < PP; p++) {
    if (compact) {
      c[p] = mx + (float)(p % P - P / 2);
      c[PP + p] = my + (float)(p / P - P / 2);
    } else {
      c[p] = ok ? c[p] - sx : mx;
      c[PP + p] = ok ? c[PP + p] - sy : my;
    }
  }
}

static int trk_edit_fill(const ramp_track *t, int cur, int64_t counter, TrkEdit &p) {
  p.dyn = t->dyn; p.mm = t->mm; p.gin = t->graph[cur]; p.gout = t->graph[1 - cur];
  p.E_cap = t->E_cap; p.M = t->M; p.r = t->patch_lifetime; p.removal_window = t->removal_window;
  p.keyframe_index = t->keyframe_index; p.log_cap = t->log_cap; p.thresh = t->keyframe_thresh; p.n_rows = t->n_rows;
  p.pad = 4 * (2 * t->patch_lifetime - 1) * t->M;
  p.nb = ramp_cdiv(t->E_cap, TRK_EB);
  p.cnt = t->edit_ws; p.off = t->edit_ws + p.nb; p.fmin = t->edit_ws + 2 * p.nb;
  p.tstamps = t->tstamps; p.poses = t->poses; p.dlog = t->dlog; p.counter = counter;
  const int PP = t->P * t->P;
  const long es = t->feat_fp32 ? 4 : 2;                // bytes per feature element
  const long fb1 = (long)t->feat_h * t->feat_w * 128 * es, fb2 = (long)(t->feat_h / 4) * (t->feat_w / 4) * 128 * es;
  void *bufs[9] = {t->tstamps, t->colors, t->poses, t->patches, t->intrinsics, t->imap, t->gmap, t->fmap1, t->fmap2};
  const long rb[9] = {8, (long)t->M * 3, 28, (long)t->M * 3 * PP * 4, 16, (long)t->M * 384 * es, (long)t->M * PP * 128 * es, fb1, fb2};
  const int md[9] = {0, 0, 0, 0, 0, t->mem, t->mem, t->mem, t->mem};
  p.nbuf = 9;
  for (int i = 0; i < 9; i++) {
    if (!bufs[i]) return RAMP_EINVAL;
    p.base[i] = (char *)bufs[i]; p.row_bytes[i] = rb[i]; p.mod[i] = md[i];
  }
  p.slot_tab = t->fmap1_slot; p.slot_buf = 7; p.slot_mod = t->mem;      // (bufs[7] = fmap1)
  return RAMP_OK;
}

static bool trk_valid(const ramp_track *t) {
  return t && t->dyn && t->M > 0 && t->P == 3 && t->E_cap > 0 && t->graph[0] && t->graph[1] && t->plan_ws &&
         t->kk_order && t->ij_order && t->ix && t->jx && t->kj && t->opt_window > 0 && t->mem > 0 && t->edit_ws;
}

// Reads the correlation planes of the frames in the window (and the patch features) once, so that the next step's
// correlation kernel finds them in the memory-side cache: between two correlation launches the update operator
// streams ~1.5 GB through L2 / MALL, and the kernel's window gathers then pay HBM latency (174 us against 117 us on
// the same factors with the planes resident; tools/corr_window_stats.py).  Runs on the front-end stream in the slack
// behind the front end, next to the previous step's bundle adjustment.
__global__ void __launch_bounds__(256) trk_warm_kernel(const uint4 *__restrict__ fmap1, const uint4 *__restrict__ fmap2,
                                                       const uint4 *__restrict__ gmap, long n1, long n2, long ng, int mem,
                                                       int frames, const int32_t *__restrict__ dyn, int32_t *sink,
                                                       const int32_t *__restrict__ slot1) {
  const int fi = blockIdx.y;                       // 0 .. frames - 1: frame n - 1 - fi; frames: the patch features
  const long stride = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  if (fi == frames) {
    for (long i = t0; i < ng; i += stride) { const uint4 v = gmap[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  } else {
    const int f = dyn[RAMP_DYN_N] - fi;            // (the keyframe test may be moving n by one right now: one frame of margin)
    if (f < 0) return;
    const int slot = f % mem;
    const uint4 *p1 = fmap1 + (size_t)(slot1 ? slot1[slot] : slot) * n1, *p2 = fmap2 + (size_t)slot * n2;
    for (long i = t0; i < n1; i += stride) { const uint4 v = p1[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    for (long i = t0; i < n2; i += stride) { const uint4 v = p2[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x9e3779b9u) *sink = (int)acc;        // (keeps the loads)
}

// One wave that sleeps: holds the front-end stream back for `ticks` of the 100 MHz wall clock.  The next frame's LSTM
// launch should reach the chip ~70 us behind the gru launch, not ~40: arriving while the gru launch is still filling
// the CUs it takes them first, and gru finishes 45 us later (tools/corun_gru_lstm.py)
__global__ void trk_delay_kernel(long ticks) {
  const long t0 = wall_clock64();
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// One sleeping wave that ends when *flag >= value (or after `ticks` of the 100 MHz clock), then sleeps `after` more.  It
// looks at the word only every ~3 us: a wave that polls a system-scope word without pauses slows the other streams'
// kernels down (tools/mb/stream_signal.hip: 64 MB copies 20.8 -> 22.3 us; the update operator 505 -> 608 us).
__global__ void trk_wait_flag_kernel(const uint32_t *flag, uint32_t value, long ticks, long after, int nap, int32_t *status) {
  const long t0 = wall_clock64();
  bool seen;
  while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= value) && (long)wall_clock64() - t0 < ticks)
    for (int i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(64);
  // gave up: what follows on this stream is no longer ordered behind the producer -- sticky status bit 128, the tracker
  // raises when it reads it (the wave still ends, so the stream cannot hang on a producer that never comes)
  if (!seen && status && threadIdx.x == 0) atomicOr(status, 128);
  const long t1 = wall_clock64();
  while ((long)wall_clock64() - t1 < after) __builtin_amdgcn_s_sleep(64);
}

extern "C" {

int ramp_stream_wait_flag(void *stream, const uint32_t *flag, uint32_t value, int timeout_us, int then_delay_us, int32_t *status) {
  if (!flag || timeout_us <= 0 || then_delay_us < 0) return RAMP_EINVAL;
  const int nap = 2;                                   // s_sleep(64) units (~1.7 us each) between two looks at the word
  hipLaunchKernelGGL(trk_wait_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, (long)timeout_us * 100,
                     (long)then_delay_us * 100, nap, status);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_signal_alloc(uint32_t **flag) {
  if (!flag) return RAMP_EINVAL;
  void *p = nullptr;
  if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess) return RAMP_EUNSUPPORTED;
  if (hipMemset(p, 0, 8) != hipSuccess) { (void)hipFree(p); return RAMP_ELAUNCH; }
  *flag = (uint32_t *)p;
  return RAMP_OK;
}
int ramp_signal_free(uint32_t *flag) { return (!flag || hipFree(flag) == hipSuccess) ? RAMP_OK : RAMP_ELAUNCH; }

__global__ void trk_signal_kernel(uint32_t *flag, uint32_t value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int ramp_stream_signal(void *stream, uint32_t *flag, uint32_t value) {
  if (!flag) return RAMP_EINVAL;
  hipLaunchKernelGGL(trk_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_stream_delay(int microseconds, void *stream) {
  if (microseconds <= 0) return RAMP_OK;
  hipLaunchKernelGGL(trk_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)microseconds * 100);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_track_warm(const ramp_track *t, int32_t *sink, void *stream) {
  if (!trk_valid(t) || !sink || !t->fmap1 || !t->fmap2 || !t->gmap) return RAMP_EINVAL;
  const long n1 = (long)t->feat_h * t->feat_w * 128 * 2 / 16, n2 = (long)(t->feat_h / 4) * (t->feat_w / 4) * 128 * 2 / 16;
  const long ng = (long)t->mem * t->M * t->P * t->P * 128 * 2 / 16;
  const int frames = t->removal_window + 3 < t->mem ? t->removal_window + 3 : t->mem;
  const int gx = 8;   // 8 x 26 workgroups: ~100 us of gentle streaming (64: the planes arrive sooner, the tail kernels slow down as much)
  hipLaunchKernelGGL(trk_warm_kernel, dim3(gx, frames + 1), dim3(256), 0, (hipStream_t)stream, (const uint4 *)t->fmap1,
                     (const uint4 *)t->fmap2, (const uint4 *)t->gmap, n1, n2, ng, t->mem, frames, t->dyn, sink, t->fmap1_slot);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

size_t ramp_track_sizeof(void) { return sizeof(ramp_track); }
int ramp_host_device_pointer(void *host, void **dev) {
  if (!host || !dev) return RAMP_EINVAL;
  void *dp = nullptr;
  if (hipHostGetDevicePointer(&dp, host, 0) != hipSuccess) { (void)hipGetLastError(); *dev = nullptr; return RAMP_EUNSUPPORTED; }
  *dev = dp;
  return RAMP_OK;
}
size_t ramp_track_plan_workspace_bytes(int E_cap, int kkey_cap, int pkey_cap) {
  return ramp_i_plan_dyn_ws(E_cap, kkey_cap, pkey_cap);
}
size_t ramp_track_ba_workspace_bytes(int E_cap, int n_rows, int M, int opt_window, int kk_cap, int ij_cap) {
  return ramp_i_ba_dyn_ws(E_cap, n_rows, n_rows * M, opt_window, kk_cap, ij_cap);
}

int ramp_track_plan(const ramp_track *t, int cur, void *stream) {
  if (!trk_valid(t) || cur < 0 || cur > 1) return RAMP_EINVAL;
  return ramp_i_plan_dyn(t->graph[cur], t->E_cap, t->E_cap, t->dyn, t->dyn + RAMP_DYN_STATUS, t->M, t->kkey_cap, t->pkey_cap,
                         t->kk_cap, t->ij_cap, t->kk_order, t->kk_gid, t->kk_seg, t->kk_ngroups, t->kk_ukeys, t->ij_order,
                         t->ij_gid, t->ij_seg, t->ij_ngroups, t->ij_ukeys, t->ix, t->jx, t->kj, t->plan_ws, t->plan_ws_bytes,
                         nullptr, (hipStream_t)stream);
}

#define TRK_PROBE(i)                                                                             \
  do {                                                                                           \
    if (t->probe[i] && hipEventRecord((hipEvent_t)t->probe[i], st) != hipSuccess) return RAMP_ELAUNCH; \
  } while (0)
#define TRK_DO(call)          \
  do {                        \
    const int rc_ = (call);   \
    if (rc_ != RAMP_OK) return rc_; \
  } while (0)

// the launch bound was below the live factor count: the frame would have been computed on a truncated graph
__global__ void trk_bound_check_kernel(int32_t *dyn, int E_bound) {
  if (dyn[RAMP_DYN_E] > E_bound) atomicOr(dyn + RAMP_DYN_STATUS, 32);
}

int ramp_track_step(const ramp_track *t, int cur, int64_t counter, int flags, int E_bound, const float *k_new,
                    void *gate_event, void *stream) {
  if (!trk_valid(t) || cur < 0 || cur > 1 || E_bound < 0) return RAMP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int Ec = t->E_cap, PP = t->P * t->P;
  // launch bound of the per-factor kernels: the caller's upper bound of dyn[RAMP_DYN_E] (0: the capacity)
  const int Eb = (E_bound > 0 && E_bound < Ec) ? E_bound : Ec;
  const int new_cap = (2 * t->patch_lifetime - 1) * t->M;          // factors one frame adds
  // (the bound check rides in the frame commit's launch when there is one)
  const int fold = 1;
  const bool folded = fold && (flags & RAMP_TRACK_COMMIT);
  if (Eb < Ec && !folded) hipLaunchKernelGGL(trk_bound_check_kernel, dim3(1), dim3(1), 0, st, t->dyn, Eb);
  const int64_t *g = t->graph[cur];
  const int64_t *ii = g, *jj = g + Ec, *kk = g + 2 * (size_t)Ec, *row = g + 3 * (size_t)Ec;
  const int32_t *dyn = t->dyn;
  bool pc_with_mm = false;
  // the host's lazy copy of the sizes: written by the plan's last launch straight into the (mapped, pinned) host buffer
  int32_t *mirror = nullptr;
  if (t->dyn_host) mirror = t->dyn_host_dev;      // resolved once by the caller (ramp_host_device_pointer)
  const int Ep_next = Eb + new_cap < Ec ? Eb + new_cap : Ec;       // the next graph: at most one frame's factors more
  if (flags & RAMP_TRACK_COMMIT) {
    if (!t->fe_colors || !t->fe_imap || !t->fe_gmap || !t->fe_fmap1 || !t->fe_fmap2 || !t->fe_patches) return RAMP_EINVAL;
    const void *src[5] = {t->fe_colors, t->fe_imap, t->fe_gmap, t->fe_fmap1, t->fe_fmap2};
    void *base[5] = {t->colors, t->imap, t->gmap, t->fmap1, t->fmap2};
    const long es = t->feat_fp32 ? 4 : 2;
    const long bytes[5] = {(long)t->M * 3, (long)t->M * 384 * es, (long)t->M * PP * 128 * es,
                           (long)t->feat_h * t->feat_w * 128 * es, (long)(t->feat_h / 4) * (t->feat_w / 4) * 128 * es};
    const int mod[5] = {0, t->mem, t->mem, t->mem, t->mem};
    TRK_DO(ramp_i_frame_commit_dyn(t->poses, t->motion_model, t->motion_damping, t->tstamps, counter, t->index_map,
                                   t->intrinsics, k_new, t->patches, 3, t->M, t->P, t->fe_patches, 5, src, base, bytes,
                                   mod, dyn, (t->median && t->keyframe_index >= 4) ? t->median : nullptr, t->dyn + RAMP_DYN_STATUS, (Eb < Ec && folded) ? Eb : 0, t->n_rows, st,
                                   t->fmap1_slot, 3, nullptr, 0));
  }
  // fp32 features: the correlation launch is corr_mfma_kernel<CorrX2> on planes of split fp16 pairs (feat_fp32 == 2: the
  // caller packed them so, RAMP_CORR_X2) or corr_mfma_kernel<float> (RAMP_CORR_F32_MFMA=0: corr_kernel<float>, the reference
  // kernel's summation order) with rows padded to 896 floats, the operator csrc/update_x3.hip's chains; PRE / POST around a
  // caller-run operator remain (RAMP_X3=0: library GEMMs)
  static int corr32_fast = -1;
  if (corr32_fast < 0) { const char *e = getenv("RAMP_CORR_F32_MFMA"); corr32_fast = e ? atoi(e) : 1; }
  if (t->feat_fp32 == 2 && t->feat_plain) return RAMP_EINVAL;
  const int f32_code = RAMP_F32 | (t->feat_fp32 == 2 ? RAMP_CORR_X2 : corr32_fast ? RAMP_CORR_MFMA32 : 0);
  if (flags & RAMP_TRACK_UPDATE_PRE) {
    if (!t->coords || !t->corr) return RAMP_EINVAL;
    TRK_DO(ramp_i_transform_dyn(t->poses, t->patches, t->intrinsics, ii, jj, kk, t->coords, Eb, dyn, st));
    ramp_corr_level lv[2];
    lv[0].fmap = t->fmap1; lv[0].H2 = t->feat_h; lv[0].W2 = t->feat_w; lv[0].coord_div = 1.0f;
    lv[1].fmap = t->fmap2; lv[1].H2 = t->feat_h / 4; lv[1].W2 = t->feat_w / 4; lv[1].coord_div = 4.0f;
    TRK_PROBE(0);
    if (t->feat_fp32)
      TRK_DO(ramp_i_corr_fwd(t->gmap, lv, 2, t->coords, kk, jj, t->ij_order, t->corr, 896, (long)t->M * t->mem, t->mem, Eb,
                             t->mem * t->M, t->mem, 128, t->P, 3, f32_code, t->feat_plain ? RAMP_NHWC : RAMP_NHWC32, dyn, st, nullptr, nullptr, nullptr, nullptr,
                             t->fmap1_slot));
    else
      TRK_DO(ramp_i_corr_fwd(t->gmap, lv, 2, t->coords, kk, jj, t->ij_order, t->corr, 896, (long)t->M * t->mem, t->mem, Eb,
                             t->mem * t->M, t->mem, 128, t->P, 3, RAMP_F16, t->feat_plain ? RAMP_NHWC : RAMP_NHWC32, dyn, st, nullptr, nullptr, nullptr, nullptr,
                             t->fmap1_slot));
    TRK_PROBE(1);
  }
  if (flags & RAMP_TRACK_UPDATE_POST) {
    if (!t->target || !t->weight || !t->ba_ws) return RAMP_EINVAL;
    TRK_PROBE(3);
    TRK_DO(ramp_i_ba_dyn(t->poses, t->patches, t->intrinsics, t->target, t->weight, t->lmbda, ii, jj, kk, Eb, t->P,
                         t->n_rows, t->n_rows * t->M, t->opt_window, 2, t->kk_order, t->kk_seg, t->kk_ngroups, t->kk_ukeys,
                         t->kk_cap, t->ij_order, t->ij_seg, t->ij_ngroups, t->ij_cap, t->ba_ws, t->ba_ws_bytes,
                         t->dyn + RAMP_DYN_STATUS, dyn, st));
    TRK_PROBE(4);
    pc_with_mm = t->points && t->ixm && (flags & RAMP_TRACK_KEYFRAME) && !(flags & RAMP_TRACK_MM_GIVEN) && t->mm;
    if (t->points && t->ixm && !pc_with_mm)
      TRK_DO(ramp_i_point_cloud_dyn(t->poses, t->patches, t->intrinsics, t->ixm, t->points, t->m_cap, dyn, t->M, st));
    if (!(flags & RAMP_TRACK_KEYFRAME))
      hipLaunchKernelGGL(trk_iota_kernel, dim3(ramp_cdiv(Eb, 256)), dim3(256), 0, st, t->graph[cur] + 3 * (size_t)Ec, t->dyn);
  }
  if (flags & RAMP_TRACK_UPDATE) {
    const ramp_track_weights &w = t->w;
    if (!t->coords || !t->corr || !t->net[0] || !t->net[1] || !t->net[2] || !t->fg || !t->ykk || !t->hkk || !t->yij ||
        !t->hij || !t->target || !t->weight || !t->ba_ws)
      return RAMP_EINVAL;
    // Ramp_vo.update(), ramp/Ramp_vo.py:276-310
    // (pops.transform as a launch of its own: riding in the correlation kernel's geometry prologue it was a wash -- the
    // nine-lane Lie algebra adds to every wave what the 7.8 us launch took, corr 158 -> 166 us)
    TRK_DO(ramp_i_transform_dyn(t->poses, t->patches, t->intrinsics, ii, jj, kk, t->coords, Eb, dyn, st));
    if (flags & (RAMP_TRACK_WRAP_COORDS | RAMP_TRACK_COMPACT_COORDS))
      hipLaunchKernelGGL(trk_wrap_coords_kernel, dim3(ramp_cdiv(Eb, 256)), dim3(256), 0, st, t->coords, dyn, t->P,
                         (float)t->feat_w, (float)t->feat_h, (flags & RAMP_TRACK_COMPACT_COORDS) ? 1 : 0);
    ramp_corr_level lv[2];
    lv[0].fmap = t->fmap1; lv[0].H2 = t->feat_h; lv[0].W2 = t->feat_w; lv[0].coord_div = 1.0f;
    lv[1].fmap = t->fmap2; lv[1].H2 = t->feat_h / 4; lv[1].W2 = t->feat_w / 4; lv[1].coord_div = 4.0f;
    TRK_PROBE(0);
    static int gate_at = -1;
    if (gate_at < 0) { const char *e = getenv("RAMP_GATE_AT"); gate_at = e ? atoi(e) : 2; if (gate_at < 0 || gate_at > 3) gate_at = 2; }
    if (t->feat_fp32) {
      // ---- fp32 features (MIXED_PRECISION off): the same step with csrc/update_x3.hip's chains (Linear layers on the f16
      // matrix cores from split fp32 operands), fp32 tables, [f | g] rows + segment softmax + h for the two SoftAggs.
      // The fp32 front end is ~1.1 ms against ~0.9 ms of step behind the second neighbour chain, so it starts at the top of the
      // step (RAMP_GATE_AT_F32: 5 = before the correlation launch, the default; 4 = before the correlation MLP, 3 = before
      // c1 / c2, 0 .. 2 as RAMP_GATE_AT).  Measured (bench.py --mixed 0, two runs each): 2 -> 434 kf/s, 3 -> 455, 4 -> 471,
      // 5 -> 492
      static int gate32 = -1;
      if (gate32 < 0) { const char *e = getenv("RAMP_GATE_AT_F32"); gate32 = e ? atoi(e) : 5; if (gate32 < 0 || gate32 > 5) gate32 = 5; }
      const int gate_at = gate32;
#define TRK_GATE32(pos)                                                                                   \
  do {                                                                                                    \
    if (gate_at == (pos) && !(t->gate_flag && (pos) == 0)) {                                              \
      if (t->gate_flag) hipLaunchKernelGGL(trk_signal_kernel, dim3(1), dim3(1), 0, st, t->gate_flag, t->gate_seq); \
      else if (gate_event && hipEventRecord((hipEvent_t)gate_event, st) != hipSuccess) return RAMP_ELAUNCH; \
    }                                                                                                     \
  } while (0)
      TRK_GATE32(5);
      TRK_DO(ramp_i_corr_fwd(t->gmap, lv, 2, t->coords, kk, jj, t->ij_order, t->corr, 896, (long)t->M * t->mem, t->mem, Eb,
                             t->mem * t->M, t->mem, 128, t->P, 3, f32_code, t->feat_plain ? RAMP_NHWC : RAMP_NHWC32, dyn, st, nullptr, nullptr, nullptr,
                             nullptr, t->fmap1_slot));
      TRK_PROBE(1);
      TRK_GATE32(4);
      const float *corr32 = (const float *)t->corr;
      float *fg32 = (float *)t->fg, *ykk = (float *)t->ykk, *hkk = (float *)t->hkk, *yij = (float *)t->yij, *hij = (float *)t->hij;
      TRK_DO(ramp_i_x3_corr_mlp(corr32, 896, w.corr_w1, w.corr_b1, w.corr_w2, w.corr_b2, w.corr_w3, w.corr_b3, w.corr_ln_w,
                                w.corr_ln_b, w.corr_ln_eps, t->net[0], row, (const float *)t->imap, kk, (long)t->M * t->mem,
                                w.norm_w, w.norm_b, w.norm_eps, t->net[1], Eb, dyn, st));
      TRK_GATE32(3);
      TRK_DO(ramp_i_x3_nbr(t->net[1], t->ix, w.c1_wa, w.c1_ba, w.c1_wb, w.c1_bb, t->net[2], Eb, dyn, st));
      TRK_DO(ramp_i_x3_nbr(t->net[2], t->jx, w.c2_wa, w.c2_ba, w.c2_wb, w.c2_bb, t->net[1], Eb, dyn, st));
      float *net32 = t->net[1];
      TRK_GATE32(2);
      TRK_DO(ramp_i_x3_fg(net32, nullptr, nullptr, nullptr, w.kk_wf, w.kk_bf, w.kk_wg, w.kk_bg, fg32, Eb, dyn, st));
      TRK_DO(ramp_x3_segment_softmax(fg32, t->kk_order, t->kk_seg, t->kk_ngroups, ykk, t->kk_cap, stream));
      TRK_DO(ramp_x3_linear(ykk, w.kk_wh, w.kk_bh, hkk, t->kk_cap, t->kk_ngroups, stream));
      TRK_GATE32(1);
      TRK_DO(ramp_i_x3_fg(net32, hkk, t->kk_gid, nullptr, w.ij_wf, w.ij_bf, w.ij_wg, w.ij_bg, fg32, Eb, dyn, st));
      TRK_DO(ramp_x3_segment_softmax(fg32, t->ij_order, t->ij_seg, t->ij_ngroups, yij, t->ij_cap, stream));
      TRK_DO(ramp_x3_linear(yij, w.ij_wh, w.ij_bh, hij, t->ij_cap, t->ij_ngroups, stream));
      TRK_GATE32(0);
      TRK_DO(ramp_i_x3_gru(net32, hkk, t->kk_gid, hij, t->ij_gid, w.ln1_w, w.ln1_b, w.ln1_eps, w.gru_w, w.gru_b, w.ln2_w, w.ln2_b,
                           w.ln2_eps, t->net[0], nullptr, Eb, dyn, (const float *)w.heads_w, w.heads_b, t->coords, t->target,
                           t->weight, t->P, (float)t->feat_w, (float)t->feat_h, gate_at == 0 ? t->gate_flag : nullptr,
                           t->gate_seq, st));
#undef TRK_GATE32
    } else {
    // (the plan's (jj, ii)-major schedule: worth 1.3 % of the frame against graph order, round 5's A/B)
    TRK_DO(ramp_i_corr_fwd(t->gmap, lv, 2, t->coords, kk, jj, t->ij_order, t->corr, 896, (long)t->M * t->mem, t->mem, Eb,
                           t->mem * t->M, t->mem, 128, t->P, 3, RAMP_F16, t->feat_plain ? RAMP_NHWC : RAMP_NHWC32, dyn, st,
                           nullptr, nullptr, nullptr, nullptr, t->fmap1_slot));
    TRK_PROBE(1);
    // the update operator, ramp/net.py:69-90 (the fp16 fused chains of csrc/update_mlp.hip)
    TRK_DO(ramp_i_upd_corr_mlp(t->corr, 896, w.corr_w1, w.corr_b1, w.corr_w2, w.corr_b2, w.corr_w3, w.corr_b3, w.corr_ln_w,
                               w.corr_ln_b, w.corr_ln_eps, t->net[0], row, t->imap, kk, (long)t->M * t->mem, w.norm_w,
                               w.norm_b, w.norm_eps, t->net[1], Eb, dyn, st));
    // where the next frame's front end may start (RAMP_GATE_AT: 2 = before the first SoftAgg, the default -- with the fused
    // SoftAgg launches next to it the front end costs the operator ~25 us and gives bundle adjustment 12 back, +1.2 % SingleScale,
    // +2.1 % MultiScale against 0; 0 = before the gru chain (rounds 2-3); 1 = before the second SoftAgg; 3 = before c1 / c2).
    // The "go" is a word stored by the first workgroup of the launch behind that point (t->gate_flag: gru, SoftAgg), a
    // one-thread launch where that kernel cannot (the three-launch SoftAgg, c1), or the caller's event.
    const bool use_sagg = t->sagg_frag != nullptr;
    const bool flag_in_kernel = t->gate_flag && (gate_at == 0 || ((gate_at == 1 || gate_at == 2) && use_sagg));
#define TRK_GATE(pos)                                                                                     \
  do {                                                                                                    \
    if (gate_at == (pos) && !flag_in_kernel) {                                                            \
      if (t->gate_flag) hipLaunchKernelGGL(trk_signal_kernel, dim3(1), dim3(1), 0, st, t->gate_flag, t->gate_seq); \
      else if (gate_event && hipEventRecord((hipEvent_t)gate_event, st) != hipSuccess) return RAMP_ELAUNCH; \
    }                                                                                                     \
  } while (0)
    TRK_GATE(3);
    TRK_DO(ramp_i_upd_nbr(t->net[1], t->ix, w.c1_wa, w.c1_ba, w.c1_wb, w.c1_bb, t->net[2], nullptr, Eb, dyn, st));
    TRK_DO(ramp_i_upd_nbr(t->net[2], t->jx, w.c2_wa, w.c2_ba, w.c2_wb, w.c2_bb, t->net[1], nullptr, Eb, dyn, st));
    float *net = t->net[1];
    TRK_GATE(2);
    // SoftAgg x 2 (ramp/net.py:84-85).  With a fragment table: gather-by-group tiles, g and f on the same tile, online
    // softmax in registers, h on the merged fragments -- 2 launches each, no [E, 768] rows (csrc/update_mlp.hip);
    // no table: [f | g] rows + segment softmax + h, 3 launches each.
    const int add2 = 1;                           // (net + hkk[.] is never written back: the gru launch forms the sum itself)
    if (use_sagg) {
      TRK_DO(ramp_i_upd_softagg(net, nullptr, nullptr, t->kk_order, t->kk_gid, w.kk_wf, w.kk_bf, w.kk_wg, w.kk_bg, t->sagg_frag,
                                Eb, dyn, st, gate_at == 2 ? t->gate_flag : nullptr, t->gate_seq));
      TRK_DO(ramp_upd_softagg_finish(t->sagg_frag, t->kk_seg, t->kk_ngroups, w.kk_wh, w.kk_bh, t->hkk, t->kk_cap, stream));
      TRK_GATE(1);
      TRK_DO(ramp_i_upd_softagg(net, t->hkk, t->kk_gid, t->ij_order, t->ij_gid, w.ij_wf, w.ij_bf, w.ij_wg, w.ij_bg, t->sagg_frag,
                                Eb, dyn, st, gate_at == 1 ? t->gate_flag : nullptr, t->gate_seq));
      TRK_DO(ramp_upd_softagg_finish(t->sagg_frag, t->ij_seg, t->ij_ngroups, w.ij_wh, w.ij_bh, t->hij, t->ij_cap, stream));
    } else {
    TRK_DO(ramp_i_upd_fg(net, nullptr, nullptr, nullptr, w.kk_wf, w.kk_bf, w.kk_wg, w.kk_bg, t->fg, Eb, dyn, st));
    TRK_DO(ramp_upd_segment_softmax(t->fg, t->kk_order, t->kk_seg, t->kk_ngroups, t->ykk, t->kk_cap, RAMP_F16, stream));
    TRK_DO(ramp_upd_linear(t->ykk, w.kk_wh, w.kk_bh, t->hkk, t->kk_cap, t->kk_ngroups, stream));
    TRK_GATE(1);
    // the pair SoftAgg's [f|g] launch reads net + hkk[patch group] and, by default, does NOT write the sum back: the gru
    // launch forms (net + hkk[.]) + hij[.] itself, in the same order (61 MB less to write in the serial part of the step;
    // the extra table rows come from L2).
    TRK_DO(ramp_i_upd_fg(net, t->hkk, t->kk_gid, add2 ? nullptr : net, w.ij_wf, w.ij_bf, w.ij_wg, w.ij_bg, t->fg, Eb, dyn, st));
    TRK_DO(ramp_upd_segment_softmax(t->fg, t->ij_order, t->ij_seg, t->ij_ngroups, t->yij, t->ij_cap, RAMP_F16, stream));
    TRK_DO(ramp_upd_linear(t->yij, w.ij_wh, w.ij_bh, t->hij, t->ij_cap, t->ij_ngroups, stream));
    }
    TRK_GATE(0);
    // (the heads and target / weight are formed in the gru launch's epilogue: no relu(net) round trip, one launch less)
    TRK_DO(ramp_i_upd_gru(net, add2 ? t->hkk : nullptr, add2 ? t->kk_gid : nullptr, t->hij, t->ij_gid, w.ln1_w, w.ln1_b, w.ln1_eps, w.gru_w, w.gru_b, w.ln2_w, w.ln2_b,
                          w.ln2_eps, t->net[0], nullptr, Eb, dyn, w.heads_w, w.heads_b, t->coords, t->target, t->weight, t->P,
                          (float)t->feat_w, (float)t->feat_h, t->E_hint, gate_at == 0 ? t->gate_flag : nullptr, t->gate_seq, st));
    }   // (fp16 features)
    TRK_PROBE(2);
    TRK_PROBE(3);
    TRK_DO(ramp_i_ba_dyn(t->poses, t->patches, t->intrinsics, t->target, t->weight, t->lmbda, ii, jj, kk, Eb, t->P,
                         t->n_rows, t->n_rows * t->M, t->opt_window, 2, t->kk_order, t->kk_seg, t->kk_ngroups, t->kk_ukeys,
                         t->kk_cap, t->ij_order, t->ij_seg, t->ij_ngroups, t->ij_cap, t->ba_ws, t->ba_ws_bytes,
                         t->dyn + RAMP_DYN_STATUS, dyn, st));
    TRK_PROBE(4);
    // (with the motion test following, the point cloud rides in its launch)
    pc_with_mm = t->points && t->ixm && (flags & RAMP_TRACK_KEYFRAME) && !(flags & RAMP_TRACK_MM_GIVEN) && t->mm;
    if (t->points && t->ixm && !pc_with_mm)
      TRK_DO(ramp_i_point_cloud_dyn(t->poses, t->patches, t->intrinsics, t->ixm, t->points, t->m_cap, dyn, t->M, st));
    if (!(flags & RAMP_TRACK_KEYFRAME)) {
      // the new hidden state is indexed by the factors themselves from here on
      hipLaunchKernelGGL(trk_iota_kernel, dim3(ramp_cdiv(Eb, 256)), dim3(256), 0, st, t->graph[cur] + 3 * (size_t)Ec, t->dyn);
    }
  }
  if (flags & RAMP_TRACK_KEYFRAME) {
    if (!t->mm || !t->dlog) return RAMP_EINVAL;
    // Ramp_vo.keyframe(), ramp/Ramp_vo.py:237-274, and the next frame's append_factors (:394-395)
    if (pc_with_mm)
      TRK_DO(ramp_i_motionmag_point_cloud_dyn(t->poses, t->patches, t->intrinsics, ii, jj, kk, t->ij_order, t->ij_seg,
                                              t->ij_ukeys, t->ij_ngroups, 0.5f, t->mm, t->dyn, t->keyframe_index, t->ixm,
                                              t->points, t->m_cap, t->M, (t->median && t->keyframe_index >= 4) ? t->median : nullptr,
                                              st));
    else if (!(flags & RAMP_TRACK_MM_GIVEN))
      TRK_DO(ramp_i_motionmag_dyn(t->poses, t->patches, t->intrinsics, ii, jj, kk, t->ij_order, t->ij_seg, t->ij_ukeys,
                                  t->ij_ngroups, 0.5f, t->mm, dyn, t->keyframe_index, st));
    TrkEdit p;
    TRK_DO(trk_edit_fill(t, cur, counter, p));
    hipLaunchKernelGGL(trk_flag_kernel, dim3(p.nb), dim3(256), 0, st, p);
    hipLaunchKernelGGL(trk_decide_kernel, dim3(1), dim3(256), 0, st, p);
    int gx = p.nb + ramp_cdiv(new_cap + p.pad, 256);
    // column chunks of the row shift: with the level-0 planes behind the slot table the largest moved row is a few hundred KB
    // (256 workgroups x 4 KB), without it 4.9 MB -- the launch deals out gx x (1 + nbuf) workgroups whether a keyframe went or not
    const int gmin = t->fmap1_slot ? 256 : 1024;
    if (gx < gmin) gx = gmin;
    hipLaunchKernelGGL(trk_apply_kernel, dim3(gx, 1 + p.nbuf), dim3(256), 0, st, p);
    RAMP_CHECK_LAUNCH();
    const int Ep = Ep_next;
    TRK_DO(ramp_i_plan_dyn(t->graph[1 - cur], Ec, Ep, t->dyn, t->dyn + RAMP_DYN_STATUS, t->M, t->kkey_cap, t->pkey_cap,
                           t->kk_cap, t->ij_cap, t->kk_order, t->kk_gid, t->kk_seg, t->kk_ngroups, t->kk_ukeys, t->ij_order,
                           t->ij_gid, t->ij_seg, t->ij_ngroups, t->ij_ukeys, t->ix, t->jx, t->kj, t->plan_ws, t->plan_ws_bytes,
                           mirror, st));
  }
  // (without a plan in this call, or without a device mapping of the host buffer: an asynchronous copy)
  if (t->dyn_host && !(mirror && (flags & RAMP_TRACK_KEYFRAME)) &&
      hipMemcpyAsync(t->dyn_host, t->dyn, RAMP_DYN_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
    return RAMP_ELAUNCH;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

}  // extern "C"
