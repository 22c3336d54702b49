// Event-biased patch selection for gfx950 (reference ramp/utils.py:186-226 + 157-183):
//   score[X][Y] = mean over bins of avgpool4x4(|events|), laid out [w][h] (transposed)
//   keep local maxima of an 11x11 neighbourhood (x * (maxpool(x) == x))
//   top-k cells, sorted by value; x = flat_index / h (TRUE division, so x carries y/h), y = index % h
// upstream: abs, avg_pool2d, transpose, mean, max_pool2d, eq, mul, topk (radix sort + merges), div,
// remainder, stack = ~20 launches.  Here: score kernel, NMS kernel, one-workgroup radix select.
#include "ramp_device.h"
#include "median.h"
#include "ramp_internal.h"

// thread per 1/4-resolution cell, X fastest: a thread reads 16 contiguous bytes per row per bin
__global__ void __launch_bounds__(256)
    event_score_kernel(const float *__restrict__ ev, float *__restrict__ score, int bins, int H, int W, int h,
                       int w) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= h * w) return;
  const int Y = c / w, X = c - Y * w;
  float tot = 0.0f;
  for (int b = 0; b < bins; b++) {
    float acc = 0.0f;      // avg_pool2d: running sum over the window in (ky, kx) order, then / 16
#pragma unroll
    for (int ky = 0; ky < 4; ky++) {
      const float4 v = *reinterpret_cast<const float4 *>(ev + ((size_t)b * H + 4 * Y + ky) * W + 4 * X);
      acc += fabsf(v.x); acc += fabsf(v.y); acc += fabsf(v.z); acc += fabsf(v.w);
    }
    tot += acc / 16.0f;
  }
  score[(size_t)X * h + Y] = tot / (float)bins;
}

// x * (max over the (2r+1)^2 window == x), window clipped at the border (-inf padding).
// A workgroup stages its 16 x 16 cells plus the halo in LDS; max == x  <=>  no neighbour is larger,
// so a cell stops at the first larger neighbour.
#define NMS_T 16
#define NMS_RMAX 8
__global__ void __launch_bounds__(NMS_T * NMS_T)
    nms_kernel(const float *__restrict__ score, float *__restrict__ out, int w, int h, int r) {
  __shared__ float tile[(NMS_T + 2 * NMS_RMAX) * (NMS_T + 2 * NMS_RMAX)];
  const int tw = NMS_T + 2 * r;
  const int X0 = blockIdx.y * NMS_T, Y0 = blockIdx.x * NMS_T;          // Y (fast axis of score) on x
  for (int i = threadIdx.x; i < tw * tw; i += NMS_T * NMS_T) {
    const int tx = i / tw, ty = i - tx * tw;
    const int xx = X0 - r + tx, yy = Y0 - r + ty;
    tile[i] = (xx >= 0 && xx < w && yy >= 0 && yy < h) ? score[(size_t)xx * h + yy] : -INFINITY;
  }
  __syncthreads();
  const int lx = threadIdx.x / NMS_T, ly = threadIdx.x - lx * NMS_T;
  const int X = X0 + lx, Y = Y0 + ly;
  if (X >= w || Y >= h) return;
  const float v = tile[(lx + r) * tw + ly + r];
  bool keep = true;
  for (int dx = 0; dx <= 2 * r && keep; dx++)
    for (int dy = 0; dy <= 2 * r; dy++)
      if (tile[(lx + dx) * tw + ly + dy] > v) { keep = false; break; }
  out[(size_t)X * h + Y] = v * (keep ? 1.0f : 0.0f);
}

// One workgroup: the k largest cells, sorted by (value desc, index asc).
//   1. one pass over the map compacts the non-zero cells (after NMS: a few hundred local maxima) into
//      an LDS candidate list; zeros are only counted;
//   2. the exact k-th largest value is found by 4 x 8-bit MSB radix passes over the float bit
//      patterns of the candidates (values are >= 0, so the patterns order like the values); the bin
//      search is a wave-wide suffix sum over the 256 counters;
//   3. everything >= the threshold is gathered and bitonic-sorted; if fewer than k cells are
//      positive the remaining slots take the zero cells of lowest index.
// A map with more non-zero cells than the list holds (no NMS, dense events) streams the map from
// memory in every pass instead.
#define TOPK_THREADS 1024
#define TOPK_MAXK 512
#define TOPK_CAP 6144
__global__ void __launch_bounds__(TOPK_THREADS)
    topk_coords_kernel(const float *__restrict__ vals, int N, int k, int hh, float *__restrict__ coords,
                       int64_t *__restrict__ idx_out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_remaining, s_count, s_ncand;
  __shared__ unsigned cand_key[TOPK_CAP], cand_idx[TOPK_CAP];
  __shared__ unsigned long long sel[2 * TOPK_MAXK];
  __shared__ unsigned s_scan[TOPK_THREADS];
  const int tid = threadIdx.x;
  const unsigned *keys = reinterpret_cast<const unsigned *>(vals);
  if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)k; s_count = 0; s_ncand = 0; }
  for (int i = tid; i < 2 * TOPK_MAXK; i += TOPK_THREADS) sel[i] = 0ull;
  __syncthreads();
  for (int i0 = tid; i0 < N; i0 += 4 * TOPK_THREADS) {
    unsigned kv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) kv[u] = (i0 + u * TOPK_THREADS < N) ? keys[i0 + u * TOPK_THREADS] : 0u;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (kv[u] != 0) {
        const unsigned pos = atomicAdd(&s_ncand, 1u);
        if (pos < TOPK_CAP) { cand_key[pos] = kv[u]; cand_idx[pos] = (unsigned)(i0 + u * TOPK_THREADS); }
      }
    }
  }
  __syncthreads();
  const unsigned nnz = s_ncand;
  const bool in_lds = nnz <= TOPK_CAP;
  const int M = in_lds ? (int)nnz : N;           // items the passes below walk over
  const unsigned zeros = (unsigned)N - nnz;
#define TOPK_ITEM(i, key, idx)                                   \
  const unsigned key = in_lds ? cand_key[i] : keys[i];           \
  const unsigned idx = in_lds ? cand_idx[i] : (unsigned)(i);
  unsigned mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int i = tid; i < M; i += TOPK_THREADS) {
      TOPK_ITEM(i, key, idx)
      (void)idx;
      if (key != 0 && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // which bin holds the rem-th largest: suffix sums over the 256 bins, 4 bins per lane of wave 0
      unsigned c[4];
#pragma unroll
      for (int b = 0; b < 4; b++) c[b] = hist[4 * tid + b];
      if (prefix == 0 && tid == 0) c[0] += zeros;
      const unsigned tot = c[0] + c[1] + c[2] + c[3];
      unsigned suf = tot;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_down(suf, o, 64);
        if (tid + o < 64) suf += v;
      }
      const unsigned rem = s_remaining, above = suf - tot;
      if (above < rem && suf >= rem) {
        unsigned rr = rem - above;
        int bin = 4 * tid;
#pragma unroll
        for (int b = 3; b >= 0; b--) {
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
  const unsigned T = s_prefix;
  // gather: T > 0: everything >= T (ties are ordered by the sort); T == 0: every positive cell
  for (int i = tid; i < M; i += TOPK_THREADS) {
    TOPK_ITEM(i, key, idx)
    if (key != 0 && key >= T) {
      const unsigned pos = atomicAdd(&s_count, 1u);
      if (pos < 2u * TOPK_MAXK) sel[pos] = ((unsigned long long)key << 32) | (0xffffffffu - idx);
    }
  }
  __syncthreads();
  if (T == 0) {
    // fewer than k positive cells: the zero cells of lowest index fill up (index-ordered scan of the map)
    const unsigned ngt = s_count, need = (unsigned)k - min((unsigned)k, ngt);
    const int chunk = (N + TOPK_THREADS - 1) / TOPK_THREADS;
    const int i0 = tid * chunk, i1 = min(N, i0 + chunk);
    unsigned cnt = 0;
    for (int i = i0; i < i1; i++) cnt += (keys[i] == 0);
    s_scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < TOPK_THREADS; off <<= 1) {
      const unsigned v = tid >= off ? s_scan[tid - off] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    unsigned rank = s_scan[tid] - cnt;
    for (int i = i0; i < i1 && rank < need; i++) {
      if (keys[i] == 0) {
        const unsigned pos = ngt + rank;
        if (pos < 2u * TOPK_MAXK) sel[pos] = (unsigned long long)(0xffffffffu - (unsigned)i);
        rank++;
      }
    }
    __syncthreads();
  }
  // bitonic sort, descending, of the gathered composite keys (unused slots are 0 = smallest)
  const unsigned gathered = T == 0 ? (unsigned)k : min(s_count, 2u * TOPK_MAXK);
  int KP = 64;
  while (KP < (int)gathered) KP <<= 1;
  for (int size = 2; size <= KP; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < KP / 2; t += TOPK_THREADS) {
        const int lo = (t / stride) * stride * 2 + (t % stride), hi = lo + stride;
        const bool desc = ((lo / size) & 1) == 0;
        const unsigned long long a = sel[lo], b = sel[hi];
        if ((a < b) == desc) { sel[lo] = b; sel[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int t = tid; t < k; t += TOPK_THREADS) {
    const unsigned idx = 0xffffffffu - (unsigned)(sel[t] & 0xffffffffull);
    // `indices / h` upstream runs on the GPU, where ATen turns a true division by a host scalar into
    // a multiplication by its float reciprocal (BinaryDivTrueKernel.cu); same arithmetic here
    coords[2 * t + 0] = (float)idx * (1.0f / (float)hh);
    coords[2 * t + 1] = (float)(idx % (unsigned)hh);
    if (idx_out) idx_out[t] = (int64_t)idx;
  }
#undef TOPK_ITEM
}

// Depth initialisation of a new frame's patches (reference ramp/Ramp_vo.py:370-371):
//   patches[:, :, 2] = torch.median(self.patches_[n-3:n, :, 2])
// i.e. the lower median of the F*M*PP inverse depths of the last F frames, broadcast into the depth plane
// of the M new patches.  One workgroup: keys in registers, 4 x 8-bit radix passes (k-th smallest), fill.
#define MED_THREADS 1024
#define MED_PER 8               // up to 8192 values (3 frames x 256 patches x 9 pixels = 6912)
__device__ __forceinline__ float depth_median_block(const float *__restrict__ src, int F, int M, int PP) {
  return depth_median_block_t<MED_THREADS, MED_PER>(src, F, M, PP);
}

__global__ void __launch_bounds__(MED_THREADS)
    depth_median_fill_kernel(const float *__restrict__ src, int F, int M, int PP, float *__restrict__ dst) {
  const float med = depth_median_block(src, F, M, PP);
  for (int i = threadIdx.x; i < M * PP; i += MED_THREADS) {
    const int m = i / PP, p = i - m * PP;
    dst[((size_t)m * 3 + 2) * PP + p] = med;
  }
}

// the median alone, into device memory: the tracker computes it right after bundle adjustment, off the critical path
// (the three newest frames are the same whether or not the keyframe test then drops an older one)
__global__ void __launch_bounds__(MED_THREADS)
    depth_median_kernel(const float *__restrict__ src, int F, int M, int PP, float *__restrict__ out) {
  const float med = depth_median_block(src, F, M, PP);
  if (threadIdx.x == 0) *out = med;
}

// Everything a steady-state Ramp_vo.__call__ writes before its reprojection (ramp/Ramp_vo.py:345-381) as ONE launch
// -- these were three dependent tiny launches on the frame's critical path.  Workgroup (0, 0): time stamp, index
// map, intrinsics row, motion-model pose (ramp_frame_begin), the depth median (ramp_depth_median_fill) and the new
// row of the patch buffer; workgroups (*, 1 + b): copy of buffer b (ramp_multi_copy).
#define FC_MAXBUF 6
struct FrameCommit {
  float *poses; int n, motion; float damping; int64_t *tstamps; int64_t counter; int64_t *index_map; int64_t index_val;
  float *intrinsics; int copy_k;
  const float *median_src; int F, M, PP; float *patches_new; float *patches_row;
  int n_copy; const char *src[FC_MAXBUF]; char *dst[FC_MAXBUF]; long bytes[FC_MAXBUF];
  const float *median_val;   // optional: the median of median_src, computed ahead (ramp_depth_median)
  // device-side row (csrc/track.hip): n = dyn[RAMP_DYN_NROW]; dst[b] is then the BASE of buffer b and the row is
  // n % mod[b] (ring buffers) or n (mod[b] == 0); median_src / patches_row are the base of the patch buffer; k_new
  // (optional) replaces the copy of the previous intrinsics row
  const int32_t *dyn; int mod[FC_MAXBUF]; const float *k_new;
  int32_t *status; int E_bound;      // optional: status bit 32 if dyn[RAMP_DYN_E] exceeds the step's launch bound
  int32_t *status_rows; int n_rows;  // with dyn: the frame buffers hold n_rows rows -- a row past them is flagged (bit 64), nothing is stored
  const int32_t *slot_tab; int slot_buf;   // optional: ring row r of buffer slot_buf lives in physical slot slot_tab[r] (ramp_track.fmap1_slot)
  uint32_t *signal; uint32_t signal_val;   // optional signal word: "everything in front of this launch on its stream is done"
};
__global__ void __launch_bounds__(MED_THREADS) frame_commit_kernel(const FrameCommit a) {
  const int t = threadIdx.x;
  int n = a.n;
  int64_t index_val = a.index_val;
  const float *median_src = a.median_src;
  float *patches_row = a.patches_row;
  if (a.signal && blockIdx.x == 0 && blockIdx.y == 0 && t == 0)
    __hip_atomic_store(a.signal, a.signal_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (a.dyn) {
    n = a.dyn[RAMP_DYN_NROW];
    if (a.n_rows > 0 && (n < 0 || n > a.n_rows - 2)) {      // (index_map is written at n + 1)
      if (a.status_rows && blockIdx.x == 0 && blockIdx.y == 0 && t == 0) atomicOr(a.status_rows, 64);
      return;
    }
    const size_t row = (size_t)a.M * 3 * a.PP;
    index_val = (int64_t)(n + 1) * a.M;
    median_src += (size_t)(n - a.F) * row;
    patches_row += (size_t)n * row;
  }
  if (blockIdx.y > 0) {
    const int b = blockIdx.y - 1;
    if (b >= a.n_copy) return;
    char *d = a.dst[b];
    if (a.dyn) {
      int row = a.mod[b] ? n % a.mod[b] : n;
      if (a.slot_tab && b == a.slot_buf) row = a.slot_tab[row];
      d += (size_t)row * a.bytes[b];
    }
    const long g0 = (long)blockIdx.x * blockDim.x + t, gs = (long)gridDim.x * blockDim.x;
    // 16-byte pieces where the row is made of them (source and buffer base are 16-byte aligned: checked on the host); a row
    // that is not -- the colours of a frame, 3 M bytes, for a PATCHES_PER_FRAME that is no multiple of 16 (precise.yaml: 300)
    // -- goes in 4-byte pieces or byte by byte
    if (!(a.bytes[b] & 15)) {
      const uint4 *s = reinterpret_cast<const uint4 *>(a.src[b]);
      uint4 *o = reinterpret_cast<uint4 *>(d);
      for (long i = g0; i < a.bytes[b] / 16; i += gs) o[i] = s[i];
    } else if (!(a.bytes[b] & 3)) {
      const uint32_t *s = reinterpret_cast<const uint32_t *>(a.src[b]);
      uint32_t *o = reinterpret_cast<uint32_t *>(d);
      for (long i = g0; i < a.bytes[b] / 4; i += gs) o[i] = s[i];
    } else {
      for (long i = g0; i < a.bytes[b]; i += gs) d[i] = a.src[b][i];
    }
    return;
  }
  if (blockIdx.x != 0) return;
  if (t == 0) {
    // (the launch bound of this step's per-factor kernels was below the live factor count: the frame would be computed on
    // a truncated graph -- flagged, csrc/track.hip)
    if (a.status && a.dyn && a.dyn[RAMP_DYN_E] > a.E_bound) atomicOr(a.status, 32);
    if (a.tstamps) a.tstamps[n] = a.counter;
    if (a.index_map) a.index_map[n + 1] = index_val;
  }
  if (a.k_new && t < 4) a.intrinsics[4 * n + t] = a.k_new[t];
  if (a.copy_k && t < 4) a.intrinsics[4 * n + t] = a.intrinsics[4 * (n - 1) + t];
  if (a.motion == 2 && t < 7) a.poses[7 * n + t] = a.poses[7 * (n - 1) + t];
  if (a.motion == 1 && t == 0) {
    float P1[7], P2[7], P2i[7], D[7], xi[6], E[7], Pn[7];
    for (int c = 0; c < 7; c++) { P1[c] = a.poses[7 * (n - 1) + c]; P2[c] = a.poses[7 * (n - 2) + c]; }
    lt_inv(P2, P2i);
    lt_mul(P1, P2i, D);
    lt_log(D, xi);
    for (int c = 0; c < 6; c++) xi[c] = a.damping * xi[c];
    lt_exp(xi, E);
    lt_mul(E, P1, Pn);
    for (int c = 0; c < 7; c++) a.poses[7 * n + c] = Pn[c];
  }
  const bool fill = a.F > 0;
  const bool ahead = a.median_val && (!a.dyn || a.dyn[RAMP_DYN_MEDOK]);      // (workgroup uniform)
  const float med = !fill ? 0.f : (ahead ? *a.median_val : depth_median_block(median_src, a.F, a.M, a.PP));
  for (int i = t; i < a.M * 3 * a.PP; i += MED_THREADS) {
    const int ch = (i / a.PP) % 3;
    float v = a.patches_new[i];
    if (fill && ch == 2) { v = med; a.patches_new[i] = med; }
    patches_row[i] = v;
  }
}


// ---- the same launch with the row taken from device memory (csrc/track.hip): bases[b] + row * bytes[b], row =
// dyn[RAMP_DYN_NROW] % mod[b] (mod[b] > 0: ring buffer) -- the host does not know the row before the previous frame's
// keyframe test has run
int ramp_i_frame_commit_dyn(float *poses, int motion, float damping, int64_t *tstamps, int64_t counter,
                            int64_t *index_map, float *intrinsics, const float *k_new, float *patches_state,
                            int median_frames, int M, int P, float *patches_new, int n_copy, const void *const *src,
                            void *const *base, const long *bytes, const int *mod, const int32_t *dyn,
                            const float *median_ahead, int32_t *status, int E_bound, int n_rows, hipStream_t st,
                            const int32_t *slot_tab, int slot_buf, uint32_t *signal, uint32_t signal_val) {
  if (!poses || !patches_state || !patches_new || !dyn || M <= 0 || P <= 0 || n_copy < 0 || n_copy > FC_MAXBUF)
    return RAMP_EINVAL;
  if ((long)median_frames * M * P * P > MED_THREADS * MED_PER) return RAMP_EUNSUPPORTED;
  FrameCommit a;
  a.poses = poses; a.n = 0; a.motion = motion; a.damping = damping; a.tstamps = tstamps; a.counter = counter;
  a.index_map = index_map; a.index_val = 0; a.intrinsics = intrinsics; a.copy_k = k_new ? 0 : 1;
  a.median_src = patches_state; a.F = median_frames; a.M = M; a.PP = P * P;
  a.patches_new = patches_new; a.patches_row = patches_state;
  a.median_val = median_ahead;      // valid while dyn[RAMP_DYN_MEDOK] (csrc/lie.hip: computed beside the previous motion test)
  a.dyn = dyn; a.k_new = k_new;
  a.status = E_bound > 0 ? status : nullptr; a.E_bound = E_bound;
  a.status_rows = status; a.n_rows = n_rows;
  a.slot_tab = slot_tab; a.slot_buf = slot_buf;
  a.signal = signal; a.signal_val = signal_val;
  if (slot_tab && (slot_buf < 0 || slot_buf >= n_copy || mod[slot_buf] <= 0)) return RAMP_EINVAL;
  a.n_copy = n_copy;
  long mx = 0;
  for (int i = 0; i < FC_MAXBUF; i++) a.mod[i] = 0;
  for (int i = 0; i < n_copy; i++) {
    if (!src[i] || !base[i] || bytes[i] <= 0 || (((uintptr_t)src[i] | (uintptr_t)base[i]) & 15)) return RAMP_EINVAL;
    a.src[i] = (const char *)src[i]; a.dst[i] = (char *)base[i]; a.bytes[i] = bytes[i]; a.mod[i] = mod[i];
    if (bytes[i] > mx) mx = bytes[i];
  }
  int bx = (int)((mx / 16 + MED_THREADS - 1) / MED_THREADS);
  bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
  hipLaunchKernelGGL(frame_commit_kernel, dim3(bx, 1 + n_copy), dim3(MED_THREADS), 0, st, a);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

extern "C" {

int ramp_depth_median_fill(const float *patches_src, int F, int M, int P, float *patches_dst, void *stream) {
  if (!patches_src || !patches_dst || F <= 0 || M <= 0 || P <= 0) return RAMP_EINVAL;
  if ((long)F * M * P * P > MED_THREADS * MED_PER) return RAMP_EUNSUPPORTED;
  hipLaunchKernelGGL(depth_median_fill_kernel, dim3(1), dim3(MED_THREADS), 0, (hipStream_t)stream, patches_src, F,
                     M, P * P, patches_dst);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

size_t ramp_event_topk_workspace_bytes(int H, int W) {
  return (size_t)2 * (H / 4) * (W / 4) * sizeof(float);
}

int ramp_event_topk(const float *events, int bins, int H, int W, int k, int nms_kernel_size, float *coords,
                    int64_t *indices, void *ws, size_t ws_bytes, void *stream) {
  if (!events || !coords || !ws || bins <= 0 || H < 4 || W < 4 || k <= 0) return RAMP_EINVAL;
  const int h = H / 4, w = W / 4, N = h * w;
  if (k > TOPK_MAXK || k > N || (W % 4) || nms_kernel_size < 0 || (nms_kernel_size && !(nms_kernel_size & 1)) ||
      nms_kernel_size > 2 * NMS_RMAX + 1)
    return RAMP_EUNSUPPORTED;
  if (ws_bytes < ramp_event_topk_workspace_bytes(H, W)) return RAMP_EWORKSPACE;
  float *score = (float *)ws, *kept = score + N;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(event_score_kernel, dim3(ramp_cdiv(N, 256)), dim3(256), 0, st, events, score, bins, H, W, h,
                     w);
  const float *src = score;
  if (nms_kernel_size > 1) {
    hipLaunchKernelGGL(nms_kernel, dim3(ramp_cdiv(h, NMS_T), ramp_cdiv(w, NMS_T)), dim3(NMS_T * NMS_T), 0, st,
                       score, kept, w, h, (nms_kernel_size - 1) / 2);
    src = kept;
  }
  hipLaunchKernelGGL(topk_coords_kernel, dim3(1), dim3(TOPK_THREADS), 0, st, src, N, k, h, coords, indices);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_depth_median(const float *patches_src, int F, int M, int P, float *out, void *stream) {
  if (!patches_src || !out || F <= 0 || M <= 0 || P <= 0) return RAMP_EINVAL;
  if ((long)F * M * P * P > MED_THREADS * MED_PER) return RAMP_EUNSUPPORTED;
  hipLaunchKernelGGL(depth_median_kernel, dim3(1), dim3(MED_THREADS), 0, (hipStream_t)stream, patches_src, F, M, P * P, out);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_frame_commit(float *poses, int n, int motion, float damping, int64_t *tstamps, int64_t counter,
                      int64_t *index_map, int64_t index_val, float *intrinsics, int copy_k, float *patches_state,
                      int median_frames, int M, int P, float *patches_new, int n_copy, const void *const *src_host,
                      void *const *dst_host, const long *bytes_host, const float *median_dev, void *stream) {
  if (!poses || !patches_state || !patches_new || n < 0 || M <= 0 || P <= 0 || n_copy < 0 || n_copy > FC_MAXBUF)
    return RAMP_EINVAL;
  if ((motion == 1 && n < 2) || ((motion == 2 || copy_k) && n < 1) || (copy_k && !intrinsics)) return RAMP_EINVAL;
  if (median_frames < 0 || n - median_frames < 0) return RAMP_EINVAL;
  if ((long)median_frames * M * P * P > MED_THREADS * MED_PER) return RAMP_EUNSUPPORTED;
  FrameCommit a;
  a.poses = poses; a.n = n; a.motion = motion; a.damping = damping; a.tstamps = tstamps; a.counter = counter;
  a.index_map = index_map; a.index_val = index_val; a.intrinsics = intrinsics; a.copy_k = copy_k;
  const size_t row = (size_t)M * 3 * P * P;
  a.median_src = patches_state + (size_t)(n - median_frames) * row; a.F = median_frames; a.M = M; a.PP = P * P;
  a.patches_new = patches_new; a.patches_row = patches_state + (size_t)n * row;
  a.median_val = median_dev;
  a.dyn = nullptr; a.k_new = nullptr; a.status = nullptr; a.E_bound = 0; a.status_rows = nullptr; a.n_rows = 0;
  a.slot_tab = nullptr; a.slot_buf = 0; a.signal = nullptr; a.signal_val = 0;
  for (int i = 0; i < FC_MAXBUF; i++) a.mod[i] = 0;
  a.n_copy = n_copy;
  long mx = 0;
  for (int i = 0; i < n_copy; i++) {
    // (a destination row that is no multiple of 16 bytes need not be 16-byte aligned: the kernel then copies 4- or 1-byte pieces)
    if (!src_host[i] || !dst_host[i] || bytes_host[i] <= 0 || ((uintptr_t)src_host[i] & 15) ||
        ((uintptr_t)dst_host[i] & ((bytes_host[i] & 15) ? ((bytes_host[i] & 3) ? 0 : 3) : 15)))
      return RAMP_EINVAL;
    a.src[i] = (const char *)src_host[i]; a.dst[i] = (char *)dst_host[i]; a.bytes[i] = bytes_host[i];
    if (bytes_host[i] > mx) mx = bytes_host[i];
  }
  int bx = (int)((mx / 16 + MED_THREADS - 1) / MED_THREADS);
  bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
  hipLaunchKernelGGL(frame_commit_kernel, dim3(bx, 1 + n_copy), dim3(MED_THREADS), 0, (hipStream_t)stream, a);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

}  // extern "C"
