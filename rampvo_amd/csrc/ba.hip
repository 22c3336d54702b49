// fastba.BA on gfx950: Gauss-Newton bundle adjustment of the sliding window.
//
// The reference accumulates the normal equations with ~340 float atomics per
// edge into a 60x60 matrix and then runs ~25 ATen launches per iteration.  Here
// every reduction is an ORDERED SEGMENT SUM (deterministic, no float atomics):
//
//   prep   group the edges by patch (kk) and by pose pair (ii,jj)  [graph.hip]
//   K1     edge kernel: residual, validity, Jacobians -> 128 B record / edge
//   K2     patch kernel: one workgroup per patch walks its edge segment and
//          emits the dense E row [6N], C, u, Q = 1/(C+lambda)
//   K3     pair kernel: one workgroup per (i,j) segment emits the 6x6 blocks
//          w*Ji*Ji', w*Jj*Jj', -w*Ji*Jj', -w*Jj*Ji' and the gradient parts
//   K4     split-K SYRK: partial  E' diag(Q) E  and  E' diag(Q) u
//   K5     assemble  S = B - sum(partials),  y = v - ..., damping
//   K6     single-workgroup LDS Cholesky + triangular solves -> dX
//   K7     dZ = Q (u - E dX), depth retraction, SE3 pose retraction
//
// Math restates ramp/fastba/ba_cuda.cu:232-376 (kernel), 433-582 (host loop),
// 178-229 (retractions).  Everything is fp32 like the reference (mtype=float).
#include "ramp_device.h"
#include <stdlib.h>
#include "ramp_internal.h"

#define BA_REC 32     // floats per edge record
#define BA_PAIR 160   // floats per pair record (156 used)
#define BA_TS 64      // SYRK tile
#define BA_KB 8       // SYRK k batch

static inline size_t al(size_t x) { return (x + 255) / 256 * 256; }

// ------------------------------------------------------------------ K1
// The per-edge record: residual, validity, Jacobians (32 floats).  ba_edge_kernel writes it to memory (the path without
// free poses); the patch / pair kernels of a regular iteration recompute it in registers from the edge's 116 bytes of
// inputs while they stage their segment -- the record never goes through HBM (it was written once and read twice per
// iteration) and the launch is gone.  Same expressions either way: identical results.
struct BaEdgeIn {
  const float *poses, *patches, *intr, *target, *weight;
  const int64_t *ii, *jj, *kk;
  int PP, c11, t0, N;
};
__device__ __forceinline__ void ba_edge_compute(const BaEdgeIn &in, int n, float (&r)[BA_REC]) {
  const float fx = in.intr[0], fy = in.intr[1], cx = in.intr[2], cy = in.intr[3];
  int ix = (int)in.ii[n], jx = (int)in.jj[n];
  const long kx = in.kk[n];
  float pi[7], pj[7];
#pragma unroll
  for (int c = 0; c < 7; c++) { pi[c] = in.poses[7 * (size_t)ix + c]; pj[c] = in.poses[7 * (size_t)jx + c]; }
  float Xi[4], Xj[4];
  Xi[0] = (in.patches[((size_t)kx * 3 + 0) * in.PP + in.c11] - cx) / fx;
  Xi[1] = (in.patches[((size_t)kx * 3 + 1) * in.PP + in.c11] - cy) / fy;
  Xi[2] = 1.0f;
  Xi[3] = in.patches[((size_t)kx * 3 + 2) * in.PP + in.c11];
  float tij[3], qij[4];
  fb_relSE3(pi, pi + 3, pj, pj + 3, tij, qij);
  fb_actSE3(tij, qij, Xi, Xj);
  const float X = Xj[0], Y = Xj[1], Z = Xj[2], W = Xj[3];
  const float d = (Z >= 0.2f) ? 1.0f / Z : 0.0f;
  const float d2 = d * d;
  const float x1 = fx * (X / Z) + cx;
  const float y1 = fy * (Y / Z) + cy;
  const float tx_ = in.target[2 * (size_t)n + 0], ty_ = in.target[2 * (size_t)n + 1];
  const float rx = tx_ - x1, ry = ty_ - y1;
  const bool in_bounds = (sqrtf(rx * rx + ry * ry) < 128) && (Z > 0.2f) && (x1 > -64) &&
                         (y1 > -64) && (x1 < 2 * cx + 64) && (y1 < 2 * cy + 64);
  const float mask = in_bounds ? 1.0f : 0.0f;
  ix -= in.t0;
  jx -= in.t0;
  if (ix >= in.N) ix = -1;  // poses >= t1 are not free (out of B in the reference)
  if (jx >= in.N) jx = -1;
  if (ix < 0) ix = -1;
  if (jx < 0) jx = -1;
  float Jj0[6] = {fx * W * d, 0, fx * -X * W * d2, fx * -X * Y * d2, fx * (1 + X * X * d2),
                  fx * -Y * d};
  float Jj1[6] = {0, fy * W * d, fy * -Y * W * d2, fy * (-1 - Y * Y * d2), fy * (X * Y * d2),
                  fy * X * d};
  float Ji0[6], Ji1[6];
  fb_adjSE3(tij, qij, Jj0, Ji0);
  fb_adjSE3(tij, qij, Jj1, Ji1);
  const float Jz0 = fx * (tij[0] * d - tij[2] * (X * d2));
  const float Jz1 = fy * (tij[1] * d - tij[2] * (Y * d2));
#pragma unroll
  for (int c = 0; c < 6; c++) { r[c] = Ji0[c]; r[6 + c] = Ji1[c]; r[12 + c] = Jj0[c]; r[18 + c] = Jj1[c]; }
  r[24] = mask * in.weight[2 * (size_t)n + 0];
  r[25] = mask * in.weight[2 * (size_t)n + 1];
  r[26] = rx; r[27] = ry; r[28] = Jz0; r[29] = Jz1;
  r[30] = __int_as_float(ix); r[31] = __int_as_float(jx);
}
// window from the device-side sizes (csrc/track.hip): [max(n - opt_window, 1), n)
__device__ __forceinline__ void ba_dyn_window(const int32_t *dyn, int opt_window, int &t0, int &N) {
  if (!dyn) return;
  const int t1 = dyn[RAMP_DYN_N];
  t0 = max(t1 - opt_window, 1);
  N = min(N, t1 - t0);
}
__global__ void __launch_bounds__(256)
    ba_edge_kernel(BaEdgeIn in, float *__restrict__ rec, int E, const int32_t *__restrict__ dyn, int opt_window) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (dyn) E = dyn[RAMP_DYN_E];              // device-side sizes: E is the launch bound, the window ends at n
  ba_dyn_window(dyn, opt_window, in.t0, in.N);
  if (n >= E) return;
  float r[BA_REC];
  ba_edge_compute(in, n, r);
  float4 *r4 = reinterpret_cast<float4 *>(rec + (size_t)n * BA_REC);
#pragma unroll
  for (int c = 0; c < 8; c++) r4[c] = make_float4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]);
}
// record field offsets
#define R_JI0 0
#define R_JI1 6
#define R_JJ0 12
#define R_JJ1 18
#define R_W0 24
#define R_W1 25
#define R_R0 26
#define R_R1 27
#define R_JZ0 28
#define R_JZ1 29
#define R_I 30
#define R_J 31

// ------------------------------------------------------------------ K2
// One workgroup per patch (group of edges with the same kk).  The group's records are staged in LDS
// with all loads in flight at once (walking them from memory is one dependent round trip per edge);
// the sums run over the staged records in segment order.
#define BA_PCHUNK 32
#ifndef BA_PBATCH
#define BA_PBATCH 48    // (128 / 224 measured on MI355X: 126 / 139 us per BA call against 124 -- the pair role is not the long one)
#endif
__device__ __forceinline__ void
    ba_patch_body(int g, const float *__restrict__ rec, const int32_t *__restrict__ order,
                  const int32_t *__restrict__ seg, const int32_t *__restrict__ ngroups,
                  const float *__restrict__ lmbda, float *__restrict__ Erow,
                  float *__restrict__ Cv, float *__restrict__ uv, float *__restrict__ Qv,
                  int n6, const BaEdgeIn &ein, const bool fused, float *__restrict__ s_buf) {
  float (*s_rec)[BA_REC + 1] = reinterpret_cast<float (*)[BA_REC + 1]>(s_buf);      // [BA_PCHUNK][BA_REC + 1]
  if (g >= *ngroups) return;
  // The merged launch's workgroups are sized for the pair role (156 sums); a patch needs n6 + 2 threads.  Waves without a
  // column leave at once (the barrier counts live waves only): a CU holds 32 waves, and 2,100 patch workgroups of four
  // live waves each were a second round of workgroups behind 2,048 places -- with one live wave they are all resident.
#ifndef BA_PATCH_KEEP_WAVES
  // (whole waves leave in front of the barriers below: s_barrier on gfx9 / CDNA counts the waves that are still alive, so the
  // remaining ones synchronise among themselves.  This library is built for gfx950 only; a target whose barrier counts
  // launched waves would hang here -- hence the check, and -DBA_PATCH_KEEP_WAVES as the portable form)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "ba_patch_body retires whole waves before __syncthreads(): valid on gfx9 / CDNA barriers only (build with -DBA_PATCH_KEEP_WAVES)"
#endif
  const int nthr = min((int)blockDim.x, ((n6 + 2 + 63) / 64) * 64);
  if ((int)threadIdx.x >= nthr) return;
#else
  const int nthr = blockDim.x;
#endif
  const int tid = threadIdx.x;
  const int s0 = seg[g], s1 = seg[g + 1];
  const int a = tid / 6, c = tid - a * 6;
  float acc = 0.0f;
  for (int b0 = s0; b0 < s1; b0 += BA_PCHUNK) {
    const int nb = min(BA_PCHUNK, s1 - b0);
    __syncthreads();
    if (fused) {                                // one thread per edge of the chunk recomputes its record
      if (tid < nb) {
        float r[BA_REC];
        ba_edge_compute(ein, order[b0 + tid], r);
#pragma unroll
        for (int k = 0; k < BA_REC; k++) s_rec[tid][k] = r[k];
      }
    } else {
      for (int e = tid; e < nb * BA_REC; e += nthr) {
        const int l = e / BA_REC, k = e - l * BA_REC;
        s_rec[l][k] = rec[(size_t)order[b0 + l] * BA_REC + k];
      }
    }
    __syncthreads();
    if (tid < n6) {
      for (int l = 0; l < nb; l++) {
        const float *r = s_rec[l];
        const int i = __float_as_int(r[R_I]), j = __float_as_int(r[R_J]);
        if (i == a) acc += (-r[R_W0] * r[R_JZ0]) * r[R_JI0 + c];
        if (j == a) acc += (r[R_W0] * r[R_JZ0]) * r[R_JJ0 + c];
        if (i == a) acc += (-r[R_W1] * r[R_JZ1]) * r[R_JI1 + c];
        if (j == a) acc += (r[R_W1] * r[R_JZ1]) * r[R_JJ1 + c];
      }
    } else if (tid == n6) {
      for (int l = 0; l < nb; l++) {
        const float *r = s_rec[l];
        acc += (r[R_W0] * r[R_JZ0]) * r[R_JZ0];
        acc += (r[R_W1] * r[R_JZ1]) * r[R_JZ1];
      }
    } else if (tid == n6 + 1) {
      for (int l = 0; l < nb; l++) {
        const float *r = s_rec[l];
        acc += (r[R_W0] * r[R_R0]) * r[R_JZ0];
        acc += (r[R_W1] * r[R_R1]) * r[R_JZ1];
      }
    }
  }
  if (tid < n6) {
    Erow[(size_t)g * n6 + tid] = acc;
  } else if (tid == n6) {
    Cv[g] = acc;
    Qv[g] = 1.0f / (acc + lmbda[0]);
  } else if (tid == n6 + 1) {
    uv[g] = acc;
  }
}

__global__ void __launch_bounds__(256)
    ba_patch_kernel(const float *__restrict__ rec, const int32_t *__restrict__ order,
                    const int32_t *__restrict__ seg, const int32_t *__restrict__ ngroups,
                    const float *__restrict__ lmbda, float *__restrict__ Erow,
                    float *__restrict__ Cv, float *__restrict__ uv, float *__restrict__ Qv,
                    int n6) {
  __shared__ __attribute__((aligned(16))) float s_buf[BA_PCHUNK * (BA_REC + 1)];
  ba_patch_body(blockIdx.x, rec, order, seg, ngroups, lmbda, Erow, Cv, uv, Qv, n6, BaEdgeIn(), false, s_buf);
}

// ------------------------------------------------------------------ K3
// pair record: [0,36) w Ji Ji', [36,72) w Jj Jj', [72,108) -w Ji Jj',
// [108,144) -w Jj Ji', [144,150) -w r Ji, [150,156) w r Jj.  Threads >= 192 only take part in the barriers.
__device__ __forceinline__ void
    ba_pair_body(int g, const float *__restrict__ rec, const int32_t *__restrict__ order,
                 const int32_t *__restrict__ seg, const int32_t *__restrict__ ngroups,
                 float *__restrict__ pairs, int32_t *__restrict__ pair_ij, const BaEdgeIn &ein, const bool fused,
                 float *__restrict__ s_rec) {
  // BA_PBATCH records per batch (any batch size sums the records in the same order: identical values)
  if (g >= *ngroups) return;
  const int tid = threadIdx.x;
  const int s0 = seg[g], s1 = seg[g + 1];
  int blk = 0, x = 0, y = 0, oa = 0, ob = 0;
  float sgn = 1.0f;
  if (tid < 144) {
    blk = tid / 36;
    const int q = tid - blk * 36;
    x = q / 6; y = q - x * 6;
    oa = (blk == 0 || blk == 2) ? R_JI0 : R_JJ0;  // left factor
    ob = (blk == 0 || blk == 3) ? R_JI0 : R_JJ0;  // right factor
    sgn = (blk >= 2) ? -1.0f : 1.0f;
  } else if (tid < 156) {
    const int q = tid - 144;
    const bool isj = q >= 6;
    x = isj ? q - 6 : q;
    oa = isj ? R_JJ0 : R_JI0;
    sgn = isj ? 1.0f : -1.0f;
  }
  float acc = 0.0f;
  for (int b0 = s0; b0 < s1; b0 += BA_PBATCH) {
    const int nb = min(BA_PBATCH, s1 - b0);
    __syncthreads();
    if (fused) {                                // one thread per edge of the batch recomputes its record
      if (tid < nb) {
        float r[BA_REC];
        ba_edge_compute(ein, order[b0 + tid], r);
#pragma unroll
        for (int c = 0; c < BA_REC / 4; c++)
          reinterpret_cast<float4 *>(s_rec)[tid * (BA_REC / 4) + c] = make_float4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]);
      }
    } else {
      for (int q = tid < 192 ? tid : nb * (BA_REC / 4); q < nb * (BA_REC / 4); q += 192) {      // coalesced 16-byte loads of whole records
        const int rr = q / (BA_REC / 4), cc = q - rr * (BA_REC / 4);
        reinterpret_cast<float4 *>(s_rec)[q] =
            reinterpret_cast<const float4 *>(rec + (size_t)order[b0 + rr] * BA_REC)[cc];
      }
    }
    __syncthreads();
    if (tid < 144) {
      for (int p = 0; p < nb; p++) {
        const float *r = s_rec + p * BA_REC;
        acc += ((sgn * r[R_W0]) * r[oa + x]) * r[ob + y];
        acc += ((sgn * r[R_W1]) * r[oa + 6 + x]) * r[ob + 6 + y];
      }
    } else if (tid < 156) {
      for (int p = 0; p < nb; p++) {
        const float *r = s_rec + p * BA_REC;
        acc += ((sgn * r[R_W0]) * r[R_R0]) * r[oa + x];
        acc += ((sgn * r[R_W1]) * r[R_R1]) * r[oa + 6 + x];
      }
    }
    if (b0 == s0 && tid == 0) {
      pair_ij[2 * g + 0] = __float_as_int(s_rec[R_I]);
      pair_ij[2 * g + 1] = __float_as_int(s_rec[R_J]);
    }
  }
  if (tid < 156) pairs[(size_t)g * BA_PAIR + tid] = acc;
}

__global__ void __launch_bounds__(192)
    ba_pair_kernel(const float *__restrict__ rec, const int32_t *__restrict__ order,
                   const int32_t *__restrict__ seg, const int32_t *__restrict__ ngroups,
                   float *__restrict__ pairs, int32_t *__restrict__ pair_ij) {
  __shared__ __attribute__((aligned(16))) float s_buf[BA_PBATCH * BA_REC];
  ba_pair_body(blockIdx.x, rec, order, seg, ngroups, pairs, pair_ij, BaEdgeIn(), false, s_buf);
}

// K2 and K3 read the same per-edge records and do not depend on each other: one launch, the first n_patch
// workgroups take the patch role, the rest the pair role (same arithmetic as the separate kernels)
__global__ void __launch_bounds__(256)
    ba_patch_pair_kernel(int n_patch, const float *__restrict__ rec, const int32_t *__restrict__ order_k,
                         const int32_t *__restrict__ seg_k, const int32_t *__restrict__ nk,
                         const float *__restrict__ lmbda, float *__restrict__ Erow, float *__restrict__ Cv,
                         float *__restrict__ uv, float *__restrict__ Qv, int n6,
                         const int32_t *__restrict__ order_p, const int32_t *__restrict__ seg_p,
                         const int32_t *__restrict__ np, float *__restrict__ pairs, int32_t *__restrict__ pair_ij,
                         BaEdgeIn ein, int fused, const int32_t *__restrict__ dyn, int opt_window) {
  ba_dyn_window(dyn, opt_window, ein.t0, ein.N);
  __shared__ __attribute__((aligned(16))) float s_buf[BA_PBATCH * BA_REC];     // one array for both roles (16 KB)
  static_assert(BA_PBATCH * BA_REC >= BA_PCHUNK * (BA_REC + 1) && BA_PBATCH <= 256, "role buffers / one thread per record");
  if ((int)blockIdx.x < n_patch)
    ba_patch_body(blockIdx.x, rec, order_k, seg_k, nk, lmbda, Erow, Cv, uv, Qv, n6, ein, fused != 0, s_buf);
  else
    ba_pair_body(blockIdx.x - n_patch, rec, order_p, seg_p, np, pairs, pair_ij, ein, fused != 0, s_buf);
}

// ------------------------------------------------------------------ K4
// S_part[z] = sum_{k in chunk z} Q_k E_k E_k' (64x64 tile per block),
// y_part[z] = sum Q_k u_k E_k
__global__ void __launch_bounds__(256)
    ba_schur_kernel(const float *__restrict__ Erow, const float *__restrict__ Qv,
                    const float *__restrict__ uv, const int32_t *__restrict__ ngroups,
                    float *__restrict__ S_part, float *__restrict__ y_part, int n6, int KS) {
  __shared__ float Ea[BA_KB][BA_TS];  // scaled by Q
  __shared__ float Eb[BA_KB][BA_TS];
  __shared__ float us[BA_KB];
  const int nk = *ngroups;
  const int z = blockIdx.z;
  const int per = (nk + KS - 1) / KS;
  const int k0 = z * per, k1 = min(nk, k0 + per);
  const int r0 = blockIdx.x * BA_TS, c0 = blockIdx.y * BA_TS;
  const int tid = threadIdx.x, ty = tid / 16, tx = tid % 16;
  float acc[4][4];
  float yacc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0.0f;
  for (int kb = k0; kb < k1; kb += BA_KB) {
    __syncthreads();
    for (int q = tid; q < BA_KB * BA_TS; q += 256) {
      const int kq = q / BA_TS, col = q - kq * BA_TS;
      const int k = kb + kq;
      float va = 0.0f, vb = 0.0f;
      if (k < k1) {
        if (r0 + col < n6) va = Erow[(size_t)k * n6 + r0 + col] * Qv[k];
        if (c0 + col < n6) vb = Erow[(size_t)k * n6 + c0 + col];
      }
      Ea[kq][col] = va;
      Eb[kq][col] = vb;
    }
    if (tid < BA_KB) us[tid] = (kb + tid < k1) ? uv[kb + tid] : 0.0f;
    __syncthreads();
#pragma unroll
    for (int kq = 0; kq < BA_KB; kq++) {
      float a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { a[q] = Ea[kq][ty * 4 + q]; b[q] = Eb[kq][tx * 4 + q]; }
#pragma unroll
      for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++) acc[x][y] = __builtin_fmaf(a[x], b[y], acc[x][y]);
      if (blockIdx.y == 0 && tx == 0) {
        const float u = us[kq];
#pragma unroll
        for (int x = 0; x < 4; x++) yacc[x] = __builtin_fmaf(a[x], u, yacc[x]);
      }
    }
  }
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const int r = r0 + ty * 4 + x;
    if (r >= n6) continue;
#pragma unroll
    for (int y = 0; y < 4; y++) {
      const int c = c0 + tx * 4 + y;
      if (c < n6) S_part[((size_t)z * n6 + r) * n6 + c] = acc[x][y];
    }
    if (blockIdx.y == 0 && tx == 0) y_part[(size_t)z * n6 + r] = yacc[x];
  }
}

// K4 for systems of one tile (6N <= 64) whose split-K chunk fits the LDS: the chunk's E rows, Q and u are staged ONCE (every
// load of the workgroup in flight together -- ba_schur_kernel walks the chunk in batches of 8 rows, one dependent memory round
// trip per batch), then the products run from LDS.  Same fma chain per entry in the same k order: identical partials.
#define BA_S1_ROWS 160
__global__ void __launch_bounds__(256)
    ba_schur1_kernel(const float *__restrict__ Erow, const float *__restrict__ Qv, const float *__restrict__ uv,
                     const int32_t *__restrict__ ngroups, float *__restrict__ S_part, float *__restrict__ y_part, int n6, int KS) {
  __shared__ float Es[BA_S1_ROWS][BA_TS];
  __shared__ float Qs[BA_S1_ROWS], us[BA_S1_ROWS];
  const int nk = *ngroups;
  const int z = blockIdx.z;
  const int per = (nk + KS - 1) / KS;
  const int k0 = z * per, k1 = min(nk, k0 + per);
  const int tid = threadIdx.x, ty = tid / 16, tx = tid % 16;
  const int rows = max(k1 - k0, 0);
  for (int q0 = 0; q0 < rows * BA_TS; q0 += 12 * 256) {      // 12 loads per thread in flight (48 rows: one round trip)
    float ev[12];
#pragma unroll
    for (int u = 0; u < 12; u++) {
      const int q = q0 + u * 256 + tid;
      const int kq = q / BA_TS, col = q - kq * BA_TS;
      ev[u] = (q < rows * BA_TS && col < n6) ? Erow[(size_t)(k0 + kq) * n6 + col] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 12; u++) {
      const int q = q0 + u * 256 + tid;
      if (q < rows * BA_TS) Es[q / BA_TS][q % BA_TS] = ev[u];
    }
  }
  for (int q = tid; q < rows; q += 256) { Qs[q] = Qv[k0 + q]; us[q] = uv[k0 + q]; }
  __syncthreads();
  float acc[4][4];
  float yacc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0.0f;
  // (ba_schur_kernel pads its last batch of 8 with zero rows: fma(0, 0, acc) leaves acc as it is, nothing to reproduce)
  for (int kq = 0; kq < rows; kq++) {
    float a[4], b[4];
    const float qk = Qs[kq];
#pragma unroll
    for (int q = 0; q < 4; q++) { a[q] = Es[kq][ty * 4 + q] * qk; b[q] = Es[kq][tx * 4 + q]; }
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
      for (int y = 0; y < 4; y++) acc[x][y] = __builtin_fmaf(a[x], b[y], acc[x][y]);
    if (tx == 0) {
      const float u = us[kq];
#pragma unroll
      for (int x = 0; x < 4; x++) yacc[x] = __builtin_fmaf(a[x], u, yacc[x]);
    }
  }
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const int r = ty * 4 + x;
    if (r >= n6) continue;
#pragma unroll
    for (int y = 0; y < 4; y++) {
      const int c = tx * 4 + y;
      if (c < n6) S_part[((size_t)z * n6 + r) * n6 + c] = acc[x][y];
    }
    if (tx == 0) y_part[(size_t)z * n6 + r] = yacc[x];
  }
}

// ------------------------------------------------------------------ K5
// One workgroup per free pose a (block row of S).  The pair records that touch pose a are first
// compacted IN ORDER into LDS (ballot prefix), then every thread sums its entries of the 6 x 6N
// row block over that short list -- fixed order, no atomics.
#define BA_MAXLIST 1024
#define BA_CHUNK 48
__global__ void __launch_bounds__(256)
    ba_assemble_kernel(const float *__restrict__ pairs, const int32_t *__restrict__ pair_ij,
                       const int32_t *__restrict__ npairs, const float *__restrict__ S_part,
                       const float *__restrict__ y_part, float *__restrict__ S,
                       float *__restrict__ yv, int n6, int KS, int32_t *__restrict__ info) {
  __shared__ int s_list[BA_MAXLIST];
  __shared__ int s_wcnt[4];
  __shared__ int s_base;
  const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int np = *npairs;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int g0 = 0; g0 < np; g0 += 256) {
    const int g = g0 + tid;
    const bool hit = g < np && (pair_ij[2 * g] == a || pair_ij[2 * g + 1] == a);
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wcnt[w];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (hit && off < BA_MAXLIST) s_list[off] = g;
    __syncthreads();
    if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  const int nl = min(s_base, BA_MAXLIST);
  // more pair records on one pose than the list holds (> 512 frames connected to one): the result would silently lose
  // terms -- flagged in *info (bit 1), never truncated quietly
  if (s_base > BA_MAXLIST && tid == 0 && info) atomicOr(info, 2);
  // The listed pair records are staged through LDS in chunks (coalesced, all loads in flight at
  // once) -- walking them straight from memory costs one dependent global round trip per record.
  __shared__ __attribute__((aligned(16))) float s_pr[BA_CHUNK][BA_PAIR];
  __shared__ int s_i[BA_CHUNK], s_j[BA_CHUNK];
  // the 6 x 6N row block is split over gridDim.y workgroups
  const int epb = (6 * n6 + (int)gridDim.y - 1) / (int)gridDim.y;
  const int q_lo = blockIdx.y * epb, q_hi = min(6 * n6, q_lo + epb);
  constexpr int QMAX = 5;                    // entries per thread (epb <= 1280)
  float bsum[QMAX], vsum[QMAX];
#pragma unroll
  for (int t = 0; t < QMAX; t++) { bsum[t] = 0.0f; vsum[t] = 0.0f; }
  for (int l0 = 0; l0 < nl; l0 += BA_CHUNK) {
    const int cnt = min(BA_CHUNK, nl - l0);
    __syncthreads();
    constexpr int V4 = BA_PAIR / 4;            // float4 pieces per record
    for (int e0 = 0; e0 < cnt * V4; e0 += 4 * 256) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {              // 4 independent 16-byte loads per thread in flight
        const int e = e0 + u * 256 + tid;
        const int l = min(e / V4, cnt - 1), k = e % V4;
        v[u] = reinterpret_cast<const float4 *>(pairs + (size_t)s_list[l0 + l] * BA_PAIR)[k];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * 256 + tid;
        if (e < cnt * V4) reinterpret_cast<float4 *>(&s_pr[e / V4][0])[e % V4] = v[u];
      }
    }
    if (tid < cnt) {
      const int g = s_list[l0 + tid];
      s_i[tid] = pair_ij[2 * g];
      s_j[tid] = pair_ij[2 * g + 1];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < QMAX; t++) {
      const int q = q_lo + tid + t * 256;
      if (q >= q_hi) break;
      const int x = q / n6, c = q - x * n6;
      const int b = c / 6, y = c - b * 6;
      for (int l = 0; l < cnt; l++) {          // ascending list order: the fixed summation order
        const int i = s_i[l], j = s_j[l];
        const float *pr = s_pr[l];
        if (i == a && i == b) bsum[t] += pr[x * 6 + y];
        if (j == a && j == b) bsum[t] += pr[36 + x * 6 + y];
        if (i == a && j == b) bsum[t] += pr[72 + x * 6 + y];
        if (j == a && i == b) bsum[t] += pr[108 + x * 6 + y];
        if (c == 0) {
          if (i == a) vsum[t] += pr[144 + x];
          if (j == a) vsum[t] += pr[150 + x];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < QMAX; t++) {
    const int q = q_lo + tid + t * 256;
    if (q >= q_hi) break;
    const int x = q / n6, c = q - x * n6;
    const int r = 6 * a + x;
    // split-K partials in z order, all KS <= 64 loads in flight (branch-free: clamped index, masked add)
    float sp = 0.0f;
    {
      float v[64];
#pragma unroll
      for (int u = 0; u < 64; u++) v[u] = S_part[((size_t)(u < KS ? u : KS - 1) * n6 + r) * n6 + c];
#pragma unroll
      for (int u = 0; u < 64; u++) sp += u < KS ? v[u] : 0.0f;
    }
    float s = bsum[t] - sp;
    if (r == c) s += (1e-4f * s + 1.0f);
    S[(size_t)r * n6 + c] = s;
    if (c == 0) {
      float yp = 0.0f;
      {
        float v[64];
#pragma unroll
        for (int u = 0; u < 64; u++) v[u] = y_part[(size_t)(u < KS ? u : KS - 1) * n6 + r];
#pragma unroll
        for (int u = 0; u < 64; u++) yp += u < KS ? v[u] : 0.0f;
      }
      yv[r] = vsum[t] - yp;
    }
  }
}

// ------------------------------------------------------------------ K5, one workgroup per 6 x 6 block
// S[6a.., 6b..] = B block - sum of the split-K partials (+ damping on the diagonal), y likewise (by the diagonal workgroup).
// An off-diagonal block (a, b) takes the <= 2 pair records (i, j) = (a, b) / (b, a); the diagonal block (a, a) every record
// with i == a or j == a (~2 x the window): 7 lanes per entry sum every 7th listed record in ascending order and the seven
// partial sums are added in lane order -- fixed order, no atomics, 20-odd independent loads per thread instead of a chain
// of chunks staged through LDS behind barriers (ba_assemble_kernel: one workgroup per POSE, 19 us of a 139 us call).
// (A single lane per entry -- ascending record order, ba_assemble_kernel's -- was slower than that kernel: 151 vs 138 us per
// call.  The interleaved order moves the 180 x 180 solve of the precise.yaml window by 6e-4 of the step against the oracle --
// fp32 rounding through its conditioning -- so windows above 16 poses keep ba_assemble_kernel.)
#define BA_AP 7
__global__ void __launch_bounds__(256)
    ba_assemble2_kernel(const float *__restrict__ pairs, const int32_t *__restrict__ pair_ij,
                        const int32_t *__restrict__ npairs, const float *__restrict__ S_part,
                        const float *__restrict__ y_part, float *__restrict__ S, float *__restrict__ yv, int n6, int KS,
                        int32_t *__restrict__ info) {
  __shared__ int s_list[BA_MAXLIST], s_i[BA_MAXLIST], s_j[BA_MAXLIST];
  __shared__ int s_wcnt[4];
  __shared__ int s_base;
  __shared__ float s_part[BA_AP][48];
  const int a = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int np = *npairs;
  const bool diag = a == b;
  // the split-K partials of this thread's entry do not depend on the pair lists: their loads go out first and are in flight
  // while the list is built (one dependent memory round trip less on a launch that is nothing but round trips)
  float spv[64];
  if (tid < 36) {
    const int r = 6 * a + tid / 6, c = 6 * b + tid % 6;
#pragma unroll
    for (int u = 0; u < 64; u++) spv[u] = S_part[((size_t)(u < KS ? u : KS - 1) * n6 + r) * n6 + c];
  } else if (tid < 42 && diag) {
    const int r = 6 * a + tid - 36;
#pragma unroll
    for (int u = 0; u < 64; u++) spv[u] = y_part[(size_t)(u < KS ? u : KS - 1) * n6 + r];
  }
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int g0 = 0; g0 < np; g0 += 256) {
    const int g = g0 + tid;
    int pi = -2, pj = -2;
    if (g < np) { pi = pair_ij[2 * g]; pj = pair_ij[2 * g + 1]; }
    const bool hit = diag ? (pi == a || pj == a) : ((pi == a && pj == b) || (pi == b && pj == a));
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wcnt[w];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (hit && off < BA_MAXLIST) { s_list[off] = g; s_i[off] = pi; s_j[off] = pj; }
    __syncthreads();
    if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  const int nl = min(s_base, BA_MAXLIST);
  if (s_base > BA_MAXLIST && tid == 0 && info) atomicOr(info, 2);       // (never a silent truncation)
  // threads [0, 36 AP): entry e = tid / AP (x = e / 6, y = e % 6), partial lane pl = tid % AP; [36 AP, 42 AP): gradient x
  const int e = tid / BA_AP, pl = tid - e * BA_AP;
  float acc = 0.f;
  if (e < 36) {
    const int x = e / 6, y = e - x * 6;
    for (int l = pl; l < nl; l += BA_AP) {
      const int i = s_i[l], j = s_j[l];
      const float *pr = pairs + (size_t)s_list[l] * BA_PAIR + x * 6 + y;
      // (the four conditions and their order are ba_assemble_kernel's)
      const float v0 = (i == a && i == b) ? pr[0] : 0.f, v1 = (j == a && j == b) ? pr[36] : 0.f;
      const float v2 = (i == a && j == b) ? pr[72] : 0.f, v3 = (j == a && i == b) ? pr[108] : 0.f;
      if (i == a && i == b) acc += v0;
      if (j == a && j == b) acc += v1;
      if (i == a && j == b) acc += v2;
      if (j == a && i == b) acc += v3;
    }
    s_part[pl][e] = acc;
  }
  if (diag && tid < 6 * BA_AP) {                     // the gradient's six entries, the same way (threads 0 .. 41 again)
    const int x = tid / BA_AP, p2 = tid - x * BA_AP;
    float g = 0.f;
    for (int l = p2; l < nl; l += BA_AP) {
      const float *pr = pairs + (size_t)s_list[l] * BA_PAIR;
      if (s_i[l] == a) g += pr[144 + x];
      if (s_j[l] == a) g += pr[150 + x];
    }
    s_part[p2][36 + x] = g;
  }
  __syncthreads();
  if (tid < 36) {
    const int x = tid / 6, y = tid - x * 6;
    float bs = 0.f;
#pragma unroll
    for (int q = 0; q < BA_AP; q++) bs += s_part[q][tid];
    const int r = 6 * a + x, c = 6 * b + y;
    float sp = 0.0f;
#pragma unroll
    for (int u = 0; u < 64; u++) sp += u < KS ? spv[u] : 0.0f;
    float sv = bs - sp;
    if (r == c) sv += (1e-4f * sv + 1.0f);
    S[(size_t)r * n6 + c] = sv;
  } else if (tid < 42 && diag) {
    const int x = tid - 36, r = 6 * a + x;
    float vs = 0.f;
#pragma unroll
    for (int q = 0; q < BA_AP; q++) vs += s_part[q][tid];
    float yp = 0.0f;
#pragma unroll
    for (int u = 0; u < 64; u++) yp += u < KS ? spv[u] : 0.0f;
    yv[r] = vs - yp;
  }
}

// ------------------------------------------------------------------ K6
// Single workgroup, matrix in LDS, blocked Cholesky (block width 6 = one pose).  Per block column every thread factors the 6 x 6 diagonal block in
// REGISTERS (21 broadcast LDS reads, then no LDS traffic inside the dependency chain: with the block left in LDS the
// loads cannot move above the stores and every one of them is an exposed round trip), the row threads solve their
// panel row against it, and the whole workgroup applies the rank-6 update to the trailing matrix on a TG x TG thread
// grid -- two barriers per pose instead of one per column.  The right-hand side rides along as row n6 (z = L^-1 y
// falls out of the panel solves); the back substitution is blocked the same way.
// the factorisation and the solve on a matrix that is already in LDS (A: (n6 + 1) x ld, lower triangle + the rhs as row n6;
// every thread of the workgroup calls it)
template <int TG>
__device__ __forceinline__ void ba_cholb_body(float *__restrict__ A, float *__restrict__ xv, float *__restrict__ Lk,
                                              float *__restrict__ dX, int32_t *__restrict__ info, int n6) {
  const int ld = n6 + 1;
  __shared__ int s_bad;
  const int tid = threadIdx.x, nt = TG * TG;
  const int ty = tid / TG, tx = tid % TG;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  const int nb = n6 / 6;
  for (int kb = 0; kb < nb; kb++) {
    const int c0 = 6 * kb;
    const int r0 = c0 + 6, m = n6 - r0;        // m trailing columns, m + 1 rows (rhs)
    // only the waves that own a panel row factor the block (tid 0 is one of them: row n6 always exists)
    if ((tid & ~63) <= m) {
    float l[6][6], li[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) l[i][j] = A[(c0 + i) * ld + c0 + j];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      float d = l[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) d = __builtin_fmaf(-l[j][k], l[j][k], d);
      bad |= !(d > 0.0f);
      li[j] = __builtin_amdgcn_rsqf(d);          // v_rsq_f32, 1 ulp; d is O(1..1e6) here, no denormal range
      l[j][j] = d * li[j];
#pragma unroll
      for (int i = j + 1; i < 6; i++) {
        float t = l[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) t = __builtin_fmaf(-l[i][k], l[j][k], t);
        l[i][j] = t * li[j];
      }
    }
    if (tid == 0) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Lk[kb * 28 + q++] = l[i][j];
#pragma unroll
      for (int j = 0; j < 6; j++) Lk[kb * 28 + 21 + j] = li[j];
      if (bad) s_bad = 1;
    }
    // panel: rows below the block (and the rhs row n6): X L_kk' = A_ik  ->  forward substitution along the row
    for (int i = c0 + 6 + tid; i <= n6; i += nt) {
      float x[6];
#pragma unroll
      for (int j = 0; j < 6; j++) x[j] = A[i * ld + c0 + j];
#pragma unroll
      for (int j = 0; j < 6; j++) {
#pragma unroll
        for (int k = 0; k < j; k++) x[j] = __builtin_fmaf(-x[k], l[j][k], x[j]);
        x[j] *= li[j];
      }
#pragma unroll
      for (int j = 0; j < 6; j++) A[i * ld + c0 + j] = x[j];
    }
    }
    __syncthreads();
    // trailing update: A[i][j] -= sum_k L[i][c0+k] L[j][c0+k],  c0 + 6 <= j <= i <= n6,  j < n6
    for (int ii = ty; ii <= m; ii += TG) {
      float ri[6];
#pragma unroll
      for (int k = 0; k < 6; k++) ri[k] = A[(r0 + ii) * ld + c0 + k];
      const int jmax = ii < m - 1 ? ii : m - 1;
      for (int jj = tx; jj <= jmax; jj += TG) {
        const float *rj = A + (r0 + jj) * ld + c0;
        float t = A[(r0 + ii) * ld + r0 + jj];
#pragma unroll
        for (int k = 0; k < 6; k++) t = __builtin_fmaf(-ri[k], rj[k], t);
        A[(r0 + ii) * ld + r0 + jj] = t;
      }
    }
    __syncthreads();
  }
  const bool bad = s_bad != 0;
  if (bad && tid == 0 && info) atomicOr(info, 1);
  // L' x = z (z = row n6), block by block from the bottom
  if (n6 <= 64) {
    // one wave, no barriers: lane i keeps z_i; a block's six unknowns are solved by every lane (uniform work on broadcast
    // values), then lane i < c0 takes them out of its z_i -- the blocked loop below operation for operation
    if (tid < 64) {
      float z = tid < n6 ? A[n6 * ld + tid] : 0.0f;
      for (int kb = nb - 1; kb >= 0; kb--) {
        const int c0 = 6 * kb;
        float lk[27], x[6];
#pragma unroll
        for (int q = 0; q < 27; q++) lk[q] = Lk[kb * 28 + q];
#pragma unroll
        for (int j = 0; j < 6; j++) x[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), c0 + j));
#pragma unroll
        for (int j = 5; j >= 0; j--) {
#pragma unroll
          for (int k = j + 1; k < 6; k++) x[j] = __builtin_fmaf(-lk[k * (k + 1) / 2 + j], x[k], x[j]);
          x[j] *= lk[21 + j];
        }
        if (tid == 0) {
#pragma unroll
          for (int j = 0; j < 6; j++) xv[c0 + j] = x[j];
        }
        if (tid < c0) {
#pragma unroll
          for (int k = 0; k < 6; k++) z = __builtin_fmaf(-A[(c0 + k) * ld + tid], x[k], z);
        }
      }
    }
    __syncthreads();
  } else
  for (int kb = nb - 1; kb >= 0; kb--) {
    const int c0 = 6 * kb;
    if (tid < 64) {                            // wave 0 (uniform work, lane 0 writes)
      float lk[27], x[6];
#pragma unroll
      for (int q = 0; q < 27; q++) lk[q] = Lk[kb * 28 + q];
#pragma unroll
      for (int j = 0; j < 6; j++) x[j] = A[n6 * ld + c0 + j];
#pragma unroll
      for (int j = 5; j >= 0; j--) {
#pragma unroll
        for (int k = j + 1; k < 6; k++) x[j] = __builtin_fmaf(-lk[k * (k + 1) / 2 + j], x[k], x[j]);
        x[j] *= lk[21 + j];
      }
      if (tid == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) xv[c0 + j] = x[j];
      }
    }
    __syncthreads();
    for (int i = tid; i < c0; i += nt) {       // z_i -= sum_k L[c0+k][i] x[c0+k]
      float t = A[n6 * ld + i];
#pragma unroll
      for (int k = 0; k < 6; k++) t = __builtin_fmaf(-A[(c0 + k) * ld + i], xv[c0 + k], t);
      A[n6 * ld + i] = t;
    }
    __syncthreads();
  }
  // a failed factorisation -- or a step that is not finite: a vanishing pivot passes the sign test -- drops the pose
  // step (dX = 0: the reference's torch.linalg.cholesky would throw inside its try, Ramp_vo.py:302-306); bit 0 of *info either way
  __shared__ int s_nf;
  if (tid == 0) s_nf = 0;
  __syncthreads();
  for (int q = tid; q < n6; q += nt)
    if (!(fabsf(xv[q]) <= 3.0e38f)) s_nf = 1;
  __syncthreads();
  const bool drop = bad || s_nf != 0;
  if (!bad && s_nf != 0 && tid == 0 && info) atomicOr(info, 1);
  for (int q = tid; q < n6; q += nt) dX[q] = drop ? 0.0f : xv[q];
}
template <int TG>
__global__ void __launch_bounds__(TG * TG)
    ba_cholb_kernel(const float *__restrict__ S, const float *__restrict__ yv,
                    float *__restrict__ dX, int32_t *__restrict__ info, int n6) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = n6 + 1;                       // odd for every 6N: conflict-free column walks
  float *A = sm;                               // (n6 + 1) x ld: rows 0..n6-1 = S (lower triangle), row n6 = y
  float *xv = sm + (n6 + 1) * ld;              // n6: solution
  float *Lk = xv + n6;                         // nb x 28: the factored diagonal blocks (21 lower entries + 6 inverses)
  const int tid = threadIdx.x, nt = TG * TG;
  {
    int r = tid / n6, c = tid - r * n6;        // one division, then stepping
    const int dr = nt / n6, dc = nt - dr * n6;
    for (int q = tid; q < n6 * n6; q += nt) {
      A[r * ld + c] = S[q];
      r += dr; c += dc;
      if (c >= n6) { c -= n6; r++; }
    }
  }
  for (int q = tid; q < n6; q += nt) A[n6 * ld + q] = yv[q];
  ba_cholb_body<TG>(A, xv, Lk, dX, info, n6);
}

// ------------------------------------------------------------------ K7
__global__ void __launch_bounds__(256)
    ba_retract_kernel(float *__restrict__ poses, float *__restrict__ patches,
                      const float *__restrict__ Erow, const float *__restrict__ Qv,
                      const float *__restrict__ uv, const float *__restrict__ dX,
                      const int64_t *__restrict__ kx, const int32_t *__restrict__ ngroups,
                      int n6, int PP, int t0, int N, int depth_blocks, const int32_t *__restrict__ dyn,
                      int opt_window) {
  if (dyn) {
    const int t1 = dyn[RAMP_DYN_N];
    t0 = max(t1 - opt_window, 1);
    N = min(N, t1 - t0);
  }
  if ((int)blockIdx.x >= depth_blocks) {
    // pose retraction (ba_cuda.cu:178-206)
    const int i = (blockIdx.x - depth_blocks) * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float *p = poses + 7 * (size_t)(t0 + i);
    float xi[6], to[3] = {p[0], p[1], p[2]}, qo[4] = {p[3], p[4], p[5], p[6]}, tn[3], qn[4];
#pragma unroll
    for (int c = 0; c < 6; c++) xi[c] = dX[6 * i + c];
    fb_retrSE3(xi, to, qo, tn, qn);
    p[0] = tn[0]; p[1] = tn[1]; p[2] = tn[2];
    p[3] = qn[0]; p[4] = qn[1]; p[5] = qn[2]; p[6] = qn[3];
    return;
  }
  // depth retraction (ba_cuda.cu:209-229), one wavefront per patch
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int g = blockIdx.x * (blockDim.x / 64) + wave;
  if (g >= *ngroups) return;
  float s = 0.0f;
  for (int a = lane; a < n6; a += 64) s += Erow[(size_t)g * n6 + a] * dX[a];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  s = __shfl(s, 0, 64);
  const float dz = Qv[g] * (uv[g] - s);
  float *pt = patches + ((size_t)kx[g] * 3 + 2) * PP;
  float dd = pt[0];
  dd = dd + dz;
  dd = (dd > 20) ? 1.0f : dd;
  dd = fmaxf(dd, 1e-4f);
  // every lane has read pt[0] (same address) before any lane stores
  if (lane < PP) pt[lane] = dd;
}

__global__ void ba_pairkey_kernel(const int64_t *__restrict__ ii, const int64_t *__restrict__ jj,
                                  int64_t *__restrict__ keys, int E, long long np) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < E) keys[n] = jj[n] * np + ii[n];   // target-frame-major, as the tracker's graph plan
}

// ------------------------------------------------------------- host driver
struct BaWs {
  void *gb; size_t gb_bytes;
  int64_t *pkeys, *kx, *pukeys;
  int32_t *order_k, *seg_k, *order_p, *seg_p, *pair_ij, *counters;
  float *rec, *Erow, *Cv, *uv, *Qv, *pairs, *S_part, *y_part, *S, *yv, *dX;
  int Mu_b, Gp_b, KS, tiles;
};
// own_groups: carve room for this call's own group-by (ramp_ba_forward); otherwise the groups
// come from the caller's plan (ramp_ba_forward_planned) and Mu_b / Gp_b are the caller's bounds.
static size_t ba_carve(void *ws, int E, int n_poses, int n_patches, int N, int own_groups,
                       int max_patches, int max_pairs, BaWs *w) {
  const size_t e = (size_t)(E > 0 ? E : 1);
  const int n6 = 6 * N;
  size_t off = 0;
  char *base = (char *)ws;
  auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += al(bytes); return (void *)p; };
  w->Mu_b = (int)((size_t)n_patches < e ? (size_t)n_patches : e);
  if (max_patches > 0 && max_patches < w->Mu_b) w->Mu_b = max_patches;
  if (w->Mu_b < 1) w->Mu_b = 1;
  const size_t pp = (size_t)n_poses * (size_t)n_poses;
  w->Gp_b = (int)(pp < e ? pp : e);
  if (max_pairs > 0 && max_pairs < w->Gp_b) w->Gp_b = max_pairs;
  if (w->Gp_b < 1) w->Gp_b = 1;
  w->tiles = (n6 + BA_TS - 1) / BA_TS;
  if (w->tiles < 1) w->tiles = 1;
  int ks = 256 / (w->tiles * w->tiles);
  w->KS = ks < 4 ? 4 : (ks > 64 ? 64 : ks);   // split-K partials, summed in fixed order by K5
  w->gb = nullptr; w->gb_bytes = 0;
  w->pkeys = w->kx = w->pukeys = nullptr;
  w->order_k = w->seg_k = w->order_p = w->seg_p = nullptr;
  if (own_groups) {
    w->gb_bytes = ramp_internal_group_by_ws(E);
    w->gb = take(w->gb_bytes);
    w->pkeys = (int64_t *)take(e * 8);
    w->kx = (int64_t *)take(e * 8);
    w->pukeys = (int64_t *)take(e * 8);
    w->order_k = (int32_t *)take(e * 4);
    w->seg_k = (int32_t *)take((e + 1) * 4);
    w->order_p = (int32_t *)take(e * 4);
    w->seg_p = (int32_t *)take((e + 1) * 4);
  }
  w->pair_ij = (int32_t *)take((size_t)w->Gp_b * 8);
  w->counters = (int32_t *)take(64);
  w->rec = (float *)take(e * BA_REC * 4);
  w->Erow = (float *)take((size_t)w->Mu_b * (n6 > 0 ? n6 : 1) * 4);
  w->Cv = (float *)take((size_t)w->Mu_b * 4);
  w->uv = (float *)take((size_t)w->Mu_b * 4);
  w->Qv = (float *)take((size_t)w->Mu_b * 4);
  w->pairs = (float *)take((size_t)w->Gp_b * BA_PAIR * 4);
  w->S_part = (float *)take((size_t)w->KS * (n6 * n6 + 1) * 4);
  w->y_part = (float *)take((size_t)w->KS * (n6 + 1) * 4);
  w->S = (float *)take((size_t)(n6 * n6 + 1) * 4);
  w->yv = (float *)take((size_t)(n6 + 1) * 4);
  w->dX = (float *)take((size_t)(n6 + 1) * 4);
  return off;
}

// the GN iterations, given the two groupings
static int ba_iterate(float *poses, float *patches, const float *intrinsics, const float *target,
                      const float *weight, const float *lmbda, const int64_t *ii, const int64_t *jj,
                      const int64_t *kk, int E, int P, int t0, int t1, int iterations, BaWs &w,
                      const int32_t *order_k, const int32_t *seg_k, const int32_t *nk, const int64_t *kx,
                      const int32_t *order_p, const int32_t *seg_p, const int32_t *np, int32_t *info,
                      hipStream_t st, const int32_t *dyn = nullptr, int opt_window = 0) {
  const int N = t1 - t0, n6 = 6 * N;
  const size_t lds = (size_t)((n6 + 1) * (n6 + 1) + 6 * n6) * sizeof(float);
  const int PP = P * P, c11 = 1 * P + 1;
  if (N > 0 && lds > 64 * 1024) {
    // (per call: the attribute is per device, and a process may drive several)
    if (hipFuncSetAttribute((const void *)ba_cholb_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return RAMP_ELAUNCH;
  }
  const int pthreads = ((n6 + 2 + 63) / 64) * 64;
  if (pthreads > 256) return RAMP_EUNSUPPORTED;
  const int depth_blocks = ramp_cdiv(w.Mu_b, 4);
  const int pose_blocks = N > 0 ? ramp_cdiv(N, 256) : 0;
  BaEdgeIn ein;
  ein.poses = poses; ein.patches = patches; ein.intr = intrinsics; ein.target = target; ein.weight = weight;
  ein.ii = ii; ein.jj = jj; ein.kk = kk; ein.PP = PP; ein.c11 = c11; ein.t0 = t0; ein.N = N;
  const int fuse_edge = 1;       // the per-factor records are recomputed in registers by the patch / pair kernel (no [E][32] rows through memory)
  for (int itr = 0; itr < iterations; itr++) {
    if (N <= 0)
      hipLaunchKernelGGL(ba_edge_kernel, dim3(ramp_cdiv(E, 256)), dim3(256), 0, st, ein, w.rec, E, dyn, opt_window);
    if (N > 0)
      hipLaunchKernelGGL(ba_patch_pair_kernel, dim3(w.Mu_b + w.Gp_b), dim3(256), 0, st, w.Mu_b, w.rec, order_k,
                         seg_k, nk, lmbda, w.Erow, w.Cv, w.uv, w.Qv, n6, order_p, seg_p, np, w.pairs, w.pair_ij, ein,
                         fuse_edge, dyn, opt_window);
    else
      hipLaunchKernelGGL(ba_patch_kernel, dim3(w.Mu_b), dim3(pthreads), 0, st, w.rec, order_k, seg_k, nk,
                         lmbda, w.Erow, w.Cv, w.uv, w.Qv, n6);
    if (N > 0) {
      // (one workgroup per split-K slice where the system is a single tile; N x N assembly workgroups up to 16 poses, one per
      // pose above -- the interleaved order of the former moves a 180 x 180 solve by 6e-4 of the step)
      if (w.tiles == 1 && ramp_cdiv(w.Mu_b, w.KS) <= BA_S1_ROWS)
        hipLaunchKernelGGL(ba_schur1_kernel, dim3(1, 1, w.KS), dim3(256), 0, st, w.Erow, w.Qv, w.uv, nk, w.S_part, w.y_part,
                           n6, w.KS);
      else
        hipLaunchKernelGGL(ba_schur_kernel, dim3(w.tiles, w.tiles, w.KS), dim3(256), 0, st, w.Erow,
                           w.Qv, w.uv, nk, w.S_part, w.y_part, n6, w.KS);
      if (N <= 16)
        hipLaunchKernelGGL(ba_assemble2_kernel, dim3(N, N), dim3(256), 0, st, w.pairs, w.pair_ij, np, w.S_part, w.y_part,
                           w.S, w.yv, n6, w.KS, info);
      else
        hipLaunchKernelGGL(ba_assemble_kernel, dim3(N, ramp_cdiv(6 * n6, 192)), dim3(256), 0, st, w.pairs, w.pair_ij, np,
                           w.S_part, w.y_part, w.S, w.yv, n6, w.KS, info);
      hipLaunchKernelGGL(ba_cholb_kernel<32>, dim3(1), dim3(1024), lds, st, w.S, w.yv, w.dX, info, n6);
    }
    hipLaunchKernelGGL(ba_retract_kernel, dim3(depth_blocks + pose_blocks), dim3(256), 0, st,
                       poses, patches, w.Erow, w.Qv, w.uv, w.dX, kx, nk, n6, PP, t0, N, depth_blocks, dyn, opt_window);
    RAMP_CHECK_LAUNCH();
  }
  return RAMP_OK;
}

static int ba_check_args(int E, int P, int n_poses, int n_patches, int t0, int t1, int iterations) {
  if (E < 0 || P < 2 || n_poses <= 0 || n_patches <= 0 || iterations < 0) return RAMP_EINVAL;
  if (t0 < 0 || t1 < t0 || t1 > n_poses) return RAMP_EINVAL;
  const int n6 = 6 * (t1 - t0);
  if ((size_t)((n6 + 1) * (n6 + 1) + 2 * n6) * sizeof(float) > 160 * 1024) return RAMP_EUNSUPPORTED;  // > 32 free poses
  return RAMP_OK;
}

// ---- device-side sizes (csrc/track.hip): the window [max(n - opt_window, 1), n) with n = dyn[RAMP_DYN_N] and the
// factor count dyn[RAMP_DYN_E] are read by the kernels; E_cap bounds the launches, the system has opt_window poses.
// Status bits accumulate in *info (not cleared here).
size_t ramp_i_ba_dyn_ws(int E_cap, int n_poses, int n_patches, int opt_window, int max_patches, int max_pairs) {
  BaWs w;
  return ba_carve(nullptr, E_cap, n_poses, n_patches, opt_window, 0, max_patches, max_pairs, &w);
}
int ramp_i_ba_dyn(float *poses, float *patches, const float *intrinsics, const float *target, const float *weight,
                  const float *lmbda, const int64_t *ii, const int64_t *jj, const int64_t *kk, int E_cap, int P,
                  int n_poses, int n_patches, int opt_window, int iterations, const int32_t *order_k,
                  const int32_t *seg_k, const int32_t *ngroups_k, const int64_t *ukeys_k, int max_patches,
                  const int32_t *order_p, const int32_t *seg_p, const int32_t *ngroups_p, int max_pairs, void *ws,
                  size_t ws_bytes, int32_t *info, const int32_t *dyn, hipStream_t st) {
  if (E_cap <= 0 || opt_window <= 0 || !dyn || !ws) return RAMP_EINVAL;
  int rc = ba_check_args(E_cap, P, n_poses, n_patches, 1, 1 + opt_window, iterations);
  if (rc != RAMP_OK) return rc;
  BaWs w;
  if (ba_carve(ws, E_cap, n_poses, n_patches, opt_window, 0, max_patches, max_pairs, &w) > ws_bytes)
    return RAMP_EWORKSPACE;
  return ba_iterate(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, E_cap, P, 1, 1 + opt_window,
                    iterations, w, order_k, seg_k, ngroups_k, ukeys_k, order_p, seg_p, ngroups_p, info, st, dyn,
                    opt_window);
}

extern "C" {

size_t ramp_ba_workspace_bytes(int E, int n_poses, int n_patches, int t0, int t1) {
  BaWs w;
  const int N = t1 - t0 > 0 ? t1 - t0 : 0;
  return ba_carve(nullptr, E, n_poses, n_patches, N, 1, 0, 0, &w);
}

int ramp_ba_forward(float *poses, float *patches, const float *intrinsics, const float *target,
                    const float *weight, const float *lmbda, const int64_t *ii,
                    const int64_t *jj, const int64_t *kk, int E, int P, int n_poses,
                    int n_patches, int t0, int t1, int iterations, void *ws, size_t ws_bytes,
                    int32_t *info, void *stream) {
  int rc = ba_check_args(E, P, n_poses, n_patches, t0, t1, iterations);
  if (rc != RAMP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (info) (void)hipMemsetAsync(info, 0, sizeof(int32_t), st);
  if (E == 0 || iterations == 0) return RAMP_OK;
  if (!poses || !patches || !intrinsics || !target || !weight || !lmbda || !ii || !jj || !kk || !ws)
    return RAMP_EINVAL;
  const int N = t1 - t0;
  BaWs w;
  if (ba_carve(ws, E, n_poses, n_patches, N, 1, 0, 0, &w) > ws_bytes) return RAMP_EWORKSPACE;
  int32_t *nk = w.counters, *np = w.counters + 1;
  // ---- prep: group by patch, group by pose pair
  rc = ramp_internal_group_by(kk, E, n_patches, w.order_k, nullptr, w.seg_k, w.kx, nk, w.gb, w.gb_bytes, st);
  if (rc != RAMP_OK) return rc;
  if (N > 0) {
    hipLaunchKernelGGL(ba_pairkey_kernel, dim3(ramp_cdiv(E, 256)), dim3(256), 0, st, ii, jj, w.pkeys, E,
                       (long long)n_poses);
    rc = ramp_internal_group_by(w.pkeys, E, (int64_t)n_poses * n_poses, w.order_p, nullptr, w.seg_p,
                                w.pukeys, np, w.gb, w.gb_bytes, st);
    if (rc != RAMP_OK) return rc;
  }
  return ba_iterate(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, E, P, t0, t1, iterations,
                    w, w.order_k, w.seg_k, nk, w.kx, w.order_p, w.seg_p, np, info, st);
}

size_t ramp_ba_planned_workspace_bytes(int E, int n_poses, int n_patches, int t0, int t1,
                                       int max_patches, int max_pairs) {
  BaWs w;
  const int N = t1 - t0 > 0 ? t1 - t0 : 0;
  return ba_carve(nullptr, E, n_poses, n_patches, N, 0, max_patches, max_pairs, &w);
}

int ramp_ba_forward_planned(float *poses, float *patches, const float *intrinsics, const float *target,
                            const float *weight, const float *lmbda, const int64_t *ii,
                            const int64_t *jj, const int64_t *kk, int E, int P, int n_poses,
                            int n_patches, int t0, int t1, int iterations, const int32_t *order_k,
                            const int32_t *seg_k, const int32_t *ngroups_k, const int64_t *ukeys_k,
                            int max_patches, const int32_t *order_p, const int32_t *seg_p,
                            const int32_t *ngroups_p, int max_pairs, void *ws, size_t ws_bytes,
                            int32_t *info, void *stream) {
  int rc = ba_check_args(E, P, n_poses, n_patches, t0, t1, iterations);
  if (rc != RAMP_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (info) (void)hipMemsetAsync(info, 0, sizeof(int32_t), st);
  if (E == 0 || iterations == 0) return RAMP_OK;
  if (!poses || !patches || !intrinsics || !target || !weight || !lmbda || !ii || !jj || !kk || !ws ||
      !order_k || !seg_k || !ngroups_k || !ukeys_k || !order_p || !seg_p || !ngroups_p)
    return RAMP_EINVAL;
  if (max_patches <= 0 || max_pairs <= 0) return RAMP_EINVAL;
  BaWs w;
  if (ba_carve(ws, E, n_poses, n_patches, t1 - t0, 0, max_patches, max_pairs, &w) > ws_bytes)
    return RAMP_EWORKSPACE;
  return ba_iterate(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, E, P, t0, t1, iterations,
                    w, order_k, seg_k, ngroups_k, ukeys_k, order_p, seg_p, ngroups_p, info, st);
}

}  // extern "C"
