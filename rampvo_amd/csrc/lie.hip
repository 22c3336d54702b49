// SE3 batched forward ops, the fused projective transform and the fastba
// reprojection kernel.  One lane per group element / edge: these are pure
// latency kernels (a few hundred bytes per element), the win over the
// reference is fusing ~13 launches (inv, mul, 9x broadcast act4, iproj, proj)
// of pops.transform into one.
#include "ramp_device.h"
#include "median.h"

#define LIE_THREADS 256

// ------------------------------------------------------------------ SE3 ops
// reference: lietorch_gpu.cu forward kernels (one thread per element),
// math from se3.h / so3.h (see ramp_device.h)
__global__ void __launch_bounds__(LIE_THREADS) se3_exp_kernel(const float *a, float *X, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xi[6], o[7];
  for (int k = 0; k < 6; k++) xi[k] = a[6 * (size_t)i + k];
  lt_exp(xi, o);
  for (int k = 0; k < 7; k++) X[7 * (size_t)i + k] = o[k];
}
__global__ void __launch_bounds__(LIE_THREADS) se3_log_kernel(const float *X, float *a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[7], o[6];
  for (int k = 0; k < 7; k++) x[k] = X[7 * (size_t)i + k];
  lt_log(x, o);
  for (int k = 0; k < 6; k++) a[6 * (size_t)i + k] = o[k];
}
__global__ void __launch_bounds__(LIE_THREADS) se3_inv_kernel(const float *X, float *Y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[7], o[7];
  for (int k = 0; k < 7; k++) x[k] = X[7 * (size_t)i + k];
  lt_inv(x, o);
  for (int k = 0; k < 7; k++) Y[7 * (size_t)i + k] = o[k];
}
__global__ void __launch_bounds__(LIE_THREADS)
    se3_mul_kernel(const float *X, const float *Y, float *Z, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[7], y[7], o[7];
  for (int k = 0; k < 7; k++) { x[k] = X[7 * (size_t)i + k]; y[k] = Y[7 * (size_t)i + k]; }
  lt_mul(x, y, o);
  for (int k = 0; k < 7; k++) Z[7 * (size_t)i + k] = o[k];
}
__global__ void __launch_bounds__(LIE_THREADS)
    se3_act4_kernel(const float *X, const float *p, float *q, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[7], t[3], r[4], pp[4], o[4];
  for (int k = 0; k < 7; k++) x[k] = X[7 * (size_t)i + k];
  for (int k = 0; k < 4; k++) pp[k] = p[4 * (size_t)i + k];
  lt_load(x, t, r);
  lt_act4_tq(t, r, pp, o);
  for (int k = 0; k < 4; k++) q[4 * (size_t)i + k] = o[k];
}
template <bool TRANSPOSE>
__global__ void __launch_bounds__(LIE_THREADS)
    se3_adj_kernel(const float *X, const float *a, float *b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[7], Ad[36], v[6];
  for (int k = 0; k < 7; k++) x[k] = X[7 * (size_t)i + k];
  for (int k = 0; k < 6; k++) v[k] = a[6 * (size_t)i + k];
  lt_Adj(x, Ad);
#pragma unroll
  for (int r = 0; r < 6; r++) {
    float s = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) s += (TRANSPOSE ? Ad[c * 6 + r] : Ad[r * 6 + c]) * v[c];
    b[6 * (size_t)i + r] = s;
  }
}

#define LIE_LAUNCH(kern, n, ...)                                                        \
  do {                                                                                  \
    if ((n) < 0) return RAMP_EINVAL;                                                    \
    if ((n) == 0) return RAMP_OK;                                                       \
    hipLaunchKernelGGL(kern, dim3(ramp_cdiv((n), LIE_THREADS)), dim3(LIE_THREADS), 0,   \
                       (hipStream_t)stream, __VA_ARGS__);                               \
    RAMP_CHECK_LAUNCH();                                                                \
    return RAMP_OK;                                                                     \
  } while (0)

extern "C" {
int ramp_se3_exp(const float *a, float *X, int n, void *stream) {
  if (n > 0 && (!a || !X)) return RAMP_EINVAL;
  LIE_LAUNCH(se3_exp_kernel, n, a, X, n);
}
int ramp_se3_log(const float *X, float *a, int n, void *stream) {
  if (n > 0 && (!a || !X)) return RAMP_EINVAL;
  LIE_LAUNCH(se3_log_kernel, n, X, a, n);
}
int ramp_se3_inv(const float *X, float *Y, int n, void *stream) {
  if (n > 0 && (!Y || !X)) return RAMP_EINVAL;
  LIE_LAUNCH(se3_inv_kernel, n, X, Y, n);
}
int ramp_se3_mul(const float *X, const float *Y, float *Z, int n, void *stream) {
  if (n > 0 && (!X || !Y || !Z)) return RAMP_EINVAL;
  LIE_LAUNCH(se3_mul_kernel, n, X, Y, Z, n);
}
int ramp_se3_act4(const float *X, const float *p, float *q, int n, void *stream) {
  if (n > 0 && (!X || !p || !q)) return RAMP_EINVAL;
  LIE_LAUNCH(se3_act4_kernel, n, X, p, q, n);
}
int ramp_se3_adj(const float *X, const float *a, float *b, int n, void *stream) {
  if (n > 0 && (!X || !a || !b)) return RAMP_EINVAL;
  LIE_LAUNCH(se3_adj_kernel<false>, n, X, a, b, n);
}
int ramp_se3_adjT(const float *X, const float *a, float *b, int n, void *stream) {
  if (n > 0 && (!X || !a || !b)) return RAMP_EINVAL;
  LIE_LAUNCH(se3_adj_kernel<true>, n, X, a, b, n);
}
}  // extern "C"

// ----------------------------------------------------- pops.transform fused
// reference: ramp/projective_ops.py:16-101 (iproj, Gij = Tj * Ti^-1 through
// lietorch inv/mul, act4 broadcast over the patch, proj with Z clamped at 0.1)
template <int P>
__global__ void __launch_bounds__(LIE_THREADS)
    transform_kernel(const float *__restrict__ poses, const float *__restrict__ patches,
                     const float *__restrict__ intr, const int64_t *__restrict__ ii,
                     const int64_t *__restrict__ jj, const int64_t *__restrict__ kk,
                     float *__restrict__ out, int E, int tonly, const int32_t *__restrict__ dyn) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (dyn) E = dyn[RAMP_DYN_E];                 // device-side size: the argument is the launch bound
  if (e >= E) return;
  const long i = ii[e], j = jj[e], k = kk[e];
  float Ti[7], Tj[7], Tinv[7], G[7];
#pragma unroll
  for (int c = 0; c < 7; c++) { Ti[c] = poses[7 * i + c]; Tj[c] = poses[7 * j + c]; }
  lt_inv(Ti, Tinv);
  lt_mul(Tj, Tinv, G);
  if (tonly) { G[3] = 0; G[4] = 0; G[5] = 0; G[6] = 1; }
  float t[3], q[4];
  lt_load(G, t, q);
  const float fxi = intr[4 * i + 0], fyi = intr[4 * i + 1], cxi = intr[4 * i + 2],
              cyi = intr[4 * i + 3];
  const float fxj = intr[4 * j + 0], fyj = intr[4 * j + 1], cxj = intr[4 * j + 2],
              cyj = intr[4 * j + 3];
  const float *pt = patches + (size_t)k * 3 * P * P;
  float *o = out + (size_t)e * 2 * P * P;
#pragma unroll
  for (int a = 0; a < P * P; a++) {
    float X0[4], X1[4];
    X0[0] = (pt[a] - cxi) / fxi;
    X0[1] = (pt[P * P + a] - cyi) / fyi;
    X0[2] = 1.0f;
    X0[3] = pt[2 * P * P + a];
    lt_act4_tq(t, q, X0, X1);
    const float Z = X1[2] < 0.1f ? 0.1f : X1[2];
    const float d = 1.0f / Z;
    o[a] = fxj * (d * X1[0]) + cxj;
    o[P * P + a] = fyj * (d * X1[1]) + cyj;
  }
}

// reference: ramp/fastba/ba_cuda.cu:379-429
template <int P>
__global__ void __launch_bounds__(LIE_THREADS)
    reproject_kernel(const float *__restrict__ poses, const float *__restrict__ patches,
                     const float *__restrict__ intr, const int64_t *__restrict__ ii,
                     const int64_t *__restrict__ jj, const int64_t *__restrict__ kk,
                     float *__restrict__ out, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const long i = ii[e], j = jj[e], k = kk[e];
  float pi[7], pj[7], tij[3], qij[4];
#pragma unroll
  for (int c = 0; c < 7; c++) { pi[c] = poses[7 * i + c]; pj[c] = poses[7 * j + c]; }
  fb_relSE3(pi, pi + 3, pj, pj + 3, tij, qij);
  const float *pt = patches + (size_t)k * 3 * P * P;
  float *o = out + (size_t)e * 2 * P * P;
#pragma unroll
  for (int a = 0; a < P * P; a++) {
    float Xi[4] = {(pt[a] - cx) / fx, (pt[P * P + a] - cy) / fy, 1.0f, pt[2 * P * P + a]};
    float Xj[4];
    fb_actSE3(tij, qij, Xi, Xj);
    o[a] = fx * (Xj[0] / Xj[2]) + cx;
    o[P * P + a] = fy * (Xj[1] / Xj[2]) + cy;
  }
}

// reference: ramp/projective_ops.py:103-105 + ramp/Ramp_vo.py:308-310
template <int P>
__device__ __forceinline__ void point_cloud_block(int bid, const float *__restrict__ poses, const float *__restrict__ patches,
                                                  const float *__restrict__ intr, const int64_t *__restrict__ ix,
                                                  float *__restrict__ out, int m, const int32_t *__restrict__ dyn, int M) {
  const int n = bid * LIE_THREADS + threadIdx.x;
  if (dyn) m = min(m, dyn[RAMP_DYN_N] * M);     // device-side size: the argument is the launch bound
  if (n >= m) return;
  const long f = ix[n];
  float T[7], Tinv[7], t[3], q[4];
#pragma unroll
  for (int c = 0; c < 7; c++) T[c] = poses[7 * f + c];
  lt_inv(T, Tinv);
  lt_load(Tinv, t, q);
  const float *pt = patches + (size_t)n * 3 * P * P;
  const int a = (P / 2) * P + (P / 2);
  float X0[4], X1[4];
  X0[0] = (pt[a] - intr[4 * f + 2]) / intr[4 * f + 0];
  X0[1] = (pt[P * P + a] - intr[4 * f + 3]) / intr[4 * f + 1];
  X0[2] = 1.0f;
  X0[3] = pt[2 * P * P + a];
  lt_act4_tq(t, q, X0, X1);
  out[3 * (size_t)n + 0] = X1[0] / X1[3];
  out[3 * (size_t)n + 1] = X1[1] / X1[3];
  out[3 * (size_t)n + 2] = X1[2] / X1[3];
}
template <int P>
__global__ void __launch_bounds__(LIE_THREADS)
    point_cloud_kernel(const float *__restrict__ poses, const float *__restrict__ patches,
                       const float *__restrict__ intr, const int64_t *__restrict__ ix,
                       float *__restrict__ out, int m, const int32_t *__restrict__ dyn, int M) {
  point_cloud_block<P>(blockIdx.x, poses, patches, intr, ix, out, m, dyn, M);
}

// Ramp_vo.motionmag both ways in one launch (ramp/Ramp_vo.py:227-243, pops.flow_mag :108-118):
// block b (0: i->j, 1: j->i) finds its (ii,jj) segment in the pair grouping by binary search on
// the sorted unique pair keys and reduces mean(beta*|x_full - x_0| + (1-beta)*|x_tonly - x_0|)
// over the segment's edges x 9 patch pixels in a fixed order.
template <int P>
__device__ __forceinline__ void mm_project(const float *t, const float *q, const float *pt, int a,
                                           const float *Ki, const float *Kj, float *xy) {
  float X0[4], X1[4];
  X0[0] = (pt[a] - Ki[2]) / Ki[0];
  X0[1] = (pt[P * P + a] - Ki[3]) / Ki[1];
  X0[2] = 1.0f;
  X0[3] = pt[2 * P * P + a];
  lt_act4_tq(t, q, X0, X1);
  const float Z = X1[2] < 0.1f ? 0.1f : X1[2];
  const float d = 1.0f / Z;
  xy[0] = Kj[0] * (d * X1[0]) + Kj[2];
  xy[1] = Kj[1] * (d * X1[1]) + Kj[3];
}

template <int P>
__device__ __forceinline__ void motionmag_block(int bid, const float *__restrict__ poses, const float *__restrict__ patches,
                                                const float *__restrict__ intr, const int64_t *__restrict__ ii,
                                                const int64_t *__restrict__ jj, const int64_t *__restrict__ kk,
                                                const int32_t *__restrict__ order, const int32_t *__restrict__ seg,
                                                const int64_t *__restrict__ ukeys, const int32_t *__restrict__ ngroups,
                                                long key0, long key1, float beta, float *__restrict__ out,
                                                const int32_t *__restrict__ dyn, int keyframe_index) {
  __shared__ float s_part[256];
  __shared__ int s_seg[2];
  if (dyn) {                                    // Ramp_vo.keyframe(): i = n - KEYFRAME_INDEX - 1, j = i + 2; keys are jj * W + ii
    const long n = dyn[RAMP_DYN_N], W = dyn[RAMP_DYN_W];
    const long i = n - keyframe_index - 1, j = n - keyframe_index + 1;
    key0 = j * W + i;
    key1 = i * W + j;
  }
  const long key = bid == 0 ? key0 : key1;
  if (threadIdx.x == 0) {
    int lo = 0, hi = *ngroups - 1, g = -1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const long v = ukeys[mid];
      if (v == key) { g = mid; break; }
      if (v < key) lo = mid + 1; else hi = mid - 1;
    }
    s_seg[0] = g < 0 ? 0 : seg[g];
    s_seg[1] = g < 0 ? 0 : seg[g + 1];
  }
  __syncthreads();
  const int s0 = s_seg[0], n = s_seg[1] - s0;
  float acc = 0.0f;
  for (int p = threadIdx.x; p < n; p += 256) {
    const int e = order[s0 + p];
    const long i = ii[e], j = jj[e], k = kk[e];
    float Ti[7], Tj[7], Tinv[7], G[7], G0[7], t[3], q[4], t0[3], q0[4], tt[3];
    for (int c = 0; c < 7; c++) { Ti[c] = poses[7 * i + c]; Tj[c] = poses[7 * j + c]; }
    lt_inv(Ti, Tinv);
    lt_mul(Tj, Tinv, G);       // i -> j
    lt_mul(Ti, Tinv, G0);      // i -> i (the reference evaluates this too)
    lt_load(G, t, q);
    lt_load(G0, t0, q0);
    float Gt[7] = {G[0], G[1], G[2], 0.f, 0.f, 0.f, 1.f}, qt[4];
    lt_load(Gt, tt, qt);
    const float *pt = patches + (size_t)k * 3 * P * P;
    const float *Ki = intr + 4 * i, *Kj = intr + 4 * j;
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < P * P; a++) {
      float c0[2], c1[2], c2[2];
      mm_project<P>(t0, q0, pt, a, Ki, Ki, c0);
      mm_project<P>(t, q, pt, a, Ki, Kj, c1);
      mm_project<P>(tt, qt, pt, a, Ki, Kj, c2);
      const float f1 = sqrtf((c1[0] - c0[0]) * (c1[0] - c0[0]) + (c1[1] - c0[1]) * (c1[1] - c0[1]));
      const float f2 = sqrtf((c2[0] - c0[0]) * (c2[0] - c0[0]) + (c2[1] - c0[1]) * (c2[1] - c0[1]));
      s += beta * f1 + (1 - beta) * f2;
    }
    acc += s;
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) s_part[threadIdx.x] += s_part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[bid] = n > 0 ? s_part[0] / (float)(n * P * P) : __int_as_float(0x7fc00000);
}
template <int P>
__global__ void __launch_bounds__(256)
    motionmag_kernel(const float *__restrict__ poses, const float *__restrict__ patches,
                     const float *__restrict__ intr, const int64_t *__restrict__ ii,
                     const int64_t *__restrict__ jj, const int64_t *__restrict__ kk,
                     const int32_t *__restrict__ order, const int32_t *__restrict__ seg,
                     const int64_t *__restrict__ ukeys, const int32_t *__restrict__ ngroups,
                     long key0, long key1, float beta, float *__restrict__ out, const int32_t *__restrict__ dyn,
                     int keyframe_index) {
  motionmag_block<P>(blockIdx.x, poses, patches, intr, ii, jj, kk, order, seg, ukeys, ngroups, key0, key1, beta, out, dyn,
                     keyframe_index);
}
// the motion test's two flow magnitudes (workgroups 0, 1: the longer ones first) and the point cloud in one launch: both
// read the poses / patches bundle adjustment just wrote, neither reads the other's output
template <int P>
__global__ void __launch_bounds__(256)
    motionmag_point_cloud_kernel(const float *__restrict__ poses, const float *__restrict__ patches,
                                 const float *__restrict__ intr, const int64_t *__restrict__ ii,
                                 const int64_t *__restrict__ jj, const int64_t *__restrict__ kk,
                                 const int32_t *__restrict__ order, const int32_t *__restrict__ seg,
                                 const int64_t *__restrict__ ukeys, const int32_t *__restrict__ ngroups, float beta,
                                 float *__restrict__ out2, int32_t *__restrict__ dyn, int keyframe_index,
                                 const int64_t *__restrict__ ix, float *__restrict__ points, int m_cap, int M,
                                 float *__restrict__ median) {
  if (median && blockIdx.x == gridDim.x - 1) {
    // the next frame's depth initialisation: lower median over the three newest frames' patches (they stay the three
    // newest whether or not the test below drops keyframe n - KEYFRAME_INDEX, KEYFRAME_INDEX >= 4)
    const int n = dyn[RAMP_DYN_N];
    if (n < 3) return;
    const float med = depth_median_block_t<256, 32>(patches + (size_t)(n - 3) * M * 3 * P * P, 3, M, P * P);
    if (threadIdx.x == 0) { *median = med; dyn[RAMP_DYN_MEDOK] = 1; }
    return;
  }
  if (blockIdx.x < 2)
    motionmag_block<P>(blockIdx.x, poses, patches, intr, ii, jj, kk, order, seg, ukeys, ngroups, 0L, 0L, beta, out2, dyn,
                       keyframe_index);
  else
    point_cloud_block<P>(blockIdx.x - 2, poses, patches, intr, ix, points, m_cap, dyn, M);
}

// DAMPED_LINEAR motion model (ramp/Ramp_vo.py:356-363): poses[n] = Exp(d * Log(P1 * P2^-1)) * P1
__global__ void motion_model_kernel(float *poses, int n, float damping) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float P1[7], P2[7], P2i[7], D[7], xi[6], E[7], Pn[7];
  for (int c = 0; c < 7; c++) { P1[c] = poses[7 * (n - 1) + c]; P2[c] = poses[7 * (n - 2) + c]; }
  lt_inv(P2, P2i);
  lt_mul(P1, P2i, D);
  lt_log(D, xi);
  for (int c = 0; c < 6; c++) xi[c] = damping * xi[c];
  lt_exp(xi, E);
  lt_mul(E, P1, Pn);
  for (int c = 0; c < 7; c++) poses[7 * n + c] = Pn[c];
}

// Per-frame bookkeeping of Ramp_vo.__call__ (ramp/Ramp_vo.py:345-363) as ONE launch: time stamp, patch index
// map, intrinsics row (copied from the previous frame when unchanged) and the motion-model pose of frame n.
// motion: 0 = leave poses[n]; 1 = DAMPED_LINEAR; 2 = copy poses[n-1]
__global__ void frame_begin_kernel(float *poses, int n, int motion, float damping, int64_t *tstamps, int64_t counter,
                                   int64_t *index_map, int64_t index_val, float *intrinsics, int copy_k) {
  const int t = threadIdx.x;
  if (t == 0) {
    if (tstamps) tstamps[n] = counter;
    if (index_map) index_map[n + 1] = index_val;
  }
  if (copy_k && t < 4) intrinsics[4 * n + t] = intrinsics[4 * (n - 1) + t];
  if (motion == 2 && t < 7) poses[7 * n + t] = poses[7 * (n - 1) + t];
  if (motion != 1 || t != 0) return;
  float P1[7], P2[7], P2i[7], D[7], xi[6], E[7], Pn[7];
  for (int c = 0; c < 7; c++) { P1[c] = poses[7 * (n - 1) + c]; P2[c] = poses[7 * (n - 2) + c]; }
  lt_inv(P2, P2i);
  lt_mul(P1, P2i, D);
  lt_log(D, xi);
  for (int c = 0; c < 6; c++) xi[c] = damping * xi[c];
  lt_exp(xi, E);
  lt_mul(E, P1, Pn);
  for (int c = 0; c < 7; c++) poses[7 * n + c] = Pn[c];
}

// ---- small multi-buffer copies (tracker bookkeeping: one launch instead of ~10-20 tiny ATen ops)
#define RAMP_MAXBUF 10
struct CopyDesc {
  const char *src[RAMP_MAXBUF];
  char *dst[RAMP_MAXBUF];
  long bytes[RAMP_MAXBUF];   // multiples of 4
  int n;
};
// dst[b][0:bytes[b]) = src[b][0:bytes[b]) for every buffer b (disjoint src/dst)
__global__ void __launch_bounds__(256) multi_copy_kernel(const CopyDesc d) {
  const int b = blockIdx.y;
  if (b >= d.n) return;
  const long n4 = d.bytes[b] / 4;
  const uint32_t *s = reinterpret_cast<const uint32_t *>(d.src[b]);
  uint32_t *o = reinterpret_cast<uint32_t *>(d.dst[b]);
  if ((((uintptr_t)s | (uintptr_t)o) & 15) == 0 && (n4 & 3) == 0) {
    const long n16 = n4 / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
      reinterpret_cast<uint4 *>(o)[i] = reinterpret_cast<const uint4 *>(s)[i];
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
      o[i] = s[i];
  }
}
// keyframe removal: rows k+1..n-1 move down by one in every buffer.  Plain buffers are
// [rows][row_bytes]; ring buffers hold row r at slot r % mod.  Rows are moved in ascending order by
// ONE workgroup per (buffer, column chunk), so a chunk is read before it is overwritten.
struct ShiftDesc {
  char *base[RAMP_MAXBUF];
  long row_bytes[RAMP_MAXBUF];
  int mod[RAMP_MAXBUF];      // 0: plain
  int n;
};
__global__ void __launch_bounds__(256) shift_rows_kernel(const ShiftDesc d, int k, int nrows) {
  const int b = blockIdx.y;
  if (b >= d.n) return;
  const long n4 = d.row_bytes[b] / 4;
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // each thread owns columns c, c + stride, ... and walks the rows itself: no cross-thread hazard
  for (long col = c; col < n4; col += (long)gridDim.x * blockDim.x) {
    for (int r = k; r < nrows - 1; r++) {
      const int sd = d.mod[b] ? r % d.mod[b] : r, ss = d.mod[b] ? (r + 1) % d.mod[b] : r + 1;
      reinterpret_cast<uint32_t *>(d.base[b] + (size_t)sd * d.row_bytes[b])[col] =
          reinterpret_cast<const uint32_t *>(d.base[b] + (size_t)ss * d.row_bytes[b])[col];
    }
  }
}

// ---- launchers with device-side sizes (csrc/track.hip): E / m are launch bounds, the live counts come from dyn
int ramp_i_transform_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ii,
                         const int64_t *jj, const int64_t *kk, float *out, int E_cap, const int32_t *dyn,
                         hipStream_t st) {
  hipLaunchKernelGGL(transform_kernel<3>, dim3(ramp_cdiv(E_cap, LIE_THREADS)), dim3(LIE_THREADS), 0, st, poses, patches,
                     intrinsics, ii, jj, kk, out, E_cap, 0, dyn);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_i_point_cloud_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ix,
                           float *out, int m_cap, const int32_t *dyn, int M, hipStream_t st) {
  hipLaunchKernelGGL(point_cloud_kernel<3>, dim3(ramp_cdiv(m_cap, LIE_THREADS)), dim3(LIE_THREADS), 0, st, poses,
                     patches, intrinsics, ix, out, m_cap, dyn, M);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_i_motionmag_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ii,
                         const int64_t *jj, const int64_t *kk, const int32_t *order, const int32_t *seg,
                         const int64_t *ukeys, const int32_t *ngroups, float beta, float *out2, const int32_t *dyn,
                         int keyframe_index, hipStream_t st) {
  hipLaunchKernelGGL(motionmag_kernel<3>, dim3(2), dim3(256), 0, st, poses, patches, intrinsics, ii, jj, kk, order, seg,
                     ukeys, ngroups, 0L, 0L, beta, out2, dyn, keyframe_index);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_i_motionmag_point_cloud_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ii,
                                     const int64_t *jj, const int64_t *kk, const int32_t *order, const int32_t *seg,
                                     const int64_t *ukeys, const int32_t *ngroups, float beta, float *out2,
                                     int32_t *dyn, int keyframe_index, const int64_t *ix, float *points, int m_cap,
                                     int M, float *median, hipStream_t st) {
  if (median && 3 * M * 9 > 256 * 32) median = nullptr;
  hipLaunchKernelGGL(motionmag_point_cloud_kernel<3>, dim3(2 + ramp_cdiv(m_cap, LIE_THREADS) + (median ? 1 : 0)), dim3(256), 0,
                     st, poses, patches, intrinsics, ii, jj, kk, order, seg, ukeys, ngroups, beta, out2, dyn, keyframe_index,
                     ix, points, m_cap, M, median);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

extern "C" {
int ramp_multi_copy(const void *const *src_host, void *const *dst_host, const long *bytes_host, int n,
                    void *stream) {
  if (n < 0 || n > RAMP_MAXBUF) return RAMP_EINVAL;
  if (n == 0) return RAMP_OK;
  CopyDesc d;
  long mx = 0;
  for (int i = 0; i < n; i++) {
    if (!src_host[i] || !dst_host[i] || bytes_host[i] < 0 || (bytes_host[i] & 3)) return RAMP_EINVAL;
    d.src[i] = (const char *)src_host[i]; d.dst[i] = (char *)dst_host[i]; d.bytes[i] = bytes_host[i];
    if (bytes_host[i] > mx) mx = bytes_host[i];
  }
  d.n = n;
  int bx = (int)((mx / 16 + 255) / 256);
  bx = bx < 1 ? 1 : (bx > 512 ? 512 : bx);
  hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, d);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_shift_rows(void *const *base_host, const long *row_bytes_host, const int *mod_host, int n, int k,
                    int nrows, void *stream) {
  if (n < 0 || n > RAMP_MAXBUF || k < 0) return RAMP_EINVAL;
  if (n == 0 || k >= nrows - 1) return RAMP_OK;
  ShiftDesc d;
  long mx = 0;
  for (int i = 0; i < n; i++) {
    if (!base_host[i] || row_bytes_host[i] <= 0 || (row_bytes_host[i] & 3) || mod_host[i] < 0) return RAMP_EINVAL;
    d.base[i] = (char *)base_host[i]; d.row_bytes[i] = row_bytes_host[i]; d.mod[i] = mod_host[i];
    if (row_bytes_host[i] > mx) mx = row_bytes_host[i];
  }
  d.n = n;
  int bx = (int)((mx / 4 + 255) / 256);
  bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
  hipLaunchKernelGGL(shift_rows_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, d, k, nrows);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_motionmag(const float *poses, const float *patches, const float *intrinsics,
                   const int64_t *ii, const int64_t *jj, const int64_t *kk, const int32_t *order,
                   const int32_t *seg, const int64_t *ukeys, const int32_t *ngroups, int64_t key_ij,
                   int64_t key_ji, float beta, float *out2, int P, void *stream) {
  if (!poses || !patches || !intrinsics || !ii || !jj || !kk || !order || !seg || !ukeys || !ngroups ||
      !out2)
    return RAMP_EINVAL;
  if (P != 3) return RAMP_EUNSUPPORTED;
  hipLaunchKernelGGL(motionmag_kernel<3>, dim3(2), dim3(256), 0, (hipStream_t)stream, poses, patches,
                     intrinsics, ii, jj, kk, order, seg, ukeys, ngroups, (long)key_ij, (long)key_ji, beta,
                     out2, (const int32_t *)nullptr, 0);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_frame_begin(float *poses, int n, int motion, float damping, int64_t *tstamps, int64_t counter,
                     int64_t *index_map, int64_t index_val, float *intrinsics, int copy_k, void *stream) {
  if (!poses || n < 0 || (motion == 1 && n < 2) || ((motion == 2 || copy_k) && n < 1) || (copy_k && !intrinsics))
    return RAMP_EINVAL;
  hipLaunchKernelGGL(frame_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, poses, n, motion, damping,
                     tstamps, counter, index_map, index_val, intrinsics, copy_k);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_motion_model(float *poses, int n, float damping, void *stream) {
  if (!poses || n < 2) return RAMP_EINVAL;
  hipLaunchKernelGGL(motion_model_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, poses, n, damping);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_transform(const float *poses, const float *patches, const float *intrinsics,
                   const int64_t *ii, const int64_t *jj, const int64_t *kk, float *out, int E,
                   int P, int tonly, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!poses || !patches || !intrinsics || !ii || !jj || !kk || !out) return RAMP_EINVAL;
  if (P != 3) return RAMP_EUNSUPPORTED;
  hipLaunchKernelGGL(transform_kernel<3>, dim3(ramp_cdiv(E, LIE_THREADS)), dim3(LIE_THREADS), 0,
                     (hipStream_t)stream, poses, patches, intrinsics, ii, jj, kk, out, E, tonly, (const int32_t *)nullptr);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_reproject(const float *poses, const float *patches, const float *intrinsics,
                   const int64_t *ii, const int64_t *jj, const int64_t *kk, float *out, int E,
                   int P, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!poses || !patches || !intrinsics || !ii || !jj || !kk || !out) return RAMP_EINVAL;
  if (P != 3) return RAMP_EUNSUPPORTED;
  hipLaunchKernelGGL(reproject_kernel<3>, dim3(ramp_cdiv(E, LIE_THREADS)), dim3(LIE_THREADS), 0,
                     (hipStream_t)stream, poses, patches, intrinsics, ii, jj, kk, out, E);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_point_cloud(const float *poses, const float *patches, const float *intrinsics,
                     const int64_t *ix, float *out, int m, int P, void *stream) {
  if (m < 0) return RAMP_EINVAL;
  if (m == 0) return RAMP_OK;
  if (!poses || !patches || !intrinsics || !ix || !out) return RAMP_EINVAL;
  if (P != 3) return RAMP_EUNSUPPORTED;
  hipLaunchKernelGGL(point_cloud_kernel<3>, dim3(ramp_cdiv(m, LIE_THREADS)), dim3(LIE_THREADS), 0,
                     (hipStream_t)stream, poses, patches, intrinsics, ix, out, m, (const int32_t *)nullptr, 0);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
const char *ramp_version(void) { return "rampvo-mi355x libramp_hip 0.1 (gfx950)"; }
}  // extern "C"
