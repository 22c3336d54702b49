/*
 * ramp_oracle.c -- CPU restatement of the RAMP-VO hot-path native operators.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * kernels in rampvo_amd/csrc and the "port" CPU baseline timed by bench.py.
 * Nothing under rampvo_amd/ may import, link or call it.
 *
 * Every function restates the algorithm of a reference source (paths relative
 * to the upstream repository, file:line cited per function).  Plain C,
 * float32 arithmetic, no Eigen / torch.  The reference's native code cannot be
 * compiled in the build image (Eigen 3.4.0 is an un-vendored download, nvcc is
 * absent), so the pin for these functions is:
 *   - lietorch SE3 ops: the identities of ramp/lietorch/run_tests.py:16-52
 *   - BA: cross-check against the reference's own python ramp/ba.py::BA
 *   - corr/patchify/transform: the reference's python call sites executed in the
 *     build container over these functions (oracle/make_golden.py)
 * altcorr / fastba have no golden vectors upstream: "parity unpinned" beyond
 * those cross-checks.
 *
 * Summation conventions (documented because float results depend on them):
 *   - channel dot products are a k-ordered fmaf chain (nvcc contracts
 *     `s += a*b` of correlation_kernel.cu:130 into FMA)
 *   - everything else is compiled with -ffp-contract=off and follows the
 *     reference's expression order; reductions the reference performs with
 *     atomics (order undefined upstream) are done here in edge order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* Diagnostic build (oracle/Makefile target libramp_oracle_f64.so): the same
 * source with every `float` promoted to double, used by tests to measure the
 * fp32 rounding envelope of ill-conditioned problems.  Never a parity target. */
#ifdef ORC_F64
#define float double
#define __builtin_fmaf __builtin_fma
/* (the math library's double forms too: the fp64 pin of the bundle-adjustment algebra against ramp/ba.py run in float64 --
 * tests/golden/ba_f64_pin.npz -- compares to 1e-8) */
#define sqrtf sqrt
#define sinf sin
#define cosf cos
#define atanf atan
#define atan2f atan2
#define acosf acos
#define fabsf fabs
#define fmaxf fmax
#define fminf fmin
#define floorf floor
#define expf exp
#endif

/* float -> int with the saturating semantics of the GPU conversion the
 * reference relies on (static_cast<int>(floor(x)) in device code). */
static inline int f2i_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483520.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

/* ------------------------------------------------------------------------- */
/* altcorr: patchify (ramp/altcorr/correlation_kernel.cu:16-47, 288-307)      */
/* ------------------------------------------------------------------------- */

/* raw (2R+2)^2 window gather, zero outside the image.
 * net [n][C][H][W], coords [n][M][2] (x,y), out [n][M][C][D][D] */
ORC_API void orc_patchify_raw(const float *net, const float *coords, float *out,
                              int n, int C, int H, int W, int M, int R) {
  const int D = 2 * R + 2;
  memset(out, 0, sizeof(float) * (size_t)n * M * C * D * D);
  for (int b = 0; b < n; b++)
    for (int m = 0; m < M; m++) {
      const float x = coords[((size_t)b * M + m) * 2 + 0];
      const float y = coords[((size_t)b * M + m) * 2 + 1];
      const int fy = f2i_sat(floorf(y)), fx = f2i_sat(floorf(x));
      for (int ii = 0; ii < D; ii++)
        for (int jj = 0; jj < D; jj++) {
          const long i = (long)fy + (ii - R);
          const long j = (long)fx + (jj - R);
          if (i < 0 || i >= H || j < 0 || j >= W) continue;
          for (int k = 0; k < C; k++)
            out[((((size_t)b * M + m) * C + k) * D + ii) * D + jj] =
                net[(((size_t)b * C + k) * H + i) * W + j];
        }
    }
}

/* patchify + bilinear blend (ramp/altcorr/correlation.py:51-68).
 * out [n][M][C][d][d], d = 2R+1 */
ORC_API void orc_patchify(const float *net, const float *coords, float *out,
                          int n, int C, int H, int W, int M, int R) {
  const int D = 2 * R + 2, d = 2 * R + 1;
  float *raw = (float *)malloc(sizeof(float) * (size_t)n * M * C * D * D);
  orc_patchify_raw(net, coords, raw, n, C, H, W, M, R);
  for (int b = 0; b < n; b++)
    for (int m = 0; m < M; m++) {
      const float x = coords[((size_t)b * M + m) * 2 + 0];
      const float y = coords[((size_t)b * M + m) * 2 + 1];
      const float dx = x - floorf(x), dy = y - floorf(y);
      const float w00 = (1 - dy) * (1 - dx), w01 = (1 - dy) * dx;
      const float w10 = dy * (1 - dx), w11 = dy * dx;
      for (int k = 0; k < C; k++) {
        const float *p = raw + (((size_t)b * M + m) * C + k) * D * D;
        float *o = out + (((size_t)b * M + m) * C + k) * d * d;
        for (int a = 0; a < d; a++)
          for (int c = 0; c < d; c++) {
            float s = w00 * p[a * D + c];
            s = s + w01 * p[a * D + c + 1];
            s = s + w10 * p[(a + 1) * D + c];
            s = s + w11 * p[(a + 1) * D + c + 1];
            o[a * d + c] = s;
          }
      }
    }
  free(raw);
}

/* ------------------------------------------------------------------------- */
/* altcorr: corr (correlation_kernel.cu:82-136 kernel, 193-233 host)          */
/* ------------------------------------------------------------------------- */

/* fmap1 [N1][C][P][P] (patch features), fmap2 [N2][C][H2][W2],
 * coords [E][2][P][P], us[E] -> fmap1 index, vs[E] -> fmap2 index.
 * out [E][d(x-off)][d(y-off)][P][P], d = 2R+1  (the permute(0,1,3,2,4,5) of
 * correlation_kernel.cu:232 is applied). */
ORC_API void orc_corr(const float *fmap1, const float *fmap2, const float *coords,
                      const int64_t *us, const int64_t *vs, float *out, int E,
                      int C, int P, int H2, int W2, int R) {
  const int D = 2 * R + 2, d = 2 * R + 1;
  /* edges are independent: OpenMP over edges changes no result (only the CPU-baseline time) */
#pragma omp parallel for schedule(dynamic, 16)
  for (int e = 0; e < E; e++) {
    float raw[D * D];
    const float *f1 = fmap1 + (size_t)us[e] * C * P * P;
    const float *f2 = fmap2 + (size_t)vs[e] * C * H2 * W2;
    for (int i0 = 0; i0 < P; i0++)
      for (int j0 = 0; j0 < P; j0++) {
        const float x = coords[(((size_t)e * 2 + 0) * P + i0) * P + j0];
        const float y = coords[(((size_t)e * 2 + 1) * P + i0) * P + j0];
        const int fy = f2i_sat(floorf(y)), fx = f2i_sat(floorf(x));
        for (int ii = 0; ii < D; ii++)
          for (int jj = 0; jj < D; jj++) {
            const long i1 = (long)fy + (ii - R);
            const long j1 = (long)fx + (jj - R);
            float s = 0.0f;
            if (i1 >= 0 && i1 < H2 && j1 >= 0 && j1 < W2) {
              for (int c = 0; c < C; c++)
                s = __builtin_fmaf(f1[((size_t)c * P + i0) * P + j0],
                                   f2[((size_t)c * H2 + i1) * W2 + j1], s);
            }
            raw[ii * D + jj] = s;
          }
        const float dx = x - floorf(x), dy = y - floorf(y);
        const float w00 = (1 - dx) * (1 - dy), w01 = dx * (1 - dy);
        const float w10 = (1 - dx) * dy, w11 = dx * dy;
        for (int a = 0; a < d; a++)   /* y offset */
          for (int b = 0; b < d; b++) { /* x offset */
            float s = w00 * raw[a * D + b];
            s = s + w01 * raw[a * D + b + 1];
            s = s + w10 * raw[(a + 1) * D + b];
            s = s + w11 * raw[(a + 1) * D + b + 1];
            out[((((size_t)e * d + b) * d + a) * P + i0) * P + j0] = s;
          }
      }
  }
}

/* ------------------------------------------------------------------------- */
/* lietorch SE3 / SO3 float32 forward ops                                     */
/* (ramp/lietorch/include/so3.h:31-201, se3.h:30-142, common.h:7)             */
/* data layout: [tx ty tz qx qy qz qw]                                        */
/* ------------------------------------------------------------------------- */
#define LT_EPS 1e-6f
#define LT_PI 3.14159265358979323846f

static inline void q_normalize(const float *q, float *o) { /* so3.h:35-41 */
  const float n = sqrtf((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
  o[0] = q[0] / n; o[1] = q[1] / n; o[2] = q[2] / n; o[3] = q[3] / n;
}
static inline void q_mul_raw(const float *a, const float *b, float *o) {
  /* Eigen quaternion product, coefficients (x,y,z,w) */
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static inline void q_rot(const float *q, const float *p, float *o) { /* so3.h:55-60 */
  float uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2],
                 q[0] * p[1] - q[1] * p[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = p[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = p[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = p[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static inline void q_to_R(const float *q, float R[9]) { /* Eigen toRotationMatrix */
  const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
static inline void so3_exp(const float *phi, float *q) { /* so3.h:153-170 */
  const float theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta = sqrtf(theta2);
  float imag, real;
  if (theta < LT_EPS) {
    const float theta4 = theta2 * theta2;
    imag = 0.5f - (1.0f / 48.0f) * theta2 + (1.0f / 3840.0f) * theta4;
    real = 1.0f - (1.0f / 8.0f) * theta2 + (1.0f / 384.0f) * theta4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  float r[4] = {imag * phi[0], imag * phi[1], imag * phi[2], real};
  q_normalize(r, q);
}
static inline void so3_log(const float *q, float *phi) { /* so3.h:115-151 */
  const float sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const float w = q[3];
  float k;
  if (sn < LT_EPS * LT_EPS) {
    const float sw = w * w;
    k = 2.0f / w - (2.0f / 3.0f) * sn / (w * sw);
  } else {
    const float n = sqrtf(sn);
    if (fabsf(w) < LT_EPS) k = (w > 0 ? LT_PI : -LT_PI) / n;
    else k = 2.0f * atanf(n / w) / n;
  }
  phi[0] = k * q[0]; phi[1] = k * q[1]; phi[2] = k * q[2];
}
static inline void hat3(const float *p, float M[9]) {
  M[0] = 0; M[1] = -p[2]; M[2] = p[1];
  M[3] = p[2]; M[4] = 0; M[5] = -p[0];
  M[6] = -p[1]; M[7] = p[0]; M[8] = 0;
}
static inline void mat3mul(const float *A, const float *B, float *C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
static inline void mat3vec(const float *A, const float *v, float *o) {
  for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
static void so3_left_jacobian(const float *phi, float J[9]) { /* so3.h:172-189 */
  float Phi[9], Phi2[9];
  hat3(phi, Phi); mat3mul(Phi, Phi, Phi2);
  const float t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float t = sqrtf(t2);
  const float c1 = (t < LT_EPS) ? 0.5f - (1.0f / 24.0f) * t2 : (1.0f - cosf(t)) / t2;
  const float c2 = (t < LT_EPS) ? (1.0f / 6.0f) - (1.0f / 120.0f) * t2 : (t - sinf(t)) / (t2 * t);
  for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + c1 * Phi[i] + c2 * Phi2[i];
}
static void so3_left_jacobian_inv(const float *phi, float J[9]) { /* so3.h:191-208 */
  float Phi[9], Phi2[9];
  hat3(phi, Phi); mat3mul(Phi, Phi, Phi2);
  const float t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float t = sqrtf(t2), ht = 0.5f * t;
  const float c2 = (t < LT_EPS) ? (1.0f / 12.0f)
                                : (1.0f - t * cosf(ht) / (2.0f * sinf(ht))) / (t * t);
  for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + (-0.5f) * Phi[i] + c2 * Phi2[i];
}

static inline void se3_load(const float *d, float *t, float *q) { /* se3.h:34, so3.h:39 */
  t[0] = d[0]; t[1] = d[1]; t[2] = d[2];
  q_normalize(d + 3, q);
}
static inline void se3_inv1(const float *X, float *Y) { /* se3.h:36-38 */
  float t[3], q[4], qi[4], qin[4], r[3];
  se3_load(X, t, q);
  qi[0] = -q[0]; qi[1] = -q[1]; qi[2] = -q[2]; qi[3] = q[3];
  q_normalize(qi, qin);
  q_rot(qin, t, r);
  Y[0] = -r[0]; Y[1] = -r[1]; Y[2] = -r[2];
  /* SE3(so3.inv(), ..) : so3.inv() already normalised once, copy ctor keeps it */
  Y[3] = qin[0]; Y[4] = qin[1]; Y[5] = qin[2]; Y[6] = qin[3];
}
static inline void se3_mul1(const float *X, const float *Y, float *Z) { /* se3.h:45-47 */
  float tx[3], qx[4], ty[3], qy[4], qz[4], qzn[4], r[3];
  se3_load(X, tx, qx); se3_load(Y, ty, qy);
  q_mul_raw(qx, qy, qz); q_normalize(qz, qzn);
  q_rot(qx, ty, r);
  Z[0] = tx[0] + r[0]; Z[1] = tx[1] + r[1]; Z[2] = tx[2] + r[2];
  Z[3] = qzn[0]; Z[4] = qzn[1]; Z[5] = qzn[2]; Z[6] = qzn[3];
}
static inline void se3_act41(const float *X, const float *p, float *o) { /* se3.h:53-56 */
  float t[3], q[4], r[3];
  se3_load(X, t, q);
  q_rot(q, p, r);
  o[0] = r[0] + t[0] * p[3]; o[1] = r[1] + t[1] * p[3]; o[2] = r[2] + t[2] * p[3];
  o[3] = p[3];
}
static inline void se3_exp1(const float *xi, float *X) { /* se3.h:134-142 */
  float q[4], J[9], t[3];
  so3_exp(xi + 3, q);
  so3_left_jacobian(xi + 3, J);
  mat3vec(J, xi, t);
  X[0] = t[0]; X[1] = t[1]; X[2] = t[2];
  X[3] = q[0]; X[4] = q[1]; X[5] = q[2]; X[6] = q[3];
}
static inline void se3_log1(const float *X, float *xi) { /* se3.h:124-132 */
  float t[3], q[4], phi[3], Vi[9], tau[3];
  se3_load(X, t, q);
  so3_log(q, phi);
  so3_left_jacobian_inv(phi, Vi);
  mat3vec(Vi, t, tau);
  xi[0] = tau[0]; xi[1] = tau[1]; xi[2] = tau[2];
  xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
}
static void se3_Adj(const float *X, float Ad[36]) { /* se3.h:58-68 */
  float t[3], q[4], R[9], T[9], TR[9];
  se3_load(X, t, q);
  q_to_R(q, R); hat3(t, T); mat3mul(T, R, TR);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ad[i * 6 + j] = R[i * 3 + j];
      Ad[i * 6 + 3 + j] = TR[i * 3 + j];
      Ad[(i + 3) * 6 + j] = 0;
      Ad[(i + 3) * 6 + 3 + j] = R[i * 3 + j];
    }
}

/* batched entry points, mirror lietorch_backends.{expm,logm,inv,mul,act4,adj,adjT}
 * (ramp/lietorch/src/lietorch.cpp:286-316) for group_id 3 (SE3), float32 */
ORC_API void orc_se3_exp(const float *a, float *X, int n) { for (int i = 0; i < n; i++) se3_exp1(a + 6 * i, X + 7 * i); }
ORC_API void orc_se3_log(const float *X, float *a, int n) { for (int i = 0; i < n; i++) se3_log1(X + 7 * i, a + 6 * i); }
ORC_API void orc_se3_inv(const float *X, float *Y, int n) { for (int i = 0; i < n; i++) se3_inv1(X + 7 * i, Y + 7 * i); }
ORC_API void orc_se3_mul(const float *X, const float *Y, float *Z, int n) { for (int i = 0; i < n; i++) se3_mul1(X + 7 * i, Y + 7 * i, Z + 7 * i); }
ORC_API void orc_se3_act4(const float *X, const float *p, float *q, int n) { for (int i = 0; i < n; i++) se3_act41(X + 7 * i, p + 4 * i, q + 4 * i); }
ORC_API void orc_se3_adj(const float *X, const float *a, float *b, int n) {
  for (int i = 0; i < n; i++) {
    float Ad[36]; se3_Adj(X + 7 * i, Ad);
    for (int r = 0; r < 6; r++) { float s = 0; for (int c = 0; c < 6; c++) s += Ad[r * 6 + c] * a[6 * i + c]; b[6 * i + r] = s; }
  }
}
ORC_API void orc_se3_adjT(const float *X, const float *a, float *b, int n) {
  for (int i = 0; i < n; i++) {
    float Ad[36]; se3_Adj(X + 7 * i, Ad);
    for (int r = 0; r < 6; r++) { float s = 0; for (int c = 0; c < 6; c++) s += Ad[c * 6 + r] * a[6 * i + c]; b[6 * i + r] = s; }
  }
}

/* ------------------------------------------------------------------------- */
/* projective transform (ramp/projective_ops.py:16-101, jacobian=False path)  */
/* poses [Np][7], patches [Nk][3][P][P], intr [Np][4]; out [E][2][P][P]       */
/* (already in the permute(0,1,4,2,3) layout Ramp_vo.reproject returns,        */
/*  ramp/Ramp_vo.py:184-192).  tonly: projective_ops.py:59-60.                 */
/* ------------------------------------------------------------------------- */
ORC_API void orc_transform(const float *poses, const float *patches, const float *intr,
                           const int64_t *ii, const int64_t *jj, const int64_t *kk,
                           float *out, int E, int P, int tonly) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; e++) {
    const float *Ki = intr + 4 * ii[e], *Kj = intr + 4 * jj[e];
    float Ti_inv[7], G[7];
    se3_inv1(poses + 7 * ii[e], Ti_inv);
    se3_mul1(poses + 7 * jj[e], Ti_inv, G);
    if (tonly) { G[3] = 0; G[4] = 0; G[5] = 0; G[6] = 1; }
    const float *pt = patches + (size_t)kk[e] * 3 * P * P;
    for (int a = 0; a < P * P; a++) {
      float X0[4], X1[4];
      X0[0] = (pt[a] - Ki[2]) / Ki[0];
      X0[1] = (pt[P * P + a] - Ki[3]) / Ki[1];
      X0[2] = 1.0f;
      X0[3] = pt[2 * P * P + a];
      se3_act41(G, X0, X1);
      const float Z = X1[2] < 0.1f ? 0.1f : X1[2]; /* clamp(min=0.1) */
      const float d = 1.0f / Z;
      out[((size_t)e * 2 + 0) * P * P + a] = Kj[0] * (d * X1[0]) + Kj[2];
      out[((size_t)e * 2 + 1) * P * P + a] = Kj[1] * (d * X1[1]) + Kj[3];
    }
  }
}

/* ------------------------------------------------------------------------- */
/* fastba device helpers (ramp/fastba/ba_cuda.cu:36-174)                      */
/* ------------------------------------------------------------------------- */
static inline void actSO3(const float *q, const float *X, float *Y) {
  float uv[3];
  uv[0] = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  uv[1] = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  uv[2] = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  Y[1] = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  Y[2] = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static inline void actSE3(const float *t, const float *q, const float *X, float *Y) {
  actSO3(q, X, Y);
  Y[3] = X[3];
  Y[0] += X[3] * t[0]; Y[1] += X[3] * t[1]; Y[2] += X[3] * t[2];
}
static inline void adjSE3(const float *t, const float *q, const float *X, float *Y) {
  float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  actSO3(qinv, &X[0], &Y[0]);
  actSO3(qinv, &X[3], &Y[3]);
  float u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  actSO3(qinv, u, v);
  Y[3] += v[0]; Y[4] += v[1]; Y[5] += v[2];
}
static inline void relSE3(const float *ti, const float *qi, const float *tj, const float *qj,
                          float *tij, float *qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  actSO3(qij, ti, tij);
  tij[0] = tj[0] - tij[0]; tij[1] = tj[1] - tij[1]; tij[2] = tj[2] - tij[2];
}
static inline void expSO3(const float *phi, float *q) {
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta_p4 = theta_sq * theta_sq;
  const float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
}
static inline void crossInplace(const float *a, float *b) {
  float x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
}
static inline void expSE3(const float *xi, float *t, float *q) {
  expSO3(xi + 3, q);
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi[5]};
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta = sqrtf(theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4f) {
    const float a = (1 - cosf(theta)) / theta_sq;
    crossInplace(phi, tau);
    t[0] += a * tau[0]; t[1] += a * tau[1]; t[2] += a * tau[2];
    const float b = (theta - sinf(theta)) / (theta * theta_sq);
    crossInplace(phi, tau);
    t[0] += b * tau[0]; t[1] += b * tau[1]; t[2] += b * tau[2];
  }
}
static inline void retrSE3(const float *xi, const float *t, const float *q, float *t1, float *q1) {
  float dt[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1};
  expSE3(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  actSO3(dq, t, t1);
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
}

/* fastba.reproject (ba_cuda.cu:379-429, 585-617): frame-0 intrinsics, no Z clamp.
 * out [E][2][P][P] */
ORC_API void orc_reproject(const float *poses, const float *patches, const float *intr,
                           const int64_t *ii, const int64_t *jj, const int64_t *kk,
                           float *out, int E, int P) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  for (int e = 0; e < E; e++) {
    const float *pi = poses + 7 * ii[e], *pj = poses + 7 * jj[e];
    float tij[3], qij[4];
    relSE3(pi, pi + 3, pj, pj + 3, tij, qij);
    const float *pt = patches + (size_t)kk[e] * 3 * P * P;
    for (int a = 0; a < P * P; a++) {
      float Xi[4] = {(pt[a] - cx) / fx, (pt[P * P + a] - cy) / fy, 1.0f, pt[2 * P * P + a]}, Xj[4];
      actSE3(tij, qij, Xi, Xj);
      out[((size_t)e * 2 + 0) * P * P + a] = fx * (Xj[0] / Xj[2]) + cx;
      out[((size_t)e * 2 + 1) * P * P + a] = fy * (Xj[1] / Xj[2]) + cy;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* fastba.neighbors (ramp/fastba/ba.cpp:59-97): for every edge, the previous / */
/* next edge of the same key ii ordered by jj (stable), -1 at the ends.       */
/* ------------------------------------------------------------------------- */
typedef struct { int64_t key, sub; int idx; } nb_t;
static int nb_cmp(const void *a, const void *b) {
  const nb_t *x = (const nb_t *)a, *y = (const nb_t *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->sub != y->sub) return x->sub < y->sub ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
ORC_API void orc_neighbors(const int64_t *ii, const int64_t *jj, int64_t *ix, int64_t *jx, int E) {
  nb_t *v = (nb_t *)malloc(sizeof(nb_t) * (size_t)(E > 0 ? E : 1));
  for (int i = 0; i < E; i++) { v[i].key = ii[i]; v[i].sub = jj[i]; v[i].idx = i; }
  qsort(v, E, sizeof(nb_t), nb_cmp); /* (key, sub, idx) total order == stable sort by sub within key */
  for (int p = 0; p < E; p++) {
    const int e = v[p].idx;
    ix[e] = (p > 0 && v[p - 1].key == v[p].key) ? v[p - 1].idx : -1;
    jx[e] = (p + 1 < E && v[p + 1].key == v[p].key) ? v[p + 1].idx : -1;
  }
  free(v);
}

/* ------------------------------------------------------------------------- */
/* fastba.BA forward (ba_cuda.cu:433-582 host loop; 232-376 residual/Hessian  */
/* kernel; 178-229 retraction kernels), eff_impl=False path.                  */
/* poses [Np][7] and patches [Nk][3][P][P] are updated IN PLACE.              */
/* returns 0, or 1 if the Cholesky factorisation met a non-positive pivot     */
/* (the reference ignores linalg_cholesky_ex's info, ba_cuda.cu:561).         */
/* ------------------------------------------------------------------------- */
static int i64cmp(const void *a, const void *b) {
  const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : (x > y);
}
/* One factor's terms of reprojection_residuals_and_hessian (ba_cuda.cu:232-376): relative pose, projection of the patch
 * centre, residual, validity mask and the three Jacobians.  Ji carries the kernel's sign convention (Ji = adjSE3(Jj); the
 * derivative with respect to pose i is -Ji: ba_cuda.cu:337-347 folds the sign into the accumulation).  Shared by the
 * Gauss-Newton loop below and by orc_ba_edge_terms (the fp64 pin against ramp/ba.py + ramp/projective_ops.py). */
typedef struct { float Ji[2][6], Jj[2][6], Jz[2], r[2], x1, y1, mask; } ba_edge_t;
static void ba_edge_terms(const float *poses, const float *patches, const float *intr, int ix, int jx, int64_t kxn, int P,
                          const float *target2, ba_edge_t *o) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int PP = P * P, c11 = 1 * P + 1;        /* the kernel reads patches[kx][c][1][1] (ba_cuda.cu:282-285) */
  const float *pi = poses + 7 * ix, *pj = poses + 7 * jx;
  float Xi[4], Xj[4];
  Xi[0] = (patches[((size_t)kxn * 3 + 0) * PP + c11] - cx) / fx;
  Xi[1] = (patches[((size_t)kxn * 3 + 1) * PP + c11] - cy) / fy;
  Xi[2] = 1.0f;
  Xi[3] = patches[((size_t)kxn * 3 + 2) * PP + c11];
  float tij[3], qij[4];
  relSE3(pi, pi + 3, pj, pj + 3, tij, qij);
  actSE3(tij, qij, Xi, Xj);
  const float X = Xj[0], Y = Xj[1], Z = Xj[2], W = Xj[3];
  const float d = (Z >= 0.2f) ? 1.0f / Z : 0.0f;
  const float d2 = d * d;
  const float x1 = fx * (X / Z) + cx;
  const float y1 = fy * (Y / Z) + cy;
  const float rx = target2[0] - x1;
  const float ry = target2[1] - y1;
  const int in_bounds = (sqrtf(rx * rx + ry * ry) < 128) && (Z > 0.2f) && (x1 > -64) &&
                        (y1 > -64) && (x1 < 2 * cx + 64) && (y1 < 2 * cy + 64);
  o->mask = in_bounds ? 1.0f : 0.0f;
  o->x1 = x1; o->y1 = y1;
  for (int row = 0; row < 2; row++) {
    float *Jj = o->Jj[row];
    if (row == 0) {
      o->r[0] = target2[0] - x1;
      o->Jz[0] = fx * (tij[0] * d - tij[2] * (X * d2));
      Jj[0] = fx * W * d; Jj[1] = 0; Jj[2] = fx * -X * W * d2;
      Jj[3] = fx * -X * Y * d2; Jj[4] = fx * (1 + X * X * d2); Jj[5] = fx * -Y * d;
    } else {
      o->r[1] = target2[1] - y1;
      o->Jz[1] = fy * (tij[1] * d - tij[2] * (Y * d2));
      Jj[0] = 0; Jj[1] = fy * W * d; Jj[2] = fy * -Y * W * d2;
      Jj[3] = fy * (-1 - Y * Y * d2); Jj[4] = fy * (X * Y * d2); Jj[5] = fy * X * d;
    }
    adjSE3(tij, qij, Jj, o->Ji[row]);
  }
}

/* (tests) the per-factor terms above for E factors: Ji, Jj [E][2][6], Jz [E][2], xy [E][2] (projected centre), mask [E] */
ORC_API void orc_ba_edge_terms(const float *poses, const float *patches, const float *intr, const int64_t *ii,
                               const int64_t *jj, const int64_t *kk, int E, int P, float *Ji, float *Jj, float *Jz,
                               float *xy, float *mask) {
  const float zero2[2] = {0, 0};
  for (int n = 0; n < E; n++) {
    ba_edge_t et;
    ba_edge_terms(poses, patches, intr, (int)ii[n], (int)jj[n], kk[n], P, zero2, &et);
    for (int r = 0; r < 2; r++) {
      for (int c = 0; c < 6; c++) { Ji[(n * 2 + r) * 6 + c] = et.Ji[r][c]; Jj[(n * 2 + r) * 6 + c] = et.Jj[r][c]; }
      Jz[n * 2 + r] = et.Jz[r];
    }
    xy[2 * n] = et.x1; xy[2 * n + 1] = et.y1;
    mask[n] = (et.x1 == et.x1) ? 1.0f : 0.0f;     /* (the residual gate needs a target; not part of these terms) */
  }
}

/* ppf > 0: the reference's eff_impl = true path (ba_cuda.cu:261-275, 353-362, 472-478, 538-550): the coupling
 * matrix E is not stored as a dense [6N x M] matrix but as E_lookup[(i, j) block][patch within frame i][6]
 * (EfficentE, fastba/block_e.cu:43-145), and E Q E', E (Q u), E' dX are formed from the lookup by
 * EEt_kernel / Ev_kernel / Etv_kernel (block_e.cu:147-283).  Same algebra as the dense path, other summation
 * order (the reference's is atomic, i.e. unordered; here: block order).  ppf = patches per frame.          */
static int ba_impl(float *poses, float *patches, const float *intr, const float *target,
                   const float *weight, const float *lmbda_p, const int64_t *ii,
                   const int64_t *jj, const int64_t *kk, int E, int P, int t0, int t1,
                   int iterations, int ppf) {
  const float lmbda = lmbda_p[0];
  int status = 0;
  /* torch::_unique(kk, sorted, return_inverse)  (ba_cuda.cu:447-449) */
  int64_t *kx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(E > 0 ? E : 1));
  int *ku = (int *)malloc(sizeof(int) * (size_t)(E > 0 ? E : 1));
  memcpy(kx, kk, sizeof(int64_t) * (size_t)E);
  qsort(kx, E, sizeof(int64_t), i64cmp);
  int M = 0;
  for (int i = 0; i < E; i++) if (i == 0 || kx[i] != kx[i - 1]) kx[M++] = kx[i];
  for (int i = 0; i < E; i++) {
    int lo = 0, hi = M - 1;
    while (lo < hi) { int mid = (lo + hi) / 2; if (kx[mid] < kk[i]) lo = mid + 1; else hi = mid; }
    ku[i] = lo;
  }
  const int N = t1 - t0;
  const int n6 = 6 * N;
  float *B = (float *)malloc(sizeof(float) * (size_t)(n6 * n6 + 1));
  float *Em = (float *)malloc(sizeof(float) * ((size_t)n6 * M + 1));
  float *C = (float *)malloc(sizeof(float) * (size_t)(M + 1));
  float *v = (float *)malloc(sizeof(float) * (size_t)(n6 + 1));
  float *u = (float *)malloc(sizeof(float) * (size_t)(M + 1));
  float *Q = (float *)malloc(sizeof(float) * (size_t)(M + 1));
  float *S = (float *)malloc(sizeof(float) * (size_t)(n6 * n6 + 1));
  float *y = (float *)malloc(sizeof(float) * (size_t)(n6 + 1));
  float *dX = (float *)malloc(sizeof(float) * (size_t)(n6 + 1));
  float *dZ = (float *)malloc(sizeof(float) * (size_t)(M + 1));
  const int PP = P * P;

  /* ---- EfficentE::EfficentE (block_e.cu:43-145) */
  int nf = 0, nblk = 0, nidx = 0;
  int *blk_of = NULL, *ijx = NULL, *ijs = NULL, *blk_i = NULL, *blk_j = NULL, *p2ku = NULL, *idx5 = NULL;
  float *Elk = NULL;
  if (ppf > 0) {
    for (int n = 0; n < E; n++) { if (ii[n] + 1 > nf) nf = (int)ii[n] + 1; if (jj[n] + 1 > nf) nf = (int)jj[n] + 1; }
    blk_of = (int *)malloc(sizeof(int) * (size_t)nf * nf);          /* frame_to_idx: (i, j) -> block or -1 */
    for (int q = 0; q < nf * nf; q++) blk_of[q] = -1;
    for (int n = 0; n < E; n++) { blk_of[ii[n] * nf + jj[n]] = 0; blk_of[ii[n] * nf + ii[n]] = 0; }
    for (int q = 0; q < nf * nf; q++) if (blk_of[q] == 0) blk_of[q] = nblk++;       /* sorted unique of i * nf + j */
    blk_i = (int *)malloc(sizeof(int) * (size_t)(nblk + 1)); blk_j = (int *)malloc(sizeof(int) * (size_t)(nblk + 1));
    for (int q = 0; q < nf * nf; q++) if (blk_of[q] >= 0) { blk_i[blk_of[q]] = q / nf; blk_j[blk_of[q]] = q % nf; }
    ijx = (int *)malloc(sizeof(int) * (size_t)(E + 1)); ijs = (int *)malloc(sizeof(int) * (size_t)(E + 1));
    for (int n = 0; n < E; n++) { ijx[n] = blk_of[ii[n] * nf + jj[n]]; ijs[n] = blk_of[ii[n] * nf + ii[n]]; }
    Elk = (float *)malloc(sizeof(float) * ((size_t)nblk * ppf * 6 + 1));
    p2ku = (int *)malloc(sizeof(int) * ((size_t)nf * ppf + 1));       /* (frame, patch) -> row of Q or -1 */
    for (int q = 0; q < nf * ppf; q++) p2ku[q] = -1;
    for (int m = 0; m < M; m++) if (kx[m] / ppf < nf) p2ku[(kx[m] / ppf) * ppf + kx[m] % ppf] = m;
    /* index_tensor: for every source frame i all pairs (j1, j2) of frames it is connected to (incl. itself) */
    for (int i = 0; i < nf; i++) { int c = 0; for (int j = 0; j < nf; j++) c += blk_of[i * nf + j] >= 0; nidx += c * c; }
    idx5 = (int *)malloc(sizeof(int) * ((size_t)nidx * 5 + 1));
    int cx_ = 0;
    for (int i = 0; i < nf; i++)
      for (int j1 = 0; j1 < nf; j1++) {
        if (blk_of[i * nf + j1] < 0) continue;
        for (int j2 = 0; j2 < nf; j2++) {
          if (blk_of[i * nf + j2] < 0) continue;
          int *r = idx5 + 5 * (size_t)cx_++;
          r[0] = i; r[1] = j1; r[2] = j2; r[3] = blk_of[i * nf + j1]; r[4] = blk_of[i * nf + j2];
        }
      }
  }

  for (int itr = 0; itr < iterations; itr++) {
    memset(B, 0, sizeof(float) * (size_t)n6 * n6);
    memset(Em, 0, sizeof(float) * (size_t)n6 * M);
    memset(C, 0, sizeof(float) * (size_t)M);
    memset(v, 0, sizeof(float) * (size_t)n6);
    memset(u, 0, sizeof(float) * (size_t)M);
    if (ppf > 0) memset(Elk, 0, sizeof(float) * (size_t)nblk * ppf * 6);     /* blockE->E_lookup.zero_() */

    for (int n = 0; n < E; n++) {
      const int k = ku[n];
      int ix = (int)ii[n], jx = (int)jj[n];
      const int64_t kxn = kk[n];
      ba_edge_t et;
      ba_edge_terms(poses, patches, intr, ix, jx, kxn, P, target + 2 * n, &et);
      ix = ix - t0; jx = jx - t0;
      /* poses >= t1 are outside B in the reference (would be an out-of-bounds
       * write there); treated as fixed here and in the HIP path. */
      if (ix >= N) ix = -1;
      if (jx >= N) jx = -1;
      for (int row = 0; row < 2; row++) {
        const float *Jj = et.Jj[row], *Ji = et.Ji[row];
        const float Jz = et.Jz[row], r = et.r[row], w = et.mask * weight[2 * n + row];
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) {
            if (ix >= 0) B[(6 * ix + i) * n6 + 6 * ix + j] += w * Ji[i] * Ji[j];
            if (jx >= 0) B[(6 * jx + i) * n6 + 6 * jx + j] += w * Jj[i] * Jj[j];
            if (ix >= 0 && jx >= 0) {
              B[(6 * ix + i) * n6 + 6 * jx + j] += -w * Ji[i] * Jj[j];
              B[(6 * jx + i) * n6 + 6 * ix + j] += -w * Jj[i] * Ji[j];
            }
          }
        for (int i = 0; i < 6; i++) {
          if (ppf > 0) {   /* ba_cuda.cu:353-356: no test of ix / jx here, the lookup kernels test the block's frame */
            Elk[((size_t)ijs[n] * ppf + kxn % ppf) * 6 + i] += -w * Jz * Ji[i];
            Elk[((size_t)ijx[n] * ppf + kxn % ppf) * 6 + i] += w * Jz * Jj[i];
          } else {
            if (ix >= 0) Em[(size_t)(6 * ix + i) * M + k] += -w * Jz * Ji[i];
            if (jx >= 0) Em[(size_t)(6 * jx + i) * M + k] += w * Jz * Jj[i];
          }
        }
        for (int i = 0; i < 6; i++) {
          if (ix >= 0) v[6 * ix + i] += -w * r * Ji[i];
          if (jx >= 0) v[6 * jx + i] += w * r * Jj[i];
        }
        C[k] += w * Jz * Jz;
        u[k] += w * r * Jz;
      }
    }

    for (int k = 0; k < M; k++) Q[k] = 1.0f / (C[k] + lmbda);

    if (N == 0) { /* ba_cuda.cu:521-531 */
      for (int k = 0; k < M; k++) dZ[k] = Q[k] * u[k];
    } else {
      if (ppf > 0) {   /* ba_cuda.cu:538-550 over block_e.cu:147-283 */
      memcpy(S, B, sizeof(float) * (size_t)n6 * n6);
      memcpy(y, v, sizeof(float) * (size_t)n6);
      for (int r = 0; r < nidx; r++) {                       /* EEt_kernel: S -= E Q E' */
        const int *ix5 = idx5 + 5 * (size_t)r;
        const int j1 = ix5[1] - t0, j2 = ix5[2] - t0;
        if (j1 < 0 || j2 < 0 || j1 >= N || j2 >= N) continue;
        for (int kq = 0; kq < ppf; kq++) {
          const int qi = p2ku[ix5[0] * ppf + kq];
          if (qi < 0) continue;                                /* no such patch: its lookup slices are zero */
          const float *s1 = Elk + ((size_t)ix5[3] * ppf + kq) * 6, *s2 = Elk + ((size_t)ix5[4] * ppf + kq) * 6;
          for (int xi = 0; xi < 6; xi++)
            for (int xj = 0; xj < 6; xj++) S[(6 * j1 + xi) * n6 + 6 * j2 + xj] -= s1[xi] * s2[xj] * Q[qi];
        }
      }
      for (int bq = 0; bq < nblk; bq++) {                     /* Ev_kernel with vec = Q * u: y -= E (Q u) */
        const int jb = blk_j[bq] - t0;
        if (jb < 0 || jb >= N) continue;
        for (int kq = 0; kq < ppf; kq++) {
          const int qi = p2ku[blk_i[bq] * ppf + kq];
          if (qi < 0) continue;
          const float *sl = Elk + ((size_t)bq * ppf + kq) * 6;
          for (int r = 0; r < 6; r++) y[jb * 6 + r] -= sl[r] * (Q[qi] * u[qi]);
        }
      }
      } else {   /* ba_cuda.cu:552-565 */
      for (int a = 0; a < n6; a++) {
        for (int b = 0; b < n6; b++) {
          float s = 0;
          for (int k = 0; k < M; k++) s += (Em[(size_t)a * M + k] * Q[k]) * Em[(size_t)b * M + k];
          S[a * n6 + b] = B[a * n6 + b] - s;
        }
        float s = 0;
        for (int k = 0; k < M; k++) s += (Em[(size_t)a * M + k] * Q[k]) * u[k];
        y[a] = v[a] - s;
      }
      }
      for (int a = 0; a < n6; a++) S[a * n6 + a] += (1e-4f * S[a * n6 + a] + 1.0f);
      /* lower Cholesky in place + solve  (linalg_cholesky_ex / cholesky_solve) */
      for (int j = 0; j < n6; j++) {
        float s = S[j * n6 + j];
        for (int k = 0; k < j; k++) s -= S[j * n6 + k] * S[j * n6 + k];
        if (!(s > 0)) status = 1;
        const float ljj = sqrtf(s);
        S[j * n6 + j] = ljj;
        for (int i = j + 1; i < n6; i++) {
          float t = S[i * n6 + j];
          for (int k = 0; k < j; k++) t -= S[i * n6 + k] * S[j * n6 + k];
          S[i * n6 + j] = t / ljj;
        }
      }
      for (int i = 0; i < n6; i++) {
        float t = y[i];
        for (int k = 0; k < i; k++) t -= S[i * n6 + k] * dX[k];
        dX[i] = t / S[i * n6 + i];
      }
      for (int i = n6 - 1; i >= 0; i--) {
        float t = dX[i];
        for (int k = i + 1; k < n6; k++) t -= S[k * n6 + i] * dX[k];
        dX[i] = t / S[i * n6 + i];
      }
      if (ppf > 0) {                                          /* Etv_kernel: E' dX per patch */
        for (int k = 0; k < M; k++) dZ[k] = 0.0f;
        for (int bq = 0; bq < nblk; bq++) {
          const int jb = blk_j[bq] - t0;
          if (jb < 0 || jb >= N) continue;
          for (int kq = 0; kq < ppf; kq++) {
            const int qi = p2ku[blk_i[bq] * ppf + kq];
            if (qi < 0) continue;
            const float *sl = Elk + ((size_t)bq * ppf + kq) * 6;
            for (int r = 0; r < 6; r++) dZ[qi] += sl[r] * dX[jb * 6 + r];
          }
        }
        for (int k = 0; k < M; k++) dZ[k] = Q[k] * (u[k] - dZ[k]);
      } else {
        for (int k = 0; k < M; k++) {
          float s = 0;
          for (int a = 0; a < n6; a++) s += Em[(size_t)a * M + k] * dX[a];
          dZ[k] = Q[k] * (u[k] - s);
        }
      }
      for (int i = 0; i < N; i++) { /* pose_retr_kernel */
        float *p = poses + 7 * (t0 + i);
        float tn[3], qn[4], to[3] = {p[0], p[1], p[2]}, qo[4] = {p[3], p[4], p[5], p[6]};
        retrSE3(dX + 6 * i, to, qo, tn, qn);
        p[0] = tn[0]; p[1] = tn[1]; p[2] = tn[2];
        p[3] = qn[0]; p[4] = qn[1]; p[5] = qn[2]; p[6] = qn[3];
      }
    }
    for (int k = 0; k < M; k++) { /* patch_retr_kernel */
      float *pt = patches + ((size_t)kx[k] * 3 + 2) * PP;
      float dd = pt[0];
      dd = dd + dZ[k];
      dd = (dd > 20) ? 1.0f : dd;
      dd = fmaxf(dd, 1e-4f);
      for (int a = 0; a < PP; a++) pt[a] = dd;
    }
  }
  free(kx); free(ku); free(B); free(Em); free(C); free(v); free(u); free(Q);
  free(S); free(y); free(dX); free(dZ);
  free(blk_of); free(ijx); free(ijs); free(blk_i); free(blk_j); free(p2ku); free(idx5); free(Elk);
  return status;
}

/* cuda_ba.forward(..., eff_impl = false) */
ORC_API int orc_ba(float *poses, float *patches, const float *intr, const float *target,
                   const float *weight, const float *lmbda_p, const int64_t *ii,
                   const int64_t *jj, const int64_t *kk, int E, int P, int t0, int t1,
                   int iterations) {
  return ba_impl(poses, patches, intr, target, weight, lmbda_p, ii, jj, kk, E, P, t0, t1, iterations, 0);
}

/* cuda_ba.forward(..., PPF = ppf, eff_impl = true) */
ORC_API int orc_ba_eff(float *poses, float *patches, const float *intr, const float *target,
                       const float *weight, const float *lmbda_p, const int64_t *ii,
                       const int64_t *jj, const int64_t *kk, int E, int P, int t0, int t1,
                       int iterations, int ppf) {
  return ba_impl(poses, patches, intr, target, weight, lmbda_p, ii, jj, kk, E, P, t0, t1, iterations, ppf);
}

/* ------------------------------------------------------------------------- */
/* SoftAgg core (ramp/blocks.py:42-50 over torch_scatter semantics):          */
/* per group g and channel c:  y[g][c] = sum_e softmax_e(gx[e][c]) * fx[e][c] */
/* group[e] in [0,G).  Written the way torch_scatter does it: max, exp, sum,  */
/* divide, then weighted sum, all in edge order.                              */
/* ------------------------------------------------------------------------- */
ORC_API void orc_segment_softmax_sum(const float *fx, const float *gx, const int64_t *group,
                                     float *y, int E, int G, int C) {
  float *mx = (float *)malloc(sizeof(float) * (size_t)G * C);
  float *sm = (float *)malloc(sizeof(float) * (size_t)G * C);
  for (size_t i = 0; i < (size_t)G * C; i++) { mx[i] = -INFINITY; sm[i] = 0; y[i] = 0; }
  for (int e = 0; e < E; e++)
    for (int c = 0; c < C; c++) {
      float *m = mx + (size_t)group[e] * C + c;
      if (gx[(size_t)e * C + c] > *m) *m = gx[(size_t)e * C + c];
    }
  for (int e = 0; e < E; e++)
    for (int c = 0; c < C; c++)
      sm[(size_t)group[e] * C + c] += expf(gx[(size_t)e * C + c] - mx[(size_t)group[e] * C + c]);
  for (int e = 0; e < E; e++)
    for (int c = 0; c < C; c++) {
      const size_t g = (size_t)group[e] * C + c;
      const float w = expf(gx[(size_t)e * C + c] - mx[g]) / sm[g];
      y[g] += fx[(size_t)e * C + c] * w;
    }
  free(mx); free(sm);
}
