// Device-side helpers shared by the gfx950 kernels of libramp_hip.so.
// All translation units are compiled with -ffp-contract=off: FMA is used only
// where it is written explicitly (__builtin_fmaf), so float results follow the
// expression order of the reference and are comparable with the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/ramp_hip.h"

#define RAMP_WAVE 64

#define RAMP_CHECK_LAUNCH()                               \
  do {                                                    \
    hipError_t _e = hipGetLastError();                    \
    if (_e != hipSuccess) return RAMP_ELAUNCH;            \
  } while (0)

static inline int ramp_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// float -> int, saturating (v_cvt_i32_f32 semantics; NaN -> 0)
__device__ __forceinline__ int ramp_f2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483520.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

// ---------------------------------------------------------------------------
// fastba-flavoured SE3 helpers (quaternions taken as stored, no normalisation)
// reference: ramp/fastba/ba_cuda.cu:36-174
// ---------------------------------------------------------------------------
__device__ __forceinline__ void fb_actSO3(const float *q, const float *X, float *Y) {
  float uv[3];
  uv[0] = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  uv[1] = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  uv[2] = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  Y[1] = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  Y[2] = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
__device__ __forceinline__ void fb_actSE3(const float *t, const float *q, const float *X,
                                          float *Y) {
  fb_actSO3(q, X, Y);
  Y[3] = X[3];
  Y[0] += X[3] * t[0];
  Y[1] += X[3] * t[1];
  Y[2] += X[3] * t[2];
}
__device__ __forceinline__ void fb_adjSE3(const float *t, const float *q, const float *X,
                                          float *Y) {
  float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  fb_actSO3(qinv, &X[0], &Y[0]);
  fb_actSO3(qinv, &X[3], &Y[3]);
  float u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  fb_actSO3(qinv, u, v);
  Y[3] += v[0];
  Y[4] += v[1];
  Y[5] += v[2];
}
__device__ __forceinline__ void fb_relSE3(const float *ti, const float *qi, const float *tj,
                                          const float *qj, float *tij, float *qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  fb_actSO3(qij, ti, tij);
  tij[0] = tj[0] - tij[0];
  tij[1] = tj[1] - tij[1];
  tij[2] = tj[2] - tij[2];
}
__device__ __forceinline__ void fb_expSO3(const float *phi, float *q) {
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
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = real;
}
__device__ __forceinline__ void fb_cross_inplace(const float *a, float *b) {
  float x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2],
                a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0];
  b[1] = x[1];
  b[2] = x[2];
}
__device__ __forceinline__ void fb_expSE3(const float *xi, float *t, float *q) {
  fb_expSO3(xi + 3, q);
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi[5]};
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta = sqrtf(theta_sq);
  t[0] = tau[0];
  t[1] = tau[1];
  t[2] = tau[2];
  if (theta > 1e-4f) {
    const float a = (1 - cosf(theta)) / theta_sq;
    fb_cross_inplace(phi, tau);
    t[0] += a * tau[0];
    t[1] += a * tau[1];
    t[2] += a * tau[2];
    const float b = (theta - sinf(theta)) / (theta * theta_sq);
    fb_cross_inplace(phi, tau);
    t[0] += b * tau[0];
    t[1] += b * tau[1];
    t[2] += b * tau[2];
  }
}
__device__ __forceinline__ void fb_retrSE3(const float *xi, const float *t, const float *q,
                                           float *t1, float *q1) {
  float dt[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1};
  fb_expSE3(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  fb_actSO3(dq, t, t1);
  t1[0] += dt[0];
  t1[1] += dt[1];
  t1[2] += dt[2];
}

// ---------------------------------------------------------------------------
// lietorch-flavoured SE3 (quaternion normalised on construction, EPS branches)
// reference: ramp/lietorch/include/so3.h:31-208, se3.h:30-142, common.h:7
// ---------------------------------------------------------------------------
#define LT_EPS 1e-6f
#define LT_PI 3.14159265358979323846f

__device__ __forceinline__ void lt_qnorm(const float *q, float *o) {
  const float n = sqrtf((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
  o[0] = q[0] / n;
  o[1] = q[1] / n;
  o[2] = q[2] / n;
  o[3] = q[3] / n;
}
__device__ __forceinline__ void lt_qmul(const float *a, const float *b, float *o) {
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void lt_qrot(const float *q, const float *p, float *o) {
  float uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2],
                 q[0] * p[1] - q[1] * p[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  o[0] = p[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = p[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = p[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
__device__ __forceinline__ void lt_q2R(const float *q, float *R) {
  const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void lt_hat(const float *p, float *M) {
  M[0] = 0; M[1] = -p[2]; M[2] = p[1];
  M[3] = p[2]; M[4] = 0; M[5] = -p[0];
  M[6] = -p[1]; M[7] = p[0]; M[8] = 0;
}
__device__ __forceinline__ void lt_m3mul(const float *A, const float *B, float *C) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void lt_m3vec(const float *A, const float *v, float *o) {
#pragma unroll
  for (int i = 0; i < 3; i++)
    o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void lt_so3_exp(const float *phi, float *q) {
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
  lt_qnorm(r, q);
}
__device__ __forceinline__ void lt_so3_log(const float *q, float *phi) {
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
  phi[0] = k * q[0];
  phi[1] = k * q[1];
  phi[2] = k * q[2];
}
__device__ __forceinline__ void lt_left_jacobian(const float *phi, float *J) {
  float Phi[9], Phi2[9];
  lt_hat(phi, Phi);
  lt_m3mul(Phi, Phi, Phi2);
  const float t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float t = sqrtf(t2);
  const float c1 = (t < LT_EPS) ? 0.5f - (1.0f / 24.0f) * t2 : (1.0f - cosf(t)) / t2;
  const float c2 =
      (t < LT_EPS) ? (1.0f / 6.0f) - (1.0f / 120.0f) * t2 : (t - sinf(t)) / (t2 * t);
#pragma unroll
  for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + c1 * Phi[i] + c2 * Phi2[i];
}
__device__ __forceinline__ void lt_left_jacobian_inv(const float *phi, float *J) {
  float Phi[9], Phi2[9];
  lt_hat(phi, Phi);
  lt_m3mul(Phi, Phi, Phi2);
  const float t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float t = sqrtf(t2), ht = 0.5f * t;
  const float c2 = (t < LT_EPS) ? (1.0f / 12.0f)
                                : (1.0f - t * cosf(ht) / (2.0f * sinf(ht))) / (t * t);
#pragma unroll
  for (int i = 0; i < 9; i++)
    J[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + (-0.5f) * Phi[i] + c2 * Phi2[i];
}
__device__ __forceinline__ void lt_load(const float *d, float *t, float *q) {
  t[0] = d[0];
  t[1] = d[1];
  t[2] = d[2];
  lt_qnorm(d + 3, q);
}
__device__ __forceinline__ void lt_inv(const float *X, float *Y) {
  float t[3], q[4], qi[4], qn[4], r[3];
  lt_load(X, t, q);
  qi[0] = -q[0]; qi[1] = -q[1]; qi[2] = -q[2]; qi[3] = q[3];
  lt_qnorm(qi, qn);
  lt_qrot(qn, t, r);
  Y[0] = -r[0]; Y[1] = -r[1]; Y[2] = -r[2];
  Y[3] = qn[0]; Y[4] = qn[1]; Y[5] = qn[2]; Y[6] = qn[3];
}
__device__ __forceinline__ void lt_mul(const float *X, const float *Y, float *Z) {
  float tx[3], qx[4], ty[3], qy[4], qz[4], qn[4], r[3];
  lt_load(X, tx, qx);
  lt_load(Y, ty, qy);
  lt_qmul(qx, qy, qz);
  lt_qnorm(qz, qn);
  lt_qrot(qx, ty, r);
  Z[0] = tx[0] + r[0]; Z[1] = tx[1] + r[1]; Z[2] = tx[2] + r[2];
  Z[3] = qn[0]; Z[4] = qn[1]; Z[5] = qn[2]; Z[6] = qn[3];
}
// act4 with t,q already loaded (normalised)
__device__ __forceinline__ void lt_act4_tq(const float *t, const float *q, const float *p,
                                           float *o) {
  float r[3];
  lt_qrot(q, p, r);
  o[0] = r[0] + t[0] * p[3];
  o[1] = r[1] + t[1] * p[3];
  o[2] = r[2] + t[2] * p[3];
  o[3] = p[3];
}
__device__ __forceinline__ void lt_exp(const float *xi, float *X) {
  float q[4], J[9], t[3];
  lt_so3_exp(xi + 3, q);
  lt_left_jacobian(xi + 3, J);
  lt_m3vec(J, xi, t);
  X[0] = t[0]; X[1] = t[1]; X[2] = t[2];
  X[3] = q[0]; X[4] = q[1]; X[5] = q[2]; X[6] = q[3];
}
__device__ __forceinline__ void lt_log(const float *X, float *xi) {
  float t[3], q[4], phi[3], Vi[9], tau[3];
  lt_load(X, t, q);
  lt_so3_log(q, phi);
  lt_left_jacobian_inv(phi, Vi);
  lt_m3vec(Vi, t, tau);
  xi[0] = tau[0]; xi[1] = tau[1]; xi[2] = tau[2];
  xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
}
__device__ __forceinline__ void lt_Adj(const float *X, float *Ad) {
  float t[3], q[4], R[9], T[9], TR[9];
  lt_load(X, t, q);
  lt_q2R(q, R);
  lt_hat(t, T);
  lt_m3mul(T, R, TR);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Ad[i * 6 + j] = R[i * 3 + j];
      Ad[i * 6 + 3 + j] = TR[i * 3 + j];
      Ad[(i + 3) * 6 + j] = 0;
      Ad[(i + 3) * 6 + 3 + j] = R[i * 3 + j];
    }
}
