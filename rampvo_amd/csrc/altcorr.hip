// altcorr: patch gather (patchify) and the fused sparse patch correlation.
//
// corr design (gfx950):
//   one wavefront (64 lanes) per edge, both pyramid levels in the same block.
//   The 9 patch pixels of an edge look up 8x8 windows that overlap almost
//   entirely, so instead of 9*64 independent 128-long dot products the block
//   forms the UNION of the 9 windows (typically 10x10 at level 0, 9x9 at level
//   1), gives every union pixel to a lane (2 per lane), streams that pixel's
//   128 channels ONCE from HBM (channels-last: 512 contiguous bytes) and
//   accumulates it against all 9 patch pixels (held in LDS, read as wave-wide
//   broadcasts).  The 9 x T dot-product matrix goes to LDS, the bilinear blend
//   + permute + level interleave of the reference's host code is applied from
//   there and the edge's 882 outputs leave as one contiguous 3.5 KB store.
//   Edges whose reprojected patch is so distorted that the union exceeds 128
//   pixels fall back to one 8x8 window per patch pixel (same code, 9 groups).
//   Dots are a channel-ordered fmaf chain == the reference kernel's
//   accumulation (correlation_kernel.cu:121-131) in fp32.
#include "ramp_device.h"
#include <stdlib.h>

// ---------------------------------------------------------------- patchify
template <typename T>
__device__ __forceinline__ float ld_as_float(const T *p) { return (float)(*p); }
template <>
__device__ __forceinline__ float ld_as_float<__half>(const __half *p) { return __half2float(*p); }
template <typename T>
__device__ __forceinline__ void st_from_float(T *p, float v) { *p = (T)v; }
template <>
__device__ __forceinline__ void st_from_float<__half>(__half *p, float v) { *p = __float2half(v); }

// reference: correlation_kernel.cu:16-47 (gather) + correlation.py:51-68 (blend)
template <typename T>
__global__ void __launch_bounds__(256)
    patchify_kernel(const T *__restrict__ net, const float *__restrict__ coords,
                    T *__restrict__ out, int C, int H, int W, int M, int R, int bilinear,
                    int layout, int out_layout) {
  const int bm = blockIdx.x;  // n*M + m
  const int b = bm / M;
  const float x = coords[2 * (size_t)bm + 0];
  const float y = coords[2 * (size_t)bm + 1];
  const float flx = floorf(x), fly = floorf(y);
  const int fx = ramp_f2i(flx), fy = ramp_f2i(fly);
  const int d = bilinear ? 2 * R + 1 : 2 * R + 2;
  const float dx = x - flx, dy = y - fly;
  const float w00 = (1 - dy) * (1 - dx), w01 = (1 - dy) * dx, w10 = dy * (1 - dx),
              w11 = dy * dx;
  const int total = C * d * d;
  const T *nb = net + (size_t)b * C * H * W;
  for (int o = threadIdx.x; o < total; o += blockDim.x) {
    int k, a, c;
    if (out_layout == RAMP_NHWC) { k = o % C; c = (o / C) % d; a = o / (C * d); }
    else { c = o % d; a = (o / d) % d; k = o / (d * d); }
    auto tap = [&](int aa, int cc) -> float {
      const long i = (long)fy + (aa - R), j = (long)fx + (cc - R);
      if (i < 0 || i >= H || j < 0 || j >= W) return 0.0f;
      const size_t off = (layout == RAMP_NHWC) ? ((size_t)i * W + j) * C + k
                                               : ((size_t)k * H + i) * W + j;
      return ld_as_float(nb + off);
    };
    float s;
    if (bilinear) {
      s = w00 * tap(a, c);
      s = s + w01 * tap(a, c + 1);
      s = s + w10 * tap(a + 1, c);
      s = s + w11 * tap(a + 1, c + 1);
    } else {
      s = tap(a, c);
    }
    st_from_float(out + (size_t)bm * total + o, s);
  }
}

// Everything the tracker gathers at a frame's patch centres, one launch (reference ramp/net.py:167-203: four patchify
// calls -- gmap 3x3x128 from fmap, imap 1x1x384, the 3x3 (x, y, disparity) patches from the coordinate grid, the colours from
// the full-resolution image -- plus the elementwise steps around them, and Ramp_vo.py:353-354's uint8 BGR colours).
// Same blend expression as patchify_kernel, tap by tap, so the results are bit-identical to the separate launches.
// The coordinate grid is not read: its value at (i, j) is (j, i, 1).
struct FrameGather {
  const void *fmap, *imap;        // NHWC [h][w][CF], [h][w][CI]
  const float *image;             // [3][H][W]
  const float *coords;            // [M][2] at feature resolution
  void *gmap, *imap_p;            // [M][3][3][CF], [M][CI]
  float *patches, *clr;           // [M][3][3][3] (channel-major), [M][3]
  unsigned char *colors;          // [M][3] BGR
  int h, w, H, W, CF, CI;
};

template <typename T>
__global__ void __launch_bounds__(256) frame_gather_kernel(const FrameGather p) {
  const int m = blockIdx.x;
  const float x = p.coords[2 * (size_t)m + 0], y = p.coords[2 * (size_t)m + 1];
  auto blend = [](float xx, float yy, int &fx, int &fy, float &w00, float &w01, float &w10, float &w11) {
    const float flx = floorf(xx), fly = floorf(yy);
    fx = ramp_f2i(flx); fy = ramp_f2i(fly);
    const float dx = xx - flx, dy = yy - fly;
    w00 = (1 - dy) * (1 - dx); w01 = (1 - dy) * dx; w10 = dy * (1 - dx); w11 = dy * dx;
  };
  int fx, fy;
  float w00, w01, w10, w11;
  blend(x, y, fx, fy, w00, w01, w10, w11);
  const T *fm = reinterpret_cast<const T *>(p.fmap), *im = reinterpret_cast<const T *>(p.imap);
  auto feat = [&](const T *base, int C, int k, long i, long j) -> float {
    if (i < 0 || i >= p.h || j < 0 || j >= p.w) return 0.0f;
    return ld_as_float(base + ((size_t)i * p.w + j) * C + k);
  };
  // gmap: radius 1, channels-last out
  for (int o = threadIdx.x; o < 9 * p.CF; o += blockDim.x) {
    const int k = o % p.CF, c = (o / p.CF) % 3, a = o / (p.CF * 3);
    const long i = (long)fy + (a - 1), j = (long)fx + (c - 1);
    float s = w00 * feat(fm, p.CF, k, i, j);
    s = s + w01 * feat(fm, p.CF, k, i, j + 1);
    s = s + w10 * feat(fm, p.CF, k, i + 1, j);
    s = s + w11 * feat(fm, p.CF, k, i + 1, j + 1);
    st_from_float(reinterpret_cast<T *>(p.gmap) + (size_t)m * 9 * p.CF + o, s);
  }
  // imap: radius 0
  for (int k = threadIdx.x; k < p.CI; k += blockDim.x) {
    float s = w00 * feat(im, p.CI, k, fy, fx);
    s = s + w01 * feat(im, p.CI, k, fy, (long)fx + 1);
    s = s + w10 * feat(im, p.CI, k, (long)fy + 1, fx);
    s = s + w11 * feat(im, p.CI, k, (long)fy + 1, (long)fx + 1);
    st_from_float(reinterpret_cast<T *>(p.imap_p) + (size_t)m * p.CI + k, s);
  }
  // patches: the (x, y, disparity = 1) grid, radius 1, channel-major out
  if (threadIdx.x < 27) {
    const int o = threadIdx.x, c = o % 3, a = (o / 3) % 3, k = o / 9;
    auto g = [&](long i, long j) -> float {
      if (i < 0 || i >= p.h || j < 0 || j >= p.w) return 0.0f;
      return k == 0 ? (float)j : (k == 1 ? (float)i : 1.0f);
    };
    const long i = (long)fy + (a - 1), j = (long)fx + (c - 1);
    float s = w00 * g(i, j);
    s = s + w01 * g(i, j + 1);
    s = s + w10 * g(i + 1, j);
    s = s + w11 * g(i + 1, j + 1);
    p.patches[(size_t)m * 27 + o] = s;
  }
  // colours: the image at 4 (coords + 0.5), radius 0
  if (threadIdx.x >= 64 && threadIdx.x < 67) {
    const int k = threadIdx.x - 64;
    int qx, qy;
    float v00, v01, v10, v11;
    blend(4.0f * (x + 0.5f), 4.0f * (y + 0.5f), qx, qy, v00, v01, v10, v11);
    auto px = [&](long i, long j) -> float {
      if (i < 0 || i >= p.H || j < 0 || j >= p.W) return 0.0f;
      return p.image[((size_t)k * p.H + i) * p.W + j];
    };
    float s = v00 * px(qy, qx);
    s = s + v01 * px(qy, (long)qx + 1);
    s = s + v10 * px((long)qy + 1, qx);
    s = s + v11 * px((long)qy + 1, (long)qx + 1);
    p.clr[(size_t)m * 3 + k] = s;
    // torch's float -> uint8 goes through int64 (c10 static_cast_with_inter_type)
    p.colors[(size_t)m * 3 + (2 - k)] = (unsigned char)(long long)((s + 0.5f) * 127.5f);
  }
}

// -------------------------------------------------------------------- corr
#define CORR_MAXLEV 2
#ifndef CORR_PGB
#define CORR_PGB 4    // pixel groups (of 16) whose loads are in flight together, MFMA kernel
#define CORR_WAVES 4  // waves per SIMD the MFMA kernel is register-budgeted for
#endif
#define CORR_T 128  // union pixels handled per group (2 per lane)

struct CorrParams {
  const void *fmap1;
  const void *fmap2[CORR_MAXLEV];
  int H2[CORR_MAXLEV], W2[CORR_MAXLEV];
  float cdiv[CORR_MAXLEV];
  int nlevels;
  const float *coords;
  const int64_t *ii, *jj;
  void *out;
  int E, N1, N2;
  long mod_ii, mod_jj;    // > 0: ii / jj are taken modulo these (the tracker's ring buffers)
  int row_elems;          // elements per edge row of `out` (>= 441 * nlevels; the tail is zero filled)
  const int32_t *order;   // optional schedule: position -> edge (any permutation of 0..E-1)
  int chunk;              // ceil(E / CORR_XCDS)
  const int32_t *dyn;     // optional device-side sizes (RAMP_DYN_E): E / chunk above are then the launch bound
  const int32_t *list;    // list mode (corr_mfma_list_kernel): the edges to compute, *list_n of them
  int32_t *list_n;        // (... and three ints behind it: a ticket; the last workgroup out zeroes both)
};

// Workgroup ids are dealt round-robin to the 8 XCDs, each with a private L2.  Position p of the
// schedule is given to XCD p / chunk, so that a run of consecutive positions -- edges that look at
// the same target frame when the caller passes a jj-major `order` -- shares one L2 instead of
// pulling the frame's feature plane into all eight.
constexpr int CORR_XCDS = 8;
static __device__ __forceinline__ int corr_edge_of_block(const CorrParams &prm) {
  const int b = blockIdx.x;
  int E = prm.E, chunk = prm.chunk;
  if (prm.dyn) {                                   // the live edge count lives in device memory
    E = prm.dyn[RAMP_DYN_E];
    chunk = (E + CORR_XCDS - 1) / CORR_XCDS;
    if (b / CORR_XCDS >= chunk) return -1;
  }
  const int pos = (b % CORR_XCDS) * chunk + b / CORR_XCDS;
  if (pos >= E) return -1;
  return prm.order ? prm.order[pos] : pos;
}

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ void load(const float *p, float *o) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
};
template <> struct Vec4<__half> {
  static __device__ __forceinline__ void load(const __half *p, float *o) {
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    const __half2 a = *reinterpret_cast<const __half2 *>(&v.x);
    const __half2 b = *reinterpret_cast<const __half2 *>(&v.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    o[0] = fa.x; o[1] = fa.y; o[2] = fb.x; o[3] = fb.y;
  }
};

template <typename T, int LAYOUT>
__global__ void __launch_bounds__(64)
    corr_kernel(const CorrParams prm) {
  constexpr int C = 128, P = 3, PP = 9, R = 3, D = 8, d = 7;
  constexpr int NOUT = d * d * PP;  // 441
  __shared__ __attribute__((aligned(16))) float f1s[PP * C];     // [p][c]
  __shared__ __attribute__((aligned(16))) float Cs[PP * CORR_T];  // [p][t]
  __shared__ float outs[NOUT * CORR_MAXLEV];
  __shared__ int s_ox[PP], s_oy[PP], s_live[PP];
  __shared__ float s_dx[PP], s_dy[PP];

  const int e = corr_edge_of_block(prm);
  if (e < 0) return;
  const int lane = threadIdx.x;
  const long i1 = prm.mod_ii > 0 ? prm.ii[e] % prm.mod_ii : prm.ii[e];   // ring-buffer slots (Ramp_vo.py:178-179)
  const long j2 = prm.mod_jj > 0 ? prm.jj[e] % prm.mod_jj : prm.jj[e];
  const int L = prm.nlevels;

  // ---- stage the patch features as fp32 [p][c]
  {
    const T *src = reinterpret_cast<const T *>(prm.fmap1) + (size_t)i1 * C * PP;
    if (LAYOUT == RAMP_NHWC) {
      for (int q = lane; q < PP * C / 4; q += 64) {
        float v[4];
        Vec4<T>::load(src + 4 * q, v);
        *reinterpret_cast<float4 *>(&f1s[4 * q]) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
      for (int q = lane; q < PP * C; q += 64) {  // src index q = c*9 + p
        const int c = q / PP, p = q - c * PP;
        f1s[p * C + c] = ld_as_float(src + q);
      }
    }
  }

  for (int lvl = 0; lvl < L; lvl++) {
    const int H2 = prm.H2[lvl], W2 = prm.W2[lvl];
    const T *f2 = reinterpret_cast<const T *>(prm.fmap2[lvl]) + (size_t)j2 * C * H2 * W2;
    if (lane < PP) {
      const float cdv = prm.cdiv[lvl];
      const float x = prm.coords[((size_t)e * 2 + 0) * PP + lane] / cdv;
      const float y = prm.coords[((size_t)e * 2 + 1) * PP + lane] / cdv;
      const float flx = floorf(x), fly = floorf(y);
      const int ox = ramp_f2i(flx), oy = ramp_f2i(fly);
      s_dx[lane] = x - flx;
      s_dy[lane] = y - fly;
      // window [o-R, o-R+D) x [o-R, o-R+D) intersects the image?
      const bool live = ((long)ox - R < W2) && ((long)ox - R + D > 0) &&
                        ((long)oy - R < H2) && ((long)oy - R + D > 0);
      s_live[lane] = live ? 1 : 0;
      s_ox[lane] = live ? ox - R : 0;
      s_oy[lane] = live ? oy - R : 0;
    }
    __syncthreads();
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30), nlive = 0;
#pragma unroll
    for (int p = 0; p < PP; p++) {
      if (s_live[p]) {
        nlive++;
        minx = min(minx, s_ox[p]); maxx = max(maxx, s_ox[p]);
        miny = min(miny, s_oy[p]); maxy = max(maxy, s_oy[p]);
      }
    }
    const long bw = (long)maxx - minx + D, bh = (long)maxy - miny + D;
    const bool uni = (nlive > 0) && (bw * bh <= CORR_T);
    const int ngroups = (nlive == 0) ? 0 : (uni ? 1 : PP);

    if (nlive == 0) {
      // every window is outside the image: the raw correlations are all zero,
      // the blend still multiplies them by the (possibly NaN) weights
      for (int o = lane; o < NOUT; o += 64) {
        const int p = o % PP;
        const float dx = s_dx[p], dy = s_dy[p];
        float s = ((1 - dx) * (1 - dy)) * 0.0f;
        s = s + (dx * (1 - dy)) * 0.0f;
        s = s + ((1 - dx) * dy) * 0.0f;
        s = s + (dx * dy) * 0.0f;
        outs[o * L + lvl] = s;
      }
    }

    for (int g = 0; g < ngroups; g++) {
      if (!uni && !s_live[g]) {
        for (int ab = lane; ab < d * d; ab += 64) {
          const float dx = s_dx[g], dy = s_dy[g];
          float s = ((1 - dx) * (1 - dy)) * 0.0f;
          s = s + (dx * (1 - dy)) * 0.0f;
          s = s + ((1 - dx) * dy) * 0.0f;
          s = s + (dx * dy) * 0.0f;
          outs[(ab * PP + g) * L + lvl] = s;
        }
        continue;
      }
      const int gx0 = uni ? minx : s_ox[g], gy0 = uni ? miny : s_oy[g];
      const int gw = uni ? (int)bw : D, gh = uni ? (int)bh : D;
      const int Tn = gw * gh;

      float acc[2][PP];
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int p = 0; p < PP; p++) acc[s][p] = 0.0f;
      bool inb[2];
      size_t poff[2];
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int t = lane + 64 * s;
        const int ty = t / gw, tx = t - ty * gw;
        const int px = gx0 + tx, py = gy0 + ty;
        inb[s] = (t < Tn) && px >= 0 && px < W2 && py >= 0 && py < H2;
        poff[s] = (LAYOUT == RAMP_NHWC) ? ((size_t)py * W2 + px) * C : ((size_t)py * W2 + px);
      }
      if (LAYOUT == RAMP_NHWC) {
#pragma unroll 2
        for (int c4 = 0; c4 < C / 4; c4++) {
          float v[2][4];
#pragma unroll
          for (int s = 0; s < 2; s++) {
            if (inb[s]) Vec4<T>::load(f2 + poff[s] + 4 * c4, v[s]);
            else { v[s][0] = v[s][1] = v[s][2] = v[s][3] = 0.0f; }
          }
#pragma unroll
          for (int p = 0; p < PP; p++) {
            const float4 a = *reinterpret_cast<const float4 *>(&f1s[p * C + 4 * c4]);
#pragma unroll
            for (int s = 0; s < 2; s++) {
              float r = acc[s][p];
              r = __builtin_fmaf(a.x, v[s][0], r);
              r = __builtin_fmaf(a.y, v[s][1], r);
              r = __builtin_fmaf(a.z, v[s][2], r);
              r = __builtin_fmaf(a.w, v[s][3], r);
              acc[s][p] = r;
            }
          }
        }
      } else {
        const size_t cstride = (size_t)H2 * W2;
#pragma unroll 4
        for (int c = 0; c < C; c++) {
          float v[2];
#pragma unroll
          for (int s = 0; s < 2; s++) v[s] = inb[s] ? ld_as_float(f2 + c * cstride + poff[s]) : 0.0f;
#pragma unroll
          for (int p = 0; p < PP; p++) {
            const float a = f1s[p * C + c];
            acc[0][p] = __builtin_fmaf(a, v[0], acc[0][p]);
            acc[1][p] = __builtin_fmaf(a, v[1], acc[1][p]);
          }
        }
      }
      // out-of-image pixels contribute exact zeros (reference: s = 0)
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int t = lane + 64 * s;
        if (t < Tn) {
#pragma unroll
          for (int p = 0; p < PP; p++) Cs[p * CORR_T + t] = inb[s] ? acc[s][p] : 0.0f;
        }
      }
      __syncthreads();
      // blend 8x8 -> 7x7 and apply the reference's permute
      const int nout = uni ? NOUT : d * d;
      for (int o = lane; o < nout; o += 64) {
        const int p = uni ? (o % PP) : g;
        const int ab = uni ? (o / PP) : o;
        const int b = ab / d, a = ab - b * d;  // b: x offset, a: y offset
        float c00 = 0, c01 = 0, c10 = 0, c11 = 0;
        if (s_live[p]) {
          const int wx = s_ox[p] - gx0 + b, wy = s_oy[p] - gy0 + a;
          const float *row = &Cs[p * CORR_T + wy * gw + wx];
          c00 = row[0]; c01 = row[1]; c10 = row[gw]; c11 = row[gw + 1];
        }
        const float dx = s_dx[p], dy = s_dy[p];
        float s = ((1 - dx) * (1 - dy)) * c00;
        s = s + (dx * (1 - dy)) * c01;
        s = s + ((1 - dx) * dy) * c10;
        s = s + (dx * dy) * c11;
        outs[(ab * PP + p) * L + lvl] = s;
      }
      __syncthreads();
    }
    __syncthreads();
  }
  __syncthreads();
  T *o = reinterpret_cast<T *>(prm.out) + (size_t)e * prm.row_elems;
  for (int q = lane; q < NOUT * L; q += 64) st_from_float(o + q, outs[q]);
  for (int q = NOUT * L + lane; q < prm.row_elems; q += 64) st_from_float(o + q, 0.0f);
}


// ---------------------------------------------------------------- fp16 / MFMA
// Mixed-precision path (fp16 features, the reference's default MIXED_PRECISION): the 9 x T
// dot-product matrix of an edge is a [16(9 used) x 128] x [128 x 16] product per group of 16
// union pixels -> v_mfma_f32_16x16x32_f16, fp32 accumulation (the reference accumulates in
// half).  A (patch features) lives in 16 VGPRs for the whole edge; B is one 16-byte
// channels-last load per lane per MFMA; no LDS in the main loop.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

//
// Target maps come either as plain NHWC or (CHUNKED) as [H][C/8][W][8]: there the 16 lanes of a
// quarter-wave -- 16 neighbouring window pixels, same 8-channel chunk -- read one or two
// contiguous runs instead of 16 cache lines 256 B apart.  The vector L1 looks up one line per
// cycle, and with NHWC those lookups (64 per load instruction) were what the kernel waited on.
// element-type traits of the MFMA correlation kernel below: fp16 -> v_mfma_f32_16x16x32_f16 (4 steps of 32 channels,
// lane (q, .) supplies 8 channels per step); fp32 -> v_mfma_f32_16x16x4_f32 (exact fp32 products; the channel axis is
// permuted so that ONE 16-byte load per lane feeds 4 MFMA steps on both operands: lane (q, .) of load g holds
// channels 16 g + 4 q + t, t = 0..3, and step (g, t) contracts them: 32 MFMAs per group of 16 window pixels).
// The fp32 variant is opt-in (dtype | RAMP_CORR_MFMA32): 2.1x faster than corr_kernel<float> (445 vs 940 us at E = 40k)
// but its accumulation order is the MFMA's, not the reference kernel's channel-ordered fmaf chain that
// corr_kernel<float> reproduces bit for bit -- so the exact-parity path stays the default.
template <typename T> struct CorrMma;
template <> struct CorrMma<_Float16> {
  typedef f16x8_t frag;
  static constexpr int STEPS = 4, PER = 8, PGB = CORR_PGB, WAVES = CORR_WAVES;
  static __device__ __forceinline__ frag zero() { return (frag){0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ f32x4_t mma(const frag &a, const frag &b, f32x4_t acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void st(_Float16 *p, float v) { *p = (_Float16)v; }
  static __device__ __forceinline__ void st2(_Float16 *p, float a, float b) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    *reinterpret_cast<h2v *>(p) = (h2v){(_Float16)a, (_Float16)b};
  }
};
template <> struct CorrMma<float> {
  typedef f32x4_t frag;
  static constexpr int STEPS = 8, PER = 4, PGB = 2, WAVES = 3;
  static __device__ __forceinline__ frag zero() { return (frag){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4_t mma(const frag &a, const frag &b, f32x4_t acc) {
#pragma unroll
    for (int t4 = 0; t4 < 4; t4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t4], b[t4], acc, 0, 0, 0);
    return acc;
  }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
  static __device__ __forceinline__ void st2(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
};

#ifdef CORR_TRACE
// phase timeline of corr_mfma_kernel (tools/corr_trace.py): shader-clock stamps of every wave, summed per phase
#define CORR_TRACE_WAVES 65536
__device__ unsigned long long g_corr_trace[CORR_TRACE_WAVES * 8];    // one row per workgroup (= wave = edge), no atomics
#define CTS(var) const unsigned long long var = __builtin_readcyclecounter()
#define CTACC(k, v) do { if (threadIdx.x == 0 && blockIdx.x < CORR_TRACE_WAVES) g_corr_trace[blockIdx.x * 8 + (k)] = (unsigned long long)(v); } while (0)
#else
#define CTS(var)
#define CTACC(k, v)
#endif
template <typename T, bool CHUNKED>
__device__ __forceinline__ void corr_mfma_edge(const CorrParams &prm, const int e) {
  typedef CorrMma<T> M;
  typedef typename M::frag frag_t;
  constexpr int STEPS = M::STEPS, PER = M::PER;   // MFMA-operand loads per pixel, channels per lane and load
  constexpr int C = 128, PP = 9, R = 3, D = 8, d = 7;
  constexpr int NOUT = d * d * PP;           // 441 values per level
  constexpr int KOUT = d;                    // 7 per lane (lanes 0..62), held until both levels are done
  constexpr int PGB = M::PGB;                     // pixel groups whose loads are in flight together
  __shared__ __attribute__((aligned(16))) float Cs[PP * CORR_T];
  __shared__ float outs[NOUT];               // staging for the ragged (non-union) paths only
  __shared__ int s_ox[PP], s_oy[PP], s_live[PP];
  __shared__ float s_dx[PP], s_dy[PP];

  CTS(ct_start);
#ifdef CORR_TRACE
  unsigned long long ct_load = 0, ct_mma = 0, ct_blend = 0, ct_setup = 0;
#endif
  const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
  const long i1 = prm.mod_ii > 0 ? prm.ii[e] % prm.mod_ii : prm.ii[e];   // ring-buffer slots (Ramp_vo.py:178-179)
  const long j2 = prm.mod_jj > 0 ? prm.jj[e] % prm.mod_jj : prm.jj[e];
  const int L = prm.nlevels;

  frag_t afrag[STEPS];
  {
    const T *src = reinterpret_cast<const T *>(prm.fmap1) + (size_t)i1 * C * PP;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      if (j < PP) afrag[s] = *reinterpret_cast<const frag_t *>(src + j * C + 4 * PER * s + PER * q);
      else afrag[s] = M::zero();
    }
  }

  float res[CORR_MAXLEV][KOUT];
  const int op_p = lane % PP, op_a = lane / PP;   // output ownership, see the union epilogue
#pragma unroll
  for (int lvl = 0; lvl < CORR_MAXLEV; lvl++) {
    if (lvl >= L) break;
    CTS(ct_l0);
    const int H2 = prm.H2[lvl], W2 = prm.W2[lvl];
    const T *f2 = reinterpret_cast<const T *>(prm.fmap2[lvl]) + (size_t)j2 * C * H2 * W2;
    if (lane < PP) {
      const float cdv = prm.cdiv[lvl];
      const float x = prm.coords[((size_t)e * 2 + 0) * PP + lane] / cdv;
      const float y = prm.coords[((size_t)e * 2 + 1) * PP + lane] / cdv;
      const float flx = floorf(x), fly = floorf(y);
      const int ox = ramp_f2i(flx), oy = ramp_f2i(fly);
      s_dx[lane] = x - flx;
      s_dy[lane] = y - fly;
      const bool live = ((long)ox - R < W2) && ((long)ox - R + D > 0) &&
                        ((long)oy - R < H2) && ((long)oy - R + D > 0);
      s_live[lane] = live ? 1 : 0;
      s_ox[lane] = live ? ox - R : 0;
      s_oy[lane] = live ? oy - R : 0;
    }
    __syncthreads();
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30), nlive = 0;
#pragma unroll
    for (int p = 0; p < PP; p++) {
      if (s_live[p]) {
        nlive++;
        minx = min(minx, s_ox[p]); maxx = max(maxx, s_ox[p]);
        miny = min(miny, s_oy[p]); maxy = max(maxy, s_oy[p]);
      }
    }
    const long bw = (long)maxx - minx + D, bh = (long)maxy - miny + D;
    const bool uni = (nlive > 0) && (bw * bh <= CORR_T);
    const int ngroups = (nlive == 0) ? 0 : (uni ? 1 : PP);
    if (nlive == 0) {
      for (int o = lane; o < NOUT; o += 64) {
        const int p = o % PP;
        const float dx = s_dx[p], dy = s_dy[p];
        float s = ((1 - dx) * (1 - dy)) * 0.0f;
        s = s + (dx * (1 - dy)) * 0.0f;
        s = s + ((1 - dx) * dy) * 0.0f;
        s = s + (dx * dy) * 0.0f;
        outs[o] = s;
      }
    }
    for (int g = 0; g < ngroups; g++) {
      if (!uni && !s_live[g]) {
        for (int ab = lane; ab < d * d; ab += 64) {
          const float dx = s_dx[g], dy = s_dy[g];
          float s = ((1 - dx) * (1 - dy)) * 0.0f;
          s = s + (dx * (1 - dy)) * 0.0f;
          s = s + ((1 - dx) * dy) * 0.0f;
          s = s + (dx * dy) * 0.0f;
          outs[ab * PP + g] = s;
        }
        continue;
      }
      const int gx0 = uni ? minx : s_ox[g], gy0 = uni ? miny : s_oy[g];
      const int gw = uni ? (int)bw : D, gh = uni ? (int)bh : D;
      const int Tn = gw * gh;                      // <= CORR_T = 128
      const int npg = (Tn + 15) / 16;
      const int inv_gw = (65536 + gw - 1) / gw;    // t / gw == (t * inv_gw) >> 16 while t * gw < 65536
#ifdef CORR_TRACE
      { CTS(ct_l1); ct_setup += ct_l1 - ct_l0; }
#endif
      for (int pg0 = 0; pg0 < npg; pg0 += PGB) {
        CTS(ct_b0);
        // all PGB x 4 sixteen-byte loads of the batch are issued before the first MFMA waits on
        // one: the address is always a valid pixel, out-of-window lanes are zeroed afterwards
        frag_t bfr[PGB][STEPS];
        bool inb[PGB];
#pragma unroll
        for (int u = 0; u < PGB; u++) {
          const int t = (pg0 + u) * 16 + j;
          const int ty = (t * inv_gw) >> 16, tx = t - ty * gw;
          const int px = gx0 + tx, py = gy0 + ty;
          inb[u] = (t < Tn) && px >= 0 && px < W2 && py >= 0 && py < H2;
          const int cy = inb[u] ? py : 0, cx = inb[u] ? px : 0;
          // load s, quarter q <-> channels [4 PER s + PER q, + PER) (fp16: = chunk 4 s + q of the [h][C/8][w][8] layout)
          const T *pp = CHUNKED ? f2 + (((size_t)cy * (C / 8) + q) * W2 + cx) * 8
                                : f2 + ((size_t)cy * W2 + cx) * C + PER * q;
          const size_t sstride = CHUNKED ? (size_t)4 * W2 * 8 : 4 * PER;
#pragma unroll
          for (int s = 0; s < STEPS; s++) bfr[u][s] = *reinterpret_cast<const frag_t *>(pp + s * sstride);
        }
#ifdef CORR_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CTS(ct_b1);
        ct_load += ct_b1 - ct_b0;
#endif
#pragma unroll
        for (int u = 0; u < PGB; u++) {
          const int t = (pg0 + u) * 16 + j;
          f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < STEPS; s++) acc = M::mma(afrag[s], bfr[u][s], acc);
          // D: rows 4q..4q+3 = patch pixels, column j = union pixel t; an out-of-map pixel
          // contributes zeros (the loads above fetched a valid stand-in pixel for it)
          if (t < Tn) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int p = 4 * q + r;
              if (p < PP) Cs[p * CORR_T + t] = inb[u] ? acc[r] : 0.0f;
            }
          }
        }
#ifdef CORR_TRACE
        { CTS(ct_b2); ct_mma += ct_b2 - ct_b1; }
#endif
      }
      __syncthreads();
      CTS(ct_c0);
      if (uni) {
        // lane = 9 a + p owns output row a of patch pixel p; its 7 outputs (b = 0..6) are
        // o = (7 b + a) 9 + p = lane + 63 b and need two 8-wide rows of Cs
        if (lane < 63) {
          float r0[D], r1[D];
          if (s_live[op_p]) {
            const float *row = &Cs[op_p * CORR_T + (s_oy[op_p] - gy0 + op_a) * gw + (s_ox[op_p] - gx0)];
#pragma unroll
            for (int b = 0; b < D; b++) { r0[b] = row[b]; r1[b] = row[gw + b]; }
          } else {
#pragma unroll
            for (int b = 0; b < D; b++) { r0[b] = 0.f; r1[b] = 0.f; }
          }
          const float dx = s_dx[op_p], dy = s_dy[op_p];
#pragma unroll
          for (int b = 0; b < d; b++) {
            float s = ((1 - dx) * (1 - dy)) * r0[b];
            s = s + (dx * (1 - dy)) * r0[b + 1];
            s = s + ((1 - dx) * dy) * r1[b];
            s = s + (dx * dy) * r1[b + 1];
            res[lvl][b] = s;
          }
        }
      } else {
        for (int ab = lane; ab < d * d; ab += 64) {
          const int b = ab / d, a = ab - b * d;
          const int wx = s_ox[g] - gx0 + b, wy = s_oy[g] - gy0 + a;
          const float *row = &Cs[g * CORR_T + wy * gw + wx];
          const float c00 = row[0], c01 = row[1], c10 = row[gw], c11 = row[gw + 1];
          const float dx = s_dx[g], dy = s_dy[g];
          float s = ((1 - dx) * (1 - dy)) * c00;
          s = s + (dx * (1 - dy)) * c01;
          s = s + ((1 - dx) * dy) * c10;
          s = s + (dx * dy) * c11;
          outs[ab * PP + g] = s;
        }
      }
      __syncthreads();
#ifdef CORR_TRACE
      { CTS(ct_c1); ct_blend += ct_c1 - ct_c0; }
#endif
    }
    if (!uni) {
      __syncthreads();
      if (lane < 63) {
#pragma unroll
        for (int k = 0; k < KOUT; k++) res[lvl][k] = outs[lane + 63 * k];
      }
    }
    __syncthreads();
  }
  // out[e][o][lvl]: with two levels a lane's pair is one 4-byte store, consecutive over lanes
  T *op = reinterpret_cast<T *>(prm.out) + (size_t)e * prm.row_elems;
  for (int q = NOUT * L + lane; q < prm.row_elems; q += 64) M::st(op + q, 0.0f);   // row padding
  if (lane < 63) {
    if (L == 2) {
#pragma unroll
      for (int k = 0; k < KOUT; k++)
        M::st2(op + 2 * (lane + 63 * k), res[0][k], res[1][k]);
    } else {
#pragma unroll
      for (int k = 0; k < KOUT; k++) M::st(op + lane + 63 * k, res[0][k]);
    }
  }
#ifdef CORR_TRACE
  {
    CTS(ct_end);
    CTACC(0, 1); CTACC(1, ct_end - ct_start); CTACC(2, ct_setup); CTACC(3, ct_load); CTACC(4, ct_mma); CTACC(5, ct_blend);
  }
#endif
}

template <typename T, bool CHUNKED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CorrMma<T>::WAVES, 8)))
    corr_mfma_kernel(const CorrParams prm) {
  const int e = corr_edge_of_block(prm);
  if (e < 0) return;
  corr_mfma_edge<T, CHUNKED>(prm, e);
}

// the factors corr_tile_kernel left out (prm.list, *prm.list_n of them): a fixed grid walks the list
template <typename T, bool CHUNKED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CorrMma<T>::WAVES, 8)))
    corr_mfma_list_kernel(const CorrParams prm) {
  const int n = *prm.list_n;
  for (int pos = blockIdx.x; pos < n; pos += gridDim.x) {
    corr_mfma_edge<T, CHUNKED>(prm, prm.list[pos]);
    __syncthreads();
  }
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = (atomicAdd(prm.list_n + 1, 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (s_last && threadIdx.x == 0) { prm.list_n[0] = 0; prm.list_n[1] = 0; }
}


// ------------------------------------------------------------- tile-resident level
// The coarse level of the pyramid is read ~170 times per pixel and step (2112 factors per target frame, each a
// 10 x 10 window of a 30 x 40 plane), every time through the vector L1 -- the limiter of corr_mfma_kernel
// (DESIGN.md section 5).  Here the factors of one level are binned by (target slot, window origin): a bin owns the
// origins of a CT_SX x CT_SY cell, all of its windows fall inside one CT_TW x CT_TH pixel tile, and a workgroup
// brings that tile into LDS ONCE (coalesced 16-byte pieces, zeros outside the plane) and hands the bin's factors
// to its waves.  A wave does exactly what corr_mfma_kernel does for the level -- the same union window, the same
// four v_mfma_f32_16x16x32_f16 steps in the same channel order, the same blend -- with ds_read_b128 as the source of
// the B operands: bit-identical values.  The FINE level of the same factor is computed by the same wave with the
// gather kernel's window loads, issued before the coarse level's LDS work and consumed after it: the vector L1's
// latency runs under the tile work, and both levels leave as the half2 pairs of the reference layout.  Factors that
// do not fit (a union window wider than CT_MAXB at the coarse level or of more than 128 pixels at either, nothing in
// a plane, bin list full) go to a fallback list for corr_mfma_list_kernel.
//   corr_bin_kernel : one thread per factor -> record in its bin (slot by atomic counter; the order inside a bin
//                     is irrelevant: factors are independent) or the fallback list
//   corr_tile_kernel: persistent workgroups, work items = (bin, chunk of CT_CHUNK factors), one contiguous run of
//                     items per XCD; the last workgroup to finish clears the counters for the next call (ticket)
#define CT_SX 10
#define CT_SY 10
#define CT_MAXB 12                       // widest union window a tile holds
#define CT_TW (CT_SX + CT_MAXB - 1)      // 21
#define CT_TH (CT_SY + CT_MAXB - 1)      // 21
#define CT_WAVES 8
#define CT_NT (64 * CT_WAVES)
#define CT_CHUNK 32                      // factors per work item
#define CT_CAP 384                       // records per bin
#define CT_REC 64                        // words per record
#define CT_MAXBINS 2560
#define CT_HDR 16                        // ints in front of the counters: [0] ticket, [1] fallback count, [2] its ticket
#define CT_UNITS (CT_TH * 16 * CT_TW)    // 16-byte pieces of a tile
#define CT_FILL ((CT_UNITS + CT_NT - 1) / CT_NT)
#define CT_FG 8                          // fine-level pixel groups of a factor (128 union pixels)
#define CT_DEADBINS 128                   // extra bins (no tile) for factors with nothing in the coarse plane
#if CT_CHUNK * CT_REC * 4 > CT_NT * 16
#error "one 16-byte piece per thread moves a work item's records"
#endif

// Record of a binned factor (written by corr_bin_kernel, 256 bytes, so that the tile kernel's loads do not depend
// on one another): [0] factor, [1] patch slot, [2] target slot; per level (fine at +4, coarse at +32): [0] window
// origin x | y << 16 (int16 each), [1] gw | gh << 8 | live mask << 16, [2..10] dx, [11..19] dy of the nine patch
// pixels, [20..22] their window offsets inside the union window, one byte each (x | y << 4)
#define CT_L0 4
#define CT_L1 32
struct CorrTile {
  const _Float16 *fmap1;       // [N1][9][128]
  const _Float16 *plane0;      // fine level   [mod_jj slots][H0][16][W0][8]
  const _Float16 *plane;       // coarse level [mod_jj slots][H2][16][W2][8]
  const float *coords;
  const int64_t *ii, *jj;
  _Float16 *out;
  int32_t *head;               // [CT_HDR], then count [nbins]
  int32_t *list;               // [nbins][CT_CAP][CT_REC]
  int32_t *fallback;           // [E] factors left to corr_mfma_list_kernel
  const int32_t *dyn;
  long mod_ii, mod_jj;
  float cdv0, cdv;
  int H0, W0, H2, W2, nbx, nby, nbins, nbins_all, E, row_elems;    // bins [nbins, nbins_all): no tile
};

// window geometry of the nine patch pixels at one level: the arithmetic of corr_mfma_kernel, line for line
struct CorrGeom {
  int minx, miny, bw, bh, nlive;
  unsigned lmask;
  int ox[9], oy[9];
  float dx[9], dy[9];
};
__device__ __forceinline__ void corr_geom(const float *__restrict__ coords, int e, float cdv, int H2, int W2, CorrGeom &g) {
  constexpr int PP = 9, R = 3, D = 8;
  int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
  g.nlive = 0; g.lmask = 0;
#pragma unroll
  for (int p = 0; p < PP; p++) {
    const float x = coords[((size_t)e * 2 + 0) * PP + p] / cdv;
    const float y = coords[((size_t)e * 2 + 1) * PP + p] / cdv;
    const float flx = floorf(x), fly = floorf(y);
    const int fx = ramp_f2i(flx), fy = ramp_f2i(fly);
    g.dx[p] = x - flx;
    g.dy[p] = y - fly;
    const bool live = ((long)fx - R < W2) && ((long)fx - R + D > 0) && ((long)fy - R < H2) && ((long)fy - R + D > 0);
    g.ox[p] = live ? fx - R : 0;
    g.oy[p] = live ? fy - R : 0;
    if (live) {
      g.nlive++;
      g.lmask |= 1u << p;
      minx = min(minx, g.ox[p]); maxx = max(maxx, g.ox[p]);
      miny = min(miny, g.oy[p]); maxy = max(maxy, g.oy[p]);
    }
  }
  g.minx = minx; g.miny = miny;
  g.bw = g.nlive ? maxx - minx + D : 0;
  g.bh = g.nlive ? maxy - miny + D : 0;
}
__device__ __forceinline__ void corr_geom_pack(const CorrGeom &g, int32_t *w) {    // 23 words
  w[0] = (g.minx & 0xffff) | (g.miny << 16);
  w[1] = g.bw | (g.bh << 8) | (int)(g.lmask << 16);
#pragma unroll
  for (int p = 0; p < 9; p++) { w[2 + p] = __float_as_int(g.dx[p]); w[11 + p] = __float_as_int(g.dy[p]); }
  w[20] = w[21] = w[22] = 0;
#pragma unroll
  for (int p = 0; p < 9; p++) {
    const int off = (g.lmask >> p) & 1 ? ((g.ox[p] - g.minx) | ((g.oy[p] - g.miny) << 4)) : 0;
    w[20 + p / 4] |= off << (8 * (p % 4));
  }
}

__global__ void __launch_bounds__(256) corr_bin_kernel(const CorrTile t) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int E = t.dyn ? t.dyn[RAMP_DYN_E] : t.E;
  if (e >= E) return;
  int32_t w[CT_REC];
  CorrGeom g;
  corr_geom(t.coords, e, t.cdv0, t.H0, t.W0, g);
  // a level with nothing in its plane costs the tile kernel nothing (no window, the blend's zero formula)
  bool ok = g.nlive == 0 || (long)g.bw * g.bh <= 16 * CT_FG;       // (the union path of corr_mfma_kernel: <= 128)
  corr_geom_pack(g, w + CT_L0);
  corr_geom(t.coords, e, t.cdv, t.H2, t.W2, g);
  ok = ok && (g.nlive == 0 || ((long)g.bw * g.bh <= CORR_T && g.bw <= CT_MAXB && g.bh <= CT_MAXB));
  corr_geom_pack(g, w + CT_L1);
  if (ok) {
    const long j2 = t.jj[e] % t.mod_jj;
    int bin;
    if (g.nlive > 0) {
      const int bx = (g.minx + 7) / CT_SX, by = (g.miny + 7) / CT_SY;          // live: origin >= -7
      bin = ((int)j2 * t.nby + by) * t.nbx + bx;
      ok = j2 >= 0 && bx < t.nbx && by < t.nby && bin < t.nbins;
    } else {
      bin = t.nbins + e % CT_DEADBINS;
      ok = j2 >= 0;
    }
    if (ok) {
      const int slot = atomicAdd(t.head + CT_HDR + bin, 1);
      ok = slot < CT_CAP;
      if (ok) {
        w[0] = e; w[1] = (int)(t.mod_ii > 0 ? t.ii[e] % t.mod_ii : t.ii[e]); w[2] = (int)j2; w[3] = 0;
#pragma unroll
        for (int c = CT_L0 + 23; c < CT_L1; c++) w[c] = 0;
#pragma unroll
        for (int c = CT_L1 + 23; c < CT_REC; c++) w[c] = 0;
        int4 *dst = reinterpret_cast<int4 *>(t.list + ((size_t)bin * CT_CAP + slot) * CT_REC);
#pragma unroll
        for (int c = 0; c < CT_REC / 4; c++) dst[c] = make_int4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
      }
    }
  }
  if (!ok) t.fallback[atomicAdd(t.head + 1, 1)] = e;
}

// blend of corr_mfma_kernel's union epilogue: lane = 9 a + p owns output row a of patch pixel p (lanes 0..62)
__device__ __forceinline__ void corr_tile_blend(const float *Cs, const int *lv, int gw, int op_p, int op_a, float (&res)[7]) {
  constexpr int D = 8, d = 7;
  const float pdx = __int_as_float(lv[2 + op_p]), pdy = __int_as_float(lv[11 + op_p]);
  const int off = (lv[20 + op_p / 4] >> (8 * (op_p % 4))) & 0xff;
  float r0[D], r1[D];
  if ((lv[1] >> (16 + op_p)) & 1) {
    const float *row = &Cs[op_p * CORR_T + ((off >> 4) + op_a) * gw + (off & 15)];
#pragma unroll
    for (int b = 0; b < D; b++) { r0[b] = row[b]; r1[b] = row[gw + b]; }
  } else {
#pragma unroll
    for (int b = 0; b < D; b++) { r0[b] = 0.f; r1[b] = 0.f; }
  }
#pragma unroll
  for (int b = 0; b < d; b++) {
    float s = ((1 - pdx) * (1 - pdy)) * r0[b];
    s = s + (pdx * (1 - pdy)) * r0[b + 1];
    s = s + ((1 - pdx) * pdy) * r1[b];
    s = s + (pdx * pdy) * r1[b + 1];
    res[b] = s;
  }
}

__global__ void __launch_bounds__(CT_NT) corr_tile_kernel(const CorrTile t) {
  constexpr int C = 128, PP = 9;
  extern __shared__ __attribute__((aligned(16))) unsigned char ct_smem[];
  uint4 *tile = reinterpret_cast<uint4 *>(ct_smem);                          // [CT_TH * CT_TW pixels][16 chunk slots]
  float *Cs_all = reinterpret_cast<float *>(ct_smem + (size_t)CT_UNITS * 16);  // [CT_WAVES][9][CORR_T]
  int *recs = reinterpret_cast<int *>(Cs_all + CT_WAVES * PP * CORR_T);       // [CT_CHUNK][CT_REC]
  int *pref = recs + CT_CHUNK * CT_REC;                                       // [nbins + 1] work items before bin b
  int *s_part = reinterpret_cast<int *>(Cs_all);                              // (prologue only)
  __shared__ int s_last;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int32_t *count = t.head + CT_HDR;

  // ---- work items: bin b has ceil(min(count, CAP) / CHUNK) of them
  const int per = (t.nbins_all + CT_NT - 1) / CT_NT;
  {
    int sum = 0;
    for (int k = 0; k < per; k++) {
      const int b = tid * per + k;
      if (b < t.nbins_all) sum += (min(count[b], CT_CAP) + CT_CHUNK - 1) / CT_CHUNK;
    }
    s_part[tid] = sum;
  }
  __syncthreads();
  if (wave == 0) {                                   // exclusive scan of the partial sums (CT_WAVES per lane)
    int v[CT_WAVES], run = 0;
#pragma unroll
    for (int k = 0; k < CT_WAVES; k++) { v[k] = s_part[lane * CT_WAVES + k]; run += v[k]; }
    int inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    int base = inc - run;
#pragma unroll
    for (int k = 0; k < CT_WAVES; k++) { s_part[lane * CT_WAVES + k] = base; base += v[k]; }
  }
  __syncthreads();
  {
    int base = s_part[tid];
    for (int k = 0; k < per; k++) {
      const int b = tid * per + k;
      if (b < t.nbins_all) {
        pref[b] = base;
        base += (min(count[b], CT_CAP) + CT_CHUNK - 1) / CT_CHUNK;
        if (b == t.nbins_all - 1) pref[t.nbins_all] = base;
      }
    }
  }
  __syncthreads();
  const int nitems = pref[t.nbins_all];
  // workgroup ids are dealt round-robin to the XCDs: each XCD takes one contiguous run of the items (target slot
  // major), so a frame's planes stay in one L2
  const int xcd = blockIdx.x % CORR_XCDS, nx = (gridDim.x - xcd + CORR_XCDS - 1) / CORR_XCDS;
  const int run = (nitems + CORR_XCDS - 1) / CORR_XCDS;
  const int item_end = min(nitems, (xcd + 1) * run);

  float *Cs = Cs_all + wave * PP * CORR_T;
  const int op_p = lane % PP, op_a = lane / PP;
  for (int item = xcd * run + blockIdx.x / CORR_XCDS; item < item_end; item += nx) {
    int lo = 0, hi = t.nbins_all - 1;                   // last bin with pref[b] <= item
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pref[mid] <= item) lo = mid; else hi = mid - 1;
    }
    const int bin = lo, chunk = item - pref[bin];
    const int bx = bin % t.nbx, by = (bin / t.nbx) % t.nby, slot2 = bin / (t.nbx * t.nby);
    const int X0 = bx * CT_SX - 7, Y0 = by * CT_SY - 7;
    const int n = min(min(count[bin], CT_CAP) - chunk * CT_CHUNK, CT_CHUNK);
    const bool with_tile = bin < t.nbins_all - CT_DEADBINS;
#ifdef CT_NO_FILL
    const int e0 = t.E;
#endif
    __syncthreads();                                // the previous item's waves are done with the tile
    {
      const _Float16 *f2 = t.plane + (size_t)slot2 * C * t.H2 * t.W2;
      uint4 v[CT_FILL], vrec = make_uint4(0, 0, 0, 0);
      if (tid < n * (CT_REC / 4))
        vrec = reinterpret_cast<const uint4 *>(t.list + ((size_t)bin * CT_CAP + chunk * CT_CHUNK) * CT_REC)[tid];
#pragma unroll
      for (int k = 0; k < CT_FILL; k++) {
        int u = tid + k * CT_NT;
        asm volatile("" : "+v"(u));                  // (keeps the index arithmetic out of the registers between items)
        const int py = u / (16 * CT_TW), rem = u - py * (16 * CT_TW), c = rem / CT_TW, px = rem - c * CT_TW;
        const int gy = Y0 + py, gx = X0 + px;
        v[k] = make_uint4(0, 0, 0, 0);
#ifdef CT_NO_FILL
        if (with_tile && e0 < 0 && u < CT_UNITS && gy >= 0 && gy < t.H2 && gx >= 0 && gx < t.W2)
#else
        if (with_tile && u < CT_UNITS && gy >= 0 && gy < t.H2 && gx >= 0 && gx < t.W2)
#endif
          v[k] = *reinterpret_cast<const uint4 *>(f2 + (((size_t)gy * (C / 8) + c) * t.W2 + gx) * 8);
      }
      if (tid < CT_CHUNK * (CT_REC / 4)) reinterpret_cast<uint4 *>(recs)[tid] = vrec;
#pragma unroll
      for (int k = 0; k < CT_FILL; k++) {
        int u = tid + k * CT_NT;
        asm volatile("" : "+v"(u));
        const int py = u / (16 * CT_TW), rem = u - py * (16 * CT_TW), c = rem / CT_TW, px = rem - c * CT_TW;
        const int pix = py * CT_TW + px;
        if (with_tile && u < CT_UNITS) tile[pix * 16 + ((c + pix) & 15)] = v[k];
      }
    }
    __syncthreads();

    for (int k = wave; k < n; k += CT_WAVES) {
      const int *rec = recs + k * CT_REC;
      const int e = rec[0];
      // the patch features and every fine-level window load go out together; the coarse level's products wait for
      // the former only
      f16x8_t afrag[4];
      {
        const _Float16 *src = t.fmap1 + (size_t)rec[1] * C * PP;
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if (j < PP) afrag[s] = *reinterpret_cast<const f16x8_t *>(src + j * C + 32 * s + 8 * q);
          else afrag[s] = (f16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      // ---- fine level: every window load of the factor goes out now (the vector L1's latency is spent on the coarse
      // level's LDS work below); the address is always a valid pixel, out-of-plane lanes are zeroed afterwards
      const int *l0 = rec + CT_L0;
      const int fx0 = (int)(short)(l0[0] & 0xffff), fy0 = l0[0] >> 16, fw = l0[1] & 0xff, fh = (l0[1] >> 8) & 0xff;
      const int fTn = fw * fh, fnpg = (fTn + 15) / 16, finv = (65536 + fw - 1) / (fw > 0 ? fw : 1);   // (fTn == 0: nothing in the plane)
      f16x8_t bF[CT_FG][4];
      unsigned inb = 0;
      {
        const _Float16 *f0 = t.plane0 + (size_t)rec[2] * C * t.H0 * t.W0;
#pragma unroll
        for (int u = 0; u < CT_FG; u++) {
#ifdef CT_NO_FINE
          if (u < fnpg && e < 0) {
#else
          if (u < fnpg) {                             // (wave uniform)
#endif
            const int tt = u * 16 + j;
            const int ty = (tt * finv) >> 16, tx = tt - ty * fw;
            const int px = fx0 + tx, py = fy0 + ty;
            const bool in = (tt < fTn) && px >= 0 && px < t.W0 && py >= 0 && py < t.H0;
            inb |= (unsigned)in << u;
            const int cy = in ? py : 0, cx = in ? px : 0;
            const _Float16 *pp = f0 + (((size_t)cy * (C / 8) + q) * t.W0 + cx) * 8;
            const size_t sstride = (size_t)4 * t.W0 * 8;
#pragma unroll
            for (int s = 0; s < 4; s++) bF[u][s] = *reinterpret_cast<const f16x8_t *>(pp + s * sstride);
          }
        }
      }
      // ---- coarse level from the tile
      float res1[7], res0[7];
      {
        const int *l1 = rec + CT_L1;
        const int gx0 = (int)(short)(l1[0] & 0xffff), gy0 = l1[0] >> 16, gw = l1[1] & 0xff, gh = (l1[1] >> 8) & 0xff;
        const int Tn = gw * gh, npg = (Tn + 15) / 16, inv_gw = (65536 + gw - 1) / (gw > 0 ? gw : 1);
        const int lx0 = gx0 - X0, ly0 = gy0 - Y0;     // the bin guarantees [lx0, lx0 + gw) x [ly0, ly0 + gh) inside the tile
        auto load_b = [&](int pg, f16x8_t (&b)[4]) {
          const int tt = pg * 16 + j;
          const int ty = (tt * inv_gw) >> 16, tx = tt - ty * gw;
          const int pix = tt < Tn ? (ly0 + ty) * CT_TW + lx0 + tx : 0;
#pragma unroll
          for (int s = 0; s < 4; s++) b[s] = *reinterpret_cast<const f16x8_t *>(&tile[pix * 16 + ((4 * s + q + pix) & 15)]);
        };
#ifdef CT_NO_COARSE
        for (int pg = 0; pg < (e < 0 ? npg : 0); pg++) {
#else
        for (int pg = 0; pg < npg; pg++) {
#endif
          f16x8_t bc[4];
          load_b(pg, bc);
          const int tt = pg * 16 + j;
          f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[s], bc[s], acc, 0, 0, 0);
          if (tt < Tn) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int p = 4 * q + r;
              if (p < PP) Cs[p * CORR_T + tt] = acc[r];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef CT_NO_BLEND
        if (lane < 63 && e < 0) corr_tile_blend(Cs, l1, gw, op_p, op_a, res1);
#else
        if (lane < 63) corr_tile_blend(Cs, l1, gw, op_p, op_a, res1);
#endif
        __builtin_amdgcn_wave_barrier();
      }
      // ---- fine level: products of the loads issued above
#pragma unroll
      for (int u = 0; u < CT_FG; u++) {
#ifdef CT_NO_FINE
        if (u < fnpg && e < 0) {
#else
        if (u < fnpg) {
#endif
          const int tt = u * 16 + j;
          f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[s], bF[u][s], acc, 0, 0, 0);
          if (tt < fTn) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int p = 4 * q + r;
              if (p < PP) Cs[p * CORR_T + tt] = ((inb >> u) & 1) ? acc[r] : 0.0f;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
#ifdef CT_NO_BLEND
      if (lane < 63 && e < 0) {
#else
      if (lane < 63) {
#endif
        corr_tile_blend(Cs, l0, fw, op_p, op_a, res0);
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        _Float16 *op = t.out + (size_t)e * t.row_elems;
#pragma unroll
        for (int b = 0; b < 7; b++)
          *reinterpret_cast<h2v *>(op + 2 * (lane + 63 * b)) = (h2v){(_Float16)res0[b], (_Float16)res1[b]};
      }
      {
        _Float16 *op = t.out + (size_t)e * t.row_elems;
        for (int z = 882 + lane; z < t.row_elems; z += 64) op[z] = (_Float16)0.0f;   // row padding
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- the last workgroup out clears the counters (every workgroup is past its last read of them)
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    s_last = (atomicAdd(t.head, 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (s_last) {
    for (int b = tid; b < t.nbins_all; b += CT_NT) t.head[CT_HDR + b] = 0;
    if (tid == 0) t.head[0] = 0;
  }
}

// ------------------------------------------------------------- pyramid pack
// One frame's fp16 NHWC feature map [H][W][128] -> the two correlation levels in the chunked
// layout: level 1 = the map itself as [H][16][W][8]; level 4 = its 4x4 average (fp32 sum / 16,
// rounded once; Ramp_vo.py:378-381's avg_pool2d) as [H/4][16][W/4][8].  A workgroup owns a 4-row x
// 16-pixel tile; thread (xl, c8) moves 16 bytes per row through an LDS transpose so that both the
// reads (pixel-major) and the writes (chunk-major) are contiguous.
__global__ void __launch_bounds__(256) pyramid_pack_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out1,
                                                           uint4 *__restrict__ out4, int H, int W) {
  __shared__ uint4 tile[16][17];
  const int t = threadIdx.x;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 4;
  const int xl = t >> 4, c8 = t & 15;      // read role
  const int wc = t >> 4, wx = t & 15;      // write role: chunk wc, pixel wx
  float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < 4; r++) {
    const int y = y0 + r;
    const uint4 v = in[((size_t)y * W + x0 + xl) * 16 + c8];
    const __half2 *h = reinterpret_cast<const __half2 *>(&v);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float2 f = __half22float2(h[k]);
      sum[2 * k] += f.x;
      sum[2 * k + 1] += f.y;
    }
    __syncthreads();
    tile[c8][xl] = v;
    __syncthreads();
    out1[((size_t)y * 16 + wc) * W + x0 + wx] = tile[wc][wx];
  }
  // 4 neighbouring pixels = lanes t, t^16, t^32, t^48 of one wave
#pragma unroll
  for (int k = 0; k < 8; k++) {
    sum[k] += __shfl_xor(sum[k], 16);
    sum[k] += __shfl_xor(sum[k], 32);
  }
  if ((xl & 3) == 0) {
    uint4 o;
    __half2 *h = reinterpret_cast<__half2 *>(&o);
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = __floats2half2_rn(sum[2 * k] * 0.0625f, sum[2 * k + 1] * 0.0625f);
    out4[((size_t)blockIdx.y * 16 + c8) * (W / 4) + (x0 + xl) / 4] = o;
  }
}

extern "C" {

#ifdef CORR_TRACE
int ramp_debug_corr_trace(unsigned long long *host, int n_waves) {     // host [n_waves][8]
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_corr_trace), (size_t)n_waves * 8 * 8) == hipSuccess ? 0 : -1;
}
#endif
int ramp_pyramid_pack(const void *fmap, void *level1, void *level4, int H, int W, int C, int dtype,
                      void *stream) {
  if (!fmap || !level1 || !level4 || H <= 0 || W <= 0) return RAMP_EINVAL;
  if (C != 128 || dtype != RAMP_F16 || (W % 16) || (H % 4)) return RAMP_EUNSUPPORTED;
  hipLaunchKernelGGL(pyramid_pack_kernel, dim3(W / 16, H / 4), dim3(256), 0, (hipStream_t)stream,
                     (const uint4 *)fmap, (uint4 *)level1, (uint4 *)level4, H, W);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_patchify_fwd(const void *net, const float *coords, void *out, int n, int C, int H,
                      int W, int M, int radius, int bilinear, int dtype, int layout,
                      int out_layout, void *stream) {
  if (n < 0 || M < 0 || C <= 0 || H <= 0 || W <= 0 || radius < 0) return RAMP_EINVAL;
  if (n * M == 0) return RAMP_OK;
  if (!net || !coords || !out) return RAMP_EINVAL;
  if (layout != RAMP_NCHW && layout != RAMP_NHWC) return RAMP_EINVAL;
  const int d = bilinear ? 2 * radius + 1 : 2 * radius + 2;
  const int total = C * d * d;
  const int threads = total >= 256 ? 256 : (total > 64 ? 128 : 64);
  if (dtype == RAMP_F32)
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(n * M), dim3(threads), 0, (hipStream_t)stream,
                       (const float *)net, coords, (float *)out, C, H, W, M, radius, bilinear,
                       layout, out_layout);
  else if (dtype == RAMP_F16)
    hipLaunchKernelGGL(patchify_kernel<__half>, dim3(n * M), dim3(threads), 0,
                       (hipStream_t)stream, (const __half *)net, coords, (__half *)out, C, H, W,
                       M, radius, bilinear, layout, out_layout);
  else
    return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_frame_gather(const void *fmap, const void *imap, const float *image, const float *coords, void *gmap,
                      void *imap_p, float *patches, float *clr, unsigned char *colors, int M, int h, int w, int H,
                      int W, int CF, int CI, int dtype, void *stream) {
  if (M < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || CF <= 0 || CI <= 0) return RAMP_EINVAL;
  if (M == 0) return RAMP_OK;
  if (!fmap || !imap || !image || !coords || !gmap || !imap_p || !patches || !clr || !colors) return RAMP_EINVAL;
  FrameGather p;
  p.fmap = fmap; p.imap = imap; p.image = image; p.coords = coords; p.gmap = gmap; p.imap_p = imap_p;
  p.patches = patches; p.clr = clr; p.colors = colors;
  p.h = h; p.w = w; p.H = H; p.W = W; p.CF = CF; p.CI = CI;
  if (dtype == RAMP_F32)
    hipLaunchKernelGGL(frame_gather_kernel<float>, dim3(M), dim3(256), 0, (hipStream_t)stream, p);
  else if (dtype == RAMP_F16)
    hipLaunchKernelGGL(frame_gather_kernel<__half>, dim3(M), dim3(256), 0, (hipStream_t)stream, p);
  else
    return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

static void corr_tile_geometry(int H2, int W2, long slots, int &nbx, int &nby, long &nbins) {
  nbx = (W2 + 7 + CT_SX - 1) / CT_SX;      // window origins of live pixels: -7 .. W2 - 1
  nby = (H2 + 7 + CT_SY - 1) / CT_SY;
  nbins = slots * nbx * nby;
}
static size_t corr_tile_head_bytes(long nbins) { return (((size_t)(CT_HDR + nbins + CT_DEADBINS) * 4 + 255) / 256) * 256; }

size_t ramp_corr_tile_workspace_bytes(int E_cap, int slots, int H2, int W2) {
  int nbx, nby; long nbins;
  if (E_cap <= 0 || slots <= 0 || H2 <= 0 || W2 <= 0) return 0;
  corr_tile_geometry(H2, W2, slots, nbx, nby, nbins);
  if (nbins + CT_DEADBINS > CT_MAXBINS) return 0;             // no tile path at this size
  return corr_tile_head_bytes(nbins) + (size_t)(nbins + CT_DEADBINS) * CT_CAP * CT_REC * 4 + (size_t)E_cap * 4;
}

int ramp_i_corr_fwd(const void *fmap1, const ramp_corr_level *levels, int nlevels,
                    const float *coords, const int64_t *ii, const int64_t *jj,
                    const int32_t *order, void *out, int out_row_elems, long mod_ii, long mod_jj, int E,
                    int N1, int N2, int C, int P, int radius, int dtype, int layout, const int32_t *dyn,
                    void *tile_ws, size_t tile_ws_bytes, void *stream) {
  if (E < 0 || nlevels < 1 || nlevels > CORR_MAXLEV || !levels) return RAMP_EINVAL;
  if (C != 128 || P != 3 || radius != 3) return RAMP_EUNSUPPORTED;
  if (E == 0) return RAMP_OK;
  if (!fmap1 || !coords || !ii || !jj || !out) return RAMP_EINVAL;
  CorrParams prm;
  prm.fmap1 = fmap1;
  for (int l = 0; l < CORR_MAXLEV; l++) {
    const int s = l < nlevels ? l : 0;
    if (!levels[s].fmap || levels[s].H2 <= 0 || levels[s].W2 <= 0) return RAMP_EINVAL;
    prm.fmap2[l] = levels[s].fmap;
    prm.H2[l] = levels[s].H2;
    prm.W2[l] = levels[s].W2;
    prm.cdiv[l] = levels[s].coord_div;
  }
  prm.nlevels = nlevels;
  prm.coords = coords;
  prm.ii = ii;
  prm.jj = jj;
  prm.out = out;
  prm.E = E;
  prm.N1 = N1;
  prm.N2 = N2;
  prm.order = order;
  prm.mod_ii = mod_ii;
  prm.mod_jj = mod_jj;
  prm.row_elems = out_row_elems > 0 ? out_row_elems : 49 * 9 * nlevels;
  if (prm.row_elems < 49 * 9 * nlevels || (nlevels == 2 && (prm.row_elems & 1))) return RAMP_EINVAL;   // half2 stores
  prm.chunk = (E + CORR_XCDS - 1) / CORR_XCDS;
  prm.dyn = dyn;
  prm.list = nullptr;
  prm.list_n = nullptr;
  const dim3 grid(prm.chunk * CORR_XCDS);
  hipStream_t st = (hipStream_t)stream;
  // both levels of the regular factors in corr_tile_kernel (fp16 chunked pyramid, ring slots known), the rest from a list
  static int tile_on = -1;                          // RAMP_CORR_TILE=0: everything in corr_mfma_kernel (A/B runs)
  if (tile_on < 0) { const char *ev = getenv("RAMP_CORR_TILE"); tile_on = ev ? atoi(ev) : 1; }
  if (tile_ws && tile_on && nlevels == 2 && dtype == RAMP_F16 && layout == RAMP_NHWC8 && mod_jj > 0) {
    int nbx, nby; long nbins;
    corr_tile_geometry(levels[1].H2, levels[1].W2, mod_jj, nbx, nby, nbins);
    const size_t need = nbins + CT_DEADBINS <= CT_MAXBINS ? corr_tile_head_bytes(nbins) + (size_t)(nbins + CT_DEADBINS) * CT_CAP * CT_REC * 4 + (size_t)E * 4 : 0;
    if (!need || tile_ws_bytes < need) return RAMP_EWORKSPACE;
    CorrTile t;
    t.fmap1 = (const _Float16 *)fmap1; t.plane0 = (const _Float16 *)levels[0].fmap; t.plane = (const _Float16 *)levels[1].fmap;
    t.coords = coords; t.ii = ii; t.jj = jj; t.out = (_Float16 *)out;
    t.head = (int32_t *)tile_ws;
    t.list = (int32_t *)((char *)tile_ws + corr_tile_head_bytes(nbins));
    t.fallback = t.list + (size_t)(nbins + CT_DEADBINS) * CT_CAP * CT_REC;
    t.dyn = dyn; t.mod_ii = mod_ii; t.mod_jj = mod_jj; t.cdv0 = levels[0].coord_div; t.cdv = levels[1].coord_div;
    t.H0 = levels[0].H2; t.W0 = levels[0].W2; t.H2 = levels[1].H2; t.W2 = levels[1].W2;
    t.nbx = nbx; t.nby = nby; t.nbins = (int)nbins; t.nbins_all = (int)nbins + CT_DEADBINS; t.E = E; t.row_elems = prm.row_elems;
    const size_t lds = (size_t)CT_UNITS * 16 + (size_t)CT_WAVES * 9 * CORR_T * 4 + (size_t)CT_CHUNK * CT_REC * 4 + ((size_t)nbins + CT_DEADBINS + 1) * 4;
    if (lds > 160 * 1024 - 64) return RAMP_EUNSUPPORTED;
    static int n_wg = 0;
    if (!n_wg) {
      if (hipFuncSetAttribute((const void *)corr_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) !=
          hipSuccess)
        return RAMP_ELAUNCH;
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return RAMP_ELAUNCH;
      n_wg = cus;
    }
    hipLaunchKernelGGL(corr_bin_kernel, dim3(ramp_cdiv(E, 256)), dim3(256), 0, st, t);
    hipLaunchKernelGGL(corr_tile_kernel, dim3(n_wg), dim3(CT_NT), lds, st, t);
    prm.list = t.fallback;
    prm.list_n = t.head + 1;
    hipLaunchKernelGGL((corr_mfma_list_kernel<_Float16, true>), dim3(2048), dim3(64), 0, st, prm);
    RAMP_CHECK_LAUNCH();
    return RAMP_OK;
  }
  const bool fast32 = (dtype & RAMP_CORR_MFMA32) != 0;
  dtype &= ~RAMP_CORR_MFMA32;
  if (dtype == RAMP_F32 && layout == RAMP_NHWC && fast32)
    hipLaunchKernelGGL((corr_mfma_kernel<float, false>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F32 && layout == RAMP_NHWC)
    hipLaunchKernelGGL((corr_kernel<float, RAMP_NHWC>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F32 && layout == RAMP_NCHW)
    hipLaunchKernelGGL((corr_kernel<float, RAMP_NCHW>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F16 && layout == RAMP_NHWC)
    hipLaunchKernelGGL((corr_mfma_kernel<_Float16, false>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F16 && layout == RAMP_NHWC8)
    hipLaunchKernelGGL((corr_mfma_kernel<_Float16, true>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F16 && layout == RAMP_NCHW)
    hipLaunchKernelGGL((corr_kernel<__half, RAMP_NCHW>), grid, dim3(64), 0, st, prm);
  else
    return RAMP_EINVAL;
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}

int ramp_corr_fwd_ordered(const void *fmap1, const ramp_corr_level *levels, int nlevels,
                          const float *coords, const int64_t *ii, const int64_t *jj,
                          const int32_t *order, void *out, int out_row_elems, long mod_ii, long mod_jj, int E,
                          int N1, int N2, int C, int P, int radius, int dtype, int layout, void *stream) {
  return ramp_i_corr_fwd(fmap1, levels, nlevels, coords, ii, jj, order, out, out_row_elems, mod_ii, mod_jj, E, N1, N2,
                         C, P, radius, dtype, layout, nullptr, nullptr, 0, stream);
}

int ramp_corr_fwd_tiled(const void *fmap1, const ramp_corr_level *levels, int nlevels,
                        const float *coords, const int64_t *ii, const int64_t *jj,
                        const int32_t *order, void *out, int out_row_elems, long mod_ii, long mod_jj, int E,
                        int N1, int N2, int C, int P, int radius, int dtype, int layout, void *ws, size_t ws_bytes,
                        void *stream) {
  if (!ws) return RAMP_EINVAL;
  return ramp_i_corr_fwd(fmap1, levels, nlevels, coords, ii, jj, order, out, out_row_elems, mod_ii, mod_jj, E, N1, N2,
                         C, P, radius, dtype, layout, nullptr, ws, ws_bytes, stream);
}

int ramp_corr_fwd(const void *fmap1, const ramp_corr_level *levels, int nlevels,
                  const float *coords, const int64_t *ii, const int64_t *jj, void *out, int E,
                  int N1, int N2, int C, int P, int radius, int dtype, int layout,
                  void *stream) {
  return ramp_corr_fwd_ordered(fmap1, levels, nlevels, coords, ii, jj, nullptr, out, 0, 0, 0, E, N1, N2, C,
                               P, radius, dtype, layout, stream);
}

}  // extern "C"
