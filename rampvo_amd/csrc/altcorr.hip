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
#ifndef CORR_KPLANE
#define CORR_KPLANE 32  // channels per plane of the packed target maps: [h][128 / KPLANE][w][KPLANE] (8: round 2/3's layout)
#endif
#ifndef CORR_PGB
#define CORR_PGB 4    // pixel groups (of 16) whose loads are in flight together, MFMA kernel
#define CORR_WAVES 4  // waves per SIMD the MFMA kernel is register-budgeted for
#endif
#define CORR_T 128  // union pixels handled per group (2 per lane), VALU kernel
#ifndef CORR_TM
#define CORR_TM 192 // ... of the MFMA kernel: twelve 16-pixel products.  In the bench's steady state a tenth of the live
#endif              // factors have a 12 x 11 .. 13 x 13 union window at the fine level (tools/corr_window_stats.py); at 128 they
                    // took the nine-separate-windows path (36 products instead of 9-11)

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
  // optional fused reprojection (MFMA kernel, P = 3): the wave computes its edge's coordinates itself -- pops.transform,
  // csrc/lie.hip::transform_kernel's arithmetic -- from poses / patches / intrinsics and the source frame of the edge, and
  // writes them to `coords` (the heads and BA's targets read the patch centre): one launch and one 72-byte round trip
  // per edge less in the serial part of the tracked frame
  const float *tf_poses, *tf_patches, *tf_intr;
  const int64_t *tf_src;  // [E] source frame of each edge (ii of the tracker's graph)
  const int32_t *slot0;   // optional [mod_jj]: physical slot of ring row r of the LEVEL-0 target maps (ramp_track.fmap1_slot)
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
    const T *f2 = reinterpret_cast<const T *>(prm.fmap2[lvl]) + (size_t)((lvl == 0 && prm.slot0) ? (long)prm.slot0[j2] : j2) * C * H2 * W2;
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
// Target maps come either as plain NHWC or (CHUNKED) as [H][C/32][W][32] -- one plane per MFMA K step: the 64 lanes of a
// load (16 neighbouring window pixels x the four 8-channel quarters of the step) read one or two contiguous runs of up
// to 640 bytes instead of 16 pieces 256 B apart.  (Rounds 2-3: [H][C/8][W][8], four 160-byte runs per quarter-wave;
// CORR_KPLANE=8 builds it.)  The vector L1 looks up one line per
// cycle, and with NHWC those lookups (64 per load instruction) were what the kernel waited on.
// element-type traits of the MFMA correlation kernel below: fp16 -> v_mfma_f32_16x16x32_f16 (4 steps of 32 channels,
// lane (q, .) supplies 8 channels per step); fp32 -> v_mfma_f32_16x16x4_f32 (exact fp32 products; the channel axis is
// permuted so that ONE 16-byte load per lane feeds 4 MFMA steps on both operands: lane (q, .) of load g holds
// channels 16 g + 4 q + t, t = 0..3, and step (g, t) contracts them: 32 MFMAs per group of 16 window pixels).
// The fp32 variant is opt-in (dtype | RAMP_CORR_MFMA32): 2.1x faster than corr_kernel<float> (445 vs 940 us at E = 40k)
// but its accumulation order is the MFMA's, not the reference kernel's channel-ordered fmaf chain that
// corr_kernel<float> reproduces bit for bit -- so the exact-parity path stays the default.
#ifndef CORR_NT
#define CORR_NT 0     // (measured with csrc/update_mlp.hip's UPD_NT: the hint makes the step slower)
#endif
template <typename T> struct CorrMma;
template <> struct CorrMma<_Float16> {
  typedef f16x8_t frag;
  typedef frag afrag_t;
  typedef frag bfrag_t;
  typedef f32x4_t acc_t;
  typedef _Float16 plane_t;                           // element of the target planes ...
  typedef _Float16 out_t;                             // ... and of the output rows
  static constexpr int STEPS = 4, PER = 8, PGB = CORR_PGB, WAVES = CORR_WAVES, FE = 1;
  static __device__ __forceinline__ frag zero() { return (frag){0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ afrag_t load_a(const void *fmap1, size_t off) {
    return *reinterpret_cast<const frag *>(reinterpret_cast<const _Float16 *>(fmap1) + off);
  }
  static __device__ __forceinline__ bfrag_t load_b(const plane_t *p, size_t) { return *reinterpret_cast<const frag *>(p); }
  static __device__ __forceinline__ acc_t acc_zero() { return (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ float val(const acc_t &acc, int r) { return acc[r]; }
  static __device__ __forceinline__ f32x4_t mma(const frag &a, const frag &b, f32x4_t acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void st(_Float16 *p, float v) { *p = (_Float16)v; }
  static __device__ __forceinline__ void st2(_Float16 *p, float a, float b) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    // the [E, 896] volume is written once and read once (by the correlation MLP): streamed (CORR_NT=0: plain stores)
#if CORR_NT
    __builtin_nontemporal_store((h2v){(_Float16)a, (_Float16)b}, reinterpret_cast<h2v *>(p));
#else
    *reinterpret_cast<h2v *>(p) = (h2v){(_Float16)a, (_Float16)b};
#endif
  }
};
template <> struct CorrMma<float> {
  typedef f32x4_t frag;
  typedef frag afrag_t;
  typedef frag bfrag_t;
  typedef f32x4_t acc_t;
  typedef float plane_t;
  typedef float out_t;
  static constexpr int STEPS = 8, PER = 4, PGB = 2, WAVES = 3, FE = 1;
  static __device__ __forceinline__ frag zero() { return (frag){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ afrag_t load_a(const void *fmap1, size_t off) {
    return *reinterpret_cast<const frag *>(reinterpret_cast<const float *>(fmap1) + off);
  }
  static __device__ __forceinline__ bfrag_t load_b(const plane_t *p, size_t) { return *reinterpret_cast<const frag *>(p); }
  static __device__ __forceinline__ acc_t acc_zero() { return (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ float val(const acc_t &acc, int r) { return acc[r]; }
  static __device__ __forceinline__ f32x4_t mma(const frag &a, const frag &b, f32x4_t acc) {
#pragma unroll
    for (int t4 = 0; t4 < 4; t4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t4], b[t4], acc, 0, 0, 0);
    return acc;
  }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
  static __device__ __forceinline__ void st2(float *p, float a, float b) { *reinterpret_cast<float2 *>(p) = make_float2(a, b); }
};

// fp32 features on the f16 matrix cores, "x2" (round 6, dtype RAMP_F32 | RAMP_CORR_X2, chunked planes only): gfx950 multiplies
// fp32 operands at the vector rate (v_mfma_f32_16x16x4_f32: the fp32 volume's 37.6 GFLOP at E = 41k are >= 240 us of the
// matrix pipe), fp16 operands 16x faster.  Every feature is kept as two fp16 numbers,
//     x = xh + xl 2^-11,   xh = fp16(x),   xl = fp16((x - xh) 2^11)      (22 significant bits; corr_split2 below),
// and a dot product is three f16 MFMA products into two fp32 accumulators,
//     acc0 += ah bh;   acc1 += ah bl + al bh;   result = acc0 + 2^-11 acc1      (dropped: al bl, 2^-22 of |a||b|),
// the scheme of csrc/update_x3.hip's Linear layers.  The target planes hold the parts as [h][4][2][w][32] fp16 -- per row
// and K step a plane of high parts, then one of low parts, each the fp16 kernel's [w][32] run (ramp_pyramid_pack with
// RAMP_CORR_X2; 512 bytes per pixel like the fp32 planes they replace, so the tracker's ring buffers and copies do not
// change) -- and the
// patch features stay fp32 rows that the wave splits once while loading its A fragments.  The kernel is the fp16 kernel
// with twice the window bytes: bound by the vector L1 like it, not by the matrix pipe.
static __device__ __forceinline__ void corr_split2(const float x, _Float16 &h, _Float16 &l) {
  // (below the fp16 normal range the value goes into the scaled low part whole: no subnormal operand is relied on)
  const float xh = fabsf(x) < 6.103515625e-5f ? 0.0f : (float)(_Float16)x;
  h = (_Float16)xh;
  l = (_Float16)((x - xh) * 2048.0f);
}
#ifndef CORR_X2_PGB
#define CORR_X2_PGB 2     // pixel groups in flight (16 sixteen-byte loads per lane, as the fp16 kernel's four)
#define CORR_X2_WAVES 3   // waves per SIMD of the register budget (132 VGPRs)
#endif
struct CorrX2 {};
template <> struct CorrMma<CorrX2> {
  typedef f16x8_t frag;
  struct afrag_t { f16x8_t h, l; };
  typedef afrag_t bfrag_t;
  struct acc_t { f32x4_t a0, a1; };
  typedef _Float16 plane_t;
  typedef float out_t;
  static constexpr int STEPS = 4, PER = 8, PGB = CORR_X2_PGB, WAVES = CORR_X2_WAVES, FE = 2;
  static __device__ __forceinline__ afrag_t zero() { return (afrag_t){(frag){0, 0, 0, 0, 0, 0, 0, 0}, (frag){0, 0, 0, 0, 0, 0, 0, 0}}; }
  static __device__ __forceinline__ afrag_t load_a(const void *fmap1, size_t off) {
    const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(fmap1) + off);
    const float4 u = p[0], v = p[1];
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    afrag_t a;
#pragma unroll
    for (int c = 0; c < 8; c++) { _Float16 h, l; corr_split2(x[c], h, l); a.h[c] = h; a.l[c] = l; }
    return a;
  }
  static __device__ __forceinline__ bfrag_t load_b(const plane_t *p, size_t part) {       // part: elements from a high to its low part
    return (bfrag_t){*reinterpret_cast<const frag *>(p), *reinterpret_cast<const frag *>(p + part)};
  }
  static __device__ __forceinline__ acc_t acc_zero() { return (acc_t){(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}}; }
  static __device__ __forceinline__ float val(const acc_t &acc, int r) { return acc.a0[r] + acc.a1[r] * 0x1p-11f; }
  static __device__ __forceinline__ acc_t mma(const afrag_t &a, const bfrag_t &b, acc_t acc) {
    acc.a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, acc.a0, 0, 0, 0);
    acc.a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.l, acc.a1, 0, 0, 0);
    acc.a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, b.h, acc.a1, 0, 0, 0);
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
// One factor's correlation rows on one wave: res[lvl][k] = output lane + 63 k of level lvl (lanes 0..62; 441 = 63 x 7
// values per level, index order [x-off(7)][y-off(7)][py(3)][px(3)] as the reference's permute leaves them).  Cs / outs:
// this wave's LDS scratch.  MULTI: the wave is one of several in its workgroup (the fused correlation + Linear1 kernel
// below) -- the LDS hand-overs between the lanes of the wave are then ordered by a wave-level fence instead of the
// workgroup barrier (LDS instructions of one wave execute in issue order; waves take data-dependent paths here).
template <bool MULTI>
static __device__ __forceinline__ void corr_wave_sync() {
  if constexpr (MULTI) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}
template <typename T, bool CHUNKED, bool MULTI>
static __device__ __forceinline__ void corr_edge_rows(const CorrParams &prm, const int e, const int lane, float *__restrict__ Cs,
                                                      float *__restrict__ outs, float (&res)[CORR_MAXLEV][7]) {
  typedef CorrMma<T> M;
  typedef typename M::afrag_t afrag_t;
  typedef typename M::bfrag_t bfrag_t;
  typedef typename M::plane_t PT;
  constexpr int STEPS = M::STEPS, PER = M::PER;   // MFMA-operand loads per pixel, channels per lane and load
  constexpr int FE = M::FE;                       // plane elements per channel (2: high and low parts, CorrMma<CorrX2>)
  static_assert(FE == 1 || CHUNKED, "split planes exist in the chunked layout only");
  constexpr int C = 128, PP = 9, R = 3, D = 8, d = 7;
  constexpr int KOUT = d;                    // 7 per lane (lanes 0..62), held until both levels are done
  constexpr int PGB = M::PGB;                     // pixel groups whose loads are in flight together
  CTS(ct_start);
#ifdef CORR_TRACE
  unsigned long long ct_load = 0, ct_mma = 0, ct_blend = 0, ct_setup = 0;
#endif
  const int q = lane >> 4, j = lane & 15;
  const long i1 = prm.mod_ii > 0 ? prm.ii[e] % prm.mod_ii : prm.ii[e];   // ring-buffer slots (Ramp_vo.py:178-179)
  const long j2 = prm.mod_jj > 0 ? prm.jj[e] % prm.mod_jj : prm.jj[e];
  const int L = prm.nlevels;

  // window geometry of the nine patch pixels at every level: lane p < 9 holds pixel p's in registers; what the wave
  // needs as a whole comes over v_readlane / ds_bpermute (an LDS round trip + barrier per level before: a level with
  // nothing in the plane -- every second one in the bench's steady state -- cost 2 us of LDS latency for a row of zeros)
  int g_ox_[CORR_MAXLEV], g_oy_[CORR_MAXLEV];
  float g_dx_[CORR_MAXLEV], g_dy_[CORR_MAXLEV];
  unsigned g_lm_[CORR_MAXLEV];
  // pixel `lane`'s reprojection: loaded, or (fused pops.transform) computed here and written out
  float cx0 = 0.f, cy0 = 0.f;
  if (lane < PP) {
    float *cw = const_cast<float *>(prm.coords) + (size_t)e * 2 * PP;
    if (prm.tf_poses) {
      const long si = prm.tf_src[e], sj = prm.jj[e], sk = prm.ii[e];      // (ii of this launch = the patch index kk)
      float Ti[7], Tj[7], Tinv[7], G[7], t[3], qr[4];
#pragma unroll
      for (int c = 0; c < 7; c++) { Ti[c] = prm.tf_poses[7 * si + c]; Tj[c] = prm.tf_poses[7 * sj + c]; }
      lt_inv(Ti, Tinv);
      lt_mul(Tj, Tinv, G);
      lt_load(G, t, qr);
      const float fxi = prm.tf_intr[4 * si + 0], fyi = prm.tf_intr[4 * si + 1], cxi = prm.tf_intr[4 * si + 2], cyi = prm.tf_intr[4 * si + 3];
      const float fxj = prm.tf_intr[4 * sj + 0], fyj = prm.tf_intr[4 * sj + 1], cxj = prm.tf_intr[4 * sj + 2], cyj = prm.tf_intr[4 * sj + 3];
      const float *pt = prm.tf_patches + (size_t)sk * 3 * PP;
      float X0[4], X1[4];
      X0[0] = (pt[lane] - cxi) / fxi;
      X0[1] = (pt[PP + lane] - cyi) / fyi;
      X0[2] = 1.0f;
      X0[3] = pt[2 * PP + lane];
      lt_act4_tq(t, qr, X0, X1);
      const float Z = X1[2] < 0.1f ? 0.1f : X1[2];
      const float dz = 1.0f / Z;
      cx0 = fxj * (dz * X1[0]) + cxj;
      cy0 = fyj * (dz * X1[1]) + cyj;
      cw[lane] = cx0; cw[PP + lane] = cy0;
    } else {
      cx0 = cw[lane]; cy0 = cw[PP + lane];
    }
  }
#pragma unroll
  for (int lvl = 0; lvl < CORR_MAXLEV; lvl++) {
    g_ox_[lvl] = 0; g_oy_[lvl] = 0; g_dx_[lvl] = 0.f; g_dy_[lvl] = 0.f;
    bool live = false;
    if (lvl < L && lane < PP) {
      const int H2 = prm.H2[lvl], W2 = prm.W2[lvl];
      const float cdv = prm.cdiv[lvl];
      const float x = cx0 / cdv;
      const float y = cy0 / cdv;
      const float flx = floorf(x), fly = floorf(y);
      const int ox = ramp_f2i(flx), oy = ramp_f2i(fly);
      g_dx_[lvl] = x - flx;
      g_dy_[lvl] = y - fly;
      live = ((long)ox - R < W2) && ((long)ox - R + D > 0) && ((long)oy - R < H2) && ((long)oy - R + D > 0);
      g_ox_[lvl] = live ? ox - R : 0;
      g_oy_[lvl] = live ? oy - R : 0;
    }
    g_lm_[lvl] = (unsigned)__ballot(live);
  }
#ifdef CORR_TRACE
  const unsigned long long ct_geo = __builtin_readcyclecounter();   // indices + coordinates have arrived (the ballots used them)
#endif
  // (a factor with nothing in any plane needs no patch features)
  bool any_live = false;
#pragma unroll
  for (int lvl = 0; lvl < CORR_MAXLEV; lvl++) any_live |= g_lm_[lvl] != 0;
  afrag_t afrag[STEPS];
  {
    const size_t src = (size_t)i1 * C * PP;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      if (j < PP && any_live) afrag[s] = M::load_a(prm.fmap1, src + j * C + 4 * PER * s + PER * q);
      else afrag[s] = M::zero();
    }
  }

  const int op_p = lane % PP, op_a = lane / PP;   // output ownership, see the union epilogue
#pragma unroll
  for (int lvl = 0; lvl < CORR_MAXLEV; lvl++) {
    if (lvl >= L) break;
    CTS(ct_l0);
    const int H2 = prm.H2[lvl], W2 = prm.W2[lvl];
    const PT *f2 = reinterpret_cast<const PT *>(prm.fmap2[lvl]) + (size_t)((lvl == 0 && prm.slot0) ? (long)prm.slot0[j2] : j2) * (C * FE) * H2 * W2;
    const int my_ox = g_ox_[lvl], my_oy = g_oy_[lvl];
    const float my_dx = g_dx_[lvl], my_dy = g_dy_[lvl];
    const unsigned lmask = g_lm_[lvl];
    const float o_dx = __shfl(my_dx, op_p, 64), o_dy = __shfl(my_dy, op_p, 64);   // of the pixel whose outputs the lane owns
    if (lmask == 0) {
      // output o = lane + 63 k belongs to pixel o % 9 = lane % 9 for every k
      float s = ((1 - o_dx) * (1 - o_dy)) * 0.0f;
      s = s + (o_dx * (1 - o_dy)) * 0.0f;
      s = s + ((1 - o_dx) * o_dy) * 0.0f;
      s = s + (o_dx * o_dy) * 0.0f;
#pragma unroll
      for (int k = 0; k < KOUT; k++) res[lvl][k] = s;
      continue;
    }
    const int o_ox = __shfl(my_ox, op_p, 64), o_oy = __shfl(my_oy, op_p, 64);
    const bool o_live = (lmask >> op_p) & 1;
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
#pragma unroll
    for (int p = 0; p < PP; p++) {
      const int px = __builtin_amdgcn_readlane(my_ox, p), py = __builtin_amdgcn_readlane(my_oy, p);
      if ((lmask >> p) & 1) {
        minx = min(minx, px); maxx = max(maxx, px);
        miny = min(miny, py); maxy = max(maxy, py);
      }
    }
    const long bw = (long)maxx - minx + D, bh = (long)maxy - miny + D;
    const bool uni = bw * bh <= CORR_TM;
    const int ngroups = uni ? 1 : PP;
    for (int g = 0; g < ngroups; g++) {
      const int g_ox = __builtin_amdgcn_readlane(my_ox, g), g_oy = __builtin_amdgcn_readlane(my_oy, g);
      const float g_dx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_dx), g));
      const float g_dy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_dy), g));
      if (!uni && !((lmask >> g) & 1)) {
        for (int ab = lane; ab < d * d; ab += 64) {
          const float dx = g_dx, dy = g_dy;
          float s = ((1 - dx) * (1 - dy)) * 0.0f;
          s = s + (dx * (1 - dy)) * 0.0f;
          s = s + ((1 - dx) * dy) * 0.0f;
          s = s + (dx * dy) * 0.0f;
          outs[ab * PP + g] = s;
        }
        continue;
      }
      const int gx0 = uni ? minx : g_ox, gy0 = uni ? miny : g_oy;
      const int gw = uni ? (int)bw : D, gh = uni ? (int)bh : D;
      const int Tn = gw * gh;                      // <= CORR_TM
      const int npg = (Tn + 15) / 16;
      const int inv_gw = (65536 + gw - 1) / gw;    // t / gw == (t * inv_gw) >> 16 while t * gw < 65536
#ifdef CORR_TRACE
      { CTS(ct_l1); ct_setup += ct_l1 - ct_l0; }
#endif
      for (int pg0 = 0; pg0 < npg; pg0 += PGB) {
        CTS(ct_b0);
        // all PGB x 4 sixteen-byte loads of the batch are issued before the first MFMA waits on
        // one: the address is always a valid pixel, out-of-window lanes are zeroed afterwards
        bfrag_t bfr[PGB][STEPS];
        bool inb[PGB];
#pragma unroll
        for (int u = 0; u < PGB; u++) {
          const int t = (pg0 + u) * 16 + j;
          const int ty = (t * inv_gw) >> 16, tx = t - ty * gw;
          const int px = gx0 + tx, py = gy0 + ty;
          inb[u] = (t < Tn) && px >= 0 && px < W2 && py >= 0 && py < H2;
          const int cy = inb[u] ? py : 0, cx = inb[u] ? px : 0;
          // load s, quarter q <-> channels [4 PER s + PER q, + PER) (fp16: = chunk 4 s + q of the [h][C/8][w][8] layout)
#if CORR_KPLANE == 32
          // [h][4][w][32]: K step s of a window row is one run of 64 bytes per pixel (18.8 vs 15.7 TB/s from the vector
          // L1 for the 10-wide windows, tools/mb/gather_patterns.hip P5 / P2).  fp32 features: [h][8][w][16] -- the same 64
          // bytes per pixel and load (load s of quarter q <-> channels 16 s + 4 q .. + 3), eight planes
          // split fp32 features (FE = 2): [h][4][2][w][32] fp16 -- per K step a plane of high parts, then one of low parts, so
          // that every load instruction reads the fp16 kernel's contiguous 64-byte-per-pixel runs (pairs side by side,
          // [h][4][w][2][32], touch 16 half-used lines per instruction: 334 us against the fp32 kernel's 336)
          constexpr int KP = 4 * PER;                       // channels per plane: 32 (fp16, split fp32), 16 (fp32)
          const PT *pp = CHUNKED ? f2 + (((size_t)cy * (C / KP) * FE) * W2 + cx) * KP + PER * q
                                 : f2 + ((size_t)cy * W2 + cx) * C + PER * q;
          const size_t sstride = CHUNKED ? (size_t)W2 * (KP * FE) : 4 * PER;
          const size_t part = (size_t)W2 * KP;
#else
          static_assert(FE == 1, "CORR_KPLANE=8 builds have no split planes");
          const PT *pp = CHUNKED ? f2 + (((size_t)cy * (C / 8) + q) * W2 + cx) * 8
                                 : f2 + ((size_t)cy * W2 + cx) * C + PER * q;
          const size_t sstride = CHUNKED ? (size_t)4 * W2 * 8 : 4 * PER;
          const size_t part = 0;
#endif
#pragma unroll
          for (int s = 0; s < STEPS; s++) bfr[u][s] = M::load_b(pp + s * sstride, part);
        }
#ifdef CORR_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CTS(ct_b1);
        ct_load += ct_b1 - ct_b0;
#endif
#pragma unroll
        for (int u = 0; u < PGB; u++) {
          const int t = (pg0 + u) * 16 + j;
          typename M::acc_t acc = M::acc_zero();
#pragma unroll
          for (int s = 0; s < STEPS; s++) acc = M::mma(afrag[s], bfr[u][s], acc);
          // D: rows 4q..4q+3 = patch pixels, column j = union pixel t; an out-of-map pixel
          // contributes zeros (the loads above fetched a valid stand-in pixel for it)
          if (t < Tn) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int p = 4 * q + r;
              if (p < PP) Cs[p * CORR_TM + t] = inb[u] ? M::val(acc, r) : 0.0f;
            }
          }
        }
#ifdef CORR_TRACE
        { CTS(ct_b2); ct_mma += ct_b2 - ct_b1; }
#endif
      }
      corr_wave_sync<MULTI>();
      CTS(ct_c0);
      if (uni) {
        // lane = 9 a + p owns output row a of patch pixel p; its 7 outputs (b = 0..6) are
        // o = (7 b + a) 9 + p = lane + 63 b and need two 8-wide rows of Cs
        if (lane < 63) {
          float r0[D], r1[D];
          if (o_live) {
            const float *row = &Cs[op_p * CORR_TM + (o_oy - gy0 + op_a) * gw + (o_ox - gx0)];
#pragma unroll
            for (int b = 0; b < D; b++) { r0[b] = row[b]; r1[b] = row[gw + b]; }
          } else {
#pragma unroll
            for (int b = 0; b < D; b++) { r0[b] = 0.f; r1[b] = 0.f; }
          }
          const float dx = o_dx, dy = o_dy;
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
          const int wx = g_ox - gx0 + b, wy = g_oy - gy0 + a;
          const float *row = &Cs[g * CORR_TM + wy * gw + wx];
          const float c00 = row[0], c01 = row[1], c10 = row[gw], c11 = row[gw + 1];
          const float dx = g_dx, dy = g_dy;
          float s = ((1 - dx) * (1 - dy)) * c00;
          s = s + (dx * (1 - dy)) * c01;
          s = s + ((1 - dx) * dy) * c10;
          s = s + (dx * dy) * c11;
          outs[ab * PP + g] = s;
        }
      }
      corr_wave_sync<MULTI>();
#ifdef CORR_TRACE
      { CTS(ct_c1); ct_blend += ct_c1 - ct_c0; }
#endif
    }
    if (!uni) {
      corr_wave_sync<MULTI>();
      if (lane < 63) {
#pragma unroll
        for (int k = 0; k < KOUT; k++) res[lvl][k] = outs[lane + 63 * k];
      }
    }
    corr_wave_sync<MULTI>();
  }
#ifdef CORR_TRACE
  {
    CTS(ct_end);
    CTACC(0, 1); CTACC(1, ct_end - ct_start); CTACC(2, ct_setup); CTACC(3, ct_load); CTACC(4, ct_mma); CTACC(5, ct_blend);
    CTACC(6, ct_geo - ct_start); CTACC(7, any_live ? 1 : 0);
  }
#endif
}

template <typename T, bool CHUNKED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CorrMma<T>::WAVES, 8)))
    corr_mfma_kernel(const CorrParams prm) {
  typedef CorrMma<T> M;
  constexpr int PP = 9, d = 7, NOUT = d * d * PP, KOUT = d;       // 441 values per level, 7 per lane
  __shared__ __attribute__((aligned(16))) float Cs[PP * CORR_TM];
  __shared__ float outs[NOUT];               // staging for the ragged (non-union) paths only
  const int e = corr_edge_of_block(prm);
  if (e < 0) return;
  const int lane = threadIdx.x, L = prm.nlevels;
  float res[CORR_MAXLEV][KOUT];
  corr_edge_rows<T, CHUNKED, false>(prm, e, lane, Cs, outs, res);
  // out[e][o][lvl]: with two levels a lane's pair is one 4-byte store, consecutive over lanes
  typedef typename M::out_t OT;
  OT *op = reinterpret_cast<OT *>(prm.out) + (size_t)e * prm.row_elems;
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
}


// ------------------------------------------------------------- pyramid pack
// One frame's fp16 NHWC feature map [H][W][128] -> the two correlation levels in the chunked
// layout: level 1 = the map itself as [H][4][W][32]; level 4 = its 4x4 average (fp32 sum / 16,
// rounded once; Ramp_vo.py:378-381's avg_pool2d) as [H/4][4][W/4][32].  A workgroup owns a 4-row x
// 16-pixel tile; thread (xl, c8) moves 16 bytes per row through an LDS transpose so that both the
// reads (pixel-major) and the writes (plane-major, 1 KB runs) are contiguous.
__global__ void __launch_bounds__(256) pyramid_pack_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out1,
                                                           uint4 *__restrict__ out4, int H, int W) {
  __shared__ uint4 tile[16][17];
  const int t = threadIdx.x;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 4;
  const int xl = t >> 4, c8 = t & 15;      // read role
#if CORR_KPLANE != 32
  const int wc = t >> 4, wx = t & 15;      // write role: chunk wc, pixel wx
#endif
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
#if CORR_KPLANE == 32
    {
      const int ws = t >> 6, px = (t >> 2) & 15, wq = t & 3;     // write role: K step, pixel, quarter -- 1 KB runs
      out1[(((size_t)y * 4 + ws) * W + x0 + px) * 4 + wq] = tile[4 * ws + wq][px];
    }
#else
    out1[((size_t)y * 16 + wc) * W + x0 + wx] = tile[wc][wx];
#endif
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
#if CORR_KPLANE == 32
    out4[(((size_t)blockIdx.y * 4 + (c8 >> 2)) * (W / 4) + (x0 + xl) / 4) * 4 + (c8 & 3)] = o;
#else
    out4[((size_t)blockIdx.y * 16 + c8) * (W / 4) + (x0 + xl) / 4] = o;
#endif
  }
}

// fp32 features: NHWC [H][W][128] -> level 1 as [H][8][W][16] (one plane per 16-byte-per-lane load of corr_mfma_kernel<float>:
// a load's 64 lanes read 16 neighbouring window pixels x 64 bytes = one or two contiguous runs instead of 16 pieces 512 bytes
// apart -- with plain NHWC the window loads were 61 % of that kernel's cycles, tools/corr_trace.py) and level 4 = the 4x4
// mean as [H/4][8][W/4][16]: the window summed in (ky, kx) order, then x 1/16 -- torch's avg_pool2d to the bit
// (Ramp_vo.py:378-381).  Thread = one float4 of the output.
__global__ void __launch_bounds__(256) pyramid_pack_f32_kernel(const float4 *__restrict__ in, float4 *__restrict__ out1,
                                                               float4 *__restrict__ out4, int H, int W) {
  const long n1 = (long)H * W * 32, n4 = (long)(H / 4) * (W / 4) * 32;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o < n1) {
    // out1 index: ((y * 8 + g) * W + x) * 4 + qq
    const int qq = (int)(o & 3);
    const long r = o >> 2;
    const int x = (int)(r % W);
    const long r2 = r / W;
    const int g = (int)(r2 & 7);
    const long y = r2 >> 3;
    out1[o] = in[(y * W + x) * 32 + g * 4 + qq];
  } else if (o < n1 + n4) {
    const long o4 = o - n1;
    const int W4 = W / 4;
    const int qq = (int)(o4 & 3);
    const long r = o4 >> 2;
    const int X = (int)(r % W4);
    const long r2 = r / W4;
    const int g = (int)(r2 & 7);
    const long Y = r2 >> 3;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < 4; ky++)
      for (int kx = 0; kx < 4; kx++) {
        const float4 v = in[((Y * 4 + ky) * W + X * 4 + kx) * 32 + g * 4 + qq];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    out4[o4] = make_float4(a.x * 0.0625f, a.y * 0.0625f, a.z * 0.0625f, a.w * 0.0625f);
  }
}

// fp32 features as split fp16 pairs (CorrMma<CorrX2>): NHWC [H][W][128] -> level 1 as [H][4][2][W][32] fp16, level 4 = the
// 4x4 mean (summed in (ky, kx) order, x 1/16: torch's avg_pool2d to the bit, as above) split the same way.  Thread = 8
// channels of one pixel and K step: one 16-byte piece of high parts, one of low parts a [W][32] plane behind it.
__global__ void __launch_bounds__(256) pyramid_pack_x2_kernel(const float4 *__restrict__ in, uint4 *__restrict__ out1,
                                                              uint4 *__restrict__ out4, int H, int W) {
  const long n1 = (long)H * W * 16, n4 = (long)(H / 4) * (W / 4) * 16;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  float4 a, b;
  uint4 *dst;
  long lo_off;
  if (o < n1) {
    // o = ((y * 4 + s) * W + x) * 4 + qq: channels 32 s + 8 qq .. + 7
    const int qq = (int)(o & 3);
    const long r = o >> 2;
    const int x = (int)(r % W);
    const long r2 = r / W;
    const int s = (int)(r2 & 3);
    const long y = r2 >> 2;
    const float4 *src = in + (y * W + x) * 32 + s * 8 + qq * 2;
    a = src[0]; b = src[1];
    dst = out1 + ((r2 * 2) * W + x) * 4 + qq;
    lo_off = (long)W * 4;
  } else if (o < n1 + n4) {
    const long o4 = o - n1;
    const int W4 = W / 4;
    const int qq = (int)(o4 & 3);
    const long r = o4 >> 2;
    const int X = (int)(r % W4);
    const long r2 = r / W4;
    const int s = (int)(r2 & 3);
    const long Y = r2 >> 2;
    a = make_float4(0.f, 0.f, 0.f, 0.f); b = a;
    for (int ky = 0; ky < 4; ky++)
      for (int kx = 0; kx < 4; kx++) {
        const float4 *src = in + ((Y * 4 + ky) * W + X * 4 + kx) * 32 + s * 8 + qq * 2;
        const float4 u = src[0], v = src[1];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
      }
    a = make_float4(a.x * 0.0625f, a.y * 0.0625f, a.z * 0.0625f, a.w * 0.0625f);
    b = make_float4(b.x * 0.0625f, b.y * 0.0625f, b.z * 0.0625f, b.w * 0.0625f);
    dst = out4 + ((r2 * 2) * W4 + X) * 4 + qq;
    lo_off = (long)W4 * 4;
  } else {
    return;
  }
  const float x8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  f16x8_t hi, lo;
#pragma unroll
  for (int c = 0; c < 8; c++) { _Float16 h, l; corr_split2(x8[c], h, l); hi[c] = h; lo[c] = l; }
  dst[0] = __builtin_bit_cast(uint4, hi);
  dst[lo_off] = __builtin_bit_cast(uint4, lo);
}

extern "C" {

#ifdef CORR_TRACE
int ramp_debug_corr_trace(unsigned long long *host, int n_waves) {     // host [n_waves][8]
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_corr_trace), (size_t)n_waves * 8 * 8) == hipSuccess ? 0 : -1;
}
#endif
int ramp_corr_kplane(void) { return CORR_KPLANE; }

int ramp_pyramid_pack(const void *fmap, void *level1, void *level4, int H, int W, int C, int dtype,
                      void *stream) {
  if (!fmap || !level1 || !level4 || H <= 0 || W <= 0) return RAMP_EINVAL;
  const bool x2 = dtype == (RAMP_F32 | RAMP_CORR_X2);      // fp32 map -> split fp16 pairs (CorrMma<CorrX2>'s planes)
  if (x2) dtype = RAMP_F32;
  if (C != 128 || (dtype != RAMP_F16 && dtype != RAMP_F32) || (W % 16) || (H % 4)) return RAMP_EUNSUPPORTED;
  if (x2) {
    const long n = (long)H * W * 16 + (long)(H / 4) * (W / 4) * 16;
    hipLaunchKernelGGL(pyramid_pack_x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)fmap, (uint4 *)level1, (uint4 *)level4, H, W);
  } else if (dtype == RAMP_F32) {
    const long n = (long)H * W * 32 + (long)(H / 4) * (W / 4) * 32;
    hipLaunchKernelGGL(pyramid_pack_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)fmap, (float4 *)level1, (float4 *)level4, H, W);
  } else {
    hipLaunchKernelGGL(pyramid_pack_kernel, dim3(W / 16, H / 4), dim3(256), 0, (hipStream_t)stream,
                       (const uint4 *)fmap, (uint4 *)level1, (uint4 *)level4, H, W);
  }
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

int ramp_i_corr_fwd(const void *fmap1, const ramp_corr_level *levels, int nlevels,
                    const float *coords, const int64_t *ii, const int64_t *jj,
                    const int32_t *order, void *out, int out_row_elems, long mod_ii, long mod_jj, int E,
                    int N1, int N2, int C, int P, int radius, int dtype, int layout, const int32_t *dyn, void *stream,
                    const float *tf_poses, const float *tf_patches, const float *tf_intr, const int64_t *tf_src,
                    const int32_t *slot0) {
  if (E < 0 || nlevels < 1 || nlevels > CORR_MAXLEV || !levels) return RAMP_EINVAL;
  if (slot0 && mod_jj <= 0) return RAMP_EINVAL;
  if (tf_poses && (!tf_patches || !tf_intr || !tf_src || (dtype & ~(RAMP_CORR_MFMA32 | RAMP_CORR_X2)) != RAMP_F16 || layout == RAMP_NCHW)) return RAMP_EINVAL;
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
  prm.tf_poses = tf_poses; prm.tf_patches = tf_patches; prm.tf_intr = tf_intr; prm.tf_src = tf_src;
  prm.slot0 = slot0;
  const dim3 grid(prm.chunk * CORR_XCDS);
  hipStream_t st = (hipStream_t)stream;
  const bool x2 = (dtype & RAMP_CORR_X2) != 0;             // chunked planes hold split fp16 pairs (ramp_pyramid_pack, x2)
  const bool fast32 = (dtype & (RAMP_CORR_MFMA32 | RAMP_CORR_X2)) != 0;
  dtype &= ~(RAMP_CORR_MFMA32 | RAMP_CORR_X2);
  if (x2 && (dtype != RAMP_F32 || layout != RAMP_NHWC32)) return RAMP_EINVAL;
  if (x2)
    hipLaunchKernelGGL((corr_mfma_kernel<CorrX2, true>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F32 && layout == RAMP_NHWC && fast32)
    hipLaunchKernelGGL((corr_mfma_kernel<float, false>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F32 && layout == RAMP_NHWC32 && fast32)        // target maps as [h][8][w][16] (ramp_pyramid_pack, fp32)
    hipLaunchKernelGGL((corr_mfma_kernel<float, true>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F32 && layout == RAMP_NHWC)
    hipLaunchKernelGGL((corr_kernel<float, RAMP_NHWC>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F32 && layout == RAMP_NCHW)
    hipLaunchKernelGGL((corr_kernel<float, RAMP_NCHW>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F16 && layout == RAMP_NHWC)
    hipLaunchKernelGGL((corr_mfma_kernel<_Float16, false>), grid, dim3(64), 0, st, prm);
  else if (dtype == RAMP_F16 && layout == RAMP_NHWC32)
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
                         C, P, radius, dtype, layout, nullptr, stream, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int ramp_corr_fwd(const void *fmap1, const ramp_corr_level *levels, int nlevels,
                  const float *coords, const int64_t *ii, const int64_t *jj, void *out, int E,
                  int N1, int N2, int C, int P, int radius, int dtype, int layout,
                  void *stream) {
  return ramp_corr_fwd_ordered(fmap1, levels, nlevels, coords, ii, jj, nullptr, out, 0, 0, 0, E, N1, N2, C,
                               P, radius, dtype, layout, stream);
}

}  // extern "C"
