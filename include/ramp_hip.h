/*
 * ramp_hip.h -- C ABI of libramp_hip.so, the MI355X (gfx950) implementation of
 * the RAMP-VO tracking hot path.
 *
 * Every entry point replaces one native entry point (or one fused group of
 * ATen calls) of the reference; the reference interface it stands in for is
 * cited as file:line (paths relative to the upstream repository).  The
 * reference binds its natives through pybind11/torch::Tensor; this boundary is
 * plain C: device pointers, sizes, a hipStream_t (passed as void*), int status.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host
 *   - index arrays are int64 (torch.long), like the reference
 *   - return value: RAMP_OK (0) or a negative RAMP_E* code; kernels are
 *     enqueued on `stream` and NOT synchronised
 *   - no entry point allocates device memory: scratch is supplied by the
 *     caller (`ws`, sized by the matching *_workspace_bytes query)
 *   - poses are [tx ty tz qx qy qz qw] float32 (lietorch SE3 embedding)
 *   - feature maps come in two layouts: RAMP_NCHW (the reference's) and
 *     RAMP_NHWC (channels-last, this library's native layout: one pixel's
 *     channels are contiguous so a wavefront reads whole 512 B rows)
 */
#ifndef RAMP_HIP_H
#define RAMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAMP_OK 0
#define RAMP_EINVAL -1      /* bad argument (null pointer, unsupported size) */
#define RAMP_ELAUNCH -2     /* HIP reported a launch/runtime error            */
#define RAMP_EWORKSPACE -3  /* workspace too small                            */
#define RAMP_EUNSUPPORTED -4

#define RAMP_F32 0
#define RAMP_F16 1

/* ramp_corr_fwd*, fp32 + RAMP_NHWC only: OR into `dtype` to use the fp32 MFMA kernel (2.1x faster; results
 * within 1e-5 of the default kernel, which keeps the reference's channel-ordered fmaf accumulation)      */
#define RAMP_CORR_MFMA32 0x40
/* fp32 features as split fp16 pairs, "x2" (round 6): every value x = xh + xl 2^-11 (xh = fp16(x), xl = fp16((x - xh) 2^11): 22
 * significant bits), a dot product = three v_mfma_f32_16x16x32_f16 products (ah bh; ah bl + al bh) into two fp32 accumulators:
 * fp32-class accuracy (<= 1e-5 of the default kernel like RAMP_CORR_MFMA32, whose fp32 MFMAs cost >= 240 us per launch at
 * 41k factors) at the fp16 kernel's speed for twice the window bytes.  OR into `dtype` (RAMP_F32) of
 *   ramp_pyramid_pack: the planes are written as [H][4][2][W][32] fp16 (per row and 32-channel step a [W][32] plane of high
 *     parts, then one of low parts; 512 bytes per pixel like the fp32 planes) -- level 4 is pooled in fp32, then split;
 *   ramp_corr_fwd* with RAMP_NHWC32: the target maps are such planes (fmap1 stays fp32 RAMP_NHWC rows, split while loading;
 *     the output stays fp32).  Any other layout / dtype with this bit: RAMP_EINVAL.                                      */
#define RAMP_CORR_X2 0x80

#define RAMP_NCHW 0
#define RAMP_NHWC 1
#define RAMP_NHWC32 2  /* [H][C/32][W][32] (fp16) / [H][C/16][W][16] (fp32: with RAMP_CORR_MFMA32 only) / [H][4][2][W][32] fp16
                          parts (fp32 with RAMP_CORR_X2): correlation target maps (ramp_pyramid_pack); 64 bytes per pixel and
                          plane = one 16-byte load of every lane quarter */

/* library / build identification: returns a static string */
const char *ramp_version(void);

/* ------------------------------------------------------------------ altcorr */

/* cuda_corr.patchify_forward + the bilinear blend of altcorr.patchify
 * (ramp/altcorr/correlation.cpp:47-50, correlation_kernel.cu:16-47,288-307,
 *  ramp/altcorr/correlation.py:51-68).
 *   net    [n][C][H][W] (NCHW) or [n][H][W][C] (NHWC), dtype
 *   coords [n][M][2] float32 (x, y)
 *   out    bilinear=1: [n][M][C][2R+1][2R+1]   (altcorr.patchify, fused)
 *          bilinear=0: [n][M][C][2R+2][2R+2]   (raw cuda_corr.patchify_forward)
 *   out_layout RAMP_NCHW: as above;  RAMP_NHWC: [n][M][d][d][C]                */
int ramp_patchify_fwd(const void *net, const float *coords, void *out, int n, int C, int H,
                      int W, int M, int radius, int bilinear, int dtype, int layout,
                      int out_layout, void *stream);

/* All per-frame gathers at the M patch centres in one launch (ramp/net.py:167-203: gmap = 3x3 patches of fmap, imap =
 * 1x1 of the context map, the 3x3 (x, y, disparity=1) patches of the coordinate grid, the colours of the full-resolution
 * image at 4*(coords+0.5); Ramp_vo.py:353-354: uint8 BGR colours).  Bit-identical to the four ramp_patchify_fwd calls.
 *   fmap [h][w][CF], imap [h][w][CI] (NHWC, dtype RAMP_F32 / RAMP_F16); image [3][H][W] fp32; coords [M][2] fp32
 *   -> gmap [M][3][3][CF], imap_p [M][CI] (dtype), patches [M][3][3][3] fp32 (channel-major), clr [M][3] fp32,
 *      colors [M][3] uint8 (BGR)                                                                                  */
int ramp_frame_gather(const void *fmap, const void *imap, const float *image, const float *coords, void *gmap,
                      void *imap_p, float *patches, float *clr, unsigned char *colors, int M, int h, int w, int H,
                      int W, int CF, int CI, int dtype, void *stream);

/* cuda_corr.forward (ramp/altcorr/correlation.cpp:28-35,
 * correlation_kernel.cu:82-136 + host blend/permute 193-233), fused over the
 * levels of the feature pyramid and with the torch.stack(..., -1) of
 * ramp/Ramp_vo.py:175-182 folded into the store.
 *   fmap1   patch features: NCHW [N1][C][P][P] or NHWC [N1][P][P][C]
 *   level l target features: NCHW [N2][C][H2_l][W2_l] or NHWC [N2][H2_l][W2_l][C]
 *   coords  [E][2][P][P] float32, divided by coord_div_l inside the kernel
 *   ii[E]   index into fmap1,  jj[E] index into the level's fmap
 *   out     [E][2R+1 (x off)][2R+1 (y off)][P][P][nlevels], dtype
 * nlevels==1 reproduces cuda_corr.forward's (permuted, contiguous) result.   */
typedef struct {
  const void *fmap;
  int H2, W2;
  float coord_div;
} ramp_corr_level;

int ramp_corr_fwd(const void *fmap1, const ramp_corr_level *levels_host, int nlevels,
                  const float *coords, const int64_t *ii, const int64_t *jj, void *out, int E,
                  int N1, int N2, int C, int P, int radius, int dtype, int layout,
                  void *stream);

/* Same result as ramp_corr_fwd (every edge is computed independently, so the
 * schedule cannot change a value).  order[E] (int32, device; NULL = identity)
 * is the order in which edges are handed to workgroups: consecutive positions
 * run on the same XCD, so a target-frame-major order keeps each frame's
 * feature plane in one L2.  The tracker passes the (jj, ii) pair grouping of
 * its graph plan.  out_row_elems (0 = dense, 441 * nlevels): elements per edge
 * row of `out`; the tail of a longer row is zero filled -- 896 instead of 882
 * makes the rows 16-byte aligned for the first Linear layer of the update
 * operator (library GEMM: 49 instead of 88 us at E = 40k).  mod_ii / mod_jj
 * > 0: ii[e] % mod_ii and jj[e] % mod_jj are used (the `kk % (M*mem)`,
 * `jj % mem` ring-buffer slots of ramp/Ramp_vo.py:178-179).                 */
int ramp_corr_fwd_ordered(const void *fmap1, const ramp_corr_level *levels_host, int nlevels,
                          const float *coords, const int64_t *ii, const int64_t *jj,
                          const int32_t *order, void *out, int out_row_elems, long mod_ii, long mod_jj, int E,
                          int N1, int N2, int C, int P, int radius, int dtype, int layout, void *stream);

/* Ramp_vo.__call__'s pyramid store (ramp/Ramp_vo.py:378-381: fmap1_[slot] =
 * fmap, fmap2_[slot] = avg_pool2d(fmap, 4, 4)) for fp16 channels-last features:
 * fmap [H][W][C] -> level1 [H][C/32][W][32] (same values) and level4
 * [H/4][C/32][W/4][32] (4x4 mean, fp32 sum, one rounding).  These are the
 * RAMP_NHWC32 target maps of ramp_corr_fwd (fmap1 stays RAMP_NHWC).  fp32 features
 * (dtype RAMP_F32): planes of 16 channels, [H][8][W][16] and [H/4][8][W/4][16]; the
 * mean is the window summed in (ky, kx) order times 1/16 = torch's avg_pool2d.
 * dtype RAMP_F32 | RAMP_CORR_X2: the same fp32 input, planes of split fp16 pairs (see RAMP_CORR_X2; same byte counts).
 * C == 128, W % 16 == 0, H % 4 == 0, else RAMP_EUNSUPPORTED.                 */
int ramp_pyramid_pack(const void *fmap, void *level1, void *level4, int H, int W, int C, int dtype,
                      void *stream);
/* channels per plane of the RAMP_NHWC32 layout THIS build packs and reads ([H][C/k][W][k]; 32 unless the library was
 * built with -DCORR_KPLANE=8, round 2/3's layout): the caller sizes and converts its ring buffers with it -- a binding
 * that assumes another width must refuse to run (rampvo_amd/_lib.py asserts it when the library loads).             */
int ramp_corr_kplane(void);

/* Event list -> int8 bin stack, the encoder's event input (reference utils/transformers.py:128-161,
 * EventToStack_Numpy; upstream a host-side np.add.at).  Event i goes to bin
 * int32(float32(bins * i) / N); polarities p[i] (+-1) are accumulated per (bin, y, x) and the sum is cast to
 * int8 (wraps).  Integer pixel coordinates only (the reference's uint16 path; its sub-pixel bilinear path is
 * not provided).  out_i8 and/or out_f32 [bins][H][W]; ws: ramp_event_stack_workspace_bytes.          */
size_t ramp_event_stack_workspace_bytes(int bins, int H, int W);
int ramp_event_stack(const int32_t *x, const int32_t *y, const int8_t *p, int N, int bins, int H, int W,
                     int8_t *out_i8, float *out_f32, void *ws, size_t ws_bytes, void *stream);

/* Depth initialisation of a new frame (ramp/Ramp_vo.py:370-371): the lower median (torch.median) of the
 * inverse depths of the last F frames' patches -- patches_src = &patches_[n-F], [F][M][3][P][P] -- written
 * into channel 2 of the M new patches patches_dst [M][3][P][P].  F*M*P*P <= 4096.                   */
int ramp_depth_median_fill(const float *patches_src, int F, int M, int P, float *patches_dst, void *stream);
/* the same median into device memory (*out), for a caller that computes it ahead of ramp_frame_commit (median_dev) */
int ramp_depth_median(const float *patches_src, int F, int M, int P, float *out, void *stream);

/* Event-biased patch-centre selection: get_coords_from_topk_events + nms_image
 * (ramp/utils.py:186-226, 157-183; upstream ~20 ATen launches) for one frame.
 *   events [bins][H][W] float32 (W % 4 == 0); score = mean over bins of the 4x4 average of |events|,
 *   laid out [W/4][H/4]; local maxima of an nms_kernel_size^2 window kept (0 = no NMS);
 *   the k largest cells in descending order (ties: lowest flat index first)
 *   coords [k][2] float32 = (flat_index / (H/4) as a TRUE division -- x carries y/h, as upstream --,
 *   flat_index % (H/4));  indices [k] int64 flat indices (optional, may be NULL)
 *   ws: ramp_event_topk_workspace_bytes(H, W);  k <= 512                                          */
size_t ramp_event_topk_workspace_bytes(int H, int W);
int ramp_event_topk(const float *events, int bins, int H, int W, int k, int nms_kernel_size, float *coords,
                    int64_t *indices, void *ws, size_t ws_bytes, void *stream);

/* ----------------------------------------------------------------- lietorch */
/* lietorch_backends.{expm,logm,inv,mul,act4,adj,adjT} for group_id 3 (SE3),
 * float32, forward only (ramp/lietorch/src/lietorch.cpp:286-316; math from
 * ramp/lietorch/include/se3.h:30-142, so3.h:31-208).  n = batch elements.   */
int ramp_se3_exp(const float *a, float *X, int n, void *stream);
int ramp_se3_log(const float *X, float *a, int n, void *stream);
int ramp_se3_inv(const float *X, float *Y, int n, void *stream);
int ramp_se3_mul(const float *X, const float *Y, float *Z, int n, void *stream);
int ramp_se3_act4(const float *X, const float *p, float *q, int n, void *stream);
int ramp_se3_adj(const float *X, const float *a, float *b, int n, void *stream);
int ramp_se3_adjT(const float *X, const float *a, float *b, int n, void *stream);

/* ---------------------------------------------------------- projective ops */
/* pops.transform(SE3(poses), patches, intrinsics, ii, jj, kk[, tonly])
 * followed by .permute(0,1,4,2,3).contiguous(), i.e. Ramp_vo.reproject
 * (ramp/projective_ops.py:16-101 jacobian=False path, ramp/Ramp_vo.py:184-192;
 * ~10 torch kernels + 3 lietorch launches upstream).
 *   poses [Np][7], patches [Nk][3][P][P], intrinsics [Np][4] (per frame)
 *   out   [E][2][P][P]                                                        */
int ramp_transform(const float *poses, const float *patches, const float *intrinsics,
                   const int64_t *ii, const int64_t *jj, const int64_t *kk, float *out, int E,
                   int P, int tonly, void *stream);

/* cuda_ba.reproject (ramp/fastba/ba.cpp:48-56, ba_cuda.cu:379-429,585-617):
 * frame-0 intrinsics, no depth clamp.  out [E][2][P][P]                      */
int ramp_reproject(const float *poses, const float *patches, const float *intrinsics,
                   const int64_t *ii, const int64_t *jj, const int64_t *kk, float *out, int E,
                   int P, void *stream);

/* pops.point_cloud (ramp/projective_ops.py:103-105) reduced to what
 * Ramp_vo.update keeps (ramp/Ramp_vo.py:308-310): the 3-D point of each patch
 * centre.  ix[m] = source frame of patch m.  out [m][3]                      */
int ramp_point_cloud(const float *poses, const float *patches, const float *intrinsics,
                     const int64_t *ix, float *out, int m, int P, void *stream);

/* Ramp_vo.motionmag(i, j) and motionmag(j, i) in one launch (ramp/Ramp_vo.py:227-243 over
 * pops.flow_mag, projective_ops.py:108-118, beta-weighted full / translation-only flow).  The
 * edges of each direction are located through the (ii, jj) grouping of ramp_group_by[_small]
 * (order / seg / sorted unique keys / ngroups); key_ij, key_ji are the two pair keys in that
 * grouping's key space.  out2[0] = mean flow i->j, out2[1] = j->i (NaN when a direction has no edge). */
int ramp_motionmag(const float *poses, const float *patches, const float *intrinsics,
                   const int64_t *ii, const int64_t *jj, const int64_t *kk, const int32_t *order,
                   const int32_t *seg, const int64_t *ukeys, const int32_t *ngroups, int64_t key_ij,
                   int64_t key_ji, float beta, float *out2, int P, void *stream);

/* DAMPED_LINEAR motion model (ramp/Ramp_vo.py:356-363; 5 lietorch launches upstream):
 * poses[n] = Exp(damping * Log(poses[n-1] * poses[n-2]^-1)) * poses[n-1]                      */
int ramp_motion_model(float *poses, int n, float damping, void *stream);

/* The per-frame bookkeeping of Ramp_vo.__call__ (ramp/Ramp_vo.py:345-363: tstamps_[n], index_map_[n+1],
 * intrinsics_[n], motion-model pose) as one launch.  motion: 0 = none, 1 = DAMPED_LINEAR (as above),
 * 2 = poses[n] = poses[n-1]; copy_k: intrinsics[n] = intrinsics[n-1] (the caller writes the row itself when
 * the intrinsics changed); tstamps / index_map may be NULL.                                       */
int ramp_frame_begin(float *poses, int n, int motion, float damping, int64_t *tstamps, int64_t counter,
                     int64_t *index_map, int64_t index_val, float *intrinsics, int copy_k, void *stream);

/* ramp_frame_begin + ramp_depth_median_fill + ramp_multi_copy of a steady-state Ramp_vo.__call__ as ONE launch
 * (ramp/Ramp_vo.py:345-381; three dependent tiny launches on the frame's critical path otherwise):
 *   frame_begin's arguments as above; patches_state [N][M][3][P][P] fp32: the depth channel of patches_new
 *   [M][3][P][P] becomes the median of rows n - median_frames .. n - 1 (median_frames = 0: left as is) and
 *   patches_new is stored as row n; then dst[b] = src[b] for n_copy <= 6 further buffers (16-byte aligned,
 *   16-byte multiples: colours, imap, gmap, fmap1, fmap2 rows).                                          */
int ramp_frame_commit(float *poses, int n, int motion, float damping, int64_t *tstamps, int64_t counter,
                      int64_t *index_map, int64_t index_val, float *intrinsics, int copy_k, float *patches_state,
                      int median_frames, int M, int P, float *patches_new, int n_copy, const void *const *src_host,
                      void *const *dst_host, const long *bytes_host, const float *median_dev, void *stream);

/* Tracker bookkeeping helpers (host-side pointer arrays, <= 10 buffers per call).
 * ramp_multi_copy: dst[b][0:bytes[b]) = src[b][...] -- the per-frame stores of imap/gmap/fmap1/fmap2/
 *   patches/colors into the state buffers (ramp/Ramp_vo.py:345-381) in one launch.
 * ramp_shift_rows: rows k+1..nrows-1 of every buffer move down by one -- the keyframe-removal loop of
 *   ramp/Ramp_vo.py:258-268; mod[b] > 0 marks a ring buffer (row r lives at slot r % mod[b]).        */
int ramp_multi_copy(const void *const *src_host, void *const *dst_host, const long *bytes_host, int n,
                    void *stream);
int ramp_shift_rows(void *const *base_host, const long *row_bytes_host, const int *mod_host, int n, int k,
                    int nrows, void *stream);

/* ------------------------------------------------------------------- graph */
/* group the E edges by an int64 key (device-side replacement of the
 * torch.unique / torch::_unique / std::stable_sort host round trips at
 * ramp/blocks.py:43, ramp/fastba/ba_cuda.cu:447, ramp/fastba/ba.cpp:59-97).
 *   key_bound : exclusive upper bound of the keys (0 = unknown: full 63 bits)
 *   order[E]      edge indices sorted by (key, edge index)  (stable)
 *   gid[E]        dense rank of the edge's key among the sorted unique keys
 *                 (== the `inverse` of torch.unique(sorted=True))
 *   seg_start[E+1] first sorted position of every group, seg_start[G] = E
 *   ukeys[E]      the sorted unique keys (first G entries valid), may be NULL
 *   ngroups       device int: G                                              */
size_t ramp_group_by_workspace_bytes(int E);
int ramp_group_by(const int64_t *keys, int E, int64_t key_bound, int32_t *order, int32_t *gid,
                  int32_t *seg_start, int64_t *ukeys, int32_t *ngroups, void *ws,
                  size_t ws_bytes, void *stream);

/* cuda_ba.neighbors(kk, jj) (ramp/fastba/ba.cpp:59-97): previous / next edge
 * of the same kk ordered by jj (stable), -1 at either end.  Entirely on the
 * device (the reference copies to the host, sorts, copies back).
 * kk_bound / jj_bound: exclusive upper bounds (0 = unknown).                 */
size_t ramp_neighbors_workspace_bytes(int E);
int ramp_neighbors(const int64_t *kk, const int64_t *jj, int64_t *ix, int64_t *jx, int E,
                   int64_t kk_bound, int64_t jj_bound, void *ws, size_t ws_bytes,
                   void *stream);

/* SoftAgg core (ramp/blocks.py:44-45; torch_scatter.scatter_softmax +
 * scatter_sum): y[g][c] = sum_{e in g} softmax_e(gx[e][c]) * fx[e][c]
 * with the groups given by ramp_group_by's (order, seg_start, ngroups).
 *   fx, gx [E][C] dtype;  y [>=G][C] dtype (rows >= G untouched)             */
int ramp_segment_softmax_sum(const void *fx, const void *gx, const int32_t *order,
                             const int32_t *seg_start, const int32_t *ngroups, void *y, int E,
                             int C, int max_groups, int dtype, void *stream);

/* ------------------------------------------------------------------ fastba */
/* cuda_ba.forward (ramp/fastba/ba.cpp:32-45, ba_cuda.cu:433-582): `iterations`
 * damped Gauss-Newton steps on the reprojection error of the patch centres,
 * poses t0..t1-1 free.  poses [n_poses][7] and patches [n_patches][3][P][P]
 * are updated IN PLACE (rows t0..t1-1 / depth channel of the patches that
 * appear in kk), exactly as the reference mutates its arguments.
 *   target, weight [E][2]; lmbda [1]; intrinsics [>=1][4] (row 0 is used,
 *   ba_cuda.cu:253-258)
 *   info: optional device int (bit mask, zeroed by the call): bit 0 = the Cholesky factorisation hit a
 *         non-positive pivot or produced a step that is not finite (the reference discards cholesky_ex's info and
 *         returns NaN poses; here the pose step of that iteration is dropped); bit 1 = more than 1024 pose-pair records touch one pose (> 512 frames
 *         connected to one frame): the normal equations are incomplete, the result must be discarded
 * Deterministic: all reductions are ordered segment sums, no float atomics.  */
size_t ramp_ba_workspace_bytes(int E, int n_poses, int n_patches, int t0, int t1);
int ramp_ba_forward(float *poses, float *patches, const float *intrinsics, const float *target,
                    const float *weight, const float *lmbda, const int64_t *ii,
                    const int64_t *jj, const int64_t *kk, int E, int P, int n_poses,
                    int n_patches, int t0, int t1, int iterations, void *ws, size_t ws_bytes,
                    int32_t *info, void *stream);

/* ramp_ba_forward with the two edge groupings supplied by the caller (the tracker builds them once
 * per graph change and shares them with the update operator's SoftAgg): by patch kk
 * (order_k / seg_k / ngroups_k / ukeys_k = sorted unique kk) and by pose pair (ii, jj) in
 * lexicographic order (order_p / seg_p / ngroups_p), as produced by ramp_group_by[_small].
 * max_patches / max_pairs: upper bounds of the group counts (grid sizes, workspace).          */
size_t ramp_ba_planned_workspace_bytes(int E, int n_poses, int n_patches, int t0, int t1,
                                       int max_patches, int max_pairs);
int ramp_ba_forward_planned(float *poses, float *patches, const float *intrinsics, const float *target,
                            const float *weight, const float *lmbda, const int64_t *ii,
                            const int64_t *jj, const int64_t *kk, int E, int P, int n_poses,
                            int n_patches, int t0, int t1, int iterations, const int32_t *order_k,
                            const int32_t *seg_k, const int32_t *ngroups_k, const int64_t *ukeys_k,
                            int max_patches, const int32_t *order_p, const int32_t *seg_p,
                            const int32_t *ngroups_p, int max_pairs, void *ws, size_t ws_bytes,
                            int32_t *info, void *stream);

/* group-by for a SMALL key range known to the caller: key = a[e]*mul + (b ? b[e] : 0) - sub must lie
 * in [0, K).  Histogram + one-workgroup scan + scatter + per-segment rank sort (5 short kernels vs a
 * radix sort); same outputs and the same (stable) ordering as ramp_group_by.  ukeys = key + sub.  */
size_t ramp_group_by_small_workspace_bytes(int E, int K);
int ramp_group_by_small(const int64_t *a, const int64_t *b, int64_t mul, int64_t sub, int K, int E,
                        int32_t *order, int32_t *gid, int32_t *seg_start, int64_t *ukeys,
                        int32_t *ngroups, int max_groups, void *ws, size_t ws_bytes, void *stream);

/* cuda_ba.neighbors derived from the per-kk groups (no second sort): every group's edges are
 * ranked by (jj, edge index).  Groups longer than 1024 edges are left untouched (use
 * ramp_neighbors for such graphs).  kj_order (optional, int32 [E]): the factors in that (kk, jj) order -- a
 * factor's temporal neighbours are its neighbours in this list.                                    */
int ramp_neighbors_from_groups(const int32_t *order, const int32_t *seg_start, const int32_t *ngroups,
                               const int64_t *jj, int64_t *ix, int64_t *jx, int32_t *kj_order, int E,
                               int max_groups, void *stream);

/* ------------------------------------------------------------------ encoder */
/* flags[0] = any(a != 0), flags[1] = any(b != 0): the "events / image present" tests of
 * ramp/extractor.py:253-254, kept on the device (the reference syncs the host on each).       */
int ramp_any_nonzero(const float *a, long na, const float *b, long nb, int32_t *flags, void *stream);
/* the same test with one result per workgroup and no memset in front: blockflags [2][1024] int32 (row 0: a, row 1: b);
 * returns the number of workgroups nblk (> 0) whose results are valid, or an error code (< 0).  The consumer
 * (ramp_lstm_superstate_blocks) ORs blockflags[r][0 .. nblk).                                                       */
int ramp_any_nonzero_blocks(const float *a, long na, const float *b, long nb, int32_t *blockflags, void *stream);
/* the same launch also zeroes `clear` [clear_bytes] (16-byte aligned, a multiple of 16): the InstanceNorm accumulators of
 * the frame's tower pass (ramp_conv job acc_out / acc_in) without a memset launch in front of the front end          */
int ramp_any_nonzero_blocks_clear(const float *a, long na, const float *b, long nb, int32_t *blockflags, void *clear,
                                  long clear_bytes, void *stream);

/* The two per-pixel LSTM cells and the super-state 1x1 convolution of the SingleScale encoder,
 * fused (ramp/extractor.py:239-259: nn.LSTM x2 on [H*W,1,C] sequences + Conv2d(30->15) x2), on the
 * matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products): the three matrix-vector products per
 * pixel are batched over 16-pixel tiles.
 *   ev [5][HW], im [3][HW] planar float32
 *   h_ev,c_ev,h_im,c_im [ceil(HW/16)][16 px][4 q][4 t] tile-major recurrent state, unit = 4 t + q (in/out, unit 15 = 0):
 *   a lane's four K-step operands / outputs are one 16-byte access
 *   ss [HW][16] channels-last super-state (in/out, channel 15 = 0)
 *   wfrag: per-lane MFMA fragments of the weights, rampvo_amd/conv_hip.py::pack_lstm_mfma
 *   flags[2]: events / image present (ramp_any_nonzero)
 *   has_state / has_ss: 0 on the first call after reinit_hidden (zero initial state)            */
int ramp_lstm_superstate_tiled(const float *ev, const float *im, float *h_ev, float *c_ev, float *h_im,
                               float *c_im, float *ss, const float *wfrag, const int32_t *flags, int HW,
                               int has_state, int has_ss, void *stream);
/* ramp_lstm_superstate_tiled with the presence flags as ramp_any_nonzero_blocks leaves them (nblk > 0) */
int ramp_lstm_superstate_blocks(const float *ev, const float *im, float *h_ev, float *c_ev, float *h_im,
                                float *c_im, float *ss, const float *wfrag, const int32_t *blockflags, int nblk, int HW,
                                int has_state, int has_ss, void *stream);

/* One scale (1, 2 or 4) of the MultiScale encoder's recurrent front end for one time step
 * (ramp/extractor.py:540-566 with LSTMEncoder :376-385 and SuperStateEncoder :432-463): strided
 * conv_1 on events / image, a per-pixel LSTM step from the zero state (hidden size D = 16*scale),
 * then s <- mix_ev([s ; h_ev]) and, if use_im (the frame's mask), s <- mix_im([s ; h_im]).
 *   ev [5][H][W], im [3][H][W] float32;  state [Hs*Ws][D] channels-last super-state, in/out
 *   weights_host: 12 device pointers (a host array): conv_1 ev W [5][5][K][K], b; conv_1 im W
 *   [3][3][K][K], b; LSTM ev W_ih [4D][5], b_ih+b_hh [4D]; LSTM im W_ih [4D][3], b [4D]; mix ev W^T
 *   [2D][D], b [D]; mix im W^T [2D][D], b [D]    (packed by rampvo_amd/conv_hip.py::pack_ms_scale)
 *   has_state: 0 on the first call after reinit_hidden (zero super-state)                        */
int ramp_ms_lstm_superstate(const float *ev, const float *im, const float *const *weights_host,
                            float *state, int H, int W, int scale, int has_state, int use_im,
                            void *stream);

/* The same step with every matrix-vector product batched over 16 pixels on v_mfma_f32_16x16x4_f32 (exact fp32 products; the
 * summation order differs from the VALU kernel's): wfrag = per-lane A fragments [nfrag][64], wsmall = conv_1 weights /
 * biases and the gate / mix biases (both packed by rampvo_amd/conv_hip.py::pack_ms_scale_mfma; layouts in csrc/conv.hip).
 * state16: optional [Hs*Ws][D] fp16 copy of the new super-state (the conv towers' second / third input).           */
int ramp_ms_lstm_superstate_mfma(const float *ev, const float *im, const float *wfrag, const float *wsmall, float *state,
                                 void *state16, int H, int W, int scale, int has_state, int use_im, void *stream);

/* nn.Conv2d (+ fused neighbours) of the encoder towers as an implicit GEMM on MFMA
 * (ramp/extractor.py:8-57, 60-130; reference: cuDNN).  NHWC activations, padding = K/2.
 *   x [H][W][Cin] (Cin % 16 == 0), y [OH][OW][Cout] (Cout % 32 == 0)
 *   wpk: weights in MFMA fragment order (rampvo_amd/conv_hip.py::pack_conv_weight)
 *   pre_scale/pre_shift [Cin] (optional): x <- relu(x*scale + shift) while loading, i.e. the
 *       producer's InstanceNorm + ReLU fused into this conv
 *   y = [relu]( conv + bias );  if res: y = relu(y + res);  y *= out_scale
 *   stats (optional) [Cout][2][ramp_conv2d_stats_blocks(...)]: per-block partial sum / sum of
 *       squares of (conv + bias), reduced by ramp_in_stats_finalize                                     */
int ramp_conv2d_nhwc(const void *x, const void *wpk, const float *bias, const float *pre_scale,
                     const float *pre_shift, const void *res, void *y, float *stats, int H, int W,
                     int Cin, int Cout, int KH, int KW, int stride, int relu, float out_scale,
                     int dtype, void *stream);

/* One layer of up to TWO independent conv problems of the same shape (the fmap and imap towers read the same input
 * through the same layer shapes and differ in weights, norm and the last layer's Cout) as ONE launch of the
 * LDS-tiled fp16 kernel; fields as the arguments of ramp_conv2d_nhwc (stats: reduce with ramp_in_stats_finalize).
 * RAMP_EUNSUPPORTED for layer shapes the tiled kernel does not cover (use ramp_conv2d_nhwc).                     */
typedef struct ramp_conv_job {
  const void *x, *wpk;
  const float *bias, *pre_scale, *pre_shift;
  const void *res;
  void *y;
  float *stats;
  int32_t Cout, relu;
  float out_scale;
  float act_scale, w_scale;   /* RAMP_CONV_FP8 only: x * act_scale and w * w_scale (wpk packed as e4m3 bytes) are the
                                 MFMA operands, saturating at +-448; the accumulator is divided by their product */
  /* accumulator mode of the InstanceNorm statistics (no ramp_in_stats_finalize launch between two layers): the layer
   * ADDS its per-workgroup partial sums, as exact 2^-20 fixed-point integers, to acc_out [RAMP_IN_ACC_R][Cout][2] uint64
   * (ZERO before the layer; order independent, so reproducible), and takes its input's normalisation from acc_in
   * [RAMP_IN_ACC_R][Cin][2] (+ the pixel count and eps of that InstanceNorm) instead of pre_scale / pre_shift: every
   * consumer workgroup sums the replicas and forms scale = rsqrt(var + eps), shift = -mean * scale itself (Cin <= 128) */
  void *acc_out;
  const void *acc_in;
  float in_count, in_eps;
  /* two-source input (half in): channels [c0, Cin) come from x2 [H][W][Cin - c0], x is [H][W][c0] -- the MultiScale
   * towers' torch.cat((x, x_down2), dim=1) (ramp/extractor.py:300, 306) without the copy; c0 and Cin - c0 multiples of
   * 8, no input normalisation.  x2 = NULL: one source                                                              */
  const void *x2;
  int32_t c0;
  /* fused residual-block tail (half in, accumulator mode; ramp/extractor.py:49-57): with `skip` [H][W][Cin] the input
   * pixel is relu(relu(x * scale + shift) + skip') -- x normalised through acc_in, skip' = skip, or skip normalised through
   * acc_skip (+ skip_count, skip_eps), or the fp16-rounded relu of that (skip_relu) -- rounded to fp16 once: what
   * ramp_norm_add_relu_f16_acc writes, without the launch.  `mat` [H][W][Cin] (stride-1 layers): the pixels are also
   * written out, each by the workgroup that owns it, for the block that takes them as its skip.  NULL: a plain input  */
  const void *skip;
  const void *acc_skip;
  float skip_count, skip_eps;
  int32_t skip_relu;
  void *mat;
} ramp_conv_job;
#define RAMP_IN_ACC_R 8
int ramp_conv2d_nhwc_multi(const ramp_conv_job *jobs, int njobs, int H, int W, int Cin, int KH, int stride,
                           int dtype, void *stream);

/* dtype of ramp_conv2d_nhwc: RAMP_F32 (fp32 in/out, exact fp32 MFMA), RAMP_F16 (half in/out, fp16
 * MFMA, fp32 accumulation / bias / statistics; Cin % 32 == 0) or RAMP_F16|RAMP_IN_F32 (fp32 in,
 * half out: the first layer of the mixed-precision tower, Cin == 16)                            */
#define RAMP_IN_F32 0x10
/* fp16 only: use the direct (one global round trip per tap) kernel instead of the LDS-tiled one;
 * both give identical results (kept for the A/B test)                                          */
#define RAMP_CONV_DIRECT 0x20
/* ramp_conv2d_nhwc_multi only, with RAMP_F16 (half in / out): the layer's products run on the fp8 MFMA
 * (v_mfma_f32_16x16x32_fp8_fp8, OCP e4m3 operands, fp32 accumulate) -- BASELINE configs[4]'s "fp16 encoder on fp8
 * MFMA"; weights packed as e4m3 fragments (rampvo_amd/conv_hip.py::pack_conv_weight mode "f8")                    */
#define RAMP_CONV_FP8 0x40

/* ramp_conv2d_nhwc only, as RAMP_F32 | RAMP_CONV_X3: fp32 in / out at fp32 accuracy on the f16 matrix cores (every operand split
 * into two fp16 numbers, three MFMA products into one fp32 accumulator: csrc/conv.hip::conv_x3_kernel); weights packed by
 * rampvo_amd/conv_hip.py::pack_conv_weight mode "x3"; layer shapes: 3x3 stride 1 (32|64 -> 32|64), 3x3 stride 2 (32 -> 64),
 * 7x7 stride 2 (16 -> 32), else RAMP_EUNSUPPORTED (ramp_conv2d_stats_blocks says so first)                                   */
#define RAMP_CONV_X3 0x80

/* number of per-block partials ramp_conv2d_nhwc writes to `stats` for this layer shape / dtype   */
int ramp_conv2d_stats_blocks(int H, int W, int Cin, int Cout, int KH, int stride, int dtype);

/* InstanceNorm2d statistics (affine=False, biased variance): scale = rsqrt(var+eps),
 * shift = -mean*scale, from the per-block partials of ramp_conv2d_nhwc                         */
int ramp_in_stats_finalize(const float *partial, int nblk, int C, float count, float eps, float *scale,
                           float *shift, void *stream);

/* accumulator-mode counterparts: ramp_norm_add_relu_f16 with y's (and a normalised skip's) statistics taken from their
 * accumulators, and the (scale, shift) arrays of an accumulator for a consumer without that path                      */
int ramp_norm_add_relu_f16_acc(const void *y, const void *acc_y, float count_y, float eps_y, const void *skip,
                               const void *acc_skip, float count_skip, float eps_skip, void *out, long n, int C,
                               int skip_relu, void *stream);
int ramp_in_acc_finalize(const void *acc, int C, float count, float eps, float *scale, float *shift, void *stream);

/* out = relu(x*s + h)   (InstanceNorm + ReLU materialised where a skip connection needs it)    */
int ramp_affine_relu(const float *x, const float *s, const float *h, float *out, long n, int C,
                     void *stream);

/* half-storage variants (x, y, skip, out are half; scale/shift stay fp32) */
int ramp_affine_relu_f16(const void *x, const float *s, const float *h, void *out, long n, int C,
                         void *stream);
/* skip_relu: the skip operand is relu(skip*ss + hs) (an InstanceNorm + ReLU that was never materialised) */
int ramp_norm_add_relu_f16(const void *y, const float *sy, const float *hy, const void *skip,
                           const float *ss, const float *hs, void *out, long n, int C, int skip_relu,
                           void *stream);

/* residual-block tail: out = relu( skip' + relu(y*sy + hy) ), skip' = skip*ss + hs if ss else skip
 * (ramp/extractor.py:49-57 with the norms folded in)                                           */
int ramp_norm_add_relu(const float *y, const float *sy, const float *hy, const float *skip,
                       const float *ss, const float *hs, float *out, long n, int C, void *stream);

/* ---------------------------------------------------------- update operator */
/* Row-fused glue of the update operator (ramp/net.py:69-90, ramp/blocks.py:15-50); rows are
 * [E][384].  `dtype` is the GEMM I/O dtype T (RAMP_F16 under MIXED_PRECISION); the hidden state
 * is fp32 as under the reference's autocast.
 *
 * ramp_upd_row_fuse: t = A[rowA(e)] + B[rowB(e)] + C[rowC(e)]; optional LayerNorm(ln_w, ln_b, eps)
 *   (nn.LayerNorm(384, eps=1e-3), net.py:45,49-52,60) and ReLU; written as fp32 (out_f32) and/or
 *   T (out_t).  rowB(e) = idxB[e] (int64) or idxB32[e] or e, taken modulo modB when modB > 0
 *   (the `kk % (M*mem)` ring-buffer gather of Ramp_vo.py:282); rowC likewise.  rowA(e) = idxA[e] or e;
 *   idxA[e] < 0 reads a zero row: the tracker keeps the hidden state of the PREVIOUS graph and maps
 *   the current edges into it (removed edges drop out, new edges start at zero: Ramp_vo.py:204-205,
 *   268-270) instead of compacting / growing the [E,384] state every frame.  A fp32, B/C of T. */
int ramp_upd_row_fuse(const float *A, const int64_t *idxA, const void *B, const void *C, const int64_t *idxB,
                      const int32_t *idxB32, long modB, const int64_t *idxC, const int32_t *idxC32,
                      const float *ln_w, const float *ln_b, float eps, int relu, float *out_f32,
                      void *out_t, int E, int dtype, void *stream);
/* out[e] = idx[e] >= 0 ? X[idx[e]] : 0  -- `mask_ix * net[:, ix]` of net.py:78-82 (X fp32, out T) */
int ramp_upd_gather_mask(const float *X, const int64_t *idx, void *out, int E, int dtype, void *stream);
/* GatedResidual tail (blocks.py:30-31): t = X + sigmoid(G) * R, optional LayerNorm; written as
 * fp32, T and ReLU(T) (each optional) */
int ramp_upd_gated(const float *X, const void *G, const void *R, const float *ln_w, const float *ln_b,
                   float eps, float *out_f32, void *out_t, void *out_relu_t, int E, int dtype,
                   void *stream);
/* heads + Ramp_vo.update's target / filter_features (net.py:64-67, Ramp_vo.py:288-294,
 * utils.py:557-570): hw [E][4] = (delta_x, delta_y, w_x, w_y) pre-sigmoid, coords [E][2][P][P];
 * target = centre + delta, weight = sigmoid(w) zeroed outside [0,wd]x[0,ht]; delta optional    */
int ramp_upd_heads(const void *hw, const float *coords, float *target, float *weight, float *delta,
                   int E, int P, float wd, float ht, int dtype, void *stream);
/* SoftAgg core over the stacked [f | g] GEMM output fg [E][768] (single-pass online softmax):
 * y[g] = sum_{e in g} softmax_e(g[e]) * f[e]  (blocks.py:44-45)                                */
int ramp_upd_segment_softmax(const void *fg, const int32_t *order, const int32_t *seg_start,
                             const int32_t *ngroups, void *y, int max_groups, int dtype, void *stream);

/* HOST helper (pointers are host memory, nothing is launched): the factor-graph edit of
 * Ramp_vo.keyframe() (ramp/Ramp_vo.py:247-274 + remove_factors :203-208) for one outcome of the motion
 * test, one pass over the host mirror.  k_remove < 0: no keyframe is dropped.  out [4][cap] int64 =
 * (ii, jj, kk, hidden-state row) of the factors kept; rows_in NULL = identity.  Returns their number.  */
int ramp_graph_edit_host(const int64_t *ii, const int64_t *jj, const int64_t *kk, const int64_t *rows_in, int E,
                         int M, int k_remove, int n_after, int removal_window, int64_t *out, int cap,
                         int64_t *ranges /* optional [4]: min/max kk, min/max frame index of the kept factors */);

/* ------------------------------------------------ fused update-operator GEMM chains (fp16) */
/* gru[1..3] of the update operator (ramp/net.py:49-54; GatedResidual: ramp/blocks.py:15-31) as ONE
 * launch: x -> x + sigmoid(Wg x) * W2 relu(W1 x) -> LayerNorm -> the same again, 6 Linear layers with the
 * 64-row activation tile resident in LDS, fp16 MFMA / fp32 accumulate.
 *   x32 [E][384] fp32 (output of gru[0], the caller's LayerNorm);  out32 [E][384] fp32;
 *   relu_t [E][384] fp16 = relu(out32), the heads' input
 *   wp_host[6]: fp16 weights of (g1.gate, g1.res[0], g1.res[2], g2.gate, g2.res[0], g2.res[2]) packed in
 *   MFMA fragment order [K/32][N/16][64 lanes][8] (lane (q, j): W[16 nt + j][32 ks + 8 q ..]);
 *   bias_host[6]: fp32 [384] each (host arrays of device pointers); ln_w, ln_b, eps: gru[2].
 *   add_t != NULL: the kernel first forms x = LayerNorm(x32 + add_t[add_idx[e]]; pre_w, pre_b, pre_eps) itself
 *   -- the expand-and-add of the second SoftAgg (ramp/net.py:85) and gru[0] -- add_t [groups][384] fp16,
 *   add_idx [E] int32; NULL: x32 is already the output of gru[0].                                       */
int ramp_upd_gru(const float *x32, const void *add_t, const int32_t *add_idx, const float *pre_w, const float *pre_b,
                 float pre_eps, const void *const *wp_host, const float *const *bias_host, const float *ln_w,
                 const float *ln_b, float eps, float *out32, void *relu_t, int E, void *stream);
/* ramp_upd_gru with the two heads and target / weight formed from the result tile in the same launch
 * (ramp/net.py:87-90 `d`, `w`; ramp/Ramp_vo.py:291-297: target = centre + delta, weight = sigmoid, zero outside the
 * image): heads_w [4][384] fp16 (d.weight rows 0..1, w.weight rows 0..1), heads_b [4], coords [E][2][P][P],
 * target / weight [E][2] fp32.  The heads' fp16 input relu(out32) is not written.  Same arithmetic as
 * ramp_upd_heads_linear up to the order of the 384-term sums.                                              */
int ramp_upd_gru_heads(const float *x32, const void *add_t, const int32_t *add_idx, const float *pre_w, const float *pre_b,
                       float pre_eps, const void *const *wp_host, const float *const *bias_host, const float *ln_w,
                       const float *ln_b, float eps, float *out32, const void *heads_w, const float *heads_b,
                       const float *coords, float *target, float *weight, int E, int P, float wd, float ht, void *stream);
size_t ramp_upd_mlp_lds_bytes(void);

/* net_out[e] = net_in[e] + Lb(relu(La(idx[e] >= 0 ? net_in[idx[e]] : 0)))  -- the temporal-neighbour MLPs
 * c1 / c2 (ramp/net.py:43-46, 77-82) with the gather, both Linear layers and the residual add in one
 * launch.  net_in != net_out (other workgroups gather from net_in); out_t: optional fp16 copy of
 * net_out.  wa / wb packed like ramp_upd_gru's weights, ba / bb fp32 [384].                          */
int ramp_upd_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb,
                 const float *bb, float *net_out, void *out_t, int E, void *stream);

/* The whole correlation MLP and Update.norm in one launch (ramp/net.py:57-62, 71-74):
 *   c = Linear3(relu(LayerNorm(Linear2(relu(Linear1(corr))))));  net_out = LayerNorm_norm((net[net_map] + inp[inp_idx % inp_mod]) + c)
 * corr [E][corr_k] fp16 (corr_k a multiple of 32: 896 = 882 + zero padding); w1 packed [corr_k/32][24][64][8] from the
 * zero-padded weight, w2 / w3 packed like ramp_upd_gru's weights, biases fp32 [384] (fp16-rounded values); net [*][384] fp32 or
 * NULL (zeros), net_map [E] (-1: zero row) or NULL (identity); inp [*][384] fp16, inp_idx NULL = identity.  Linear outputs
 * are rounded to fp16 where the reference's autocast makes them half tensors.                                             */
int ramp_upd_corr_mlp(const void *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                      const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps,
                      const float *net, const int64_t *net_map, const void *inp, const int64_t *inp_idx, long inp_mod,
                      const float *norm_w, const float *norm_b, float norm_eps, float *net_out, int E, void *stream);

/* fp16 path: the two heads' Linear layers (ramp/net.py:64-66) + ramp_upd_heads's epilogue in one launch:
 * hw = relu_t [E][384] @ heads_w [4][384]^T + heads_b (rounded to fp16 like the GEMM's output), then target = patch
 * centre of coords + hw[:2], weight = sigmoid(hw[2:]) zeroed where target is outside [0,wd]x[0,ht].     */
int ramp_upd_heads_linear(const void *relu_t, const void *heads_w, const float *heads_b, const float *coords,
                          float *target, float *weight, int E, int P, float wd, float ht, void *stream);

/* SoftAgg front half (ramp/blocks.py:42-46) in one launch: x = x32[e] (+ add_t[add_idx[e]], written to x32_out
 * when given; x32_out may be x32);  fg[e] = [ f(x) | g(x) ]  fp16 [E][768].  wf / wg packed like ramp_upd_gru's
 * weights, bf / bg fp32 [384].                                                                        */
int ramp_upd_fg(const float *x32, const void *add_t, const int32_t *add_idx, float *x32_out, const void *wf,
                const float *bf, const void *wg, const float *bg, void *fg, int E, void *stream);

/* nn.Linear(384, 384) on a small fp16 table: y[r] = fp16(x[r] W^T + b) -- SoftAgg's `h` layer on the group table
 * (ramp/blocks.py:46-47), the one GEMM of the fused update operator that used to be a library call.  w_packed like
 * ramp_upd_gru's weights, bias fp32 [384]; rows_dev (optional, device int32): only rows < *rows_dev are computed.   */
int ramp_upd_linear(const void *x, const void *w_packed, const float *bias, void *y, int rows, const int32_t *rows_dev,
                    void *stream);

/* SoftAgg (ramp/blocks.py:33-50: y = scatter_sum(f(x) * scatter_softmax(g(x))), h(y)) of the fp16 path WITHOUT the
 * [E, 768] rows of [f | g] going through memory: ramp_upd_softagg walks the grouping's sorted factor list (`order`,
 * groups = contiguous runs; `gid`: factor -> group), 80 positions per workgroup, forms g and f on the same tile and leaves
 * one fragment (running maximum, sum, weighted sum)[384] per run of a group inside a 20-position block in
 * frag[group + position / 20] (ramp_upd_softagg_frag_rows(E, max_groups) rows of 3 x 384 floats); ramp_upd_softagg_finish
 * merges a group's fragments in slot order (seg_start = the grouping's segment starts), y = a / z, and applies h:
 * hy [max_groups][384] fp16 (rows >= *ngroups are zero).  x = x32 (+ add_t[add_idx]); weights as ramp_upd_fg /
 * ramp_upd_linear.  Replaces ramp_upd_fg + ramp_upd_segment_softmax + ramp_upd_linear (same values up to the order of
 * the fp32 additions).                                                                                             */
size_t ramp_upd_softagg_frag_rows(int E, int max_groups);
int ramp_upd_softagg(const float *x32, const void *add_t, const int32_t *add_idx, const int32_t *order,
                     const int32_t *gid, const void *wf, const float *bf, const void *wg, const float *bg, float *frag,
                     int E, void *stream);
int ramp_upd_softagg_finish(const float *frag, const int32_t *seg_start, const int32_t *ngroups, const void *wh,
                            const float *bh, void *hy, int max_groups, void *stream);

/* ---------------------------------------------------------------- the update operator at fp32 accuracy (MIXED_PRECISION off)
 *
 * The same chains for fp32 features (ramp/net.py:69-90 without autocast): every Linear layer is formed on the f16 matrix
 * cores from split operands (x = xh + xl, three f16 products into one fp32 accumulator: csrc/update_x3.hip), nothing
 * between layers is rounded to fp16; biases, LayerNorm, gate, residual stream, the SoftAgg tables and the heads are fp32.
 * Agreement with an fp32 GEMM chain: ~1e-6 of the output scale (22-bit operands, fp32 accumulation).  All tables and
 * rows are float; weights `*w*` are "x3 packs" of an nn.Linear weight W [384][K] (K a multiple of 32):
 *     [K/32][24][2][64 lanes][8] fp16 -- plane 0 = fp16(W 2^s), plane 1 = fp16(W 2^s - plane 0), s the power of two that
 *     puts max |W| into [2^12, 2^13); lane (q, j) of fragment (ks, nt) holds W[16 nt + j][32 ks + 8 q .. + 8];
 *     followed by one float, 2^-s   (rampvo_amd/update_fused.py::pack_linear_x3).
 * Linear inputs must stay below 65504 in magnitude (as on the MIXED_PRECISION path).  Arguments otherwise as the
 * ramp_upd_* function of the same name.                                                                              */
int ramp_x3_corr_mlp(const float *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                     const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps, const float *net,
                     const int64_t *net_map, const float *inp, const int64_t *inp_idx, long inp_mod, const float *norm_w,
                     const float *norm_b, float norm_eps, float *net_out, int E, void *stream);
int ramp_x3_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb, const float *bb,
                float *net_out, int E, void *stream);
/* fg [E][768] fp32 = [f(x) | g(x)], x = x32 (+ add_t[add_idx], fp32 table; written to x32_out when given)              */
int ramp_x3_fg(const float *x32, const float *add_t, const int32_t *add_idx, float *x32_out, const void *wf, const float *bf,
               const void *wg, const float *bg, float *fg, int E, void *stream);
/* y [max_groups][384] fp32: the softmax-weighted segment sums of ramp_x3_fg's rows (ramp/blocks.py:46-48; 8 row lanes per
 * group, merged in lane order); rows >= *ngroups are zero                                                            */
int ramp_x3_segment_softmax(const float *fg, const int32_t *order, const int32_t *seg_start, const int32_t *ngroups, float *y,
                            int max_groups, void *stream);
/* y[r] = x[r] W^T + b on a table of rows (SoftAgg's `h`); rows_dev (optional, device int32): only rows < *rows_dev      */
int ramp_x3_linear(const float *x, const void *w_packed, const float *bias, float *y, int rows, const int32_t *rows_dev,
                   void *stream);
/* gru (LayerNorm, GatedResidual, LayerNorm, GatedResidual) with x = LayerNorm_pre((x32 (+ add0_t[add0_idx])) + add_t[add_idx])
 * when add_t is given; relu32 (optional) = relu(out32); heads_w [4][384] / heads_b [4] fp32 (optional): the d / w heads and
 * target / weight as ramp_upd_gru_heads, in fp32                                                                      */
int ramp_x3_gru(const float *x32, const float *add0_t, const int32_t *add0_idx, const float *add_t, const int32_t *add_idx,
                const float *pre_w, const float *pre_b, float pre_eps, const void *const *wp_host, const float *const *bias_host,
                const float *ln_w, const float *ln_b, float eps, float *out32, float *relu32, int E, const float *heads_w,
                const float *heads_b, const float *coords, float *target, float *weight, int P, float wd, float ht, void *stream);

/* ---------------------------------------------------------------- device-resident tracking step
 *
 * Ramp_vo.__call__ in steady state (ramp/Ramp_vo.py:327-410): the frame's state stores, update() (:276-310:
 * reproject, corr, the update operator, two BA iterations, point cloud) and keyframe() (:237-274: the motion test, the
 * removal of a keyframe and of old factors) followed by the NEXT frame's append_factors (:194-201, :312-325) -- as ONE
 * host call that never reads the device.  The reference decides keyframe() on the host (`.item()`), so every frame
 * waits for the GPU to drain; here the decision is taken by a kernel and everything that depends on it -- the row a
 * frame is stored to, the factor count, the optimisation window -- is read by the kernels from a block of int32 words
 * in device memory (`dyn`), their launch sizes being capacity bounds.  The host mirrors the state lazily (dyn_host).
 */
#define RAMP_DYN_WORDS 32
#define RAMP_DYN_N 0        /* keyframes incl. the newest frame: Ramp_vo.n during update() / keyframe(), BA's t1   */
#define RAMP_DYN_NROW 1     /* row the NEXT frame is stored to (Ramp_vo.n before its `n += 1`)                    */
#define RAMP_DYN_E 2        /* factors in the current graph (incl. the ones the newest frame added)               */
#define RAMP_DYN_KLO 3      /* lower bound of kk (offset of the counting group-by)                                */
#define RAMP_DYN_FLO 4      /* lowest frame index in ii / jj                                                      */
#define RAMP_DYN_W 5        /* pair keys are jj * W + ii                                                          */
#define RAMP_DYN_REMOVED 6  /* outcome of the last keyframe test: 1 = keyframe K was dropped                      */
#define RAMP_DYN_K 7        /* the keyframe that test looked at (n - KEYFRAME_INDEX)                              */
#define RAMP_DYN_NPREV 8    /* Ramp_vo.n when the test ran                                                        */
#define RAMP_DYN_EPREV 9    /* factors before the edit                                                            */
#define RAMP_DYN_EKEPT 10   /* factors the edit kept (next graph = kept ++ new)                                   */
#define RAMP_DYN_STATUS 11  /* sticky bits: 1 BA pose step dropped, 2 BA pair list overflow (ramp_ba_forward's
                               info), 4 factor capacity exceeded, 8 group-by key out of range, 16 delta log full,
                               32 a step's E_bound was below the live factor count, 64 frame buffers full (n_rows),
                               128 a gate wait (ramp_stream_wait_flag) timed out: the front end ran unordered       */
#define RAMP_DYN_NLOG 12    /* entries written to the delta log                                                   */
#define RAMP_DYN_FRAME 13   /* `counter` of the last frame stepped (tags the host's lazy copy)                    */
#define RAMP_DYN_MEDOK 14   /* 1: ramp_track.median holds the depth median of the three newest frames (set beside the
                             * motion test, cleared by an update-only step; 0 when the host hands the state over)     */
#define RAMP_DYN_FRAME2 31  /* = RAMP_DYN_FRAME, in the other half of the block: the two differ in a torn host copy         */
#define RAMP_TRACK_LOG 12   /* floats per delta-log entry: t1, t0 (as int32 bit patterns), dP[7], pad             */

#define RAMP_TRACK_COMMIT 1    /* store the front end's outputs as frame NROW first                               */
#define RAMP_TRACK_UPDATE 2    /* Ramp_vo.update()                                                                */
#define RAMP_TRACK_KEYFRAME 4  /* Ramp_vo.keyframe() + the next frame's append_factors + its plan                 */
#define RAMP_TRACK_MM_GIVEN 8  /* (tests) keyframe(): take the two flow magnitudes from t->mm instead of computing */
#define RAMP_TRACK_WRAP_COORDS 16 /* (measurement) move every reprojection into the target plane by whole plane sizes
                                     before the correlation launch: bench.py's roofline leg with every factor live    */
#define RAMP_TRACK_UPDATE_PRE 64   /* the part of update() in front of the update operator: reprojection + correlation        */
#define RAMP_TRACK_UPDATE_POST 128 /* the part behind it: two BA iterations + point cloud, from t->target / t->weight.  With
                                      PRE and POST as two calls the caller runs the operator in between on t->coords / t->corr
                                      and leaves the new hidden state in t->net[0], target / weight in their buffers -- the
                                      fp32 path (MIXED_PRECISION off), whose Linear layers are library GEMMs; it still never
                                      reads the device (launch sizes = E_bound)                                          */
#define RAMP_TRACK_COMPACT_COORDS 32 /* (measurement) the same, and every patch with unit pixel spacing around its centre (a
                                     converged tracker's factors: one 10 x 10 union window per level)                  */

typedef struct ramp_track_weights {      /* update operator, fp16 fused formats of ramp_upd_* */
  const void *corr_w1, *corr_w2, *corr_w3;
  const float *corr_b1, *corr_b2, *corr_b3, *corr_ln_w, *corr_ln_b, *norm_w, *norm_b;
  const void *c1_wa, *c1_wb, *c2_wa, *c2_wb;
  const float *c1_ba, *c1_bb, *c2_ba, *c2_bb;
  const void *kk_wf, *kk_wg, *kk_wh, *ij_wf, *ij_wg, *ij_wh;
  const float *kk_bf, *kk_bg, *kk_bh, *ij_bf, *ij_bg, *ij_bh;
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
  const void *gru_w[6];
  const float *gru_b[6];
  const void *heads_w;
  const float *heads_b;
  float corr_ln_eps, norm_eps, ln1_eps, ln2_eps;
} ramp_track_weights;

typedef struct ramp_track {
  /* configuration (cfg: PATCHES_PER_FRAME, PATCH_LIFETIME, REMOVAL_WINDOW, OPTIMIZATION_WINDOW, KEYFRAME_INDEX,
   * KEYFRAME_THRESH, MOTION_MODEL (1 DAMPED_LINEAR, 2 copy), MOTION_DAMPING) */
  int M, P, mem, n_rows, patch_lifetime, removal_window, opt_window, keyframe_index, motion_model;
  int feat_h, feat_w;                 /* level-1 feature plane (level 4 is feat_h/4 x feat_w/4)                  */
  int E_cap, kk_cap, ij_cap, kkey_cap, pkey_cap, log_cap, m_cap;
  float motion_damping, pad0;
  double keyframe_thresh;
  /* tracker state (ramp/Ramp_vo.py:54-100), layouts as in rampvo_amd/Ramp_vo.py */
  int32_t *dyn;                       /* [RAMP_DYN_WORDS] */
  float *poses, *patches, *intrinsics, *points;
  int64_t *tstamps, *index_map;
  const int64_t *ixm;                 /* [n_rows * M]: patch -> source frame                                     */
  void *colors, *imap, *gmap, *fmap1, *fmap2;
  const float *lmbda;
  /* the front end's outputs of the frame being committed */
  const void *fe_colors, *fe_imap, *fe_gmap, *fe_fmap1, *fe_fmap2;
  float *fe_patches;
  /* factor graph, double buffered: [4][E_cap] int64 rows (ii, jj, kk, hidden-state row) */
  int64_t *graph[2];
  /* graph plan */
  int32_t *kk_order, *kk_gid, *kk_seg, *kk_ngroups, *ij_order, *ij_gid, *ij_seg, *ij_ngroups;
  int64_t *kk_ukeys, *ij_ukeys, *ix, *jx;
  int32_t *kj;                        /* [E_cap] factors in (kk, jj) order (a by-product of the neighbour search)  */
  void *plan_ws;                      /* ZERO before the first plan (each plan leaves its histograms cleared)      */
  size_t plan_ws_bytes;
  /* update operator */
  ramp_track_weights w;
  float *coords;                      /* [E_cap][2][P][P] */
  void *corr;                         /* [E_cap][896] fp16 */
  float *net[3];                      /* [E_cap][384] fp32: [0] the hidden state (in: previous, out: new), [1], [2] scratch */
  void *fg, *ykk, *hkk, *yij, *hij, *relu_t;      /* (relu_t: unused since the heads moved into the gru launch) */
  float *sagg_frag;                   /* optional [ramp_upd_softagg_frag_rows(E_cap, max(kk_cap, ij_cap))][3][384]: with it the
                                       * two SoftAggs run as ramp_upd_softagg + _finish (fg / ykk / yij are then unused)  */
  float *target, *weight;             /* [E_cap][2] */
  /* bundle adjustment */
  void *ba_ws;
  size_t ba_ws_bytes;
  /* keyframe() */
  float *mm;                          /* [2] flow magnitudes of the motion test */
  float *median;                      /* optional [1]: depth initialisation of the next frame (ramp/Ramp_vo.py:370-371), computed in
                                       * the motion test's launch instead of at the head of the next step (KEYFRAME_INDEX >= 4:
                                       * the three newest frames are the same whether or not the test drops a keyframe)       */
  float *dlog;                        /* [log_cap][RAMP_TRACK_LOG] */
  int32_t *edit_ws;                   /* [3 * ceil(E_cap / 256) + 8]  */
  int32_t *dyn_host;                  /* optional pinned host copy of dyn, refreshed asynchronously after each step */
  int32_t *dyn_host_dev;              /* optional: device address of dyn_host (ramp_host_device_pointer, resolved once):
                                       * the plan's last launch then writes the copy itself                          */
  /* optional hipEvent_t handles recorded on `stream` by ramp_track_step (measurement only: bench.py's roofline legs):
   * [0] before / [1] after the correlation kernel, [2] after the update operator's last chain (gru), [3] before /
   * [4] after bundle adjustment                                                                                  */
  void *probe[5];
  int32_t E_hint;                     /* optional (> 0): the caller's estimate of the live factor count (E_bound is an upper
                                       * bound): picks the gru launch's tile (64 / 80 rows per workgroup)                    */
  uint32_t gate_seq;                  /* with gate_flag: the value the update operator's last launch (gru) stores into it  */
  int32_t feat_fp32;                  /* 0: fp16 features (imap / gmap / fmap rows of 2-byte elements, chunked [h][C/32][w][32] pyramid
                                       * planes, corr [E_cap][896] fp16); 2: fp32 features, chunked planes of split fp16 pairs
                                       * (RAMP_CORR_X2: corr_mfma_kernel<CorrX2>; feat_plain must be 0), everything else as 1;
                                       * 1: fp32 features, planes chunked as [h][C/16][w][16]
                                       * (feat_plain = 0) or plain NHWC, corr [E_cap][896] fp32 by corr_mfma_kernel<float>
                                       * (RAMP_CORR_F32_MFMA=0: corr_kernel<float>, the reference kernel's summation order, plain
                                       * planes only), operator csrc/update_x3.hip                                          */
  int32_t feat_plain;                 /* 1: the pyramid planes are plain NHWC [h][w][128] rows instead of the
                                       * chunked [h][C/32][w][32] layout -- feature planes whose width is no multiple of 16 or
                                       * whose height is no multiple of 4 (ramp_pyramid_pack's shapes): corr_mfma_kernel<half, false>  */
  uint32_t *gate_flag;                /* optional signal word (ramp_signal_alloc): "the next frame's front end may start"  *
                                       * without a packet on this stream -- the other stream waits with                    *
                                       * ramp_stream_wait_flag (a hipEventRecord costs ~5 us between two kernels of the   *
                                       * recording stream, tools/mb/stream_signal.hip); takes the place of gate_event     */
  int32_t *fmap1_slot;                /* optional [mem] int32, a permutation of 0 .. mem - 1 (identity at hand-over): ring row r of
                                       * the level-0 correlation planes lives in physical slot fmap1_slot[r] of `fmap1`.  A
                                       * dropped keyframe then rotates table entries instead of moving three 4.9 MB planes
                                       * (ramp/Ramp_vo.py:259-271 shifts them); every reader of `fmap1` rows (correlation, frame
                                       * commit, warm-up) goes through the table; the caller undoes the permutation when it takes
                                       * the buffers back.  NULL: rows are slots                                           */
} ramp_track;

/* cache warm-up for the next step's correlation kernel: reads the planes of the window's frames and the patch features
 * once (no effect on any value; `sink` [1] is never written in practice).  Meant for the front-end stream, in the slack
 * behind the front end.                                                                                            */
int ramp_track_warm(const ramp_track *t, int32_t *sink, void *stream);
/* holds `stream` back for about `microseconds` with one sleeping wave (frame pipelining: the next frame's encoder
 * should reach the chip a little behind the gru launch)                                         */
int ramp_stream_delay(int microseconds, void *stream);

/* Cross-stream "go" through a 32-bit word instead of an event: the producer is a KERNEL that stores a sequence number
 * (no packet between its stream's launches); the consumer stream waits with ONE sleeping wave that looks at the word
 * every ~3 us and ends when it is >= value, or after timeout_us (a producer that never comes must not hang the stream;
 * a time-out is an ERROR for whoever relied on the order: it is recorded in *status);
 * it then sleeps then_delay_us more.  (A hipStreamWaitValue32 in its place slows the other streams' launches down for
 * as long as it is pending, and so does a wave that polls without pauses: DESIGN.md section 8.0.)  The word lives in
 * signal memory (hipExtMallocWithFlags(hipMallocSignalMemory)), zero-initialised.                                    */
int ramp_signal_alloc(uint32_t **flag);
int ramp_signal_free(uint32_t *flag);
int ramp_stream_wait_flag(void *stream, const uint32_t *flag, uint32_t value, int timeout_us, int then_delay_us,
                          int32_t *status /* optional: bit 128 is ORed in when the wait times out */);
/* the producer side as a launch of its own (one thread storing `value`), for producers that are not this library's kernels */
int ramp_stream_signal(void *stream, uint32_t *flag, uint32_t value);
/* device address of a pinned (mapped) host allocation */
int ramp_host_device_pointer(void *host, void **dev);

size_t ramp_track_sizeof(void);      /* sizeof(ramp_track): lets a binding check its mirror of the struct */
size_t ramp_track_plan_workspace_bytes(int E_cap, int kkey_cap, int pkey_cap);
size_t ramp_track_ba_workspace_bytes(int E_cap, int n_rows, int M, int opt_window, int kk_cap, int ij_cap);

/* plan (groupings + temporal neighbours) of graph[cur] with the sizes in dyn: needed once, when the host hands a graph
 * over; afterwards every step builds the next one                                                                  */
int ramp_track_plan(const ramp_track *t, int cur, void *stream);

/* One tracked frame (flags = COMMIT | UPDATE | KEYFRAME), a bare update() (flags = UPDATE), ...; `cur` = which half of
 * graph[] holds the current graph (KEYFRAME writes the next one to 1 - cur).  k_new: optional device [4] intrinsics at
 * feature resolution when they differ from the previous frame's.  gate_event: optional hipEvent_t recorded before the
 * last kernel of the update operator (where the next frame's front end may start on another stream).  E_bound: the
 * caller's upper bound of the current factor count (from its lazy copy of dyn; 0 = E_cap) -- sizes the per-factor
 * launches; a bound below the live count raises status bit 32 instead of truncating silently.                      */
int ramp_track_step(const ramp_track *t, int cur, int64_t counter, int flags, int E_bound, const float *k_new,
                    void *gate_event, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RAMP_HIP_H */
