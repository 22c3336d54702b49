// internal (non-exported) helpers shared between translation units
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/ramp_hip.h"

size_t ramp_internal_group_by_ws(int E);
int ramp_internal_group_by(const int64_t *keys, int E, int64_t key_bound, int32_t *order,
                           int32_t *gid, int32_t *seg_start, int64_t *ukeys, int32_t *ngroups,
                           void *ws, size_t ws_bytes, hipStream_t st);

// ---- launchers with device-side sizes (the RAMP_DYN_* block of include/ramp_hip.h), used by csrc/track.hip: the
// integer size argument is the launch bound, the live count is read by the kernel from `dyn`
int ramp_i_transform_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ii,
                         const int64_t *jj, const int64_t *kk, float *out, int E_cap, const int32_t *dyn,
                         hipStream_t st);
int ramp_i_point_cloud_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ix,
                           float *out, int m_cap, const int32_t *dyn, int M, hipStream_t st);
int ramp_i_motionmag_point_cloud_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ii,
                                     const int64_t *jj, const int64_t *kk, const int32_t *order, const int32_t *seg,
                                     const int64_t *ukeys, const int32_t *ngroups, float beta, float *out2,
                                     int32_t *dyn, int keyframe_index, const int64_t *ix, float *points, int m_cap,
                                     int M, float *median, hipStream_t st);
int ramp_i_motionmag_dyn(const float *poses, const float *patches, const float *intrinsics, const int64_t *ii,
                         const int64_t *jj, const int64_t *kk, const int32_t *order, const int32_t *seg,
                         const int64_t *ukeys, const int32_t *ngroups, float beta, float *out2, const int32_t *dyn,
                         int keyframe_index, hipStream_t st);
int ramp_i_frame_commit_dyn(float *poses, int motion, float damping, int64_t *tstamps, int64_t counter,
                            int64_t *index_map, float *intrinsics, const float *k_new, float *patches_state,
                            int median_frames, int M, int P, float *patches_new, int n_copy, const void *const *src,
                            void *const *base, const long *bytes, const int *mod, const int32_t *dyn,
                            const float *median_ahead, int32_t *status, int E_bound, int n_rows, hipStream_t st,
                            const int32_t *slot_tab = nullptr, int slot_buf = 0, uint32_t *signal = nullptr,
                            uint32_t signal_val = 0);
size_t ramp_i_plan_dyn_ws(int E_cap, int kkey_cap, int pkey_cap);
int ramp_i_plan_dyn(const int64_t *g4, int E_cap, int E_grid, const int32_t *dyn, int32_t *status, int M, int kkey_cap,
                    int pkey_cap, int kk_cap, int ij_cap, int32_t *kk_order, int32_t *kk_gid, int32_t *kk_seg,
                    int32_t *kk_ngroups, int64_t *kk_ukeys, int32_t *ij_order, int32_t *ij_gid, int32_t *ij_seg,
                    int32_t *ij_ngroups, int64_t *ij_ukeys, int64_t *ix, int64_t *jx, int32_t *kj, void *ws,
                    size_t ws_bytes, int32_t *mirror, hipStream_t st);
size_t ramp_i_ba_dyn_ws(int E_cap, int n_poses, int n_patches, int opt_window, int max_patches, int max_pairs);
int ramp_i_ba_dyn(float *poses, float *patches, const float *intrinsics, const float *target, const float *weight,
                  const float *lmbda, const int64_t *ii, const int64_t *jj, const int64_t *kk, int E_cap, int P,
                  int n_poses, int n_patches, int opt_window, int iterations, const int32_t *order_k,
                  const int32_t *seg_k, const int32_t *ngroups_k, const int64_t *ukeys_k, int max_patches,
                  const int32_t *order_p, const int32_t *seg_p, const int32_t *ngroups_p, int max_pairs, void *ws,
                  size_t ws_bytes, int32_t *info, const int32_t *dyn, hipStream_t st);
extern "C" {
int ramp_i_corr_fwd(const void *fmap1, const ramp_corr_level *levels, int nlevels, const float *coords,
                    const int64_t *ii, const int64_t *jj, const int32_t *order, void *out, int out_row_elems,
                    long mod_ii, long mod_jj, int E, int N1, int N2, int C, int P, int radius, int dtype, int layout,
                    const int32_t *dyn, void *stream, const float *tf_poses = nullptr, const float *tf_patches = nullptr,
                    const float *tf_intr = nullptr, const int64_t *tf_src = nullptr, const int32_t *slot0 = nullptr);
int ramp_i_upd_gru(const float *x32, const void *add0_t, const int32_t *add0_idx, const void *add_t, const int32_t *add_idx,
                 const float *pre_w, const float *pre_b,
                   float pre_eps, const void *const *wp_host, const float *const *bias_host, const float *ln_w,
                   const float *ln_b, float eps, float *out32, void *relu_t, int E, const int32_t *dyn,
                   const void *heads_w, const float *heads_b, const float *coords, float *target, float *weight, int P,
                   float wd, float ht, int E_hint, uint32_t *gate_flag, uint32_t gate_seq, void *stream);
int ramp_i_upd_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb,
                   const float *bb, float *net_out, void *out_t, int E, const int32_t *dyn, void *stream);
int ramp_i_upd_corr_mlp(const void *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                        const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps,
                        const float *net, const int64_t *net_map, const void *inp, const int64_t *inp_idx, long inp_mod,
                        const float *norm_w, const float *norm_b, float norm_eps, float *net_out, int E,
                        const int32_t *dyn, void *stream);
int ramp_i_upd_softagg(const float *x32, const void *add_t, const int32_t *add_idx, const int32_t *order, const int32_t *gid,
                       const void *wf, const float *bf, const void *wg, const float *bg, float *frag, int E,
                       const int32_t *dyn, void *stream, uint32_t *gate_flag = nullptr, uint32_t gate_seq = 0);
int ramp_i_upd_fg(const float *x32, const void *add_t, const int32_t *add_idx, float *x32_out, const void *wf,
                  const float *bf, const void *wg, const float *bg, void *fg, int E, const int32_t *dyn, void *stream);
int ramp_i_upd_heads_linear(const void *relu_t, const void *heads_w, const float *heads_b, const float *coords,
                            float *target, float *weight, int E, int P, float wd, float ht, const int32_t *dyn,
                            void *stream);
int ramp_i_x3_nbr(const float *net_in, const int64_t *idx, const void *wa, const float *ba, const void *wb, const float *bb,
                  float *net_out, int E, const int32_t *dyn, void *stream);
int ramp_i_x3_corr_mlp(const float *corr, int corr_k, const void *w1, const float *b1, const void *w2, const float *b2,
                       const void *w3, const float *b3, const float *ln_w, const float *ln_b, float ln_eps, const float *net,
                       const int64_t *net_map, const float *inp, const int64_t *inp_idx, long inp_mod, const float *norm_w,
                       const float *norm_b, float norm_eps, float *net_out, int E, const int32_t *dyn, void *stream);
int ramp_i_x3_fg(const float *x32, const float *add_t, const int32_t *add_idx, float *x32_out, const void *wf, const float *bf,
                 const void *wg, const float *bg, float *fg, int E, const int32_t *dyn, void *stream);
int ramp_i_x3_gru(const float *x32, const float *add0_t, const int32_t *add0_idx, const float *add_t, const int32_t *add_idx,
                  const float *pre_w, const float *pre_b, float pre_eps, const void *const *wp_host,
                  const float *const *bias_host, const float *ln_w, const float *ln_b, float eps, float *out32, float *relu32,
                  int E, const int32_t *dyn, const float *heads_w, const float *heads_b, const float *coords, float *target,
                  float *weight, int P, float wd, float ht, uint32_t *gate_flag, uint32_t gate_seq, void *stream);
}
