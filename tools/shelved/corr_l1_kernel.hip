// SHELVED (round 6): correlation + the correlation MLP's first Linear in one launch (first clause).
// Built in round 5, bit-identical to corr + upd_corr_mlp, measured slower (profiles/r05_corr_l1_ab.txt);
// cut out of csrc/altcorr.hip (it used that file's corr_edge helpers and CorrParams) -- kept for the record, not compiled.
// ------------------------------------------------- correlation + the first Linear of the correlation MLP
// "corr-MLP first layer fused with corr output (avoid materialising [E,882])": a workgroup owns 16 consecutive positions
// of the (jj, ii)-major schedule, its NWV waves compute the 16 factors' correlation rows (16 / NWV each, one after the
// other, the per-edge code of corr_mfma_kernel) into an LDS tile [16][896] fp16 -- the values the unfused kernel would
// have stored -- and the tile is multiplied by Linear1 (ramp/net.py:60, corr[0]: 882 -> 384, K padded to 896) on
// v_mfma_f32_16x16x32_f16 in the operand order and K order of upd_corr_tail_kernel<true>, so that
// c1 = relu(Linear1(corr) + b1) [E][384] fp16 is bit-identical to what that kernel forms in its tile.  The [E,896] tensor
// (75 MB written + read back per update at 40k factors) is never formed; upd_corr_tail_kernel<false> continues from c1.
// What it costs: Linear1's 688 KB of packed weights are streamed once per SIXTEEN factors here (43 KB per factor through
// the vector L1, as much again as the window gathers) against once per 64 in the unfused pair, and a 16-row tile leaves
// room for two 4-wave workgroups per CU where the unfused kernel keeps 16 single-wave workgroups resident.  Measured on
// MI355X: DESIGN.md section 8 (round 5).  Switch: RAMP_CORR_L1=1 (tracker), ramp_corr_l1_fwd_ordered (C ABI).
struct CorrL1Params {
  CorrParams c;
  const _Float16 *w1;     // packed fragments [corr_k / 32][24][64][8] (update_fused.pack_linear_f16)
  const float *b1;        // [384] (fp16-rounded values as fp32)
  _Float16 *c1;           // [E][384]
  int nks;                // corr_k / 32
};
constexpr int CL1_ROWS = 16, CL1_K = 896, CL1_RXS = CL1_K + 8, CL1_N = 384;

template <int NWV, bool CHUNKED>
__global__ void __launch_bounds__(64 * NWV) corr_l1_kernel(const CorrL1Params p) {
  typedef _Float16 T;
  typedef f16x8_t h8;
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  constexpr int PP = 9, d = 7, NOUT = d * d * PP, KOUT = d, EPW = CL1_ROWS / NWV, NT = (CL1_N / 16) / NWV;
  static_assert(CL1_ROWS % NWV == 0 && (CL1_N / 16) % NWV == 0, "16 rows and 24 column tiles over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char cl1_smem[];
  _Float16 *Rs = reinterpret_cast<_Float16 *>(cl1_smem);                                  // [16][CL1_RXS]
  float *Cs_all = reinterpret_cast<float *>(cl1_smem + (size_t)CL1_ROWS * CL1_RXS * 2);     // [NWV][9 CORR_TM]
  float *outs_all = Cs_all + (size_t)NWV * PP * CORR_TM;                                    // [NWV][441]
  int *s_edge = reinterpret_cast<int *>(outs_all + (size_t)NWV * NOUT);                     // [16]
  const CorrParams &prm = p.c;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int E = prm.dyn ? prm.dyn[RAMP_DYN_E] : prm.E;
  const int ngroups = (E + CL1_ROWS - 1) / CL1_ROWS, chunkg = (ngroups + CORR_XCDS - 1) / CORR_XCDS;
  const int b = blockIdx.x;
  if (b / CORR_XCDS >= chunkg) return;
  const int gid = (b % CORR_XCDS) * chunkg + b / CORR_XCDS;      // consecutive positions stay on one XCD (corr_edge_of_block)
  if (gid >= ngroups) return;
  float *Cs = Cs_all + (size_t)wave * PP * CORR_TM, *outs = outs_all + (size_t)wave * NOUT;
#pragma unroll 1
  for (int k = 0; k < EPW; k++) {
    const int r = wave * EPW + k, pos = gid * CL1_ROWS + r;
    const int e = pos < E ? (prm.order ? prm.order[pos] : pos) : -1;                       // (wave-uniform)
    if (lane == 0) s_edge[r] = e;
    _Float16 *row = Rs + r * CL1_RXS;
    if (e >= 0) {
      float res[CORR_MAXLEV][KOUT];
      corr_edge_rows<T, CHUNKED, true>(prm, e, lane, Cs, outs, res);
      if (lane < 63) {
#pragma unroll
        for (int kk = 0; kk < KOUT; kk++)
          *reinterpret_cast<h2v *>(row + 2 * (lane + 63 * kk)) = (h2v){(_Float16)res[0][kk], (_Float16)res[1][kk]};
      }
      if (lane < (CL1_K - 2 * NOUT) / 2) *reinterpret_cast<h2v *>(row + 2 * NOUT + 2 * lane) = (h2v){(_Float16)0.f, (_Float16)0.f};
    } else {
      for (int c = lane; c < CL1_K / 2; c += 64) *reinterpret_cast<h2v *>(row + 2 * c) = (h2v){(_Float16)0.f, (_Float16)0.f};
    }
  }
  __syncthreads();
  // Linear1: [16 x 896] x W1^T, wave w owns column tiles [NT w, NT w + NT); same operand order (activations as A) and
  // ascending K as mlp_gemm_chunk in csrc/update_mlp.hip
  f32x4_t acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) acc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const _Float16 *wb = p.w1 + ((size_t)(wave * NT) * 64 + lane) * 8;
  const _Float16 *xb = Rs + j * CL1_RXS + 8 * q;
#pragma unroll 2
  for (int ks = 0; ks < p.nks; ks++) {
    h8 bw[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) bw[nt] = *reinterpret_cast<const h8 *>(wb + ((size_t)ks * (CL1_N / 16) + nt) * 512);
    const h8 a = *reinterpret_cast<const h8 *>(xb + ks * 32);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bw[nt], acc[nt], 0, 0, 0);
  }
  __syncthreads();                                   // every wave is past its reads of the row tile: c1 takes its place
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int col = (wave * NT + nt) * 16 + j;
    const float bias = p.b1[col];
#pragma unroll
    for (int r = 0; r < 4; r++) Rs[(4 * q + r) * CL1_RXS + col] = (_Float16)fmaxf(acc[nt][r] + bias, 0.f);
  }
  __syncthreads();
  for (int i = tid; i < CL1_ROWS * (CL1_N / 8); i += 64 * NWV) {
    const int r = i / (CL1_N / 8), c8 = i - r * (CL1_N / 8);
    const int e = s_edge[r];
    if (e >= 0) *reinterpret_cast<h8 *>(p.c1 + (size_t)e * CL1_N + 8 * c8) = *reinterpret_cast<const h8 *>(Rs + r * CL1_RXS + 8 * c8);
  }
}
template <int NWV>
static constexpr size_t corr_l1_lds() {
  return (size_t)CL1_ROWS * CL1_RXS * 2 + (size_t)NWV * 9 * CORR_TM * 4 + (size_t)NWV * 441 * 4 + CL1_ROWS * 4;
}

// ---- launchers
template <int NWV>
static int corr_l1_launch(const CorrL1Params &p, int E_bound, bool chunked, hipStream_t st) {
  const size_t lds = corr_l1_lds<NWV>();
  auto k = chunked ? corr_l1_kernel<NWV, true> : corr_l1_kernel<NWV, false>;
  static bool attr[2] = {false, false};
  if (lds > 48 * 1024 && !attr[chunked]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return RAMP_ELAUNCH;
    attr[chunked] = true;
  }
  const int ngroups = (E_bound + CL1_ROWS - 1) / CL1_ROWS, chunkg = (ngroups + CORR_XCDS - 1) / CORR_XCDS;
  hipLaunchKernelGGL(k, dim3(chunkg * CORR_XCDS), dim3(64 * NWV), lds, st, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
extern "C" {

int ramp_i_corr_l1_fwd(const void *fmap1, const ramp_corr_level *levels, const float *coords, const int64_t *ii,
                       const int64_t *jj, const int32_t *order, const void *w1_packed, const float *b1, int corr_k,
                       void *c1, long mod_ii, long mod_jj, int E, int layout, const int32_t *dyn, void *stream,
                       const float *tf_poses, const float *tf_patches, const float *tf_intr, const int64_t *tf_src,
                       const int32_t *slot0) {
  if (E < 0 || !levels) return RAMP_EINVAL;
  if (corr_k != CL1_K || (layout != RAMP_NHWC && layout != RAMP_NHWC32)) return RAMP_EUNSUPPORTED;
  if (slot0 && mod_jj <= 0) return RAMP_EINVAL;
  if (tf_poses && (!tf_patches || !tf_intr || !tf_src)) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!fmap1 || !coords || !ii || !jj || !w1_packed || !b1 || !c1) return RAMP_EINVAL;
  CorrL1Params p;
  CorrParams &prm = p.c;
  prm.fmap1 = fmap1;
  for (int l = 0; l < CORR_MAXLEV; l++) {
    if (!levels[l].fmap || levels[l].H2 <= 0 || levels[l].W2 <= 0) return RAMP_EINVAL;
    prm.fmap2[l] = levels[l].fmap; prm.H2[l] = levels[l].H2; prm.W2[l] = levels[l].W2; prm.cdiv[l] = levels[l].coord_div;
  }
  prm.nlevels = 2; prm.coords = coords; prm.ii = ii; prm.jj = jj; prm.out = nullptr; prm.E = E; prm.N1 = 0; prm.N2 = 0;
  prm.order = order; prm.mod_ii = mod_ii; prm.mod_jj = mod_jj; prm.row_elems = CL1_K;
  prm.chunk = (E + CORR_XCDS - 1) / CORR_XCDS; prm.dyn = dyn;
  prm.tf_poses = tf_poses; prm.tf_patches = tf_patches; prm.tf_intr = tf_intr; prm.tf_src = tf_src; prm.slot0 = slot0;
  p.w1 = (const _Float16 *)w1_packed; p.b1 = b1; p.c1 = (_Float16 *)c1; p.nks = corr_k / 32;
  static int nwv = 0;                               // RAMP_CORR_L1_WAVES = 2 | 4 | 8 waves per workgroup (16 factors each way)
  if (!nwv) { const char *e = getenv("RAMP_CORR_L1_WAVES"); nwv = e ? atoi(e) : 4; if (nwv != 2 && nwv != 4 && nwv != 8) nwv = 4; }
  const bool chunked = layout == RAMP_NHWC32;
  if (nwv == 2) return corr_l1_launch<2>(p, E, chunked, (hipStream_t)stream);
  if (nwv == 8) return corr_l1_launch<8>(p, E, chunked, (hipStream_t)stream);
  return corr_l1_launch<4>(p, E, chunked, (hipStream_t)stream);
}

int ramp_corr_l1_fwd_ordered(const void *fmap1, const ramp_corr_level *levels, const float *coords, const int64_t *ii,
                             const int64_t *jj, const int32_t *order, const void *w1_packed, const float *b1, int corr_k,
                             void *c1, long mod_ii, long mod_jj, int E, int layout, void *stream) {
  return ramp_i_corr_l1_fwd(fmap1, levels, coords, ii, jj, order, w1_packed, b1, corr_k, c1, mod_ii, mod_jj, E, layout,
                            nullptr, stream, nullptr, nullptr, nullptr, nullptr, nullptr);
}

