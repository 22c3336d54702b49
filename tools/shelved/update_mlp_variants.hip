// SHELVED (round 6): variants of the fp16 update-operator chains that were built, verified bit-identical / within bounds, measured
// slower or equal, and switched off (DESIGN.md sections 8.000, 8.00): the wide-tile correlation MLP (RAMP_CORR_MLP_BIG), c1 + c2 in
// one launch (RAMP_NBR2), the tail-only correlation MLP entry point that served the fused correlation + Linear1 launch.
// Cut out of csrc/update_mlp.hip (they use that file's mlp_gemm / tile_ln helpers); kept for the record, not compiled.
// ---- upd_corr_mlp_big_kernel (RAMP_CORR_MLP_BIG=1)
// ------------------------------------------------------------------ correlation MLP (big tile)
// The whole correlation MLP (CorrTailParams, FULL) on the wide-tile recipe of c1 / c2: 16 NMT rows per workgroup (80: 500
// workgroups for 40k factors, two per CU -- one round), transposed accumulators, weight fragments one K step ahead, and
// NO fp32 parking passes: a lane holds four consecutive columns of its rows, the two LayerNorms reduce through an LDS
// table of per-wave partial sums (tile_ln), the previous state / context rows are added and the result stored as
// 16-byte pieces straight from the accumulators.  Same values as upd_corr_tail_kernel up to the order of the LayerNorm
// sums (tile_ln vs row_ln).
template <int NMT, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) upd_corr_mlp_big_kernel(const CorrTailParams p) {
  static_assert(NW == MWAVES, "tile_ln's table is [rows][MWAVES]");
  constexpr int NTW = 24 / NW, ROWS = 16 * NMT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  float *T1 = reinterpret_cast<float *>(Xs + ROWS * MXS), *T2 = T1 + ROWS * NW;
  __shared__ long s_ra[ROWS], s_rb[ROWS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int row0 = blockIdx.x * ROWS;
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (row0 >= pE) return;                      // (workgroup-uniform: only with device-side sizes)
  const int col0 = wave * (16 * NTW);
  if (tid < ROWS) {                            // the rows the last pass adds: indices fetched now
    const int row = row0 + tid;
    long ra = -1, rb = 0;
    if (row < pE) {
      ra = p.net ? (p.net_map ? p.net_map[row] : (long)row) : -1;
      rb = p.inp_idx ? p.inp_idx[row] : (long)row;
      if (p.inp_mod > 0) rb %= p.inp_mod;
    }
    s_ra[tid] = ra; s_rb[tid] = rb;
  }
  f4 acc[NMT][NTW];
  big_zero<NMT, NTW>(acc);
  // Linear1 over K = corr_k in chunks of <= 12 K steps (the tile is 384 wide)
  const int nks_total = p.corr_k / 32;
  for (int ks0 = 0; ks0 < nks_total; ks0 += MKS) {
    const int nks = min(MKS, nks_total - ks0);
    const int v8 = nks * 4;                                  // 16-byte vectors per row of this chunk
    for (int i = tid; i < ROWS * v8; i += 64 * NW) {
      const int r = i / v8, c8 = i - r * v8;
      h8 v = (h8){0, 0, 0, 0, 0, 0, 0, 0};
      if (row0 + r < pE) v = ld_dead(reinterpret_cast<const h8 *>(p.corr + (size_t)(row0 + r) * p.corr_k + ks0 * 32 + 8 * c8));
      *reinterpret_cast<h8 *>(Xs + r * MXS + 8 * c8) = v;
    }
    __syncthreads();
    big_gemm<NMT, NTW>(Xs, p.w1, nks, ks0, wave, lane, acc);
    __syncthreads();                                         // before the tile is overwritten
  }
  big_store_tile<NMT, NTW, true>(Xs, acc, p.b1, col0, q, j);
  __syncthreads();
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.w2, MKS, 0, wave, lane, acc);
  {
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const float4 b = *reinterpret_cast<const float4 *>(p.b2 + col0 + nt * 16 + 4 * q);
#pragma unroll
      for (int mt = 0; mt < NMT; mt++) {
        acc[mt][nt][0] = h_round(acc[mt][nt][0] + b.x); acc[mt][nt][1] = h_round(acc[mt][nt][1] + b.y);
        acc[mt][nt][2] = h_round(acc[mt][nt][2] + b.z); acc[mt][nt][3] = h_round(acc[mt][nt][3] + b.w);
      }
    }
  }
  tile_ln_lean<NMT>(acc, p.ln_w, p.ln_b, p.ln_eps, T1, T2, wave, q, j, col0);    // (its barriers: every wave is past its reads of the tile)
#pragma unroll
  for (int nt = 0; nt < NTW; nt++)
#pragma unroll
    for (int mt = 0; mt < NMT; mt++)
      *reinterpret_cast<hh4 *>(Xs + (mt * 16 + j) * MXS + col0 + nt * 16 + 4 * q) =
          (hh4){(_Float16)fmaxf(acc[mt][nt][0], 0.f), (_Float16)fmaxf(acc[mt][nt][1], 0.f), (_Float16)fmaxf(acc[mt][nt][2], 0.f),
                (_Float16)fmaxf(acc[mt][nt][3], 0.f)};
  __syncthreads();
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.w3, MKS, 0, wave, lane, acc);
  // net_prev + inp + c (in that order), LayerNorm, fp32 store
  {
    float4 bv[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) bv[nt] = *reinterpret_cast<const float4 *>(p.b3 + col0 + nt * 16 + 4 * q);
    const char *net_base = reinterpret_cast<const char *>(p.net ? p.net : p.net_out);      // (s_ra is -1 everywhere without a state)
#pragma unroll
    for (int mt = 0; mt < NMT; mt++) {
      __builtin_amdgcn_sched_barrier(0);                       // (the loads stay behind the last product: registers)
      // (32-bit byte offsets from the uniform bases, the n-tile step as an immediate: one address register per row tile and
      // table -- with 64-bit addresses per piece the pass peaked at 186 registers)
      const long ra = s_ra[mt * 16 + j];
      const unsigned oa = (unsigned)((ra >= 0 ? ra : 0) * MD + col0 + 4 * q) * 4u;
      const unsigned ob = (unsigned)(s_rb[mt * 16 + j] * MD + col0 + 4 * q) * 2u;
      float4 a[NTW];
      hh4 t[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) {
        const f4 a_ = ld_dead(reinterpret_cast<const f4 *>(net_base + oa + nt * 64));      // (branch-free: row 0 stands in, masked)
        a[nt] = ra >= 0 ? make_float4(a_[0], a_[1], a_[2], a_[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        t[nt] = *reinterpret_cast<const hh4 *>(reinterpret_cast<const char *>(p.inp) + ob + nt * 32);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) {
        acc[mt][nt][0] = (a[nt].x + (float)t[nt][0]) + h_round(acc[mt][nt][0] + bv[nt].x);
        acc[mt][nt][1] = (a[nt].y + (float)t[nt][1]) + h_round(acc[mt][nt][1] + bv[nt].y);
        acc[mt][nt][2] = (a[nt].z + (float)t[nt][2]) + h_round(acc[mt][nt][2] + bv[nt].z);
        acc[mt][nt][3] = (a[nt].w + (float)t[nt][3]) + h_round(acc[mt][nt][3] + bv[nt].w);
      }
      __builtin_amdgcn_sched_barrier(0);                       // (one row tile's 36 bytes per lane in flight at a time: registers)
    }
  }
  __syncthreads();                                           // (tile_ln's tables are free again)
  tile_ln_lean<NMT>(acc, p.norm_w, p.norm_b, p.norm_eps, T1, T2, wave, q, j, col0);
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    const int row = row0 + mt * 16 + j;
    if (row >= pE) continue;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
      st_st(reinterpret_cast<f4 *>(p.net_out + (size_t)row * MD + col0 + nt * 16 + 4 * q), acc[mt][nt]);
  }
}


// ---- upd_nbr2_kernel (RAMP_NBR2=1)
// ------------------------------------------------------------------ c1 AND c2 in one launch
// net += c1(mask * net[ix]);  net += c2(mask * net[jx])   (ramp/net.py:77-82)
// The two launches above each read the state twice (own row + neighbour row) and write it once: 6 passes over
// [E][384] fp32.  Here a workgroup owns 78 CONSECUTIVE POSITIONS of the (kk, jj)-sorted factor list `kj` (the graph
// plan emits it): the temporal neighbours of position P are positions P - 1 and P + 1 (if in the same patch group), so
//   c1:  G[t]  = c1(net[kj[p0 - 1 + t]]),  t = 0..78         -> out1 at position p0 + t = net + (has_prev ? G[t] : c1(0))
//   c2:  G2[v] = c2(out1 at position p0 + 1 + v), v = 0..77  -> out  at position p0 + v = out1 + (has_next ? G2[v] : c2(0))
// with ONE read of every state row (plus the one-row halo at either end) and one write.  Tile row 79 is a zero row in
// both chains: it yields c1(0) / c2(0), what the reference's masked gather feeds a factor without a neighbour.  The
// row shift between the chains happens where out1 goes back to LDS as the c2 input (row u is stored as row u - 1), so
// every lane adds values of rows it owns.
struct Nbr2Params {
  const float *net_in;         // [E][384] fp32
  float *net_out;              // [E][384] fp32, a different buffer
  const int32_t *kj;           // [E] position in (kk, jj) order -> factor
  const int64_t *ix, *jx;      // [E] temporal neighbours (only their sign is read here)
  const _Float16 *w1a, *w1b, *w2a, *w2b;
  const float *b1a, *b1b, *b2a, *b2b;
  int E;
  const int32_t *dyn;          // optional device-side sizes (RAMP_DYN_*): E is then the launch bound
};
#define NBR2_OUT 78
__global__ void __launch_bounds__(512, 4) upd_nbr2_kernel(const Nbr2Params p) {
  constexpr int NMT = 5, NW = 8, NTW = 3, ROWS = 80;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *Xs = reinterpret_cast<_Float16 *>(smem_raw);
  float *Ks = reinterpret_cast<float *>(Xs + ROWS * MXS);          // [384]: the zero row's result
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int p0 = blockIdx.x * NBR2_OUT;
  const int pE = p.dyn ? p.dyn[RAMP_DYN_E] : p.E;
  if (p0 >= pE) return;                        // (workgroup-uniform)
  const int col0 = wave * (16 * NTW);
  // ---- c1 input: tile row t <- state row of position p0 - 1 + t (rows past either end of the list: zeros)
  constexpr int RPW = ROWS / NW;
  {
    int src[RPW];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const int t = wave + i * NW, P = p0 - 1 + t;
      src[i] = (t < ROWS - 1 && P >= 0 && P < pE) ? p.kj[P] : -1;
    }
    float2 v[RPW][3];
#pragma unroll
    for (int i = 0; i < RPW; i++) {
      const float *b = p.net_in + (size_t)(src[i] >= 0 ? src[i] : 0) * MD + 2 * lane;
#pragma unroll
      for (int k = 0; k < 3; k++) v[i][k] = ld_st2(b + 128 * k);
    }
#pragma unroll
    for (int i = 0; i < RPW; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float2 a = src[i] >= 0 ? v[i][k] : make_float2(0.f, 0.f);
        *reinterpret_cast<h2 *>(Xs + (wave + i * NW) * MXS + 2 * lane + 128 * k) = (h2){(_Float16)a.x, (_Float16)a.y};
      }
  }
  // ---- the rows this lane owns in accumulator layout: row u = 16 mt + j <-> position p0 + u
  unsigned ro[NMT];            // byte offset of the factor's state row + this lane's first column
  unsigned has_prev = 0, has_next = 0, live1 = 0, live2 = 0;
#pragma unroll
  for (int mt = 0; mt < NMT; mt++) {
    const int u = mt * 16 + j, P = p0 + u;
    const bool in = u < ROWS - 1 && P < pE;
    const int e = p.kj[in ? P : pE - 1];
    ro[mt] = (unsigned)(e * MD + col0 + 4 * q) * 4u;
    if (in) {
      live1 |= 1u << mt;
      if (u < NBR2_OUT) live2 |= 1u << mt;
      if (p.ix[e] >= 0) has_prev |= 1u << mt;
      if (p.jx[e] >= 0) has_next |= 1u << mt;
    }
  }
  __syncthreads();
  f4 acc[NMT][NTW];
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.w1a, MKS, 0, wave, lane, acc);
  __syncthreads();                                       // every wave is past its reads of x
  big_store_tile<NMT, NTW, true>(Xs, acc, p.b1a, col0, q, j);
  __syncthreads();
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.w1b, MKS, 0, wave, lane, acc);
  // G = round_fp16(acc + bias); the zero row's G (tile row 79: m-tile 4, j = 15) -> Ks
  float4 bv[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) bv[nt] = *reinterpret_cast<const float4 *>(p.b1b + col0 + nt * 16 + 4 * q);
  if (j == 15) {
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
      *reinterpret_cast<float4 *>(Ks + col0 + nt * 16 + 4 * q) =
          make_float4(h_round(acc[NMT - 1][nt][0] + bv[nt].x), h_round(acc[NMT - 1][nt][1] + bv[nt].y),
                      h_round(acc[NMT - 1][nt][2] + bv[nt].z), h_round(acc[NMT - 1][nt][3] + bv[nt].w));
  }
  __syncthreads();                                       // Ks written; every wave is past its reads of the hidden tile
  // out1 = net + (has_prev ? G : c1(0)): its fp16 copy is the c2 input; the fp32 value is parked in the OUTPUT row
  // (this lane re-reads its own 16-byte pieces after the second chain: program order, no fence) -- holding it in 60
  // registers instead left one workgroup per CU (176 VGPRs) and nothing to overlap a workgroup's gather with
  {
    float4 kz[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) kz[nt] = *reinterpret_cast<const float4 *>(Ks + col0 + nt * 16 + 4 * q);
#pragma unroll
    for (int mt = 0; mt < NMT; mt++) {
      const bool hp = (has_prev >> mt) & 1, lv = (live1 >> mt) & 1;
      const int u = mt * 16 + j;
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) {
        const float4 x = ldg4(p.net_in, ro[mt], nt);
        float4 g = make_float4(h_round(acc[mt][nt][0] + bv[nt].x), h_round(acc[mt][nt][1] + bv[nt].y),
                               h_round(acc[mt][nt][2] + bv[nt].z), h_round(acc[mt][nt][3] + bv[nt].w));
        if (!hp) g = kz[nt];
        const float4 o = lv ? make_float4(x.x + g.x, x.y + g.y, x.z + g.z, x.w + g.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        if ((live2 >> mt) & 1) stg4(p.net_out, ro[mt], nt, o);
        // c2 input: out1 of row u goes to tile row u - 1 (row 0 feeds the previous workgroup's last factor, not ours)
        if (u >= 1)
          *reinterpret_cast<hh4 *>(Xs + (u - 1) * MXS + col0 + nt * 16 + 4 * q) =
              (hh4){(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
      }
    }
    if (j == 15) {                                         // tile row 79: the zero row of the second chain
#pragma unroll
      for (int nt = 0; nt < NTW; nt++)
        *reinterpret_cast<hh4 *>(Xs + (ROWS - 1) * MXS + col0 + nt * 16 + 4 * q) = (hh4){0, 0, 0, 0};
    }
  }
  __syncthreads();
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.w2a, MKS, 0, wave, lane, acc);
  __syncthreads();
  big_store_tile<NMT, NTW, true>(Xs, acc, p.b2a, col0, q, j);
  __syncthreads();
  big_zero<NMT, NTW>(acc);
  big_gemm<NMT, NTW>(Xs, p.w2b, MKS, 0, wave, lane, acc);
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) bv[nt] = *reinterpret_cast<const float4 *>(p.b2b + col0 + nt * 16 + 4 * q);
  if (j == 15) {
#pragma unroll
    for (int nt = 0; nt < NTW; nt++)
      *reinterpret_cast<float4 *>(Ks + col0 + nt * 16 + 4 * q) =
          make_float4(h_round(acc[NMT - 1][nt][0] + bv[nt].x), h_round(acc[NMT - 1][nt][1] + bv[nt].y),
                      h_round(acc[NMT - 1][nt][2] + bv[nt].z), h_round(acc[NMT - 1][nt][3] + bv[nt].w));
  }
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < NTW; nt++) {
    const float4 kz = *reinterpret_cast<const float4 *>(Ks + col0 + nt * 16 + 4 * q);
#pragma unroll
    for (int mt = 0; mt < NMT; mt++) {
      if (!((live2 >> mt) & 1)) continue;
      float4 g = make_float4(h_round(acc[mt][nt][0] + bv[nt].x), h_round(acc[mt][nt][1] + bv[nt].y),
                             h_round(acc[mt][nt][2] + bv[nt].z), h_round(acc[mt][nt][3] + bv[nt].w));
      if (!((has_next >> mt) & 1)) g = kz;
      const float4 o1 = ldg4(p.net_out, ro[mt], nt);
      stg4(p.net_out, ro[mt], nt, make_float4(o1.x + g.x, o1.y + g.y, o1.z + g.z, o1.w + g.w));
    }
  }
}


// ---- ramp_upd_nbr2 launchers
int ramp_i_upd_nbr2(const float *net_in, const int32_t *kj, const int64_t *ix, const int64_t *jx, const void *w1a,
                    const float *b1a, const void *w1b, const float *b1b, const void *w2a, const float *b2a,
                    const void *w2b, const float *b2b, float *net_out, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!net_in || !kj || !ix || !jx || !w1a || !b1a || !w1b || !b1b || !w2a || !b2a || !w2b || !b2b || !net_out ||
      net_in == net_out)
    return RAMP_EINVAL;
  if ((long)E * MD * 4 >= (1l << 32)) return RAMP_EUNSUPPORTED;       // 32-bit row offsets
  Nbr2Params p;
  p.net_in = net_in; p.net_out = net_out; p.kj = kj; p.ix = ix; p.jx = jx;
  p.w1a = (const _Float16 *)w1a; p.w1b = (const _Float16 *)w1b; p.w2a = (const _Float16 *)w2a; p.w2b = (const _Float16 *)w2b;
  p.b1a = b1a; p.b1b = b1b; p.b2a = b2a; p.b2b = b2b; p.E = E; p.dyn = dyn;
  const size_t lds = (size_t)80 * MXS * 2 + MD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)upd_nbr2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return RAMP_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(upd_nbr2_kernel, dim3(ramp_cdiv(E, NBR2_OUT)), dim3(512), lds, (hipStream_t)stream, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}
int ramp_upd_nbr2(const float *net_in, const int32_t *kj, const int64_t *ix, const int64_t *jx, const void *w1a,
                  const float *b1a, const void *w1b, const float *b1b, const void *w2a, const float *b2a,
                  const void *w2b, const float *b2b, float *net_out, int E, void *stream) {
  return ramp_i_upd_nbr2(net_in, kj, ix, jx, w1a, b1a, w1b, b1b, w2a, b2a, w2b, b2b, net_out, E, nullptr, stream);
}


// ---- ramp_i_upd_corr_tail
int ramp_i_upd_corr_tail(const void *c1, const void *w2, const float *b2, const void *w3, const float *b3,
                         const float *ln_w, const float *ln_b, float ln_eps, const float *net, const int64_t *net_map,
                         const void *inp, const int64_t *inp_idx, long inp_mod, const float *norm_w,
                         const float *norm_b, float norm_eps, float *net_out, int E, const int32_t *dyn, void *stream) {
  if (E < 0) return RAMP_EINVAL;
  if (E == 0) return RAMP_OK;
  if (!c1 || !w2 || !b2 || !w3 || !b3 || !ln_w || !ln_b || !inp || !norm_w || !norm_b || !net_out || net == net_out)
    return RAMP_EINVAL;
  CorrTailParams p;
  p.c1 = (const _Float16 *)c1; p.w2 = (const _Float16 *)w2; p.w3 = (const _Float16 *)w3; p.b2 = b2; p.b3 = b3;
  p.ln_w = ln_w; p.ln_b = ln_b; p.ln_eps = ln_eps; p.net = net; p.net_map = net_map; p.inp = (const _Float16 *)inp;
  p.inp_idx = inp_idx; p.inp_mod = inp_mod; p.norm_w = norm_w; p.norm_b = norm_b; p.norm_eps = norm_eps;
  p.net_out = net_out; p.E = E; p.dyn = dyn;
  const size_t lds = (size_t)MBM * MXS * 2;
  p.corr = nullptr; p.w1 = nullptr; p.b1 = nullptr; p.corr_k = 0;
  hipLaunchKernelGGL(upd_corr_tail_kernel<false>, dim3(ramp_cdiv(E, MBM)), dim3(64 * MWAVES), lds, (hipStream_t)stream, p);
  RAMP_CHECK_LAUNCH();
  return RAMP_OK;
}


// ---- ramp_upd_corr_tail
int ramp_upd_corr_tail(const void *c1, const void *w2, const float *b2, const void *w3, const float *b3,
                       const float *ln_w, const float *ln_b, float ln_eps, const float *net, const int64_t *net_map,
                       const void *inp, const int64_t *inp_idx, long inp_mod, const float *norm_w,
                       const float *norm_b, float norm_eps, float *net_out, int E, void *stream) {
  return ramp_i_upd_corr_tail(c1, w2, b2, w3, b3, ln_w, ln_b, ln_eps, net, net_map, inp, inp_idx, inp_mod, norm_w, norm_b,
                              norm_eps, net_out, E, nullptr, stream);
}


// ---- the RAMP_CORR_MLP_BIG dispatch in ramp_i_upd_corr_mlp
  // RAMP_CORR_MLP_BIG=1: the wide-tile kernel without parking passes.  Measured (tools/mb_update.py):
  // 86.2 vs 87.8 us at 40000 factors (500 tiles, one round of two per CU), 78.2 vs 86.1 at 38400, 112.8 vs 87.5 at 41200
  // (515 tiles: a second round) -- a CU moves ~1.9 rows per us whichever kernel it runs, the launch is bound by its 220 MB
  // and the weight stream, not by the row passes; and its LayerNorm sums in another order (tile_ln), so a host-driven and
  // a device-resident step that choose by their row bound would differ in the last bit.  Off by default.
  static const bool big = getenv("RAMP_CORR_MLP_BIG") && atoi(getenv("RAMP_CORR_MLP_BIG")) != 0;
  if (big)
    if (const int nmt = big_pick_nmt(E, 5)) { BIG_DISPATCH(upd_corr_mlp_big_kernel, 8, p, E, nmt, true, (hipStream_t)stream) }
