// SHELVED (round 6): the speculative keyframe edit of round 5 (RAMP_SPEC_EDIT=1): both outcomes of keyframe()'s graph edit and the next
// graph's plan computed beside the update operator, the tail behind the motion test one select launch.  Bit-identical to the serial
// tail, 24 instead of 29 launches on the main queue, and no faster (profiles/r05_spec_edit_*).  Cut out of
// csrc/track.hip (+ ramp_i_plan_dyn_pair in csrc/graph.hip, the spec_* fields of ramp_track, DeviceTrack._init_speculative_edit);
// kept for the record, not compiled.
// ---- TrkEdit's candidate fields and trk_cand
  // speculative edit (trk_select_kernel): spec != 0 -- blockIdx.z is the OUTCOME the launch assumes (0 keep, 1 remove); the
  // sizes are read from dyn (the live block) and written to that candidate's block, the graph goes to the candidate's
  // buffer; no delta-log entry, no row shift
  int spec;
  int32_t *cdyn[2];
  int64_t *cgout[2];
  int32_t *cws[2];         // cnt / off / fmin of each candidate
};
// what a launch reads and writes: the live buffers and the motion test's decision, or candidate blockIdx.z's
struct TrkCand {
  const int32_t *dyn_in;   // sizes before the edit
  int32_t *dyn;            // sizes after
  int64_t *gout;
  int32_t *cnt, *off, *fmin;
  int force;               // -1: the motion test decides
};
__device__ __forceinline__ TrkCand trk_cand(const TrkEdit &p) {
  TrkCand c;
  c.dyn_in = p.dyn;
  if (!p.spec) {
    c.dyn = p.dyn; c.gout = p.gout; c.cnt = p.cnt; c.off = p.off; c.fmin = p.fmin; c.force = -1;
  } else {
    const int o = blockIdx.z;
    c.dyn = p.cdyn[o]; c.gout = p.cgout[o]; c.cnt = p.cws[o]; c.off = p.cws[o] + p.nb; c.fmin = p.cws[o] + 2 * p.nb; c.force = o;
  }
  return c;
}

// ---- trk_spec_log_kernel / trk_select_kernel
// ---------------------------------------------------------------------------------------- speculative keyframe edit
// The graph edit and the next graph's plan depend on the motion test's DECISION only, not on anything bundle adjustment
// computes: both possible next graphs (keyframe n - KEYFRAME_INDEX kept / dropped) are structural functions of the current
// one.  ramp_track_step therefore runs flag -> decide -> apply -> plan for BOTH outcomes on a second stream, beside the
// update operator, into candidate buffers (trk_*_kernel with TrkEdit.force = 0 / 1 on a candidate's copy of the sizes),
// and the serial tail behind the motion test shrinks from seven dependent launches (~54 us) to this one: take the
// decision, copy the chosen candidate's graph, plan and sizes into the live buffers and shift the frame buffers if the
// keyframe went (the delta-log entry, which needs the poses bundle adjustment just wrote, rides in the wait in front).  Same kernels, same inputs:
// the live buffers end up bit-identical to the seven-launch path's.
#define TRK_NCOPY 17
struct TrkSelect {
  TrkEdit e;                          // decision parameters, delta log, frame buffers of the row shift
  int32_t *dyn;                       // live sizes
  const int32_t *cand;                // [2][RAMP_DYN_WORDS]: [0] keep, [1] remove
  int32_t *mirror;                    // optional: the host's lazy copy of the sizes (mapped pinned memory)
  const char *src[2][TRK_NCOPY];
  char *dst[TRK_NCOPY];
  long bytes[TRK_NCOPY];              // multiples of 4
};
// the launch in front of trk_select_kernel: waits for "both candidates are ready" (flag; nullptr: an event ordered the
// streams) and appends the delta-log entry of a dropped keyframe -- it reads rows k - 1, k of the poses and time stamps,
// which the select launch's row shift overwrites (Ramp_vo.py:249-253: delta[t1] = (t0, poses[k] * poses[k-1]^-1))
__global__ void trk_spec_log_kernel(const TrkEdit p, const uint32_t *flag, uint32_t value, long ticks) {
  if (flag) {
    const long t0 = wall_clock64();
    bool seen;
    while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= value) && (long)wall_clock64() - t0 < ticks)
      __builtin_amdgcn_s_sleep(64);
    if (!seen && threadIdx.x == 0) atomicOr(p.dyn + RAMP_DYN_STATUS, 128);
  }
  if (threadIdx.x != 0 || !trk_remove(p.mm, p.thresh)) return;
  const int k = p.dyn[RAMP_DYN_N] - p.keyframe_index, idx = p.dyn[RAMP_DYN_NLOG];
  if (idx < p.log_cap) {
    float Pk[7], Pm[7], Pi[7], dP[7];
    for (int q = 0; q < 7; q++) { Pk[q] = p.poses[7 * k + q]; Pm[q] = p.poses[7 * (k - 1) + q]; }
    lt_inv(Pm, Pi);
    lt_mul(Pk, Pi, dP);
    float *lo = p.dlog + (size_t)idx * RAMP_TRACK_LOG;
    lo[0] = __int_as_float((int)p.tstamps[k]);
    lo[1] = __int_as_float((int)p.tstamps[k - 1]);
    for (int q = 0; q < 7; q++) lo[2 + q] = dP[q];
    p.dyn[RAMP_DYN_NLOG] = idx + 1;
  } else {
    atomicOr(p.dyn + RAMP_DYN_STATUS, 16);
  }
}

__global__ void __launch_bounds__(256) trk_select_kernel(const TrkSelect s) {
  const int tid = threadIdx.x;
  const int o = trk_remove(s.e.mm, s.e.thresh) ? 1 : 0;
  const int32_t *cd = s.cand + o * RAMP_DYN_WORDS;
  if (blockIdx.y > 0) {
    if (o) trk_shift_buffer(s.e, blockIdx.y - 1, cd[RAMP_DYN_K], cd[RAMP_DYN_NPREV], tid);
    return;
  }
  for (int b = 0; b < TRK_NCOPY; b++) {
    const long n = s.bytes[b];
    const char *src = s.src[o][b];
    char *dst = s.dst[b];
    if (!(n & 15)) {
      for (long i = (long)blockIdx.x * 256 + tid; i < n / 16; i += (long)gridDim.x * 256)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    } else {
      for (long i = (long)blockIdx.x * 256 + tid; i < n / 4; i += (long)gridDim.x * 256)
        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
    }
  }
  if (blockIdx.x != 0 || tid != 0) return;
  // the sizes trk_decide_kernel rewrote in the chosen candidate's copy; the live block keeps what the step itself set
  // meanwhile (status bits of bundle adjustment / the gate wait, MEDOK, the log length)
  const int status = cd[RAMP_DYN_STATUS];
  const int own[13] = {RAMP_DYN_NPREV, RAMP_DYN_EPREV, RAMP_DYN_EKEPT, RAMP_DYN_REMOVED, RAMP_DYN_K, RAMP_DYN_NROW, RAMP_DYN_N,
                       RAMP_DYN_E, RAMP_DYN_KLO, RAMP_DYN_FLO, RAMP_DYN_W, RAMP_DYN_FRAME, RAMP_DYN_FRAME2};
  for (int q = 0; q < 13; q++) s.dyn[own[q]] = cd[own[q]];
  if (status) atomicOr(s.dyn + RAMP_DYN_STATUS, status);
  if (s.mirror) {
    __threadfence_system();
    for (int w = 0; w < RAMP_DYN_WORDS; w++) s.mirror[w] = s.dyn[w == RAMP_DYN_FRAME2 ? RAMP_DYN_FRAME : w];
  }
}


// ---- ramp_track_step: the switch
  // speculative keyframe edit (see trk_select_kernel): both candidates of the next graph + plan on the second stream,
  // from the live graph and sizes the previous step's select launch (or the hand-over) left final
  static int spec_on = -1;                               // RAMP_SPEC_EDIT=1 (default 0: measured a wash)
  static int spec_nap = 2;                               // x s_sleep 64 (~2 us) between two looks at the signal word
  if (spec_on < 0) {
    const char *e = getenv("RAMP_SPEC_EDIT"); spec_on = e ? atoi(e) : 0;
    if ((e = getenv("RAMP_SPEC_NAP")) && atoi(e) > 0) spec_nap = atoi(e);
  }
  const int full = RAMP_TRACK_COMMIT | RAMP_TRACK_UPDATE | RAMP_TRACK_KEYFRAME;
  const bool spec = spec_on && t->spec_stream && (flags & full) == full && !(flags & RAMP_TRACK_MM_GIVEN) && !t->feat_fp32 &&
                    t->mm && t->dlog && t->spec_graph[0] && t->spec_graph[1] && t->spec_dyn && t->spec_plan_ws && t->spec_edit_ws &&
                    ((t->spec_go && t->spec_done) || (t->spec_ev_go && t->spec_ev_done));

// ---- ramp_track_step: both candidates on the second stream
  if (spec) {
    hipStream_t ax = (hipStream_t)t->spec_stream;
    // behind this step's commit launch (which stores spec_go): the live graph and sizes are the previous step's final ones
    // (RAMP_SPEC_AT=gate: the chain starts at this step's gate -- the first SoftAgg launch -- instead of its commit, i.e. on
    // the front-end stream right ahead of the next frame's front end; measured)
    static int spec_at_gate = -1;
    if (spec_at_gate < 0) { const char *e = getenv("RAMP_SPEC_AT"); spec_at_gate = e && !strcmp(e, "gate"); }
    if (t->spec_go && spec_at_gate && t->gate_flag)
      hipLaunchKernelGGL(trk_wait_flag_kernel, dim3(1), dim3(64), 0, ax, t->gate_flag, t->gate_seq, 2000000000L, 0L, spec_nap,
                         t->dyn + RAMP_DYN_STATUS);
    else if (t->spec_go)       // (a data dependency: the time-out is a hang guard, not a scheduling choice -- 20 s)
      hipLaunchKernelGGL(trk_wait_flag_kernel, dim3(1), dim3(64), 0, ax, t->spec_go, t->spec_seq, 2000000000L, 0L, spec_nap,
                         t->dyn + RAMP_DYN_STATUS);
    else if (hipEventRecord((hipEvent_t)t->spec_ev_go, st) != hipSuccess ||
             hipStreamWaitEvent(ax, (hipEvent_t)t->spec_ev_go, 0) != hipSuccess) return RAMP_ELAUNCH;
    // both outcomes in the same launches (blockIdx.z): the sizes are read from the live block, each candidate writes its own
    TrkEdit p;
    TRK_DO(trk_edit_fill(t, cur, counter, p));
    p.spec = 1;
    const size_t pws = t->plan_ws_bytes;
    int32_t *cdyn[2];
    const int64_t *cg[2];
    void *cws[2];
    for (int o = 0; o < 2; o++) {
      p.cdyn[o] = cdyn[o] = t->spec_dyn + o * RAMP_DYN_WORDS;
      p.cgout[o] = t->spec_graph[o]; cg[o] = t->spec_graph[o];
      p.cws[o] = t->spec_edit_ws + o * (3 * p.nb + 8);
      cws[o] = (char *)t->spec_plan_ws + o * pws;
    }
    hipLaunchKernelGGL(trk_flag_kernel, dim3(p.nb, 1, 2), dim3(256), 0, ax, p);
    hipLaunchKernelGGL(trk_decide_kernel, dim3(1, 1, 2), dim3(256), 0, ax, p);
    hipLaunchKernelGGL(trk_apply_kernel, dim3(p.nb + ramp_cdiv(new_cap + p.pad, 256), 1, 2), dim3(256), 0, ax, p);
    TRK_DO(ramp_i_plan_dyn_pair(cg, Ec, Ep_next, cdyn, t->M, t->kkey_cap, t->pkey_cap, t->kk_cap, t->ij_cap, t->spec_plan, cws,
                                pws, ax));
    if (t->spec_done) hipLaunchKernelGGL(trk_signal_kernel, dim3(1), dim3(1), 0, ax, t->spec_done, t->spec_seq);
    else if (hipEventRecord((hipEvent_t)t->spec_ev_done, ax) != hipSuccess) return RAMP_ELAUNCH;
    RAMP_CHECK_LAUNCH();
  }

// ---- ramp_track_step: the one-launch tail
    if (spec) {
      // both candidates are (long) ready: take the decision and copy the chosen one into the live buffers
      if (!t->spec_done && hipStreamWaitEvent(st, (hipEvent_t)t->spec_ev_done, 0) != hipSuccess) return RAMP_ELAUNCH;
      hipLaunchKernelGGL(trk_spec_log_kernel, dim3(1), dim3(64), 0, st, p, t->spec_done, t->spec_seq, 2000000000L);
      TrkSelect sel;
      sel.e = p; sel.dyn = t->dyn; sel.cand = t->spec_dyn; sel.mirror = mirror;
      int64_t *gl = t->graph[1 - cur];
      int nbuf = 0;
      auto add = [&](const void *a0, const void *a1, void *d, long bytes) {
        sel.src[0][nbuf] = (const char *)a0; sel.src[1][nbuf] = (const char *)a1; sel.dst[nbuf] = (char *)d;
        sel.bytes[nbuf] = (bytes + 3) / 4 * 4; nbuf++;
      };
      const long ep16 = ((long)Ep_next + 3) / 4 * 4 < Ec ? ((long)Ep_next + 3) / 4 * 4 : Ec;   // whole 16-byte pieces of int32 rows
      for (int r = 0; r < 4; r++)
        add(t->spec_graph[0] + r * (size_t)Ec, t->spec_graph[1] + r * (size_t)Ec, gl + r * (size_t)Ec, ep16 * 8);
      const ramp_plan_set &q0 = t->spec_plan[0], &q1 = t->spec_plan[1];
      add(q0.kk_order, q1.kk_order, t->kk_order, ep16 * 4); add(q0.kk_gid, q1.kk_gid, t->kk_gid, ep16 * 4);
      add(q0.ij_order, q1.ij_order, t->ij_order, ep16 * 4); add(q0.ij_gid, q1.ij_gid, t->ij_gid, ep16 * 4);
      add(q0.ix, q1.ix, t->ix, ep16 * 8); add(q0.jx, q1.jx, t->jx, ep16 * 8); add(q0.kj, q1.kj, t->kj, ep16 * 4);
      add(q0.kk_seg, q1.kk_seg, t->kk_seg, (long)(t->kk_cap + 2) * 4); add(q0.ij_seg, q1.ij_seg, t->ij_seg, (long)(t->ij_cap + 2) * 4);
      add(q0.kk_ukeys, q1.kk_ukeys, t->kk_ukeys, (long)(t->kk_cap + 2) * 8); add(q0.ij_ukeys, q1.ij_ukeys, t->ij_ukeys, (long)(t->ij_cap + 2) * 8);
      add(q0.kk_ngroups, q1.kk_ngroups, t->kk_ngroups, 4); add(q0.ij_ngroups, q1.ij_ngroups, t->ij_ngroups, 4);
      static_assert(TRK_NCOPY == 17, "the copy list above");
      hipLaunchKernelGGL(trk_select_kernel, dim3(t->fmap1_slot ? 256 : 1024, 1 + p.nbuf), dim3(256), 0, st, sel);
      RAMP_CHECK_LAUNCH();

// ---- csrc/graph.hip: both candidates' plans in one set of launches
// two plans (the candidates of the speculative keyframe edit) in one set of launches: graphs g4[z], sizes dyn[z] (status word
// inside: + RAMP_DYN_STATUS), outputs set[z], workspaces ws[z] of ramp_i_plan_dyn_ws bytes each
int ramp_i_plan_dyn_pair(const int64_t *const g4[2], int E_cap, int E_grid, int32_t *const dyn[2], int M, int kkey_cap,
                         int pkey_cap, int kk_cap, int ij_cap, const ramp_plan_set set[2], void *const ws[2], size_t ws_bytes,
                         hipStream_t st) {
  PlanPair pp;
  for (int z = 0; z < 2; z++) {
    const ramp_plan_set &q = set[z];
    const int rc = plan_fill(pp, z, g4[z], E_cap, E_grid, dyn[z], dyn[z] ? dyn[z] + RAMP_DYN_STATUS : nullptr, M, kkey_cap,
                             pkey_cap, kk_cap, ij_cap, q.kk_order, q.kk_gid, q.kk_seg, q.kk_ngroups, q.kk_ukeys, q.ij_order,
                             q.ij_gid, q.ij_seg, q.ij_ngroups, q.ij_ukeys, q.ix, q.jx, q.kj, ws[z], ws_bytes);
    if (rc != RAMP_OK) return rc;
  }
  return plan_launch(pp, 2, E_cap, E_grid, kkey_cap, pkey_cap, kk_cap, ij_cap, nullptr, st);
}

