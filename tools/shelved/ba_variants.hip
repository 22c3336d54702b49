// SHELVED (round 6): bundle-adjustment variants that were built, verified and measured slower: the
// per-column Cholesky kernels (round 1-2; the blocked kernel replaced them), the assembly inside the factorisation's workgroup
// (round 5, profiles/r05_ba_merge.txt).  Cut out of csrc/ba.hip; kept for the record, not compiled.
// ---- the per-column Cholesky kernels (RAMP_BA_CHOL=2): ba_chol_kernel, ba_chol64_kernel
// ------------------------------------------------------------------ K6
// Single workgroup, matrix in LDS, right-looking Cholesky with ONE barrier per column:
//   * the pivot d = A[j][j] is read by every thread (no broadcast step); the trailing update uses the
//     UNSCALED column, A[i][k] -= A[i][j] A[k][j] / d, so it does not wait for a scaling pass;
//   * the scaled column L[i][j] = A[i][j] / sqrt(d) goes to the unused upper triangle (A[j][i]) in the same
//     phase -- which is also the layout the back substitution wants (row j of L');
//   * the right-hand side rides along as row n6, so z = L^-1 y falls out of the factorisation
//     (column n6 of the upper triangle) and there is no forward substitution;
//   * 1/L[j][j] is kept per column: the back substitution multiplies, one barrier per step.
__global__ void __launch_bounds__(1024)
    ba_chol_kernel(const float *__restrict__ S, const float *__restrict__ yv,
                   float *__restrict__ dX, int32_t *__restrict__ info, int n6) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = n6 + 1;
  float *A = sm;                  // (n6 + 1) x ld: rows 0..n6-1 = S, row n6 = y
  float *t = sm + (n6 + 1) * ld;  // n6: back-substitution vector
  float *rd = t + n6;             // n6: 1 / L[j][j]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ty = tid >> 5, tx = tid & 31, nty = nt >> 5;
  for (int q = tid; q < n6 * n6; q += nt) {
    const int r = q / n6, c = q - r * n6;
    A[r * ld + c] = S[q];
  }
  for (int q = tid; q < n6; q += nt) A[n6 * ld + q] = yv[q];
  __syncthreads();
  bool bad = false;
  for (int j = 0; j < n6; j++) {
    const float d = A[j * ld + j];
    bad |= !(d > 0.0f);
    const float rinv = 1.0f / sqrtf(d), dinv = 1.0f / d;
    if (tid == 0) rd[j] = rinv;
    // scaled column -> upper triangle (rows j+1..n6 of column j)
    for (int i = j + 1 + tid; i <= n6; i += nt) A[j * ld + i] = A[i * ld + j] * rinv;
    // trailing lower triangle incl. the rhs row: (i, k), j < k <= i <= n6, k < n6
    for (int i = j + 1 + ty; i <= n6; i += nty) {
      const float aij = A[i * ld + j] * dinv;
      const int kmax = i < n6 ? i : n6 - 1;
      for (int k = j + 1 + tx; k <= kmax; k += 32) A[i * ld + k] = A[i * ld + k] - aij * A[k * ld + j];
    }
    __syncthreads();
  }
  if (bad && tid == 0 && info) atomicOr(info, 1);
  // z = column n6 of the upper triangle;  L' x = z, column oriented; L[k][i] sits at A[i][k]
  for (int q = tid; q < n6; q += nt) t[q] = A[q * ld + n6];
  __syncthreads();
  for (int k = n6 - 1; k >= 0; k--) {
    // a non-positive pivot (S indefinite by rounding): the reference's Eigen LLT would silently return NaN poses
    // (ba_cuda.cu:549-552) and the tracker never recovers; here the pose step of this iteration is dropped
    const float xk = bad ? 0.0f : t[k] * rd[k];
    if (tid == 0) dX[k] = xk;
    for (int i = tid; i < k; i += nt) t[i] = t[i] - A[i * ld + k] * xk;
    __syncthreads();
  }
}

// single-wavefront variant for 6N <= 63 (default.yaml: 60).  Lane i keeps ROW i of the matrix in 64
// registers; column values are broadcast with v_readlane (constant lane index after full
// unrolling): straight-line readlane + fma pairs, no barriers.
//   * the right-hand side rides along as row n6 of the lower triangle, so the factorisation leaves
//     z = L^-1 y in that row (no separate forward substitution);
//   * 1/L[j][j] is computed once per column (uniform) and kept by lane j, so neither the column
//     scaling nor the back substitution divides;
//   * the back substitution is column oriented: lane i reads L[r][i] from an LDS transpose and
//     subtracts L[r][i] x_r -- one readlane pair + fma per step instead of a wave reduction.
__device__ __forceinline__ float bcast_lane(float v, int lane_const) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_const));
}

__global__ void __launch_bounds__(64)
    ba_chol64_kernel(const float *__restrict__ S, const float *__restrict__ yv,
                     float *__restrict__ dX, int32_t *__restrict__ info, int n6) {
  __shared__ float Lt[64 * 65];          // Lt[c * 65 + i] = L[i][c]
  const int i = threadIdx.x;
  float a[64];
#pragma unroll
  for (int k = 0; k < 64; k++) {
    float v = (k == i) ? 1.0f : 0.0f;                       // identity padding keeps unused columns inert
    if (i < n6 && k < n6) v = (k <= i) ? S[(size_t)i * n6 + k] : 0.0f;
    if (i == n6 && k < n6) v = yv[k];                       // right-hand side as row n6
    a[k] = v;
  }
  bool bad = false;
  float rdiag = 1.0f;                                       // lane j: 1 / L[j][j]
#pragma unroll
  for (int j = 0; j < 63; j++) {
    const float dgn = bcast_lane(a[j], j);
    bad |= (j < n6) && !(dgn > 0.0f);
    const float rinv = 1.0f / sqrtf(dgn);
    if (i == j) rdiag = rinv;
    a[j] = (i == j) ? sqrtf(dgn) : a[j] * rinv;             // rows above j hold don't-care values
#pragma unroll
    for (int k = j + 1; k < 64; k++) {
      const float lkj = bcast_lane(a[j], k);
      a[k] = __builtin_fmaf(-a[j], lkj, a[k]);              // only rows i >= k are ever read
    }
  }
  if (bad && i == 0 && info) atomicOr(info, 1);
  // z = row n6;  L' x = z, column oriented
#pragma unroll
  for (int c = 0; c < 64; c++) Lt[c * 65 + i] = a[c];
  __syncthreads();
  float z = (i < n6) ? Lt[i * 65 + n6] : 0.0f;             // z_i = L[n6][i]
  float x = 0.0f;
  for (int r = n6 - 1; r >= 0; r--) {
    const float xr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), r)) *
                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rdiag), r));
    if (i == r) x = xr;
    z = __builtin_fmaf(-Lt[i * 65 + r], xr, z);             // L[r][i]; lanes i >= r are no longer needed
  }
  if (i < n6) dX[i] = bad ? 0.0f : x;                      // see ba_chol_kernel: a failed factorisation drops the pose step
}


// ---- ba_asmchol_kernel (RAMP_BA_ASMCHOL=1)
// ------------------------------------------------------------------ K5 + K6 in one launch (windows of <= 16 poses)
// The assembly was a launch of N x N small workgroups whose only consumer is the single workgroup of the factorisation: here
// that workgroup forms S's lower triangle and the right-hand side straight into its LDS matrix -- the same sums in the same
// order as ba_assemble2_kernel (pair records of a block in ascending record order, dealt to BA_AP partial sums that are added
// in lane order; split-K partials in z order; the damping), so S, y and dX are bit-identical to the two-launch path -- and
// factors it.  One wave per pose lists the pair records that touch it (ballot prefix, ascending); a table gives the record of
// every ordered pair for the off-diagonal blocks.
// (cap = list entries per pose: BA_MAXLIST where the LDS holds it -- windows of <= 10 poses --, half of it above: a pose with
// more pair records is flagged in *info, bit 1, like the two-launch path's overflow)
template <int TG>
__global__ void __launch_bounds__(TG * TG)
    ba_asmchol_kernel(const float *__restrict__ pairs, const int32_t *__restrict__ pair_ij, const int32_t *__restrict__ npairs,
                      const float *__restrict__ S_part, const float *__restrict__ y_part, float *__restrict__ dX,
                      int32_t *__restrict__ info, int n6, int KS, int BA_AC_LIST) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = n6 + 1, N = n6 / 6;
  float *A = sm;
  float *xv = sm + (n6 + 1) * ld;
  float *Lk = xv + n6;
  int *s_list = reinterpret_cast<int *>(Lk + N * 28);       // [N][BA_AC_LIST] record numbers
  int *s_li = s_list + N * BA_AC_LIST;                       // their i
  int *s_lj = s_li + N * BA_AC_LIST;                         // their j
  int *s_nl = s_lj + N * BA_AC_LIST;                         // [N]
  int *s_pid = s_nl + N;                                     // [N][N]: record of the ordered pair (i, j), -1 if none
  const int tid = threadIdx.x, nt = TG * TG, lane = tid & 63, wave = tid >> 6;
  const int np = *npairs;
  BA_AC_T(0);
  for (int q = tid; q < N * N; q += nt) s_pid[q] = -1;
  __syncthreads();
  if (wave < N) {                               // (TG = 32: 16 waves, N <= 16)
    const int a = wave;
    int base = 0;
    constexpr int LR = 8;                       // rounds loaded together
    for (int g00 = 0; g00 < np; g00 += 64 * LR) {
    int2 pv[LR];
#pragma unroll
    for (int u = 0; u < LR; u++) {
      const int g = g00 + 64 * u + lane;
      pv[u] = g < np ? reinterpret_cast<const int2 *>(pair_ij)[g] : make_int2(-2, -2);
    }
#pragma unroll
    for (int u = 0; u < LR; u++) {
      const int g0 = g00 + 64 * u;
      if (g0 >= np) break;
      const int g = g0 + lane;
      const int pi = pv[u].x, pj = pv[u].y;
      const bool hit = pi == a || pj == a;
      const unsigned long long m = __ballot(hit);
      const int off = base + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && off < BA_AC_LIST) { s_list[a * BA_AC_LIST + off] = g; s_li[a * BA_AC_LIST + off] = pi; s_lj[a * BA_AC_LIST + off] = pj; }
      if (pi == a && pj >= 0 && pj < N) s_pid[a * N + pj] = g;       // (one record per ordered pair: the records are the pair groups)
      base += __popcll(m);
    }
    }
    if (lane == 0) {
      s_nl[a] = min(base, BA_AC_LIST);
      if (base > BA_AC_LIST && info) atomicOr(info, 2);              // (never a silent truncation)
    }
  }
  __syncthreads();
  BA_AC_T(1);
  // phase A: the partial sums over the pair records of the diagonal blocks' lower entries (21 per pose) and of the gradient
  // (6 per pose), one (entry, partial lane) item per thread like ba_assemble2_kernel: lane pl sums records pl, pl + AP, ...
  // of the pose's list in ascending order; a round's loads are all in flight before its first add
  float *s_part = reinterpret_cast<float *>(s_pid + N * N);       // [N * 27][BA_AP]
  for (int it = tid; it < N * 27 * BA_AP; it += nt) {
    const int e = it / BA_AP, pl = it - e * BA_AP, a = e / 27, k = e - 27 * a;
    int x = 0, y = 0;
    if (k < 21) {
      x = k < 1 ? 0 : (k < 3 ? 1 : (k < 6 ? 2 : (k < 10 ? 3 : (k < 15 ? 4 : 5))));
      y = k - x * (x + 1) / 2;
    } else {
      x = k - 21;
    }
    const int nl = s_nl[a];
    const int *li = s_li + a * BA_AC_LIST, *lj = s_lj + a * BA_AC_LIST, *lg = s_list + a * BA_AC_LIST;
    float acc = 0.f;
    constexpr int RND = 8;
    for (int l0 = pl; l0 < nl; l0 += BA_AP * RND) {
      // a record touches pose a as i or as j: ONE load (its i-block or its j-block); only the self pair (a, a) adds all four
      // blocks (a workgroup's vector memory instructions, not its bytes, are what this single-CU launch pays for)
      float v0[RND], v1[RND], v2[RND], v3[RND];
#pragma unroll
      for (int u = 0; u < RND; u++) {
        const int l = l0 + BA_AP * u;
        const bool on = l < nl;
        const int i = on ? li[l] : -2, j = on ? lj[l] : -2;
        const bool self = i == a && j == a;
        const float *pr = pairs + (size_t)(on ? lg[l] : lg[0]) * BA_PAIR + (k < 21 ? x * 6 + y : 144 + x);
        const int step = k < 21 ? 36 : 6;
        const float first = pr[i == a ? 0 : step];
        v0[u] = i == a ? first : 0.f;
        v1[u] = j == a ? (self ? pr[step] : first) : 0.f;
        v2[u] = (self && k < 21) ? pr[72] : 0.f; v3[u] = (self && k < 21) ? pr[108] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < RND; u++) {
        const int l = l0 + BA_AP * u;
        if (l >= nl) break;
        const int i = li[l], j = lj[l];
        if (i == a) acc += v0[u];
        if (j == a) acc += v1[u];
        if (k < 21) {
          if (i == a && j == a) acc += v2[u];
          if (j == a && i == a) acc += v3[u];
        }
      }
    }
    s_part[it] = acc;
  }
  __syncthreads();
  BA_AC_T(2);
  // phase B: lower triangle in pieces of four columns (r, 4 c4 .. 4 c4 + 3), then the right-hand side; the split-K partials
  // in z order, 16 bytes per load, 16 loads in flight
  const int n4 = n6 / 4;                                     // (6N is a multiple of 4 for even N; odd N: scalar tail column pieces)
  const int row4 = (n6 + 3) / 4;
  for (int q = tid; q < n6 * row4 + n6; q += nt) {
    const bool rhs = q >= n6 * row4;
    const int r = rhs ? q - n6 * row4 : q / row4, c4 = rhs ? 0 : q - r * row4;
    if (!rhs && 4 * c4 > r) continue;
    const int a = r / 6, x = r - 6 * a;
    float sp[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = !rhs && c4 < n4 && (n6 & 3) == 0;
    const int ncol = rhs ? 1 : min(4, n6 - 4 * c4);
    for (int z0 = 0; z0 < KS; z0 += 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int z = min(z0 + u, KS - 1);
        if (rhs) {
          v[u] = make_float4(y_part[(size_t)z * n6 + r], 0.f, 0.f, 0.f);
        } else if (vec) {
          v[u] = *reinterpret_cast<const float4 *>(S_part + ((size_t)z * n6 + r) * n6 + 4 * c4);
        } else {
          const float *sr = S_part + ((size_t)z * n6 + r) * n6 + 4 * c4;
          v[u] = make_float4(sr[0], ncol > 1 ? sr[1] : 0.f, ncol > 2 ? sr[2] : 0.f, ncol > 3 ? sr[3] : 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (z0 + u >= KS) break;
        sp[0] += v[u].x; sp[1] += v[u].y; sp[2] += v[u].z; sp[3] += v[u].w;
      }
    }
    for (int e = 0; e < ncol; e++) {
      const int c = 4 * c4 + e;
      if (!rhs && c > r) break;
      const int b = c / 6, y = c - 6 * b;
      float bs = 0.f;
      if (rhs || a == b) {
        const float *pp = s_part + ((size_t)a * 27 + (rhs ? 21 + x : x * (x + 1) / 2 + y)) * BA_AP;
#pragma unroll
        for (int k = 0; k < BA_AP; k++) bs += pp[k];
      } else {
        // (a, b), a > b: the records of the ordered pairs (a, b) and (b, a), in ascending record order, on partial lanes 0 / 1
        const int g1 = s_pid[a * N + b], g2 = s_pid[b * N + a];
        const int lo = g1 < 0 ? g2 : (g2 < 0 ? g1 : min(g1, g2)), hi = (g1 >= 0 && g2 >= 0) ? max(g1, g2) : -1;
        float p0 = 0.f, p1 = 0.f;
        if (lo >= 0) p0 += pairs[(size_t)lo * BA_PAIR + (lo == g1 ? 72 : 108) + x * 6 + y];
        if (hi >= 0) p1 += pairs[(size_t)hi * BA_PAIR + (hi == g1 ? 72 : 108) + x * 6 + y];
        bs += p0; bs += p1;
      }
      const float spe = e == 0 ? sp[0] : (e == 1 ? sp[1] : (e == 2 ? sp[2] : sp[3]));
      if (rhs) {
        A[n6 * ld + r] = bs - spe;
      } else {
        float sv = bs - spe;
        if (r == c) sv += (1e-4f * sv + 1.0f);
        A[r * ld + c] = sv;
      }
    }
  }
  BA_AC_T(3);
  ba_cholb_body<TG>(A, xv, Lk, dX, info, n6);
  BA_AC_T(4);
}
#ifdef BA_AC_TIMING
extern "C" int ramp_debug_ba_times(long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(ba_ac_t), 8 * sizeof(long)) == hipSuccess ? 0 : 1; }
#endif

