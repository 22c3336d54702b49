// Which packed-fp32 instruction goes wrong, and how?  (round 6; companion of pk_hazard.hip)
// Victim on stream A: every lane evaluates v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (inline asm: exactly these
// instructions, independent of the vectoriser) and their scalar forms on the SAME static operands, REPS operand sets per
// lane, and stores all results.  Aggressor on stream B: libramp_hip.so's 1x1 64 -> 384 conv layer from a hipGraph.  Every
// result is compared with the host's correctly rounded value (float arithmetic under fesetround) -- a wrong word is
// classified: which op, which lane quarter, which half of the pair, how many ulps, and whether it equals the product / sum
// under another rounding mode.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I include -o /tmp/pk_probe tools/mb/pk_probe.hip -L rampvo_amd/csrc -lramp_hip
#include <hip/hip_runtime.h>
#include <cfenv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int REPS = 32, NOPS = 16;    // ops: pk_mul, pk_add, pk_fma, s_mul, s_add, s_fma, then the modifier forms the SLP-vectorised
                                       // transform kernel contains (op_sel / op_sel_hi / neg_lo / neg_hi, v_pk_mov_b32) -- checked against the probe's own first run

__global__ void __launch_bounds__(256) probe_kernel(const f2 *__restrict__ a, const f2 *__restrict__ b, const f2 *__restrict__ c,
                                                    f2 *__restrict__ out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
#pragma unroll 4
  for (int r = 0; r < REPS; r++) {
    const f2 x = a[(size_t)r * n + t], y = b[(size_t)r * n + t], z = c[(size_t)r * n + t];
    f2 m, s, f;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(x), "v"(y));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(f) : "v"(x), "v"(y), "v"(z));
    f2 sm, ss, sf;
    sm[0] = x[0] * y[0]; sm[1] = x[1] * y[1];
    ss[0] = x[0] + y[0]; ss[1] = x[1] + y[1];
    sf[0] = __builtin_fmaf(x[0], y[0], z[0]); sf[1] = __builtin_fmaf(x[1], y[1], z[1]);
    f2 m1, m2, a1, a2, mv, a3;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(m1) : "v"(x), "v"(y));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(m2) : "v"(x), "v"(y));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a1) : "v"(x), "v"(y));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(a2) : "v"(x), "v"(y));
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(mv) : "v"(x), "v"(y));
    asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(a3) : "v"(x));
    // dependent mixes of packed ops with the division / reciprocal / square-root sequences (the victim's quaternion
    // normalisation and projection): IEEE division result -> packed multiply; packed add -> division; v_rcp_f32 -> packed
    // multiply; packed square + sqrt + division -> packed multiply
    f2 d1, d2, d3, d4;
    {
      const float q = x[0] / y[1];
      const f2 qq = (f2){q, q};
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d1) : "v"(qq), "v"(z));
      f2 tt;
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(tt) : "v"(x), "v"(y));
      d2 = (f2){1.0f / tt[0], tt[1]};
      const float rc = __builtin_amdgcn_rcpf(x[0]);
      const f2 rr = (f2){rc, rc};
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d3) : "v"(rr), "v"(y));
      f2 sq;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(sq) : "v"(x), "v"(x));
      const float nrm = 1.0f / __builtin_sqrtf(sq[0] + sq[1]);
      const f2 nn = (f2){nrm, nrm};
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d4) : "v"(nn), "v"(x));
    }
    f2 *o = out + ((size_t)r * NOPS) * n + t;
    o[12 * (size_t)n] = d1; o[13 * (size_t)n] = d2; o[14 * (size_t)n] = d3; o[15 * (size_t)n] = d4;
    o[0 * (size_t)n] = m; o[1 * (size_t)n] = s; o[2 * (size_t)n] = f;
    o[3 * (size_t)n] = sm; o[4 * (size_t)n] = ss; o[5 * (size_t)n] = sf;
    o[6 * (size_t)n] = m1; o[7 * (size_t)n] = m2; o[8 * (size_t)n] = a1; o[9 * (size_t)n] = a2; o[10 * (size_t)n] = mv; o[11 * (size_t)n] = a3;
  }
}

extern "C" int ramp_conv2d_nhwc(const void *x, const void *wpk, const float *bias, const float *pre_scale,
                                const float *pre_shift, const void *res, void *y, float *stats, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int relu, float out_scale, int dtype, void *stream);

static float host_op(int op, float x, float y, float z, int mode) {
  fesetround(mode);
  volatile float vx = x, vy = y, vz = z, r;
  if (op == 0) r = vx * vy; else if (op == 1) r = vx + vy; else r = fmaf(vx, vy, vz);
  fesetround(FE_TONEAREST);
  return r;
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 400, aggressor = argc > 2 ? atoi(argv[2]) : 1;
  const int n = 40960;                       // 160 workgroups of 256 lanes
  std::vector<f2> a((size_t)REPS * n), b(a.size()), c(a.size());
  srand(99);
  auto rnd = [] { return ((float)rand() / RAND_MAX - 0.5f) * 200.0f; };
  for (size_t i = 0; i < a.size(); i++) { a[i] = (f2){rnd(), rnd()}; b[i] = (f2){rnd() * 0.01f, rnd()}; c[i] = (f2){rnd(), rnd() * 3.f}; }
  f2 *d_a, *d_b, *d_c, *d_out;
  const size_t nout = (size_t)REPS * NOPS * n;
  CK(hipMalloc(&d_a, a.size() * 8)); CK(hipMalloc(&d_b, a.size() * 8)); CK(hipMalloc(&d_c, a.size() * 8)); CK(hipMalloc(&d_out, nout * 8));
  CK(hipMemcpy(d_a, a.data(), a.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, b.data(), a.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_c, c.data(), a.size() * 8, hipMemcpyHostToDevice));
  // host reference (round to nearest): op o in {mul, add, fma}
  std::vector<f2> ref(nout);
  for (int r = 0; r < REPS; r++)
    for (int t = 0; t < n; t++) {
      const f2 x = a[(size_t)r * n + t], y = b[(size_t)r * n + t], z = c[(size_t)r * n + t];
      for (int o = 0; o < 6; o++)
        ref[((size_t)r * NOPS + o) * n + t] = (f2){host_op(o % 3, x[0], y[0], z[0], FE_TONEAREST), host_op(o % 3, x[1], y[1], z[1], FE_TONEAREST)};
    }
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  _Float16 *d_cx, *d_cw, *d_cy; float *d_cb;
  CK(hipMalloc(&d_cx, 120 * 160 * 64 * 2)); CK(hipMalloc(&d_cw, 64 * 384 * 2)); CK(hipMalloc(&d_cy, 120 * 160 * 384 * 2)); CK(hipMalloc(&d_cb, 384 * 4));
  CK(hipMemset(d_cx, 0x3c, 120 * 160 * 64 * 2)); CK(hipMemset(d_cw, 0x2c, 64 * 384 * 2)); CK(hipMemset(d_cb, 0, 384 * 4));
  hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
  if (aggressor) {
    CK(hipStreamBeginCapture(sb, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 8; k++)
      if (ramp_conv2d_nhwc(d_cx, d_cw, d_cb, nullptr, nullptr, nullptr, d_cy, nullptr, 120, 160, 64, 384, 1, 1, 1, 0, 0.25f, 1, sb) != 0) { printf("conv failed\n"); return 1; }
    CK(hipStreamEndCapture(sb, &graph));
    CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
  }
  std::vector<f2> got(nout);
  const char *opn[NOPS] = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "scalar mul", "scalar add", "scalar fma",
                           "pk_mul op_sel_hi:[0,1]", "pk_mul op_sel_hi:[0,1] neg", "pk_add op_sel_hi:[1,0] neg", "pk_add op_sel_hi:[0,1]",
                           "pk_mov op_sel:[1,0]", "pk_add x, 1.0 op_sel_hi", "div -> pk_mul", "pk_add -> div", "v_rcp -> pk_mul",
                           "pk_mul, sqrt, div -> pk_mul"};
  // the modifier forms: reference = the probe's own first run, alone
  hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, sa, d_a, d_b, d_c, d_out, n);
  CK(hipStreamSynchronize(sa));
  CK(hipMemcpy(got.data(), d_out, nout * 8, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nout; i++) if ((i / n) % NOPS >= 6) ref[i] = got[i];
  long wrong[NOPS][4][2] = {}, as_mode[NOPS][4] = {}, ulp1[NOPS] = {}, other[NOPS] = {}, launches_bad = 0, launches = 0;
  const int modes[4] = {FE_TOWARDZERO, FE_UPWARD, FE_DOWNWARD, FE_TONEAREST};
  int printed = 0;
  for (int rd = 0; rd < rounds; rd++) {
    if (aggressor) CK(hipGraphLaunch(gexec, sb));
    for (int rep = 0; rep < 4; rep++) {
      hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, sa, d_a, d_b, d_c, d_out, n);
      CK(hipStreamSynchronize(sa));
      CK(hipMemcpy(got.data(), d_out, nout * 8, hipMemcpyDeviceToHost));
      launches++;
      bool bad = false;
      for (size_t i = 0; i < nout; i++) {
        for (int h = 0; h < 2; h++) {
          const float g = got[i][h], e = ref[i][h];
          if (memcmp(&g, &e, 4) == 0) continue;
          bad = true;
          const int t = (int)(i % n), o = (int)((i / n) % NOPS), r = (int)(i / n / NOPS);
          wrong[o][(t % 64) / 16][h]++;
          const f2 x = a[(size_t)r * n + t], y = b[(size_t)r * n + t], z = c[(size_t)r * n + t];
          bool matched = false;
          for (int m = 0; m < 3 && !matched && o < 6; m++) {
            const float alt = host_op(o % 3, x[h], y[h], z[h], modes[m]);
            if (memcmp(&alt, &g, 4) == 0) { as_mode[o][m]++; matched = true; }
          }
          if (!matched && o < 6 && o % 3 == 2) {          // fma: the unfused result?
            fesetround(FE_TONEAREST);
            volatile float p = x[h] * y[h]; volatile float q = p + z[h];
            const float alt = q;
            if (memcmp(&alt, &g, 4) == 0) { as_mode[o][3]++; matched = true; }
          }
          int ig, ie; memcpy(&ig, &g, 4); memcpy(&ie, &e, 4);
          if (!matched) { if (abs(ig - ie) == 1) ulp1[o]++; else other[o]++; }
          if (printed < 12) {
            printed++;
            printf("  %-28s lane %2d half %d: x %.9g y %.9g z %.9g -> got %.9g (0x%08x) expected %.9g (0x%08x)\n", opn[o], t % 64, h, x[h], y[h], z[h], g, ig, e, ie);
          }
        }
      }
      launches_bad += bad;
    }
    if (aggressor) CK(hipStreamSynchronize(sb));
  }
  printf("%s: %ld of %ld launches of the probe with wrong words\n", aggressor ? "next to the conv layer" : "alone", launches_bad, launches);
  for (int o = 0; o < NOPS; o++) {
    long tot = 0;
    for (int q4 = 0; q4 < 4; q4++) tot += wrong[o][q4][0] + wrong[o][q4][1];
    printf("  %-28s wrong words %8ld | by lane quarter %ld / %ld / %ld / %ld | low / high half of the pair %ld / %ld | equals toward-zero %ld, upward %ld, downward %ld, unfused %ld; other 1-ulp %ld, other %ld\n",
           opn[o], tot, wrong[o][0][0] + wrong[o][0][1], wrong[o][1][0] + wrong[o][1][1], wrong[o][2][0] + wrong[o][2][1], wrong[o][3][0] + wrong[o][3][1],
           wrong[o][0][0] + wrong[o][1][0] + wrong[o][2][0] + wrong[o][3][0], wrong[o][0][1] + wrong[o][1][1] + wrong[o][2][1] + wrong[o][3][1],
           as_mode[o][0], as_mode[o][1], as_mode[o][2], as_mode[o][3], ulp1[o], other[o]);
  }
  return 0;
}
