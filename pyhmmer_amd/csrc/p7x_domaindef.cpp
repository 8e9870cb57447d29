// p7x_domaindef.cpp -- host-side domain definition for the targets that survive the Forward filter.
//
// Restates upstream p7_domaindef.c:p7_domaindef_ByPosteriorHeuristics (reference
// include/libhmmer/p7_domaindef.pxd:23-72) and the routines it drives:
//   impl_sse/decoding.c   p7_DomainDecoding, p7_Decoding
//   impl_sse/fwdback.c    p7_Forward, p7_Backward (full matrices, on envelopes only)
//   impl_sse/null2.c      p7_Null2_ByExpectation, p7_Null2_ByTrace
//   impl_sse/optacc.c     p7_OptimalAccuracy, p7_OATrace
//   impl_sse/stotrace.c   p7_StochasticTrace          (Easel "fast" LCG, re-seeded per region)
//   p7_spensemble.c       p7_spensemble_Add/_Cluster  (Easel single-linkage clustering)
//   p7_trace.c            p7_trace_Index;  p7_alidisplay.c p7_alidisplay_Create
// Matrices are un-striped (node k at index k).  Where upstream's results depend on the order in which
// the striped SSE code visits cells (tie-breaks in the OA traceback, cumulative sums in the stochastic
// traceback) that visiting order is reproduced.  Runs only for ~1e-5 of random targets plus true homologs.
#include "p7x_host.hpp"
#include "p7x_choice.hpp"
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace p7x {

int vit_pick_C(int M);          // p7x_vitfwd.hip: nodes per lane of the wave-per-target kernels, a function of M alone

// optional host profile (option "host_profile"): accumulated nanoseconds per phase, printed by host_prof_dump()
#define g_prof_on (debug_opt(OPT_HOST_PROFILE) > 0)
static std::atomic<long long> g_prof_ns[12];
static const char *g_prof_name[12] = { "domain_decoding", "region_forward", "stochastic_traces", "null2_by_trace", "cluster",
                                        "env_forward", "env_backward", "env_decoding", "optimal_accuracy", "oa_trace+display",
                                        "null2_expect", "other" };
struct ProfScope {
  int id; std::chrono::steady_clock::time_point t0;
  explicit ProfScope(int i) : id(i) { if (g_prof_on) t0 = std::chrono::steady_clock::now(); }
  ~ProfScope() { if (g_prof_on) g_prof_ns[id] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
void host_prof_dump()
{
  if (!g_prof_on) return;
  for (int i = 0; i < 12; ++i) { std::fprintf(stderr, "[p7x host prof] %-18s %9.3f ms\n", g_prof_name[i], g_prof_ns[i].load() / 1e6); g_prof_ns[i] = 0; }
}

namespace {

// hot loops are compiled twice (AVX2 and baseline x86-64) and dispatched at load time
#if defined(__HIP_DEVICE_COMPILE__)
#define P7X_MULTIVERSION
#else
#define P7X_MULTIVERSION __attribute__((target_clones("avx2", "default")))
#endif

enum { sM = 1, sD = 2, sI = 3, sS = 4, sN = 5, sB = 6, sE = 7, sC = 8, sT = 9, sJ = 10 };   // p7T_* (p7_trace.pxd)
enum { xE_ = 0, xN_ = 1, xJ_ = 2, xB_ = 3, xC_ = 4, xS_ = 5, NX = 6 };

// ---------------------------------------------------------------- Easel "fast" RNG (esl_randomness_CreateFast)
struct FastRng {
  uint32_t seed = 42, x = 0;
  static uint32_t mix3(uint32_t a, uint32_t b, uint32_t c)
  {
    a -= b; a -= c; a ^= (c >> 13);  b -= c; b -= a; b ^= (a << 8);   c -= a; c -= b; c ^= (b >> 13);
    a -= b; a -= c; a ^= (c >> 12);  b -= c; b -= a; b ^= (a << 16);  c -= a; c -= b; c ^= (b >> 5);
    a -= b; a -= c; a ^= (c >> 3);   b -= c; b -= a; b ^= (a << 10);  c -= a; c -= b; c ^= (b >> 15);
    return c;
  }
  // esl_randomness_Init for the LCG type: the seed is dispersed with Jenkins' mix3, never zero
  void init(uint32_t s) { seed = s; x = mix3(s, 87654321u, 12345678u); if (x == 0) x = 42; }
  uint32_t next_u32() { x = lcg_next(x); return x; }
  double next() { return (double) next_u32() / 4294967296.0; }
};

// esl_vec_FSum (Kahan).  Same arithmetic as kahan_fsum(); this translation unit is compiled without fast-math and
// without contraction, so the compensation survives without the volatile round trips through memory.
static inline float fsum(const float *v, int n)
{
  float sum = 0.0f, c = 0.0f;
  for (int i = 0; i < n; ++i) { const float y = v[i] - c; const float t = sum + y; c = (t - sum) - y; sum = t; }
  return sum;
}
void fnorm(float *v, int n)
{
  const float s = fsum(v, n);
  if (s != 0.0f) for (int i = 0; i < n; ++i) v[i] /= s;
  else           for (int i = 0; i < n; ++i) v[i] = 1. / (float) n;
}
// ---------------------------------------------------------------- the wavefront's combining trees, lane by lane
// Host restatement of p7x_wave.hpp (affine_scan_up / affine_scan_down / wave_sum_f32): the same operations on the same
// operands in the same order, so that a sum formed here and on the device is the same float.  A DPP step whose source
// lane lies outside the 16-lane row (or whose row is masked) leaves the identity (a = 0, p = 1) in place of the operand.
static inline void lanes_scan_up(float *sa, float *sp)
{ // inclusive scan of the affine maps x -> a_l + p_l x over lanes 0..l: Kogge-Stone inside the rows, then row_bcast:15 / :31
  float a[64], q[64];
  for (int s = 1; s <= 8; s <<= 1) {
    for (int l = 0; l < 64; ++l) {
      const bool src = (l & 15) >= s;
      const float pa = src ? sa[l - s] : 0.0f, pp = src ? sp[l - s] : 1.0f;
      a[l] = sa[l] + pa * sp[l]; q[l] = sp[l] * pp;
    }
    for (int l = 0; l < 64; ++l) { sa[l] = a[l]; sp[l] = q[l]; }
  }
  for (int row = 1; row < 4; row += 2) {            // rows 1 and 3 take the last lane of the row before them
    const float pa = sa[16 * row - 1], pp = sp[16 * row - 1];
    for (int l = 16 * row; l < 16 * row + 16; ++l) { sa[l] = sa[l] + pa * sp[l]; sp[l] = sp[l] * pp; }
  }
  {                                                 // rows 2 and 3 take lane 31
    const float pa = sa[31], pp = sp[31];
    for (int l = 32; l < 64; ++l) { sa[l] = sa[l] + pa * sp[l]; sp[l] = sp[l] * pp; }
  }
}
static inline void lanes_scan_down(float *sa, float *sp)
{ // the same over lanes l..63
  float a[64], q[64];
  for (int s = 1; s <= 8; s <<= 1) {
    for (int l = 0; l < 64; ++l) {
      const bool src = (l & 15) + s <= 15;
      const float pa = src ? sa[l + s] : 0.0f, pp = src ? sp[l + s] : 1.0f;
      a[l] = sa[l] + pa * sp[l]; q[l] = sp[l] * pp;
    }
    for (int l = 0; l < 64; ++l) { sa[l] = a[l]; sp[l] = q[l]; }
  }
  {
    const float a48 = sa[48], p48 = sp[48], a16 = sa[16], p16 = sp[16];
    for (int l = 0; l < 64; ++l) {
      const int row = l >> 4;
      const float pa = (row == 2) ? a48 : ((row == 0) ? a16 : 0.0f), pp = (row == 2) ? p48 : ((row == 0) ? p16 : 1.0f);
      sa[l] = sa[l] + pa * sp[l]; sp[l] = sp[l] * pp;
    }
  }
  {
    const float a32 = sa[32], p32 = sp[32];
    for (int l = 0; l < 64; ++l) {
      const float pa = (l < 32) ? a32 : 0.0f, pp = (l < 32) ? p32 : 1.0f;
      sa[l] = sa[l] + pa * sp[l]; sp[l] = sp[l] * pp;
    }
  }
}
static inline float lanes_sum(float *v)
{ // wave_sum_f32: the value lane 63 ends up with
  float t[64];
  for (int s = 1; s <= 8; s <<= 1) {
    for (int l = 0; l < 64; ++l) t[l] = v[l] + (((l & 15) >= s) ? v[l - s] : 0.0f);
    for (int l = 0; l < 64; ++l) v[l] = t[l];
  }
  for (int row = 1; row < 4; row += 2) { const float b = v[16 * row - 1]; for (int l = 16 * row; l < 16 * row + 16; ++l) v[l] = v[l] + b; }
  { const float b = v[31]; for (int l = 32; l < 64; ++l) v[l] = v[l] + b; }
  return v[63];
}

// ---------------------------------------------------------------- the query in a given configuration
// Long-target (nhmmer) variant of envelope rescoring, upstream rescore_isolated_domain(..., long_target = TRUE, ...)
struct LongTargetOpts {
  bool do_null2 = true;                // false: upstream passes scores_arr == NULL and the model is not re-parameterised
  const float *match_prob = nullptr;   // [M+1][K] match emission probabilities of the core model (fwd_emissions_arr)
  int max_env_extra = 20;              // an envelope is trimmed to its alignment +- this many residues
};

struct Model {
  const Profile *p;
  int M;
  float xf[4][2];                 // [E,N,J,C][MOVE,LOOP] for the current mode / length
  const float *rf_over = nullptr; // [Kp][M+1] replacement match odds (long targets: composition-adjusted background)
  const LongTargetOpts *lt = nullptr;
  const float *tf(int t) const { return p->tf.data() + (size_t) t * (M + 1); }
  const float *rf(int x) const { return (rf_over ? rf_over : p->rf_.data()) + (size_t) x * (M + 1); }
  // Order of operations.  The sums whose result depends on the order of the float additions -- the D->D chains, the row
  // sums xE / xB, the null2 expectation -- run in the order of the device kernels (p7x_envelope.hip, p7x_wave.hpp): lane z
  // of a 64-lane wavefront owns the C consecutive nodes zC+1 .. zC+C, walks them in order, and the lanes are combined by
  // the wavefront's scan / reduction trees (lanes_scan_up / lanes_scan_down / lanes_sum below).  Host twin and device
  // therefore produce the same bits, and every discrete decision taken from them (optimal-accuracy traceback, the
  // stochastic tracebacks' choices) is the same decision.  Upstream's own order (four striped lanes, serial D->D sweeps)
  // is a third one; what the fixtures pin is reproduced by all of them.
  int C = 1;                                        // nodes per lane: vit_pick_C(M), the device image's choice
  float ddprod[64];                                 // product of the D->D transitions of a lane's nodes, in node order
  // upstream = true (the default; option "host_order" = 1 selects the device's order instead): those sums run as
  // impl_sse runs them -- four stripes, node k in stripe (k-1)/Q at position (k-1)%Q, serial D->D sweeps, row sums
  // stripe by stripe and then (s0+s1)+(s2+s3) -- so that every float, and with it every decision, is the reference's.
  // The host stage uses it for whatever it computes itself, in particular for the envelopes and regions the device
  // flags as too close to call (near-tie guards of p7x_envelope.hip / p7x_ensemble.hip).
  bool upstream = true;
  int Q = 0;
  std::vector<float> st;                            // [8][Q][4] transitions in the striped layout (padding: 0)
  const float *sT(int t) const { return st.data() + (size_t) t * Q * 4; }
  // [Kp][Q][4] match odds in the striped layout, of whatever rf() currently stands for (rebuilt when that changes: the
  // long-target path swaps in composition-adjusted odds per envelope)
  mutable std::vector<float> sr; mutable const float *sr_of = nullptr;
  const float *sR(int x) const
  {
    const float *base = rf(0);
    if (sr_of != base || sr.size() != (size_t) p->Kp * Q * 4) {
      sr.assign((size_t) p->Kp * Q * 4, 0.0f);
      for (int y = 0; y < p->Kp; ++y) {
        const float *r = rf(y);
        float *d = sr.data() + (size_t) y * Q * 4;
        for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) { const int k = q + 1 + z * Q; if (k <= M) d[q * 4 + z] = r[k]; }
      }
      sr_of = base;
    }
    return sr.data() + (size_t) x * Q * 4;
  }
  std::vector<float> rfT;                           // [M+1][kKpad] match odds, residue-minor (null2_by_trace)
  static constexpr int kKpad = 24;
  void prepare_rfT()
  {
    rfT.assign((size_t) (M + 1) * kKpad, 0.0f);
    for (int x = 0; x < p->K && x < kKpad; ++x) { const float *r = rf(x); for (int k = 1; k <= M; ++k) rfT[(size_t) k * kKpad + x] = r[k]; }
  }
  void prepare(int order = -1)
  {
    upstream = order >= 0 ? order == 0 : debug_opt(OPT_HOST_ORDER) <= 0;
    Q = p->Q4();
    st.assign((size_t) 8 * Q * 4, 0.0f);
    for (int t = 0; t < 8; ++t) {
      const float *src = tf(t);
      const int last = (t == 4 || t == 7) ? M - 1 : M;        // M -> D and D -> D do not leave node M
      for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) { const int k = q + 1 + z * Q; if (k <= last) st[((size_t) t * Q + q) * 4 + z] = src[k]; }
    }
    C = vit_pick_C(M);
    if (C <= 0) C = (M + 63) / 64;                  // beyond the device kernels' reach: the same rule, continued
    const float *tDD = tf(7);
    for (int z = 0; z < 64; ++z) {
      float pr = 1.0f;
      for (int c = 0; c < C; ++c) { const int k = z * C + c + 1; pr *= (k <= M ? tDD[k] : 0.0f); }
      ddprod[z] = pr;
    }
  }
  void configure(bool multihit, int L)
  { // p7_oprofile_ReconfigMultihit / ReconfigUnihit (+ ReconfigLength)
    const float nj = multihit ? 1.0f : 0.0f;
    xf[XE][MOVE] = multihit ? 0.5f : 1.0f;
    xf[XE][LOOP] = multihit ? 0.5f : 0.0f;
    const float pmove = (2.0f + nj) / ((float) L + 2.0f + nj), ploop = 1.0f - pmove;
    for (int s : {XN, XJ, XC}) { xf[s][MOVE] = pmove; xf[s][LOOP] = ploop; }
  }
};

// Full DP matrix: rows 0..L, per row three arrays of M+1 floats (M, I, D) plus the specials.
struct Matrix {
  int M = 0, L = 0;
  std::vector<float> m, i, d, x, scratch;
  float totscale = 0.0f;
  bool own_scales = false;
  void resize(int M_, int L_)
  {
    M = M_; L = L_;
    const size_t n = (size_t) (L + 1) * (M + 2);
    if (m.size() < n) { m.resize(n); i.resize(n); d.resize(n); }
    if (x.size() < (size_t) (L + 1) * NX) x.resize((size_t) (L + 1) * NX);
    if (scratch.size() < (size_t) (M + 4) + 4 * (size_t) (L + 1)) scratch.resize((size_t) (M + 4) + 4 * (size_t) (L + 1));
  }
  float *M_(int r) { return m.data() + (size_t) r * (M + 2); }
  float *I_(int r) { return i.data() + (size_t) r * (M + 2); }
  float *D_(int r) { return d.data() + (size_t) r * (M + 2); }
  const float *M_(int r) const { return m.data() + (size_t) r * (M + 2); }
  const float *I_(int r) const { return i.data() + (size_t) r * (M + 2); }
  const float *D_(int r) const { return d.data() + (size_t) r * (M + 2); }
  float &X(int r, int s) { return x[(size_t) r * NX + s]; }
  float X(int r, int s) const { return x[(size_t) r * NX + s]; }
};

// ---------------------------------------------------------------- p7_Forward (full matrix)
// dsq is 1-indexed over the envelope: residues dsq[1..L].
// D(i,k) = M(i,k-1) tMD(k-1) + D(i,k-1) tDD(k-1), k = 1..M, given the finished M row; returns xE = sum_k M(i,k) + D(i,k).
// EnvForward<C>::row (p7x_envelope.hip) operation for operation: every lane runs its own chain from a zero carry, the
// lanes' affine maps are composed by the scan, the carry enters a lane's nodes through successive products, and the row
// sum takes a lane's match cells, then its delete cells, then the reduction tree.
static float dchain_forward(const Model &om, const float *__restrict mc, float *__restrict dc)
{
  const int M = om.M, C = om.C;
  const float *__restrict tMD = om.tf(4), *__restrict tDD = om.tf(7);
  float sa[64], sp[64], es[64];
  for (int z = 0; z < 64; ++z) {
    const int k0 = z * C + 1;
    float A = 0.0f, e = 0.0f;
    const int k1 = std::min(M, k0 + C - 1);
    for (int k = k0; k <= k1; ++k) { e = e + mc[k]; dc[k] = A; A = mc[k] * tMD[k] + A * tDD[k]; }
    if (k0 + C - 1 > M) A = 0.0f;                   // a padding node closes the chain (its transitions are zero)
    sa[z] = A; sp[z] = om.ddprod[z]; es[z] = e;
  }
  lanes_scan_up(sa, sp);
  for (int z = 0; z < 64; ++z) {
    const int k0 = z * C + 1, k1 = std::min(M, k0 + C - 1);
    float w = z ? sa[z - 1] : 0.0f, e = es[z];
    for (int k = k0; k <= k1; ++k) { dc[k] = dc[k] + w; e = e + dc[k]; w = w * tDD[k]; }
    es[z] = e;
  }
  return lanes_sum(es);
}

P7X_MULTIVERSION int forward_full_lanes(const Model &om, const uint8_t *dsq, int L, Matrix &ox, float *ret_sc)
{
  const int M = om.M;
  ox.resize(M, L);
  const float *__restrict bm = om.tf(0), *__restrict tMM = om.tf(1), *__restrict tIM = om.tf(2), *__restrict tDM = om.tf(3),
              *__restrict tMI = om.tf(5), *__restrict tII = om.tf(6);
  float *m0 = ox.M_(0), *i0 = ox.I_(0), *d0 = ox.D_(0);
  for (int k = 0; k <= M + 1; ++k) m0[k] = i0[k] = d0[k] = 0.0f;
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = om.xf[XN][MOVE], xC = 0.f;
  ox.X(0, xE_) = xE; ox.X(0, xN_) = xN; ox.X(0, xJ_) = xJ; ox.X(0, xB_) = xB; ox.X(0, xC_) = xC; ox.X(0, xS_) = 1.0f;
  ox.totscale = 0.0f; ox.own_scales = true;
  for (int r = 1; r <= L; ++r) {
    const float *__restrict rf = om.rf(dsq[r]);
    const float *__restrict mp = ox.M_(r - 1), *__restrict ip = ox.I_(r - 1), *__restrict dp = ox.D_(r - 1);
    float *__restrict mc = ox.M_(r), *__restrict ic = ox.I_(r), *__restrict dc = ox.D_(r);
    mc[0] = ic[0] = dc[0] = 0.0f;
    for (int k = 1; k <= M; ++k) {
      float sv = xB * bm[k];
      sv = sv + mp[k - 1] * tMM[k];
      sv = sv + ip[k - 1] * tIM[k];
      sv = sv + dp[k - 1] * tDM[k];
      mc[k] = sv * rf[k];
      ic[k] = mp[k] * tMI[k] + ip[k] * tII[k];
    }
    xE = dchain_forward(om, mc, dc);
    mc[M + 1] = ic[M + 1] = dc[M + 1] = 0.0f;
    xN = xN * om.xf[XN][LOOP];
    xC = (xC * om.xf[XC][LOOP]) + (xE * om.xf[XE][MOVE]);
    xJ = (xJ * om.xf[XJ][LOOP]) + (xE * om.xf[XE][LOOP]);
    xB = (xJ * om.xf[XJ][MOVE]) + (xN * om.xf[XN][MOVE]);
    if (xE > 1.0e4) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      const float inv = 1.0 / xE;
      for (int q = 1; q <= M; ++q) { mc[q] *= inv; dc[q] *= inv; ic[q] *= inv; }
      ox.X(r, xS_) = xE;
      ox.totscale += std::log((double) xE);
      xE = 1.0;
    } else ox.X(r, xS_) = 1.0f;
    ox.X(r, xE_) = xE; ox.X(r, xN_) = xN; ox.X(r, xJ_) = xJ; ox.X(r, xB_) = xB; ox.X(r, xC_) = xC;
  }
  if (std::isnan(xC) || (L > 0 && xC == 0.0f) || std::isinf(xC)) { if (ret_sc) *ret_sc = INFINITY; return P7X_ERANGE; }
  if (ret_sc) *ret_sc = ox.totscale + std::log((double) (xC * om.xf[XC][MOVE]));
  return P7X_OK;
}

// ---------------------------------------------------------------- p7_Backward (full matrix)
// D(i,k) = base(k) + D(i,k+1) tDD(k), k = M..1 (D(i,M+1) = 0); on entry dc[k] = base(k).  The envelope kernel's d_chain:
// every lane folds its nodes from the last to the first, the scan composes the lanes from 63 downwards, and each lane
// then recomputes its chain from the carry it was handed.
static void dchain_backward(const Model &om, float *__restrict dc)
{
  const int M = om.M, C = om.C;
  const float *__restrict tDD = om.tf(7);
  float sa[64], sp[64];
  for (int z = 0; z < 64; ++z) {
    const int k0 = z * C + 1, k1 = std::min(M, k0 + C - 1);
    float A = 0.0f;
    for (int k = k1; k >= k0; --k) A = dc[k] + A * tDD[k];
    sa[z] = A; sp[z] = om.ddprod[z];
  }
  lanes_scan_down(sa, sp);
  for (int z = 0; z < 64; ++z) {
    const int k0 = z * C + 1, k1 = std::min(M, k0 + C - 1);
    float w = (z < 63) ? sa[z + 1] : 0.0f;
    for (int k = k1; k >= k0; --k) { dc[k] = dc[k] + w * tDD[k]; w = dc[k]; }
  }
}
// sum_k v(k) w(k) as the kernels form it: a lane's nodes in order, then the reduction tree (v = nullptr: sum of w)
static float lanes_dot(const Model &om, const float *__restrict v, const float *__restrict w)
{
  const int M = om.M, C = om.C;
  float es[64];
  for (int z = 0; z < 64; ++z) {
    const int k0 = z * C + 1, k1 = std::min(M, k0 + C - 1);
    float e = 0.0f;
    for (int k = k0; k <= k1; ++k) e = e + v[k] * w[k];
    es[z] = e;
  }
  return lanes_sum(es);
}

P7X_MULTIVERSION int backward_full_lanes(const Model &om, const uint8_t *dsq, int L, const Matrix &fwd, Matrix &bck, float *ret_sc)
{
  const int M = om.M;
  bck.resize(M, L);
  const float *__restrict bm = om.tf(0), *__restrict tMM = om.tf(1), *__restrict tIM = om.tf(2), *__restrict tDM = om.tf(3),
              *__restrict tMD = om.tf(4), *__restrict tMI = om.tf(5), *__restrict tII = om.tf(6);
  bck.own_scales = false;
  float xJ = 0.f, xB = 0.f, xN = 0.f;
  float xC = om.xf[XC][MOVE];
  float xE = xC * om.xf[XE][MOVE];
  float *__restrict me = bck.scratch.data();          // [M+3] work row (no allocation inside the multiversioned body)
  for (int k = 0; k <= M + 2; ++k) me[k] = 0.0f;
  {
    float *__restrict mc = bck.M_(L), *__restrict ic = bck.I_(L), *__restrict dc = bck.D_(L);
    mc[M + 1] = ic[M + 1] = dc[M + 1] = 0.0f;
    for (int k = 1; k <= M; ++k) { dc[k] = xE; ic[k] = 0.0f; }
    dchain_backward(om, dc);
    for (int k = 1; k <= M; ++k) mc[k] = xE + dc[k + 1] * tMD[k];
    mc[0] = ic[0] = dc[0] = 0.0f;
    float sc = fwd.X(L, xS_);
    if (sc > 1.0f) {
      xE = xE / sc; xN = xN / sc; xC = xC / sc; xJ = xJ / sc; xB = xB / sc;
      const float inv = 1.0 / sc;
      for (int k = 1; k <= M; ++k) { mc[k] *= inv; dc[k] *= inv; ic[k] *= inv; }
    }
    bck.X(L, xS_) = sc;
    bck.totscale = std::log((double) sc);
    bck.X(L, xE_) = xE; bck.X(L, xN_) = xN; bck.X(L, xJ_) = xJ; bck.X(L, xB_) = xB; bck.X(L, xC_) = xC;
  }
  for (int r = L - 1; r >= 1; --r) {
    const float *__restrict rf = om.rf(dsq[r + 1]);
    const float *__restrict mn = bck.M_(r + 1), *__restrict in = bck.I_(r + 1);
    float *__restrict mc = bck.M_(r), *__restrict ic = bck.I_(r), *__restrict dc = bck.D_(r);
    for (int k = 1; k <= M; ++k) me[k] = mn[k] * rf[k];                      // M(i+1,k) e(x_{i+1},k)
    me[M + 1] = 0.0f;
    xB = lanes_dot(om, me, bm);
    xC = xC * om.xf[XC][LOOP];
    xJ = (xB * om.xf[XJ][MOVE]) + (xJ * om.xf[XJ][LOOP]);
    xN = (xB * om.xf[XN][MOVE]) + (xN * om.xf[XN][LOOP]);
    xE = (xC * om.xf[XE][MOVE]) + (xJ * om.xf[XE][LOOP]);
    mc[M + 1] = ic[M + 1] = dc[M + 1] = 0.0f;
    // entering-transitions of node k+1 are stored at index k+1; node M+1 does not exist (me[M+1] = 0)
    for (int k = 1; k < M; ++k) {
      const float mek = me[k + 1];
      ic[k] = in[k] * tII[k] + mek * tIM[k + 1];
      dc[k] = mek * tDM[k + 1] + xE;
      mc[k] = (in[k] * tMI[k] + mek * tMM[k + 1]) + xE;
    }
    ic[M] = in[M] * tII[M]; dc[M] = xE; mc[M] = in[M] * tMI[M] + xE;
    dchain_backward(om, dc);
    for (int k = 1; k <= M; ++k) mc[k] += dc[k + 1] * tMD[k];
    mc[0] = ic[0] = dc[0] = 0.0f;
    if (xB > 1.0e16) bck.own_scales = true;
    float sc = bck.own_scales ? ((xB > 1.0e4) ? xB : 1.0f) : fwd.X(r, xS_);
    bck.X(r, xS_) = sc;
    if (sc > 1.0f) {
      xE /= sc; xN /= sc; xJ /= sc; xB /= sc; xC /= sc;
      const float inv = 1.0 / sc;
      for (int k = 1; k <= M; ++k) { mc[k] *= inv; dc[k] *= inv; ic[k] *= inv; }
      bck.totscale += std::log((double) sc);
    }
    bck.X(r, xE_) = xE; bck.X(r, xN_) = xN; bck.X(r, xJ_) = xJ; bck.X(r, xB_) = xB; bck.X(r, xC_) = xC;
  }
  {
    const float *__restrict rf = om.rf(dsq[1]);
    const float *__restrict mn = bck.M_(1);
    for (int k = 1; k <= M; ++k) me[k] = mn[k] * rf[k];
    xB = lanes_dot(om, me, bm);
    xN = (xB * om.xf[XN][MOVE]) + (xN * om.xf[XN][LOOP]);
    bck.X(0, xB_) = xB; bck.X(0, xC_) = 0.0f; bck.X(0, xJ_) = 0.0f; bck.X(0, xN_) = xN; bck.X(0, xE_) = 0.0f; bck.X(0, xS_) = 1.0f;
    float *mc = bck.M_(0), *ic = bck.I_(0), *dc = bck.D_(0);
    for (int k = 0; k <= M + 1; ++k) mc[k] = ic[k] = dc[k] = 0.0f;
  }
  if (std::isnan(xN) || (L > 0 && xN == 0.0f) || std::isinf(xN)) { if (ret_sc) *ret_sc = INFINITY; return P7X_ERANGE; }
  if (ret_sc) *ret_sc = bck.totscale + std::log((double) xN);
  return P7X_OK;
}


// ---------------------------------------------------------------- p7_Forward / p7_Backward in upstream's order
// impl_sse/fwdback.c forward_engine / backward_engine with do_full = TRUE, restated on four-float vectors: vector q of a row
// holds nodes q+1, q+1+Q, q+1+2Q, q+1+3Q (padding nodes carry zero transitions and emissions, as p7_oprofile_Convert pads
// them).  Every multiplication and addition of the vector code is performed on the same operands in the same order; the
// rows go to the un-striped Matrix afterwards.  forward_parser_striped() (p7x_longtarget.inc.hpp) is the do_full = FALSE
// sibling.
typedef float V4 __attribute__((vector_size(16), aligned(4)));      // four stripes side by side (the compiler's own vector type: one SIMD operation per vector operation)
static inline V4 v4_set(float a) { return V4{ a, a, a, a }; }
static inline V4 v4_add(const V4 &a, const V4 &b) { return a + b; }
static inline V4 v4_mul(const V4 &a, const V4 &b) { return a * b; }
static inline V4 v4_shr(const V4 &a) { const V4 z = { 0.0f, 0.0f, 0.0f, 0.0f }; return __builtin_shufflevector(a, z, 4, 0, 1, 2); }   // esl_sse_rightshift_ps(a, 0)
static inline V4 v4_shl(const V4 &a) { const V4 z = { 0.0f, 0.0f, 0.0f, 0.0f }; return __builtin_shufflevector(a, z, 1, 2, 3, 4); }   // esl_sse_leftshift_ps(a, 0)
static inline float v4_hsum(const V4 &a) { return (a[0] + a[1]) + (a[2] + a[3]); }            // esl_sse_hsum_ps
struct StripedRow { std::vector<V4> m, d, i; void resize(int Q) { m.resize((size_t) Q); d.resize((size_t) Q); i.resize((size_t) Q); } };
static inline void unstripe(const StripedRow &r, int Q, int M, float *mc, float *ic, float *dc)
{
  for (int q = 0; q < Q; ++q)
    for (int z = 0; z < 4; ++z) { const int k = q + 1 + z * Q; if (k <= M) { mc[k] = r.m[q][z]; ic[k] = r.i[q][z]; dc[k] = r.d[q][z]; } }
  mc[0] = ic[0] = dc[0] = 0.0f; mc[M + 1] = ic[M + 1] = dc[M + 1] = 0.0f;
}
static inline void stripe_emissions(const float *rf, int Q, int M, std::vector<V4> &rv)
{
  for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) { const int k = q + 1 + z * Q; rv[(size_t) q][z] = k <= M ? rf[k] : 0.0f; }
}

struct StripedScratch { StripedRow a, b; std::vector<V4> rv; };     // per thread, owned by the dispatchers below (no thread_local inside a cloned function)
P7X_MULTIVERSION int forward_full_upstream(const Model &om, const uint8_t *dsq, int L, Matrix &ox, float *ret_sc, StripedScratch &ss)
{
  const int M = om.M, Q = om.Q;
  ox.resize(M, L);
  const V4 *tBM = reinterpret_cast<const V4 *>(om.sT(0)), *tMM = reinterpret_cast<const V4 *>(om.sT(1)), *tIM = reinterpret_cast<const V4 *>(om.sT(2)),
           *tDM = reinterpret_cast<const V4 *>(om.sT(3)), *tMD = reinterpret_cast<const V4 *>(om.sT(4)), *tMI = reinterpret_cast<const V4 *>(om.sT(5)),
           *tII = reinterpret_cast<const V4 *>(om.sT(6)), *tDD = reinterpret_cast<const V4 *>(om.sT(7));
  StripedRow &row = ss.a;
  row.resize(Q);
  const V4 zero = v4_set(0.0f);
  for (int q = 0; q < Q; ++q) row.m[q] = row.d[q] = row.i[q] = zero;
  { float *m0 = ox.M_(0), *i0 = ox.I_(0), *d0 = ox.D_(0); for (int k = 0; k <= M + 1; ++k) m0[k] = i0[k] = d0[k] = 0.0f; }
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = om.xf[XN][MOVE], xC = 0.f;
  ox.X(0, xE_) = xE; ox.X(0, xN_) = xN; ox.X(0, xJ_) = xJ; ox.X(0, xB_) = xB; ox.X(0, xC_) = xC; ox.X(0, xS_) = 1.0f;
  ox.totscale = 0.0f; ox.own_scales = true;
  for (int r = 1; r <= L; ++r) {
    const V4 *rv = reinterpret_cast<const V4 *>(om.sR(dsq[r]));
    V4 dcv = zero, xEv = zero;
    const V4 xBv = v4_set(xB);
    V4 mpv = v4_shr(row.m[Q - 1]), dpv = v4_shr(row.d[Q - 1]), ipv = v4_shr(row.i[Q - 1]);
    for (int q = 0; q < Q; ++q) {
      V4 sv = v4_mul(xBv, tBM[q]);
      sv = v4_add(sv, v4_mul(mpv, tMM[q]));
      sv = v4_add(sv, v4_mul(ipv, tIM[q]));
      sv = v4_add(sv, v4_mul(dpv, tDM[q]));
      sv = v4_mul(sv, rv[(size_t) q]);
      xEv = v4_add(xEv, sv);
      mpv = row.m[q]; dpv = row.d[q]; ipv = row.i[q];
      row.m[q] = sv;
      row.d[q] = dcv;
      dcv = v4_mul(sv, tMD[q]);
      sv = v4_mul(mpv, tMI[q]);
      row.i[q] = v4_add(sv, v4_mul(ipv, tII[q]));
    }
    dcv = v4_shr(dcv);
    row.d[0] = zero;
    for (int q = 0; q < Q; ++q) { row.d[q] = v4_add(dcv, row.d[q]); dcv = v4_mul(row.d[q], tDD[q]); }
    if (M < 100) {
      for (int j = 1; j < 4; ++j) {
        dcv = v4_shr(dcv);
        for (int q = 0; q < Q; ++q) { row.d[q] = v4_add(dcv, row.d[q]); dcv = v4_mul(dcv, tDD[q]); }
      }
    } else {
      for (int j = 1; j < 4; ++j) {
        bool grew = false;
        dcv = v4_shr(dcv);
        for (int q = 0; q < Q; ++q) {
          const V4 sv = v4_add(dcv, row.d[q]);
          for (int z = 0; z < 4; ++z) grew |= sv[z] > row.d[q][z];
          row.d[q] = sv;
          dcv = v4_mul(dcv, tDD[q]);
        }
        if (!grew) break;
      }
    }
    for (int q = 0; q < Q; ++q) xEv = v4_add(row.d[q], xEv);
    xE = v4_hsum(xEv);
    xN = xN * om.xf[XN][LOOP];
    xC = (xC * om.xf[XC][LOOP]) + (xE * om.xf[XE][MOVE]);
    xJ = (xJ * om.xf[XJ][LOOP]) + (xE * om.xf[XE][LOOP]);
    xB = (xJ * om.xf[XJ][MOVE]) + (xN * om.xf[XN][MOVE]);
    if (xE > 1.0e4) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      const V4 inv = v4_set((float) (1.0 / xE));
      for (int q = 0; q < Q; ++q) { row.m[q] = v4_mul(row.m[q], inv); row.d[q] = v4_mul(row.d[q], inv); row.i[q] = v4_mul(row.i[q], inv); }
      ox.X(r, xS_) = xE;
      ox.totscale += std::log((double) xE);
      xE = 1.0;
    } else ox.X(r, xS_) = 1.0f;
    ox.X(r, xE_) = xE; ox.X(r, xN_) = xN; ox.X(r, xJ_) = xJ; ox.X(r, xB_) = xB; ox.X(r, xC_) = xC;
    unstripe(row, Q, M, ox.M_(r), ox.I_(r), ox.D_(r));
  }
  if (std::isnan(xC) || (L > 0 && xC == 0.0f) || std::isinf(xC)) { if (ret_sc) *ret_sc = INFINITY; return P7X_ERANGE; }
  if (ret_sc) *ret_sc = ox.totscale + std::log((double) (xC * om.xf[XC][MOVE]));
  return P7X_OK;
}

P7X_MULTIVERSION int backward_full_upstream(const Model &om, const uint8_t *dsq, int L, const Matrix &fwd, Matrix &bck, float *ret_sc, StripedScratch &ss)
{
  const int M = om.M, Q = om.Q;
  bck.resize(M, L);
  const V4 *tBM = reinterpret_cast<const V4 *>(om.sT(0)), *tMM = reinterpret_cast<const V4 *>(om.sT(1)), *tIM = reinterpret_cast<const V4 *>(om.sT(2)),
           *tDM = reinterpret_cast<const V4 *>(om.sT(3)), *tMD = reinterpret_cast<const V4 *>(om.sT(4)), *tMI = reinterpret_cast<const V4 *>(om.sT(5)),
           *tII = reinterpret_cast<const V4 *>(om.sT(6)), *tDD = reinterpret_cast<const V4 *>(om.sT(7));
  StripedRow &rowa = ss.a, &rowb = ss.b;
  rowa.resize(Q); rowb.resize(Q);
  StripedRow *cur = &rowa, *nxt = &rowb;          // row i being built, row i + 1
  const V4 zero = v4_set(0.0f);
  bck.own_scales = false;
  float xJ = 0.f, xB = 0.f, xN = 0.f;
  float xC = om.xf[XC][MOVE];
  float xE = xC * om.xf[XE][MOVE];
  // the D -> D paths of a row: one sweep down every stripe from what the stripe after it holds so far, then three sweeps that
  // hand the remainder from stripe to stripe; then the M -> D paths.  first: the operand the first sweep starts from.
  auto close_row = [&](StripedRow &r, const V4 &first, const V4 &xEv, bool add_e) {
    V4 dpv = v4_shl(first), dcv = zero;
    for (int q = Q - 1; q >= 0; --q) {
      dcv = v4_mul(dpv, tDD[q]);
      if (add_e) { r.d[q] = v4_add(r.d[q], v4_add(dcv, xEv)); r.m[q] = v4_add(r.m[q], xEv); }
      else       r.d[q] = v4_add(r.d[q], dcv);
      dpv = r.d[q];
    }
    for (int j = 1; j < 4; ++j) {
      dcv = v4_shl(dcv);
      for (int q = Q - 1; q >= 0; --q) { dcv = v4_mul(dcv, tDD[q]); r.d[q] = v4_add(r.d[q], dcv); }
    }
    dcv = v4_shl(r.d[0]);
    for (int q = Q - 1; q >= 0; --q) { r.m[q] = v4_add(r.m[q], v4_mul(dcv, tMD[q])); dcv = r.d[q]; }
  };
  auto rescale = [&](StripedRow &r, float sc) {
    const V4 inv = v4_set((float) (1.0 / sc));
    for (int q = 0; q < Q; ++q) { r.m[q] = v4_mul(r.m[q], inv); r.d[q] = v4_mul(r.d[q], inv); r.i[q] = v4_mul(r.i[q], inv); }
  };
  {
    const V4 xEv = v4_set(xE);
    for (int q = 0; q < Q; ++q) { cur->m[q] = cur->d[q] = xEv; cur->i[q] = zero; }
    close_row(*cur, cur->d[Q - 1], xEv, false);
    const float sc = fwd.X(L, xS_);
    if (sc > 1.0f) { xE = xE / sc; xN = xN / sc; xC = xC / sc; xJ = xJ / sc; xB = xB / sc; rescale(*cur, sc); }
    bck.X(L, xS_) = sc;
    bck.totscale = std::log((double) sc);
    bck.X(L, xE_) = xE; bck.X(L, xN_) = xN; bck.X(L, xJ_) = xJ; bck.X(L, xB_) = xB; bck.X(L, xC_) = xC;
    unstripe(*cur, Q, M, bck.M_(L), bck.I_(L), bck.D_(L));
  }
  for (int r = L - 1; r >= 1; --r) {
    std::swap(cur, nxt);
    const V4 *rv = reinterpret_cast<const V4 *>(om.sR(dsq[r + 1]));
    V4 tmmv = v4_shl(tMM[0]), timv = v4_shl(tIM[0]), tdmv = v4_shl(tDM[0]);      // the transitions INTO the node after the stripe's last
    V4 mpv = v4_shl(v4_mul(nxt->m[0], rv[0]));                                   // M(i+1, k+1) e(x_{i+1}, k+1)
    V4 xBv = zero;
    for (int q = Q - 1; q >= 0; --q) {
      const V4 ipv = nxt->i[q];
      cur->i[q] = v4_add(v4_mul(ipv, tII[q]), v4_mul(mpv, timv));
      cur->d[q] = v4_mul(mpv, tdmv);
      const V4 mcv = v4_add(v4_mul(ipv, tMI[q]), v4_mul(mpv, tmmv));
      mpv = v4_mul(nxt->m[q], rv[(size_t) q]);
      cur->m[q] = mcv;
      tdmv = tDM[q]; timv = tIM[q]; tmmv = tMM[q];
      xBv = v4_add(xBv, v4_mul(mpv, tBM[q]));
    }
    xB = v4_hsum(xBv);
    xC = xC * om.xf[XC][LOOP];
    xJ = (xB * om.xf[XJ][MOVE]) + (xJ * om.xf[XJ][LOOP]);
    xN = (xB * om.xf[XN][MOVE]) + (xN * om.xf[XN][LOOP]);
    xE = (xC * om.xf[XE][MOVE]) + (xJ * om.xf[XE][LOOP]);
    const V4 xEv = v4_set(xE);
    close_row(*cur, v4_add(cur->d[0], xEv), xEv, true);
    if (xB > 1.0e16) bck.own_scales = true;
    const float sc = bck.own_scales ? ((xB > 1.0e4) ? xB : 1.0f) : fwd.X(r, xS_);
    bck.X(r, xS_) = sc;
    if (sc > 1.0f) {
      xE /= sc; xN /= sc; xJ /= sc; xB /= sc; xC /= sc;
      rescale(*cur, sc);
      bck.totscale += std::log((double) sc);
    }
    bck.X(r, xE_) = xE; bck.X(r, xN_) = xN; bck.X(r, xJ_) = xJ; bck.X(r, xB_) = xB; bck.X(r, xC_) = xC;
    unstripe(*cur, Q, M, bck.M_(r), bck.I_(r), bck.D_(r));
  }
  {
    const V4 *rv = reinterpret_cast<const V4 *>(om.sR(dsq[1]));
    V4 xBv = zero;
    for (int q = Q - 1; q >= 0; --q) xBv = v4_add(xBv, v4_mul(v4_mul(cur->m[q], rv[(size_t) q]), tBM[q]));
    xB = v4_hsum(xBv);
    xN = (xB * om.xf[XN][MOVE]) + (xN * om.xf[XN][LOOP]);
    bck.X(0, xB_) = xB; bck.X(0, xC_) = 0.0f; bck.X(0, xJ_) = 0.0f; bck.X(0, xN_) = xN; bck.X(0, xE_) = 0.0f; bck.X(0, xS_) = 1.0f;
    float *mc = bck.M_(0), *ic = bck.I_(0), *dc = bck.D_(0);
    for (int k = 0; k <= M + 1; ++k) mc[k] = ic[k] = dc[k] = 0.0f;
  }
  if (std::isnan(xN) || (L > 0 && xN == 0.0f) || std::isinf(xN)) { if (ret_sc) *ret_sc = INFINITY; return P7X_ERANGE; }
  if (ret_sc) *ret_sc = bck.totscale + std::log((double) xN);
  return P7X_OK;
}

static StripedScratch &striped_scratch() { thread_local StripedScratch *ss = new StripedScratch(); return *ss; }     // (leaked with its thread: a few KB)
static inline int forward_full(const Model &om, const uint8_t *dsq, int L, Matrix &ox, float *ret_sc)
{ return om.upstream ? forward_full_upstream(om, dsq, L, ox, ret_sc, striped_scratch()) : forward_full_lanes(om, dsq, L, ox, ret_sc); }
static inline int backward_full(const Model &om, const uint8_t *dsq, int L, const Matrix &fwd, Matrix &bck, float *ret_sc)
{ return om.upstream ? backward_full_upstream(om, dsq, L, fwd, bck, ret_sc, striped_scratch()) : backward_full_lanes(om, dsq, L, fwd, bck, ret_sc); }

// ---------------------------------------------------------------- p7_Decoding: posteriors into <bck> (in place)
// NOTE: only the M and I posteriors are formed.  bck.D_ keeps Backward's delete values afterwards (upstream zeroes them): a
// future reader of the delete "posteriors" must not take them from here.
P7X_MULTIVERSION int decoding(const Model &om, const Matrix &fwd, Matrix &bck)
{
  const int M = om.M, L = fwd.L;
  float scaleproduct = 1.0 / bck.X(0, xN_);
  // row 0 is zeroed *after* reading xN(0); upstream writes into a third matrix, we overwrite <bck> row by row,
  // which is safe because row i of the posterior only needs row i of both matrices and specials of row i-1 (fwd).
  float *bN = bck.scratch.data() + (M + 4), *bJ = bN + (L + 1), *bC = bJ + (L + 1), *bS = bC + (L + 1);
  for (int r = 0; r <= L; ++r) { bN[r] = bck.X(r, xN_); bJ[r] = bck.X(r, xJ_); bC[r] = bck.X(r, xC_); bS[r] = bck.X(r, xS_); }
  {
    float *mc = bck.M_(0), *ic = bck.I_(0), *dc = bck.D_(0);
    for (int k = 0; k <= M + 1; ++k) mc[k] = ic[k] = dc[k] = 0.0f;
    bck.X(0, xE_) = bck.X(0, xN_) = bck.X(0, xJ_) = bck.X(0, xC_) = bck.X(0, xB_) = 0.0f;
  }
  for (int r = 1; r <= L; ++r) {
    const float totr = scaleproduct * fwd.X(r, xS_);
    const float *__restrict fm = fwd.M_(r), *__restrict fi = fwd.I_(r);
    float *__restrict mc = bck.M_(r), *__restrict ic = bck.I_(r), *__restrict dc = bck.D_(r);
    (void) dc;                 // upstream zeroes the delete posteriors; nothing downstream reads them (optimal accuracy, null2 and the
                               // trace take M and I), so the row is left as Backward wrote it and a store stream is saved
    for (int k = 1; k <= M; ++k) {
      mc[k] = (fm[k] * mc[k]) * totr;
      ic[k] = (fi[k] * ic[k]) * totr;
    }
    bck.X(r, xE_) = 0.0f;
    bck.X(r, xN_) = fwd.X(r - 1, xN_) * bN[r] * om.xf[XN][LOOP] * scaleproduct;
    bck.X(r, xJ_) = fwd.X(r - 1, xJ_) * bJ[r] * om.xf[XJ][LOOP] * scaleproduct;
    bck.X(r, xB_) = 0.0f;
    bck.X(r, xC_) = fwd.X(r - 1, xC_) * bC[r] * om.xf[XC][LOOP] * scaleproduct;
    if (bck.own_scales) scaleproduct *= fwd.X(r, xS_) / bS[r];
  }
  return std::isinf(scaleproduct) ? P7X_ERANGE : P7X_OK;
}

// esl_abc_FAvgScVec + gap-like symbols
void finish_null2(const Profile &p, float *null2)
{
  const Alphabet &abc = Alphabet::get(p.abc_type);
  for (int x = p.K + 1; x <= p.Kp - 3; ++x) {
    float result = 0.0f; int n = 0;
    for (int y = 0; y < p.K; ++y) if (abc.degen[x][y]) { result += null2[y]; ++n; }
    null2[x] = result / (float) n;
  }
  null2[p.K] = 1.0f; null2[p.Kp - 2] = 1.0f; null2[p.Kp - 1] = 1.0f;
}

// ---------------------------------------------------------------- p7_Null2_ByExpectation (pp = posterior matrix)
P7X_MULTIVERSION void null2_by_expectation(const Model &om, Matrix &pp, float *null2)
{
  const int M = om.M, Ld = pp.L;
  float *m0 = pp.M_(0), *i0 = pp.I_(0);
  std::memcpy(m0, pp.M_(1), sizeof(float) * (M + 2));
  std::memcpy(i0, pp.I_(1), sizeof(float) * (M + 2));
  float eN = pp.X(1, xN_), eC = pp.X(1, xC_), eJ = pp.X(1, xJ_);
  for (int r = 2; r <= Ld; ++r) {
    const float *mr = pp.M_(r), *ir = pp.I_(r);
    for (int k = 1; k <= M; ++k) { m0[k] = mr[k] + m0[k]; i0[k] = ir[k] + i0[k]; }
    eN += pp.X(r, xN_); eC += pp.X(r, xC_); eJ += pp.X(r, xJ_);
  }
  const float norm = 1.0 / (float) Ld;
  for (int k = 1; k <= M; ++k) { m0[k] *= norm; i0[k] *= norm; }
  eN *= norm; eC *= norm; eJ *= norm;
  const float xfactor = eN + eC + eJ;
  const int C = om.C;
  if (om.upstream) {                                    // impl_sse/null2.c: four stripes, a stripe's nodes in order (match term, insert term), then hsum
    const int Q = om.Q;
    for (int x = 0; x < om.p->K; ++x) {
      const float *rf = om.rf(x);
      float sv[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      for (int q = 0; q < Q; ++q)
        for (int z = 0; z < 4; ++z) { const int k = q + 1 + z * Q; if (k <= M) { sv[z] = sv[z] + m0[k] * rf[k]; sv[z] = sv[z] + i0[k]; } }
      null2[x] = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + xfactor;
    }
    finish_null2(*om.p, null2);
    return;
  }
  for (int x = 0; x < om.p->K; ++x) {
    const float *rf = om.rf(x);
    float es[64];                                       // the envelope kernel's order: a lane's nodes, match then insert, then the tree
    for (int z = 0; z < 64; ++z) {
      const int k0 = z * C + 1, k1 = std::min(M, k0 + C - 1);
      float e = 0.0f;
      for (int k = k0; k <= k1; ++k) { e = e + m0[k] * rf[k]; e = e + i0[k]; }
      es[z] = e;
    }
    null2[x] = lanes_sum(es) + xfactor;
  }
  finish_null2(*om.p, null2);
}

// ---------------------------------------------------------------- traces
struct Trace {
  std::vector<int8_t> st; std::vector<int> k, i; std::vector<float> pp;
  int ndom = 0;
  std::vector<int> tfrom, tto, sqfrom, sqto, hmmfrom, hmmto;
  void clear() { st.clear(); k.clear(); i.clear(); pp.clear(); ndom = 0; tfrom.clear(); tto.clear(); sqfrom.clear(); sqto.clear(); hmmfrom.clear(); hmmto.clear(); }
  void append(int s, int kk, int ii, float p)
  { // p7_trace_AppendWithPP
    int iv = 0, kv = 0; float pv = 0.0f;
    switch (s) {
      case sN: case sC: case sJ:
        if (!st.empty() && st.back() == s) { iv = ii; pv = p; }
        break;
      case sD: kv = kk; break;
      case sM: case sI: iv = ii; kv = kk; pv = p; break;
      default: break;
    }
    st.push_back((int8_t) s); k.push_back(kv); i.push_back(iv); pp.push_back(pv);
  }
  void reverse()
  { // p7_trace_Reverse: N,C,J emit on transition, so their i/pp move one step when the order flips
    const int N = (int) st.size();
    for (int z = 0; z < N; ++z)
      if ((st[z] == sN || st[z] == sC || st[z] == sJ) && z + 1 < N && st[z] == st[z + 1]) {
        if (i[z] == 0 && i[z + 1] > 0) { i[z] = i[z + 1]; i[z + 1] = 0; pp[z] = pp[z + 1]; pp[z + 1] = 0.0f; }
      }
    std::reverse(st.begin(), st.end()); std::reverse(k.begin(), k.end());
    std::reverse(i.begin(), i.end()); std::reverse(pp.begin(), pp.end());
  }
  void index()
  { // p7_trace_Index
    ndom = 0; tfrom.clear(); tto.clear(); sqfrom.clear(); sqto.clear(); hmmfrom.clear(); hmmto.clear();
    for (int z = 0; z < (int) st.size(); ++z)
      switch (st[z]) {
        case sB: tfrom.push_back(z); tto.push_back(0); sqfrom.push_back(0); sqto.push_back(0); hmmfrom.push_back(0); hmmto.push_back(0); break;
        case sM:
          if (sqfrom[ndom] == 0) sqfrom[ndom] = i[z];
          if (hmmfrom[ndom] == 0) hmmfrom[ndom] = k[z];
          sqto[ndom] = i[z]; hmmto[ndom] = k[z];
          break;
        case sE: tto[ndom] = z; ndom++; break;
        default: break;
      }
  }
};

// ---------------------------------------------------------------- p7_OptimalAccuracy + p7_OATrace
inline float gate(float t, float v) { return t > 0.0f ? v : 0.0f; }   // upstream: and(cmpgt(t,0), v)
inline float vmax(float a, float b) { return a > b ? a : b; }

P7X_MULTIVERSION void optimal_accuracy(const Model &om, const Matrix &pp, Matrix &ox, float *ret_e)
{
  const int M = om.M, L = pp.L;
  ox.resize(M, L);
  const float *bm = om.tf(0), *tMM = om.tf(1), *tIM = om.tf(2), *tDM = om.tf(3), *tMD = om.tf(4), *tMI = om.tf(5), *tII = om.tf(6), *tDD = om.tf(7);
  {
    float *mc = ox.M_(0), *ic = ox.I_(0), *dc = ox.D_(0);
    for (int k = 0; k <= M + 1; ++k) mc[k] = ic[k] = dc[k] = -INFINITY;
  }
  ox.X(0, xE_) = -INFINITY; ox.X(0, xN_) = 0.f; ox.X(0, xJ_) = -INFINITY; ox.X(0, xB_) = 0.f; ox.X(0, xC_) = -INFINITY;
  for (int r = 1; r <= L; ++r) {
    const float *__restrict mp = ox.M_(r - 1), *__restrict ip = ox.I_(r - 1), *__restrict dp = ox.D_(r - 1);
    const float *__restrict pm = pp.M_(r), *__restrict pi = pp.I_(r);
    float *__restrict mc = ox.M_(r), *__restrict ic = ox.I_(r), *__restrict dc = ox.D_(r);
    const float xB = ox.X(r - 1, xB_);
    mc[0] = ic[0] = dc[0] = -INFINITY;
    for (int k = 1; k <= M; ++k) {
      float sv = gate(bm[k], xB);
      sv = vmax(sv, gate(tMM[k], mp[k - 1]));
      sv = vmax(sv, gate(tIM[k], ip[k - 1]));
      sv = vmax(sv, gate(tDM[k], dp[k - 1]));
      mc[k] = sv + pm[k];
      float iv = gate(tMI[k], mp[k]);
      iv = vmax(iv, gate(tII[k], ip[k]));
      ic[k] = iv + pi[k];
    }
    // D(i,k) = max( gate(tMD(k-1), M(i,k-1)), gate(tDD(k-1), D(i,k-1)) ), D(i,1) = -inf.  Maxima are exact: any order of
    // evaluation (the device's gated max-scan included) gives these values.
    dc[1] = -INFINITY;
    for (int k = 2; k <= M; ++k) dc[k] = vmax(gate(tMD[k - 1], mc[k - 1]), (tDD[k - 1] > 0.0f) ? dc[k - 1] : 0.0f);
    float xEmax = -INFINITY;
    {
      float a8[8]; for (int z = 0; z < 8; ++z) a8[z] = -INFINITY;
      int k = 1;
      for (; k + 7 <= M; k += 8) for (int z = 0; z < 8; ++z) a8[z] = vmax(a8[z], vmax(mc[k + z], dc[k + z]));
      for (; k <= M; ++k) a8[0] = vmax(a8[0], vmax(mc[k], dc[k]));
      for (int z = 0; z < 8; ++z) xEmax = vmax(xEmax, a8[z]);
    }
    mc[M + 1] = ic[M + 1] = dc[M + 1] = -INFINITY;
    ox.X(r, xE_) = xEmax;
    float t1, t2;
    t1 = (om.xf[XJ][LOOP] == 0.0f) ? 0.0f : ox.X(r - 1, xJ_) + pp.X(r, xJ_);
    t2 = (om.xf[XE][LOOP] == 0.0f) ? 0.0f : ox.X(r, xE_);
    ox.X(r, xJ_) = std::fmax(t1, t2);
    t1 = (om.xf[XC][LOOP] == 0.0f) ? 0.0f : ox.X(r - 1, xC_) + pp.X(r, xC_);
    t2 = (om.xf[XE][MOVE] == 0.0f) ? 0.0f : ox.X(r, xE_);
    ox.X(r, xC_) = std::fmax(t1, t2);
    ox.X(r, xN_) = (om.xf[XN][LOOP] == 0.0f) ? 0.0f : ox.X(r - 1, xN_) + pp.X(r, xN_);
    t1 = (om.xf[XN][MOVE] == 0.0f) ? 0.0f : ox.X(r, xN_);
    t2 = (om.xf[XJ][MOVE] == 0.0f) ? 0.0f : ox.X(r, xJ_);
    ox.X(r, xB_) = std::fmax(t1, t2);
  }
  *ret_e = ox.X(L, xC_);
}

int oa_trace(const Model &om, const Matrix &pp, const Matrix &ox, Trace &tr)
{
  const int M = om.M, Q = om.p->Q4();
  const float *bm = om.tf(0), *tMM = om.tf(1), *tIM = om.tf(2), *tDM = om.tf(3), *tMD = om.tf(4), *tMI = om.tf(5), *tII = om.tf(6), *tDD = om.tf(7);
  int i = ox.L, k = 0;
  tr.clear();
  tr.append(sT, k, i, 0.0f);
  tr.append(sC, k, i, 0.0f);
  int s0 = sC;
  auto postprob = [&](int scur, int sprv, int kk, int ii) -> float {
    switch (scur) {
      case sM: return pp.M_(ii)[kk];
      case sI: return pp.I_(ii)[kk];
      case sN: if (sprv == scur) return pp.X(ii, xN_); return 0.0f;
      case sC: if (sprv == scur) return pp.X(ii, xC_); return 0.0f;
      case sJ: if (sprv == scur) return pp.X(ii, xJ_); return 0.0f;
      default: return 0.0f;
    }
  };
  while (s0 != sS) {
    int s1 = -1;
    switch (s0) {
      case sM: {
        float path[4];
        path[0] = (tMM[k] == 0.0f) ? -INFINITY : ox.M_(i - 1)[k - 1];
        path[1] = (tIM[k] == 0.0f) ? -INFINITY : ox.I_(i - 1)[k - 1];
        path[2] = (tDM[k] == 0.0f) ? -INFINITY : ox.D_(i - 1)[k - 1];
        path[3] = (bm[k]  == 0.0f) ? -INFINITY : ox.X(i - 1, xB_);
        static const int state[4] = { sM, sI, sD, sB };
        int best = 0;
        for (int a = 1; a < 4; ++a) if (path[a] > path[best]) best = a;     // esl_vec_FArgMax: first maximum
        s1 = state[best]; k--; i--;
        break;
      }
      case sD: {
        const float p0 = (tMD[k - 1] == 0.0f) ? -INFINITY : ox.M_(i)[k - 1];
        const float p1 = (tDD[k - 1] == 0.0f) ? -INFINITY : ox.D_(i)[k - 1];
        s1 = (p0 >= p1) ? sM : sD; k--;
        break;
      }
      case sI: {
        const float p0 = (tMI[k] == 0.0f) ? -INFINITY : ox.M_(i - 1)[k];
        const float p1 = (tII[k] == 0.0f) ? -INFINITY : ox.I_(i - 1)[k];
        s1 = (p0 >= p1) ? sM : sI; i--;
        break;
      }
      case sN: s1 = (i == 0) ? sS : sN; break;
      case sC: {
        const float t1 = (om.xf[XC][LOOP] == 0.0f) ? 0.0f : 1.0f, t2 = (om.xf[XE][MOVE] == 0.0f) ? 0.0f : 1.0f;
        const float p0 = t1 * (ox.X(i - 1, xC_) + pp.X(i, xC_)), p1 = t2 * ox.X(i, xE_);
        s1 = (p0 > p1) ? sC : sE;
        break;
      }
      case sJ: {
        const float t1 = (om.xf[XJ][LOOP] == 0.0f) ? 0.0f : 1.0f, t2 = (om.xf[XE][LOOP] == 0.0f) ? 0.0f : 1.0f;
        const float p0 = t1 * (ox.X(i - 1, xJ_) + pp.X(i, xJ_)), p1 = t2 * ox.X(i, xE_);
        s1 = (p0 > p1) ? sJ : sE;
        break;
      }
      case sE: {   // striped visiting order: q outer, lanes inner; M beats D on ties (>= vs >)
        float mx = -INFINITY; int smax = -1, kmax = 0;
        const float *mr = ox.M_(i), *dr = ox.D_(i);
        for (int q = 0; q < Q; ++q) {
          for (int z = 0; z < 4; ++z) { const int kk = z * Q + q + 1; if (kk <= M && mr[kk] >= mx) { mx = mr[kk]; smax = sM; kmax = kk; } }
          for (int z = 0; z < 4; ++z) { const int kk = z * Q + q + 1; if (kk <= M && dr[kk] >  mx) { mx = dr[kk]; smax = sD; kmax = kk; } }
        }
        k = kmax; s1 = smax;
        break;
      }
      case sB: {
        const float t1 = (om.xf[XN][MOVE] == 0.0f) ? 0.0f : 1.0f, t2 = (om.xf[XJ][MOVE] == 0.0f) ? 0.0f : 1.0f;
        s1 = (t1 * ox.X(i, xN_) > t2 * ox.X(i, xJ_)) ? sN : sJ;
        break;
      }
      default: return P7X_EINVAL;
    }
    if (s1 == -1) return P7X_EINVAL;
    tr.append(s1, k, i, postprob(s1, s0, k, i));
    if ((s1 == sN || s1 == sJ || s1 == sC) && s1 == s0) i--;
    s0 = s1;
  }
  tr.reverse();
  return P7X_OK;
}

// ---------------------------------------------------------------- p7_StochasticTrace
// The choices are made through p7x_choice.hpp (integer thresholds against the generator's 32-bit state -- the reference's
// floating-point test restated exactly, see there): the same functions the device fills its choice records with.
int stochastic_trace(FastRng &rng, const Model &om, const Matrix &fx, int L, Trace &tr)
{
  const int M = om.M, Q = om.p->Q4();
  const float *bm = om.tf(0), *tMM = om.tf(1), *tIM = om.tf(2), *tDM = om.tf(3), *tMD = om.tf(4), *tMI = om.tf(5), *tII = om.tf(6), *tDD = om.tf(7);
  int i = L, k = 0;
  tr.clear();
  tr.append(sT, k, i, 0.0f);
  tr.append(sC, k, i, 0.0f);
  int s0 = sC;
  auto pair = [&](float a, float b) -> int {
    uint32_t T, bits;
    choice_pair(a, b, &T, &bits);
    return choice_pick_pair(T, bits, rng.next_u32());
  };
  while (s0 != sS) {
    int s1 = -1;
    switch (s0) {
      case sM: {
        if (i < 1 || k < 1) return P7X_EINVAL;
        uint32_t c4[4];
        choice_cell_m(fx.X(i - 1, xB_) * bm[k], fx.M_(i - 1)[k - 1] * tMM[k], fx.I_(i - 1)[k - 1] * tIM[k], fx.D_(i - 1)[k - 1] * tDM[k], c4);
        static const int state[4] = { sB, sM, sI, sD };
        s1 = state[choice_pick_m(c4, rng.next_u32())]; k--; i--;
        break;
      }
      case sD: {
        if (i < 1 || k < 1) return P7X_EINVAL;
        s1 = pair(fx.M_(i)[k - 1] * tMD[k - 1], fx.D_(i)[k - 1] * tDD[k - 1]) == 0 ? sM : sD; k--;
        break;
      }
      case sI: {
        if (i < 1 || k < 1) return P7X_EINVAL;
        s1 = pair(fx.M_(i - 1)[k] * tMI[k], fx.I_(i - 1)[k] * tII[k]) == 0 ? sM : sI; i--;
        break;
      }
      case sN: s1 = (i == 0) ? sS : sN; break;
      case sC: {
        if (i < 1) return P7X_EINVAL;
        s1 = pair(fx.X(i - 1, xC_) * om.xf[XC][LOOP], fx.X(i, xE_) * om.xf[XE][MOVE] * fx.X(i, xS_)) == 0 ? sC : sE;
        break;
      }
      case sJ: {
        if (i < 1) return P7X_EINVAL;
        s1 = pair(fx.X(i - 1, xJ_) * om.xf[XJ][LOOP], fx.X(i, xE_) * om.xf[XE][LOOP] * fx.X(i, xS_)) == 0 ? sJ : sE;
        break;
      }
      case sE: {
        double sum = 0.0;
        const double roll = rng.next();
        const float norm = (float) (1.0 / fx.X(i, xE_));
        const float *mr = fx.M_(i), *dr = fx.D_(i);
        bool done = false;
        for (int pass = 0; pass < 2 && !done; ++pass) {
          for (int q = 0; q < Q && !done; ++q) {
            for (int z = 0; z < 4 && !done; ++z) { const int kk = z * Q + q + 1; sum += (kk <= M ? mr[kk] * norm : 0.0f); if (roll < sum) { k = kk; s1 = sM; done = true; } }
            for (int z = 0; z < 4 && !done; ++z) { const int kk = z * Q + q + 1; sum += (kk <= M ? dr[kk] * norm : 0.0f); if (roll < sum) { k = kk; s1 = sD; done = true; } }
          }
          if (!done && sum < 0.99) return P7X_EINVAL;
        }
        if (!done) return P7X_EINVAL;
        break;
      }
      case sB: {
        s1 = pair(fx.X(i, xN_) * om.xf[XN][MOVE], fx.X(i, xJ_) * om.xf[XJ][MOVE]) == 0 ? sN : sJ;
        break;
      }
      default: return P7X_EINVAL;
    }
    if (s1 == -1) return P7X_EINVAL;
    tr.append(s1, k, i, 0.0f);
    if ((s1 == sN || s1 == sJ || s1 == sC) && s1 == s0) i--;
    s0 = s1;
  }
  tr.reverse();
  return P7X_OK;
}

// ---------------------------------------------------------------- p7_Null2_ByTrace
P7X_MULTIVERSION void null2_by_trace(const Model &om, const Trace &tr, int zstart, int zend, float *wm, float *wi, float *null2)
{ // wm / wi: [M+2] scratch.  The sums run in upstream's striped order (4 lanes, q ascending) for every residue at once:
  // residue-minor odds make the inner loop a plain vector update, and nodes the trace never visited contribute
  // exact zeros, so they are skipped.
  const int M = om.M, Q = om.p->Q4(), K = om.p->K;
  constexpr int KP = Model::kKpad;
  (void) wi;          // the insert slots stay zero (see below): adding them is adding 0.0f, which is exact, so they are left out
  for (int k = 0; k <= M + 1; ++k) wm[k] = 0.0f;
  float eN = 0.0f, eC = 0.0f, eJ = 0.0f;
  int Ld = 0;
  for (int z = zstart; z <= zend; ++z) {
    if (tr.i[z] == 0) continue;
    Ld++;
    // upstream computes a match/insert selector here and then never uses it: the usage count of an insert
    // emission lands in the MATCH slot of node k as well (the insert slot stays zero)
    if (tr.k[z] > 0) wm[tr.k[z]] += 1.0f;
    else switch (tr.st[z]) { case sN: eN += 1.0f; break; case sC: eC += 1.0f; break; case sJ: eJ += 1.0f; break; default: break; }
  }
  const float norm = 1.0 / (float) Ld;
  for (int k = 1; k <= M; ++k) wm[k] *= norm;
  eN *= norm; eC *= norm; eJ *= norm;
  const float xfactor = eN + eC + eJ;
  float acc[4][KP];
  for (int z = 0; z < 4; ++z) for (int x = 0; x < KP; ++x) acc[z][x] = 0.0f;
  const float *__restrict rfT = om.rfT.data();
  for (int q = 0; q < Q; ++q)
    for (int z = 0; z < 4; ++z) {
      const int k = q + 1 + z * Q;
      if (k > M) continue;
      const float w = wm[k];
      if (w == 0.0f) continue;
      const float *__restrict r = rfT + (size_t) k * KP;
      float *__restrict a = acc[z];
      for (int x = 0; x < KP; ++x) a[x] = a[x] + w * r[x];
    }
  for (int x = 0; x < K; ++x) null2[x] = ((acc[0][x] + acc[1][x]) + (acc[2][x] + acc[3][x])) + xfactor;
  finish_null2(*om.p, null2);
}

// ---------------------------------------------------------------- p7_spensemble
struct SpCoord { int idx, i, j, k, m; float prob; };

bool sp_link(const SpCoord &h1, const SpCoord &h2, float min_overlap, bool of_smaller, int max_diagdiff)
{
  int nov = std::min(h1.j, h2.j) - std::max(h1.i, h2.i) + 1;
  int n = of_smaller ? std::min(h1.j - h1.i + 1, h2.j - h2.i + 1) : std::max(h1.j - h1.i + 1, h2.j - h2.i + 1);
  if ((float) nov / (float) n < min_overlap) return false;
  nov = std::min(h1.m, h2.m) - std::max(h1.k, h2.k);
  n = of_smaller ? std::min(h1.m - h1.k + 1, h2.m - h2.k + 1) : std::max(h1.m - h1.k + 1, h2.m - h2.k + 1);
  if ((float) nov / (float) n < min_overlap) return false;
  int d1 = h1.i - h1.k, d2 = h2.i - h2.k; if (std::abs(d1 - d2) <= max_diagdiff) return true;
  d1 = h1.j - h1.m; d2 = h2.j - h2.m;     if (std::abs(d1 - d2) <= max_diagdiff) return true;
  return false;
}

// esl_cluster_SingleLinkage: same traversal order as Easel, so cluster numbers agree
void single_linkage(const std::vector<SpCoord> &sp, float min_overlap, bool of_smaller, int max_diagdiff,
                    std::vector<int> &assign, int &nc)
{
  const int n = (int) sp.size();
  std::vector<int> a(n), b(n);
  assign.assign(n, 0);
  for (int v = 0; v < n; ++v) a[v] = n - v - 1;
  int na = n; nc = 0;
  while (na > 0) {
    int v = a[na - 1]; na--;
    b[0] = v; int nb = 1;
    while (nb > 0) {
      v = b[nb - 1]; nb--;
      assign[v] = nc;
      for (int i = na - 1; i >= 0; --i)
        if (sp_link(sp[v], sp[a[i]], min_overlap, of_smaller, max_diagdiff)) { b[nb++] = a[i]; a[i] = a[na - 1]; na--; }
    }
    nc++;
  }
}

void sp_cluster(const std::vector<SpCoord> &sp, int nsamples, float min_overlap, bool of_smaller, int max_diagdiff,
                float min_posterior, float min_endpointp, std::vector<SpCoord> &sigc)
{
  std::vector<int> assign; int nc = 0;
  single_linkage(sp, min_overlap, of_smaller, max_diagdiff, assign, nc);
  const int n = (int) sp.size();
  sigc.clear();
  std::vector<int> epc;
  for (int c = 0; c < nc; ++c) {
    int idx_of_last = -1, ninc = 0;
    for (int h = 0; h < n; ++h) if (assign[h] == c) { if (sp[h].idx != idx_of_last) ninc++; idx_of_last = sp[h].idx; }
    if ((float) ninc / (float) nsamples < min_posterior) continue;
    int imin = 0, imax = 0, jmin = 0, jmax = 0, kmin = 0, kmax = 0, mmin = 0, mmax = 0;
    for (int h = 0; h < n; ++h) if (assign[h] == c) {
      if (imin == 0) { imin = imax = sp[h].i; jmin = jmax = sp[h].j; kmin = kmax = sp[h].k; mmin = mmax = sp[h].m; }
      else {
        imin = std::min(imin, sp[h].i); imax = std::max(imax, sp[h].i);
        jmin = std::min(jmin, sp[h].j); jmax = std::max(jmax, sp[h].j);
        kmin = std::min(kmin, sp[h].k); kmax = std::max(kmax, sp[h].k);
        mmin = std::min(mmin, sp[h].m); mmax = std::max(mmax, sp[h].m);
      }
    }
    const int epc_threshold = (int) std::ceil((float) ninc * min_endpointp);
    auto argmax = [&](int w) { int b = 0; for (int q = 1; q < w; ++q) if (epc[q] > epc[b]) b = q; return b; };
    int best_i = 0, best_j = 0, best_k = 0, best_m = 0;
    epc.assign(imax - imin + 1, 0);
    for (int h = 0; h < n; ++h) if (assign[h] == c) epc[sp[h].i - imin]++;
    for (int i = imin; i <= imax; ++i) if (epc[i - imin] >= epc_threshold) { best_i = i; break; }
    if (best_i == 0) best_i = argmax(imax - imin + 1) + imin;
    epc.assign(kmax - kmin + 1, 0);
    for (int h = 0; h < n; ++h) if (assign[h] == c) epc[sp[h].k - kmin]++;
    for (int k = kmin; k <= kmax; ++k) if (epc[k - kmin] >= epc_threshold) { best_k = k; break; }
    if (best_k == 0) best_k = argmax(kmax - kmin + 1) + kmin;
    epc.assign(jmax - jmin + 1, 0);
    for (int h = 0; h < n; ++h) if (assign[h] == c) epc[sp[h].j - jmin]++;
    for (int j = jmax; j >= jmin; --j) if (epc[j - jmin] >= epc_threshold) { best_j = j; break; }
    if (best_j == 0) best_j = argmax(jmax - jmin + 1) + jmin;
    epc.assign(mmax - mmin + 1, 0);
    for (int h = 0; h < n; ++h) if (assign[h] == c) epc[sp[h].m - mmin]++;
    for (int m = mmax; m >= mmin; --m) if (epc[m - mmin] >= epc_threshold) { best_m = m; break; }
    if (best_m == 0) best_m = argmax(mmax - mmin + 1) + mmin;
    if (best_i > best_j || best_k > best_m) continue;
    sigc.push_back(SpCoord{ c, best_i, best_j, best_k, best_m, (float) ninc / (float) nsamples });
  }
  std::stable_sort(sigc.begin(), sigc.end(), [](const SpCoord &a, const SpCoord &b) { return a.i < b.i; });
}

// ---------------------------------------------------------------- p7_alidisplay_Create (domain 0 of an OA trace)
void make_alidisplay(const Profile &p, const Trace &tr, const uint8_t *dsq, int L, Domain &dom)
{
  const Alphabet &abc = Alphabet::get(p.abc_type);
  const int N = (int) tr.st.size();
  int z1 = 0, z2;
  for (z1 = 0; z1 < N; ++z1) if (tr.st[z1] == sB) break;
  z1++;                                           // first state after B
  for (z2 = z1 + 1; z2 < N; ++z2) if (tr.st[z2] == sE) break;
  z2--;                                           // last state before E
  dom.N = z2 - z1 + 1;
  dom.hmmfrom = tr.k[z1]; dom.hmmto = tr.k[z2]; dom.M = p.M;
  dom.sqfrom = tr.i[z1]; dom.sqto = tr.i[z2]; dom.L = L;
  dom.model.assign(dom.N, ' '); dom.mline.assign(dom.N, ' '); dom.aseq.assign(dom.N, ' '); dom.ppline.assign(dom.N, ' ');
  const bool has_rf = p.rf.size() > 1 && p.rf[0] != 0, has_mm = p.mm.size() > 1 && p.mm[0] != 0, has_cs = p.cs.size() > 1 && p.cs[0] != 0;
  if (has_rf) dom.rfline.assign(dom.N, ' ');
  if (has_mm) dom.mmline.assign(dom.N, ' ');
  if (has_cs) dom.csline.assign(dom.N, ' ');
  // residue code of a consensus character (strchr over the alphabet's symbols, upper-cased): tabulated per call instead
  // of searched per aligned column -- most of this function's time was the search
  signed char code_of[256];
  std::memset(code_of, -1, sizeof code_of);
  for (int x = (int) std::strlen(abc.sym) - 1; x >= 0; --x) code_of[(unsigned char) abc.sym[x]] = (signed char) x;     // first occurrence wins, as strchr
  auto digitize = [&](char c) -> int { return code_of[(unsigned char) std::toupper((unsigned char) c)]; };
  for (int z = z1; z <= z2; ++z) {
    const int k = tr.k[z], i = tr.i[z], s = tr.st[z], o = z - z1;
    const char cons = (k >= 1 && (int) p.consensus.size() > k) ? p.consensus[k] : 'x';
    if (has_rf) dom.rfline[o] = (s == sI) ? '.' : p.rf[k];
    if (has_mm) dom.mmline[o] = (s == sI) ? '.' : p.mm[k];
    if (has_cs) dom.csline[o] = (s == sI) ? '.' : p.cs[k];
    switch (s) {
      case sM: {
        const int x = dsq[i];
        dom.model[o] = cons;
        if (x == digitize(cons)) dom.mline[o] = cons;
        else if (p.rf_[(size_t) x * (p.M + 1) + k] > 1.0f) dom.mline[o] = '+';
        else dom.mline[o] = ' ';
        dom.aseq[o] = (char) std::toupper((unsigned char) abc.sym[x]);
        break;
      }
      case sI: dom.model[o] = '.'; dom.mline[o] = ' '; dom.aseq[o] = (char) std::tolower((unsigned char) abc.sym[dsq[i]]); break;
      case sD: dom.model[o] = cons; dom.mline[o] = ' '; dom.aseq[o] = '-'; break;
      default: break;
    }
    if (s == sD) dom.ppline[o] = '.';
    else { const float pv = tr.pp[z]; dom.ppline[o] = (pv + 0.05 >= 1.0) ? '*' : (char) ((int) ((pv + 0.05) * 10.0) + '0'); }
  }
}

struct Workspace { Matrix fwd, bck; Trace tr; std::vector<float> wm, wi; };

// ---------------------------------------------------------------- rescore_isolated_domain
// upstream reparameterize_model + p7_oprofile_UpdateFwdEmissionScores: the background becomes
//   bg'[x] = (1 - s) * composition(dsq[i..j])[x] + s * bg->f[x],   s = 25 / min(100, max(50, n)),  n = window length
// (0.25 for every window of 100 residues or more), and the match odds rf'[x][k] = match_prob[k][x] / bg'[x], degenerate
// codes by expectation under bg'.  Pinned by bmyD1/bmyD2.tbl and the RF00001 answers (all windows there are > 100 nt; the
// short-window branch of s is restated from upstream and not pinned by any fixture).
static void reparameterize(const Profile &p, const LongTargetOpts &lt, const uint8_t *dsq, int n, int i, int j, std::vector<float> &rf)
{
  const int M = p.M, K = p.K, Kp = p.Kp;
  const Alphabet &abc = Alphabet::get(p.abc_type);
  const float bg_smooth = 25.0f / (float) std::min(100, std::max(50, n));
  float cnt[MAXK];
  for (int x = 0; x < K; ++x) cnt[x] = 0.0f;
  for (int pos = i; pos <= j; ++pos) {           // esl_sq_CountResidues: degenerate residues count fractionally
    const int x = dsq[pos];
    if (x < K) cnt[x] += 1.0f;
    else if (x > K && x <= Kp - 3) {
      int nd = 0; for (int y = 0; y < K; ++y) nd += abc.degen[x][y] ? 1 : 0;
      for (int y = 0; y < K; ++y) if (abc.degen[x][y]) cnt[y] += 1.0f / (float) nd;
    }
  }
  float tot = 0.0f; for (int x = 0; x < K; ++x) tot += cnt[x];
  float bgn[MAXK];
  for (int x = 0; x < K; ++x) bgn[x] = (1.0f - bg_smooth) * (tot > 0 ? cnt[x] / tot : p.bgf[x]) + bg_smooth * p.bgf[x];
  rf.assign((size_t) Kp * (M + 1), 0.0f);
  for (int k = 1; k <= M; ++k) {
    float sc[MAXKP];
    for (int x = 0; x < K; ++x) sc[x] = logf(lt.match_prob[(size_t) k * K + x] / bgn[x]);
    sc[K] = sc[Kp - 2] = sc[Kp - 1] = -INFINITY;
    for (int x = K + 1; x <= Kp - 3; ++x) {      // esl_abc_FExpectScVec with the new background
      float num = 0.f, den = 0.f;
      for (int y = 0; y < K; ++y) if (abc.degen[x][y]) { num += sc[y] * bgn[y]; den += bgn[y]; }
      sc[x] = num / den;
    }
    for (int x = 0; x < Kp; ++x) rf[(size_t) x * (M + 1) + k] = expf(sc[x]);
  }
}

int rescore_isolated_domain(const Profile &p, Model &om, const uint8_t *dsq, int L, int i, int j, bool null2_is_done,
                            Workspace &ws, DomainDefResult &dd)
{
  const LongTargetOpts *lt = om.lt;
  int Ld = j - i + 1;
  float envsc = 0.0f, oasc = 0.0f;
  float save_xf[4][2];
  thread_local std::vector<float> rf_lt;
  if (lt) std::memcpy(save_xf, om.xf, sizeof(save_xf));
  struct Restore { Model &om; const LongTargetOpts *lt; float (*xf)[2];
                   ~Restore() { if (lt) { om.rf_over = nullptr; std::memcpy(om.xf, xf, sizeof(float) * 8); } } } restore{ om, lt, save_xf };
  // Long targets (upstream rescore_isolated_domain with long_target = TRUE): the envelope is scored unihit under a length
  // model of its OWN length (p7_oprofile_ReconfigRestLength(om, j-i+1); the pipeline re-expresses the score for max_length
  // afterwards) and, with null2 on, against emissions re-derived for a background mixed with the envelope's composition.
  auto lt_setup = [&]() {
    om.configure(false, Ld);
    if (lt->do_null2) { reparameterize(p, *lt, dsq, L, i, j, rf_lt); om.rf_over = rf_lt.data(); om.sr_of = nullptr; }     // (same buffer, new odds: the striped copy is stale)
  };
  auto align = [&]() -> int {
    { ProfScope ps(5); forward_full(om, dsq + i - 1, Ld, ws.fwd, &envsc); }
    { ProfScope ps(6); backward_full(om, dsq + i - 1, Ld, ws.fwd, ws.bck, nullptr); }
    { ProfScope ps(7); if (decoding(om, ws.fwd, ws.bck) == P7X_ERANGE) return P7X_ENORESULT; }   // repetitive garbage; the envelope is dropped
    { ProfScope ps(8); optimal_accuracy(om, ws.bck, ws.fwd, &oasc); }                              // <fwd> now holds the OA matrix
    ProfScope ps9(9);
    if (oa_trace(om, ws.bck, ws.fwd, ws.tr) != P7X_OK) return P7X_EINVAL;
    for (size_t z = 0; z < ws.tr.st.size(); ++z) if (ws.tr.i[z] > 0) ws.tr.i[z] += i - 1;
    return P7X_OK;
  };
  if (lt) lt_setup();
  int st = align();
  if (st != P7X_OK) return st;
  Domain dom;
  make_alidisplay(p, ws.tr, dsq, L, dom);
  if (lt && (i < dom.sqfrom - lt->max_env_extra || j > dom.sqto + lt->max_env_extra)) {
    // long targets often give envelopes far wider than the alignment (a repetitive stretch of the model collecting
    // weak matches): trim the envelope to the alignment +- max_env_extra and do it again
    i = std::max<int>(i, (int) dom.sqfrom - lt->max_env_extra);
    j = std::min<int>(j, (int) dom.sqto + lt->max_env_extra);
    Ld = j - i + 1;
    lt_setup();
    if ((st = align()) != P7X_OK) return st;
    dom = Domain();
    make_alidisplay(p, ws.tr, dsq, L, dom);
  }
  float domcorrection = 0.0f;
  if (lt) {
    if (lt->do_null2) {
      // the bias of a long-target envelope is the score it loses against the composition-adjusted background; the
      // envelope score itself is the unmodified model's (Forward again with the original emissions)
      float orig = 0.0f;
      om.rf_over = nullptr;
      forward_full(om, dsq + i - 1, Ld, ws.fwd, &orig);
      domcorrection = std::max(0.0f, orig - envsc);
      envsc = orig;
    }
  } else {
    if (!null2_is_done) {
      float null2[MAXKP];
      ProfScope psn(10);
      null2_by_expectation(om, ws.bck, null2);
      float ln2[MAXKP];                                   // one logarithm per residue code, not per residue
      for (int x = 0; x < p.Kp; ++x) ln2[x] = logf(null2[x]);
      for (int pos = i; pos <= j; ++pos) dd.n2sc[pos] = ln2[dsq[pos]];
    }
    for (int pos = i; pos <= j; ++pos) domcorrection += dd.n2sc[pos];
  }
  dom.domcorrection = domcorrection;
  dom.ienv = i; dom.jenv = j; dom.envsc = envsc; dom.oasc = oasc;
  dom.iali = dom.sqfrom; dom.jali = dom.sqto;
  dd.dcl.push_back(std::move(dom));
  return P7X_OK;
}

} // anonymous namespace

// ---------------------------------------------------------------- p7_domaindef_ByPosteriorHeuristics
// Step 1: posterior decoding of the special states (p7_DomainDecoding) and the region scan.
int domaindef_regions(const Profile &p, int L, const float *fx, const float *bx, DomainDefResult &dd, std::vector<Region> &regs)
{
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;                         // p7_domaindef.pxd:39-41
  Model om{ &p, p.M, {} };
  dd.dcl.clear(); dd.n2sc.assign(L + 1, 0.0f); dd.multi.clear();
  dd.nregions = dd.nclustered = dd.noverlaps = dd.nenvelopes = 0;
  regs.clear();
  om.configure(true, L);
  thread_local std::vector<float> btot, etot, mocc;
  btot.resize(L + 1); etot.resize(L + 1); mocc.resize(L + 1);
  {
    float scaleproduct = 1.0 / bx[0 * NX + xN_];
    btot[0] = etot[0] = mocc[0] = 0.0f;
    for (int i = 1; i <= L; ++i) {
      btot[i] = btot[i - 1] + (fx[(i - 1) * NX + xB_] * bx[(i - 1) * NX + xB_]) * fx[(i - 1) * NX + xS_] * scaleproduct;
      etot[i] = etot[i - 1] + (fx[i * NX + xE_] * bx[i * NX + xE_]) * fx[i * NX + xS_] * scaleproduct;
      float njcp = fx[(i - 1) * NX + xN_] * bx[i * NX + xN_] * om.xf[XN][LOOP] * scaleproduct;
      njcp += fx[(i - 1) * NX + xJ_] * bx[i * NX + xJ_] * om.xf[XJ][LOOP] * scaleproduct;
      njcp += fx[(i - 1) * NX + xC_] * bx[i * NX + xC_] * om.xf[XC][LOOP] * scaleproduct;
      mocc[i] = 1. - njcp;
    }
    if (std::isinf(scaleproduct)) return P7X_ERANGE;
  }
  dd.nexpected = btot[L];
  int i = -1; bool triggered = false;
  for (int j = 1; j <= L; ++j) {
    if (!triggered) {
      if (mocc[j] - (btot[j] - btot[j - 1]) < rt2) i = j;
      else if (i == -1) i = j;
      if (mocc[j] >= rt1) triggered = true;
    } else if (mocc[j] - (etot[j] - etot[j - 1]) < rt2) {
      dd.nregions++;
      float mx = -1.0f;                                                      // is_multidomain_region
      for (int z = i; z <= j; ++z) mx = std::max(mx, std::min(etot[z] - etot[i - 1], btot[j] - btot[z - 1]));
      regs.push_back(Region{ i, j, mx >= rt3 });
      i = -1; triggered = false;
    }
  }
  return P7X_OK;
}

// Step 2 for a multi-domain region: region_trace_ensemble (sampled tracebacks from a multihit Forward matrix of the
// region, single-linkage clustering of their domain coordinates), then every surviving envelope is rescored.
static thread_local const LongTargetOpts *t_long_target = nullptr;     // set by the long-target pipeline around its calls

int domaindef_multi_region(const Profile &p, const uint8_t *dsq, int L, int i, int j, uint32_t seed, bool do_reseeding,
                           MultiRegionState &state, DomainDefResult &dd, std::vector<Domain> &out,
                           std::vector<EnvelopeRequest> *defer2, int item, const EnsembleResult *ens)
{
  const int nsamples = 200;                                                  // p7_domaindef.pxd:43-48
  const float min_overlap = 0.8f, min_posterior = 0.25f, min_endpointp = 0.02f;
  const bool of_smaller = true; const int max_diagdiff = 4;
  thread_local Workspace ws;
  Model om{ &p, p.M, {} };
  om.lt = t_long_target;
  om.prepare();
  const int Lr = j - i + 1;
  std::vector<SpCoord> sp;
  dd.nclustered++;
  if (ens && ens->status >= 0 && (ens->status & 64) && ens->status < 128) dd.nens_redone++;
  if (ens && ens->status == 0 && do_reseeding && !om.lt) {
    dd.nens_device++;
    // the ensemble was sampled on the device (p7x_ensemble.hip): end points and null2 sums are in; a sample's domains come
    // in traceback order and go into the list first domain first, as p7_trace_Index numbers them
    sp.reserve((size_t) ens->ndom);
    for (int a = 0; a < ens->ndom; ) {
      int b = a;
      while (b < ens->ndom && ens->dom[(size_t) b * 5] == ens->dom[(size_t) a * 5]) ++b;
      for (int d = b - 1; d >= a; --d) {
        const int32_t *o = ens->dom + (size_t) d * 5;
        sp.push_back(SpCoord{ o[0], o[1] + i - 1, o[2] + i - 1, o[3], o[4], 0.0f });
      }
      a = b;
    }
    for (int pos = 1; pos <= Lr; ++pos) dd.n2sc[i + pos - 1] = logf(ens->n2[pos] / (float) nsamples);
  } else {
    om.prepare_rfT();
    ws.wm.resize(p.M + 2); ws.wi.resize(p.M + 2);
    FastRng rng; rng.seed = state.rng_seed; rng.x = state.rng_x;
    if (!state.started) { rng.init(seed); state.started = true; }
    om.configure(true, L);
    { ProfScope ps(1); forward_full(om, dsq + i - 1, j - i + 1, ws.fwd, nullptr); }
    for (int pos = i; pos <= j; ++pos) dd.n2sc[pos] = 0.0f;
    if (do_reseeding) rng.init(seed);
    float null2[MAXKP];
    for (int t = 0; t < nsamples; ++t) {
      { ProfScope ps(2); if (stochastic_trace(rng, om, ws.fwd, Lr, ws.tr) != P7X_OK) return P7X_EINVAL; }
      ws.tr.index();
      int pos = 1;
      for (int d = 0; d < ws.tr.ndom; ++d) {
        sp.push_back(SpCoord{ t, ws.tr.sqfrom[d] + i - 1, ws.tr.sqto[d] + i - 1, ws.tr.hmmfrom[d], ws.tr.hmmto[d], 0.0f });
        { ProfScope ps(3); null2_by_trace(om, ws.tr, ws.tr.tfrom[d], ws.tr.tto[d], ws.wm.data(), ws.wi.data(), null2); }
        for (; pos <= ws.tr.sqfrom[d]; ++pos) dd.n2sc[i + pos - 1] += 1.0f;   // sic: the first domain residue counts as "outside"
        for (; pos <= ws.tr.sqto[d]; ++pos) dd.n2sc[i + pos - 1] += null2[dsq[i + pos - 1]];
      }
      for (; pos <= Lr; ++pos) dd.n2sc[i + pos - 1] += 1.0f;
    }
    state.rng_seed = rng.seed; state.rng_x = rng.x;
    if (ens && ens->raw_out) {            // test seam: the ensemble as sampled, before the logarithm and the clustering
      ens->raw_out->dom.clear();
      for (const SpCoord &c : sp) { ens->raw_out->dom.push_back(c.idx); ens->raw_out->dom.push_back(c.i - i + 1); ens->raw_out->dom.push_back(c.j - i + 1); ens->raw_out->dom.push_back(c.k); ens->raw_out->dom.push_back(c.m); }
      ens->raw_out->n2.assign((size_t) Lr + 1, 0.0f);
      for (int pos = 1; pos <= Lr; ++pos) ens->raw_out->n2[(size_t) pos] = dd.n2sc[i + pos - 1];
    }
    for (int pos = i; pos <= j; ++pos) dd.n2sc[pos] = logf(dd.n2sc[pos] / (float) nsamples);
  }
  std::vector<SpCoord> sigc;
  { ProfScope ps(4); sp_cluster(sp, nsamples, min_overlap, of_smaller, max_diagdiff, min_posterior, min_endpointp, sigc); }
  // remove envelopes dominated (>= 80% overlap of the smaller) by a more probable one
  const int nc0 = (int) sigc.size();
  std::vector<char> dominated(nc0, 0);
  for (int d = 0; d < nc0; ++d)
    for (int d2 = d + 1; d2 < nc0; ++d2) {
      const int nov = std::min(sigc[d].j, sigc[d2].j) - std::max(sigc[d].i, sigc[d2].i) + 1;
      if (nov == 0) break;
      const int n = std::min(sigc[d].j - sigc[d].i + 1, sigc[d2].j - sigc[d2].i + 1);
      if ((float) nov / (float) n >= 0.8f) { if (sigc[d].prob > sigc[d2].prob) dominated[d2] = 1; else dominated[d] = 1; }
    }
  om.configure(false, L);
  if (defer2 && !om.lt) {                  // the device rescores the clustered envelopes; overlaps are counted when its answers are in
    out.clear();
    for (int d = 0; d < nc0; ++d) {
      if (dominated[d]) continue;
      dd.nenvelopes++;
      Domain ph; ph.ienv = sigc[d].i; ph.jenv = sigc[d].j; ph.deferred2 = (int) defer2->size();
      defer2->push_back(EnvelopeRequest{ item, (int32_t) sigc[d].i, (int32_t) sigc[d].j });
      out.push_back(std::move(ph));
    }
    return P7X_OK;
  }
  int last_j2 = 0;
  std::vector<Domain> keep;
  keep.swap(dd.dcl);                       // rescore_isolated_domain() appends to dd.dcl: collect this region's domains apart
  for (int d = 0; d < nc0; ++d) {
    if (dominated[d]) continue;
    const int i2 = sigc[d].i, j2 = sigc[d].j;
    if (i2 <= last_j2) dd.noverlaps++;
    dd.nenvelopes++;
    if (rescore_isolated_domain(p, om, dsq, L, i2, j2, true, ws, dd) == P7X_OK) last_j2 = j2;
  }
  out = std::move(dd.dcl);
  dd.dcl.swap(keep);
  return P7X_OK;
}

static int dispatch_regions(const Profile &p, const uint8_t *dsq, int L, const Region *regs, int nregs, uint32_t seed,
                            bool do_reseeding, DomainDefResult &dd, std::vector<EnvelopeRequest> *defer, int item);

int domaindef_by_posterior_heuristics(const Profile &p, const uint8_t *dsq, int L, const float *fx, const float *bx,
                                      uint32_t seed, bool do_reseeding, DomainDefResult &dd,
                                      std::vector<EnvelopeRequest> *defer, int item)
{
  thread_local std::vector<Region> regs;
  const int st = domaindef_regions(p, L, fx, bx, dd, regs);
  if (st != P7X_OK) return st;
  return dispatch_regions(p, dsq, L, regs.data(), (int) regs.size(), seed, do_reseeding, dd, defer, item);
}

int domaindef_from_regions(const Profile &p, const uint8_t *dsq, int L, float nexpected, const Region *regs, int nregs,
                           uint32_t seed, bool do_reseeding, DomainDefResult &dd, std::vector<EnvelopeRequest> *defer, int item)
{
  dd.dcl.clear(); dd.n2sc.assign(L + 1, 0.0f); dd.multi.clear();
  dd.nclustered = dd.noverlaps = dd.nenvelopes = 0;
  dd.nregions = nregs; dd.nexpected = nexpected;
  return dispatch_regions(p, dsq, L, regs, nregs, seed, do_reseeding, dd, defer, item);
}

static int dispatch_regions(const Profile &p, const uint8_t *dsq, int L, const Region *regs, int nregs, uint32_t seed,
                            bool do_reseeding, DomainDefResult &dd, std::vector<EnvelopeRequest> *defer, int item)
{
  Model om{ &p, p.M, {} };
  om.lt = t_long_target;
  thread_local Workspace ws;
  bool prepared = false;
  MultiRegionState state;
  for (int ri = 0; ri < nregs; ++ri) {
    const Region &r = regs[ri];
    if (r.multi) {
      if (defer) {                    // resolved later by domaindef_multi_region(): leave a marker in domain order
        Domain ph; ph.ienv = r.i; ph.jenv = r.j; ph.deferred = -2; ph.multi_slot = (int) dd.multi.size();
        dd.multi.emplace_back();
        dd.dcl.push_back(std::move(ph));
        continue;
      }
      std::vector<Domain> doms;
      const int st2 = domaindef_multi_region(p, dsq, L, r.i, r.j, seed, do_reseeding, state, dd, doms);
      if (st2 != P7X_OK) return st2;
      for (Domain &d : doms) dd.dcl.push_back(std::move(d));
    } else {
      dd.nenvelopes++;
      if (defer) {                   // rescored on the device; domaindef_finish_deferred() fills the placeholder
        Domain ph; ph.ienv = r.i; ph.jenv = r.j; ph.deferred = (int) defer->size();
        defer->push_back(EnvelopeRequest{ item, r.i, r.j });
        dd.dcl.push_back(std::move(ph));
      } else {
        if (!prepared) { om.prepare(); om.configure(false, L); prepared = true; }
        rescore_isolated_domain(p, om, dsq, L, r.i, r.j, false, ws, dd);
      }
    }
  }
  return P7X_OK;
}

// The multi-domain regions left behind by the deferring call above, in order.
int domaindef_finish_multi(const Profile &p, const uint8_t *dsq, int L, uint32_t seed, bool do_reseeding, DomainDefResult &dd,
                           std::vector<EnvelopeRequest> *defer2, int item, const EnsembleResult *const *ens)
{
  MultiRegionState state;
  int nth = 0;
  for (Domain &d : dd.dcl) {
    if (d.deferred != -2) continue;
    const int st = domaindef_multi_region(p, dsq, L, (int) d.ienv, (int) d.jenv, seed, do_reseeding, state, dd, dd.multi[(size_t) d.multi_slot],
                                          defer2, item, ens ? ens[nth] : nullptr);
    ++nth;
    if (st != P7X_OK) return st;
  }
  return P7X_OK;
}

uint32_t fast_rng_state(uint32_t seed) { FastRng r; r.init(seed); return r.x; }

int parser_rows_upstream(const Profile &p, const uint8_t *dsq, int L, std::vector<float> &fx, std::vector<float> &bx)
{
  thread_local Workspace ws;
  Model om{ &p, p.M, {} };
  om.prepare(0);
  om.configure(true, L);
  int st = forward_full(om, dsq, L, ws.fwd, nullptr);
  if (st != P7X_OK) return st;
  st = backward_full(om, dsq, L, ws.fwd, ws.bck, nullptr);
  if (st != P7X_OK) return st;
  fx.assign(ws.fwd.x.begin(), ws.fwd.x.begin() + (size_t) (L + 1) * NX);
  bx.assign(ws.bck.x.begin(), ws.bck.x.begin() + (size_t) (L + 1) * NX);
  return P7X_OK;
}

// Second half of rescore_isolated_domain() for envelopes rescored by the device kernel: trace -> alignment
// display, null2 odds -> per-residue corrections.  req_index[n] is the position in <res> of local request n
// (Domain::deferred of the placeholders of this target).
int domaindef_finish_deferred(const Profile &p, const uint8_t *dsq, int L, const std::vector<EnvelopeResult> &res,
                              const std::vector<int> &req_index, DomainDefResult &dd,
                              const std::vector<EnvelopeResult> *res2, const std::vector<int> *req_index2)
{
  thread_local Workspace ws;
  std::vector<Domain> kept;
  kept.reserve(dd.dcl.size());
  // the device's answer for envelope i..j -> a Domain; null2_done: the region's ensemble already set dd.n2sc on i..j
  Model om_exact{ &p, p.M, {} };
  bool om_exact_ready = false;
  auto from_result = [&](const EnvelopeResult &r, int i, int j, bool null2_done, Domain &dom) -> bool {
    if (r.status & 2) return false;                  // p7_Decoding range error: the envelope is dropped
    if (r.status & 0xff & ~(3 | 64)) return false;   // traceback failure: upstream's rescore returns without a domain
    if (r.status & 64) {
      // a near-tie on the device's optimal-accuracy trace (p7x_envelope.hip, cfg.oa_guard): this envelope again with
      // the host twin, which performs the reference's operations in the reference's order
      if (!om_exact_ready) { om_exact.prepare(); om_exact.configure(false, L); om_exact_ready = true; }      // upstream's order (unless the test seam "host_order" asks for the device's)
      DomainDefResult one;
      one.n2sc.swap(dd.n2sc);
      const int st = rescore_isolated_domain(p, om_exact, dsq, L, i, j, null2_done, ws, one);
      dd.n2sc.swap(one.n2sc);
      dd.nneartie++;
      for (int b = 0; b < 8; ++b) if (r.status & (1 << (8 + b))) dd.neartie_why[b]++;
      if (st != P7X_OK || one.dcl.empty()) return false;
      dom = std::move(one.dcl[0]);
      return true;
    }
    Trace &tr = ws.tr;
    tr.clear();
    for (int z = 0; z < r.ntrace; ++z) tr.append((int) (r.ta[z] & 0xffu), (int) ((r.ta[z] >> 8) & 0xffffu), r.ti[z], r.tp[z]);
    tr.reverse();
    for (size_t z = 0; z < tr.st.size(); ++z) if (tr.i[z] > 0) tr.i[z] += i - 1;
    make_alidisplay(p, tr, dsq, L, dom);
    if (!null2_done) {
      float null2[MAXKP];
      for (int x = 0; x < p.K; ++x) null2[x] = r.null2[x];
      finish_null2(p, null2);
      float ln2[MAXKP];                                   // one logarithm per residue code, not per residue
      for (int x = 0; x < p.Kp; ++x) ln2[x] = logf(null2[x]);
      for (int pos = i; pos <= j; ++pos) dd.n2sc[pos] = ln2[dsq[pos]];
    }
    float domcorrection = 0.0f;
    for (int pos = i; pos <= j; ++pos) domcorrection += dd.n2sc[pos];
    dom.domcorrection = domcorrection;
    dom.ienv = i; dom.jenv = j; dom.envsc = r.envsc; dom.oasc = r.oasc;
    dom.iali = dom.sqfrom; dom.jali = dom.sqto;
    return true;
  };
  for (Domain &d : dd.dcl) {
    if (d.deferred == -2) {
      int last_j2 = 0;
      for (Domain &m : dd.multi[(size_t) d.multi_slot]) {
        if (m.deferred2 < 0 || !res2 || !req_index2) { kept.push_back(std::move(m)); continue; }
        const int i2 = (int) m.ienv, j2 = (int) m.jenv;
        if (i2 <= last_j2) dd.noverlaps++;
        Domain dom;
        if (from_result((*res2)[(size_t) (*req_index2)[(size_t) m.deferred2]], i2, j2, true, dom)) { last_j2 = j2; kept.push_back(std::move(dom)); }
      }
      continue;
    }
    if (d.deferred < 0) { kept.push_back(std::move(d)); continue; }
    Domain dom;
    if (from_result(res[(size_t) req_index[(size_t) d.deferred]], (int) d.ienv, (int) d.jenv, false, dom)) kept.push_back(std::move(dom));
  }
  dd.dcl = std::move(kept);
  return P7X_OK;
}

} // namespace p7x

// Test seam: one choice point through the integer thresholds of p7x_choice.hpp and through esl_rnd_FChoose as the
// reference writes it (floating-point test against roll = x / 2^32); x is the generator's state AFTER the draw.
extern "C" int p7x_debug_choice(const float *p, int n, uint32_t x, int *via_thresholds, int *via_fchoose)
{
  using namespace p7x;
  if (!p || n < 2 || n > 4 || !via_thresholds || !via_fchoose) return P7X_EINVAL;
  float a[4], b[4];
  for (int i = 0; i < n; ++i) a[i] = b[i] = p[i];
  uint32_t T[3] = { 0, 0, 0 }, fb = 0;
  choice_thresholds(a, n, T, &fb);
  *via_thresholds = choice_pick(T, fb, n, x);
  fnorm(b, n);
  FastRng r; r.x = x;
  const double roll = (double) x / 4294967296.0;
  const double norm = fsum(b, n);
  double sum = 0.0;
  int pick = -1;
  for (int i = 0; i < n && pick < 0; ++i) { sum += b[i]; if (roll < sum / norm) pick = i; }
  if (pick < 0) { pick = 0; for (int i = n - 1; i >= 0; --i) if (b[i] > 0.0f) { pick = i; break; } }
  *via_fchoose = pick;
  return P7X_OK;
}

// Calibration seam of the ensemble walk's near-threshold guard (p7x_ensemble.hip): the Forward matrix of region i..j of
// dsq1[1..L] in both summation orders -- the device's lane chunks and upstream's stripes -- and, for every cell, the integer
// thresholds of the three choice points a traceback can meet there (p7x_choice.hpp).  out[0] = thresholds compared,
// out[1] = the largest |T_lanes - T_upstream| / 2^32, out[2 + b] = how many differ by more than 2^-(24 - b), b = 0..9.
extern "C" int p7x_debug_order_spread(const p7x_oprofile *om, const uint8_t *dsq1, int32_t L, int32_t i, int32_t j, int multihit, double *out12)
{
  using namespace p7x;
  if (!om || !dsq1 || !out12 || i < 1 || j < i || j > L) { set_error("p7x_debug_order_spread: bad arguments"); return P7X_EINVAL; }
  const Profile &p = om->p;
  Model a{ &p, p.M, {} }, b{ &p, p.M, {} };
  a.prepare(1); b.prepare(0);
  a.configure(multihit != 0, L); b.configure(multihit != 0, L);
  Matrix fa, fb;
  const int Lr = j - i + 1, M = p.M;
  forward_full(a, dsq1 + i - 1, Lr, fa, nullptr);
  forward_full(b, dsq1 + i - 1, Lr, fb, nullptr);
  for (int q = 0; q < 12; ++q) out12[q] = 0.0;
  auto note = [&](uint32_t ta, uint32_t tb) {
    const double d = std::fabs((double) ta - (double) tb) / 4294967296.0;
    out12[0] += 1.0;
    if (d > out12[1]) out12[1] = d;
    for (int q = 0; q < 10; ++q) if (d > std::ldexp(1.0, -(24 - q))) out12[2 + q] += 1.0;
  };
  const float *bm = a.tf(0), *tMM = a.tf(1), *tIM = a.tf(2), *tDM = a.tf(3), *tMD = a.tf(4), *tMI = a.tf(5), *tII = a.tf(6), *tDD = a.tf(7);
  for (int r = 1; r <= Lr; ++r)
    for (int k = 1; k <= M; ++k) {
      uint32_t ca[4], cb[4], ta, tb, fa_, fb_;
      choice_cell_m(fa.X(r - 1, xB_) * bm[k], fa.M_(r - 1)[k - 1] * tMM[k], fa.I_(r - 1)[k - 1] * tIM[k], fa.D_(r - 1)[k - 1] * tDM[k], ca);
      choice_cell_m(fb.X(r - 1, xB_) * bm[k], fb.M_(r - 1)[k - 1] * tMM[k], fb.I_(r - 1)[k - 1] * tIM[k], fb.D_(r - 1)[k - 1] * tDM[k], cb);
      for (int q = 0; q < 3; ++q) note(ca[q], cb[q]);
      choice_pair(fa.M_(r - 1)[k] * tMI[k], fa.I_(r - 1)[k] * tII[k], &ta, &fa_);
      choice_pair(fb.M_(r - 1)[k] * tMI[k], fb.I_(r - 1)[k] * tII[k], &tb, &fb_);
      note(ta, tb);
      if (k > 1) {
        choice_pair(fa.M_(r)[k - 1] * tMD[k - 1], fa.D_(r)[k - 1] * tDD[k - 1], &ta, &fa_);
        choice_pair(fb.M_(r)[k - 1] * tMD[k - 1], fb.D_(r)[k - 1] * tDD[k - 1], &tb, &fb_);
        note(ta, tb);
      }
    }
  return P7X_OK;
}

#include "p7x_longtarget.inc.hpp"
