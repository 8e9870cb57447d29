// p7x_profile.cpp -- host side of libp7x: alphabets, query preparation and small statistics.
//
// Replaces, for the search path only, what pyhmmer reaches through
//   Profile.configure      -> p7_ProfileConfig      (reference plan7.pyx:8082, modelconfig.pxd:7-10)
//   Profile.to_optimized   -> p7_oprofile_Convert   (reference plan7.pyx:4961, impl_sse/p7_oprofile.pxd:121)
// The numbers produced are the reference's (they are checked bit-for-bit against the pressed
// .h3f/.h3p fixtures in tests/), but the storage is un-striped: node k lives at index k.  Farrar
// striping is an SSE artefact; the device kernels build their own layouts from these arrays, and
// p7x_oprofile_striped() re-creates the impl_sse views on demand for API parity.
#include "p7x_internal.hpp"
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>

namespace p7x {
std::atomic<int> g_debug_opt[OPT_COUNT];
static const bool g_debug_opt_ready = [] { for (auto &o : g_debug_opt) o.store(-1); return true; }();     // every knob starts unset, however many there are
int debug_opt(int which) { return (which >= 0 && which < OPT_COUNT) ? g_debug_opt[which].load(std::memory_order_relaxed) : -1; }


static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

// ---------------------------------------------------------------- alphabets (easel.pyx:313-347)
static Alphabet make_alphabet(int type)
{
  Alphabet a{};
  a.type = type;
  std::memset(a.degen, 0, sizeof(a.degen));
  if (type == P7X_AMINO) {
    a.K = 20; a.Kp = 29; a.sym = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
    struct { int x; const char *set; } d[] = { {21, "ND"}, {22, "IL"}, {23, "QE"}, {24, "K"}, {25, "C"} };
    for (auto &e : d) for (const char *c = e.set; *c; ++c) a.degen[e.x][std::strchr(a.sym, *c) - a.sym] = 1;
    for (int y = 0; y < 20; ++y) a.degen[26][y] = 1;
  } else {
    a.K = 4; a.Kp = 18; a.sym = (type == P7X_RNA) ? "ACGU-RYMKSWHBVDN*~" : "ACGT-RYMKSWHBVDN*~";
    const char *ref = "ACGT";
    struct { int x; const char *set; } d[] = { {5, "AG"}, {6, "CT"}, {7, "AC"}, {8, "GT"}, {9, "CG"}, {10, "AT"},
                                               {11, "ACT"}, {12, "CGT"}, {13, "ACG"}, {14, "AGT"}, {15, "ACGT"} };
    for (auto &e : d) for (const char *c = e.set; *c; ++c) a.degen[e.x][std::strchr(ref, *c) - ref] = 1;
  }
  for (int x = 0; x < a.K; ++x) a.degen[x][x] = 1;
  return a;
}

const Alphabet &Alphabet::get(int type)
{
  static const Alphabet amino = make_alphabet(P7X_AMINO), dna = make_alphabet(P7X_DNA), rna = make_alphabet(P7X_RNA);
  return type == P7X_AMINO ? amino : (type == P7X_RNA ? rna : dna);
}

// ---------------------------------------------------------------- Easel's vector expf, one lane
// (esl_sse_expf; fb_conversion applies it to every emission/transition score.)
float sse_expf(float x)
{
  static const float P0 = 1.9875691500E-4f, P1 = 1.3981999507E-3f, P2 = 8.3334519073E-3f,
                     P3 = 4.1665795894E-2f, P4 = 1.6666665459E-1f, P5 = 5.0000001201E-1f;
  static const float C1 = 0.693359375f, C2 = -2.12194440e-4f;
  const bool over = x > 88.72283905206835f, under = x <= -103.27892990343185f;
  if (std::isnan(x)) return x;
  if (over)  return INFINITY;
  if (under) return 0.0f;
  volatile float fx = x * (float) kLog2R;   // volatile: one IEEE rounding per operation, no contraction
  fx = fx + 0.5f;
  volatile float fl = (float) (int) fx;
  if (fl > fx) fl = fl - 1.0f;
  const int k = (int) fl;
  volatile float r = x;
  volatile float t1 = fl * C1, t2 = fl * C2;
  r = r - t1;
  r = r - t2;
  volatile float z = r * r;
  volatile float y = P0;
  y = y * r; y = y + P1;
  y = y * r; y = y + P2;
  y = y * r; y = y + P3;
  y = y * r; y = y + P4;
  y = y * r; y = y + P5;
  y = y * z;
  y = y + r;
  y = y + 1.0f;
  union { int32_t i; float f; } two_k;
  two_k.i = (int32_t) ((uint32_t) (k + 127) << 23);
  y = y * two_k.f;
  return y;
}

uint8_t unbiased_byteify(float scale_b, float sc)
{
  sc = -1.0f * roundf(scale_b * sc);
  return (sc > 255.) ? 255 : (uint8_t) (int) sc;
}
static uint8_t biased_byteify(float scale_b, int bias_b, float sc)
{
  sc = -1.0f * roundf(scale_b * sc);
  return (sc > 255 - bias_b) ? 255 : (uint8_t) ((int) sc + bias_b);
}
int16_t wordify(float scale_w, float sc)
{
  sc = roundf(scale_w * sc);
  if (sc >= 32767.0) return 32767;
  if (sc <= -32768.0) return -32768;
  return (int16_t) sc;
}

// ---------------------------------------------------------------- statistics (Easel gumbel / exponential)
double gumbel_surv(double x, double mu, double lambda)
{
  const double y = lambda * (x - mu), ey = -std::exp(-y);
  return (std::fabs(ey) < 5e-9) ? -ey : 1 - std::exp(ey);
}
double exp_surv(double x, double mu, double lambda)    { return x < mu ? 1.0 : std::exp(-lambda * (x - mu)); }
double exp_logsurv(double x, double mu, double lambda) { return x < mu ? 0.0 : -lambda * (x - mu); }

float null1_score(int L)
{ // p7_bg_SetLength + p7_bg_NullOne (p7_bg.pxd:10-30; plan7.pyx:6435)
  const float p1 = (float) L / (float) (L + 1);
  return (float) L * std::log((double) p1) + std::log(1. - p1);     // C's log(): double, whatever the argument
}

static float g_flogsum[16000];
static std::once_flag g_flogsum_once;
void flogsum_init()
{
  std::call_once(g_flogsum_once, [] {
    for (int i = 0; i < 16000; ++i) g_flogsum[i] = std::log(1. + std::exp((double) -i / 1000.0));
  });
}
float flogsum(float a, float b)
{ // upstream logsum.c p7_FLogsum: table lookup, 1/1000 nat resolution
  const float mx = a > b ? a : b, mn = a > b ? b : a;
  return (mn == -INFINITY || (mx - mn) >= 15.7f) ? mx : mx + g_flogsum[(int) ((mx - mn) * 1000.0f)];
}

// ---------------------------------------------------------------- query preparation
static void configure_generic(Profile &p, const p7x_hmm_view &h)
{
  const Alphabet &abc = Alphabet::get(p.abc_type);
  const int M = p.M, K = p.K, Kp = p.Kp;
  const float *t = h.t;
  p.tsc.assign((size_t) (M + 1) * 8, -INFINITY);
  p.msc.assign((size_t) Kp * (M + 1), -INFINITY);
  // local entry: occ[k] / sum_j occ[j] (M-j+1)   (p7_hmm_CalculateOccupancy + p7_ProfileConfig)
  std::vector<float> occ(M + 1);
  occ[0] = 0.f;
  occ[1] = t[0 * 7 + 1] + t[0 * 7 + 0];
  for (int k = 2; k <= M; ++k)
    occ[k] = occ[k - 1] * (t[(k - 1) * 7 + 0] + t[(k - 1) * 7 + 1]) + (1.0 - occ[k - 1]) * t[(k - 1) * 7 + 5];
  float Z = 0.f;
  for (int k = 1; k <= M; ++k) Z += occ[k] * (float) (M - k + 1);
  for (int k = 1; k <= M; ++k) p.tsc[(size_t) (k - 1) * 8 + gBM] = std::log((double) (occ[k] / Z));
  // multihit: E->J = E->C = 1/2
  p.xsc[XE][MOVE] = -kLog2;
  p.xsc[XE][LOOP] = -kLog2;
  p.nj = 1.0f;
  for (int k = 1; k < M; ++k) {
    float *g = &p.tsc[(size_t) k * 8];
    const float *tk = t + (size_t) k * 7;   // MM,MI,MD,IM,II,DM,DD
    // (upstream is C: log() takes and returns double; std::log(float) would be logf)
    g[gMM] = std::log((double) tk[0]); g[gMI] = std::log((double) tk[1]); g[gMD] = std::log((double) tk[2]);
    g[gIM] = std::log((double) tk[3]); g[gII] = std::log((double) tk[4]);
    g[gDM] = std::log((double) tk[5]); g[gDD] = std::log((double) tk[6]);
  }
  for (int k = 1; k <= M; ++k) {
    float sc[MAXKP];
    for (int x = 0; x < K; ++x) sc[x] = std::log((double) h.mat[(size_t) k * K + x] / p.bgf[x]);
    sc[K] = sc[Kp - 2] = sc[Kp - 1] = -INFINITY;
    for (int x = K + 1; x <= Kp - 3; ++x) {   // esl_abc_FExpectScVec
      float num = 0.f, den = 0.f;
      for (int y = 0; y < K; ++y) if (abc.degen[x][y]) { num += sc[y] * p.bgf[y]; den += p.bgf[y]; }
      sc[x] = num / den;
    }
    for (int x = 0; x < Kp; ++x) p.msc[(size_t) x * (M + 1) + k] = sc[x];
  }
  // length model (p7_ReconfigLength)
  const float pmove = (2.0f + p.nj) / ((float) p.L + 2.0f + p.nj), ploop = 1.0f - pmove;
  p.xsc[XN][LOOP] = p.xsc[XC][LOOP] = p.xsc[XJ][LOOP] = std::log((double) ploop);
  p.xsc[XN][MOVE] = p.xsc[XC][MOVE] = p.xsc[XJ][MOVE] = std::log((double) pmove);
}

static void convert_msv(Profile &p)
{
  const int M = p.M;
  float mx = 0.0f;   // rsc also holds the insert scores (0), so the maximum is never negative
  for (int x = 0; x < p.K; ++x)
    for (int k = 1; k <= M; ++k) mx = std::fmax(mx, p.msc[(size_t) x * (M + 1) + k]);
  p.scale_b = 3.0 / kLog2;
  p.base_b = 190;
  p.bias_b = unbiased_byteify(p.scale_b, -1.0 * mx);
  p.rb.assign((size_t) p.Kp * (M + 1), 255);
  for (int x = 0; x < p.Kp; ++x)
    for (int k = 1; k <= M; ++k)
      p.rb[(size_t) x * (M + 1) + k] = biased_byteify(p.scale_b, p.bias_b, p.msc[(size_t) x * (M + 1) + k]);
  p.tbm_b = unbiased_byteify(p.scale_b, logf(2.0f / ((float) M * (float) (M + 1))));
  p.tec_b = unbiased_byteify(p.scale_b, logf(0.5f));
  p.tjb_b = unbiased_byteify(p.scale_b, logf(3.0f / (float) (p.L + 3)));
}

static void convert_viterbi(Profile &p)
{
  const int M = p.M;
  p.scale_w = 500.0 / kLog2;
  p.base_w = 12000;
  p.rw.assign((size_t) p.Kp * (M + 1), -32768);
  for (int x = 0; x < p.Kp; ++x)
    for (int k = 1; k <= M; ++k) p.rw[(size_t) x * (M + 1) + k] = wordify(p.scale_w, p.msc[(size_t) x * (M + 1) + k]);
  p.tw.assign((size_t) NTRANS * (M + 1), -32768);
  auto W = [&](int t, int k) -> int16_t & { return p.tw[(size_t) t * (M + 1) + k]; };
  auto cap = [](int16_t v, int16_t maxval) { return v <= maxval ? v : maxval; };
  for (int k = 1; k <= M; ++k) {
    // entering node k: generic node k-1 (tsc[0] carries only B->M1)
    W(tBM, k) = cap(wordify(p.scale_w, p.tsc[(size_t) (k - 1) * 8 + gBM]), 0);
    W(tMM, k) = cap(wordify(p.scale_w, p.tsc[(size_t) (k - 1) * 8 + gMM]), 0);
    W(tIM, k) = cap(wordify(p.scale_w, p.tsc[(size_t) (k - 1) * 8 + gIM]), 0);
    W(tDM, k) = cap(wordify(p.scale_w, p.tsc[(size_t) (k - 1) * 8 + gDM]), 0);
    if (k < M) {   // leaving node k
      W(tMD, k) = cap(wordify(p.scale_w, p.tsc[(size_t) k * 8 + gMD]), 0);
      W(tMI, k) = cap(wordify(p.scale_w, p.tsc[(size_t) k * 8 + gMI]), 0);
      W(tII, k) = cap(wordify(p.scale_w, p.tsc[(size_t) k * 8 + gII]), -1);   // never a zero-cost I->I
      W(tDD, k) = wordify(p.scale_w, p.tsc[(size_t) k * 8 + gDD]);
    }
  }
  p.xw[XE][LOOP] = wordify(p.scale_w, p.xsc[XE][LOOP]);
  p.xw[XE][MOVE] = wordify(p.scale_w, p.xsc[XE][MOVE]);
  for (int s : {XN, XC, XJ}) { p.xw[s][MOVE] = wordify(p.scale_w, p.xsc[s][MOVE]); p.xw[s][LOOP] = 0; }
  p.ncj_roundoff = 0.0f;
  int bound = -32768;
  for (int k = 2; k < M - 1; ++k) {
    int dd = (int) wordify(p.scale_w, p.tsc[(size_t) k * 8 + gDD]);
    dd += (int) wordify(p.scale_w, p.tsc[(size_t) (k + 1) * 8 + gDM]);
    dd -= (int) wordify(p.scale_w, p.tsc[(size_t) (k + 1) * 8 + gBM]);
    if (dd > bound) bound = dd;
  }
  p.ddbound_w = (int16_t) bound;
}

static void convert_forward(Profile &p)
{
  const int M = p.M;
  p.rf_.assign((size_t) p.Kp * (M + 1), 0.0f);
  for (int x = 0; x < p.Kp; ++x)
    for (int k = 1; k <= M; ++k) p.rf_[(size_t) x * (M + 1) + k] = sse_expf(p.msc[(size_t) x * (M + 1) + k]);
  p.tf.assign((size_t) NTRANS * (M + 1), 0.0f);
  auto F = [&](int t, int k) -> float & { return p.tf[(size_t) t * (M + 1) + k]; };
  for (int k = 1; k <= M; ++k) {
    F(tBM, k) = sse_expf(p.tsc[(size_t) (k - 1) * 8 + gBM]);
    F(tMM, k) = sse_expf(p.tsc[(size_t) (k - 1) * 8 + gMM]);
    F(tIM, k) = sse_expf(p.tsc[(size_t) (k - 1) * 8 + gIM]);
    F(tDM, k) = sse_expf(p.tsc[(size_t) (k - 1) * 8 + gDM]);
    if (k < M) {
      F(tMD, k) = sse_expf(p.tsc[(size_t) k * 8 + gMD]);
      F(tMI, k) = sse_expf(p.tsc[(size_t) k * 8 + gMI]);
      F(tII, k) = sse_expf(p.tsc[(size_t) k * 8 + gII]);
      F(tDD, k) = sse_expf(p.tsc[(size_t) k * 8 + gDD]);
    }
  }
  for (int s = 0; s < 4; ++s) for (int m = 0; m < 2; ++m) p.xf[s][m] = expf(p.xsc[s][m]);
}

} // namespace p7x

using namespace p7x;

extern "C" {

int p7x_abi_version(void) { return P7X_ABI_VERSION; }

int p7x_debug_set_option(const char *name, int value)
{
  static const char *const names[p7x::OPT_COUNT] = { "small_block", "vit_wave", "msv_exact", "msv_long_groups", "msv_blocks_per_cu", "env_workspace_gb",
                                                     "device_clustered", "trace_finish", "trace_longtarget", "trace_envelope", "host_profile", "ssv_kernel", "msv_f16", "ens_lds_kb", "ens_fail", "msv_tiers", "stage_merge", "early_pack", "host_order", "vit_long_cut", "fwd_grouped", "region_guard_ppm", "msv_k8", "msv_lane_blocks" };
  if (!name) { p7x::set_error("p7x_debug_set_option: no name"); return P7X_EINVAL; }
  for (int i = 0; i < p7x::OPT_COUNT; ++i)
    if (std::strcmp(name, names[i]) == 0) { p7x::g_debug_opt[i].store(value); return P7X_OK; }
  p7x::set_error(std::string("p7x_debug_set_option: unknown option ") + name);
  return P7X_EINVAL;
}

const char *p7x_last_error(void) { return g_err.c_str(); }

void p7x_expf_neg(const double *in, float *out, size_t n)
{
  for (size_t i = 0; i < n; ++i) out[i] = std::isinf(in[i]) ? 0.0f : expf((float) (-1.0 * in[i]));
}

int p7x_hmm_max_length(const p7x_hmm_view *h, double beta, int32_t *out)
{
  if (!h || !out || h->M < 1 || !h->t || !(beta > 0.0)) { set_error("p7x_hmm_max_length: bad arguments"); return P7X_EINVAL; }
  const int M = h->M;
  if (M == 1) { *out = 1; return P7X_OK; }
  // column L of a table T[state][L] = P(the model, entered at M1, is in <state> having emitted exactly L residues);
  // two columns are kept.  The model emits an L-th residue with probability sum_k M[k][L] + I[k][L]: the first L for which
  // that falls below beta is the bound.  (Measuring what is still inside the model -- rather than one minus what has
  // left it -- keeps the bound right for transition rows that sum to 1 only to the five decimals of an HMM file.)
  const float *t = h->t;                       // [M+1][7]: MM MI MD IM II DM DD
  auto T = [&](int k, int x) -> double { return (double) t[(size_t) k * 7 + x]; };
  std::vector<double> Mm(2 * (size_t) (M + 1), 0.0), I(2 * (size_t) (M + 1), 0.0), D(2 * (size_t) (M + 1), 0.0);
  auto at = [&](std::vector<double> &v, int k, int c) -> double & { return v[(size_t) c * (M + 1) + k]; };
  at(Mm, 1, 0) = 1.0;
  for (int k = 2; k <= M; ++k) at(D, k, 0) = T(k - 1, 2) * at(Mm, k - 1, 0) + T(k - 1, 6) * at(D, k - 1, 0);
  const int length_bound = 200000;
  for (int L = 2; L <= length_bound; ++L) {
    const int c = (L - 1) & 1, pc = L & 1;
    at(Mm, 1, c) = 0.0; at(D, 1, c) = 0.0;
    at(I, 1, c) = T(1, 1) * at(Mm, 1, pc) + T(1, 4) * at(I, 1, pc);
    double alive = at(I, 1, c);
    for (int k = 2; k <= M; ++k) {
      const double m = T(k - 1, 0) * at(Mm, k - 1, pc) + T(k - 1, 3) * at(I, k - 1, pc) + T(k - 1, 5) * at(D, k - 1, pc);
      const double i = k < M ? T(k, 1) * at(Mm, k, pc) + T(k, 4) * at(I, k, pc) : 0.0;      // no insert state after the last node
      at(Mm, k, c) = m; at(I, k, c) = i;
      at(D, k, c) = T(k - 1, 2) * at(Mm, k - 1, c) + T(k - 1, 6) * at(D, k - 1, c);
      alive += m + i;
    }
    if (alive < beta) { *out = L; return P7X_OK; }
  }
  set_error("p7x_hmm_max_length: no bound below 200000 residues");
  return P7X_ERANGE;
}

int p7x_oprofile_create(const p7x_hmm_view *h, const float *bg_f, int32_t L, p7x_oprofile **out)
{
  if (!h || !bg_f || !out || h->M < 1 || !h->t || !h->mat || !h->name) { set_error("p7x_oprofile_create: bad arguments"); return P7X_EINVAL; }
  if (h->abc_type != P7X_AMINO && h->abc_type != P7X_DNA && h->abc_type != P7X_RNA) { set_error("unknown alphabet"); return P7X_EINVAL; }
  flogsum_init();
  auto *om = new p7x_oprofile();
  Profile &p = om->p;
  const Alphabet &abc = Alphabet::get(h->abc_type);
  p.M = h->M; p.K = abc.K; p.Kp = abc.Kp; p.abc_type = h->abc_type; p.L = L; p.max_length = h->max_length;
  p.mode = 1;                                  // p7_LOCAL: multihit local alignment, the only mode the pipeline configures
  p.name = h->name;
  if (h->acc)  { p.acc = h->acc;   p.has_acc = true; }
  if (h->desc) { p.desc = h->desc; p.has_desc = true; }
  if (h->consensus) p.consensus = h->consensus;
  if (h->rf) p.rf = h->rf;
  if (h->mm) p.mm = h->mm;
  if (h->cs) p.cs = h->cs;
  std::memcpy(p.evparam, h->evparam, sizeof(p.evparam));
  std::memcpy(p.cutoff, h->cutoff, sizeof(p.cutoff));
  std::memset(p.compo, 0, sizeof(p.compo));
  if (h->compo) std::memcpy(p.compo, h->compo, sizeof(float) * p.K);
  std::memset(p.bgf, 0, sizeof(p.bgf));
  std::memcpy(p.bgf, bg_f, sizeof(float) * p.K);
  configure_generic(p, *h);
  convert_msv(p);
  convert_viterbi(p);
  convert_forward(p);
  attach_dev_cache(om);
  *out = om;
  return P7X_OK;
}

int p7x_oprofile_get_info(const p7x_oprofile *om, p7x_oprofile_info *o)
{
  if (!om || !o) return P7X_EINVAL;
  const Profile &p = om->p;
  std::memset(o, 0, sizeof(*o));
  o->M = p.M; o->K = p.K; o->Kp = p.Kp; o->abc_type = p.abc_type; o->L = p.L; o->max_length = p.max_length; o->mode = p.mode;
  o->Q16 = p.Q16(); o->Q8 = p.Q8(); o->Q4 = p.Q4();
  o->tbm_b = p.tbm_b; o->tec_b = p.tec_b; o->tjb_b = p.tjb_b; o->base_b = p.base_b; o->bias_b = p.bias_b; o->scale_b = p.scale_b;
  std::memcpy(o->xw, p.xw, sizeof(p.xw));
  o->scale_w = p.scale_w; o->base_w = p.base_w; o->ddbound_w = p.ddbound_w; o->ncj_roundoff = p.ncj_roundoff;
  std::memcpy(o->xf, p.xf, sizeof(p.xf));
  std::memcpy(o->evparam, p.evparam, sizeof(p.evparam));
  std::memcpy(o->cutoff, p.cutoff, sizeof(p.cutoff));
  std::memcpy(o->compo, p.compo, sizeof(p.compo));
  o->nj = p.nj;
  return P7X_OK;
}

// Farrar striping: lane z of vector q holds node k = q + 1 + z*Q.
int64_t p7x_oprofile_striped(const p7x_oprofile *om, int which, void *out, size_t out_bytes)
{
  if (!om || !out) return -1;
  const Profile &p = om->p;
  const int M = p.M, Kp = p.Kp;
  auto node = [](int q, int z, int Q) { return q + 1 + z * Q; };
  switch (which) {
  case 0: {  // rbv
    const int Q = p.Q16(); const size_t need = (size_t) Kp * Q * 16;
    if (out_bytes < need) return -1;
    auto *o = (uint8_t *) out;
    for (int x = 0; x < Kp; ++x) for (int q = 0; q < Q; ++q) for (int z = 0; z < 16; ++z) {
      const int k = node(q, z, Q);
      o[((size_t) x * Q + q) * 16 + z] = k <= M ? p.rb[(size_t) x * (M + 1) + k] : 255;
    }
    return (int64_t) need;
  }
  case 1: {  // sbv = signed (rbv - bias), saturating; 17 extra wrap-around vectors
    const int Q = p.Q16(), QS = Q + 17; const size_t need = (size_t) Kp * QS * 16;
    if (out_bytes < need) return -1;
    auto *o = (int8_t *) out;
    const uint8_t top = (uint8_t) (p.bias_b + 127);
    for (int x = 0; x < Kp; ++x) for (int q = 0; q < QS; ++q) for (int z = 0; z < 16; ++z) {
      const int k = node(q % Q, z, Q);
      const uint8_t r = k <= M ? p.rb[(size_t) x * (M + 1) + k] : 255;
      const uint8_t d = top > r ? (uint8_t) (top - r) : 0;
      o[((size_t) x * QS + q) * 16 + z] = (int8_t) (d ^ 127);
    }
    return (int64_t) need;
  }
  case 2: {  // rwv
    const int Q = p.Q8(); const size_t need = (size_t) Kp * Q * 8 * 2;
    if (out_bytes < need) return -1;
    auto *o = (int16_t *) out;
    for (int x = 0; x < Kp; ++x) for (int q = 0; q < Q; ++q) for (int z = 0; z < 8; ++z) {
      const int k = node(q, z, Q);
      o[((size_t) x * Q + q) * 8 + z] = k <= M ? p.rw[(size_t) x * (M + 1) + k] : -32768;
    }
    return (int64_t) need;
  }
  case 3: {  // twv: per q the 7 vectors BM,MM,IM,DM,MD,MI,II; then all DD vectors
    const int Q = p.Q8(); const size_t need = (size_t) 8 * Q * 8 * 2;
    if (out_bytes < need) return -1;
    auto *o = (int16_t *) out;
    for (int q = 0; q < Q; ++q) for (int z = 0; z < 8; ++z) {
      const int k = node(q, z, Q);
      for (int t = tBM; t <= tII; ++t) {
        int16_t v = k <= M ? p.tw[(size_t) t * (M + 1) + k] : -32768;
        if (k > M) v = (t == tII) ? -32768 : -32768;
        o[((size_t) q * 7 + t) * 8 + z] = v;
      }
      o[((size_t) 7 * Q + q) * 8 + z] = k <= M ? p.tw[(size_t) tDD * (M + 1) + k] : -32768;
    }
    return (int64_t) need;
  }
  case 4: {  // rfv
    const int Q = p.Q4(); const size_t need = (size_t) Kp * Q * 4 * 4;
    if (out_bytes < need) return -1;
    auto *o = (float *) out;
    for (int x = 0; x < Kp; ++x) for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) {
      const int k = node(q, z, Q);
      o[((size_t) x * Q + q) * 4 + z] = k <= M ? p.rf_[(size_t) x * (M + 1) + k] : 0.0f;
    }
    return (int64_t) need;
  }
  case 5: {  // tfv
    const int Q = p.Q4(); const size_t need = (size_t) 8 * Q * 4 * 4;
    if (out_bytes < need) return -1;
    auto *o = (float *) out;
    for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) {
      const int k = node(q, z, Q);
      for (int t = tBM; t <= tII; ++t) o[((size_t) q * 7 + t) * 4 + z] = k <= M ? p.tf[(size_t) t * (M + 1) + k] : 0.0f;
      o[((size_t) 7 * Q + q) * 4 + z] = k <= M ? p.tf[(size_t) tDD * (M + 1) + k] : 0.0f;
    }
    return (int64_t) need;
  }
  default: return -1;
  }
}


} // extern "C"

// ---------------------------------------------------------------- pressed profiles (.h3f / .h3p)
// Record layouts of upstream's p7_oprofile_Write (impl_sse/io.c; reference impl_sse/io.pxd:11-16, magics in
// patches/p7_hmmfile.c.patch:15-20), restated in SURVEY.md section 8(f) and checked byte for byte against the
// reference's pressed fixtures (tests/test_host.py).  Little-endian, as HMMER writes on x86.
namespace {
constexpr uint32_t kMagicF = 0xb3e6e6f3u, kMagicP = 0xb3e6f0f3u;

struct Writer {
  uint8_t *buf; size_t cap; size_t n = 0;
  void bytes(const void *p, size_t len) { if (buf && n + len <= cap) std::memcpy(buf + n, p, len); n += len; }
  template <typename T> void pod(T v) { bytes(&v, sizeof(T)); }
  void str_f(const std::string &s) { pod<int32_t>((int32_t) s.size()); bytes(s.c_str(), s.size() + 1); }          // .h3f name: always written
  void str_p(const std::string &s, bool have) { if (!have || s.empty()) { pod<int32_t>(0); return; } str_f(s); }   // .h3p: n == 0 means absent
  void line(const std::string &s, int M) { std::vector<char> l((size_t) M + 2, 0); std::memcpy(l.data(), s.data(), std::min(s.size(), (size_t) M + 1)); bytes(l.data(), l.size()); }
};
struct Rd {
  const uint8_t *buf; size_t len; size_t n = 0; bool ok = true;
  void bytes(void *p, size_t k) { if (n + k > len) { ok = false; return; } std::memcpy(p, buf + n, k); n += k; }
  template <typename T> T pod() { T v{}; bytes(&v, sizeof(T)); return v; }
  std::string str(int32_t k) { std::string s; if (k < 0 || n + (size_t) k + 1 > len) { ok = false; return s; } s.assign((const char *) buf + n, (size_t) k); n += (size_t) k + 1; return s; }
  std::string line(int M) { std::string s; if (n + (size_t) M + 2 > len) { ok = false; return s; } s.assign((const char *) buf + n, strnlen((const char *) buf + n, (size_t) M + 2)); n += (size_t) M + 2; return s; }
};

void write_pressed(const Profile &p, const int64_t offs[3], Writer &f, Writer &q, const p7x_oprofile *om)
{
  const int M = p.M, Kp = p.Kp;
  std::vector<uint8_t> tmp;
  auto striped = [&](int which, size_t bytes) { tmp.resize(bytes); p7x_oprofile_striped(om, which, tmp.data(), bytes); return tmp.data(); };
  // ---- MSV part
  f.pod(kMagicF); f.pod<int32_t>(M); f.pod<int32_t>(p.abc_type);
  f.str_f(p.name);
  f.pod<int32_t>(p.max_length);
  f.pod(p.tbm_b); f.pod(p.tec_b); f.pod(p.tjb_b); f.pod(p.scale_b); f.pod(p.base_b); f.pod(p.bias_b);
  { const size_t n = (size_t) Kp * (p.Q16() + 17) * 16; f.bytes(striped(1, n), n); }
  { const size_t n = (size_t) Kp * p.Q16() * 16; f.bytes(striped(0, n), n); }
  f.bytes(p.evparam, sizeof(p.evparam));
  f.bytes(offs, 3 * sizeof(int64_t));
  { float compo[20]; std::memset(compo, 0, sizeof(compo)); std::memcpy(compo, p.compo, sizeof(float) * std::min(20, (int) MAXK)); f.bytes(compo, sizeof(compo)); }
  f.pod(kMagicF);
  // ---- the rest
  q.pod(kMagicP); q.pod<int32_t>(M); q.pod<int32_t>(p.abc_type);
  q.str_p(p.name, true); q.str_p(p.acc, p.has_acc); q.str_p(p.desc, p.has_desc);
  q.line(p.rf, M); q.line(p.mm, M); q.line(p.cs, M); q.line(p.consensus, M);
  { const size_t n = (size_t) 8 * p.Q8() * 8 * 2; q.bytes(striped(3, n), n); }
  { const size_t n = (size_t) Kp * p.Q8() * 8 * 2; q.bytes(striped(2, n), n); }
  q.bytes(p.xw, sizeof(p.xw));
  q.pod(p.scale_w); q.pod(p.base_w); q.pod(p.ddbound_w); q.pod(p.ncj_roundoff);
  { const size_t n = (size_t) 8 * p.Q4() * 4 * 4; q.bytes(striped(5, n), n); }
  { const size_t n = (size_t) Kp * p.Q4() * 4 * 4; q.bytes(striped(4, n), n); }
  q.bytes(p.xf, sizeof(p.xf));
  q.bytes(p.cutoff, sizeof(p.cutoff));
  q.pod(p.nj); q.pod<int32_t>(p.mode); q.pod<int32_t>(p.L);
  q.pod(kMagicP);
}
} // namespace

extern "C" {

// FASTA text -> the packed block of the C-ABI (255 x1..xL 255 ...) in one pass over the buffer, at memory speed.
// lut[c]: digital code of character c, 255 = illegal, 254 = ignored (white space, digits).  Call once with
// dsq == NULL to size the outputs (nseq, nres, bytes of the string table), then again to fill them.
// strtab receives, per record, the NUL-terminated name (first word of the header line) followed by the
// NUL-terminated description (rest of the line, trimmed); name_off / desc_off index it.
int p7x_fasta_parse(const char *text, size_t n, const uint8_t *lut, size_t *nseq, size_t *nres, size_t *strbytes,
                    uint8_t *dsq, int64_t *offsets, int32_t *lengths, char *strtab, int64_t *name_off, int64_t *desc_off,
                    size_t *bad_pos)
{
  if (!text || !lut || !nseq || !nres || !strbytes) { set_error("p7x_fasta_parse: bad arguments"); return P7X_EINVAL; }
  const bool fill = dsq != nullptr;
  if (fill && (!offsets || !lengths || !strtab || !name_off || !desc_off)) { set_error("p7x_fasta_parse: bad arguments"); return P7X_EINVAL; }
  size_t ns = 0, nr = 0, sb = 0, pos = 0, w = 0;
  if (fill) dsq[w] = 255;
  ++w;
  while (pos < n && text[pos] != '>') {                       // anything before the first record must be blank
    const unsigned char c = (unsigned char) text[pos];
    if (c != ' ' && c != '\t' && c != '\r' && c != '\n') { if (bad_pos) *bad_pos = pos; set_error("FASTA text does not start with '>'"); return P7X_EFORMAT; }
    ++pos;
  }
  while (pos < n) {
    // header line
    size_t h0 = pos + 1, h1 = h0;
    while (h1 < n && text[h1] != '\n') ++h1;
    size_t he = h1;
    while (he > h0 && (text[he - 1] == '\r' || text[he - 1] == ' ' || text[he - 1] == '\t')) --he;
    size_t a = h0;
    while (a < he && (text[a] == ' ' || text[a] == '\t')) ++a;
    size_t b = a;
    while (b < he && text[b] != ' ' && text[b] != '\t') ++b;
    size_t d = b;
    while (d < he && (text[d] == ' ' || text[d] == '\t')) ++d;
    if (fill) {
      name_off[ns] = (int64_t) sb; std::memcpy(strtab + sb, text + a, b - a); strtab[sb + (b - a)] = 0;
      desc_off[ns] = (int64_t) (sb + (b - a) + 1); std::memcpy(strtab + sb + (b - a) + 1, text + d, he - d); strtab[sb + (b - a) + 1 + (he - d)] = 0;
    }
    sb += (b - a) + 1 + (he - d) + 1;
    pos = h1 < n ? h1 + 1 : n;
    // residues up to the next '>' at the start of a line
    const size_t start = w;
    bool bol = true;
    while (pos < n) {
      const unsigned char c = (unsigned char) text[pos];
      if (bol && c == '>') break;
      if (c == '\n') { bol = true; ++pos; continue; }
      bol = false;
      const uint8_t code = lut[c];
      if (code < 254) { if (fill) dsq[w] = code; ++w; }
      else if (code == 255) { if (bad_pos) *bad_pos = pos; set_error(std::string("invalid symbol '") + (char) c + "' in sequence data"); return P7X_EFORMAT; }
      ++pos;
    }
    const size_t L = w - start;
    if (L > 0x7fffffff) { set_error("sequence too long"); return P7X_ERANGE; }
    if (fill) { offsets[ns] = (int64_t) start; lengths[ns] = (int32_t) L; dsq[w] = 255; }
    ++w;
    nr += L; ++ns;
  }
  *nseq = ns; *nres = nr; *strbytes = sb;
  return P7X_OK;
}

int p7x_oprofile_get_string(const p7x_oprofile *om, int which, char *buf, size_t n)
{
  if (!om || !buf || n == 0) return -1;
  const Profile &p = om->p;
  const std::string *s = nullptr;
  switch (which) {
    case 0: s = &p.name; break;
    case 1: if (p.has_acc) s = &p.acc; break;
    case 2: if (p.has_desc) s = &p.desc; break;
    case 3: s = &p.consensus; break;
    default: return -1;
  }
  buf[0] = 0;
  if (!s || s->empty()) return 0;
  std::snprintf(buf, n, "%s", s->c_str());
  return (int) s->size();
}

int p7x_oprofile_write_pressed(const p7x_oprofile *om, const int64_t offs[3], uint8_t *h3f, size_t cap_f, size_t *len_f,
                               uint8_t *h3p, size_t cap_p, size_t *len_p)
{
  if (!om || !len_f || !len_p) { set_error("p7x_oprofile_write_pressed: bad arguments"); return P7X_EINVAL; }
  const int64_t zero[3] = {0, 0, 0};
  Writer f{ h3f, h3f ? cap_f : 0 }, q{ h3p, h3p ? cap_p : 0 };
  write_pressed(om->p, offs ? offs : zero, f, q, om);
  *len_f = f.n; *len_p = q.n;
  if ((h3f && f.n > cap_f) || (h3p && q.n > cap_p)) { set_error("p7x_oprofile_write_pressed: buffer too small"); return P7X_EINVAL; }
  return P7X_OK;
}

int p7x_oprofile_read_pressed(const uint8_t *h3f, size_t nf, const uint8_t *h3p, size_t np, const float *bg_f,
                              p7x_oprofile **out, size_t *used_f, size_t *used_p, int64_t offs[3])
{
  if (!h3f || !h3p || !bg_f || !out) { set_error("p7x_oprofile_read_pressed: bad arguments"); return P7X_EINVAL; }
  Rd f{ h3f, nf }, q{ h3p, np };
  auto om = std::make_unique<p7x_oprofile>();
  Profile &p = om->p;
  auto bad = [&](const char *what) { set_error(std::string("pressed profile: ") + what); return P7X_EFORMAT; };
  // ---- .h3f
  if (f.pod<uint32_t>() != kMagicF || !f.ok) return bad("bad magic in the MSV filter file (.h3f)");
  p.M = f.pod<int32_t>(); p.abc_type = f.pod<int32_t>();
  if (!f.ok || p.M < 1 || p.M > 100000 || (p.abc_type != P7X_AMINO && p.abc_type != P7X_DNA && p.abc_type != P7X_RNA)) return bad("bad model header (.h3f)");
  const Alphabet &abc = Alphabet::get(p.abc_type);
  p.K = abc.K; p.Kp = abc.Kp;
  const int M = p.M, Kp = p.Kp;
  p.name = f.str(f.pod<int32_t>());
  p.max_length = f.pod<int32_t>();
  p.tbm_b = f.pod<uint8_t>(); p.tec_b = f.pod<uint8_t>(); p.tjb_b = f.pod<uint8_t>(); p.scale_b = f.pod<float>();
  p.base_b = f.pod<uint8_t>(); p.bias_b = f.pod<uint8_t>();
  const int Q16 = p.Q16(), Q8 = p.Q8(), Q4 = p.Q4();
  { std::vector<int8_t> sbv((size_t) Kp * (Q16 + 17) * 16); f.bytes(sbv.data(), sbv.size()); }      // derived from rbv: not kept
  {
    std::vector<uint8_t> rbv((size_t) Kp * Q16 * 16); f.bytes(rbv.data(), rbv.size());
    p.rb.assign((size_t) Kp * (M + 1), 255);
    if (f.ok)
      for (int x = 0; x < Kp; ++x) for (int q2 = 0; q2 < Q16; ++q2) for (int z = 0; z < 16; ++z) {
        const int k = q2 + 1 + z * Q16;
        if (k <= M) p.rb[(size_t) x * (M + 1) + k] = rbv[((size_t) x * Q16 + q2) * 16 + z];
      }
  }
  f.bytes(p.evparam, sizeof(p.evparam));
  int64_t o3[3] = {0, 0, 0}; f.bytes(o3, sizeof(o3));
  { float compo[20]; f.bytes(compo, sizeof(compo)); std::memset(p.compo, 0, sizeof(p.compo)); std::memcpy(p.compo, compo, sizeof(float) * std::min(20, (int) MAXK)); }
  if (f.pod<uint32_t>() != kMagicF || !f.ok) return bad("truncated or corrupt record (.h3f)");
  // ---- .h3p
  if (q.pod<uint32_t>() != kMagicP || !q.ok) return bad("bad magic in the profile file (.h3p)");
  if (q.pod<int32_t>() != M || q.pod<int32_t>() != p.abc_type || !q.ok) return bad(".h3f and .h3p records disagree");
  { const int32_t n = q.pod<int32_t>(); const std::string nm = n > 0 ? q.str(n) : std::string(); if (nm != p.name) return bad(".h3f and .h3p records carry different names"); }
  { const int32_t n = q.pod<int32_t>(); if (n > 0) { p.acc = q.str(n); p.has_acc = true; } }
  { const int32_t n = q.pod<int32_t>(); if (n > 0) { p.desc = q.str(n); p.has_desc = true; } }
  p.rf = q.line(M); p.mm = q.line(M); p.cs = q.line(M); p.consensus = q.line(M);
  {
    std::vector<int16_t> twv((size_t) 8 * Q8 * 8), rwv((size_t) Kp * Q8 * 8);
    q.bytes(twv.data(), twv.size() * 2); q.bytes(rwv.data(), rwv.size() * 2);
    p.tw.assign((size_t) NTRANS * (M + 1), -32768); p.rw.assign((size_t) Kp * (M + 1), -32768);
    if (q.ok)
      for (int q2 = 0; q2 < Q8; ++q2) for (int z = 0; z < 8; ++z) {
        const int k = q2 + 1 + z * Q8;
        if (k > M) continue;
        for (int t = tBM; t <= tII; ++t) p.tw[(size_t) t * (M + 1) + k] = twv[((size_t) q2 * 7 + t) * 8 + z];
        p.tw[(size_t) tDD * (M + 1) + k] = twv[((size_t) 7 * Q8 + q2) * 8 + z];
        for (int x = 0; x < Kp; ++x) p.rw[(size_t) x * (M + 1) + k] = rwv[((size_t) x * Q8 + q2) * 8 + z];
      }
  }
  q.bytes(p.xw, sizeof(p.xw));
  p.scale_w = q.pod<float>(); p.base_w = q.pod<int16_t>(); p.ddbound_w = q.pod<int16_t>(); p.ncj_roundoff = q.pod<float>();
  {
    std::vector<float> tfv((size_t) 8 * Q4 * 4), rfv((size_t) Kp * Q4 * 4);
    q.bytes(tfv.data(), tfv.size() * 4); q.bytes(rfv.data(), rfv.size() * 4);
    p.tf.assign((size_t) NTRANS * (M + 1), 0.0f); p.rf_.assign((size_t) Kp * (M + 1), 0.0f);
    if (q.ok)
      for (int q2 = 0; q2 < Q4; ++q2) for (int z = 0; z < 4; ++z) {
        const int k = q2 + 1 + z * Q4;
        if (k > M) continue;
        for (int t = tBM; t <= tII; ++t) p.tf[(size_t) t * (M + 1) + k] = tfv[((size_t) q2 * 7 + t) * 4 + z];
        p.tf[(size_t) tDD * (M + 1) + k] = tfv[((size_t) 7 * Q4 + q2) * 4 + z];
        for (int x = 0; x < Kp; ++x) p.rf_[(size_t) x * (M + 1) + k] = rfv[((size_t) x * Q4 + q2) * 4 + z];
      }
  }
  q.bytes(p.xf, sizeof(p.xf));
  q.bytes(p.cutoff, sizeof(p.cutoff));
  p.nj = q.pod<float>(); p.mode = q.pod<int32_t>(); p.L = q.pod<int32_t>();
  if (q.pod<uint32_t>() != kMagicP || !q.ok) return bad("truncated or corrupt record (.h3p)");
  std::memset(p.bgf, 0, sizeof(p.bgf));
  std::memcpy(p.bgf, bg_f, sizeof(float) * p.K);
  flogsum_init();
  attach_dev_cache(om.get());
  if (used_f) *used_f = f.n;
  if (used_p) *used_p = q.n;
  if (offs) std::memcpy(offs, o3, sizeof(o3));
  *out = om.release();
  return P7X_OK;
}

void p7x_pipeline_cfg_default(p7x_pipeline_cfg *c)
{ // p7_pipeline_Create(NULL, ...) defaults; pyhmmer plan7.pyx:5413-5421
  std::memset(c, 0, sizeof(*c));
  c->by_E = 1; c->E = 10.0; c->T = 0.0; c->dom_by_E = 1; c->domE = 10.0; c->domT = 0.0; c->use_bit_cutoffs = 0;
  c->inc_by_E = 1; c->incE = 0.01; c->incT = 0.0; c->incdom_by_E = 1; c->incdomE = 0.01; c->incdomT = 0.0;
  c->Z = 0.0; c->domZ = 0.0; c->Z_setby = P7X_ZSETBY_NTARGETS; c->domZ_setby = P7X_ZSETBY_NTARGETS;
  c->F1 = 0.02; c->F2 = 1e-3; c->F3 = 1e-5;
  c->do_max = 0; c->do_biasfilter = 1; c->do_null2 = 1;
  c->seed = 42; c->mode = P7X_SEARCH_SEQS; c->host_threads = 0; c->host_envelopes = 0; c->host_regions = 0; c->host_ensembles = 0;
  c->long_targets = 0; c->strands = P7X_STRAND_BOTH; c->B1 = 100; c->B2 = 240; c->B3 = 1000;      // p7_pipeline_Create
  c->block_length = 0x40000; c->window_length = -1; c->evalue_window_length = -1; c->oa_guard = 4e-6f; c->ens_guard = 2.5e-7f; c->lt_part = 0; c->lt_nparts = 1;
  c->f3_guard = 4e-3f;
  c->lt_resident_key = 0;
}

} // extern "C"
