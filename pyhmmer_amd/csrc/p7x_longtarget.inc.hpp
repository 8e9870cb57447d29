// p7x_longtarget.inc.hpp -- host tail of p7_Pipeline_LongTarget (nhmmer), included at the end of p7x_domaindef.cpp
// (it drives that file's envelope machinery).
//
// Restates, from upstream HMMER 3.3/3.4 (absent from the reference checkout; entry points and structs in
// include/libhmmer/p7_pipeline.pxd:131-143, p7_scoredata.pxd:16-30, called from plan7.pyx:7541-7664):
//   p7_pipeline.c   p7_Pipeline_LongTarget, p7_pli_ExtendAndMergeWindows, p7_pli_postSSV_LongTarget,
//                   p7_pli_postViterbi_LongTarget and the per-domain hit construction
//   impl_sse/msvfilter.c  the bookkeeping of p7_SSVFilter_longtarget around a row that reaches the threshold
//   impl_sse/vitfilter.c  p7_ViterbiFilter_longtarget
//   p7_scoredata.c  p7_hmm_ScoreDataComputeRest (prefix / suffix lengths)
//   p7_tophits.c    p7_tophits_ComputeNhmmerEvalues, p7_tophits_RemoveDuplicates
// The SSV scan itself runs on the device (p7x_ssvlong.hip); everything here sees only the few windows it seeds.
// Parity status: restated from memory of upstream and pinned only by the reference's nhmmer fixtures
// (tests/golden/tables/bmyD{1,2}.tbl, the RF00001 known answers); see DESIGN.md.
#include <mutex>

namespace p7x {

namespace {

struct LtWindow { int64_t n = 0; int k = 0; int64_t length = 0; };     // first residue (1-based in the block), model node of the last cell, length

// ---------------------------------------------------------------- scalar filters on one window
struct LtLengthModel { uint8_t tjb_b; int16_t xw_move; float nullsc; };

static float lt_null1(int64_t L)
{
  const float p1 = (float) L / (float) (L + 1);
  return (float) L * std::log((double) p1) + std::log(1. - p1);        // p7_bg_NullOne
}

// p7_MSVFilter on dsq[1..L] with the length model of L (u8 arithmetic of impl_sse/msvfilter.c)
static float lt_msv(const Profile &p, const uint8_t *dsq, int64_t L)
{
  const int M = p.M;
  const int tjb = unbiased_byteify(p.scale_b, logf(3.0f / (float) (L + 3)));
  const int bias = p.bias_b, base = p.base_b, tjbm = tjb + p.tbm_b, tec = p.tec_b;
  std::vector<int> row(M + 1, 0), nxt(M + 1, 0);
  int xJ = 0, xB = std::max(base - tjbm, 0);
  for (int64_t i = 1; i <= L; ++i) {
    const uint8_t *rb = p.rb.data() + (size_t) dsq[i] * (M + 1);
    int xE = 0;
    for (int k = 1; k <= M; ++k) {
      int sv = std::max(row[k - 1], xB);
      sv = std::min(sv + bias, 255);
      sv = std::max(sv - (int) rb[k], 0);
      xE = std::max(xE, sv);
      nxt[k] = sv;
    }
    if (xE + bias >= 255) return INFINITY;
    xE = std::max(xE - tec, 0);
    xJ = std::max(xJ, xE);
    xB = std::max(std::max(base, xJ) - tjbm, 0);
    row.swap(nxt);
  }
  float sc = ((float) (xJ - tjb) - (float) p.base_b);
  sc /= p.scale_b;
  sc -= 3.0f;
  return sc;
}

// p7_bg_FilterScore: Forward score of the two-state composition HMM (esl_hmm_Forward), float arithmetic as upstream
static float lt_bias_filter(const Profile &p, const uint8_t *dsq, int64_t L)
{
  const Alphabet &abc = Alphabet::get(p.abc_type);
  float eo[MAXKP][2];
  for (int x = 0; x < p.Kp; ++x) { eo[x][0] = 1.0f; eo[x][1] = 1.0f; }
  for (int x = 0; x < p.K; ++x) { eo[x][0] = p.bgf[x] / p.bgf[x]; eo[x][1] = p.compo[x] / p.bgf[x]; }
  for (int x = p.K + 1; x <= p.Kp - 3; ++x)
    for (int s = 0; s < 2; ++s) {
      float e = 0.0f, den = 0.0f;
      for (int y = 0; y < p.K; ++y) if (abc.degen[x][y]) { e += (s == 0 ? p.bgf[y] : p.compo[y]); den += p.bgf[y]; }
      eo[x][s] = den > 0.0f ? e / den : 0.0f;
    }
  const float p1 = (float) L / (float) (L + 1);
  const float L1 = (float) ((double) (float) p.M / 8.0);
  const float t00 = p1, t01 = 1.0f - p1, t10 = 1.0f / (L1 + 1.0f), t11 = L1 / (L1 + 1.0f);
  float dp0 = eo[dsq[1]][0] * 0.999f, dp1 = eo[dsq[1]][1] * 0.001f;
  float mx = std::max(0.0f, std::max(dp0, dp1));
  dp0 /= mx; dp1 /= mx;
  float logsc = 0.0f;
  logsc += (float) std::log((double) mx);
  for (int64_t i = 2; i <= L; ++i) {
    const int x = dsq[i];
    float n0 = 0.0f; n0 += dp0 * t00; n0 += dp1 * t10; n0 *= eo[x][0];
    float n1 = 0.0f; n1 += dp0 * t01; n1 += dp1 * t11; n1 *= eo[x][1];
    mx = std::max(0.0f, std::max(n0, n1));
    dp0 = n0 / mx; dp1 = n1 / mx;
    logsc += (float) std::log((double) mx);
  }
  float last = 0.0f; last += dp0 * 1.0f; last += dp1 * 1.0f;
  logsc += (float) std::log((double) last);
  return logsc + (float) L * logf(p1) + logf((float) (1. - (double) p1));
}

static inline int16_t sat16(int v) { return (int16_t) std::max(-32768, std::min(32767, v)); }

// the row score that corresponds to P = F2 for a window of length L with bias-adjusted null score <filtersc>
static int lt_viterbi_threshold(const Profile &p, int64_t L, float filtersc, double F2)
{
  const int16_t xw_move = wordify(p.scale_w, logf(3.0f / (float) (L + 3)));
  const double invP = p.evparam[P7X_VMU] - std::log(-1.0 * std::log(1.0 - F2)) / p.evparam[P7X_VLAMBDA];     // esl_gumbel_invsurv
  return (int) std::ceil(((filtersc + (float) (kLog2 * invP) + 3.0) * p.scale_w) - (float) p.xw[XE][MOVE] - (float) xw_move + (float) p.base_w);
}

// p7_ViterbiFilter_longtarget: every row whose best match cell reaches the score that corresponds to P = F2 seeds a
// window (one per cell that holds that score) and clears the row.  Un-striped; the D->D path is evaluated in full,
// which gives the same M cells as upstream's lazy-F evaluation.
static void lt_viterbi_longtarget(const Profile &p, const uint8_t *dsq, int64_t L, float filtersc, double F2, std::vector<LtWindow> &out)
{
  const int M = p.M;
  const int16_t xw_move = wordify(p.scale_w, logf(3.0f / (float) (L + 3)));
  const int16_t xw_e_move = p.xw[XE][MOVE], xw_e_loop = p.xw[XE][LOOP];
  const int sc_thresh = lt_viterbi_threshold(p, L, filtersc, F2);
  auto tw = [&](int t, int k) -> int { return p.tw[(size_t) t * (M + 1) + k]; };
  std::vector<int16_t> mm(M + 2, -32768), im(M + 2, -32768), dm(M + 2, -32768), mn(M + 2), in_(M + 2), dn(M + 2);
  const int xN = p.base_w;
  int xB = sat16(xN + xw_move), xJ = -32768, xC = -32768;
  for (int64_t i = 1; i <= L; ++i) {
    const int16_t *rw = p.rw.data() + (size_t) dsq[i] * (M + 1);
    int xE = -32768;
    mn[0] = in_[0] = dn[0] = -32768;
    for (int k = 1; k <= M; ++k) {
      int sv = sat16(xB + tw(0, k));
      sv = std::max(sv, (int) sat16(mm[k - 1] + tw(1, k)));
      sv = std::max(sv, (int) sat16(im[k - 1] + tw(2, k)));
      sv = std::max(sv, (int) sat16(dm[k - 1] + tw(3, k)));
      sv = sat16(sv + rw[k]);
      mn[k] = (int16_t) sv;
      xE = std::max(xE, sv);
      in_[k] = (int16_t) std::max((int) sat16(mm[k] + tw(5, k)), (int) sat16(im[k] + tw(6, k)));
    }
    if (xE >= sc_thresh) {
      for (int k = 1; k <= M; ++k) if (mn[k] == xE) out.push_back(LtWindow{ i, k, 1 });
      std::fill(mm.begin(), mm.end(), (int16_t) -32768); std::fill(im.begin(), im.end(), (int16_t) -32768); std::fill(dm.begin(), dm.end(), (int16_t) -32768);
      continue;
    }
    xC = std::max(xC, xE + xw_e_move);
    xJ = std::max(xJ, xE + xw_e_loop);
    xB = std::max(xJ + xw_move, xN + xw_move);
    dn[1] = -32768;
    for (int k = 2; k <= M; ++k) dn[k] = (int16_t) std::max((int) sat16(mn[k - 1] + tw(4, k - 1)), (int) sat16(dn[k - 1] + tw(7, k - 1)));
    mm.swap(mn); im.swap(in_); dm.swap(dn);
  }
}

// Forward / Backward parsers on a window (multihit, length model of the window): special-state rows only, two rolling
// DP rows.  Same arithmetic and scaling as forward_full / backward_full above; rows are (L+1) x [E,N,J,B,C,SCALE].
static int lt_forward_parser(Model &om, const uint8_t *dsq, int L, std::vector<float> &xmx, float *ret_sc)
{
  const int M = om.M;
  const float *__restrict bm = om.tf(0), *__restrict tMM = om.tf(1), *__restrict tIM = om.tf(2), *__restrict tDM = om.tf(3),
              *__restrict tMI = om.tf(5), *__restrict tII = om.tf(6);
  std::vector<float> buf((size_t) 6 * (M + 2), 0.0f);
  float *mp = buf.data(), *ip = mp + (M + 2), *dp = ip + (M + 2), *mc = dp + (M + 2), *ic = mc + (M + 2), *dc = ic + (M + 2);
  xmx.assign((size_t) (L + 1) * NX, 0.0f);
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = om.xf[XN][MOVE], xC = 0.f, totscale = 0.0f;
  xmx[xN_] = xN; xmx[xB_] = xB; xmx[xS_] = 1.0f;
  for (int r = 1; r <= L; ++r) {
    const float *__restrict rf = om.rf(dsq[r]);
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
    float *row = xmx.data() + (size_t) r * NX;
    if (xE > 1.0e4) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      const float inv = 1.0 / xE;
      for (int q = 1; q <= M; ++q) { mc[q] *= inv; dc[q] *= inv; ic[q] *= inv; }
      row[xS_] = xE;
      totscale += std::log((double) xE);
      xE = 1.0;
    } else row[xS_] = 1.0f;
    row[xE_] = xE; row[xN_] = xN; row[xJ_] = xJ; row[xB_] = xB; row[xC_] = xC;
    std::swap(mp, mc); std::swap(ip, ic); std::swap(dp, dc);
  }
  if (std::isnan(xC) || (L > 0 && xC == 0.0f) || std::isinf(xC)) { if (ret_sc) *ret_sc = INFINITY; return P7X_ERANGE; }
  if (ret_sc) *ret_sc = totscale + std::log((double) (xC * om.xf[XC][MOVE]));
  return P7X_OK;
}

// Backward parser on a window: the special-state rows only, two rolling DP rows.  Operation for operation backward_full()
// (same D chains, same sums), without its (L+1) x M matrices -- a 2,400-residue window of a 1,200-node model would
// stream 70 MB through the host caches for rows the region scan never reads.  fx: Forward's rows (scale factors).
static int lt_backward_parser(const Model &om, const uint8_t *dsq, int L, const std::vector<float> &fx, std::vector<float> &bx)
{
  const int M = om.M;
  const float *__restrict bm = om.tf(0), *__restrict tMM = om.tf(1), *__restrict tIM = om.tf(2), *__restrict tDM = om.tf(3),
              *__restrict tMD = om.tf(4), *__restrict tMI = om.tf(5), *__restrict tII = om.tf(6);
  std::vector<float> buf((size_t) 7 * (M + 3), 0.0f);
  float *mc = buf.data(), *ic = mc + (M + 3), *dc = ic + (M + 3), *mn = dc + (M + 3), *in = mn + (M + 3), *dn = in + (M + 3), *me = dn + (M + 3);
  bx.assign((size_t) (L + 1) * NX, 0.0f);
  auto X = [&](int r, int s) -> float & { return bx[(size_t) r * NX + s]; };
  auto FS = [&](int r) { return fx[(size_t) r * NX + xS_]; };
  bool own_scales = false;
  float xJ = 0.f, xB = 0.f, xN = 0.f;
  float xC = om.xf[XC][MOVE];
  float xE = xC * om.xf[XE][MOVE];
  {
    mc[M + 1] = ic[M + 1] = dc[M + 1] = 0.0f;
    for (int k = 1; k <= M; ++k) { dc[k] = xE; ic[k] = 0.0f; }
    dchain_backward(om, dc);
    for (int k = 1; k <= M; ++k) mc[k] = xE + dc[k + 1] * tMD[k];
    mc[0] = ic[0] = dc[0] = 0.0f;
    const float sc = FS(L);
    if (sc > 1.0f) {
      xE = xE / sc; xN = xN / sc; xC = xC / sc; xJ = xJ / sc; xB = xB / sc;
      const float inv = 1.0 / sc;
      for (int k = 1; k <= M; ++k) { mc[k] *= inv; dc[k] *= inv; ic[k] *= inv; }
    }
    X(L, xS_) = sc;
    X(L, xE_) = xE; X(L, xN_) = xN; X(L, xJ_) = xJ; X(L, xB_) = xB; X(L, xC_) = xC;
  }
  for (int r = L - 1; r >= 1; --r) {
    std::swap(mc, mn); std::swap(ic, in); std::swap(dc, dn);           // row r+1 becomes "next"
    const float *__restrict rf = om.rf(dsq[r + 1]);
    for (int k = 1; k <= M; ++k) me[k] = mn[k] * rf[k];
    me[M + 1] = 0.0f;
    xB = lanes_dot(om, me, bm);
    xC = xC * om.xf[XC][LOOP];
    xJ = (xB * om.xf[XJ][MOVE]) + (xJ * om.xf[XJ][LOOP]);
    xN = (xB * om.xf[XN][MOVE]) + (xN * om.xf[XN][LOOP]);
    xE = (xC * om.xf[XE][MOVE]) + (xJ * om.xf[XE][LOOP]);
    mc[M + 1] = ic[M + 1] = dc[M + 1] = 0.0f;
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
    if (xB > 1.0e16) own_scales = true;
    const float sc = own_scales ? ((xB > 1.0e4) ? xB : 1.0f) : FS(r);
    X(r, xS_) = sc;
    if (sc > 1.0f) {
      xE /= sc; xN /= sc; xJ /= sc; xB /= sc; xC /= sc;
      const float inv = 1.0 / sc;
      for (int k = 1; k <= M; ++k) { mc[k] *= inv; dc[k] *= inv; ic[k] *= inv; }
    }
    X(r, xE_) = xE; X(r, xN_) = xN; X(r, xJ_) = xJ; X(r, xB_) = xB; X(r, xC_) = xC;
  }
  {
    const float *__restrict rf = om.rf(dsq[1]);
    for (int k = 1; k <= M; ++k) me[k] = mc[k] * rf[k];
    xB = lanes_dot(om, me, bm);
    xN = (xB * om.xf[XN][MOVE]) + (xN * om.xf[XN][LOOP]);
    X(0, xB_) = xB; X(0, xC_) = 0.0f; X(0, xJ_) = 0.0f; X(0, xN_) = xN; X(0, xE_) = 0.0f; X(0, xS_) = 1.0f;
  }
  if (std::isnan(xN) || (L > 0 && xN == 0.0f) || std::isinf(xN)) return P7X_ERANGE;
  return P7X_OK;
}

// ---------------------------------------------------------------- Forward parser in upstream's summation order
// impl_sse/fwdback.c forward_engine() adds its floats in the order the striped 4-lane vectors impose: node k lives in
// lane z = (k-1) / Q of vector q = (k-1) % Q; xE is four per-lane sums over q (match cells first, delete cells after
// the delete chain) folded as (l0 + l1) + (l2 + l3); the D->D chain is one serial sweep per lane and then up to three
// carry sweeps from lane to lane (always three when M < 100, else until no cell grows).  The device kernels and
// forward_full() above use other (faster) association orders, a few ulps apart -- which only matters for a target
// whose P-value sits on the F3 threshold.  This routine is the tie-breaker for exactly those targets (the guard in
// p7x_tophits.cpp): plain scalar code, lane by lane, same operations in the same order.
// dsq[1..L]; multihit, length model of L as the pipeline configures the parser.
static int forward_parser_striped(const Profile &p, const uint8_t *dsq, int L, float *ret_sc)
{
  const int M = p.M, Q = p.Q4();
  Model om{ &p, M, {} };
  om.configure(true, L);
  const float *bm = om.tf(0), *tMM = om.tf(1), *tIM = om.tf(2), *tDM = om.tf(3), *tMD = om.tf(4), *tMI = om.tf(5), *tII = om.tf(6), *tDD = om.tf(7);
  struct V { float v[4]; };
  std::vector<V> mmo((size_t) Q, V{{0, 0, 0, 0}}), dmo = mmo, imo = mmo;
  // transitions of vector q, lane z (node k = q + 1 + z Q); padding nodes, and the transitions that would leave node M, are zero
  std::vector<V> vBM((size_t) Q), vMM = vBM, vIM = vBM, vDM = vBM, vMD = vBM, vMI = vBM, vII = vBM, vDD = vBM;
  for (int q = 0; q < Q; ++q)
    for (int z = 0; z < 4; ++z) {
      const int k = q + 1 + z * Q;
      const bool in = k <= M;
      vBM[q].v[z] = in ? bm[k] : 0.0f; vMM[q].v[z] = in ? tMM[k] : 0.0f; vIM[q].v[z] = in ? tIM[k] : 0.0f; vDM[q].v[z] = in ? tDM[k] : 0.0f;
      vMI[q].v[z] = in ? tMI[k] : 0.0f; vII[q].v[z] = in ? tII[k] : 0.0f;
      vMD[q].v[z] = (k < M) ? tMD[k] : 0.0f; vDD[q].v[z] = (k < M) ? tDD[k] : 0.0f;
    }
  auto rightshift = [](const V &a) { return V{{ 0.0f, a.v[0], a.v[1], a.v[2] }}; };
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = om.xf[XN][MOVE], xC = 0.f, totscale = 0.0f;
  std::vector<V> rv((size_t) Q);
  for (int i = 1; i <= L; ++i) {
    const float *rf = om.rf(dsq[i]);
    for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) { const int k = q + 1 + z * Q; rv[q].v[z] = k <= M ? rf[k] : 0.0f; }
    V dcv{{0, 0, 0, 0}}, xEv{{0, 0, 0, 0}};
    V mpv = rightshift(mmo[Q - 1]), dpv = rightshift(dmo[Q - 1]), ipv = rightshift(imo[Q - 1]);
    for (int q = 0; q < Q; ++q) {
      V sv;
      for (int z = 0; z < 4; ++z) {
        float s = xB * vBM[q].v[z];
        s = s + mpv.v[z] * vMM[q].v[z];
        s = s + ipv.v[z] * vIM[q].v[z];
        s = s + dpv.v[z] * vDM[q].v[z];
        s = s * rv[q].v[z];
        sv.v[z] = s;
        xEv.v[z] = xEv.v[z] + s;
      }
      mpv = mmo[q]; dpv = dmo[q]; ipv = imo[q];
      mmo[q] = sv;
      dmo[q] = dcv;
      for (int z = 0; z < 4; ++z) {
        dcv.v[z] = sv.v[z] * vMD[q].v[z];
        const float t = mpv.v[z] * vMI[q].v[z];
        imo[q].v[z] = t + ipv.v[z] * vII[q].v[z];
      }
    }
    dcv = rightshift(dcv);
    dmo[0] = V{{0, 0, 0, 0}};
    for (int q = 0; q < Q; ++q)
      for (int z = 0; z < 4; ++z) { dmo[q].v[z] = dcv.v[z] + dmo[q].v[z]; dcv.v[z] = dmo[q].v[z] * vDD[q].v[z]; }
    if (M < 100) {
      for (int j = 1; j < 4; ++j) {
        dcv = rightshift(dcv);
        for (int q = 0; q < Q; ++q)
          for (int z = 0; z < 4; ++z) { dmo[q].v[z] = dcv.v[z] + dmo[q].v[z]; dcv.v[z] = dcv.v[z] * vDD[q].v[z]; }
      }
    } else {
      for (int j = 1; j < 4; ++j) {
        bool grew = false;
        dcv = rightshift(dcv);
        for (int q = 0; q < Q; ++q)
          for (int z = 0; z < 4; ++z) {
            const float s = dcv.v[z] + dmo[q].v[z];
            if (s > dmo[q].v[z]) grew = true;
            dmo[q].v[z] = s;
            dcv.v[z] = dcv.v[z] * vDD[q].v[z];
          }
        if (!grew) break;
      }
    }
    for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) xEv.v[z] = dmo[q].v[z] + xEv.v[z];
    {
      const float a0 = xEv.v[0] + xEv.v[1], a2 = xEv.v[2] + xEv.v[3];
      xE = a0 + a2;
    }
    xN = xN * om.xf[XN][LOOP];
    xC = (xC * om.xf[XC][LOOP]) + (xE * om.xf[XE][MOVE]);
    xJ = (xJ * om.xf[XJ][LOOP]) + (xE * om.xf[XE][LOOP]);
    xB = (xJ * om.xf[XJ][MOVE]) + (xN * om.xf[XN][MOVE]);
    if (xE > 1.0e4) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      const float inv = 1.0 / xE;
      for (int q = 0; q < Q; ++q) for (int z = 0; z < 4; ++z) { mmo[q].v[z] = mmo[q].v[z] * inv; dmo[q].v[z] = dmo[q].v[z] * inv; imo[q].v[z] = imo[q].v[z] * inv; }
      totscale += std::log((double) xE);
      xE = 1.0;
    }
  }
  if (std::isnan(xC) || (L > 0 && xC == 0.0f) || std::isinf(xC)) { *ret_sc = INFINITY; return P7X_ERANGE; }
  *ret_sc = totscale + std::log((double) (xC * om.xf[XC][MOVE]));
  return P7X_OK;
}

// ---------------------------------------------------------------- windows
struct LtScoreData { std::vector<float> prefix, suffix; };       // [M+1] fractions of the model's maximal length up to / from node k

// p7_hmm_ScoreDataComputeRest: per node the longest insert that still carries a tail mass of 1e-7, cumulated and
// normalised to fractions of the whole model
static void lt_scoredata(const Profile &p, LtScoreData &sd)
{
  const int M = p.M;
  const double beta = 1e-7;                                   // p7_DEFAULT_WINDOW_BETA
  sd.prefix.assign(M + 2, 0.0f); sd.suffix.assign(M + 2, 0.0f);
  float sum = 0.0f;
  for (int k = 1; k < M; ++k) {
    const float tmi = p.tf[(size_t) tMI * (M + 1) + k], tii = p.tf[(size_t) tII * (M + 1) + k];
    float len = 2.0f;
    if (tmi > 0.0f && tii > 0.0f && tii < 1.0f) len = 2.0f + (float) (int) (std::log(beta / tmi) / std::log(tii));
    if (len < 2.0f) len = 2.0f;
    sd.prefix[k] = len;
    sum += len;
  }
  sd.prefix[M] = 1.0f; sum += 1.0f;
  for (int k = 1; k <= M; ++k) sd.prefix[k] /= sum;
  sd.suffix[M] = sd.prefix[M];
  for (int k = M - 1; k >= 1; --k) sd.suffix[k] = sd.suffix[k + 1] + sd.prefix[k];
  for (int k = 2; k <= M; ++k) sd.prefix[k] += sd.prefix[k - 1];
}

// p7_pli_ExtendAndMergeWindows (one strand, positions on that strand)
static void lt_extend_and_merge(const LtScoreData &sd, int max_length, int64_t target_len, float pct_overlap, std::vector<LtWindow> &w)
{
  if (w.empty()) return;
  for (LtWindow &c : w) {
    const int kfirst = std::max<int64_t>(1, (int64_t) c.k - c.length + 1);
    const int64_t ws = std::max<int64_t>(1, c.n - (int64_t) (max_length * (0.1 + sd.prefix[(size_t) kfirst])));
    const int64_t we = std::min<int64_t>(target_len, c.n + c.length + (int64_t) (max_length * (0.1 + sd.suffix[(size_t) c.k])));
    c.length = we - ws + 1; c.n = ws;
  }
  size_t cnt = 0;
  for (size_t i = 1; i < w.size(); ++i) {
    LtWindow &prev = w[cnt]; const LtWindow &cur = w[i];
    const int64_t os = std::max(prev.n, cur.n), oe = std::min(prev.n + prev.length - 1, cur.n + cur.length - 1);
    if ((float) (oe - os + 1) / (float) std::min(prev.length, cur.length) > pct_overlap) {
      const int64_t ms = std::min(prev.n, cur.n), me = std::max(prev.n + prev.length - 1, cur.n + cur.length - 1);
      prev.n = ms; prev.length = me - ms + 1;
    } else { ++cnt; w[cnt] = w[i]; }
  }
  w.resize(cnt + 1);
}

static double lt_gumbel_invsurv(double P, double mu, double lambda) { return mu - std::log(-1.0 * std::log(1.0 - P)) / lambda; }

// the score threshold of p7_SSVFilter_longtarget (byte units) and the constant begin score
static void lt_ssv_threshold(const Profile &p, int max_length, double F1, int *sc_thresh, int *xB, int *tjb)
{
  const double invP = lt_gumbel_invsurv(F1, p.evparam[P7X_MMU], p.evparam[P7X_MLAMBDA]);
  const float nullsc = lt_null1(max_length);
  const int tjb_b = unbiased_byteify(p.scale_b, logf(3.0f / (float) (max_length + 3)));
  *sc_thresh = (int) std::ceil(((nullsc + (invP * kLog2) + 3.0) * p.scale_b) + p.base_b + p.tec_b + tjb_b);
  *xB = std::max((int) p.base_b - tjb_b - (int) p.tbm_b, 0);
  *tjb = tjb_b;
}

// What p7_SSVFilter_longtarget does with a row i whose cell (k, sc) reached the threshold: recover the diagonal back
// to where it left the begin score, extend it forward while it keeps (nearly) rising, emit the window.  Returns the
// last row of the extended diagonal (upstream resumes scanning behind it).
static int64_t lt_seed_from_cell(const Profile &p, const uint8_t *dsq, int64_t L, int64_t i, int k, int sc, int xB, LtWindow *out)
{
  const int M = p.M, bias = p.bias_b;
  auto cost = [&](int kk, int64_t pos) -> int { return (int) p.rb[(size_t) dsq[pos] * (M + 1) + kk]; };
  int start = k; int64_t tstart = i; int rem = sc;
  while (rem > xB && start >= 1 && tstart >= 1) { rem -= bias - cost(start, tstart); --start; --tstart; }
  ++start; ++tstart;
  int kk = k + 1; int64_t n = i + 1, max_end = i; int max_sc = sc, cur = sc, since = 0;
  while (kk < M && n <= L) {
    cur += bias - cost(kk, n);
    if (cur >= max_sc) { max_sc = cur; max_end = n; since = 0; }
    else if (++since == 5) break;
    ++kk; ++n;
  }
  const int end = k + (int) (max_end - i);
  out->n = tstart; out->k = end; out->length = end - start + 1;
  return max_end;
}

// A short exact replay of the scan from row <from> (all cells at the begin score) to <to>: the first row whose best
// cell reaches the threshold, with upstream's choice of cell.  Used in the shadow of a previous seed, where the
// reset-free device scan over-reports.
static bool lt_replay(const Profile &p, const uint8_t *dsq, int64_t from, int64_t to, int sc_thresh, int xB, int64_t *row_out, int *k_out, int *sc_out)
{
  const int M = p.M, bias = p.bias_b, Q = p.Q16();
  std::vector<int> row(M + 1, 0), nxt(M + 1, 0);
  for (int64_t i = from; i <= to; ++i) {
    const uint8_t *rb = p.rb.data() + (size_t) dsq[i] * (M + 1);
    int best = -1, bestkey = INT_MAX;
    for (int k = 1; k <= M; ++k) {
      int sv = std::max(row[k - 1], xB);
      sv = std::min(sv + bias, 255);
      sv = std::max(sv - (int) rb[k], 0);
      nxt[k] = sv;
      if (sv >= sc_thresh) {
        const int key = ((k - 1) % Q) * 16 + (k - 1) / Q;
        if (sv > best || (sv == best && key < bestkey)) { best = sv; bestkey = key; }
      }
    }
    if (best >= 0) { *row_out = i; *k_out = (bestkey / 16) + Q * (bestkey % 16) + 1; *sc_out = best; return true; }
    row.swap(nxt);
  }
  return false;
}

struct LtRow { int64_t pos; int k, sc; };       // a row the device scan reported: position on the strand, upstream's cell

// Upstream's sequential bookkeeping over the rows the reset-free scan reported for one block of one strand.  Rows are
// positions inside the block (1..L), ascending.
static void lt_seeds_from_rows(const Profile &p, const uint8_t *dsq, int64_t L, const std::vector<LtRow> &rows, int sc_thresh, int xB,
                               std::vector<LtWindow> &seeds)
{
  const int M = p.M;
  // The scan restarts (all cells at the begin score) behind every seed and at the block start; rows up to M behind a
  // restart are its "shadow": a diagonal through them may have begun before the restart, so the reset-free scores of the
  // device do not apply there.  Outside shadows they are upstream's.
  int64_t restart = 0;             // the scan restarted at row restart + 1
  bool shadow = true;              // rows restart + 1 .. restart + M still have to be looked at
  int64_t skip_to = 0;             // rows up to here are behind us (consumed by a seed's diagonal, or replayed)
  size_t idx = 0;
  while (idx < rows.size()) {
    const LtRow &r = rows[idx];
    if (r.pos <= skip_to) { ++idx; continue; }
    int64_t row = r.pos; int k = r.k, sc = r.sc;
    if (shadow && r.pos <= restart + M) {
      // replay the scan from the restart; a true crossing can only be at a row the device reported, so the replay ends at
      // the last reported row inside the shadow
      int64_t to = r.pos;
      for (size_t j = idx; j < rows.size() && rows[j].pos <= restart + M; ++j) to = rows[j].pos;
      to = std::min(to, L);
      if (!lt_replay(p, dsq, restart + 1, to, sc_thresh, xB, &row, &k, &sc)) { skip_to = to; shadow = false; continue; }
    }
    LtWindow w;
    const int64_t end = lt_seed_from_cell(p, dsq, L, row, k, sc, xB, &w);
    seeds.push_back(w);
    restart = end; shadow = true; skip_to = end;
  }
}

// ---------------------------------------------------------------- one window past the SSV filter
struct LtTarget { int64_t idx; const char *name, *acc, *desc; int64_t length; };

struct LtBlock {                 // one block of one strand, as p7_Pipeline_LongTarget receives it
  const uint8_t *dsq;            // 1-based residues of the block on this strand (dsq[0] and dsq[n+1] are sentinels)
  int64_t n;                     // residues in the block
  int64_t start;                 // sq->start: original coordinate of dsq[1] (for the complement strand: the block's last residue)
  bool complement;
};

struct LtCounters { uint64_t n_past_msv = 0, n_past_bias = 0, n_past_vit = 0, n_past_fwd = 0, pos_past_msv = 0, pos_past_bias = 0, pos_past_vit = 0, pos_past_fwd = 0; };

// blk.dsq is not used here: <subseq>[1..window_len] are the window's residues; blk.start / blk.complement and
// window_start (the window's first residue in the block) map coordinates back to the target.  fwd_given: the window's
// Forward parser score when it was computed elsewhere (the device batch).
// The Forward filter of one Viterbi window (p7_pli_postViterbi_LongTarget up to the F3 test): true when it passes.
static bool lt_forward_test(const p7x_pipeline_cfg &cfg, const Profile &p, int64_t window_len, const uint8_t *subseq, float fwdsc)
{
  const int64_t F3_L = std::min<int64_t>(window_len, cfg.B3);
  const float nullsc = lt_null1(window_len);
  float filtersc = nullsc;
  if (cfg.do_biasfilter) {
    float bias_filtersc = lt_bias_filter(p, subseq, window_len);
    bias_filtersc -= nullsc;
    filtersc = nullsc + (bias_filtersc * (F3_L > window_len ? 1.0f : (float) F3_L / (float) window_len));
  }
  const float seq_score = (fwdsc - filtersc) / (float) kLog2;
  return !(exp_surv(seq_score, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]) > cfg.F3);
}

static int lt_window_hits(const p7x_pipeline_cfg &cfg, const Profile &p, int max_length, uint64_t nres_so_far, const LtBlock &blk, const LtTarget &tg,
                          int64_t window_start, DomainDefResult &dd, std::vector<Hit> &hits);

// The rest of p7_pli_postViterbi_LongTarget for a window that passed the Forward filter: Backward, domain definition
// with long_target = TRUE, one hit per domain.  <dev>: the device's region scan of this window (n >= 0), else the host
// runs the parsers itself.
static int lt_post_viterbi(const p7x_pipeline_cfg &cfg, const Profile &p, const LongTargetOpts &lto, int max_length, uint64_t nres_so_far,
                           const LtBlock &blk, const LtTarget &tg, int64_t window_start, int64_t window_len, const uint8_t *subseq,
                           float fwdsc, const LongTargetWindowRegions *dev, std::vector<Hit> &hits, LtCounters &ctr)
{
  ctr.n_past_fwd++; ctr.pos_past_fwd += (uint64_t) window_len;
  DomainDefResult dd;
  int st = P7X_OK;
  if (dev && dev->n == -1) return P7X_OK;                      // p7_DomainDecoding: eslERANGE, nothing comes of this window
  t_long_target = &lto;
  if (dev && dev->n >= 0) {
    st = domaindef_from_regions(p, subseq, (int) window_len, dev->nexpected, dev->regs.data(), dev->n, cfg.seed, cfg.seed != 0, dd, nullptr, 0);
  } else {
    // Backward parser rows: the full-matrix routine on the window would need L x M floats; the region scan only needs the
    // special states, which the generic Backward delivers row by row.  Windows are a few max_length long.
    Model om{ &p, p.M, {} };
    om.prepare();
    om.configure(true, (int) window_len);
    std::vector<float> fx, bx;
    float sc2 = 0.0f;
    lt_forward_parser(om, subseq, (int) window_len, fx, &sc2);
    lt_backward_parser(om, subseq, (int) window_len, fx, bx);
    st = domaindef_by_posterior_heuristics(p, subseq, (int) window_len, fx.data(), bx.data(), cfg.seed, cfg.seed != 0, dd, nullptr, 0);
  }
  t_long_target = nullptr;
  if (st != P7X_OK) return st == P7X_ERANGE ? P7X_OK : st;
  if ((debug_opt(OPT_TRACE_LONGTARGET) > 0)) {
    std::fprintf(stderr, "[lt] window start %lld len %lld compl %d fwd %.3f: nregions %d nclustered %d nenvelopes %d ndom %zu\n",
                 (long long) window_start, (long long) window_len, (int) blk.complement, fwdsc, dd.nregions, dd.nclustered, dd.nenvelopes, dd.dcl.size());
    for (const Domain &d : dd.dcl) std::fprintf(stderr, "[lt]   env %lld-%lld ali %lld-%lld hmm %d-%d envsc %.3f domcorr %.3f\n", (long long) d.ienv, (long long) d.jenv,
                                                 (long long) d.iali, (long long) d.jali, d.hmmfrom, d.hmmto, d.envsc, d.domcorrection);
  }
  return lt_window_hits(cfg, p, max_length, nres_so_far, blk, tg, window_start, dd, hits);
}

// The domains of one window (coordinates relative to the window) -> hits: p7_pli_postDomainDef of the long-target pipeline
static int lt_window_hits(const p7x_pipeline_cfg &cfg, const Profile &p, int max_length, uint64_t nres_so_far, const LtBlock &blk, const LtTarget &tg,
                          int64_t window_start, DomainDefResult &dd, std::vector<Hit> &hits)
{
  if (dd.nregions == 0 || dd.nenvelopes == 0) return P7X_OK;
  for (Domain &dom : dd.dcl) {
    const int64_t env_len = dom.jenv - dom.ienv + 1, ali_len = dom.jali - dom.iali + 1;
    float bitscore = dom.envsc;
    // the envelope was scored (unihit) under a length model of its own length: take out what that model charged for
    // entering, leaving and the envelope's flanks, then re-express the score as if every window had the length
    // max_length, so that scores do not depend on how the windows happened to merge
    bitscore -= 2 * log(2. / (env_len + 2)) + (env_len - ali_len) * log((float) env_len / (float) (env_len + 2));
    bitscore += 2 * log(2. / (max_length + 2));
    bitscore += (std::max<int64_t>(max_length, env_len) - ali_len) * log((float) max_length / (float) (max_length + 2));
    const float dom_nullsc = lt_null1(std::max<int64_t>(max_length, env_len));
    const float dom_bias = cfg.do_null2 ? dom.domcorrection : 0.0f;      // long targets: the correction is the bias (no omega prior)
    const float dom_score = (bitscore - (dom_nullsc + dom_bias)) / (float) kLog2;
    const double dom_lnP = exp_logsurv(dom_score, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]);
    // conservative test with the residues seen so far; the final E-values use the whole search (ComputeNhmmerEvalues)
    const double lnP_test = dom_lnP + std::log((double) std::max<uint64_t>(nres_so_far, 1) / (double) max_length);
    if (!tophits_target_reportable(cfg, dom_score, lnP_test)) continue;
    Hit h;
    h.ndom = 1; h.best_domain = 0; h.window_length = max_length; h.seqidx = tg.idx;
    if (tg.name) h.name = tg.name;
    if (tg.acc && tg.acc[0]) { h.acc = tg.acc; h.has_acc = true; }
    if (tg.desc && tg.desc[0]) { h.desc = tg.desc; h.has_desc = true; }
    // positions in the original target: blk.start is the original coordinate of the block's first residue on this strand
    auto map_pos = [&](int64_t x) -> int64_t {
      return blk.complement ? blk.start - (window_start + x) + 2 : (blk.start - 1) + (window_start - 1) + x;
    };
    dom.ienv = map_pos(dom.ienv); dom.jenv = map_pos(dom.jenv);
    dom.iali = map_pos(dom.iali); dom.jali = map_pos(dom.jali);
    dom.sqfrom = map_pos(dom.sqfrom); dom.sqto = map_pos(dom.sqto);
    dom.L = tg.length;
    dom.dombias = dom_bias; dom.bitscore = dom_score; dom.lnP = dom_lnP;
    h.pre_score = bitscore / (float) kLog2;
    h.pre_lnP = exp_logsurv(h.pre_score, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]);
    h.sum_score = h.score = dom_score;
    h.sum_lnP = h.lnP = dom_lnP;
    h.sortkey = cfg.inc_by_E ? -dom_lnP : dom_score;
    h.nexpected = dd.nexpected; h.nregions = dd.nregions; h.nclustered = dd.nclustered; h.noverlaps = dd.noverlaps; h.nenvelopes = dd.nenvelopes;
    h.dcl.push_back(std::move(dom));
    hits.push_back(std::move(h));
  }
  return P7X_OK;
}

// Filter scores of a window computed elsewhere (the device batch of p7x_longtarget.hip): MSV score, bias filter score,
// and the standard Viterbi filter score, which bounds every row of the long-target Viterbi scan from above.
struct LtWindowFilters { bool have = false; float usc = 0.0f, bias_filtersc = 0.0f; bool have_vit = false; float vfsc = 0.0f; };

// p7_pli_postSSV_LongTarget up to the Viterbi step: MSV and bias tests of one window.  state: 0 dropped, 2 goes through the
// long-target Viterbi scan with score threshold <vit_thresh>.  (Every surviving window does, also one whose MSV P-value is
// already below F2: the scan is what cuts a merged SSV window into the Viterbi windows Forward is run on.  Passing such
// windows on whole -- the protein pipeline's shortcut -- made bmyD2.tbl's third row come out of a 4.6 kb window together
// with a weak envelope nhmmer never sees.)
struct LtPrefilter { int state = 0; int vit_thresh = 0; float filtersc_f2 = 0.0f; };

static LtPrefilter lt_window_prefilter(const p7x_pipeline_cfg &cfg, const Profile &p, const uint8_t *subseq, int64_t window_len,
                                       const LtWindowFilters *wf, LtCounters &ctr)
{
  LtPrefilter out;
  const int64_t F1_L = std::min<int64_t>(window_len, cfg.B1), F2_L = std::min<int64_t>(window_len, cfg.B2);
  const float nullsc = lt_null1(window_len);
  // the full MSV score of the window (SSV only seeded it)
  const float usc = (wf && wf->have) ? wf->usc : lt_msv(p, subseq, window_len);
  double P = gumbel_surv((usc - nullsc) / kLog2, p.evparam[P7X_MMU], p.evparam[P7X_MLAMBDA]);
  if (P > cfg.F1) return out;
  ctr.n_past_msv++; ctr.pos_past_msv += (uint64_t) window_len;
  float bias_filtersc = 0.0f, filtersc = nullsc;
  if (cfg.do_biasfilter) {
    bias_filtersc = ((wf && wf->have) ? wf->bias_filtersc : lt_bias_filter(p, subseq, window_len)) - nullsc;
    filtersc = nullsc + (bias_filtersc * (F1_L > window_len ? 1.0f : (float) F1_L / (float) window_len));
    P = gumbel_surv((usc - filtersc) / kLog2, p.evparam[P7X_MMU], p.evparam[P7X_MLAMBDA]);
    if (P > cfg.F1) return out;
  }
  ctr.n_past_bias++; ctr.pos_past_bias += (uint64_t) window_len;
  if (cfg.do_biasfilter) filtersc = nullsc + (bias_filtersc * (F2_L > window_len ? 1.0f : (float) F2_L / (float) window_len));
  // The standard Viterbi filter score of the window is at least the score any single row reaches in the long-target
  // scan (its C state collects every row's E; clearing rows can only lower later ones): a window that fails P <= F2
  // with it cannot seed a Viterbi window.
  if (wf && wf->have_vit && gumbel_surv((wf->vfsc - filtersc) / kLog2, p.evparam[P7X_VMU], p.evparam[P7X_VLAMBDA]) > cfg.F2) return out;
  out.state = 2; out.filtersc_f2 = filtersc; out.vit_thresh = lt_viterbi_threshold(p, window_len, filtersc, cfg.F2);
  return out;
}

// p7_Pipeline_LongTarget behind the SSV scan, first half: the seeds of one block of one strand become its windows
static void lt_block_windows(const LtScoreData &sd, int max_length, int64_t block_len, std::vector<LtWindow> seeds, std::vector<LtWindow> &windows)
{
  windows.clear();
  if (seeds.empty()) return;
  lt_extend_and_merge(sd, max_length, block_len, 0.0f, seeds);
  // very long merged windows are cut into overlapping pieces (upstream: longer than 80 kb -> 40 kb pieces)
  const int64_t max_window = 80000, piece = 40000;
  for (const LtWindow &w : seeds) {
    if (w.length <= max_window) { windows.push_back(w); continue; }
    for (int64_t off = 0; off < w.length; off += piece - max_length) {
      const int64_t len = std::min<int64_t>(piece, w.length - off);
      windows.push_back(LtWindow{ w.n + off, 0, len });
      if (off + len >= w.length) break;
    }
  }
}

// ---------------------------------------------------------------- hit list: E-values, duplicates
// E-values, duplicates, order and thresholds of a long-target hit list whose hits still carry per-window P-values
static void lt_finalize(p7x_tophits *th, int max_length, double res_count)
{
  std::vector<Hit> &hits = th->hits;
  // p7_tophits_ComputeNhmmerEvalues: the P-value of a hit refers to one window of max_length; scale by the windows searched
  for (Hit &h : hits) {
    h.lnP += std::log((double) ((float) res_count / (float) max_length));
    h.dcl[0].lnP = h.lnP;
    h.sortkey = -1.0 * h.lnP;
  }
  // p7_tophits_SortBySeqidxAndAlipos + p7_tophits_RemoveDuplicates: the same region found in two overlapping blocks or
  // windows; the hit with the better E-value stays
  std::vector<size_t> ord(hits.size());
  for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
  auto lo = [&](const Hit &h) { return std::min(h.dcl[0].iali, h.dcl[0].jali); };
  std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
    if (hits[a].seqidx != hits[b].seqidx) return hits[a].seqidx < hits[b].seqidx;
    const bool ca = hits[a].dcl[0].iali > hits[a].dcl[0].jali, cb = hits[b].dcl[0].iali > hits[b].dcl[0].jali;
    if (ca != cb) return !ca;
    if (lo(hits[a]) != lo(hits[b])) return lo(hits[a]) < lo(hits[b]);
    return a < b;
  });
  size_t j = 0;
  for (size_t q = 1; q < ord.size(); ++q) {
    Hit &hj = hits[ord[j]], &hi = hits[ord[q]];
    const bool cj = hj.dcl[0].iali > hj.dcl[0].jali, ci = hi.dcl[0].iali > hi.dcl[0].jali;
    bool dup = false;
    if (hj.seqidx == hi.seqidx && cj == ci) {
      const int64_t sj = std::min(hj.dcl[0].iali, hj.dcl[0].jali), ej = std::max(hj.dcl[0].iali, hj.dcl[0].jali);
      const int64_t si = std::min(hi.dcl[0].iali, hi.dcl[0].jali), ei = std::max(hi.dcl[0].iali, hi.dcl[0].jali);
      const int64_t inter = std::min(ei, ej) - std::max(si, sj) + 1;
      const int64_t li = ei - si + 1, lj = ej - sj + 1;
      // upstream: only hits on similar parts of the model (their model ranges intersect) can be duplicates of each
      // other -- tandem or partial repeat copies with flush ends but different model ranges are both kept
      const int hmm_inter = std::min(hi.dcl[0].hmmto, hj.dcl[0].hmmto) - std::max(hi.dcl[0].hmmfrom, hj.dcl[0].hmmfrom) + 1;
      if (hmm_inter > 0 && ((std::llabs(si - sj) <= 3) || (std::llabs(ei - ej) <= 3) || inter >= 0.95 * (double) li || inter >= 0.95 * (double) lj)) dup = true;
    }
    if (dup) {
      const size_t remove = hi.lnP < hj.lnP ? j : q;
      hits[ord[remove]].flags |= P7X_IS_DUPLICATE;
      if (remove == j) j = q;
    } else j = q;
  }
  th->order.clear(); th->sorted_by_key = false;
  tophits_sort_by_key(*th);
  tophits_threshold(*th);
  th->ctr.n_output = th->ctr.pos_output = 0;
  for (const Hit &h : th->hits)
    if (h.flags & (P7X_IS_REPORTED | P7X_IS_INCLUDED)) { th->ctr.n_output++; th->ctr.pos_output += 1 + (uint64_t) std::llabs(h.dcl[0].jali - h.dcl[0].iali); }
  th->lt_unfinished = false;
}

static double lt_res_count(const p7x_pipeline_cfg &cfg, uint64_t nres)
{
  double res_count = (double) nres;
  if (cfg.Z_setby != P7X_ZSETBY_NTARGETS) { res_count = 1000000.0 * cfg.Z; if (cfg.strands == P7X_STRAND_BOTH) res_count *= 2; }
  return res_count;
}

static void lt_finish_tophits(const p7x_pipeline_cfg &cfg_in, const Profile &p, int max_length, uint64_t nres, uint64_t nseqs,
                              const LtCounters &ctr, std::vector<Hit> &hits, p7x_tophits **out)
{
  auto th = std::make_unique<p7x_tophits>();
  th->cfg = cfg_in;
  th->qname = p.name; th->qacc = p.acc; th->qdesc = p.desc; th->q_has_acc = p.has_acc; th->q_has_desc = p.has_desc;
  th->M = p.M;
  th->ctr.nmodels = 1; th->ctr.nnodes = (uint64_t) p.M; th->ctr.nseqs = nseqs; th->ctr.nres = nres;
  th->ctr.n_past_msv = ctr.n_past_msv; th->ctr.n_past_bias = ctr.n_past_bias; th->ctr.n_past_vit = ctr.n_past_vit; th->ctr.n_past_fwd = ctr.n_past_fwd;
  th->ctr.pos_past_msv = ctr.pos_past_msv; th->ctr.pos_past_bias = ctr.pos_past_bias; th->ctr.pos_past_vit = ctr.pos_past_vit; th->ctr.pos_past_fwd = ctr.pos_past_fwd;
  th->hits = std::move(hits);
  th->lt_evalue_window = max_length;
  th->lt_unfinished = true;
  // one part of several: the parts are finished together (p7x_tophits_merge_longtargets)
  if (cfg_in.lt_nparts <= 1) lt_finalize(th.get(), max_length, lt_res_count(cfg_in, nres));
  *out = th.release();
}

static void lt_match_probabilities(const Profile &p, std::vector<float> &mp)
{ // p7_oprofile_GetFwdEmissionArray: match emission probabilities back from the odds ratios
  mp.assign((size_t) (p.M + 1) * p.K, 0.0f);
  for (int k = 1; k <= p.M; ++k)
    for (int x = 0; x < p.K; ++x) mp[(size_t) k * p.K + x] = p.rf_[(size_t) x * (p.M + 1) + k] * p.bgf[x];
}

static const uint8_t *lt_complement_table(int abc_type)
{ // Easel's complement of every DNA / RNA residue code: ACGT-RYMKSWHBVDN*~
  static const uint8_t comp[18] = { 3, 2, 1, 0, 4, 6, 5, 8, 7, 9, 10, 14, 13, 12, 11, 15, 16, 17 };
  (void) abc_type;
  return comp;
}

struct LtSeedIn { int64_t target, block_start; int strand; LtWindow w; };

// The whole host side for a set of targets, given the SSV seeds of every (target, block, strand).  <filters>, when
// given, scores all windows of a target in one device batch before the host sees them.
static int lt_run_host(const p7x_pipeline_cfg &cfg, const p7x_oprofile *om, const uint8_t *dsq, const int64_t *offsets, const int64_t *lengths, size_t n,
                       const char *const *names, const char *const *accs, const char *const *descs,
                       const std::vector<LtSeedIn> &seeds_in, p7x_tophits **out, LongTargetWindowScorer *filters)
{
  const Profile &p = om->p;
  if (p.abc_type != P7X_DNA && p.abc_type != P7X_RNA) { set_error("long-target pipeline needs a nucleotide model"); return P7X_EINVAL; }
  const int max_length = cfg.window_length > 0 ? cfg.window_length : p.max_length;
  if (max_length <= 0) { set_error("model has no max_length (MAXL) and no window_length was given"); return P7X_EINVAL; }
  const int64_t W = cfg.block_length, C = max_length;
  if (W <= C) { set_error("block_length must exceed the model's max_length"); return P7X_EINVAL; }
  flogsum_init();
  LtScoreData sd; lt_scoredata(p, sd);
  std::vector<float> mp; lt_match_probabilities(p, mp);
  LongTargetOpts lto; lto.do_null2 = cfg.do_null2 != 0; lto.match_prob = mp.data();
  const uint8_t *comp = lt_complement_table(p.abc_type);
  std::vector<Hit> hits;
  LtCounters ctr;
  uint64_t nres = 0;
  size_t sidx = 0;
  LongTargetUnits units; units.count(cfg, max_length, lengths, n);
  uint64_t unit = 0;
  struct BlockJob { size_t t; int64_t i, bn, bc, bw; int strand; uint64_t nres_at; std::vector<LtWindow> windows; size_t first_window; };
  // One pass over the targets collects the windows of every (target, block, strand); everything after that runs once
  // over all of them -- one device batch per stage for the whole search, however many records the target file has
  // (an assembly of 1e5 contigs costs the same number of launches as one chromosome).  Window references handed to
  // the device stages are positions in `dsq` itself (offsets[t] - 1 + position in the target).
  std::vector<BlockJob> jobs;
  size_t nwin = 0;
  auto seq_of = [&](const BlockJob &job) { return dsq + offsets[job.t] - 1; };      // seq[1..Lt] of the job's target
  auto target_of = [&](const BlockJob &job) { const size_t t = job.t; return LtTarget{ (int64_t) t, names ? names[t] : nullptr, accs ? accs[t] : nullptr, descs ? descs[t] : nullptr, lengths[t] }; };
  // P7X_LT_DEBUG: wall time of the phases of this function
  const bool dbg = debug_opt(OPT_TRACE_LONGTARGET) > 0;
  auto t_last = std::chrono::steady_clock::now();
  auto tick = [&](const char *what) {
    if (!dbg) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[lt] host phase %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  for (size_t t = 0; t < n; ++t) {
    const int64_t Lt = lengths[t];
    // the windows of every block and strand of this target (and the residue accounting of the block loop)
    for (int64_t i = 0; i < Lt; i += W - C) {
      const int64_t bc = i == 0 ? 0 : std::min<int64_t>(C, Lt - i);
      const int64_t bw = std::min<int64_t>(W, Lt - i - bc);
      const int64_t bn = bc + bw;
      if (bn <= 0) break;
      nres += (uint64_t) bn;                                 // p7_pli_NewSeq
      for (int strand = 0; strand < 2; ++strand) {
        if (strand == 0 && cfg.strands == P7X_STRAND_BOTTOMONLY) { nres -= (uint64_t) bn; continue; }
        if (strand == 0) nres -= (uint64_t) bc;              // the overlap with the previous block was counted there
        if (strand == 1 && cfg.strands == P7X_STRAND_TOPONLY) continue;
        const bool mine = units.mine(unit++);                // another part's unit: only its residues are counted here
        std::vector<LtWindow> seeds;
        while (sidx < seeds_in.size() && seeds_in[sidx].target == (int64_t) t && seeds_in[sidx].block_start == i && seeds_in[sidx].strand == strand)
          seeds.push_back(seeds_in[sidx++].w);
        BlockJob job{ t, i, bn, bc, bw, strand, nres, {}, nwin };
        if (mine) lt_block_windows(sd, max_length, bn, std::move(seeds), job.windows);
        nwin += job.windows.size();
        if (!job.windows.empty()) jobs.push_back(std::move(job));
        if (strand == 1) nres += (uint64_t) bw;
      }
      // no early exit: the reference loop (plan7.pyx:7604 `for i from 0 <= i < sq[t].n by W - C`) visits every start
      // below the target length, also a trailing block that only repeats residues the previous block has seen
    }
  }
  {
    tick("windows of the blocks");
    // the windows' filter scores, one device batch
    std::vector<LtWindowFilters> wf;
    std::vector<LongTargetWindowRef> refs;
    if (filters && nwin > 0) {
      refs.reserve(nwin);
      for (const BlockJob &job : jobs)
        for (const LtWindow &w : job.windows) {
          // original coordinates of the window's first residue on its strand and its length
          LongTargetWindowRef r; r.strand = job.strand; r.length = w.length;
          r.start = job.strand == 0 ? job.i + w.n : job.i + job.bn - w.n + 1;       // strand 1: original position of the window's first (revcomp) residue
          r.start += offsets[job.t] - 1;
          refs.push_back(r);
        }
      std::vector<LongTargetWindowScore> sc(nwin);
      const int st = filters->score(dsq, 0, comp, refs.data(), nwin, cfg.F1, cfg.do_biasfilter != 0, sc.data());
      if (st != P7X_OK) return st;
      wf.resize(nwin);
      for (size_t q = 0; q < nwin; ++q) { wf[q].have = true; wf[q].usc = sc[q].usc; wf[q].bias_filtersc = sc[q].bias_filtersc; wf[q].have_vit = sc[q].have_vit != 0; wf[q].vfsc = sc[q].vfsc; }
    }
    // MSV / bias tests of every window (with the device's scores these are a few flops each; without, the host filters
    // run here, in parallel)
    std::vector<LtPrefilter> pf(nwin);
    {
      std::vector<std::pair<const BlockJob *, size_t>> flat0; flat0.reserve(nwin);
      for (const BlockJob &job : jobs) for (size_t w = 0; w < job.windows.size(); ++w) flat0.emplace_back(&job, w);
      std::vector<LtCounters> pc(nwin);
      host_parallel_for((int) nwin, wf.empty() ? cfg.host_threads : 1, [&](int q) {
        const BlockJob &job = *flat0[(size_t) q].first; const LtWindow &w = job.windows[flat0[(size_t) q].second];
        std::vector<uint8_t> sub;
        const uint8_t *subseq = nullptr;
        if (wf.empty()) {            // host filters need the residues
          sub.assign((size_t) w.length + 2, 255);
          const uint8_t *seq = seq_of(job);
          for (int64_t r = 1; r <= w.length; ++r) {
            const int64_t bp = w.n + r - 1;        // block position
            sub[(size_t) r] = job.strand == 0 ? seq[job.i + bp] : comp[seq[job.i + job.bn - bp + 1]];
          }
          subseq = sub.data();
        }
        pf[(size_t) q] = lt_window_prefilter(cfg, p, subseq, w.length, wf.empty() ? nullptr : &wf[(size_t) q], pc[(size_t) q]);
      });
      for (const LtCounters &c : pc) { ctr.n_past_msv += c.n_past_msv; ctr.n_past_bias += c.n_past_bias; ctr.pos_past_msv += c.pos_past_msv; ctr.pos_past_bias += c.pos_past_bias; }
    }
    tick("window scores + prefilter");
    // residues of a stretch of a block on its strand: out[1..len], sentinels around
    auto fetch = [&](const BlockJob &job, int64_t first_block_pos, int64_t len, std::vector<uint8_t> &outv) {
      outv.assign((size_t) len + 2, 255);
      const uint8_t *seq = seq_of(job);
      if (job.strand == 0) std::memcpy(outv.data() + 1, seq + job.i + first_block_pos, (size_t) len);
      else for (int64_t r = 0; r < len; ++r) outv[(size_t) r + 1] = comp[seq[job.i + job.bn - (first_block_pos + r) + 1]];
    };
    std::vector<std::pair<const BlockJob *, size_t>> flat; flat.reserve(nwin);
    for (const BlockJob &job : jobs) for (size_t w = 0; w < job.windows.size(); ++w) flat.emplace_back(&job, w);
    // long-target Viterbi scan of the windows that need it: on the device when there is one, else on the host workers
    std::vector<std::vector<LtWindow>> vit_of(nwin);
    {
      std::vector<int> need; std::vector<int> thr;
      for (size_t q = 0; q < nwin; ++q) if (pf[q].state == 2) { need.push_back((int) q); thr.push_back(pf[q].vit_thresh); }
      if (filters && !need.empty()) {
        std::vector<int> rec;            // (index into need, row, node) triples, sorted
        const int st = filters->viterbi(need.data(), thr.data(), need.size(), rec);
        if (st != P7X_OK) return st;
        for (size_t r = 0; r + 2 < rec.size(); r += 3) vit_of[(size_t) need[(size_t) rec[r]]].push_back(LtWindow{ rec[r + 1], rec[r + 2], 1 });
      } else if (!need.empty()) {
        host_parallel_for((int) need.size(), cfg.host_threads, [&](int z) {
          const size_t q = (size_t) need[(size_t) z];
          const BlockJob &job = *flat[q].first; const LtWindow &w = job.windows[flat[q].second];
          std::vector<uint8_t> sub;
          fetch(job, w.n, w.length, sub);
          lt_viterbi_longtarget(p, sub.data(), w.length, pf[q].filtersc_f2, cfg.F2, vit_of[q]);
        });
      }
      for (size_t q = 0; q < nwin; ++q) {
        const LtWindow &w = flat[q].first->windows[flat[q].second];
        if (pf[q].state == 2) lt_extend_and_merge(sd, max_length, w.length, 0.5f, vit_of[q]);
        else if (pf[q].state == 1) vit_of[q].assign(1, LtWindow{ 1, 0, w.length });
      }
    }
    tick("long-target Viterbi scan");
    // every Viterbi window: Forward (one device batch when there is a device), then Backward / domain definition for
    // the few that pass, on the host workers; hits stay in window order
    struct VitJob { size_t q; LtWindow vw; };
    std::vector<VitJob> vj;
    for (size_t q = 0; q < nwin; ++q) for (const LtWindow &vw : vit_of[q]) vj.push_back(VitJob{ q, vw });
    std::vector<float> fwd_dev;
    if (filters && !vj.empty()) {
      std::vector<LongTargetWindowRef> vrefs(vj.size());
      for (size_t z = 0; z < vj.size(); ++z) {
        const BlockJob &job = *flat[vj[z].q].first; const LtWindow &w = job.windows[flat[vj[z].q].second];
        const int64_t bp = w.n + vj[z].vw.n - 1;                // block position of the Viterbi window's first residue
        vrefs[z].strand = job.strand; vrefs[z].length = vj[z].vw.length;
        vrefs[z].start = (job.strand == 0 ? job.i + bp : job.i + job.bn - bp + 1) + offsets[job.t] - 1;
      }
      fwd_dev.resize(vj.size());
      const int st = filters->forward(dsq, comp, vrefs.data(), vrefs.size(), fwd_dev.data());
      if (st != P7X_OK) return st;
    }
    tick("Forward of the Viterbi windows");
    // the Forward filter of every Viterbi window (with the device's scores a bias-filter pass over the window and a few
    // flops; without a device the Forward parser itself runs here), then -- one device batch again -- the parsers' rows
    // and the region scan of the windows that passed, so that the host starts at the envelopes
    std::vector<std::vector<uint8_t>> subs(vj.size());
    std::vector<float> fwd_of(vj.size(), 0.0f);
    std::vector<char> pass(vj.size(), 0);
    auto geometry = [&](size_t z, const BlockJob *&job, int64_t &bp) {
      job = flat[vj[z].q].first; const LtWindow &w = job->windows[flat[vj[z].q].second];
      bp = w.n + vj[z].vw.n - 1;
    };
    host_parallel_for((int) vj.size(), cfg.host_threads, [&](int zi) {
      const size_t z = (size_t) zi;
      flogsum_init();
      const BlockJob *job; int64_t bp; geometry(z, job, bp);
      fetch(*job, bp, vj[z].vw.length, subs[z]);
      if (!fwd_dev.empty()) fwd_of[z] = fwd_dev[z];
      else {
        Model om{ &p, p.M, {} };
        om.prepare(); om.configure(true, (int) vj[z].vw.length);
        std::vector<float> fx;
        lt_forward_parser(om, subs[z].data(), (int) vj[z].vw.length, fx, &fwd_of[z]);
      }
      pass[z] = lt_forward_test(cfg, p, vj[z].vw.length, subs[z].data(), fwd_of[z]) ? 1 : 0;
      if (!pass[z]) std::vector<uint8_t>().swap(subs[z]);
    });
    tick("Forward filter of the Viterbi windows");
    std::vector<LongTargetWindowRegions> dev_regions;
    std::vector<int> dev_of(vj.size(), -1);
    if (filters) {
      std::vector<LongTargetWindowRef> prefs;
      for (size_t z = 0; z < vj.size(); ++z) if (pass[z]) {
        const BlockJob *job; int64_t bp; geometry(z, job, bp);
        LongTargetWindowRef r; r.strand = job->strand; r.length = vj[z].vw.length;
        r.start = (job->strand == 0 ? job->i + bp : job->i + job->bn - bp + 1) + offsets[job->t] - 1;
        dev_of[z] = (int) prefs.size();
        prefs.push_back(r);
      }
      if (!prefs.empty()) {
        const int st = filters->regions(dsq, comp, prefs.data(), prefs.size(), dev_regions);
        if (st != P7X_OK) return st;
      }
    }
    tick("parsers + region scan of the survivors");
    std::vector<std::vector<Hit>> jh(vj.size());
    std::vector<LtCounters> jc(vj.size());
    std::vector<int> jst(vj.size(), P7X_OK);
    // With a device and null2 on, the single-domain envelopes of all windows are rescored by the envelope kernel's
    // long-target instantiation in two batches (the second for the envelopes that are trimmed to their alignment):
    // phase A queues them, phase B builds their composition-adjusted odds and runs them, phase C makes the hits.
    // Whether that pays is a matter of numbers: the kernel's time is the latency of its longest envelope (two rounds of
    // ~19 ms for envelopes of a 1,203-node model, 1,024 of them at a time, plus ~10 ms of host work around them), the
    // host workers need ~7.5 ms of one thread for such an envelope (measured on the benchmark: 110 envelopes, 16
    // threads: 51.6 ms on the host, 49.8 ms through the device; before the envelope kernel's fences and waits were fixed
    // the two rounds took 53 ms).  A handful of hits stays with the host workers; more than a hundred per 16 threads, or
    // a repeat family with thousands of copies, go to the device.
    // cfg.host_envelopes: 1 always the host, 2 always the device (tests), 0 by that estimate.
    bool dev_env = filters != nullptr && lto.do_null2 && !dev_regions.empty() && cfg.host_envelopes != 1;
    if (dev_env && cfg.host_envelopes != 2) {
      size_t nsingle = 0;
      for (const LongTargetWindowRegions &w : dev_regions) for (const Region &r : w.regs) nsingle += r.multi ? 0 : 1;
      const int threads = cfg.host_threads > 0 ? cfg.host_threads : tophits_usable_cpus();
      const double host_ms = (double) nsingle * 7.5 * ((double) p.M / 1203.0) / (double) std::max(1, threads);
      const double dev_ms = 48.0 * (double) ((nsingle + 1023) / 1024);
      dev_env = nsingle > 0 && host_ms > dev_ms;
    }
    struct WinDD { DomainDefResult dd; std::vector<EnvelopeRequest> defer; bool active = false; };
    std::vector<WinDD> wdd(dev_env ? vj.size() : 0);
    host_parallel_for((int) vj.size(), cfg.host_threads, [&](int zi) {
      const size_t z = (size_t) zi;
      flogsum_init();
      const BlockJob *job; int64_t bp; geometry(z, job, bp);
      const LtWindow &vw = vj[z].vw;
      LtBlock blk{ nullptr, job->bn, job->strand == 0 ? job->i + 1 : job->i + job->bn, job->strand == 1 };
      jc[z].n_past_vit++; jc[z].pos_past_vit += (uint64_t) vw.length;
      if (!pass[z]) return;
      const LongTargetWindowRegions *dr = dev_of[z] >= 0 && (size_t) dev_of[z] < dev_regions.size() ? &dev_regions[(size_t) dev_of[z]] : nullptr;
      if (dev_env && dr && dr->n >= 0) {           // phase A: regions -> queued envelopes; multi-domain regions resolved here, on the host
        jc[z].n_past_fwd++; jc[z].pos_past_fwd += (uint64_t) vw.length;
        WinDD &w = wdd[z];
        t_long_target = &lto;
        int st = domaindef_from_regions(p, subs[z].data(), (int) vw.length, dr->nexpected, dr->regs.data(), dr->n, cfg.seed, cfg.seed != 0, w.dd, &w.defer, (int) z);
        if (st == P7X_OK) st = domaindef_finish_multi(p, subs[z].data(), (int) vw.length, cfg.seed, cfg.seed != 0, w.dd, nullptr, (int) z);
        t_long_target = nullptr;
        if (st != P7X_OK && st != P7X_ERANGE) jst[z] = st;
        w.active = st == P7X_OK;
        return;
      }
      jst[z] = lt_post_viterbi(cfg, p, lto, max_length, job->nres_at, blk, target_of(*job), bp, vw.length, subs[z].data(), fwd_of[z], dr, jh[z], jc[z]);
    });
    if (dev_env) {
      for (size_t z = 0; z < vj.size(); ++z) if (jst[z] != P7X_OK) return jst[z];
      // phase B: every queued envelope, two rounds
      struct EnvState { size_t z; int slot; int i, j; std::vector<float> rf; LongTargetEnvResult res; bool host = false, done = false; };
      std::vector<EnvState> es;
      for (size_t z = 0; z < vj.size(); ++z) if (wdd[z].active)
        for (size_t q = 0; q < wdd[z].defer.size(); ++q) { EnvState e; e.z = z; e.slot = (int) q; e.i = wdd[z].defer[q].i; e.j = wdd[z].defer[q].j; es.push_back(std::move(e)); }
      auto trace_of = [&](const LongTargetEnvResult &r, int i, Trace &tr) {
        tr.clear();
        for (size_t q = 0; q < r.ta.size(); ++q) tr.append((int) (r.ta[q] & 0xffu), (int) ((r.ta[q] >> 8) & 0xffffu), r.ti[q], r.tp[q]);
        tr.reverse();
        for (size_t q = 0; q < tr.st.size(); ++q) if (tr.i[q] > 0) tr.i[q] += i - 1;
      };
      std::vector<size_t> todo(es.size());
      for (size_t e = 0; e < es.size(); ++e) todo[e] = e;
      for (int round = 0; round < 2 && !todo.empty(); ++round) {
        host_parallel_for((int) todo.size(), cfg.host_threads, [&](int q) {
          EnvState &e = es[todo[(size_t) q]];
          reparameterize(p, lto, subs[e.z].data(), (int) vj[e.z].vw.length, e.i, e.j, e.rf);
        });
        std::vector<LongTargetEnvRequest> rq(todo.size());
        for (size_t q = 0; q < todo.size(); ++q) { const EnvState &e = es[todo[q]]; rq[q] = LongTargetEnvRequest{ dev_of[e.z], e.i, e.j, e.rf.data() }; }
        std::vector<LongTargetEnvResult> rs;
        const int st = filters->envelopes(rq.data(), rq.size(), rs);
        if (st != P7X_OK) return st;
        std::vector<size_t> next;
        std::vector<char> again(todo.size(), 0);
        host_parallel_for((int) todo.size(), cfg.host_threads, [&](int q) {
          EnvState &e = es[todo[(size_t) q]];
          e.res = std::move(rs[(size_t) q]);
          const int stt = e.res.status;
          if ((stt & 2) || (stt & 0xff & ~(3 | 64))) { e.done = true; return; }      // dropped (range error / traceback failure), as the host would
          if (stt & 64) { e.host = true; e.done = true; return; }                   // a near-tie: the host twin repeats this envelope from the start
          if (round == 0) {
            thread_local Trace tr;
            trace_of(e.res, e.i, tr);
            Domain d;
            make_alidisplay(p, tr, subs[e.z].data(), (int) vj[e.z].vw.length, d);
            if (e.i < d.sqfrom - lto.max_env_extra || e.j > d.sqto + lto.max_env_extra) {
              e.i = std::max<int>(e.i, (int) d.sqfrom - lto.max_env_extra);
              e.j = std::min<int>(e.j, (int) d.sqto + lto.max_env_extra);
              again[(size_t) q] = 1;
              return;
            }
          }
          e.done = true;
        });
        for (size_t q = 0; q < todo.size(); ++q) if (again[q]) next.push_back(todo[q]);
        todo.swap(next);
      }
      tick("envelopes on the device");
      // phase C: the windows' domains in order, then their hits
      std::vector<std::vector<size_t>> of_win(vj.size());
      for (size_t e = 0; e < es.size(); ++e) of_win[es[e].z].push_back(e);
      host_parallel_for((int) vj.size(), cfg.host_threads, [&](int zi) {
        const size_t z = (size_t) zi;
        if (!wdd[z].active) return;
        flogsum_init();
        WinDD &w = wdd[z];
        const BlockJob *job; int64_t bp; geometry(z, job, bp);
        const int L = (int) vj[z].vw.length;
        const uint8_t *dsq = subs[z].data();
        std::vector<Domain> kept;
        thread_local Workspace hws;
        thread_local Trace tr;
        for (Domain &d : w.dd.dcl) {
          if (d.deferred == -2) { for (Domain &m : w.dd.multi[(size_t) d.multi_slot]) kept.push_back(std::move(m)); continue; }
          if (d.deferred < 0) { kept.push_back(std::move(d)); continue; }
          EnvState &e = es[of_win[z][(size_t) d.deferred]];
          if (e.host) {                 // the reference's order of operations, from the untrimmed envelope
            Model om{ &p, p.M, {} };
            om.lt = &lto; om.prepare(); om.configure(false, L);
            DomainDefResult one; one.n2sc.assign((size_t) L + 1, 0.0f);
            if (rescore_isolated_domain(p, om, dsq, L, (int) d.ienv, (int) d.jenv, false, hws, one) == P7X_OK && !one.dcl.empty()) kept.push_back(std::move(one.dcl[0]));
            continue;
          }
          const int stt = e.res.status;
          if ((stt & 2) || (stt & 0xff & ~(3 | 64)) || e.res.ta.empty()) continue;
          trace_of(e.res, e.i, tr);
          Domain dom;
          make_alidisplay(p, tr, dsq, L, dom);
          dom.domcorrection = std::max(0.0f, e.res.orig - e.res.envsc);      // the score lost to the composition-adjusted background
          dom.ienv = e.i; dom.jenv = e.j; dom.envsc = e.res.orig; dom.oasc = e.res.oasc;
          dom.iali = dom.sqfrom; dom.jali = dom.sqto;
          kept.push_back(std::move(dom));
        }
        w.dd.dcl = std::move(kept);
        LtBlock blk{ nullptr, job->bn, job->strand == 0 ? job->i + 1 : job->i + job->bn, job->strand == 1 };
        jst[z] = lt_window_hits(cfg, p, max_length, job->nres_at, blk, target_of(*job), bp, w.dd, jh[z]);
      });
    }
    for (size_t z = 0; z < vj.size(); ++z) {
      if (jst[z] != P7X_OK) return jst[z];
      for (Hit &h : jh[z]) hits.push_back(std::move(h));
      ctr.n_past_vit += jc[z].n_past_vit; ctr.n_past_fwd += jc[z].n_past_fwd;
      ctr.pos_past_vit += jc[z].pos_past_vit; ctr.pos_past_fwd += jc[z].pos_past_fwd;
    }
  }
  tick("Backward + domain definition");
  lt_finish_tophits(cfg, p, cfg.evalue_window_length > 0 ? cfg.evalue_window_length : max_length, nres, (uint64_t) n, ctr, hits, out);
  tick("hit list");
  return P7X_OK;
}

} // namespace

// ---- what the device half (p7x_longtarget.hip) needs from here
// the F3 tie-breaker of p7x_tophits.cpp: Forward parser score in upstream's summation order, and the filter's null score
int host_forward_parser_exact(const Profile &p, const uint8_t *dsq1, int L, float *sc) { return forward_parser_striped(p, dsq1, L, sc); }
float host_filter_null_score(const Profile &p, const uint8_t *dsq1, int L, bool do_bias) { return do_bias ? lt_bias_filter(p, dsq1, L) : lt_null1(L); }

int longtarget_setup(const p7x_pipeline_cfg &cfg, const Profile &p, int *max_length, int *sc_thresh, int *xB)
{
  if (p.abc_type != P7X_DNA && p.abc_type != P7X_RNA) { set_error("long-target pipeline needs a nucleotide model"); return P7X_EINVAL; }
  *max_length = cfg.window_length > 0 ? cfg.window_length : p.max_length;
  if (*max_length <= 0) { set_error("model has no max_length (MAXL) and no window_length was given"); return P7X_EINVAL; }
  if (cfg.block_length <= *max_length) { set_error("block_length must exceed the model's max_length"); return P7X_EINVAL; }
  int tjb = 0;
  lt_ssv_threshold(p, *max_length, cfg.F1, sc_thresh, xB, &tjb);
  return P7X_OK;
}

const uint8_t *longtarget_complement(int abc_type) { return lt_complement_table(abc_type); }

void LongTargetUnits::count(const p7x_pipeline_cfg &cfg, int max_length, const int64_t *lengths, size_t n)
{
  W = cfg.block_length; C = max_length; strands = cfg.strands == P7X_STRAND_BOTH ? 2 : 1;
  part = cfg.lt_part; nparts = std::max(1, cfg.lt_nparts);
  total = 0;
  for (size_t t = 0; t < n; ++t) for (int64_t i = 0; i < lengths[t]; i += W - C) total += (uint64_t) strands;
}

void longtarget_finalize(p7x_tophits *th, int evalue_window, double res_count) { lt_finalize(th, evalue_window, res_count); }
double longtarget_res_count(const p7x_pipeline_cfg &cfg, uint64_t nres) { return lt_res_count(cfg, nres); }

void longtarget_seeds_from_rows(const Profile &p, const uint8_t *block_dsq, int64_t L, const LongTargetRow *rows, size_t nrows,
                                int sc_thresh, int xB, std::vector<int64_t> &seeds3)
{
  std::vector<LtRow> r(nrows);
  for (size_t i = 0; i < nrows; ++i) r[i] = LtRow{ rows[i].pos, rows[i].k, rows[i].sc };
  std::vector<LtWindow> w;
  lt_seeds_from_rows(p, block_dsq, L, r, sc_thresh, xB, w);
  for (const LtWindow &x : w) { seeds3.push_back(x.n); seeds3.push_back(x.k); seeds3.push_back(x.length); }
}

int longtarget_run_host(const p7x_pipeline_cfg &cfg, const p7x_oprofile *om, const uint8_t *dsq, const int64_t *offsets, const int64_t *lengths,
                        size_t n, const char *const *names, const char *const *accs, const char *const *descs,
                        const std::vector<LongTargetSeed> &seeds, p7x_tophits **out, LongTargetWindowScorer *filters)
{
  std::vector<LtSeedIn> in(seeds.size());
  for (size_t s = 0; s < seeds.size(); ++s) in[s] = LtSeedIn{ seeds[s].target, seeds[s].block_start, seeds[s].strand, LtWindow{ seeds[s].n, seeds[s].k, seeds[s].length } };
  return lt_run_host(cfg, om, dsq, offsets, lengths, n, names, accs, descs, in, out, filters);
}

} // namespace p7x

using namespace p7x;

extern "C" {

int p7x_forward_parser_exact(const p7x_oprofile *om, const uint8_t *dsq, int32_t L, float *sc)
{
  if (!om || !sc || L < 0 || (L > 0 && !dsq)) { set_error("p7x_forward_parser_exact: bad arguments"); return P7X_EINVAL; }
  return forward_parser_striped(om->p, dsq, L, sc);
}

int p7x_longtarget_from_seeds(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om,
                              const uint8_t *dsq, const int64_t *offsets, const int64_t *lengths, size_t n,
                              const char *const *names, const char *const *accs, const char *const *descs,
                              const int64_t *seed_target, const int64_t *seed_block, const int32_t *seed_strand,
                              const int64_t *seeds, size_t nseeds, p7x_tophits **out)
{
  if (!cfg || !om || !out || (n && (!dsq || !offsets || !lengths))) { set_error("p7x_longtarget_from_seeds: bad arguments"); return P7X_EINVAL; }
  std::vector<LtSeedIn> in(nseeds);
  for (size_t s = 0; s < nseeds; ++s)
    in[s] = LtSeedIn{ seed_target[s], seed_block[s], seed_strand[s], LtWindow{ seeds[3 * s], (int) seeds[3 * s + 1], seeds[3 * s + 2] } };
  return lt_run_host(*cfg, om, dsq, offsets, lengths, n, names, accs, descs, in, out, nullptr);
}

} // extern "C"
