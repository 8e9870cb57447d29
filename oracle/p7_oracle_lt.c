/* p7_oracle_lt.c -- TEST INFRASTRUCTURE ONLY (see p7_oracle.h).
 *
 * CPU restatement of the tail of HMMER 3.4's p7_Pipeline_LongTarget behind the SSV scan (reference
 * include/libhmmer/p7_pipeline.pxd:131-143; driven by LongTargetsPipeline._search_loop_longtarget,
 * src/pyhmmer/plan7.pyx:7541-7664), for ONE strand of ONE block of a target: dsq[1..L], the bottom strand already
 * reverse-complemented by the caller, as the reference hands it to the pipeline.  Plain scalar C, written apart from the
 * product's p7x_longtarget.inc.hpp and sharing no code with it; it builds on this directory's own filters
 * (p7o_ssv_longtarget, p7o_msv, p7o_bias_filter, p7o_fwd).  What it restates, from upstream p7_pipeline.c / p7_scoredata.c /
 * impl_sse/vitfilter.c / p7_domaindef.c:
 *
 *   p7_hmm_ScoreDataComputeRest   prefix / suffix length fractions of the model
 *   p7_pli_ExtendAndMergeWindows  SSV seeds -> windows (overlap 0), Viterbi seeds -> windows (overlap 0.5)
 *   (windows above 80 kb are cut into 40 kb pieces that overlap by max_length)
 *   p7_pli_postSSV_LongTarget     whole-window MSV and bias tests at F1, the bias scaled by B1 / window length
 *   p7_ViterbiFilter_longtarget   16-bit rows; a row whose best match cell reaches the score of P = F2 seeds a window per
 *                                 cell holding that score and is cleared
 *   p7_pli_postViterbi_LongTarget Forward parser per Viterbi window, bias scaled by B3 / window length, exponential tail at F3
 *   the long_target = TRUE scoring of an envelope (p7_Pipeline_LongTarget's per-domain block): length-model corrections for
 *   the residues of the window outside the envelope, the exponential tail
 *   reparameterize_model          the background mixed with the envelope's composition, smoothing 25 / min(100, max(50, n))
 *
 * The domains inside a Forward-passing window are defined by p7o_lt_domains (p7_oracle_dd.c).
 *
 * Pinned by: the reference's nhmmer tables through the product (tests/test_host_longtarget.py), and against the product's
 * device path on a 2 Mbp synthetic chromosome (tests/test_gpu_longtarget.py): stage counts equal, every hit inside a
 * window this file lets through Forward, scores re-derived from the hit's envelope / alignment coordinates to 1e-3 bit.
 */
#include "p7_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LT_LOG2 0.69314718055994529

typedef struct { int64_t n; int k; int64_t length; } LTW;            /* first residue, last node (seeds), residues */

static inline int sat16i(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
static float lt_null1(int64_t L) { float p1 = (float) L / (float) (L + 1); return (float) L * log(p1) + log(1. - p1); }   /* p7_bg_NullOne: C's double log() */

/* node k of the striped 4-lane float tables: vector q = (k-1) % Q, lane z = (k-1) / Q */
static float tf_at(const P7O_PROFILE *p, int t, int k)
{
  int Q = p->Q4, q = (k - 1) % Q, z = (k - 1) / Q;
  return t < 7 ? p->tfv[(q * 7 + t) * 4 + z] : p->tfv[(7 * Q + q) * 4 + z];
}

/* p7_hmm_ScoreDataComputeRest: per node the longest insert run that keeps a tail mass of 1e-7 (at least 2), normalised
 * over the model; prefix[k] = share of the model's length up to node k, suffix[k] = from node k on */
static void lt_scoredata(const P7O_PROFILE *p, float *prefix, float *suffix)
{
  int M = p->M;
  float sum = 0.0f;
  for (int k = 1; k < M; k++) {
    float tmi = tf_at(p, p7O_MI, k), tii = tf_at(p, p7O_II, k), len = 2.0f;
    if (tmi > 0.0f && tii > 0.0f && tii < 1.0f) len = 2.0f + (float) (int) (log(1e-7 / tmi) / log(tii));
    if (len < 2.0f) len = 2.0f;
    prefix[k] = len; sum += len;
  }
  prefix[M] = 1.0f; sum += 1.0f;
  for (int k = 1; k <= M; k++) prefix[k] /= sum;
  suffix[M] = prefix[M];
  for (int k = M - 1; k >= 1; k--) suffix[k] = suffix[k + 1] + prefix[k];
  for (int k = 2; k <= M; k++) prefix[k] += prefix[k - 1];
}

/* p7_pli_ExtendAndMergeWindows, one strand */
static int64_t lt_extend_merge(const float *prefix, const float *suffix, int max_length, int64_t L, float pct, LTW *w, int64_t n)
{
  if (n == 0) return 0;
  for (int64_t i = 0; i < n; i++) {
    int kfirst = (int) (w[i].k - w[i].length + 1); if (kfirst < 1) kfirst = 1;
    int64_t ws = w[i].n - (int64_t) (max_length * (0.1 + prefix[kfirst]));
    int64_t we = w[i].n + w[i].length + (int64_t) (max_length * (0.1 + suffix[w[i].k]));
    if (ws < 1) ws = 1;
    if (we > L) we = L;
    w[i].n = ws; w[i].length = we - ws + 1;
  }
  int64_t cnt = 0;
  for (int64_t i = 1; i < n; i++) {
    LTW *prev = &w[cnt], *cur = &w[i];
    int64_t os = prev->n > cur->n ? prev->n : cur->n;
    int64_t pe = prev->n + prev->length - 1, ce = cur->n + cur->length - 1;
    int64_t oe = pe < ce ? pe : ce;
    int64_t shorter = prev->length < cur->length ? prev->length : cur->length;
    if ((float) (oe - os + 1) / (float) shorter > pct) {
      int64_t ms = prev->n < cur->n ? prev->n : cur->n, me = pe > ce ? pe : ce;
      prev->n = ms; prev->length = me - ms + 1;
    } else { cnt++; w[cnt] = w[i]; }
  }
  return cnt + 1;
}

/* p7_ViterbiFilter_longtarget over dsq[1..L] (length model of L): returns the number of seeds, (row, node, 1) each */
static int64_t lt_viterbi_scan(const P7O_PROFILE *p, const uint8_t *dsq, int64_t L, float filtersc, double F2, LTW **out)
{
  int M = p->M, Q = p->Q8, N = Q * 8;
  int *T = (int *) malloc(sizeof(int) * 8 * (N + 1));
  int *blk = (int *) malloc(sizeof(int) * (N + 1) * 6);
  int *Mr = blk, *Ir = Mr + (N + 1), *Dr = Ir + (N + 1), *Mn = Dr + (N + 1), *In = Mn + (N + 1), *Dn = In + (N + 1);
  for (int q = 0; q < Q; q++) for (int z = 0; z < 8; z++) {
    int k = q + 1 + z * Q;
    for (int t = 0; t < 7; t++) T[t * (N + 1) + k] = p->twv[(q * 7 + t) * 8 + z];
    T[7 * (N + 1) + k] = p->twv[(7 * Q + q) * 8 + z];
  }
  /* the length model of this window (p7_oprofile_ReconfigRestLength): N, C, J move at log(3 / (L + 3)), loops cost nothing
   * (the 3-nat correction of the filter's score stands for them) */
  int xw_move = p->xw[p7O_N][p7O_MOVE];                       /* the caller configured the profile for this length */
  int xw_e_move = p->xw[p7O_E][p7O_MOVE], xw_e_loop = p->xw[p7O_E][p7O_LOOP];
  double invP = p->evparam[p7_VMU] - log(-1.0 * log(1.0 - F2)) / p->evparam[p7_VLAMBDA];      /* esl_gumbel_invsurv */
  /* xE >= thresh  <=>  ((xE + E->C + C->T) - base) / scale - 3.0 >= filtersc + ln2 * invP */
  int sc_thresh = (int) ceil(((filtersc + (float) (LT_LOG2 * invP) + 3.0) * p->scale_w) - (float) xw_e_move - (float) xw_move + (float) p->base_w);
  int64_t cap = 1024, ns = 0;
  LTW *seeds = (LTW *) malloc(sizeof(LTW) * cap);
  for (int k = 0; k <= N; k++) Mr[k] = Ir[k] = Dr[k] = -32768;
  int xN = p->base_w, xB = sat16i(xN + xw_move), xJ = -32768, xC = -32768;
  for (int64_t i = 1; i <= L; i++) {
    int x = dsq[i], xE = -32768;
    Mn[0] = In[0] = Dn[0] = -32768;
    for (int k = 1; k <= N; k++) {
      int q = (k - 1) % Q, z = (k - 1) / Q;
      int sv = sat16i(xB + T[p7O_BM * (N + 1) + k]), a;
      a = sat16i(Mr[k - 1] + T[p7O_MM * (N + 1) + k]); if (a > sv) sv = a;
      a = sat16i(Ir[k - 1] + T[p7O_IM * (N + 1) + k]); if (a > sv) sv = a;
      a = sat16i(Dr[k - 1] + T[p7O_DM * (N + 1) + k]); if (a > sv) sv = a;
      sv = sat16i(sv + p->rwv[((size_t) x * Q + q) * 8 + z]);
      Mn[k] = sv;
      if (k <= M && sv > xE) xE = sv;
      sv = sat16i(Mr[k] + T[p7O_MI * (N + 1) + k]);
      a = sat16i(Ir[k] + T[p7O_II * (N + 1) + k]);
      In[k] = a > sv ? a : sv;
    }
    if (xE >= sc_thresh) {                        /* a window per cell that holds the row's best score; the row is cleared */
      for (int k = 1; k <= M; k++) if (Mn[k] == xE) {
        if (ns == cap) { cap *= 2; seeds = (LTW *) realloc(seeds, sizeof(LTW) * cap); }
        seeds[ns].n = i; seeds[ns].k = k; seeds[ns].length = 1; ns++;
      }
      for (int k = 0; k <= N; k++) Mr[k] = Ir[k] = Dr[k] = -32768;
      continue;
    }
    { int b = xE + xw_e_move; if (b > xC) xC = sat16i(b); }
    { int b = xE + xw_e_loop; if (b > xJ) xJ = sat16i(b); }
    { int a = xJ + xw_move, b = xN + xw_move; xB = sat16i(a > b ? a : b); }
    Dn[1] = -32768;
    for (int k = 2; k <= N; k++) {
      int a = sat16i(Mn[k - 1] + T[p7O_MD * (N + 1) + k - 1]);
      int b = sat16i(Dn[k - 1] + T[p7O_DD * (N + 1) + k - 1]);
      Dn[k] = a > b ? a : b;
    }
    { int *t; t = Mr; Mr = Mn; Mn = t; t = Ir; Ir = In; In = t; t = Dr; Dr = Dn; Dn = t; }
  }
  free(T); free(blk);
  *out = seeds;
  return ns;
}

/* One strand block through the tail.  out: windows that passed the Forward test, [n][6] doubles: first residue, length (block
 * coordinates), Forward score, null1 score, bias-adjusted null score at F3 (nats), 0.  counts[8]: windows and residues past
 * MSV, bias, Viterbi, Forward (n_past_*, pos_past_* of p7_pipeline.pxd:88-101).  Returns the number of windows (may exceed cap). */
int64_t p7o_lt_block(P7O_PROFILE *p, const uint8_t *dsq, int64_t L, int max_length, double F1, double F2, double F3,
                     int B1, int B2, int B3, int do_bias, double *out, int64_t cap, uint64_t *counts)
{
  int M = p->M;
  int64_t nout = 0;
  float *prefix = (float *) calloc(M + 2, sizeof(float)), *suffix = (float *) calloc(M + 2, sizeof(float));
  lt_scoredata(p, prefix, suffix);
  int64_t scap = 1 << 16;
  int64_t *s3 = (int64_t *) malloc(sizeof(int64_t) * 3 * scap);
  int64_t nseeds = p7o_ssv_longtarget(p, dsq, L, max_length, F1, s3, scap);
  if (nseeds > scap) { scap = nseeds; s3 = (int64_t *) realloc(s3, sizeof(int64_t) * 3 * scap); nseeds = p7o_ssv_longtarget(p, dsq, L, max_length, F1, s3, scap); }
  LTW *w = (LTW *) malloc(sizeof(LTW) * (nseeds + 1));
  for (int64_t i = 0; i < nseeds; i++) { w[i].n = s3[3 * i]; w[i].k = (int) s3[3 * i + 1]; w[i].length = s3[3 * i + 2]; }
  free(s3);
  int64_t nw = lt_extend_merge(prefix, suffix, max_length, L, 0.0f, w, nseeds);
  /* merged windows above 80 kb are cut into 40 kb pieces that overlap by max_length */
  int64_t pcap = nw + 16, np = 0;
  LTW *pw = (LTW *) malloc(sizeof(LTW) * pcap);
  for (int64_t i = 0; i < nw; i++) {
    if (w[i].length <= 80000) { if (np == pcap) { pcap *= 2; pw = (LTW *) realloc(pw, sizeof(LTW) * pcap); } pw[np++] = w[i]; continue; }
    for (int64_t off = 0; off < w[i].length; off += 40000 - max_length) {
      int64_t len = w[i].length - off < 40000 ? w[i].length - off : 40000;
      if (np == pcap) { pcap *= 2; pw = (LTW *) realloc(pw, sizeof(LTW) * pcap); }
      pw[np].n = w[i].n + off; pw[np].k = 0; pw[np].length = len; np++;
      if (off + len >= w[i].length) break;
    }
  }
  free(w);
  for (int64_t wi = 0; wi < np; wi++) {
    const int64_t wlen = pw[wi].length;
    const uint8_t *sub = dsq + pw[wi].n - 1;                 /* sub[1..wlen] */
    float nullsc = lt_null1(wlen), usc = 0.0f;
    p7o_reconfig_length(p, (int) wlen);
    if (p7o_msv(p, sub, (int) wlen, &usc, NULL) != 0) usc = INFINITY;          /* overflow: passes */
    double P = p7o_gumbel_surv((usc - nullsc) / LT_LOG2, p->evparam[p7_MMU], p->evparam[p7_MLAMBDA]);
    if (P > F1) continue;
    counts[0]++; counts[4] += (uint64_t) wlen;
    float bias_sc = 0.0f, filtersc = nullsc;
    if (do_bias) {
      bias_sc = p7o_bias_filter(p, sub, (int) wlen) - nullsc;
      int64_t F1_L = wlen < B1 ? wlen : B1;
      filtersc = nullsc + bias_sc * ((float) F1_L / (float) wlen);
      P = p7o_gumbel_surv((usc - filtersc) / LT_LOG2, p->evparam[p7_MMU], p->evparam[p7_MLAMBDA]);
      if (P > F1) continue;
    }
    counts[1]++; counts[5] += (uint64_t) wlen;
    if (do_bias) { int64_t F2_L = wlen < B2 ? wlen : B2; filtersc = nullsc + bias_sc * ((float) F2_L / (float) wlen); }
    LTW *vs = NULL;
    int64_t nv = lt_viterbi_scan(p, sub, wlen, filtersc, F2, &vs);
    nv = lt_extend_merge(prefix, suffix, max_length, wlen, 0.5f, vs, nv);
    int64_t prev_end = 0;
    for (int64_t v = 0; v < nv; v++) {
      const int64_t vlen = vs[v].length, vstart = vs[v].n;
      int64_t overlap = prev_end - vstart + 1; if (overlap < 0 || v == 0) overlap = 0;
      counts[2]++; counts[6] += (uint64_t) (vlen - overlap);
      prev_end = vstart + vlen - 1;
      const uint8_t *vsub = sub + vstart - 1;
      float vnull = lt_null1(vlen), fwdsc = 0.0f, vbias = 0.0f;
      p7o_reconfig_length(p, (int) vlen);
      if (do_bias) vbias = p7o_bias_filter(p, vsub, (int) vlen) - vnull;
      if (p7o_fwd(p, vsub, (int) vlen, NULL, &fwdsc) != 0) continue;
      int64_t F3_L = vlen < B3 ? vlen : B3;
      float fsc = vnull + vbias * ((float) F3_L / (float) vlen);
      P = p7o_exp_surv((fwdsc - fsc) / LT_LOG2, p->evparam[p7_FTAU], p->evparam[p7_FLAMBDA]);
      if (P > F3) continue;
      counts[3]++; counts[7] += (uint64_t) (vlen - overlap);
      if (nout < cap) {
        double *o = out + 6 * nout;
        o[0] = (double) (pw[wi].n + vstart - 1); o[1] = (double) vlen; o[2] = fwdsc; o[3] = vnull; o[4] = fsc; o[5] = 0.0;
      }
      nout++;
    }
    free(vs);
  }
  free(pw); free(prefix); free(suffix);
  return nout;
}

/* The long_target = TRUE scoring of one envelope (p7_Pipeline_LongTarget's per-domain block): the envelope's Forward score
 * <envsc> was computed under a length model of the envelope's own length <env_len>; what that model charged for entering,
 * leaving and the flanks is taken out and the score re-expressed for a window of max_length (so that it does not depend on
 * how windows happened to merge), the null model is that of the same span, the bias is the envelope's correction, and the
 * exponential tail gives ln P for one window.  The rule is the one the reference's nhmmer tables pin (every printed score
 * and bias of bmyD1.tbl / bmyD2.tbl and the RF00001 answers, tests/test_host_longtarget.py).
 * Returns the bit score; *bias (bits), *lnP optional. */
float p7o_lt_domain_score(const P7O_PROFILE *p, int max_length, int64_t env_len, int64_t ali_len,
                          float envsc, float domcorrection, int do_null2, float *ret_bias_bits, double *ret_lnP)
{
  int64_t span = max_length > env_len ? max_length : env_len;
  float nullsc = lt_null1(span);
  float bitscore = envsc;
  bitscore -= 2 * log(2. / (env_len + 2)) + (env_len - ali_len) * log((float) env_len / (float) (env_len + 2));
  bitscore += 2 * log(2. / (max_length + 2));
  bitscore += (span - ali_len) * log((float) max_length / (float) (max_length + 2));
  float dom_bias = do_null2 ? domcorrection : 0.0f;
  float score = (bitscore - (nullsc + dom_bias)) / (float) LT_LOG2;
  if (ret_bias_bits) *ret_bias_bits = dom_bias / (float) LT_LOG2;
  if (ret_lnP) *ret_lnP = p7o_exp_logsurv(score, p->evparam[p7_FTAU], p->evparam[p7_FLAMBDA]);
  return score;
}

/* reparameterize_model: the background an envelope is rescored against when null2 is on -- the canonical frequencies
 * counted over the envelope env[0..n_env-1] (degenerate residues spread evenly over what they stand for), mixed with the
 * model's background at smoothing s = 25 / min(100, max(50, window_len)).  degen: [Kp][K] membership.  bg_out: K floats. */
void p7o_lt_envelope_background(const P7O_PROFILE *p, const uint8_t *env, int64_t n_env, int64_t window_len,
                                const uint8_t *degen, float *bg_out)
{
  int K = p->K, Kp = p->Kp;
  int64_t m = window_len < 50 ? 50 : window_len; if (m > 100) m = 100;
  float s = 25.0f / (float) m;
  float cnt[P7O_MAXK]; for (int x = 0; x < K; x++) cnt[x] = 0.0f;
  for (int64_t i = 0; i < n_env; i++) {
    int x = env[i];
    if (x < K) cnt[x] += 1.0f;
    else if (x > K && x <= Kp - 3) {
      int nd = 0; for (int y = 0; y < K; y++) nd += degen[x * K + y] ? 1 : 0;
      for (int y = 0; y < K; y++) if (degen[x * K + y]) cnt[y] += 1.0f / (float) nd;
    }
  }
  float tot = 0.0f; for (int x = 0; x < K; x++) tot += cnt[x];
  for (int x = 0; x < K; x++) bg_out[x] = (1.0f - s) * (tot > 0.0f ? cnt[x] / tot : p->bgf[x]) + s * p->bgf[x];
}

/* rescore_isolated_domain(long_target = TRUE) of one envelope env[1..n] cut from a window of <window_len> residues: Forward,
 * unihit, under a length model of the envelope's own length -- once with the profile's own match odds (*orig: the envelope
 * score proper) and once with odds re-derived for the background of p7o_lt_envelope_background (*adj); the bias correction
 * is max(0, orig - adj).  Canonical residues only (degenerate codes keep the profile's odds). */
int p7o_lt_envelope_scores(const P7O_PROFILE *p, const uint8_t *env, int n, int64_t window_len, const uint8_t *degen,
                           float *orig, float *adj)
{
  int M = p->M, K = p->K, Kp = p->Kp, Q = p->Q4;
  P7O_PROFILE q = *p;                                        /* shallow copy: only rfv and xf differ */
  float *rfv = (float *) malloc(sizeof(float) * (size_t) Kp * Q * 4);
  memcpy(rfv, p->rfv, sizeof(float) * (size_t) Kp * Q * 4);
  float pmove = 2.0f / ((float) n + 2.0f), ploop = 1.0f - pmove;         /* nj = 0 */
  q.xf[p7O_E][p7O_MOVE] = 1.0f; q.xf[p7O_E][p7O_LOOP] = 0.0f;
  q.xf[p7O_N][p7O_MOVE] = q.xf[p7O_J][p7O_MOVE] = q.xf[p7O_C][p7O_MOVE] = pmove;
  q.xf[p7O_N][p7O_LOOP] = q.xf[p7O_J][p7O_LOOP] = q.xf[p7O_C][p7O_LOOP] = ploop;
  int st = p7o_fwd(&q, env, n, NULL, orig);
  float bg[P7O_MAXK];
  p7o_lt_envelope_background(p, env + 1, n, window_len, degen, bg);
  for (int k = 1; k <= M; k++) {
    int qq = (k - 1) % Q, z = (k - 1) / Q;
    for (int x = 0; x < K; x++) {
      float prob = expf(p->msc[(size_t) x * (M + 1) + k]) * p->bgf[x];       /* the core model's match emission */
      rfv[((size_t) x * Q + qq) * 4 + z] = expf(logf(prob / bg[x]));
    }
  }
  q.rfv = rfv;
  int st2 = p7o_fwd(&q, env, n, NULL, adj);
  free(rfv);
  return st ? st : st2;
}
