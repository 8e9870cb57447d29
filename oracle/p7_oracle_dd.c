/* p7_oracle_dd.c -- TEST INFRASTRUCTURE ONLY (see p7_oracle.h).
 *
 * CPU restatement of HMMER 3.4's domain definition, p7_domaindef_ByPosteriorHeuristics (reference
 * include/libhmmer/p7_domaindef.pxd:23-72, p7_spensemble.pxd:3-39; reached from Pipeline._search_loop,
 * src/pyhmmer/plan7.pyx:6393-6453, through p7_Pipeline; the generator is re-seeded per region, plan7.pyx:5684-5688):
 *
 *   p7_DomainDecoding          posterior begin / end / occupancy totals from the parsers' special-state rows
 *   the region scan            rt1 = 0.25, rt2 = 0.10: where the occupancy rises and falls
 *   is_multidomain_region      rt3 = 0.20
 *   rescore_isolated_domain    unihit Forward / Backward over the envelope (upstream impl_sse/fwdback.c, odds space with
 *                              sparse rescaling), p7_Decoding, p7_Null2_ByExpectation, p7_OptimalAccuracy, p7_OATrace
 *   region_trace_ensemble      multihit Forward of the region, 200 x p7_StochasticTrace from Easel's fast generator
 *                              (esl_random.c: LCG, Jenkins-mixed seed; esl_rnd_FChoose on Kahan-normalised weights),
 *                              p7_trace_Index, p7_Null2_ByTrace per sampled domain, the per-residue null2 scores
 *   p7_spensemble_Cluster      single linkage (link_spsamples), clusters with posterior >= 0.25, consensus end points
 *   the scoring of a domain    p7_pipeline.c: envelope score + length correction - null1 - null2, in bits; exponential tail
 *   long targets               rescore_isolated_domain with long_target = TRUE (p7o_lt_domains), the alignment display's lines
 *                              (p7o_domain_alignment), the sequence's scores (seqout)
 *
 * Summation order: UPSTREAM's.  The full Forward / Backward matrices of envelopes and regions are computed by the striped vector
 * code of impl_sse/fwdback.c with do_full = TRUE (dd_forward / dd_backward below: SSE2 intrinsics, the same loops as the parsers
 * p7o_fwd / p7o_bck of p7_oracle.c, every row kept), the null2 sums stripe by stripe as impl_sse/null2.c forms them; the cells are
 * then held un-striped (node k at index k) for the routines that only READ them (decoding is elementwise, optimal accuracy is
 * max-plus over them, the tracebacks compare them).  So every posterior, every optimal-accuracy candidate and every sampled
 * traceback's choice is formed from the same floats as upstream's, and no decision here falls on "a tie of two summation
 * orders".  (Until round 5 this file summed in node order -- a third order beside upstream's and the product's.)  What this file
 * is for: a second, independently structured implementation of the logic -- thresholds, recursions, tie-break orders,
 * coordinate conventions, the order and number of the generator's draws -- pinned by the reference's own domain tables
 * (tests/golden/tables/ *.domtbl: all 53 rows, envelope / alignment / model coordinates exactly, scores and biases at print
 * precision) and compared with the product on thousands of synthetic targets: the product's host stage sums in upstream's order
 * too and must agree to the last bit of every score (tests/test_oracle_domains.py), the device path flags what its own order
 * cannot decide and hands it to that host code (tests/test_gpu_oracle_domains.py).  It shares no code with
 * pyhmmer_amd/csrc/p7x_domaindef.cpp.  Some of upstream's peculiarities were written wrongly at first from memory and
 * corrected after the tables (and the product, which reproduces them) disagreed -- four that the tables pin: two sampled domains link
 * when their START points OR their END points lie on nearby diagonals; p7_Null2_ByTrace counts an insert state's residue
 * in the match slot of its node; region_trace_ensemble counts a sampled domain's first residue as outside the domain; the
 * E state's draw multiplies the cells by a single-precision reciprocal -- and two that only models of a few nodes show (a
 * soak with 3-node models): link_spsamples' overlap on the model side is counted without the + 1 of the sequence side, so
 * sampled domains of fewer than five nodes never link and their regions yield no envelope; and of two clusters that
 * overlap by 80 % of the shorter one only the more probable is rescored.
 */
#include "p7_oracle.h"
#include <emmintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ---------------------------------------------------------------------------------------------------------------------
 * un-striped model: odds (probability / background), nodes 1..M
 * tfv holds, per striped position, (BM MM IM DM MD MI II) and then the DD's: the first four ENTER node k, the last four
 * LEAVE node k (impl_sse/p7_oprofile.pxd:41-49; p7_oprofile.c: fb_conversion). */
typedef struct {
  int M, K, Kp;
  float *bm, *mm, *im, *dm;        /* [M+2] into M_k from B, M_{k-1}, I_{k-1}, D_{k-1} */
  float *md, *mi, *ii, *dd;        /* [M+2] M_k->D_{k+1}, M_k->I_k, I_k->I_k, D_k->D_{k+1} */
  float *em;                       /* [Kp][M+1] match emission odds (insert odds are 1) */
  float nloop, nmove, cloop, cmove, jloop, jmove, eloop, emove;
  int Q;                           /* vectors per row of the striped layout, p7O_NQF(M) */
  const float *tfv;                /* the profile's striped transitions, [8Q][4] (borrowed) */
  float *rfs;                      /* [Kp][Q][4] match emission odds, striped (the profile's rfv, or <em> after dd_lt_adjust) */
} DDModel;

static void ddmodel_free(DDModel *m)
{
  free(m->bm); free(m->mm); free(m->im); free(m->dm); free(m->md); free(m->mi); free(m->ii); free(m->dd); free(m->em); free(m->rfs);
}

/* <em> into the striped layout (nodes beyond M: 0, as p7_oprofile_Convert pads) */
static void ddmodel_stripe_emissions(DDModel *m)
{
  const int M = m->M, Q = m->Q;
  for (int x = 0; x < m->Kp; x++)
    for (int q = 0; q < Q; q++)
      for (int z = 0; z < 4; z++) {
        const int k = z * Q + q + 1;
        m->rfs[((size_t) x * Q + q) * 4 + z] = (k <= M) ? m->em[(size_t) x * (M + 1) + k] : 0.0f;
      }
}

static int ddmodel_build(const P7O_PROFILE *p, int L, int multihit, DDModel *m)
{
  const int M = p->M, Q = p->Q4;
  memset(m, 0, sizeof(*m));
  m->M = M; m->K = p->K; m->Kp = p->Kp;
  float **t[8] = { &m->bm, &m->mm, &m->im, &m->dm, &m->md, &m->mi, &m->ii, &m->dd };
  for (int s = 0; s < 8; s++) { *t[s] = (float *) calloc((size_t) M + 2, sizeof(float)); if (!*t[s]) return -1; }
  m->em = (float *) calloc((size_t) p->Kp * (M + 1), sizeof(float));
  m->Q = Q; m->tfv = p->tfv;
  m->rfs = (float *) aligned_alloc(16, sizeof(float) * 4 * (size_t) p->Kp * (size_t) Q);
  if (!m->em || !m->rfs) return -1;
  for (int k = 1; k <= M; k++) {
    const int q = (k - 1) % Q, z = (k - 1) / Q;
    for (int s = 0; s < 7; s++) (*t[s])[k] = p->tfv[(size_t) (7 * q + s) * 4 + z];
    m->dd[k] = p->tfv[(size_t) (7 * Q + q) * 4 + z];
    for (int x = 0; x < p->Kp; x++) m->em[(size_t) x * (M + 1) + k] = p->rfv[(size_t) x * Q * 4 + (size_t) q * 4 + z];
  }
  /* p7_oprofile_ReconfigLength + ReconfigMultihit / ReconfigUnihit (impl_sse/p7_oprofile.c) */
  const float nj = multihit ? 1.0f : 0.0f;
  const float pmove = (2.0f + nj) / ((float) L + 2.0f + nj);
  const float ploop = 1.0f - pmove;
  m->nloop = m->cloop = m->jloop = ploop;
  m->nmove = m->cmove = m->jmove = pmove;
  m->eloop = multihit ? 0.5f : 0.0f;
  m->emove = multihit ? 0.5f : 1.0f;
  ddmodel_stripe_emissions(m);
  return 0;
}

/* full matrices of an envelope: rows 0..L, nodes 0..M (node 0 unused, kept at 0) */
typedef struct {
  int L, M;
  float *mx, *ix, *dx;      /* [(L+1)*(M+1)] */
  float *xE, *xN, *xJ, *xB, *xC, *scale;   /* [L+1] */
  int own_scales;           /* Backward left Forward's scale factors behind (has_own_scales) */
} DDMatrix;

static int ddmx_alloc(DDMatrix *x, int L, int M)
{
  memset(x, 0, sizeof(*x));
  x->L = L; x->M = M;
  const size_t n = (size_t) (L + 1) * (M + 1);
  x->mx = (float *) calloc(n, sizeof(float)); x->ix = (float *) calloc(n, sizeof(float)); x->dx = (float *) calloc(n, sizeof(float));
  float **s[6] = { &x->xE, &x->xN, &x->xJ, &x->xB, &x->xC, &x->scale };
  for (int i = 0; i < 6; i++) { *s[i] = (float *) calloc((size_t) L + 1, sizeof(float)); if (!*s[i]) return -1; }
  return (x->mx && x->ix && x->dx) ? 0 : -1;
}
static void ddmx_free(DDMatrix *x)
{
  free(x->mx); free(x->ix); free(x->dx); free(x->xE); free(x->xN); free(x->xJ); free(x->xB); free(x->xC); free(x->scale);
}
#define MX(x, i, k) ((x)->mx[(size_t) (i) * ((x)->M + 1) + (k)])
#define IX(x, i, k) ((x)->ix[(size_t) (i) * ((x)->M + 1) + (k)])
#define DX(x, i, k) ((x)->dx[(size_t) (i) * ((x)->M + 1) + (k)])

/* The envelope's matrices are kept un-striped (node k at index k), but every VALUE is formed by upstream's striped vector code:
 * the rows are computed in __m128 vectors of four stripes exactly as impl_sse/fwdback.c does (the same operations on the same
 * operands in the same order: vector q holds nodes q+1, q+1+Q, q+1+2Q, q+1+3Q; the D->D paths in up to four serial sweeps; the
 * row sums lane by lane, then (s0+s1)+(s2+s3)) and then scattered to their nodes.  So a cell here is bit for bit the cell of
 * upstream's P7_OMX, and every decision taken from the cells is upstream's decision. */
static inline __m128 dd_rightshift(__m128 a, __m128 b) { return _mm_move_ss(_mm_shuffle_ps(a, a, _MM_SHUFFLE(2, 1, 0, 0)), b); }
static inline __m128 dd_leftshift(__m128 a, __m128 zerov) { a = _mm_move_ss(a, zerov); return _mm_shuffle_ps(a, a, _MM_SHUFFLE(0, 3, 2, 1)); }
static inline float dd_hsum(__m128 a)
{ /* esl_sse_hsum_ps */
  float r;
  a = _mm_add_ps(a, _mm_shuffle_ps(a, a, _MM_SHUFFLE(0, 3, 2, 1)));
  a = _mm_add_ps(a, _mm_shuffle_ps(a, a, _MM_SHUFFLE(1, 0, 3, 2)));
  _mm_store_ss(&r, a);
  return r;
}
/* a striped row (Q vectors of [M D I]) to nodes 1..M of row i */
static void dd_scatter(const __m128 *dp, int Q, DDMatrix *x, int i)
{
  const int M = x->M;
  for (int q = 0; q < Q; q++) {
    float m[4], d[4], v[4];
    _mm_storeu_ps(m, dp[q * 3 + 0]); _mm_storeu_ps(d, dp[q * 3 + 1]); _mm_storeu_ps(v, dp[q * 3 + 2]);
    for (int z = 0; z < 4; z++) { const int k = z * Q + q + 1; if (k <= M) { MX(x, i, k) = m[z]; DX(x, i, k) = d[z]; IX(x, i, k) = v[z]; } }
  }
}

/* p7_Forward (impl_sse/fwdback.c forward_engine, do_full = TRUE): odds space; a row whose xE exceeds 1e4 is divided by it and
 * the logarithm of the divisor kept.  dsq[1..L].  Returns 0, or 1 when the score is not a finite number. */
static int dd_forward(const DDModel *m, const uint8_t *dsq, int L, DDMatrix *f, float *ret_sc)
{
  const int Q = m->Q;
  __m128 *dpc = (__m128 *) aligned_alloc(16, sizeof(__m128) * 3 * (size_t) Q);
  if (!dpc) { *ret_sc = -INFINITY; return 1; }
#define MMO(q) (dpc[(q)*3 + 0])
#define DMO(q) (dpc[(q)*3 + 1])
#define IMO(q) (dpc[(q)*3 + 2])
  const __m128 *tfv = (const __m128 *) m->tfv;
  const __m128 zerov = _mm_setzero_ps();
  float xN, xE, xB, xC, xJ, totscale = 0.0f;
  for (int q = 0; q < Q; q++) MMO(q) = IMO(q) = DMO(q) = zerov;
  xE = 0.f; xN = 1.f; xJ = 0.f; xB = m->nmove; xC = 0.f;
  f->xE[0] = xE; f->xN[0] = xN; f->xJ[0] = xJ; f->xB[0] = xB; f->xC[0] = xC; f->scale[0] = 1.0f;
  for (int i = 1; i <= L; i++) {
    const __m128 *rp = (const __m128 *) (m->rfs + (size_t) dsq[i] * Q * 4);
    const __m128 *tp = tfv;
    __m128 dcv = zerov, xEv = zerov, xBv = _mm_set1_ps(xB);
    __m128 mpv = dd_rightshift(MMO(Q-1), zerov);
    __m128 dpv = dd_rightshift(DMO(Q-1), zerov);
    __m128 ipv = dd_rightshift(IMO(Q-1), zerov);
    __m128 sv;
    int q, j;
    for (q = 0; q < Q; q++) {
      sv  =                _mm_mul_ps(xBv, *tp);  tp++;
      sv  = _mm_add_ps(sv, _mm_mul_ps(mpv, *tp)); tp++;
      sv  = _mm_add_ps(sv, _mm_mul_ps(ipv, *tp)); tp++;
      sv  = _mm_add_ps(sv, _mm_mul_ps(dpv, *tp)); tp++;
      sv  = _mm_mul_ps(sv, rp[q]);
      xEv = _mm_add_ps(xEv, sv);
      mpv = MMO(q); dpv = DMO(q); ipv = IMO(q);
      MMO(q) = sv;
      DMO(q) = dcv;
      dcv = _mm_mul_ps(sv, *tp); tp++;
      sv     =                _mm_mul_ps(mpv, *tp);  tp++;
      IMO(q) = _mm_add_ps(sv, _mm_mul_ps(ipv, *tp)); tp++;
    }
    dcv    = dd_rightshift(dcv, zerov);
    DMO(0) = zerov;
    tp     = tfv + 7*Q;
    for (q = 0; q < Q; q++) {
      DMO(q) = _mm_add_ps(dcv, DMO(q));
      dcv    = _mm_mul_ps(DMO(q), *tp); tp++;
    }
    if (m->M < 100) {
      for (j = 1; j < 4; j++) {
        dcv = dd_rightshift(dcv, zerov);
        tp  = tfv + 7*Q;
        for (q = 0; q < Q; q++) {
          DMO(q) = _mm_add_ps(dcv, DMO(q));
          dcv    = _mm_mul_ps(dcv, *tp); tp++;
        }
      }
    } else {
      for (j = 1; j < 4; j++) {
        __m128 cv = zerov;
        dcv = dd_rightshift(dcv, zerov);
        tp  = tfv + 7*Q;
        for (q = 0; q < Q; q++) {
          sv     = _mm_add_ps(dcv, DMO(q));
          cv     = _mm_or_ps(cv, _mm_cmpgt_ps(sv, DMO(q)));
          DMO(q) = sv;
          dcv    = _mm_mul_ps(dcv, *tp); tp++;
        }
        if (!_mm_movemask_ps(cv)) break;
      }
    }
    for (q = 0; q < Q; q++) xEv = _mm_add_ps(DMO(q), xEv);
    xE = dd_hsum(xEv);
    xN =  xN * m->nloop;
    xC = (xC * m->cloop) + (xE * m->emove);
    xJ = (xJ * m->jloop) + (xE * m->eloop);
    xB = (xJ * m->jmove) + (xN * m->nmove);
    if (xE > 1.0e4) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      xEv = _mm_set1_ps(1.0 / xE);
      for (q = 0; q < Q; q++) {
        MMO(q) = _mm_mul_ps(MMO(q), xEv);
        DMO(q) = _mm_mul_ps(DMO(q), xEv);
        IMO(q) = _mm_mul_ps(IMO(q), xEv);
      }
      f->scale[i] = xE;
      totscale += log(xE);
      xE = 1.0;
    } else f->scale[i] = 1.0f;
    f->xE[i] = xE; f->xN[i] = xN; f->xJ[i] = xJ; f->xB[i] = xB; f->xC[i] = xC;
    dd_scatter(dpc, Q, f, i);
  }
#undef MMO
#undef DMO
#undef IMO
  free(dpc);
  if (isnan(xC) || (L > 0 && xC == 0.0f) || isinf(xC)) { *ret_sc = -INFINITY; return 1; }
  *ret_sc = totscale + log(xC * m->cmove);
  return 0;
}

/* p7_Backward (impl_sse/fwdback.c backward_engine, do_full = TRUE): row i is divided by Forward's scale factor of row i -- or,
 * once a row's xB exceeds 1e16, by its own (own_scales; p7_Decoding then tracks the ratio of the two). */
static void dd_backward(const DDModel *m, const uint8_t *dsq, int L, const DDMatrix *f, DDMatrix *b)
{
  const int Q = m->Q;
  __m128 *buf = (__m128 *) aligned_alloc(16, sizeof(__m128) * 6 * (size_t) Q);
  __m128 *dpc = buf, *dpp = buf + 3 * Q;
#define MMOx(d,q) ((d)[(q)*3 + 0])
#define DMOx(d,q) ((d)[(q)*3 + 1])
#define IMOx(d,q) ((d)[(q)*3 + 2])
  const __m128 *tfv = (const __m128 *) m->tfv;
  const __m128 *tp, *rp;
  const __m128 zerov = _mm_setzero_ps();
  __m128 mpv, ipv, dpv, mcv, dcv, tmmv, timv, tdmv, xBv, xEv;
  float xN, xE, xB, xC, xJ, sc;
  int q, j;
  b->own_scales = 0;
  xJ = 0.f; xB = 0.f; xN = 0.f;
  xC = m->cmove;
  xE = xC * m->emove;
  xEv = _mm_set1_ps(xE);
  dcv = zerov;
  for (q = 0; q < Q; q++) MMOx(dpc,q) = DMOx(dpc,q) = xEv;
  for (q = 0; q < Q; q++) IMOx(dpc,q) = zerov;
  tp  = tfv + 8*Q - 1;
  dpv = dd_leftshift(DMOx(dpc,Q-1), zerov);
  for (q = Q-1; q >= 0; q--) {
    dcv = _mm_mul_ps(dpv, *tp); tp--;
    DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), dcv);
    dpv = DMOx(dpc,q);
  }
  for (j = 1; j < 4; j++) {
    tp  = tfv + 8*Q - 1;
    dcv = dd_leftshift(dcv, zerov);
    for (q = Q-1; q >= 0; q--) {
      dcv = _mm_mul_ps(dcv, *tp); tp--;
      DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), dcv);
    }
  }
  tp  = tfv + 7*Q - 3;
  dcv = dd_leftshift(DMOx(dpc,0), zerov);
  for (q = Q-1; q >= 0; q--) {
    MMOx(dpc,q) = _mm_add_ps(MMOx(dpc,q), _mm_mul_ps(dcv, *tp)); tp -= 7;
    dcv = DMOx(dpc,q);
  }
  sc = f->scale[L];
  if (sc > 1.0f) {
    xE = xE / sc; xN = xN / sc; xC = xC / sc; xJ = xJ / sc; xB = xB / sc;
    xEv = _mm_set1_ps(1.0 / sc);
    for (q = 0; q < Q; q++) {
      MMOx(dpc,q) = _mm_mul_ps(MMOx(dpc,q), xEv);
      DMOx(dpc,q) = _mm_mul_ps(DMOx(dpc,q), xEv);
      IMOx(dpc,q) = _mm_mul_ps(IMOx(dpc,q), xEv);
    }
  }
  b->scale[L] = sc;
  b->xE[L] = xE; b->xN[L] = xN; b->xJ[L] = xJ; b->xB[L] = xB; b->xC[L] = xC;
  dd_scatter(dpc, Q, b, L);

  for (int i = L-1; i >= 1; i--) {
    { __m128 *t = dpc; dpc = dpp; dpp = t; }   /* dpp = row i+1, dpc = row i (being built) */
    const __m128 *rfrow = (const __m128 *) (m->rfs + (size_t) dsq[i+1] * Q * 4);
    rp  = rfrow + Q-1;
    tp  = tfv + 7*Q - 1;
    tmmv = dd_leftshift(tfv[1], zerov);
    timv = dd_leftshift(tfv[2], zerov);
    tdmv = dd_leftshift(tfv[3], zerov);
    mpv = _mm_mul_ps(MMOx(dpp,0), rfrow[0]);
    mpv = dd_leftshift(mpv, zerov);
    xBv = zerov;
    for (q = Q-1; q >= 0; q--) {
      ipv = IMOx(dpp,q);
      IMOx(dpc,q) = _mm_add_ps(_mm_mul_ps(ipv, *tp), _mm_mul_ps(mpv, timv)); tp--;
      DMOx(dpc,q) =                                  _mm_mul_ps(mpv, tdmv);
      mcv         = _mm_add_ps(_mm_mul_ps(ipv, *tp), _mm_mul_ps(mpv, tmmv)); tp -= 2;
      mpv         = _mm_mul_ps(MMOx(dpp,q), *rp); rp--;
      MMOx(dpc,q) = mcv;
      tdmv = *tp; tp--;
      timv = *tp; tp--;
      tmmv = *tp; tp--;
      xBv = _mm_add_ps(xBv, _mm_mul_ps(mpv, *tp)); tp--;
    }
    xB = dd_hsum(xBv);
    xC =  xC * m->cloop;
    xJ = (xB * m->jmove) + (xJ * m->jloop);
    xN = (xB * m->nmove) + (xN * m->nloop);
    xE = (xC * m->emove) + (xJ * m->eloop);
    xEv = _mm_set1_ps(xE);
    tp  = tfv + 8*Q - 1;
    dpv = _mm_add_ps(DMOx(dpc,0), xEv);
    dpv = dd_leftshift(dpv, zerov);
    for (q = Q-1; q >= 0; q--) {
      dcv = _mm_mul_ps(dpv, *tp); tp--;
      DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), _mm_add_ps(dcv, xEv));
      dpv = DMOx(dpc,q);
      MMOx(dpc,q) = _mm_add_ps(MMOx(dpc,q), xEv);
    }
    for (j = 1; j < 4; j++) {
      dcv = dd_leftshift(dcv, zerov);
      tp  = tfv + 8*Q - 1;
      for (q = Q-1; q >= 0; q--) {
        dcv = _mm_mul_ps(dcv, *tp); tp--;
        DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), dcv);
      }
    }
    dcv = dd_leftshift(DMOx(dpc,0), zerov);
    tp  = tfv + 7*Q - 3;
    for (q = Q-1; q >= 0; q--) {
      MMOx(dpc,q) = _mm_add_ps(MMOx(dpc,q), _mm_mul_ps(dcv, *tp)); tp -= 7;
      dcv = DMOx(dpc,q);
    }
    if (xB > 1.0e16) b->own_scales = 1;
    if (b->own_scales) sc = (xB > 1.0e4) ? xB : 1.0;
    else               sc = f->scale[i];
    b->scale[i] = sc;
    if (sc > 1.0f) {
      xE /= sc; xN /= sc; xJ /= sc; xB /= sc; xC /= sc;
      xBv = _mm_set1_ps(1.0 / sc);
      for (q = 0; q < Q; q++) {
        MMOx(dpc,q) = _mm_mul_ps(MMOx(dpc,q), xBv);
        DMOx(dpc,q) = _mm_mul_ps(DMOx(dpc,q), xBv);
        IMOx(dpc,q) = _mm_mul_ps(IMOx(dpc,q), xBv);
      }
    }
    b->xE[i] = xE; b->xN[i] = xN; b->xJ[i] = xJ; b->xB[i] = xB; b->xC[i] = xC;
    dd_scatter(dpc, Q, b, i);
  }
  /* row 0: only B and N are reachable */
  if (L >= 1) {
    const __m128 *row1 = dpc;   /* after the loop dpc holds row 1 (row L when L == 1) */
    rp  = (const __m128 *) (m->rfs + (size_t) dsq[1] * Q * 4) + Q-1;
    tp  = tfv + 7*(Q-1);
    xBv = zerov;
    for (q = Q-1; q >= 0; q--) {
      mpv = _mm_mul_ps(MMOx(row1,q), *rp); rp--;
      mpv = _mm_mul_ps(mpv, *tp);          tp -= 7;
      xBv = _mm_add_ps(xBv, mpv);
    }
    xB = dd_hsum(xBv);
    xN = (xB * m->nmove) + (xN * m->nloop);
    b->xB[0] = xB; b->xC[0] = 0.0f; b->xJ[0] = 0.0f; b->xN[0] = xN; b->xE[0] = 0.0f; b->scale[0] = 1.0f;
  }
#undef MMOx
#undef DMOx
#undef IMOx
  free(buf);
}

/* p7_Decoding (impl_sse/decoding.c): posterior probabilities of the emitting states, written over <b>; delete states are
 * not decoded (0).  Returns 1 when the scale product is not finite (upstream: eslERANGE, the domain is dropped). */
static int dd_decoding(const DDModel *m, int L, const DDMatrix *f, DDMatrix *b)
{
  const int M = m->M;
  float scaleproduct = 1.0 / b->xN[0];
  /* the specials need Backward's row i and Forward's row i-1: walk upwards keeping Backward's values before overwriting */
  float *bN = (float *) malloc(sizeof(float) * (size_t) (L + 1)), *bJ = (float *) malloc(sizeof(float) * (size_t) (L + 1)),
        *bC = (float *) malloc(sizeof(float) * (size_t) (L + 1));
  if (!bN || !bJ || !bC) { free(bN); free(bJ); free(bC); return 1; }
  memcpy(bN, b->xN, sizeof(float) * (size_t) (L + 1)); memcpy(bJ, b->xJ, sizeof(float) * (size_t) (L + 1)); memcpy(bC, b->xC, sizeof(float) * (size_t) (L + 1));
  b->xE[0] = b->xN[0] = b->xJ[0] = b->xB[0] = b->xC[0] = 0.0f;
  for (int k = 0; k <= M; k++) { MX(b, 0, k) = IX(b, 0, k) = DX(b, 0, k) = 0.0f; }
  for (int i = 1; i <= L; i++) {
    const float totr = scaleproduct * f->scale[i];
    for (int k = 1; k <= M; k++) {
      MX(b, i, k) = MX(f, i, k) * MX(b, i, k) * totr;
      IX(b, i, k) = IX(f, i, k) * IX(b, i, k) * totr;
      DX(b, i, k) = 0.0f;
    }
    b->xE[i] = 0.0f; b->xB[i] = 0.0f;
    b->xN[i] = f->xN[i - 1] * bN[i] * m->nloop * scaleproduct;
    b->xJ[i] = f->xJ[i - 1] * bJ[i] * m->jloop * scaleproduct;
    b->xC[i] = f->xC[i - 1] * bC[i] * m->cloop * scaleproduct;
    if (b->own_scales) scaleproduct *= f->scale[i] / b->scale[i];
  }
  free(bN); free(bJ); free(bC);
  return (isinf(scaleproduct) || isnan(scaleproduct)) ? 1 : 0;
}

/* p7_Null2_ByExpectation (impl_sse/null2.c): the envelope's own residue composition as the posterior-weighted mean of the
 * emission odds of the states that explain it; null2[x], x < Kp, odds ratios.  degen[x*K + y] != 0: residue code x stands
 * for canonical residue y (esl_abc_FAvgScVec: a degenerate code gets the plain mean of its residues' ratios). */
/* the weighted sum over the nodes as the vector code forms it: four stripes, each its nodes in order (match term, then insert
 * term), then (s0 + s1) + (s2 + s3) */
static float dd_null2_sum(const DDModel *m, const float *wm, const float *wi, int x)
{
  const int M = m->M, Q = m->Q;
  const float *e = m->em + (size_t) x * (M + 1);
  float sv[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  for (int q = 0; q < Q; q++)
    for (int z = 0; z < 4; z++) {
      const int k = z * Q + q + 1;
      if (k > M) continue;                        /* padding: + 0 * 0, + 0 */
      sv[z] = sv[z] + wm[k] * e[k];
      sv[z] = sv[z] + wi[k];
    }
  return (sv[0] + sv[1]) + (sv[2] + sv[3]);
}

static void dd_null2_by_expectation(const DDModel *m, int L, const DDMatrix *pp, const uint8_t *degen, float *null2)
{
  const int M = m->M, K = m->K, Kp = m->Kp;
  float *wm = (float *) calloc((size_t) M + 1, sizeof(float)), *wi = (float *) calloc((size_t) M + 1, sizeof(float));
  float xN = 0.0f, xC = 0.0f, xJ = 0.0f;
  for (int i = 1; i <= L; i++) {
    for (int k = 1; k <= M; k++) { wm[k] += MX(pp, i, k); wi[k] += IX(pp, i, k); }
    xN += pp->xN[i]; xC += pp->xC[i]; xJ += pp->xJ[i];
  }
  const float norm = 1.0 / (float) L;
  for (int k = 1; k <= M; k++) { wm[k] *= norm; wi[k] *= norm; }
  xN *= norm; xC *= norm; xJ *= norm;
  const float xfactor = xN + xC + xJ;
  for (int x = 0; x < K; x++) null2[x] = dd_null2_sum(m, wm, wi, x) + xfactor;
  for (int x = K; x < Kp; x++) null2[x] = 1.0f;                 /* gap, *, ~ */
  for (int x = K + 1; x < Kp - 2; x++) {                         /* the degenerate codes */
    float sum = 0.0f; int n = 0;
    for (int y = 0; y < K; y++) if (degen[(size_t) x * K + y]) { sum += null2[y]; n++; }
    null2[x] = n ? sum / (float) n : 1.0f;
  }
  free(wm); free(wi);
}

/* p7_OptimalAccuracy (impl_sse/optacc.c): the alignment that maximises the expected number of correctly aligned
 * residues; a transition that the model does not have contributes 0 (the vector code masks it), impossible cells of row 0
 * are -infinity.  Written over <f>.  */
static float dd_optimal_accuracy(const DDModel *m, int L, const DDMatrix *pp, DDMatrix *ox)
{
  const int M = m->M;
  ox->xE[0] = -INFINITY; ox->xN[0] = 0.0f; ox->xJ[0] = -INFINITY; ox->xB[0] = 0.0f; ox->xC[0] = -INFINITY;
  for (int k = 0; k <= M; k++) { MX(ox, 0, k) = IX(ox, 0, k) = DX(ox, 0, k) = -INFINITY; }
  for (int i = 1; i <= L; i++) {
    float xE = -INFINITY;
    MX(ox, i, 0) = IX(ox, i, 0) = DX(ox, i, 0) = -INFINITY;
    for (int k = 1; k <= M; k++) {
      float sv = (m->bm[k] > 0.0f) ? ox->xB[i - 1] : 0.0f;
      if (k > 1) {
        sv = fmaxf(sv, (m->mm[k] > 0.0f) ? MX(ox, i - 1, k - 1) : 0.0f);
        sv = fmaxf(sv, (m->im[k] > 0.0f) ? IX(ox, i - 1, k - 1) : 0.0f);
        sv = fmaxf(sv, (m->dm[k] > 0.0f) ? DX(ox, i - 1, k - 1) : 0.0f);
      } else {                                    /* node 0 does not exist: the vector code shifts zeros in */
        sv = fmaxf(sv, 0.0f);
      }
      sv += MX(pp, i, k);
      MX(ox, i, k) = sv;
      xE = fmaxf(xE, sv);
      float iv = (m->mi[k] > 0.0f) ? MX(ox, i - 1, k) : 0.0f;
      iv = fmaxf(iv, (m->ii[k] > 0.0f) ? IX(ox, i - 1, k) : 0.0f);
      IX(ox, i, k) = iv + IX(pp, i, k);
      float dv = 0.0f;
      if (k > 1) {
        dv = (m->md[k - 1] > 0.0f) ? MX(ox, i, k - 1) : 0.0f;
        dv = fmaxf(dv, (m->dd[k - 1] > 0.0f) ? DX(ox, i, k - 1) : 0.0f);
      }
      DX(ox, i, k) = dv;
      xE = fmaxf(xE, dv);
    }
    ox->xE[i] = xE;
    float t1 = (m->jloop == 0.0f) ? 0.0f : ox->xJ[i - 1] + pp->xJ[i];
    float t2 = (m->eloop == 0.0f) ? 0.0f : xE;
    ox->xJ[i] = fmaxf(t1, t2);
    t1 = (m->cloop == 0.0f) ? 0.0f : ox->xC[i - 1] + pp->xC[i];
    t2 = (m->emove == 0.0f) ? 0.0f : xE;
    ox->xC[i] = fmaxf(t1, t2);
    ox->xN[i] = (m->nloop == 0.0f) ? 0.0f : ox->xN[i - 1] + pp->xN[i];
    t1 = (m->nmove == 0.0f) ? 0.0f : ox->xN[i];
    t2 = (m->jmove == 0.0f) ? 0.0f : ox->xJ[i];
    ox->xB[i] = fmaxf(t1, t2);
  }
  return (m->cmove == 0.0f) ? 0.0f : ox->xC[L];
}

/* p7_OATrace (impl_sse/optacc.c): back from C(L); a state's predecessor is the FIRST of its candidates, in upstream's
 * order, that holds the maximum (esl_vec_FArgMax); E looks for the best M or D cell of its row in the order in which the
 * striped vectors are scanned (vector q outer, then the four M's, then the four D's: node k = z Q + q + 1, strictly greater).
 * Only the unihit configuration is traced here (one domain).  Returns 0 and the first / last match state's residue and
 * node, or 1 when the trace holds no match state. */
enum { ST_M = 1, ST_I, ST_D, ST_B, ST_N, ST_C, ST_E, ST_S };
typedef struct { int n, cap; int *st, *k, *i; float *pp; } DDPath;     /* the states between B and E, last first */
static void dd_path_push(DDPath *p, int st, int k, int i, float pp)
{
  if (p && p->n < p->cap) { p->st[p->n] = st; p->k[p->n] = k; p->i[p->n] = i; p->pp[p->n] = pp; p->n++; }
  else if (p) p->n++;
}
static int dd_oa_trace_path(const DDModel *m, int L, const DDMatrix *pp, const DDMatrix *ox, int Q,
                            int *ia, int *ja, int *ka, int *kb, DDPath *path);
static int dd_oa_trace(const DDModel *m, int L, const DDMatrix *pp, const DDMatrix *ox, int Q,
                       int *ia, int *ja, int *ka, int *kb)
{
  return dd_oa_trace_path(m, L, pp, ox, Q, ia, ja, ka, kb, NULL);
}
static int dd_oa_trace_path(const DDModel *m, int L, const DDMatrix *pp, const DDMatrix *ox, int Q,
                            int *ia, int *ja, int *ka, int *kb, DDPath *path)
{
  const int M = m->M;
  int i = L, k = 0, st = ST_C;
  *ia = *ja = *ka = *kb = 0;
  long guard = 4L * (L + 2) * (M + 2) + 16;
  while (st != ST_S && guard-- > 0) {
    switch (st) {
    case ST_C: {
      const float t1 = (m->cloop == 0.0f) ? -INFINITY : ox->xC[i - 1] + pp->xC[i];
      const float t2 = (m->emove == 0.0f) ? -INFINITY : ox->xE[i];
      if (i < 1) return 1;
      if (t1 >= t2) { st = ST_C; i--; } else st = ST_E;      /* argmax over (C, E): C first */
      if (i == 0 && st == ST_C) return 1;                     /* C cannot begin a path */
      break;
    }
    case ST_E: {
      float best = -INFINITY; int bk = 0, bs = 0;
      for (int q = 0; q < Q; q++) {
        for (int z = 0; z < 4; z++) { const int kk = z * Q + q + 1; if (kk <= M && MX(ox, i, kk) > best) { best = MX(ox, i, kk); bk = kk; bs = ST_M; } }
        for (int z = 0; z < 4; z++) { const int kk = z * Q + q + 1; if (kk <= M && DX(ox, i, kk) > best) { best = DX(ox, i, kk); bk = kk; bs = ST_D; } }
      }
      if (!bs) return 1;
      st = bs; k = bk;
      break;
    }
    case ST_M: {
      if (*ja == 0) { *ja = i; *kb = k; }
      *ia = i; *ka = k;
      dd_path_push(path, ST_M, k, i, MX(pp, i, k));
      float path[4];
      path[0] = (k > 1 && m->mm[k] != 0.0f) ? MX(ox, i - 1, k - 1) : -INFINITY;
      path[1] = (k > 1 && m->im[k] != 0.0f) ? IX(ox, i - 1, k - 1) : -INFINITY;
      path[2] = (k > 1 && m->dm[k] != 0.0f) ? DX(ox, i - 1, k - 1) : -INFINITY;
      path[3] = (m->bm[k] != 0.0f) ? ox->xB[i - 1] : -INFINITY;
      if (k == 1) { path[0] = (m->mm[k] != 0.0f) ? 0.0f : -INFINITY; path[1] = (m->im[k] != 0.0f) ? 0.0f : -INFINITY; path[2] = (m->dm[k] != 0.0f) ? 0.0f : -INFINITY; }
      int arg = 0;
      for (int s = 1; s < 4; s++) if (path[s] > path[arg]) arg = s;
      i--;
      if (arg == 0) { st = ST_M; k--; } else if (arg == 1) { st = ST_I; k--; } else if (arg == 2) { st = ST_D; k--; } else st = ST_B;
      if (st != ST_B && k < 1) return 1;
      break;
    }
    case ST_D: {
      dd_path_push(path, ST_D, k, 0, 0.0f);
      const float p0 = (k > 1 && m->md[k - 1] != 0.0f) ? MX(ox, i, k - 1) : -INFINITY;
      const float p1 = (k > 1 && m->dd[k - 1] != 0.0f) ? DX(ox, i, k - 1) : -INFINITY;
      if (k <= 1) return 1;
      st = (p1 > p0) ? ST_D : ST_M; k--;
      break;
    }
    case ST_I: {
      dd_path_push(path, ST_I, k, i, IX(pp, i, k));
      const float p0 = (m->mi[k] != 0.0f) ? MX(ox, i - 1, k) : -INFINITY;
      const float p1 = (m->ii[k] != 0.0f) ? IX(ox, i - 1, k) : -INFINITY;
      st = (p1 > p0) ? ST_I : ST_M; i--;
      break;
    }
    case ST_B: st = ST_N; break;       /* unihit: J is not reachable */
    case ST_N: st = ST_S; break;       /* the N run back to row 0 emits nothing that is recorded here */
    default: return 1;
    }
  }
  return (*ia > 0 && guard > 0) ? 0 : 1;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * the heuristics on the parsers' rows (p7_domaindef.c) */

/* p7_DomainDecoding (impl_sse/decoding.c): fx / bx are the parsers' rows, (L+1) x [E N J B C SCALE] */
static void dd_domain_decoding(const P7O_PROFILE *p, int L, const float *fx, const float *bx, float *btot, float *etot, float *mocc)
{
  enum { E = 0, N = 1, J = 2, B = 3, C = 4, S = 5 };
  const float scaleproduct = 1.0f / bx[N];
  btot[0] = etot[0] = mocc[0] = 0.0f;
  for (int i = 1; i <= L; i++) {
    const float *f1 = fx + (size_t) (i - 1) * 6, *f = fx + (size_t) i * 6, *b1 = bx + (size_t) (i - 1) * 6, *b = bx + (size_t) i * 6;
    btot[i] = btot[i - 1] + f1[B] * b1[B] * f1[S] * scaleproduct;
    etot[i] = etot[i - 1] + f[E] * b[E] * f[S] * scaleproduct;
    float njcp = f1[N] * b[N] * p->xf[p7O_N][p7O_LOOP] * scaleproduct;
    njcp += f1[J] * b[J] * p->xf[p7O_J][p7O_LOOP] * scaleproduct;
    njcp += f1[C] * b[C] * p->xf[p7O_C][p7O_LOOP] * scaleproduct;
    mocc[i] = 1.0f - njcp;
  }
}

static int dd_is_multidomain(const float *btot, const float *etot, int i, int j, float rt3)
{
  float max = -1.0f;
  for (int z = i; z <= j; z++) {
    const float a = etot[z] - etot[i - 1], b = btot[j] - btot[z - 1];
    const float expected_n = a < b ? a : b;
    if (expected_n > max) max = expected_n;
  }
  return max >= rt3;
}

/* p7_FLogsum (logsum.c): log(e^a + e^b) through a table of log(1 + e^-x) at steps of 1/1000 nat, as upstream evaluates it */
static float dd_flogsum(float a, float b)
{
  static float table[16000];
  static int ready = 0;
  if (!ready) { for (int i = 0; i < 16000; i++) table[i] = (float) log(1.0 + exp((double) -i / 1000.0)); ready = 1; }
  const float hi = a > b ? a : b, lo = a > b ? b : a;
  return (lo == -INFINITY || (hi - lo) >= 15.7f) ? hi : hi + table[(int) ((hi - lo) * 1000.0f)];
}

/* ---------------------------------------------------------------------------------------------------------------------
 * regions that hold more than one domain: p7_domaindef.c region_trace_ensemble (reference p7_domaindef.pxd:23-59) */

/* Easel's fast generator (esl_random.c, eslRND_FAST): x <- 69069 x + 1, a draw is x / 2^32; esl_randomness_Init disperses the
 * seed with Jenkins' mix3 */
typedef struct { uint32_t x; } DDRng;
static uint32_t dd_mix3(uint32_t a, uint32_t b, uint32_t c)
{
  a -= b; a -= c; a ^= (c >> 13);  b -= c; b -= a; b ^= (a << 8);   c -= a; c -= b; c ^= (b >> 13);
  a -= b; a -= c; a ^= (c >> 12);  b -= c; b -= a; b ^= (a << 16);  c -= a; c -= b; c ^= (b >> 5);
  a -= b; a -= c; a ^= (c >> 3);   b -= c; b -= a; b ^= (a << 10);  c -= a; c -= b; c ^= (b >> 15);
  return c;
}
static void dd_rng_init(DDRng *r, uint32_t seed) { r->x = dd_mix3(seed, 87654321u, 12345678u); if (r->x == 0) r->x = 42; }
static double dd_random(DDRng *r) { r->x = r->x * 69069u + 1u; return (double) r->x / 4294967296.0; }

/* esl_vec_FNorm (its sum is Kahan-compensated, esl_vectorops.c) + esl_rnd_FChoose (esl_random.c: roll and running sum in
 * double -- a float roll could be 1.0 --, and when rounding leaves the roll above the last sum, uniform draws until one hits
 * a path of nonzero weight) */
static int dd_fchoose(DDRng *r, float *p, int n)
{
  float sum = 0.0f, c = 0.0f;
  for (int i = 0; i < n; i++) { const float y = p[i] - c, t = sum + y; c = (t - sum) - y; sum = t; }
  if (sum != 0.0f) for (int i = 0; i < n; i++) p[i] /= sum; else for (int i = 0; i < n; i++) p[i] = 1.0f / (float) n;
  const double roll = dd_random(r);
  double acc = 0.0;
  for (int i = 0; i < n; i++) { acc += p[i]; if (roll < acc) return i; }
  int i;
  do { i = (int) (dd_random(r) * n); } while (p[i] == 0.0f);
  return i;
}

/* one sampled domain of a traceback: residues ia..ja of the region, nodes ka..kb, and the states that emitted inside it */
typedef struct { int ia, ja, ka, kb; } DDSeg;

/* p7_StochasticTrace (impl_sse/stotrace.c) on the region's multihit Forward matrix, p7_trace_Index, and for every domain
 * of the trace p7_Null2_ByTrace (impl_sse/null2.c): cnt_m / cnt_i count the match / insert states the domain used.
 * segs: the trace's domains in sequence order.  n2acc[1..Lr] is bumped as region_trace_ensemble bumps ddef->n2sc. */
static int dd_sample_trace(const DDModel *m, const uint8_t *dsq /* region, 1..Lr */, int Lr, const DDMatrix *f, int Q, DDRng *r,
                           const uint8_t *degen, DDSeg *segs, int segcap, float *n2acc, float *cnt_m, float *cnt_i)
{
  const int M = m->M, K = m->K, Kp = m->Kp;
  /* the walk, from C at row Lr back to S; domains are met last to first */
  enum { tS, tN, tB, tM, tD, tI, tE, tJ, tC };
  int i = Lr, k = 0, sprv = tC, nseg = 0;
  /* the emitting states of the domain being walked: list of (state, k, i) is not needed, only the counts per domain, so the
   * counts are collected per domain on the way and committed when its B is reached */
  int cur_ja = 0, cur_kb = 0, cur_ia = 0, cur_ka = 0, in_dom = 0;
  /* per-domain emitting states: stored so that null2 can be applied after the whole trace is known (domains come out last
   * first, the accumulation into n2acc runs first to last over positions) */
  int *dom_state_k = (int *) malloc(sizeof(int) * (size_t) (2 * (Lr + M) + 8));       /* +k for M, -k for I */
  int *dom_first = (int *) malloc(sizeof(int) * (size_t) (segcap + 1));
  int nstates = 0;
  if (!dom_state_k || !dom_first) { free(dom_state_k); free(dom_first); return -1; }
  long guard = 8L * (Lr + 2) * (M + 2) + 64;
  while (sprv != tS && guard-- > 0) {
    int scur = -1;
    float path[4];
    switch (sprv) {
    case tM:
      path[0] = f->xB[i - 1] * m->bm[k];
      path[1] = (k > 1) ? MX(f, i - 1, k - 1) * m->mm[k] : 0.0f;
      path[2] = (k > 1) ? IX(f, i - 1, k - 1) * m->im[k] : 0.0f;
      path[3] = (k > 1) ? DX(f, i - 1, k - 1) * m->dm[k] : 0.0f;
      { static const int st[4] = { tB, tM, tI, tD }; scur = st[dd_fchoose(r, path, 4)]; }
      k--; i--;
      break;
    case tD:
      path[0] = MX(f, i, k - 1) * m->md[k - 1];
      path[1] = DX(f, i, k - 1) * m->dd[k - 1];
      scur = dd_fchoose(r, path, 2) == 0 ? tM : tD;
      k--;
      break;
    case tI:
      path[0] = MX(f, i - 1, k) * m->mi[k];
      path[1] = IX(f, i - 1, k) * m->ii[k];
      scur = dd_fchoose(r, path, 2) == 0 ? tM : tI;
      i--;
      break;
    case tN: scur = (i == 0) ? tS : tN; break;
    case tC:
      path[0] = f->xC[i - 1] * m->cloop;
      path[1] = f->xE[i] * m->emove * f->scale[i];
      scur = dd_fchoose(r, path, 2) == 0 ? tC : tE;
      break;
    case tJ:
      path[0] = f->xJ[i - 1] * m->jloop;
      path[1] = f->xE[i] * m->eloop * f->scale[i];
      scur = dd_fchoose(r, path, 2) == 0 ? tJ : tE;
      break;
    case tE: {
      /* E(i) from M(i,k) or D(i,k): one draw against the running sum, cells in the order of the striped vectors */
      const double roll = dd_random(r);
      const float norm = (float) (1.0 / f->xE[i]);
      double sum = 0.0;
      int tries = 0;
      while (scur < 0 && tries++ < 4) {
        for (int q = 0; q < Q && scur < 0; q++) {
          for (int z = 0; z < 4 && scur < 0; z++) { const int kk = z * Q + q + 1; if (kk <= M) { sum += MX(f, i, kk) * norm; if (roll < sum) { scur = tM; k = kk; } } }
          for (int z = 0; z < 4 && scur < 0; z++) { const int kk = z * Q + q + 1; if (kk <= M) { sum += DX(f, i, kk) * norm; if (roll < sum) { scur = tD; k = kk; } } }
        }
      }
      if (scur < 0) { free(dom_state_k); free(dom_first); return -1; }
      in_dom = 1; cur_ja = cur_kb = cur_ia = cur_ka = 0;
      if (nseg < segcap) dom_first[nseg] = nstates;
      break;
    }
    case tB:
      path[0] = f->xN[i] * m->nmove;
      path[1] = f->xJ[i] * m->jmove;
      scur = dd_fchoose(r, path, 2) == 0 ? tN : tJ;
      break;
    default: break;
    }
    if (scur < 0) { free(dom_state_k); free(dom_first); return -1; }
    /* the state just chosen sits at (k, i) */
    if (scur == tM) {
      if (cur_ja == 0) { cur_ja = i; cur_kb = k; }
      cur_ia = i; cur_ka = k;
      dom_state_k[nstates++] = k;
    } else if (scur == tI) {
      dom_state_k[nstates++] = -k;
    } else if (scur == tB && in_dom) {
      if (nseg < segcap) { segs[nseg].ia = cur_ia; segs[nseg].ja = cur_ja; segs[nseg].ka = cur_ka; segs[nseg].kb = cur_kb; }
      nseg++;
      in_dom = 0;
    }
    if ((scur == tN || scur == tJ || scur == tC) && scur == sprv) i--;
    sprv = scur;
  }
  if (guard <= 0 || nseg > segcap) { free(dom_state_k); free(dom_first); return -1; }
  dom_first[nseg] = nstates;
  /* first to last */
  for (int a = 0, b = nseg - 1; a < b; a++, b--) { DDSeg t = segs[a]; segs[a] = segs[b]; segs[b] = t; }
  int pos = 1;
  for (int d = 0; d < nseg; d++) {
    const int w = nseg - 1 - d;                                /* where this domain's states were stored */
    /* p7_Null2_ByTrace */
    for (int kk = 0; kk <= M; kk++) cnt_m[kk] = cnt_i[kk] = 0.0f;
    int Ld = 0;
    /* upstream works out whether the emitting state is a match or an insert state and then does not use it: an insert
     * state's residue is counted in the MATCH slot of its node, and the insert slots stay zero */
    for (int z = dom_first[w]; z < dom_first[w + 1]; z++) { const int v = dom_state_k[z]; cnt_m[v > 0 ? v : -v] += 1.0f; Ld++; }
    float null2[P7O_MAXKP];
    const float norm = 1.0 / (float) Ld;
    for (int kk = 1; kk <= M; kk++) { cnt_m[kk] *= norm; cnt_i[kk] *= norm; }
    for (int x = 0; x < K; x++) null2[x] = dd_null2_sum(m, cnt_m, cnt_i, x) + 0.0f;   /* no N / C / J residue inside a domain: xfactor = 0 */
    for (int x = K; x < Kp; x++) null2[x] = 1.0f;
    for (int x = K + 1; x < Kp - 2; x++) {
      float sum = 0.0f; int n = 0;
      for (int y = 0; y < K; y++) if (degen[(size_t) x * K + y]) { sum += null2[y]; n++; }
      null2[x] = n ? sum / (float) n : 1.0f;
    }
    /* residues outside the domains count 1, residues inside their null2 ratio (upstream's loop bounds: the domain's
     * first residue is still counted as outside) */
    for (; pos <= segs[d].ia; pos++) n2acc[pos] += 1.0f;
    for (; pos <= segs[d].ja; pos++) n2acc[pos] += null2[dsq[pos]];
  }
  for (; pos <= Lr; pos++) n2acc[pos] += 1.0f;
  free(dom_state_k); free(dom_first);
  return nseg;
}

/* p7_spensemble_Cluster (p7_spensemble.c): single-linkage clustering of the sampled domains (esl_cluster_SingleLinkage with
 * link_spsamples: overlap of at least min_overlap of the smaller one in the sequence AND in the model, and the start points
 * or the end points on nearby diagonals), then for every cluster that at least min_posterior of the samples take part in, consensus end points: the
 * leftmost start and the rightmost end that at least min_endpointp of its samples use. */
typedef struct { int i, j, k, m, idx; float prob; } DDCoord;
static int dd_linked(const DDCoord *a, const DDCoord *b)
{
  const float min_overlap = 0.8f; const int max_diagdiff = 4;
  int nov = (a->j < b->j ? a->j : b->j) - (a->i > b->i ? a->i : b->i) + 1;
  int la = a->j - a->i + 1, lb = b->j - b->i + 1;
  int n = la < lb ? la : lb;
  if ((float) nov / (float) n < min_overlap) return 0;
  nov = (a->m < b->m ? a->m : b->m) - (a->k > b->k ? a->k : b->k);        /* sic: no + 1 on the model side */
  la = a->m - a->k + 1; lb = b->m - b->k + 1;
  n = la < lb ? la : lb;
  if ((float) nov / (float) n < min_overlap) return 0;
  if (abs((a->i - a->k) - (b->i - b->k)) <= max_diagdiff) return 1;      /* the start points on nearby diagonals, */
  if (abs((a->j - a->m) - (b->j - b->m)) <= max_diagdiff) return 1;      /* or the end points */
  return 0;
}
static int dd_coord_by_start(const void *x, const void *y)
{
  const DDCoord *a = (const DDCoord *) x, *b = (const DDCoord *) y;
  return (a->i > b->i) - (a->i < b->i);
}
static int dd_cluster(const DDCoord *seg, int n, int nsamples, DDCoord *out, int outcap)
{
  const float min_posterior = 0.25f, min_endpointp = 0.02f;
  int *assign = (int *) malloc(sizeof(int) * (size_t) (n + 1)), *stack = (int *) malloc(sizeof(int) * (size_t) (n + 1));
  char *seen = (char *) calloc((size_t) nsamples + 1, 1);
  if (!assign || !stack || !seen) { free(assign); free(stack); free(seen); return -1; }
  for (int h = 0; h < n; h++) assign[h] = -1;
  int nc = 0;
  for (int h = 0; h < n; h++) {                 /* connected components */
    if (assign[h] >= 0) continue;
    int ns = 0; stack[ns++] = h; assign[h] = nc;
    while (ns > 0) {
      const int v = stack[--ns];
      for (int w = 0; w < n; w++) if (assign[w] < 0 && dd_linked(seg + v, seg + w)) { assign[w] = nc; stack[ns++] = w; }
    }
    nc++;
  }
  int nout = 0;
  for (int c = 0; c < nc; c++) {
    memset(seen, 0, (size_t) nsamples + 1);
    int ninc = 0, imin = 1 << 30, imax = 0, jmin = 1 << 30, jmax = 0, kmin = 1 << 30, kmax = 0, mmin = 1 << 30, mmax = 0;
    for (int h = 0; h < n; h++) if (assign[h] == c) {
      if (!seen[seg[h].idx]) { seen[seg[h].idx] = 1; ninc++; }
      if (seg[h].i < imin) imin = seg[h].i;
      if (seg[h].i > imax) imax = seg[h].i;
      if (seg[h].j < jmin) jmin = seg[h].j;
      if (seg[h].j > jmax) jmax = seg[h].j;
      if (seg[h].k < kmin) kmin = seg[h].k;
      if (seg[h].k > kmax) kmax = seg[h].k;
      if (seg[h].m < mmin) mmin = seg[h].m;
      if (seg[h].m > mmax) mmax = seg[h].m;
    }
    if ((float) ninc / (float) nsamples < min_posterior) continue;
    /* end points: histogram over the cluster's members, threshold in members-per-sample units */
    int span = imax - imin; if (jmax - jmin > span) span = jmax - jmin; if (kmax - kmin > span) span = kmax - kmin; if (mmax - mmin > span) span = mmax - mmin;
    int *epc = (int *) calloc((size_t) span + 2, sizeof(int));
    if (!epc) { free(assign); free(stack); free(seen); return -1; }
    int best[4];
    for (int which = 0; which < 4; which++) {
      const int lo = which == 0 ? imin : which == 1 ? jmin : which == 2 ? kmin : mmin;
      const int hi = which == 0 ? imax : which == 1 ? jmax : which == 2 ? kmax : mmax;
      memset(epc, 0, sizeof(int) * (size_t) (span + 2));
      for (int h = 0; h < n; h++) if (assign[h] == c) epc[(which == 0 ? seg[h].i : which == 1 ? seg[h].j : which == 2 ? seg[h].k : seg[h].m) - lo]++;
      int b = -1;
      if (which == 0 || which == 2) { for (int v = lo; v <= hi; v++) if ((float) epc[v - lo] / (float) ninc >= min_endpointp) { b = v; break; } }
      else                          { for (int v = hi; v >= lo; v--) if ((float) epc[v - lo] / (float) ninc >= min_endpointp) { b = v; break; } }
      if (b < 0) { int arg = 0; for (int v = 1; v <= hi - lo; v++) if (epc[v] > epc[arg]) arg = v; b = lo + arg; }
      best[which] = b;
    }
    free(epc);
    if (best[0] > best[1] || best[2] > best[3]) continue;
    if (nout < outcap) { out[nout].i = best[0]; out[nout].j = best[1]; out[nout].k = best[2]; out[nout].m = best[3]; out[nout].idx = c; out[nout].prob = (float) ninc / (float) nsamples; }
    nout++;
  }
  free(assign); free(stack); free(seen);
  if (nout > outcap) return -1;
  qsort(out, (size_t) nout, sizeof(DDCoord), dd_coord_by_start);
  /* p7_domaindef.c: of two clusters that overlap by at least 80 % of the shorter one only the more probable is kept */
  for (int d = 0; d < nout; d++) out[d].idx = 1;
  for (int d = 0; d < nout; d++)
    for (int d2 = d + 1; d2 < nout; d2++) {
      const int nov = (out[d].j < out[d2].j ? out[d].j : out[d2].j) - (out[d].i > out[d2].i ? out[d].i : out[d2].i) + 1;
      if (nov == 0) break;
      const int la = out[d].j - out[d].i + 1, lb = out[d2].j - out[d2].i + 1, n = la < lb ? la : lb;
      if ((float) nov / (float) n >= 0.8f) { if (out[d].prob > out[d2].prob) out[d2].idx = 0; else out[d].idx = 0; }
    }
  int kept = 0;
  for (int d = 0; d < nout; d++) if (out[d].idx) out[kept++] = out[d];
  return kept;
}

/* rescore_isolated_domain: the envelope i..j of the target; n2sc != NULL: the null2 log ratios of the target's residues are
 * already there (null2_is_done: an ensemble region), else they come from the envelope's own posterior expectation and are
 * left there.  Appends one row. */
static int dd_rescore(const P7O_PROFILE *p, const DDModel *uni, const uint8_t *dsq, int L, int i, int j, const uint8_t *degen, int do_null2,
                      float *n2sc, int null2_is_done, double *out, int64_t *nout, int64_t cap, double kind)
{
  const int Ld = j - i + 1;
  DDMatrix f, b;
  if (ddmx_alloc(&f, Ld, uni->M) != 0 || ddmx_alloc(&b, Ld, uni->M) != 0) { ddmx_free(&f); ddmx_free(&b); return -1; }
  float envsc = 0.0f;
  const uint8_t *sub = dsq + i - 1;                       /* sub[1..Ld] */
  const int bad = dd_forward(uni, sub, Ld, &f, &envsc);
  dd_backward(uni, sub, Ld, &f, &b);
  const int range = dd_decoding(uni, Ld, &f, &b);          /* b holds the posteriors now */
  int ok = 0;
  if (!bad && !range) {
    float null2[P7O_MAXKP];
    float domcorrection = 0.0f;
    if (do_null2) {
      if (!null2_is_done) {
        dd_null2_by_expectation(uni, Ld, &b, degen, null2);
        for (int pos = i; pos <= j; pos++) n2sc[pos] = logf(null2[dsq[pos]]);
      }
      for (int pos = i; pos <= j; pos++) domcorrection += n2sc[pos];
    }
    const float oasc = dd_optimal_accuracy(uni, Ld, &b, &f);     /* f holds the optimal-accuracy matrix now */
    int ia, ja, ka, kb;
    if (dd_oa_trace(uni, Ld, &b, &f, p->Q4, &ia, &ja, &ka, &kb) == 0) {
      ok = 1;
      if (*nout < cap) {
        double *o = out + *nout * 13;
        const float nullsc = p7o_null1(L), omega = 1.0f / 256.0f;
        /* p7_pipeline.c: the domain's bit score */
        /* (upstream's C: the logarithms are double, the sums are rounded to float where it assigns to floats) */
        float bitscore = envsc + (L - Ld) * log((float) L / (float) (L + 3));
        const float dombias = do_null2 ? dd_flogsum(0.0f, log(omega) + domcorrection) : 0.0f;
        bitscore = (bitscore - (nullsc + dombias)) / 0.69314718055994529;
        o[0] = i; o[1] = j; o[2] = ia + i - 1; o[3] = ja + i - 1; o[4] = ka; o[5] = kb;
        o[6] = envsc; o[7] = domcorrection; o[8] = oasc; o[9] = bitscore; o[10] = dombias / 0.69314718055994529;
        o[11] = p7o_exp_logsurv((double) bitscore, (double) p->evparam[p7_FTAU], (double) p->evparam[p7_FLAMBDA]);
        o[12] = kind;
      }
      (*nout)++;
    }
  }
  ddmx_free(&f); ddmx_free(&b);
  return ok;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * long targets: rescore_isolated_domain with long_target = TRUE (p7_domaindef.c; reference p7_domaindef.pxd:69-72 as called by
 * p7_Pipeline_LongTarget, p7_pipeline.pxd:131-143).  An envelope is scored unihit under a length model of its OWN length, with
 * null2 on against match emissions re-derived for a background mixed with the envelope's composition (reparameterize_model:
 * p7o_lt_envelope_background, p7_oracle_lt.c); when the envelope reaches further than max_env_extra residues beyond its
 * optimal-accuracy alignment it is cut back to that and aligned again; its score is then Forward's with the profile's own
 * odds, and its bias what the adjusted odds take from that. */
static int dd_lt_adjust(const P7O_PROFILE *p, DDModel *m, const uint8_t *first, int Ld, int64_t window_len, const uint8_t *degen)
{
  const int M = m->M, K = m->K, Kp = m->Kp;
  float bg[P7O_MAXK];
  p7o_lt_envelope_background(p, first, Ld, window_len, degen, bg);
  for (int k = 1; k <= M; k++) {
    float sc[P7O_MAXKP];
    for (int x = 0; x < K; x++) {
      const float prob = expf(p->msc[(size_t) x * (M + 1) + k]) * p->bgf[x];      /* the core model's match emission */
      sc[x] = logf(prob / bg[x]);
    }
    sc[K] = sc[Kp - 2] = sc[Kp - 1] = -INFINITY;
    for (int x = K + 1; x <= Kp - 3; x++) {                                        /* esl_abc_FExpectScVec under the new background */
      float num = 0.0f, den = 0.0f;
      for (int y = 0; y < K; y++) if (degen[(size_t) x * K + y]) { num += sc[y] * bg[y]; den += bg[y]; }
      sc[x] = num / den;
    }
    for (int x = 0; x < Kp; x++) m->em[(size_t) x * (M + 1) + k] = expf(sc[x]);
  }
  ddmodel_stripe_emissions(m);
  return 0;
}

static int dd_lt_rescore(const P7O_PROFILE *p, const uint8_t *win, int W, int i, int j, const uint8_t *degen, int do_null2, int max_env_extra,
                         double *out, int64_t *nout, int64_t cap, double kind)
{
  int ia = 0, ja = 0, ka = 0, kb = 0, ok = 0;
  float envsc = 0.0f, oasc = 0.0f;
  for (int pass = 0; pass < 2; pass++) {
    const int Ld = j - i + 1;
    DDModel m;
    DDMatrix f, b;
    if (ddmodel_build(p, Ld, 0, &m) != 0) return -1;
    if (do_null2) dd_lt_adjust(p, &m, win + i, Ld, W, degen);
    if (ddmx_alloc(&f, Ld, m.M) != 0 || ddmx_alloc(&b, Ld, m.M) != 0) { ddmx_free(&f); ddmx_free(&b); ddmodel_free(&m); return -1; }
    const uint8_t *sub = win + i - 1;
    const int bad = dd_forward(&m, sub, Ld, &f, &envsc);
    dd_backward(&m, sub, Ld, &f, &b);
    const int range = dd_decoding(&m, Ld, &f, &b);
    ok = 0;
    if (!bad && !range) {
      oasc = dd_optimal_accuracy(&m, Ld, &b, &f);
      int a1, a2;
      if (dd_oa_trace(&m, Ld, &b, &f, p->Q4, &a1, &a2, &ka, &kb) == 0) { ok = 1; ia = a1 + i - 1; ja = a2 + i - 1; }
    }
    ddmx_free(&f); ddmx_free(&b); ddmodel_free(&m);
    if (!ok) return 0;
    if (pass == 0 && (i < ia - max_env_extra || j > ja + max_env_extra)) {
      if (i < ia - max_env_extra) i = ia - max_env_extra;
      if (j > ja + max_env_extra) j = ja + max_env_extra;
      continue;
    }
    break;
  }
  float domcorrection = 0.0f;
  if (do_null2) {
    const int Ld = j - i + 1;
    DDModel m;
    DDMatrix f;
    float orig = 0.0f;
    if (ddmodel_build(p, Ld, 0, &m) != 0) return -1;
    if (ddmx_alloc(&f, Ld, m.M) != 0) { ddmx_free(&f); ddmodel_free(&m); return -1; }
    dd_forward(&m, win + i - 1, Ld, &f, &orig);
    ddmx_free(&f); ddmodel_free(&m);
    domcorrection = orig - envsc > 0.0f ? orig - envsc : 0.0f;
    envsc = orig;
  }
  if (*nout < cap) {
    double *o = out + *nout * 13;
    o[0] = i; o[1] = j; o[2] = ia; o[3] = ja; o[4] = ka; o[5] = kb; o[6] = envsc; o[7] = domcorrection; o[8] = oasc;
    o[9] = o[10] = o[11] = 0.0; o[12] = kind;
  }
  (*nout)++;
  return 1;
}

/* One window of a long target that passed the Forward test: win[1..W] on its strand; fx / bx: the parsers' rows of the window in
 * the multihit configuration of length W.  out: as p7o_domains, columns 0-8 and 12 (window coordinates; the scoring of a
 * long-target domain is p7o_lt_domain_score's).  counts as p7o_domains. */
int64_t p7o_lt_domains(P7O_PROFILE *p, const uint8_t *win, int W, const float *fx, const float *bx, const uint8_t *degen,
                       int do_null2, uint32_t seed, int max_env_extra, double *out, int64_t cap, int64_t *counts)
{
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
  const int nsamples = 200;
  int64_t nout = 0;
  for (int c = 0; c < 5; c++) counts[c] = 0;
  float *btot = (float *) calloc((size_t) W + 1, sizeof(float)), *etot = (float *) calloc((size_t) W + 1, sizeof(float)),
        *mocc = (float *) calloc((size_t) W + 1, sizeof(float));
  if (!btot || !etot || !mocc) { free(btot); free(etot); free(mocc); return -1; }
  dd_domain_decoding(p, W, fx, bx, btot, etot, mocc);
  DDModel multi;
  if (ddmodel_build(p, W, 1, &multi) != 0) { free(btot); free(etot); free(mocc); return -1; }
  int i = -1, triggered = 0, failed = 0;
  for (int j = 1; j <= W && !failed; j++) {
    if (!triggered) {
      if (mocc[j] - (btot[j] - btot[j - 1]) < rt2) i = j;
      else if (i == -1) i = j;
      if (mocc[j] >= rt1) triggered = 1;
    } else if (mocc[j] - (etot[j] - etot[j - 1]) < rt2) {
      counts[0]++;
      if (dd_is_multidomain(btot, etot, i, j, rt3)) {
        counts[2]++;
        if (seed != 0) {
          const int Lr = j - i + 1, M = multi.M;
          const uint8_t *sub = win + i - 1;
          DDMatrix f;
          float fsc;
          const int segcap = 4 + Lr;
          DDSeg *segs = (DDSeg *) malloc(sizeof(DDSeg) * (size_t) segcap);
          DDCoord *all = (DDCoord *) malloc(sizeof(DDCoord) * (size_t) nsamples * (size_t) segcap);
          DDCoord *cl = (DDCoord *) malloc(sizeof(DDCoord) * (size_t) (nsamples * 4 + 16));
          float *n2acc = (float *) calloc((size_t) Lr + 2, sizeof(float));
          float *cm = (float *) calloc((size_t) M + 1, sizeof(float)), *ci = (float *) calloc((size_t) M + 1, sizeof(float));
          if (ddmx_alloc(&f, Lr, M) != 0 || !segs || !all || !cl || !n2acc || !cm || !ci) failed = 1;
          int nall = 0;
          if (!failed) {
            dd_forward(&multi, sub, Lr, &f, &fsc);
            DDRng rng;
            dd_rng_init(&rng, seed);
            for (int t = 0; t < nsamples && !failed; t++) {
              const int ns = dd_sample_trace(&multi, sub, Lr, &f, p->Q4, &rng, degen, segs, segcap, n2acc, cm, ci);
              if (ns < 0) { failed = 1; break; }
              for (int d = 0; d < ns; d++) { all[nall].i = segs[d].ia + i - 1; all[nall].j = segs[d].ja + i - 1; all[nall].k = segs[d].ka; all[nall].m = segs[d].kb; all[nall].idx = t; nall++; }
            }
          }
          if (!failed) {
            const int nc = dd_cluster(all, nall, nsamples, cl, nsamples * 4 + 16);
            if (nc < 0) failed = 1;
            int last_j2 = 0;
            for (int d = 0; d < nc && !failed; d++) {
              counts[3]++;
              if (cl[d].i <= last_j2) counts[4]++;
              const int st = dd_lt_rescore(p, win, W, cl[d].i, cl[d].j, degen, do_null2, max_env_extra, out, &nout, cap, 1.0);
              if (st < 0) failed = 1; else if (st > 0) { last_j2 = cl[d].j; counts[1]++; }
            }
          }
          ddmx_free(&f);
          free(segs); free(all); free(cl); free(n2acc); free(cm); free(ci);
        }
      } else {
        const int st = dd_lt_rescore(p, win, W, i, j, degen, do_null2, max_env_extra, out, &nout, cap, 0.0);
        if (st < 0) failed = 1; else if (st > 0) counts[1]++;
      }
      i = -1; triggered = 0;
    }
  }
  ddmodel_free(&multi);
  free(btot); free(etot); free(mocc);
  return failed ? -1 : nout;
}

/* One target: dsq[1..L]; fx / bx: Forward / Backward parser rows of the whole target in the multihit configuration of
 * length L ((L+1) x 6 floats each, p7o_fwd / p7o_bck).  The profile <p> must be configured for L (p7o_reconfig_length).
 * seed: the pipeline's seed (the generator is re-seeded for every ensemble region, plan7.pyx:5684-5688); seed 0 or
 * ensembles == 0: regions that need the ensemble are counted and left out.
 * out: cap x 13 doubles per envelope, in the order the reference defines them:
 *   ienv jenv iali jali hmmfrom hmmto  envsc domcorrection oasc (nats / residues)  bitscore(bits) dombias(bits) lnP
 *   kind (0: a region that holds one domain, 1: a cluster of an ensemble region)
 * counts[0..4] = regions, envelopes, ensemble regions, clusters, overlapping clusters.
 * seqout (may be NULL), given the parser's Forward score fwdsc (nats): the sequence's bit score, its score before the null2
 * correction, the sum-of-domains score, ln P, the number of domains, their total length.
 * Returns the number of envelopes (may exceed cap: only cap are written), or -1. */
int64_t p7o_domains(P7O_PROFILE *p, const uint8_t *dsq, int L, const float *fx, const float *bx, const uint8_t *degen,
                    int do_null2, uint32_t seed, int ensembles, double *out, int64_t cap, int64_t *counts, float fwdsc, double *seqout)
{
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
  const int nsamples = 200;
  int64_t nout = 0;
  for (int c = 0; c < 5; c++) counts[c] = 0;
  float *btot = (float *) calloc((size_t) L + 1, sizeof(float)), *etot = (float *) calloc((size_t) L + 1, sizeof(float)),
        *mocc = (float *) calloc((size_t) L + 1, sizeof(float)), *n2sc = (float *) calloc((size_t) L + 2, sizeof(float));
  if (!btot || !etot || !mocc || !n2sc) { free(btot); free(etot); free(mocc); free(n2sc); return -1; }
  dd_domain_decoding(p, L, fx, bx, btot, etot, mocc);
  DDModel uni, multi;
  if (ddmodel_build(p, L, 0, &uni) != 0 || ddmodel_build(p, L, 1, &multi) != 0) { free(btot); free(etot); free(mocc); free(n2sc); return -1; }
  int i = -1, triggered = 0, failed = 0;
  for (int j = 1; j <= L && !failed; j++) {
    if (!triggered) {
      if (mocc[j] - (btot[j] - btot[j - 1]) < rt2) i = j;
      else if (i == -1) i = j;
      if (mocc[j] >= rt1) triggered = 1;
    } else if (mocc[j] - (etot[j] - etot[j - 1]) < rt2) {
      counts[0]++;
      if (dd_is_multidomain(btot, etot, i, j, rt3)) {
        counts[2]++;
        if (ensembles && seed != 0) {
          const int Lr = j - i + 1, M = multi.M;
          const uint8_t *sub = dsq + i - 1;
          DDMatrix f;
          float fsc;
          const int segcap = 4 + Lr;
          DDSeg *segs = (DDSeg *) malloc(sizeof(DDSeg) * (size_t) segcap);
          DDCoord *all = (DDCoord *) malloc(sizeof(DDCoord) * (size_t) nsamples * (size_t) segcap);
          DDCoord *cl = (DDCoord *) malloc(sizeof(DDCoord) * (size_t) (nsamples * 4 + 16));
          float *n2acc = (float *) calloc((size_t) Lr + 2, sizeof(float));
          float *cm = (float *) calloc((size_t) M + 1, sizeof(float)), *ci = (float *) calloc((size_t) M + 1, sizeof(float));
          if (ddmx_alloc(&f, Lr, M) != 0 || !segs || !all || !cl || !n2acc || !cm || !ci) failed = 1;
          int nall = 0;
          if (!failed) {
            dd_forward(&multi, sub, Lr, &f, &fsc);
            DDRng rng;
            dd_rng_init(&rng, seed);
            for (int t = 0; t < nsamples && !failed; t++) {
              const int ns = dd_sample_trace(&multi, sub, Lr, &f, p->Q4, &rng, degen, segs, segcap, n2acc, cm, ci);
              if (ns < 0) { failed = 1; break; }
              for (int d = 0; d < ns; d++) { all[nall].i = segs[d].ia + i - 1; all[nall].j = segs[d].ja + i - 1; all[nall].k = segs[d].ka; all[nall].m = segs[d].kb; all[nall].idx = t; nall++; }
            }
          }
          if (!failed) {
            for (int pos = 1; pos <= Lr; pos++) n2sc[i + pos - 1] = logf(n2acc[pos] / (float) nsamples);
            const int nc = dd_cluster(all, nall, nsamples, cl, nsamples * 4 + 16);
            if (nc < 0) failed = 1;
            int last_j2 = 0;
            for (int d = 0; d < nc && !failed; d++) {
              counts[3]++;
              if (cl[d].i <= last_j2) counts[4]++;
              const int st = dd_rescore(p, &uni, dsq, L, cl[d].i, cl[d].j, degen, do_null2, n2sc, 1, out, &nout, cap, 1.0);
              if (st < 0) failed = 1; else if (st > 0) { last_j2 = cl[d].j; counts[1]++; }
            }
          }
          ddmx_free(&f);
          free(segs); free(all); free(cl); free(n2acc); free(cm); free(ci);
        }
      } else {
        const int st = dd_rescore(p, &uni, dsq, L, i, j, degen, do_null2, n2sc, 0, out, &nout, cap, 0.0);
        if (st < 0) failed = 1; else counts[1]++;
      }
      i = -1; triggered = 0;
    }
  }
  if (seqout && !failed) {
    /* p7_pipeline.c, after domain definition: the sequence's score with the null2 correction of all its residues, or, if
     * that is higher, the sum of its domains that are worth more than their correction */
    const float nullsc = p7o_null1(L), omega = 1.0f / 256.0f;
    float seqbias = 0.0f;
    if (do_null2) {
      float sum = 0.0f, c = 0.0f;                                  /* esl_vec_FSum */
      for (int pos = 0; pos <= L; pos++) { const float y = n2sc[pos] - c, t = sum + y; c = (t - sum) - y; sum = t; }
      seqbias = dd_flogsum(0.0f, log(omega) + sum);
    }
    float pre_score = (fwdsc - nullsc) / 0.69314718055994529;
    float seq_score = (fwdsc - (nullsc + seqbias)) / 0.69314718055994529;
    float sum_score = 0.0f; int Ld = 0; seqbias = 0.0f;
    const int64_t nd = nout < cap ? nout : cap;
    for (int64_t d = 0; d < nd; d++) {
      const double *o = out + d * 13;
      if (!do_null2 || o[6] - o[7] > 0.0) { sum_score += (float) o[6]; Ld += (int) (o[1] - o[0] + 1); seqbias += (float) o[7]; }
    }
    seqbias = do_null2 ? dd_flogsum(0.0f, log(omega) + seqbias) : 0.0f;
    sum_score += (L - Ld) * log((float) L / (float) (L + 3));
    const float pre2_score = (sum_score - nullsc) / 0.69314718055994529;
    sum_score = (sum_score - (nullsc + seqbias)) / 0.69314718055994529;
    if (Ld > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2_score; }
    seqout[0] = seq_score; seqout[1] = pre_score; seqout[2] = sum_score;
    seqout[3] = p7o_exp_logsurv((double) seq_score, (double) p->evparam[p7_FTAU], (double) p->evparam[p7_FLAMBDA]);
    seqout[4] = (double) nout; seqout[5] = (double) Ld;
  }
  ddmodel_free(&uni); ddmodel_free(&multi);
  free(btot); free(etot); free(mocc); free(n2sc);
  return failed ? -1 : nout;
}

/* p7_alidisplay_Create (p7_alidisplay.c) for the one domain of an envelope's optimal-accuracy trace: the states between B
 * and E from the first to the last match state, one column each.  model: the consensus letter of the node for match and delete states, '.' for inserts; mline: the
 * model's letter when the residue is the consensus residue, '+' when its emission odds exceed 1, else a blank; aseq: the
 * residue, upper case in match columns, lower case in insert columns, '-' in delete columns; ppline: the posterior
 * probability of the state in tenths, '*' from 0.95, '.' for deletes.  consensus[1..M]; sym: the alphabet's symbols.
 * Returns the number of columns (the strings are terminated when they fit cap), or -1. */
int p7o_domain_alignment(P7O_PROFILE *p, const uint8_t *dsq, int L, int ienv, int jenv, const char *consensus, const char *sym,
                         char *model, char *mline, char *aseq, char *ppline, int cap)
{
  DDModel uni;
  if (ddmodel_build(p, L, 0, &uni) != 0) return -1;
  const int Ld = jenv - ienv + 1, M = uni.M;
  DDMatrix f, b;
  int ncol = -1;
  if (ddmx_alloc(&f, Ld, M) == 0 && ddmx_alloc(&b, Ld, M) == 0) {
    float envsc;
    const uint8_t *sub = dsq + ienv - 1;
    const int bad = dd_forward(&uni, sub, Ld, &f, &envsc);
    dd_backward(&uni, sub, Ld, &f, &b);
    const int range = dd_decoding(&uni, Ld, &f, &b);
    if (!bad && !range) {
      dd_optimal_accuracy(&uni, Ld, &b, &f);
      DDPath path;
      path.n = 0; path.cap = 2 * (Ld + M) + 8;
      path.st = (int *) malloc(sizeof(int) * (size_t) path.cap); path.k = (int *) malloc(sizeof(int) * (size_t) path.cap);
      path.i = (int *) malloc(sizeof(int) * (size_t) path.cap); path.pp = (float *) malloc(sizeof(float) * (size_t) path.cap);
      int ia, ja, ka, kb;
      if (path.st && path.k && path.i && path.pp && dd_oa_trace_path(&uni, Ld, &b, &f, p->Q4, &ia, &ja, &ka, &kb, &path) == 0 && path.n <= path.cap) {
        /* the display runs from the first to the last match state: an optimal-accuracy trace may leave through a run of
         * delete states (they cost nothing, and E takes the first best cell of its row) */
        int zlast = 0, zfirst = path.n - 1;
        while (zlast < path.n && path.st[zlast] != ST_M) zlast++;
        while (zfirst >= 0 && path.st[zfirst] != ST_M) zfirst--;
        ncol = zfirst - zlast + 1;
        for (int c = 0; c < ncol && c < cap - 1; c++) {
          const int z = zfirst - c;                         /* the path was recorded last state first */
          const int st = path.st[z], k = path.k[z];
          if (st == ST_M) {
            const int x = sub[path.i[z]];
            const char cons = consensus[k];
            const char up = (cons >= 'a' && cons <= 'z') ? (char) (cons - 'a' + 'A') : cons;
            model[c] = cons;
            mline[c] = (sym[x] == up) ? cons : (uni.em[(size_t) x * (M + 1) + k] > 1.0f ? '+' : ' ');
            aseq[c] = sym[x];
          } else if (st == ST_I) {
            const int x = sub[path.i[z]];
            const char lo = (sym[x] >= 'A' && sym[x] <= 'Z') ? (char) (sym[x] - 'A' + 'a') : sym[x];
            model[c] = '.'; mline[c] = ' '; aseq[c] = lo;
          } else {
            model[c] = consensus[k]; mline[c] = ' '; aseq[c] = '-';
          }
          const float pr = path.pp[z];
          ppline[c] = (st == ST_D) ? '.' : ((pr + 0.05f >= 1.0f) ? '*' : (char) ((int) ((pr + 0.05f) * 10.0f) + '0'));
        }
        const int end = ncol < cap - 1 ? ncol : cap - 1;
        model[end] = mline[end] = aseq[end] = ppline[end] = 0;
      }
      free(path.st); free(path.k); free(path.i); free(path.pp);
    }
  }
  ddmx_free(&f); ddmx_free(&b);
  ddmodel_free(&uni);
  return ncol;
}
