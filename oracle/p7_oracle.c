/* p7_oracle.c -- TEST INFRASTRUCTURE ONLY (see p7_oracle.h).
 *
 * CPU restatement of the HMMER 3.4 `p7_Pipeline` acceleration cascade as the
 * reference's impl_sse build executes it.  Upstream sources are absent from
 * /root/reference (empty submodules), so each function names the upstream
 * routine it restates and the reference line that calls / declares it.
 *
 * Parity status: PINNED by the reference's fixtures (tests/test_oracle_*.py):
 *   striped tables vs tests/golden/db/ *.h3f|*.h3p  (bit-exact, incl. float),
 *   cascade survivors vs tests/golden/tables/ *.tbl  (identical hit lists).
 * Raw integer filter scores are not recorded by any reference fixture
 * (SURVEY.md 8c); they are pinned only through those end-to-end results.
 *
 * Build: see oracle/Makefile (gcc -O2 -msse2 -ffp-contract=off).
 */
#include "p7_oracle.h"
#include <emmintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define LOG2   0.69314718055994529
#define LOG2R  1.44269504088896341
#define SMALLX1 5e-9

static void *amalloc(size_t n) { void *p = NULL; if (posix_memalign(&p, 16, n ? n : 16)) return NULL; memset(p, 0, n); return p; }

/* ---------------------------------------------------------------- alphabet
 * Easel digital alphabets; reference src/pyhmmer/easel.pyx:313-347.
 * amino "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~" (K=20,Kp=29); DNA/RNA "ACGT-RYMKSWHBVDN*~" (K=4,Kp=18).
 */
static void degen_matrix(int K, int Kp, unsigned char dg[P7O_MAXKP][P7O_MAXK])
{
  memset(dg, 0, P7O_MAXKP * P7O_MAXK);
  for (int x = 0; x < K; x++) dg[x][x] = 1;
  if (K == 20) {
    const char *sym = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
    const char *sets[29] = {0};
    sets[21] = "ND"; sets[22] = "IL"; sets[23] = "QE"; sets[24] = "K"; sets[25] = "C";
    for (int x = 21; x <= 25; x++)
      for (const char *c = sets[x]; *c; c++) dg[x][(int)(strchr(sym, *c) - sym)] = 1;
    for (int y = 0; y < K; y++) dg[26][y] = 1;                        /* X */
  } else {
    const char *sym = "ACGT-RYMKSWHBVDN*~";
    const char *sets[18] = {0};
    sets[5] = "AG"; sets[6] = "CT"; sets[7] = "AC"; sets[8] = "GT"; sets[9] = "CG"; sets[10] = "AT";
    sets[11] = "ACT"; sets[12] = "CGT"; sets[13] = "ACG"; sets[14] = "AGT"; sets[15] = "ACGT";
    for (int x = 5; x <= 15; x++)
      for (const char *c = sets[x]; *c; c++) dg[x][(int)(strchr(sym, *c) - sym)] = 1;
  }
  (void)Kp;
}

/* ---------------------------------------------------------------- esl_sse_expf
 * Easel esl_sse.c: Cephes-derived vector expf used by fb_conversion
 * (reference include/libhmmer/impl_sse/p7_oprofile.pxd:121 -> p7_oprofile_Convert).
 * Scalar restatement of the per-lane operations, IEEE single, no contraction.
 */
float p7o_sse_expf_scalar(float x)
{
  static const float cp[6] = { 1.9875691500E-4f, 1.3981999507E-3f, 8.3334519073E-3f,
                               4.1665795894E-2f, 1.6666665459E-1f, 5.0000001201E-1f };
  static const float cc[2] = { 0.693359375f, -2.12194440e-4f };
  static const float maxlogf =  88.72283905206835f;
  static const float minlogf = -103.27892990343185f;
  int is_max = (x >  maxlogf);
  int is_min = (x <= minlogf);
  volatile float fx, tmp, z, y, xx = x;
  int k;
  if (isnan(x)) return x;
  fx = xx * (float) LOG2R;
  fx = fx + 0.5f;
  if (!(fx > -2.0e9f && fx < 2.0e9f)) { return is_max ? INFINITY : 0.0f; }   /* cvttps would give INT_MIN; result masked anyway */
  k   = (int) fx;                 /* truncation */
  tmp = (float) k;
  if (tmp > fx) tmp = tmp - 1.0f; /* floor */
  fx  = tmp;
  k   = (int) fx;
  tmp = fx * cc[0];
  z   = fx * cc[1];
  xx  = xx - tmp;
  xx  = xx - z;
  z   = xx * xx;
  y = cp[0];        y = y * xx;
  y = y + cp[1];    y = y * xx;
  y = y + cp[2];    y = y * xx;
  y = y + cp[3];    y = y * xx;
  y = y + cp[4];    y = y * xx;
  y = y + cp[5];    y = y * z;
  y = y + xx;
  y = y + 1.0f;
  {
    union { int32_t i; float f; } u;
    u.i = (int32_t)((uint32_t)(k + 127) << 23);
    y = y * u.f;
  }
  if (is_max) return INFINITY;
  if (is_min) return 0.0f;
  return y;
}

void p7o_expf_neg(const double *in, float *out, size_t n)
{ /* upstream p7_hmmfile.c read_asc30hmm: '*' -> 0.0, else expf(-1.0 * atof(tok)) */
  for (size_t i = 0; i < n; i++) out[i] = isinf(in[i]) ? 0.0f : expf((float)(-1.0 * in[i]));
}

/* ---------------------------------------------------------------- profile config
 * upstream modelconfig.c p7_ProfileConfig / p7_ReconfigLength, local multihit mode
 * (reference include/libhmmer/modelconfig.pxd:7-10; called plan7.pyx:8082).
 */
static void calculate_occupancy(int M, const float *t, float *mocc)
{ /* upstream p7_hmm.c p7_hmm_CalculateOccupancy */
  mocc[0] = 0.f;
  mocc[1] = t[0*7+p7H_MI] + t[0*7+p7H_MM];
  for (int k = 2; k <= M; k++)
    mocc[k] = mocc[k-1] * (t[(k-1)*7+p7H_MM] + t[(k-1)*7+p7H_MI]) + (1.0 - mocc[k-1]) * t[(k-1)*7+p7H_DM];
}

static void profile_config(P7O_PROFILE *p, const float *t, const float *mat)
{
  int M = p->M, K = p->K, Kp = p->Kp;
  unsigned char dg[P7O_MAXKP][P7O_MAXK];
  degen_matrix(K, Kp, dg);
  float *occ = (float *) malloc(sizeof(float) * (M + 1));
  float Z = 0.f;
  for (int i = 0; i < (M+1)*8; i++) p->tsc[i] = -INFINITY;
  calculate_occupancy(M, t, occ);
  for (int k = 1; k <= M; k++) Z += occ[k] * (float)(M - k + 1);
  for (int k = 1; k <= M; k++) p->tsc[(k-1)*8 + p7P_BM] = log(occ[k] / Z);
  free(occ);
  p->xsc[p7O_E][p7O_MOVE] = -LOG2;
  p->xsc[p7O_E][p7O_LOOP] = -LOG2;
  p->nj = 1.0f;
  for (int k = 1; k < M; k++) {
    float *tp = p->tsc + k*8;
    tp[p7P_MM] = log(t[k*7+p7H_MM]);
    tp[p7P_MI] = log(t[k*7+p7H_MI]);
    tp[p7P_MD] = log(t[k*7+p7H_MD]);
    tp[p7P_IM] = log(t[k*7+p7H_IM]);
    tp[p7P_II] = log(t[k*7+p7H_II]);
    tp[p7P_DM] = log(t[k*7+p7H_DM]);
    tp[p7P_DD] = log(t[k*7+p7H_DD]);
  }
  /* match emission scores; degenerate residues by esl_abc_FExpectScVec */
  for (int x = 0; x < Kp; x++) p->msc[x*(M+1) + 0] = -INFINITY;
  for (int k = 1; k <= M; k++) {
    float sc[P7O_MAXKP];
    for (int x = 0; x < K; x++) sc[x] = log((double) mat[k*K+x] / p->bgf[x]);
    sc[K] = -INFINITY; sc[Kp-2] = -INFINITY; sc[Kp-1] = -INFINITY;
    for (int x = K+1; x <= Kp-3; x++) {
      float result = 0.f, denom = 0.f;
      for (int i = 0; i < K; i++) if (dg[x][i]) { result += sc[i] * p->bgf[i]; denom += p->bgf[i]; }
      sc[x] = result / denom;
    }
    for (int x = 0; x < Kp; x++) p->msc[x*(M+1) + k] = sc[x];
  }
}

static void reconfig_generic_length(P7O_PROFILE *p, int L)
{ /* upstream p7_ReconfigLength */
  float pmove = (2.0f + p->nj) / ((float) L + 2.0f + p->nj);
  float ploop = 1.0f - pmove;
  p->xsc[p7O_N][p7O_LOOP] = p->xsc[p7O_C][p7O_LOOP] = p->xsc[p7O_J][p7O_LOOP] = log(ploop);
  p->xsc[p7O_N][p7O_MOVE] = p->xsc[p7O_C][p7O_MOVE] = p->xsc[p7O_J][p7O_MOVE] = log(pmove);
  p->L = L;
}

/* ---------------------------------------------------------------- oprofile convert
 * upstream impl_sse/p7_oprofile.c: mf_conversion, sf_conversion, vf_conversion, fb_conversion
 * (reference include/libhmmer/impl_sse/p7_oprofile.pxd:121; plan7.pyx:4961).
 */
static uint8_t unbiased_byteify(const P7O_PROFILE *p, float sc)
{
  sc = -1.0f * roundf(p->scale_b * sc);
  return (sc > 255.) ? 255 : (uint8_t)(int) sc;
}
static uint8_t biased_byteify(const P7O_PROFILE *p, float sc)
{
  sc = -1.0f * roundf(p->scale_b * sc);
  return (sc > 255 - p->bias_b) ? 255 : (uint8_t)((int) sc + p->bias_b);
}
static int16_t wordify(const P7O_PROFILE *p, float sc)
{
  sc = roundf(p->scale_w * sc);
  if      (sc >=  32767.0) return  32767;
  else if (sc <= -32768.0) return -32768;
  else return (int16_t) sc;
}

static void mf_conversion(P7O_PROFILE *p)
{
  int M = p->M, nq = p->Q16;
  float max = 0.0f;
  /* max over rsc[x][(M+1)*2] for x<K: match scores (node 0 = -inf) and insert scores (0 or -inf) */
  for (int x = 0; x < p->K; x++)
    for (int k = 0; k <= M; k++) if (p->msc[x*(M+1)+k] > max) max = p->msc[x*(M+1)+k];
  p->scale_b = 3.0 / LOG2;
  p->base_b  = 190;
  p->bias_b  = unbiased_byteify(p, -1.0 * max);
  for (int x = 0; x < p->Kp; x++)
    for (int q = 0, k = 1; q < nq; q++, k++)
      for (int z = 0; z < 16; z++)
        p->rbv[(x*nq + q)*16 + z] = (k + z*nq <= M) ? biased_byteify(p, p->msc[x*(M+1) + k + z*nq]) : 255;
  p->tbm_b = unbiased_byteify(p, logf(2.0f / ((float) M * (float)(M+1))));
  p->tec_b = unbiased_byteify(p, logf(0.5f));
  p->tjb_b = unbiased_byteify(p, logf(3.0f / (float)(p->L + 3)));
  /* sf_conversion: sbv = ((bias+127) -sat_u8 rbv) ^ 127 ; extra vectors repeat q % nq */
  {
    int nqs = nq + P7O_EXTRA_SB;
    uint8_t t1 = (uint8_t)(p->bias_b + 127);
    for (int x = 0; x < p->Kp; x++) {
      for (int q = 0; q < nq; q++)
        for (int z = 0; z < 16; z++) {
          uint8_t r = p->rbv[(x*nq+q)*16+z];
          uint8_t d = (t1 > r) ? (uint8_t)(t1 - r) : 0;
          p->sbv[(x*nqs+q)*16+z] = (int8_t)(d ^ 127);
        }
      for (int q = nq; q < nqs; q++)
        memcpy(p->sbv + (x*nqs+q)*16, p->sbv + (x*nqs + (q % nq))*16, 16);
    }
  }
}

static void vf_conversion(P7O_PROFILE *p)
{
  int M = p->M, nq = p->Q8, j = 0;
  p->scale_w = 500.0 / LOG2;
  p->base_w  = 12000;
  for (int x = 0; x < p->Kp; x++)
    for (int k = 1, q = 0; q < nq; q++, k++)
      for (int z = 0; z < 8; z++)
        p->rwv[(x*nq+q)*8+z] = (k + z*nq <= M) ? wordify(p, p->msc[x*(M+1) + k + z*nq]) : -32768;
  for (int k = 1, q = 0; q < nq; q++, k++)
    for (int t = p7O_BM; t <= p7O_II; t++) {
      int tg = 0, kb = 0; int16_t maxval = 0;
      switch (t) {
        case p7O_BM: tg = p7P_BM; kb = k-1; maxval = 0;  break;
        case p7O_MM: tg = p7P_MM; kb = k-1; maxval = 0;  break;
        case p7O_IM: tg = p7P_IM; kb = k-1; maxval = 0;  break;
        case p7O_DM: tg = p7P_DM; kb = k-1; maxval = 0;  break;
        case p7O_MD: tg = p7P_MD; kb = k;   maxval = 0;  break;
        case p7O_MI: tg = p7P_MI; kb = k;   maxval = 0;  break;
        case p7O_II: tg = p7P_II; kb = k;   maxval = -1; break;
      }
      for (int z = 0; z < 8; z++) {
        int16_t val = (kb + z*nq < M) ? wordify(p, p->tsc[(kb + z*nq)*8 + tg]) : -32768;
        p->twv[j*8+z] = (val <= maxval) ? val : maxval;
      }
      j++;
    }
  for (int k = 1, q = 0; q < nq; q++, k++) {
    for (int z = 0; z < 8; z++)
      p->twv[j*8+z] = (k + z*nq < M) ? wordify(p, p->tsc[(k + z*nq)*8 + p7P_DD]) : -32768;
    j++;
  }
  p->xw[p7O_E][p7O_LOOP] = wordify(p, p->xsc[p7O_E][p7O_LOOP]);
  p->xw[p7O_E][p7O_MOVE] = wordify(p, p->xsc[p7O_E][p7O_MOVE]);
  p->xw[p7O_N][p7O_MOVE] = wordify(p, p->xsc[p7O_N][p7O_MOVE]);
  p->xw[p7O_N][p7O_LOOP] = 0;
  p->xw[p7O_C][p7O_MOVE] = wordify(p, p->xsc[p7O_C][p7O_MOVE]);
  p->xw[p7O_C][p7O_LOOP] = 0;
  p->xw[p7O_J][p7O_MOVE] = wordify(p, p->xsc[p7O_J][p7O_MOVE]);
  p->xw[p7O_J][p7O_LOOP] = 0;
  p->ncj_roundoff = 0.0f;
  p->ddbound_w = -32768;
  for (int k = 2; k < M-1; k++) {
    int ddtmp = (int) wordify(p, p->tsc[k*8 + p7P_DD]);
    ddtmp += (int) wordify(p, p->tsc[(k+1)*8 + p7P_DM]);
    ddtmp -= (int) wordify(p, p->tsc[(k+1)*8 + p7P_BM]);
    if (ddtmp > p->ddbound_w) p->ddbound_w = (int16_t) ddtmp;
  }
}

static void fb_conversion(P7O_PROFILE *p)
{
  int M = p->M, nq = p->Q4, j = 0;
  for (int x = 0; x < p->Kp; x++)
    for (int k = 1, q = 0; q < nq; q++, k++)
      for (int z = 0; z < 4; z++)
        p->rfv[(x*nq+q)*4+z] = p7o_sse_expf_scalar((k + z*nq <= M) ? p->msc[x*(M+1) + k + z*nq] : -INFINITY);
  for (int k = 1, q = 0; q < nq; q++, k++)
    for (int t = p7O_BM; t <= p7O_II; t++) {
      int tg = 0, kb = 0;
      switch (t) {
        case p7O_BM: tg = p7P_BM; kb = k-1; break;
        case p7O_MM: tg = p7P_MM; kb = k-1; break;
        case p7O_IM: tg = p7P_IM; kb = k-1; break;
        case p7O_DM: tg = p7P_DM; kb = k-1; break;
        case p7O_MD: tg = p7P_MD; kb = k;   break;
        case p7O_MI: tg = p7P_MI; kb = k;   break;
        case p7O_II: tg = p7P_II; kb = k;   break;
      }
      for (int z = 0; z < 4; z++)
        p->tfv[j*4+z] = p7o_sse_expf_scalar((kb + z*nq < M) ? p->tsc[(kb + z*nq)*8 + tg] : -INFINITY);
      j++;
    }
  for (int k = 1, q = 0; q < nq; q++, k++) {
    for (int z = 0; z < 4; z++)
      p->tfv[j*4+z] = p7o_sse_expf_scalar((k + z*nq < M) ? p->tsc[(k + z*nq)*8 + p7P_DD] : -INFINITY);
    j++;
  }
  for (int s = 0; s < 4; s++) for (int m = 0; m < 2; m++) p->xf[s][m] = expf(p->xsc[s][m]);
}

void p7o_reconfig_length(P7O_PROFILE *p, int L)
{ /* upstream p7_oprofile_ReconfigLength = ReconfigMSVLength + ReconfigRestLength
   * (reference impl_sse/p7_oprofile.pxd:122-126; plan7.pyx:6438) */
  float pmove, ploop;
  p->tjb_b = unbiased_byteify(p, logf(3.0f / (float)(L + 3)));
  pmove = (2.0f + p->nj) / ((float) L + 2.0f + p->nj);
  ploop = 1.0f - pmove;
  p->xf[p7O_N][p7O_LOOP] = p->xf[p7O_C][p7O_LOOP] = p->xf[p7O_J][p7O_LOOP] = ploop;
  p->xf[p7O_N][p7O_MOVE] = p->xf[p7O_C][p7O_MOVE] = p->xf[p7O_J][p7O_MOVE] = pmove;
  p->xw[p7O_N][p7O_MOVE] = p->xw[p7O_C][p7O_MOVE] = p->xw[p7O_J][p7O_MOVE] = wordify(p, logf(pmove));
  p->L = L;
}

P7O_PROFILE *p7o_profile_build(int M, int K, const float *t, const float *mat, const float *bgf,
                               const float *compo, const float *evparam, int L)
{
  P7O_PROFILE *p = (P7O_PROFILE *) calloc(1, sizeof(P7O_PROFILE));
  int Kp = (K == 20) ? 29 : 18;
  p->M = M; p->K = K; p->Kp = Kp;
  p->Q16 = (M-1)/16 + 1; if (p->Q16 < 2) p->Q16 = 2;
  p->Q8  = (M-1)/8  + 1; if (p->Q8  < 2) p->Q8  = 2;
  p->Q4  = (M-1)/4  + 1; if (p->Q4  < 2) p->Q4  = 2;
  p->tsc = (float *)   amalloc(sizeof(float) * (M+1) * 8);
  p->msc = (float *)   amalloc(sizeof(float) * Kp * (M+1));
  p->rbv = (uint8_t *) amalloc((size_t) Kp * p->Q16 * 16);
  p->sbv = (int8_t *)  amalloc((size_t) Kp * (p->Q16 + P7O_EXTRA_SB) * 16);
  p->rwv = (int16_t *) amalloc(sizeof(int16_t) * Kp * p->Q8 * 8);
  p->twv = (int16_t *) amalloc(sizeof(int16_t) * 8 * p->Q8 * 8);
  p->rfv = (float *)   amalloc(sizeof(float) * Kp * p->Q4 * 4);
  p->tfv = (float *)   amalloc(sizeof(float) * 8 * p->Q4 * 4);
  memcpy(p->bgf, bgf, sizeof(float) * K);
  if (compo) memcpy(p->compo, compo, sizeof(float) * K);
  if (evparam) memcpy(p->evparam, evparam, sizeof(float) * 6);
  profile_config(p, t, mat);
  reconfig_generic_length(p, L);
  mf_conversion(p);
  vf_conversion(p);
  fb_conversion(p);
  return p;
}

void p7o_profile_free(P7O_PROFILE *p)
{
  if (!p) return;
  free(p->tsc); free(p->msc); free(p->rbv); free(p->sbv); free(p->rwv); free(p->twv); free(p->rfv); free(p->tfv);
  free(p);
}

/* ---------------------------------------------------------------- MSV filter
 * upstream impl_sse/msvfilter.c p7_MSVFilter (reference impl_sse/__init__.pxd:12; plan7.pyx:5005).
 * p7_MSVFilter first tries p7_SSVFilter, which by construction returns the same score whenever it
 * returns eslOK (plan7.pyx:5033-5038), so the MSV recurrence alone defines the result.
 */
static inline uint8_t hmax_epu8(__m128i a)
{
  a = _mm_max_epu8(a, _mm_srli_si128(a, 8));
  a = _mm_max_epu8(a, _mm_srli_si128(a, 4));
  a = _mm_max_epu8(a, _mm_srli_si128(a, 2));
  a = _mm_max_epu8(a, _mm_srli_si128(a, 1));
  return (uint8_t) _mm_cvtsi128_si32(a);
}
static inline int16_t hmax_epi16(__m128i a)
{
  a = _mm_max_epi16(a, _mm_srli_si128(a, 8));
  a = _mm_max_epi16(a, _mm_srli_si128(a, 4));
  a = _mm_max_epi16(a, _mm_srli_si128(a, 2));
  return (int16_t) _mm_cvtsi128_si32(a);
}

int p7o_msv(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xJ)
{
  int Q = p->Q16;
  __m128i dpbuf[Q];
  __m128i *dp = dpbuf;
  __m128i biasv = _mm_set1_epi8((int8_t) p->bias_b);
  __m128i ceilingv = _mm_cmpeq_epi8(biasv, biasv);
  __m128i basev = _mm_set1_epi8((int8_t) p->base_b);
  __m128i tjbmv = _mm_set1_epi8((int8_t)(p->tjb_b + p->tbm_b));
  __m128i tecv  = _mm_set1_epi8((int8_t) p->tec_b);
  __m128i xJv   = _mm_setzero_si128();
  __m128i xBv   = _mm_subs_epu8(basev, tjbmv);
  uint8_t xJ;
  for (int q = 0; q < Q; q++) dp[q] = _mm_setzero_si128();
  for (int i = 1; i <= L; i++) {
    const __m128i *rsc = (const __m128i *)(p->rbv + (size_t) dsq[i] * Q * 16);
    __m128i xEv = _mm_setzero_si128();
    __m128i mpv = _mm_slli_si128(dp[Q-1], 1);
    for (int q = 0; q < Q; q++) {
      __m128i sv = _mm_max_epu8(mpv, xBv);
      sv  = _mm_adds_epu8(sv, biasv);
      sv  = _mm_subs_epu8(sv, rsc[q]);
      xEv = _mm_max_epu8(xEv, sv);
      mpv = dp[q];
      dp[q] = sv;
    }
    {
      __m128i tempv = _mm_adds_epu8(xEv, biasv);
      tempv = _mm_cmpeq_epi8(tempv, ceilingv);
      if (_mm_movemask_epi8(tempv) != 0) { *ret_sc = INFINITY; if (ret_xJ) *ret_xJ = -1; return 16; /* eslERANGE */ }
    }
    xEv = _mm_set1_epi8((int8_t) hmax_epu8(xEv));
    xEv = _mm_subs_epu8(xEv, tecv);
    xJv = _mm_max_epu8(xJv, xEv);
    xBv = _mm_max_epu8(basev, xJv);
    xBv = _mm_subs_epu8(xBv, tjbmv);
  }
  xJ = (uint8_t) _mm_extract_epi16(xJv, 0);
  if (ret_xJ) *ret_xJ = xJ;
  *ret_sc = ((float)(xJ - p->tjb_b) - (float) p->base_b);
  *ret_sc /= p->scale_b;
  *ret_sc -= 3.0;
  return 0;
}

/* Un-striped scalar twin used to cross-check the SSE restatement. */
static inline int rb_unstriped(const P7O_PROFILE *p, int x, int k)
{ int q = (k-1) % p->Q16, z = (k-1) / p->Q16; return p->rbv[((size_t) x * p->Q16 + q)*16 + z]; }

int p7o_msv_scalar(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xJ)
{
  int M = p->M;
  int *mrow = (int *) calloc(M+1, sizeof(int)), *nrow = (int *) calloc(M+1, sizeof(int));
  int bias = p->bias_b, base = p->base_b, tjbm = p->tjb_b + p->tbm_b, tec = p->tec_b;
  int xJ = 0, xB = base - tjbm; if (xB < 0) xB = 0;
  for (int i = 1; i <= L; i++) {
    int xE = 0, x = dsq[i];
    for (int k = 1; k <= M; k++) {
      int sv = mrow[k-1] > xB ? mrow[k-1] : xB;
      sv += bias; if (sv > 255) sv = 255;
      sv -= rb_unstriped(p, x, k); if (sv < 0) sv = 0;
      if (sv > xE) xE = sv;
      nrow[k] = sv;
    }
    if (xE + bias >= 255) { free(mrow); free(nrow); *ret_sc = INFINITY; if (ret_xJ) *ret_xJ = -1; return 16; }
    xE -= tec; if (xE < 0) xE = 0;
    if (xE > xJ) xJ = xE;
    xB = (base > xJ ? base : xJ) - tjbm; if (xB < 0) xB = 0;
    { int *t = mrow; mrow = nrow; nrow = t; }
  }
  free(mrow); free(nrow);
  if (ret_xJ) *ret_xJ = xJ;
  *ret_sc = ((float)(xJ - p->tjb_b) - (float) p->base_b);
  *ret_sc /= p->scale_b;
  *ret_sc -= 3.0;
  return 0;
}

/* ---------------------------------------------------------------- Viterbi filter
 * upstream impl_sse/vitfilter.c p7_ViterbiFilter (only reachable through p7_Pipeline,
 * reference include/libhmmer/p7_pipeline.pxd:130).  Includes the "lazy F" DD evaluation.
 */
int p7o_vit(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xC)
{
  int Q = p->Q8;
  __m128i dpbuf[3*Q];
  __m128i *dp = dpbuf;
#define MMXo(q) (dp[(q)*3 + 0])
#define DMXo(q) (dp[(q)*3 + 1])
#define IMXo(q) (dp[(q)*3 + 2])
  __m128i negInfv = _mm_srli_si128(_mm_set1_epi16(-32768), 14);
  const __m128i *twv = (const __m128i *) p->twv;
  int16_t xE, xB, xC, xJ, xN, Dmax;
  for (int q = 0; q < Q; q++) MMXo(q) = IMXo(q) = DMXo(q) = _mm_set1_epi16(-32768);
  xN = p->base_w;
  xB = xN + p->xw[p7O_N][p7O_MOVE];
  xJ = -32768; xC = -32768; xE = -32768;
  for (int i = 1; i <= L; i++) {
    const __m128i *rsc = (const __m128i *)(p->rwv + (size_t) dsq[i] * Q * 8);
    const __m128i *tsc = twv;
    __m128i dcv = _mm_set1_epi16(-32768), xEv = dcv, Dmaxv = dcv, xBv = _mm_set1_epi16(xB);
    __m128i mpv, dpv, ipv, sv;
    int q;
    mpv = _mm_or_si128(_mm_slli_si128(MMXo(Q-1), 2), negInfv);
    dpv = _mm_or_si128(_mm_slli_si128(DMXo(Q-1), 2), negInfv);
    ipv = _mm_or_si128(_mm_slli_si128(IMXo(Q-1), 2), negInfv);
    for (q = 0; q < Q; q++) {
      sv  =                   _mm_adds_epi16(xBv, *tsc);  tsc++;
      sv  = _mm_max_epi16(sv, _mm_adds_epi16(mpv, *tsc)); tsc++;
      sv  = _mm_max_epi16(sv, _mm_adds_epi16(ipv, *tsc)); tsc++;
      sv  = _mm_max_epi16(sv, _mm_adds_epi16(dpv, *tsc)); tsc++;
      sv  = _mm_adds_epi16(sv, rsc[q]);
      xEv = _mm_max_epi16(xEv, sv);
      mpv = MMXo(q); dpv = DMXo(q); ipv = IMXo(q);
      MMXo(q) = sv;
      DMXo(q) = dcv;
      dcv   = _mm_adds_epi16(sv, *tsc); tsc++;
      Dmaxv = _mm_max_epi16(dcv, Dmaxv);
      sv      =                   _mm_adds_epi16(mpv, *tsc);  tsc++;
      IMXo(q) = _mm_max_epi16(sv, _mm_adds_epi16(ipv, *tsc)); tsc++;
    }
    xE = hmax_epi16(xEv);
    if (xE >= 32767) { *ret_sc = INFINITY; if (ret_xC) *ret_xC = 32767; return 16; }
    xN = xN + p->xw[p7O_N][p7O_LOOP];
    { int a = xC + p->xw[p7O_C][p7O_LOOP], b = xE + p->xw[p7O_E][p7O_MOVE]; xC = (int16_t)(a > b ? a : b); }
    { int a = xJ + p->xw[p7O_J][p7O_LOOP], b = xE + p->xw[p7O_E][p7O_LOOP]; xJ = (int16_t)(a > b ? a : b); }
    { int a = xJ + p->xw[p7O_J][p7O_MOVE], b = xN + p->xw[p7O_N][p7O_MOVE]; xB = (int16_t)(a > b ? a : b); }
    Dmax = hmax_epi16(Dmaxv);
    if ((int) Dmax + (int) p->ddbound_w > (int) xB) {
      dcv = _mm_or_si128(_mm_slli_si128(dcv, 2), negInfv);
      tsc = twv + 7*Q;
      for (q = 0; q < Q; q++) {
        DMXo(q) = _mm_max_epi16(dcv, DMXo(q));
        dcv     = _mm_adds_epi16(DMXo(q), *tsc); tsc++;
      }
      do {
        dcv = _mm_or_si128(_mm_slli_si128(dcv, 2), negInfv);
        tsc = twv + 7*Q;
        for (q = 0; q < Q; q++) {
          if (_mm_movemask_epi8(_mm_cmpgt_epi16(dcv, DMXo(q))) == 0) break;
          DMXo(q) = _mm_max_epi16(dcv, DMXo(q));
          dcv     = _mm_adds_epi16(DMXo(q), *tsc); tsc++;
        }
      } while (q == Q);
    } else {
      DMXo(0) = _mm_or_si128(_mm_slli_si128(dcv, 2), negInfv);
    }
  }
#undef MMXo
#undef DMXo
#undef IMXo
  if (ret_xC) *ret_xC = xC;
  if (xC > -32768) {
    *ret_sc = (float) xC + (float) p->xw[p7O_C][p7O_MOVE] - (float) p->base_w;
    *ret_sc /= p->scale_w;
    *ret_sc -= 3.0;
  } else *ret_sc = -INFINITY;
  return 0;
}

/* Un-striped scalar twin with the D->D chain fully evaluated. */
static inline int sat16(int a) { return a > 32767 ? 32767 : (a < -32768 ? -32768 : a); }
int p7o_vit_scalar(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xC)
{
  int M = p->M, Q = p->Q8;
  /* un-stripe: T[t][k] = value held for node k (k=1..Q*8) */
  int N = Q*8;
  int *T = (int *) malloc(sizeof(int) * 8 * (N+1));
  int *blk = (int *) malloc(sizeof(int)*(N+1)*6);
  int *Mr = blk;
  int *Ir = Mr + (N+1), *Dr = Ir + (N+1), *Mn = Dr + (N+1), *In = Mn + (N+1), *Dn = In + (N+1);
  for (int q = 0; q < Q; q++) for (int z = 0; z < 8; z++) {
    int k = q + 1 + z*Q;
    for (int t = 0; t < 7; t++) T[t*(N+1)+k] = p->twv[(q*7 + t)*8 + z];
    T[7*(N+1)+k] = p->twv[(7*Q + q)*8 + z];
  }
  for (int k = 0; k <= N; k++) Mr[k] = Ir[k] = Dr[k] = -32768;
  int xN = p->base_w, xB = xN + p->xw[p7O_N][p7O_MOVE], xJ = -32768, xC = -32768, xE;
  for (int i = 1; i <= L; i++) {
    int x = dsq[i];
    xE = -32768;
    Mn[0] = In[0] = Dn[0] = -32768;
    for (int k = 1; k <= N; k++) {
      int q = (k-1) % Q, z = (k-1) / Q;
      int sv = sat16(xB + T[p7O_BM*(N+1)+k]);
      int a;
      a = sat16(Mr[k-1] + T[p7O_MM*(N+1)+k]); if (a > sv) sv = a;
      a = sat16(Ir[k-1] + T[p7O_IM*(N+1)+k]); if (a > sv) sv = a;
      a = sat16(Dr[k-1] + T[p7O_DM*(N+1)+k]); if (a > sv) sv = a;
      sv = sat16(sv + p->rwv[((size_t) x*Q + q)*8 + z]);
      Mn[k] = sv;
      if (sv > xE) xE = sv;                 /* padded lanes (k > M) hold -32768 */
      sv = sat16(Mr[k] + T[p7O_MI*(N+1)+k]);
      a  = sat16(Ir[k] + T[p7O_II*(N+1)+k]);
      In[k] = a > sv ? a : sv;
    }
    Dn[1] = -32768;
    for (int k = 2; k <= N; k++) {
      int a = sat16(Mn[k-1] + T[p7O_MD*(N+1)+k-1]);
      int b = sat16(Dn[k-1] + T[p7O_DD*(N+1)+k-1]);
      Dn[k] = a > b ? a : b;
    }
    if (xE >= 32767) { free(T); free(blk); *ret_sc = INFINITY; if (ret_xC) *ret_xC = 32767; return 16; }
    { int a = xC, b = xE + p->xw[p7O_E][p7O_MOVE]; xC = (int16_t)(a > b ? a : b); }
    { int a = xJ, b = xE + p->xw[p7O_E][p7O_LOOP]; xJ = (int16_t)(a > b ? a : b); }
    { int a = xJ + p->xw[p7O_J][p7O_MOVE], b = xN + p->xw[p7O_N][p7O_MOVE]; xB = (int16_t)(a > b ? a : b); }
    { int *t; t = Mr; Mr = Mn; Mn = t; t = Ir; Ir = In; In = t; t = Dr; Dr = Dn; Dn = t; }
  }
  free(T); free(blk);
  if (ret_xC) *ret_xC = xC;
  if (xC > -32768) {
    *ret_sc = (float) xC + (float) p->xw[p7O_C][p7O_MOVE] - (float) p->base_w;
    *ret_sc /= p->scale_w;
    *ret_sc -= 3.0;
  } else *ret_sc = -INFINITY;
  return 0;
}

/* ---------------------------------------------------------------- Forward / Backward parsers
 * upstream impl_sse/fwdback.c forward_engine / backward_engine with do_full = FALSE
 * (p7_ForwardParser / p7_BackwardParser; only reachable through p7_Pipeline).
 * xmx layout per row: [E,N,J,B,C,SCALE] (reference include/libhmmer/impl_sse/p7_omx.pxd:16-38).
 */
enum { X_E = 0, X_N, X_J, X_B, X_C, X_SCALE, X_NCELLS };

static inline __m128 rightshift_ps(__m128 a, __m128 b)
{ return _mm_move_ss(_mm_shuffle_ps(a, a, _MM_SHUFFLE(2, 1, 0, 0)), b); }
static inline __m128 leftshift_ps(__m128 a, __m128 zerov)
{ a = _mm_move_ss(a, zerov); return _mm_shuffle_ps(a, a, _MM_SHUFFLE(0, 3, 2, 1)); }

int p7o_fwd(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *xmx, float *ret_sc)
{
  int Q = p->Q4;
  __m128 dpbuf[3*Q];
  __m128 *dpc = dpbuf;
#define MMO(q) (dpc[(q)*3 + 0])
#define DMO(q) (dpc[(q)*3 + 1])
#define IMO(q) (dpc[(q)*3 + 2])
  const __m128 *tfv = (const __m128 *) p->tfv;
  __m128 zerov = _mm_setzero_ps();
  float xN, xE, xB, xC, xJ, totscale = 0.0f;
  for (int q = 0; q < Q; q++) MMO(q) = IMO(q) = DMO(q) = zerov;
  xE = 0.f; xN = 1.f; xJ = 0.f; xB = p->xf[p7O_N][p7O_MOVE]; xC = 0.f;
  if (xmx) { xmx[X_E] = xE; xmx[X_N] = xN; xmx[X_J] = xJ; xmx[X_B] = xB; xmx[X_C] = xC; xmx[X_SCALE] = 1.0f; }
  for (int i = 1; i <= L; i++) {
    const __m128 *rp = (const __m128 *)(p->rfv + (size_t) dsq[i] * Q * 4);
    const __m128 *tp = tfv;
    __m128 dcv = zerov, xEv = zerov, xBv = _mm_set1_ps(xB);
    __m128 mpv = rightshift_ps(MMO(Q-1), zerov);
    __m128 dpv = rightshift_ps(DMO(Q-1), zerov);
    __m128 ipv = rightshift_ps(IMO(Q-1), zerov);
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
    dcv    = rightshift_ps(dcv, zerov);
    DMO(0) = zerov;
    tp     = tfv + 7*Q;
    for (q = 0; q < Q; q++) {
      DMO(q) = _mm_add_ps(dcv, DMO(q));
      dcv    = _mm_mul_ps(DMO(q), *tp); tp++;
    }
    if (p->M < 100) {
      for (j = 1; j < 4; j++) {
        dcv = rightshift_ps(dcv, zerov);
        tp  = tfv + 7*Q;
        for (q = 0; q < Q; q++) {
          DMO(q) = _mm_add_ps(dcv, DMO(q));
          dcv    = _mm_mul_ps(dcv, *tp); tp++;
        }
      }
    } else {
      for (j = 1; j < 4; j++) {
        __m128 cv = zerov;
        dcv = rightshift_ps(dcv, zerov);
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
    xEv = _mm_add_ps(xEv, _mm_shuffle_ps(xEv, xEv, _MM_SHUFFLE(0, 3, 2, 1)));
    xEv = _mm_add_ps(xEv, _mm_shuffle_ps(xEv, xEv, _MM_SHUFFLE(1, 0, 3, 2)));
    _mm_store_ss(&xE, xEv);
    xN =  xN * p->xf[p7O_N][p7O_LOOP];
    xC = (xC * p->xf[p7O_C][p7O_LOOP]) + (xE * p->xf[p7O_E][p7O_MOVE]);
    xJ = (xJ * p->xf[p7O_J][p7O_LOOP]) + (xE * p->xf[p7O_E][p7O_LOOP]);
    xB = (xJ * p->xf[p7O_J][p7O_MOVE]) + (xN * p->xf[p7O_N][p7O_MOVE]);
    if (xE > 1.0e4) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      xEv = _mm_set1_ps(1.0 / xE);
      for (q = 0; q < Q; q++) {
        MMO(q) = _mm_mul_ps(MMO(q), xEv);
        DMO(q) = _mm_mul_ps(DMO(q), xEv);
        IMO(q) = _mm_mul_ps(IMO(q), xEv);
      }
      if (xmx) xmx[i*X_NCELLS + X_SCALE] = xE;
      totscale += log(xE);
      xE = 1.0;
    } else if (xmx) xmx[i*X_NCELLS + X_SCALE] = 1.0;
    if (xmx) {
      xmx[i*X_NCELLS+X_E] = xE; xmx[i*X_NCELLS+X_N] = xN; xmx[i*X_NCELLS+X_J] = xJ;
      xmx[i*X_NCELLS+X_B] = xB; xmx[i*X_NCELLS+X_C] = xC;
    }
  }
#undef MMO
#undef DMO
#undef IMO
  if (isnan(xC) || (L > 0 && xC == 0.0) || isinf(xC)) { *ret_sc = isnan(xC) ? NAN : INFINITY; return 16; }
  *ret_sc = totscale + log(xC * p->xf[p7O_C][p7O_MOVE]);
  return 0;
}

int p7o_bck(const P7O_PROFILE *p, const uint8_t *dsq, int L, const float *fwd_xmx, float *bxmx, float *ret_sc)
{
  int Q = p->Q4;
  __m128 buf[6*Q];
  __m128 *dpc = buf, *dpp = buf + 3*Q;
#define MMOx(d,q) ((d)[(q)*3 + 0])
#define DMOx(d,q) ((d)[(q)*3 + 1])
#define IMOx(d,q) ((d)[(q)*3 + 2])
  const __m128 *tfv = (const __m128 *) p->tfv;
  const __m128 *tp, *rp;
  __m128 zerov = _mm_setzero_ps();
  __m128 mpv, ipv, dpv, mcv, dcv, tmmv, timv, tdmv, xBv, xEv;
  float xN, xE, xB, xC, xJ, totscale, sc;
  int has_own_scales = 0;
  int q, j;
  xJ = 0.f; xB = 0.f; xN = 0.f;
  xC = p->xf[p7O_C][p7O_MOVE];
  xE = xC * p->xf[p7O_E][p7O_MOVE];
  xEv = _mm_set1_ps(xE);
  dcv = zerov;
  for (q = 0; q < Q; q++) MMOx(dpc,q) = DMOx(dpc,q) = xEv;
  for (q = 0; q < Q; q++) IMOx(dpc,q) = zerov;
  tp  = tfv + 8*Q - 1;
  dpv = leftshift_ps(DMOx(dpc,Q-1), zerov);
  for (q = Q-1; q >= 0; q--) {
    dcv = _mm_mul_ps(dpv, *tp); tp--;
    DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), dcv);
    dpv = DMOx(dpc,q);
  }
  for (j = 1; j < 4; j++) {
    tp  = tfv + 8*Q - 1;
    dcv = leftshift_ps(dcv, zerov);
    for (q = Q-1; q >= 0; q--) {
      dcv = _mm_mul_ps(dcv, *tp); tp--;
      DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), dcv);
    }
  }
  tp  = tfv + 7*Q - 3;
  dcv = leftshift_ps(DMOx(dpc,0), zerov);
  for (q = Q-1; q >= 0; q--) {
    MMOx(dpc,q) = _mm_add_ps(MMOx(dpc,q), _mm_mul_ps(dcv, *tp)); tp -= 7;
    dcv = DMOx(dpc,q);
  }
  sc = fwd_xmx[L*X_NCELLS + X_SCALE];
  if (sc > 1.0f) {
    xE = xE / sc; xN = xN / sc; xC = xC / sc; xJ = xJ / sc; xB = xB / sc;
    xEv = _mm_set1_ps(1.0 / sc);
    for (q = 0; q < Q; q++) {
      MMOx(dpc,q) = _mm_mul_ps(MMOx(dpc,q), xEv);
      DMOx(dpc,q) = _mm_mul_ps(DMOx(dpc,q), xEv);
      IMOx(dpc,q) = _mm_mul_ps(IMOx(dpc,q), xEv);
    }
  }
  bxmx[L*X_NCELLS + X_SCALE] = sc;
  totscale = log(sc);
  bxmx[L*X_NCELLS+X_E] = xE; bxmx[L*X_NCELLS+X_N] = xN; bxmx[L*X_NCELLS+X_J] = xJ;
  bxmx[L*X_NCELLS+X_B] = xB; bxmx[L*X_NCELLS+X_C] = xC;

  for (int i = L-1; i >= 1; i--) {
    { __m128 *t = dpc; dpc = dpp; dpp = t; }   /* dpp = row i+1, dpc = row i (being built) */
    rp  = (const __m128 *)(p->rfv + (size_t) dsq[i+1] * Q * 4) + Q-1;
    tp  = tfv + 7*Q - 1;
    tmmv = leftshift_ps(tfv[1], zerov);
    timv = leftshift_ps(tfv[2], zerov);
    tdmv = leftshift_ps(tfv[3], zerov);
    mpv = _mm_mul_ps(MMOx(dpp,0), *((const __m128 *)(p->rfv + (size_t) dsq[i+1] * Q * 4)));
    mpv = leftshift_ps(mpv, zerov);
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
    xBv = _mm_add_ps(xBv, _mm_shuffle_ps(xBv, xBv, _MM_SHUFFLE(0, 3, 2, 1)));
    xBv = _mm_add_ps(xBv, _mm_shuffle_ps(xBv, xBv, _MM_SHUFFLE(1, 0, 3, 2)));
    _mm_store_ss(&xB, xBv);
    xC =  xC * p->xf[p7O_C][p7O_LOOP];
    xJ = (xB * p->xf[p7O_J][p7O_MOVE]) + (xJ * p->xf[p7O_J][p7O_LOOP]);
    xN = (xB * p->xf[p7O_N][p7O_MOVE]) + (xN * p->xf[p7O_N][p7O_LOOP]);
    xE = (xC * p->xf[p7O_E][p7O_MOVE]) + (xJ * p->xf[p7O_E][p7O_LOOP]);
    xEv = _mm_set1_ps(xE);
    tp  = tfv + 8*Q - 1;
    dpv = _mm_add_ps(DMOx(dpc,0), xEv);
    dpv = leftshift_ps(dpv, zerov);
    for (q = Q-1; q >= 0; q--) {
      dcv = _mm_mul_ps(dpv, *tp); tp--;
      DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), _mm_add_ps(dcv, xEv));
      dpv = DMOx(dpc,q);
      MMOx(dpc,q) = _mm_add_ps(MMOx(dpc,q), xEv);
    }
    for (j = 1; j < 4; j++) {
      dcv = leftshift_ps(dcv, zerov);
      tp  = tfv + 8*Q - 1;
      for (q = Q-1; q >= 0; q--) {
        dcv = _mm_mul_ps(dcv, *tp); tp--;
        DMOx(dpc,q) = _mm_add_ps(DMOx(dpc,q), dcv);
      }
    }
    dcv = leftshift_ps(DMOx(dpc,0), zerov);
    tp  = tfv + 7*Q - 3;
    for (q = Q-1; q >= 0; q--) {
      MMOx(dpc,q) = _mm_add_ps(MMOx(dpc,q), _mm_mul_ps(dcv, *tp)); tp -= 7;
      dcv = DMOx(dpc,q);
    }
    if (xB > 1.0e16) has_own_scales = 1;
    if (has_own_scales) bxmx[i*X_NCELLS+X_SCALE] = (xB > 1.0e4) ? xB : 1.0;
    else                bxmx[i*X_NCELLS+X_SCALE] = fwd_xmx[i*X_NCELLS+X_SCALE];
    sc = bxmx[i*X_NCELLS+X_SCALE];
    if (sc > 1.0f) {
      xE /= sc; xN /= sc; xJ /= sc; xB /= sc; xC /= sc;
      xBv = _mm_set1_ps(1.0 / sc);
      for (q = 0; q < Q; q++) {
        MMOx(dpc,q) = _mm_mul_ps(MMOx(dpc,q), xBv);
        DMOx(dpc,q) = _mm_mul_ps(DMOx(dpc,q), xBv);
        IMOx(dpc,q) = _mm_mul_ps(IMOx(dpc,q), xBv);
      }
      totscale += log(sc);
    }
    bxmx[i*X_NCELLS+X_E] = xE; bxmx[i*X_NCELLS+X_N] = xN; bxmx[i*X_NCELLS+X_J] = xJ;
    bxmx[i*X_NCELLS+X_B] = xB; bxmx[i*X_NCELLS+X_C] = xC;
  }
  /* row 0: only B and N are reachable */
  {
    __m128 *row1 = dpc;   /* after the loop dpc holds row 1 (or row L if L==1) */
    rp  = (const __m128 *)(p->rfv + (size_t) dsq[1] * Q * 4) + Q-1;
    tp  = tfv + 7*(Q-1);
    xBv = zerov;
    for (q = Q-1; q >= 0; q--) {
      mpv = _mm_mul_ps(MMOx(row1,q), *rp); rp--;
      mpv = _mm_mul_ps(mpv, *tp);          tp -= 7;
      xBv = _mm_add_ps(xBv, mpv);
    }
    xBv = _mm_add_ps(xBv, _mm_shuffle_ps(xBv, xBv, _MM_SHUFFLE(0, 3, 2, 1)));
    xBv = _mm_add_ps(xBv, _mm_shuffle_ps(xBv, xBv, _MM_SHUFFLE(1, 0, 3, 2)));
    _mm_store_ss(&xB, xBv);
    xN = (xB * p->xf[p7O_N][p7O_MOVE]) + (xN * p->xf[p7O_N][p7O_LOOP]);
    bxmx[X_B] = xB; bxmx[X_C] = 0.0f; bxmx[X_J] = 0.0f; bxmx[X_N] = xN; bxmx[X_E] = 0.0f; bxmx[X_SCALE] = 1.0f;
  }
#undef MMOx
#undef DMOx
#undef IMOx
  if (isnan(xN) || (L > 0 && xN == 0.0) || isinf(xN)) { if (ret_sc) *ret_sc = INFINITY; return 16; }
  if (ret_sc) *ret_sc = totscale + log(xN);
  return 0;
}

/* ---------------------------------------------------------------- null models
 * upstream p7_bg.c p7_bg_SetLength + p7_bg_NullOne (reference p7_bg.pxd:10-30; plan7.pyx:6435, 593);
 * p7_bg_SetFilter + p7_bg_FilterScore -> Easel esl_hmm_Configure + esl_hmm_Forward (libeasel/hmm.pxd:43).
 */
float p7o_null1(int L)
{
  float p1 = (float) L / (float)(L + 1);
  return (float) L * log(p1) + log(1. - p1);
}

float p7o_bias_filter(const P7O_PROFILE *p, const uint8_t *dsq, int L)
{
  int K = p->K, Kp = p->Kp;
  unsigned char dg[P7O_MAXKP][P7O_MAXK];
  float eo[P7O_MAXKP][2];
  float t[2][3], pi[2];
  float p1 = (float) L / (float)(L + 1);
  float L1 = (float) p->M / 8.0;
  float dp0, dp1, n0, n1, max, logsc, fsc, last;
  degen_matrix(K, Kp, dg);
  t[0][0] = p1;               t[0][1] = 1.0f - p1;        t[0][2] = 1.0f;
  t[1][0] = 1.0f / (L1+1.0f); t[1][1] = L1 / (L1+1.0f);   t[1][2] = 1.0f;
  pi[0] = 0.999; pi[1] = 0.001;
  for (int x = 0; x < K; x++) { eo[x][0] = p->bgf[x] / p->bgf[x]; eo[x][1] = p->compo[x] / p->bgf[x]; }
  eo[K][0] = eo[K][1] = 1.0f; eo[Kp-2][0] = eo[Kp-2][1] = 1.0f; eo[Kp-1][0] = eo[Kp-1][1] = 1.0f;
  for (int x = K+1; x <= Kp-3; x++)
    for (int k = 0; k < 2; k++) {
      float e = 0.f, denom = 0.f;
      for (int y = 0; y < K; y++) if (dg[x][y]) { e += (k == 0 ? p->bgf[y] : p->compo[y]); denom += p->bgf[y]; }
      eo[x][k] = (denom > 0.0f) ? e / denom : 0.0f;
    }
  if (L == 0) return log(0.0) ;  /* pi[M] = 0: never used (n==0 targets are skipped) */
  dp0 = eo[dsq[1]][0] * pi[0];
  dp1 = eo[dsq[1]][1] * pi[1];
  max = 0.0f; if (dp0 > max) max = dp0; if (dp1 > max) max = dp1;
  dp0 /= max; dp1 /= max;
  logsc = 0.0f;
  fsc = log(max); logsc += fsc;
  for (int i = 2; i <= L; i++) {
    n0 = 0.0f; n0 += dp0 * t[0][0]; n0 += dp1 * t[1][0]; n0 *= eo[dsq[i]][0];
    n1 = 0.0f; n1 += dp0 * t[0][1]; n1 += dp1 * t[1][1]; n1 *= eo[dsq[i]][1];
    max = 0.0f; if (n0 > max) max = n0; if (n1 > max) max = n1;
    dp0 = n0 / max; dp1 = n1 / max;
    fsc = log(max); logsc += fsc;
  }
  last = 0.0f; last += dp0 * t[0][2]; last += dp1 * t[1][2];
  last = log(last);
  logsc += last;
  return logsc + (float) L * logf(p1) + logf(1. - p1);
}

/* ---------------------------------------------------------------- statistics (Easel)
 * esl_gumbel_surv (libeasel/gumbel.pxd:6), esl_exp_surv / esl_exp_logsurv (libeasel/exponential.pxd:7-8)
 */
double p7o_gumbel_surv(double x, double mu, double lambda)
{
  double y  = lambda * (x - mu);
  double ey = -exp(-y);
  if (fabs(ey) < SMALLX1) return -ey;
  else                    return 1 - exp(ey);
}
double p7o_exp_surv(double x, double mu, double lambda)    { if (x < mu) return 1.0; return exp(-lambda * (x - mu)); }
double p7o_exp_logsurv(double x, double mu, double lambda) { if (x < mu) return 0.0; return -lambda * (x - mu); }

/* ---------------------------------------------------------------- the cascade
 * upstream p7_pipeline.c p7_Pipeline up to (not including) the Backward/domain-definition step
 * (reference include/libhmmer/p7_pipeline.pxd:130; called plan7.pyx:6442).
 */
int p7o_cascade(P7O_PROFILE *p, const uint8_t *dsq, int L, double F1, double F2, double F3, int do_bias, P7O_RECORD *rec)
{
  float usc, vfsc, fwdsc, filtersc, nullsc, seq_score;
  double P;
  int xJ = 0, xC = INT_MIN;
  memset(rec, 0, sizeof(*rec));
  rec->xC_vit = INT_MIN;
  if (L == 0) { rec->stage = -1; return 0; }
  p7o_reconfig_length(p, L);
  nullsc = p7o_null1(L);
  rec->nullsc = nullsc;
  p7o_msv(p, dsq, L, &usc, &xJ);
  rec->usc = usc; rec->xJ_msv = xJ;
  seq_score = (usc - nullsc) / LOG2;
  P = p7o_gumbel_surv(seq_score, p->evparam[p7_MMU], p->evparam[p7_MLAMBDA]);
  rec->P_msv = P;
  if (P > F1) { rec->stage = 0; return 0; }
  if (do_bias) {
    filtersc = p7o_bias_filter(p, dsq, L);
    seq_score = (usc - filtersc) / LOG2;
    P = p7o_gumbel_surv(seq_score, p->evparam[p7_MMU], p->evparam[p7_MLAMBDA]);
    rec->filtersc = filtersc; rec->P_bias = P;
    if (P > F1) { rec->stage = 1; return 0; }
  } else { filtersc = nullsc; rec->filtersc = filtersc; rec->P_bias = P; }
  if (P > F2) {
    p7o_vit(p, dsq, L, &vfsc, &xC);
    rec->vfsc = vfsc; rec->xC_vit = xC; rec->ran_vit = 1;
    seq_score = (vfsc - filtersc) / LOG2;
    P = p7o_gumbel_surv(seq_score, p->evparam[p7_VMU], p->evparam[p7_VLAMBDA]);
    rec->P_vit = P;
    if (P > F2) { rec->stage = 2; return 0; }
  }
  p7o_fwd(p, dsq, L, NULL, &fwdsc);
  rec->fwdsc = fwdsc;
  seq_score = (fwdsc - filtersc) / LOG2;
  P = p7o_exp_surv(seq_score, p->evparam[p7_FTAU], p->evparam[p7_FLAMBDA]);
  rec->P_fwd = P;
  if (P > F3) { rec->stage = 3; return 0; }
  rec->stage = 4;
  return 0;
}

int p7o_cascade_block(P7O_PROFILE *p, const uint8_t *dsq_concat, const int64_t *offsets, const int32_t *lengths,
                      size_t n, double F1, double F2, double F3, int do_bias, P7O_RECORD *recs, P7O_COUNTERS *ctr)
{
  P7O_RECORD r;
  memset(ctr, 0, sizeof(*ctr));
  for (size_t t = 0; t < n; t++) {
    p7o_cascade(p, dsq_concat + offsets[t] - 1, lengths[t], F1, F2, F3, do_bias, &r);
    if (recs) recs[t] = r;
    ctr->nseqs++; ctr->nres += (uint64_t) lengths[t];
    if (r.stage >= 1) ctr->n_past_msv++;
    if (r.stage >= 2) ctr->n_past_bias++;
    if (r.stage >= 3) ctr->n_past_vit++;
    if (r.stage >= 4) ctr->n_past_fwd++;
  }
  return 0;
}

int p7o_msv_block(P7O_PROFILE *p, const uint8_t *dsq_concat, const int64_t *offsets, const int32_t *lengths,
                  size_t n, int32_t *out_xJ)
{
  float sc; int xJ;
  for (size_t t = 0; t < n; t++) {
    p7o_reconfig_length(p, lengths[t]);
    p7o_msv(p, dsq_concat + offsets[t] - 1, lengths[t], &sc, &xJ);
    out_xJ[t] = xJ;
  }
  return 0;
}

/* ---------------------------------------------------------------- long targets (nhmmer)
 * upstream impl_sse/msvfilter.c p7_SSVFilter_longtarget (reference include/libhmmer/p7_pipeline.pxd:131-143, called by
 * p7_Pipeline_LongTarget from plan7.pyx:7628/7645): the sequential SSV scan of one strand block.  Un-striped u8
 * arithmetic; the choice among several cells at or above the threshold follows the order in which upstream unstripes
 * the row (vector q outer, byte z inner).  Emits (first residue of the diagonal, model node of its last cell, length)
 * per window seed.  Parity: restated from memory of upstream; pinned end to end by tests/golden/tables/bmyD{1,2}.tbl.
 */
static float p7o_null1_len(int L) { float p1 = (float) L / (float) (L + 1); return (float) L * logf(p1) + logf(1.0f - p1); }

int64_t p7o_ssv_longtarget(P7O_PROFILE *p, const uint8_t *dsq, int64_t L, int max_length, double F1, int64_t *seeds, int64_t cap)
{
  int M = p->M, Q = p->Q16;
  /* score threshold for P = F1 under the length model of max_length (upstream's comment block in p7_SSVFilter_longtarget) */
  double invP = p->evparam[p7_MMU] - log(-1.0 * log(1.0 - F1)) / p->evparam[p7_MLAMBDA];      /* esl_gumbel_invsurv */
  float nullsc = p7o_null1_len(max_length);
  int tjb = unbiased_byteify(p, logf(3.0f / (float) (max_length + 3)));
  int sc_thresh = (int) ceil(((nullsc + (invP * 0.69314718055994529) + 3.0) * p->scale_b) + p->base_b + p->tec_b + tjb);
  int bias = p->bias_b, xB = p->base_b - tjb - p->tbm_b; if (xB < 0) xB = 0;
  int *row = (int *) calloc(M + 1, sizeof(int)), *nxt = (int *) calloc(M + 1, sizeof(int));
  int64_t nseeds = 0;
  for (int64_t i = 1; i <= L; i++) {
    int x = dsq[i], hit = 0;
    for (int k = 1; k <= M; k++) {
      int sv = row[k-1] > xB ? row[k-1] : xB;
      sv += bias; if (sv > 255) sv = 255;
      sv -= rb_unstriped(p, x, k); if (sv < 0) sv = 0;
      nxt[k] = sv;
      if (sv >= sc_thresh) hit = 1;
    }
    { int *t = row; row = nxt; nxt = t; }
    if (!hit) continue;
    /* which model state hit the threshold: the best one, the first such in unstriping order */
    int end = -1, rem_sc = -1;
    for (int q = 0; q < Q; q++)
      for (int z = 0; z < 16; z++) {
        int k = q + Q * z + 1;
        if (k <= M && row[k] >= sc_thresh && row[k] > rem_sc) { end = k; rem_sc = row[k]; }
      }
    for (int k = 0; k <= M; k++) row[k] = 0;        /* values restart from xB in the next row */
    /* recover the diagonal that hit the threshold */
    int start = end, sc = rem_sc;
    int64_t target_end = i, target_start = i;
    while (rem_sc > p->base_b - tjb - p->tbm_b && start >= 1 && target_start >= 1) {
      rem_sc -= bias - rb_unstriped(p, dsq[target_start], start);
      --start; --target_start;
    }
    start++; target_start++;
    /* extend it forward while it keeps (nearly) rising */
    int k = end + 1; int64_t n = target_end + 1, max_end = target_end; int max_sc = sc, pos_since_max = 0;
    while (k < M && n <= L) {
      sc += bias - rb_unstriped(p, dsq[n], k);
      if (sc >= max_sc) { max_sc = sc; max_end = n; pos_since_max = 0; }
      else if (++pos_since_max == 5) break;
      k++; n++;
    }
    end += (int) (max_end - target_end);
    target_end = max_end;
    if (nseeds < cap) { seeds[3*nseeds] = target_start; seeds[3*nseeds+1] = end; seeds[3*nseeds+2] = end - start + 1; }
    nseeds++;
    i = target_end;                                 /* skip forward */
  }
  free(row); free(nxt);
  return nseeds;
}
