/* p7_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + SSE2 intrinsics, the ISA of the reference's
 * `impl_sse` build) of the HMMER 3.4 / Easel 0.49 algorithms that
 * pyhmmer's `plan7.Pipeline._search_loop` (reference
 * src/pyhmmer/plan7.pyx:6393-6453) reaches through `p7_Pipeline`
 * (reference include/libhmmer/p7_pipeline.pxd:130).
 *
 * The arithmetic itself is NOT under /root/reference (vendor/hmmer and
 * vendor/easel are empty, un-vendored submodules pinned at HMMER 3.4 /
 * Easel 0.49: reference src/hmmer/CMakeLists.txt:10-14,
 * src/easel/CMakeLists.txt:35-39).  Every function here restates the
 * published upstream algorithm and is pinned against the reference's own
 * fixtures (tests/golden/, see oracle/README.md):
 *   - striped int8/int16 tables: bit-exact vs the pressed .h3f/.h3p files,
 *   - float tables: bit-exact vs .h3p (Easel's vector expf polynomial),
 *   - filter cascade: survivors == golden hit lists of the .tbl files.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pyhmmer_amd/) never links or calls it.
 */
#ifndef P7_ORACLE_H
#define P7_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { p7O_E = 0, p7O_N = 1, p7O_J = 2, p7O_C = 3 };        /* impl_sse/p7_oprofile.pxd:29-33 */
enum { p7O_MOVE = 0, p7O_LOOP = 1 };                        /* impl_sse/p7_oprofile.pxd:35-38 */
enum { p7O_BM = 0, p7O_MM, p7O_IM, p7O_DM, p7O_MD, p7O_MI, p7O_II, p7O_DD }; /* :41-49 */
enum { p7_MMU = 0, p7_MLAMBDA, p7_VMU, p7_VLAMBDA, p7_FTAU, p7_FLAMBDA };  /* libhmmer/__init__.pxd:30-37 */
enum { p7H_MM = 0, p7H_MI, p7H_MD, p7H_IM, p7H_II, p7H_DM, p7H_DD };       /* p7_hmm.pxd transitions */
/* generic profile transition order (upstream p7_profile.h) */
enum { p7P_MM = 0, p7P_IM, p7P_DM, p7P_BM, p7P_MD, p7P_DD, p7P_MI, p7P_II };

#define P7O_EXTRA_SB 17
#define P7O_MAXK 20
#define P7O_MAXKP 29

typedef struct p7o_profile {
  int M, K, Kp;
  int Q16, Q8, Q4;
  int L;                 /* current length configuration */
  float nj;
  /* generic log-odds profile (p7_ProfileConfig) */
  float *tsc;            /* [(M+1)*8], p7P_* order */
  float *msc;            /* [Kp][M+1] match scores (insert scores are hard-wired 0) */
  float xsc[4][2];       /* [E,N,J,C][MOVE,LOOP] */
  /* MSV / SSV */
  uint8_t *rbv;          /* [Kp][Q16*16] striped */
  int8_t  *sbv;          /* [Kp][(Q16+17)*16] striped */
  uint8_t tbm_b, tec_b, tjb_b, base_b, bias_b;
  float scale_b;
  /* Viterbi */
  int16_t *rwv;          /* [Kp][Q8*8] */
  int16_t *twv;          /* [8*Q8*8] */
  int16_t xw[4][2];
  float scale_w; int16_t base_w, ddbound_w; float ncj_roundoff;
  /* Forward/Backward */
  float *rfv;            /* [Kp][Q4*4] */
  float *tfv;            /* [8*Q4*4] */
  float xf[4][2];
  /* stats + bias filter */
  float evparam[6];
  float compo[P7O_MAXK];
  float bgf[P7O_MAXK];
} P7O_PROFILE;

typedef struct p7o_record {
  float usc, filtersc, nullsc, vfsc, fwdsc;   /* nats */
  double P_msv, P_bias, P_vit, P_fwd;
  int32_t xJ_msv;        /* raw integer MSV result (xJ), -1 if overflow */
  int32_t xC_vit;        /* raw integer Viterbi xC, 32767 if overflow, -32768 if -inf; INT32_MIN if not run */
  int32_t stage;         /* 0: failed MSV, 1: failed bias, 2: failed Vit, 3: failed Fwd, 4: passed Fwd */
  int32_t ran_vit;
} P7O_RECORD;

typedef struct p7o_counters {
  uint64_t nseqs, nres, n_past_msv, n_past_bias, n_past_vit, n_past_fwd;
} P7O_COUNTERS;

P7O_PROFILE *p7o_profile_build(int M, int K, const float *t, const float *mat,
                               const float *bgf, const float *compo,
                               const float *evparam, int L);
void  p7o_profile_free(P7O_PROFILE *p);
void  p7o_reconfig_length(P7O_PROFILE *p, int L);

/* dsq is 1-indexed: dsq[1..L] residues, dsq[0] and dsq[L+1] sentinels (never read) */
int   p7o_msv(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xJ);
int   p7o_msv_scalar(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xJ);
int   p7o_vit(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xC);
int   p7o_vit_scalar(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *ret_sc, int *ret_xC);
/* xmx: NULL or (L+1)*6 floats [E,N,J,B,C,SCALE] per row */
int   p7o_fwd(const P7O_PROFILE *p, const uint8_t *dsq, int L, float *xmx, float *ret_sc);
int   p7o_bck(const P7O_PROFILE *p, const uint8_t *dsq, int L, const float *fwd_xmx, float *bck_xmx, float *ret_sc);
float p7o_null1(int L);
float p7o_bias_filter(const P7O_PROFILE *p, const uint8_t *dsq, int L);
double p7o_gumbel_surv(double x, double mu, double lambda);
double p7o_exp_surv(double x, double mu, double lambda);
double p7o_exp_logsurv(double x, double mu, double lambda);

int   p7o_cascade(P7O_PROFILE *p, const uint8_t *dsq, int L, double F1, double F2, double F3,
                  int do_bias, P7O_RECORD *rec);
/* packed block: residues of target t are dsq_concat[offsets[t] .. offsets[t]+lengths[t]-1], with
 * at least one byte of padding before and after each target. */
int   p7o_cascade_block(P7O_PROFILE *p, const uint8_t *dsq_concat, const int64_t *offsets,
                        const int32_t *lengths, size_t n, double F1, double F2, double F3,
                        int do_bias, P7O_RECORD *recs /* may be NULL */, P7O_COUNTERS *ctr);
/* MSV only over a block (for cpu_baseline timing + parity of raw xJ); out_xJ[t] = xJ or -1 on overflow */
int   p7o_msv_block(P7O_PROFILE *p, const uint8_t *dsq_concat, const int64_t *offsets,
                    const int32_t *lengths, size_t n, int32_t *out_xJ);

void  p7o_expf_neg(const double *in, float *out, size_t n);   /* expf(-1.0*v), '*' encoded as +inf */
float p7o_sse_expf_scalar(float x);

/* long targets: upstream p7_SSVFilter_longtarget for one strand block; seeds = cap x (first residue, last node, length) */
int64_t p7o_ssv_longtarget(P7O_PROFILE *p, const uint8_t *dsq, int64_t L, int max_length, double F1, int64_t *seeds, int64_t cap);

/* long targets, behind the SSV scan (p7_oracle_lt.c): one strand block through window merging, the MSV / bias tests, the
 * long-target Viterbi scan and the Forward test; the scoring of an envelope; the background an envelope is rescored against */
int64_t p7o_lt_block(P7O_PROFILE *p, const uint8_t *dsq, int64_t L, int max_length, double F1, double F2, double F3,
                     int B1, int B2, int B3, int do_bias, double *out, int64_t cap, uint64_t *counts);
float p7o_lt_domain_score(const P7O_PROFILE *p, int max_length, int64_t env_len, int64_t ali_len,
                          float envsc, float domcorrection, int do_null2, float *ret_bias_bits, double *ret_lnP);
int   p7o_lt_envelope_scores(const P7O_PROFILE *p, const uint8_t *env, int n, int64_t window_len, const uint8_t *degen,
                             float *orig, float *adj);
void  p7o_lt_envelope_background(const P7O_PROFILE *p, const uint8_t *env, int64_t n_env, int64_t window_len,
                                 const uint8_t *degen, float *bg_out);

/* domain definition (p7_oracle_dd.c): regions from the parsers' rows; a region that holds one domain is rescored as it is,
 * a region that holds several through the ensemble of 200 sampled tracebacks, their clustering and the null2 scores by
 * trace (seed != 0 and ensembles != 0; else such regions are only counted).  out: 13 doubles per envelope (ienv jenv iali
 * jali hmmfrom hmmto envsc domcorrection oasc bitscore dombias lnP kind); counts: regions, envelopes, ensemble regions,
 * clusters, overlapping clusters.  seqout (or NULL), given the parser's Forward score: the sequence's bit score, pre-score,
 * sum-of-domains score, ln P, number of domains, their length (p7_pipeline.c).  degen: [Kp][K] residue-code sets. */
int64_t p7o_domains(P7O_PROFILE *p, const uint8_t *dsq, int L, const float *fx, const float *bx, const uint8_t *degen,
                    int do_null2, uint32_t seed, int ensembles, double *out, int64_t cap, int64_t *counts, float fwdsc, double *seqout);

/* domain definition of one Forward-passing window of a long target (rescore_isolated_domain with long_target = TRUE: own
 * length model, composition-adjusted emissions, the envelope cut back to its alignment + max_env_extra): rows as p7o_domains,
 * columns 0-8 and 12, window coordinates. */
int64_t p7o_lt_domains(P7O_PROFILE *p, const uint8_t *win, int W, const float *fx, const float *bx, const uint8_t *degen,
                       int do_null2, uint32_t seed, int max_env_extra, double *out, int64_t cap, int64_t *counts);

/* the alignment display of an envelope's optimal-accuracy alignment (p7_alidisplay_Create): model / match / sequence /
 * posterior lines, one column per state between B and E.  consensus[1..M], sym = the alphabet's symbols. */
int p7o_domain_alignment(P7O_PROFILE *p, const uint8_t *dsq, int L, int ienv, int jenv, const char *consensus, const char *sym,
                         char *model, char *mline, char *aseq, char *ppline, int cap);

#ifdef __cplusplus
}
#endif
#endif
