/* p7x.h -- C ABI of libp7x, the MI355X-native replacement for the libhmmer calls that
 * pyhmmer's plan7.Pipeline makes on its hot path.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the
 * pyhmmer tree, v0.12.3).  Plain pointers and sizes only: no torch types, no C++ types.
 * All functions return an Easel-style status (include/libeasel/__init__.pxd:29-58):
 *   P7X_OK 0, P7X_EINVAL 11 (-> MissingCutoffs, plan7.pyx:6424-6425), P7X_ERANGE 16
 *   (-> OverflowError, plan7.pyx:6445-6446), anything else -> UnexpectedError (plan7.pyx:6447-6448).
 * P7X_ENODEVICE is returned -- never a CPU fallback -- when a device entry point is called
 * and no HIP device is usable.
 *
 * Ownership: inputs are borrowed for the duration of the call (as plan7.pyx:5116-5119);
 * handles returned by *_create() are owned by the caller and released with *_destroy().
 * Thread-safety: one p7x_pipeline per host thread (as hmmer/_base.py:324-325); profiles and
 * sequence databases are immutable after creation and may be shared between pipelines
 * (the per-target length model is computed inside the kernels instead of mutating the
 * profile as p7_oprofile_ReconfigLength does, plan7.pyx:6438).
 */
#ifndef P7X_H
#define P7X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P7X_ABI_VERSION 8

enum {
  P7X_OK = 0, P7X_EMEM = 5, P7X_EFORMAT = 7, P7X_EINVAL = 11, P7X_ERANGE = 16, P7X_ENORESULT = 19,
  P7X_ENODEVICE = 100, P7X_EDEVICE = 101
};

enum { P7X_RNA = 1, P7X_DNA = 2, P7X_AMINO = 3 };                 /* libeasel/alphabet.pxd */
enum { P7X_SEARCH_SEQS = 0, P7X_SCAN_MODELS = 1 };                /* p7_pipeline.pxd:26-28 */
enum { P7X_ZSETBY_NTARGETS = 0, P7X_ZSETBY_OPTION = 1, P7X_ZSETBY_FILEINFO = 2 }; /* p7_pipeline.pxd:30-33 */
enum { P7X_MMU = 0, P7X_MLAMBDA, P7X_VMU, P7X_VLAMBDA, P7X_FTAU, P7X_FLAMBDA };  /* libhmmer/__init__.pxd:30-37 */
enum { P7X_GA1 = 0, P7X_GA2, P7X_TC1, P7X_TC2, P7X_NC1, P7X_NC2 };               /* libhmmer/__init__.pxd:39-46 */
#define P7X_CUTOFF_UNSET (-99999.0f)
#define P7X_EVPARAM_UNSET (-99999.0f)
enum { P7X_IS_INCLUDED = 1, P7X_IS_REPORTED = 2, P7X_IS_NEW = 4, P7X_IS_DROPPED = 8, P7X_IS_DUPLICATE = 16 }; /* p7_tophits.pxd:13-18 */
enum { P7X_BITCUT_NONE = 0, P7X_BITCUT_GA = 1, P7X_BITCUT_NC = 2, P7X_BITCUT_TC = 3 }; /* p7_pipeline.pxd p7_pipemodes */

/* ------------------------------------------------------------------ host utilities */

int p7x_abi_version(void);

/* expf(-v) exactly as upstream read_asc30hmm parses an HMMER3/f ASCII save file
 * ('*' is passed as +inf and yields 0).  Replaces the C parser behind HMMFile (plan7.pyx:3656-4050). */
void p7x_expf_neg(const double *in, float *out, size_t n);

/* ------------------------------------------------------------------ profiles
 * p7x_hmm_view mirrors the P7_HMM fields the path needs (include/libhmmer/p7_hmm.pxd:48-78). */
typedef struct p7x_hmm_view {
  int32_t M;
  int32_t abc_type;            /* P7X_AMINO | P7X_DNA | P7X_RNA */
  const float *t;              /* [(M+1)*7] MM,MI,MD,IM,II,DM,DD probabilities */
  const float *mat;            /* [(M+1)*K] */
  const float *ins;            /* [(M+1)*K] */
  const float *compo;          /* [K] or NULL (model had no COMPO line) */
  float evparam[6];
  float cutoff[6];
  int32_t max_length;
  const char *name;            /* required */
  const char *acc;             /* or NULL */
  const char *desc;            /* or NULL */
  const char *consensus;       /* [M+2] as P7_HMM.consensus, or NULL */
  const char *rf, *mm, *cs;    /* optional annotation lines [M+2], or NULL */
} p7x_hmm_view;

typedef struct p7x_oprofile p7x_oprofile;   /* opaque: replaces P7_PROFILE + P7_OPROFILE */

/* p7_ProfileConfig (modelconfig.pxd:7-10; plan7.pyx:8082) followed by p7_oprofile_Convert
 * (impl_sse/p7_oprofile.pxd:121; plan7.pyx:4961), local multihit mode, length model L.
 * bg_f is P7_BG.f (p7_bg.pxd:10-30); K floats. */
int  p7x_oprofile_create(const p7x_hmm_view *hmm, const float *bg_f, int32_t L, p7x_oprofile **out);
/* p7_Builder_MaxLength(hmm, emit_thresh) (include/libhmmer/p7_builder.pxd; called by LongTargetsPipeline.search_hmm,
 * plan7.pyx:7346-7354 with window_beta): the smallest length W such that the core model, entered at its first match
 * state, emits a W-th residue with probability below <beta>.  Reproduces the MAXL lines hmmbuild wrote into the
 * nucleotide fixtures (beta = 1e-7: RF00001 305, bmyD 1736).  P7X_ERANGE if no such length below 200,000. */
int  p7x_hmm_max_length(const p7x_hmm_view *hmm, double beta, int32_t *out);
void p7x_oprofile_destroy(p7x_oprofile *om);

typedef struct p7x_oprofile_info {          /* scalars of P7_OPROFILE (impl_sse/p7_oprofile.pxd:52-108) */
  int32_t M, K, Kp, abc_type, L, max_length, mode;
  int32_t Q16, Q8, Q4;                       /* p7O_NQB/NQW/NQF (p7_oprofile.pxd:24-26) */
  uint8_t tbm_b, tec_b, tjb_b, base_b, bias_b;
  float   scale_b;
  int16_t xw[4][2];                          /* [E,N,J,C][MOVE,LOOP] */
  float   scale_w; int16_t base_w, ddbound_w; float ncj_roundoff;
  float   xf[4][2];
  float   evparam[6], cutoff[6], compo[20];
  float   nj;
} p7x_oprofile_info;
int p7x_oprofile_get_info(const p7x_oprofile *om, p7x_oprofile_info *info);

/* Striped (Farrar) views exactly as impl_sse stores them, for the OptimizedProfile properties
 * rbv/sbv/rwv/twv/rfv/tfv (plan7.pyx:4623-4813) and for checking against pressed .h3f/.h3p files.
 * which: 0 rbv u8 [Kp][Q16*16]; 1 sbv i8 [Kp][(Q16+17)*16]; 2 rwv i16 [Kp][Q8*8]; 3 twv i16 [8*Q8*8];
 *        4 rfv f32 [Kp][Q4*4]; 5 tfv f32 [8*Q4*4].  Returns bytes written, or -1. */
int64_t p7x_oprofile_striped(const p7x_oprofile *om, int which, void *out, size_t out_bytes);

/* Pressed profiles: one record of a `.h3f` (MSV part) and of a `.h3p` (everything else) file, the formats
 * `hmmpress` writes (reference hmmer/_hmmpress.py:29-66, OptimizedProfile.write plan7.pyx:5078-5105,
 * HMMPressedFile plan7.pyx:4051-4197).  write: pass NULL buffers to obtain the sizes; offs = byte offsets of this
 * model in the .h3m/.h3f/.h3p files (NULL: zeros).  read: parses one record from each buffer and returns the bytes
 * consumed; the log-odds tables come out exactly as stored. */
int p7x_oprofile_write_pressed(const p7x_oprofile *om, const int64_t offs[3], uint8_t *h3f, size_t cap_f, size_t *len_f,
                               uint8_t *h3p, size_t cap_p, size_t *len_p);
/* name (0), accession (1), description (2), consensus line with its leading pad (3): copies at most n-1 characters,
 * returns the full length, 0 when absent, -1 on bad arguments. */
int p7x_oprofile_get_string(const p7x_oprofile *om, int which, char *buf, size_t n);
int p7x_oprofile_read_pressed(const uint8_t *h3f, size_t nf, const uint8_t *h3p, size_t np, const float *bg_f,
                              p7x_oprofile **out, size_t *used_f, size_t *used_p, int64_t offs[3]);

/* FASTA text -> packed block (the input format of p7x_seqdb_create) in one pass; the reference reads sequence files
 * through Easel's C parser (easel.pyx SequenceFile.read_block, 8168-8192).  lut[256]: digital code per character,
 * 255 = illegal, 254 = ignored.  First call with dsq == NULL returns the sizes (nseq, nres, bytes of the string
 * table); second call fills dsq[nres + nseq + 1], offsets/lengths[nseq], strtab (NUL-terminated name then
 * description per record) and its two index arrays.  P7X_EFORMAT + *bad_pos on an illegal character. */
int p7x_fasta_parse(const char *text, size_t n, const uint8_t *lut, size_t *nseq, size_t *nres, size_t *strbytes,
                    uint8_t *dsq, int64_t *offsets, int32_t *lengths, char *strtab, int64_t *name_off, int64_t *desc_off,
                    size_t *bad_pos);

/* ------------------------------------------------------------------ devices */
int p7x_device_count(void);                       /* 0 when no HIP device is usable */
int p7x_device_name(int device, char *buf, size_t n);

/* ------------------------------------------------------------------ sequence database (device resident)
 * Replaces the `const ESL_SQ**` array of a DigitalSequenceBlock handed to _search_loop
 * (plan7.pyx:6393-6398; easel.pyx:8168-8192).  dsq holds the residues of target t at
 * dsq[offsets[t] .. offsets[t]+lengths[t]-1], digital codes 0..Kp-1.  Packed once, searched
 * by any number of profiles. */
typedef struct p7x_seqdb p7x_seqdb;
int  p7x_seqdb_create(int device, int32_t abc_type, const uint8_t *dsq, const int64_t *offsets,
                      const int32_t *lengths, size_t n, p7x_seqdb **out);
void p7x_seqdb_destroy(p7x_seqdb *db);
int64_t p7x_seqdb_ntargets(const p7x_seqdb *db);
int64_t p7x_seqdb_nresidues(const p7x_seqdb *db);

/* ------------------------------------------------------------------ single-stage entry points (unit-test seam)
 * Mirror OptimizedProfile.msv_filter / ssv_filter (plan7.pyx:4969-5070): one digital sequence,
 * score in nats; P7X_ERANGE + *sc=+inf on overflow.  dsq[0..L-1] are the residues (no sentinels). */
int p7x_msv_filter(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc);
int p7x_vit_filter(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc);
int p7x_fwd_parser(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc);
int p7x_bck_parser(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc);

/* Batched raw filter outputs over a whole database (parity tests and bench).
 * xJ[t]  : integer MSV end state (p7_MSVFilter's xJ), -1 on overflow.
 * xC[t]  : integer Viterbi end state (p7_ViterbiFilter's xC), 32767 on overflow.
 * fwd[t] : Forward score in nats (p7_ForwardParser). Any output pointer may be NULL. */
int p7x_filters_batch(const p7x_oprofile *om, const p7x_seqdb *db, int32_t *xJ, int32_t *xC,
                      float *fwd, float *bias_filtersc);

/* ------------------------------------------------------------------ the pipeline
 * p7x_pipeline_cfg mirrors the P7_PIPELINE fields reachable through the Python property
 * setters (p7_pipeline.pxd:58-107; plan7.pyx:5635-5950). */
typedef struct p7x_pipeline_cfg {
  int32_t by_E; double E, T; int32_t dom_by_E; double domE, domT; int32_t use_bit_cutoffs;
  int32_t inc_by_E; double incE, incT; int32_t incdom_by_E; double incdomE, incdomT;
  double  Z, domZ; int32_t Z_setby, domZ_setby;
  double  F1, F2, F3;
  int32_t do_max, do_biasfilter, do_null2;
  uint32_t seed;             /* do_reseeding = (seed != 0), plan7.pyx:5684-5688 */
  int32_t mode;              /* P7X_SEARCH_SEQS | P7X_SCAN_MODELS */
  int32_t host_threads;      /* workers for host-side domain definition; 0 = hardware_concurrency */
  int32_t host_envelopes;    /* 0 (default): single-domain envelopes are rescored by the device kernel; 1: on the host.  Long targets:
                              * 0 the device when the number of envelopes makes it pay (a few long envelopes are faster on the host
                              * workers), 1 always the host, 2 always the device */
  int32_t host_regions;      /* 0 (default): posterior decoding of the specials + region scan on the device; 1: on the host */
  /* long targets (nhmmer): p7_pipeline.pxd:80-87, 103-107; LongTargetsPipeline.__init__ plan7.pyx:6957-7060 */
  int32_t long_targets;      /* p7_Pipeline_LongTarget semantics: every domain is a hit, E-values from residues searched */
  int32_t strands;           /* P7X_STRAND_BOTH | _TOPONLY | _BOTTOMONLY */
  int32_t B1, B2, B3;        /* window lengths of the biased-composition modifier for the MSV / Viterbi / Forward filters */
  int32_t block_length;      /* residues per block read from a long target (W); consecutive blocks overlap by max_length */
  int32_t window_length;     /* > 0: overrides the model's max_length (nhmmer --w_length) */
  int32_t evalue_window_length; /* > 0: the window length p7_tophits_ComputeNhmmerEvalues is given when it differs from the scan's
                              * (an HMM query without window_length: p7_Builder_MaxLength(hmm, window_beta), plan7.pyx:7346-7354,
                              * while the scan keeps the max_length the optimized profile was built with); <= 0: the scan's */
  int32_t lt_part, lt_nparts; /* nhmmer over several devices: this call takes part lt_part of lt_nparts of the (target, block, strand)
                              * units of the search -- consecutive units, units counted in the order of the reference's loop
                              * (plan7.pyx:7582-7655) -- and returns an unfinished hit list; p7x_tophits_merge_longtargets finishes the
                              * parts together (E-values for all residues searched, duplicates, thresholds).  Default 0 of 1 */
  float   oa_guard;          /* near-tie guard of the device's optimal-accuracy traceback: with g > 0 a choice on the trace between
                              * candidates within |v| * g + g of each other (or a posterior that close to the next printed digit) sends
                              * the envelope to the host stage, which repeats it with Forward / Backward / null2 summed in UPSTREAM's
                              * order (impl_sse's four stripes: forward_full_upstream, p7x_domaindef.cpp), so that a decision the
                              * device's summation order cannot be trusted with is the reference's.  Default 4e-6 since ABI 8 (about
                              * 2 % of the envelopes; scripts/oa_guard_sweep2.py: no differing domain left at 3e-6 over 26,000); 0:
                              * off, and a kernel without the guard's arithmetic (the device is then compared with its own host twin,
                              * option "host_order" = 1, which sums in the device's order) */
  float   f3_guard;          /* relative half-width of the band around F3 inside which a target's Forward P-value is not trusted to the
                              * device's summation order: the device passes P <= F3 (1 + g), the host stage re-scores the targets with
                              * P > F3 (1 - g) with p7x_forward_parser_exact and applies F3 to that.  Default 4e-3 (a band of about 6e-3 bit, three times the stated tolerance of the device Forward score); 0: no guard */
  uint64_t lt_resident_key;  /* p7x_search_longtargets: nonzero = the caller's promise that a target buffer passed with this key always
                              * has the same contents (a token of the packed image, not of its address).  The device copy is then kept
                              * after the call and a later search with the same key and size does not upload the targets again (one
                              * copy per device; a new key replaces it).  0 (default): upload per call, nothing kept */
  int32_t host_ensembles;    /* 0 (default): the stochastic traceback ensembles of multi-domain regions (p7_domaindef.c
                              * region_trace_ensemble: 200 sampled tracebacks per region, p7_Null2_ByTrace of every sampled domain) run
                              * on the device and the clustered envelopes are rescored there in a second round; 1: on the host
                              * workers.  Identical results (one generator stream per region, the same choices: p7x_choice.hpp).
                              * Without re-seeding (seed 0) the regions of a search share one stream and the host samples them. */
  float   ens_guard;         /* ABI 8.  Near-threshold guard of the device's stochastic tracebacks: a choice of a sampled traceback
                              * whose deviate lies within g (as a fraction of the deviate's range) of one of its thresholds -- they
                              * come from a Forward matrix summed in the device's order -- flags the region, and the host stage
                              * samples flagged regions itself from a Forward matrix in upstream's order (status bit 6 of
                              * EnsembleResult).  Default 2.5e-7 = 2^-22: between the two orders no threshold of a reachable cell was seen
                              * further apart than 2^-21, and 5 in a million further than 2^-22 (scripts/order_spread.py, 3.6e8
                              * thresholds, profiles/r06_order_spread.txt); 0: off */
} p7x_pipeline_cfg;
enum { P7X_STRAND_BOTH = 0, P7X_STRAND_TOPONLY = 1, P7X_STRAND_BOTTOMONLY = 2 };
void p7x_pipeline_cfg_default(p7x_pipeline_cfg *cfg);   /* p7_pipeline_Create(NULL,...) defaults, plan7.pyx:5413-5421 */

typedef struct p7x_counters {   /* accounting, p7_pipeline.pxd:88-101 */
  uint64_t nmodels, nseqs, nres, nnodes;
  uint64_t n_past_msv, n_past_bias, n_past_vit, n_past_fwd;
  uint64_t n_output, pos_past_msv, pos_past_bias, pos_past_vit, pos_past_fwd, pos_output;
} p7x_counters;

typedef struct p7x_domain {     /* P7_DOMAIN, p7_domain.pxd:10-26 */
  int64_t ienv, jenv, iali, jali, iorf, jorf;
  float envsc, domcorrection, dombias, oasc, bitscore;
  double lnP;
  int32_t is_reported, is_included;
  /* alignment display, P7_ALIDISPLAY p7_alidisplay.pxd:26-53 (strings owned by the tophits handle) */
  int32_t N, hmmfrom, hmmto, M;
  int64_t sqfrom, sqto, L;
  const char *model, *mline, *aseq, *ppline, *rfline, *mmline, *csline;
  const char *hmmname, *hmmacc, *hmmdesc, *sqname, *sqacc, *sqdesc;
} p7x_domain;

typedef struct p7x_hit {        /* P7_HIT, p7_hit.pxd:27-58 */
  const char *name, *acc, *desc;
  int64_t seqidx;               /* index of the target in the searched block */
  int32_t window_length;
  double sortkey;
  float score, pre_score, sum_score;
  double lnP, pre_lnP, sum_lnP;
  float nexpected;
  int32_t nregions, nclustered, noverlaps, nenvelopes, ndom;
  uint32_t flags;
  int32_t nreported, nincluded, best_domain;
} p7x_hit;

typedef struct p7x_tophits p7x_tophits;   /* opaque: replaces P7_TOPHITS + the copied P7_PIPELINE (plan7.pyx:8813-8817) */

/* Pipeline._search_loop (plan7.pyx:6393-6453): p7_pli_NewModel once, then for every target
 * p7_pli_NewSeq / p7_bg_SetLength / p7_oprofile_ReconfigLength / p7_Pipeline / p7_pipeline_Reuse;
 * followed by p7_tophits_SortBySortkey + p7_tophits_Threshold (plan7.pyx:6255-6256).
 * targets' names/accessions/descriptions are NUL-separated string tables indexed by target
 * (may be NULL: hits then carry only seqidx). */
int p7x_search_block(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const float *bg_f,
                     const p7x_seqdb *db, const char *const *names, const char *const *accs,
                     const char *const *descs, p7x_tophits **out);

/* The same search in two stages, for callers that overlap consecutive queries (the reference runs its queries on
 * worker threads, _hmmsearch.py:294-436): begin = filters + parsers on the device (blocks until they are done),
 * finish = domain definition + hit list; finish consumes the handle.  begin/finish of different searches may run
 * on different host threads at the same time. */
typedef struct p7x_pending p7x_pending;
int  p7x_search_block_begin(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const float *bg_f,
                            const p7x_seqdb *db, p7x_pending **out);
int  p7x_search_block_finish(p7x_pending *pending, const char *const *names, const char *const *accs,
                             const char *const *descs, p7x_tophits **out);
void p7x_pending_destroy(p7x_pending *pending);
/* begin in two halves, for a caller that keeps several searches of one host thread in flight (hmmscan: one per model,
 * the OptimizedProfileBlock loop of plan7.pyx:6624-6677 turned inside out): enqueue queues stage 1 on its own device
 * stream and returns at once; wait blocks until that work is done (idempotent; finish calls it when the caller did
 * not).  Both halves of one search, and destroy of an un-waited handle, belong to the thread that called enqueue. */
int  p7x_search_block_enqueue(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const float *bg_f,
                              const p7x_seqdb *db, p7x_pending **out);
int  p7x_search_block_wait(p7x_pending *pending);
/* A batch of query profiles against one resident target block: the many-query loops of the reference -- hmmsearch
 * over an iterable of HMMs (_hmmsearch.py:294-436, one Pipeline.search_hmm per query) and Pipeline._scan_loop over an
 * OptimizedProfileBlock (plan7.pyx:5072-5338, 6624-6677) -- as ONE set of device launches: every kernel of the cascade
 * takes the profile of its work item from the batch (blockIdx.y), so a launch serves all nq profiles.  Results are
 * those of nq separate searches (tests/test_gpu_search.py).  oms[q] may have any lengths; out of finish is a caller
 * array of nq hit lists, in the order of oms.  wait is p7x_search_block_wait; the threading rules are the same. */
int  p7x_search_batch_enqueue(const p7x_pipeline_cfg *cfg, const p7x_oprofile *const *oms, size_t nq, const float *bg_f,
                              const p7x_seqdb *db, p7x_pending **out);
int  p7x_search_batch_finish(p7x_pending *pending, const char *const *names, const char *const *accs,
                             const char *const *descs, p7x_tophits **outs);
size_t p7x_pending_nqueries(const p7x_pending *pending);
/* Test seam of the bias filter's logarithm: out[i] = the device's (float) log((double) in[i]) as bias_kernel takes it at every
 * residue (a table + series evaluation in double; claimed to round to the same float as the C library's logarithm, which
 * is what the reference's esl_hmm_Forward calls: tests/test_gpu_filters.py checks every float of [2^-7, 2^7)). */
int  p7x_debug_log_of_float(int device, const float *in, float *out, size_t n);
/* Test seams of the stochastic traceback ensembles (p7_domaindef.c region_trace_ensemble; p7_domaindef.pxd:23-59).
 * p7x_debug_choice: one choice point of p7_StochasticTrace with n paths of weights p[], for the generator state x after the
 *   draw: the path taken through the integer thresholds the product uses and through esl_rnd_FChoose as the reference
 *   writes it (they must agree for every p and x).
 * p7x_debug_ensemble: the 200 sampled tracebacks of region i..j (1-based) of one target of a resident block, on the device
 *   (use_device != 0) or by the host twin: dom[ndom][5] = sample, first / last residue inside the region, first / last
 *   node, a sample's domains first to last; n2[pos], pos = 1..j-i+1: the summed null2 odds ratios of the residue;
 *   status 0 = sampled. */
int  p7x_debug_choice(const float *p, int n, uint32_t x, int *via_thresholds, int *via_fchoose);
/* Calibration seam of the ensemble walk's near-threshold guard: the Forward matrix of region i..j of dsq1[1..L] in the device's
 * summation order and in upstream's, and for every cell the integer thresholds of its choice points in both: out12[0] =
 * thresholds compared, [1] = largest difference / 2^32, [2 + b] = how many differ by more than 2^-(24 - b), b = 0..9. */
int  p7x_debug_order_spread(const p7x_oprofile *om, const uint8_t *dsq1, int32_t L, int32_t i, int32_t j, int multihit, double *out12);
/* Test seam of the long-target SSV scan's tables (host code, no device): the registers per lane R the kernel takes for this
 * model, the most a cell can lose in one row with a canonical residue (byte units), and the table it stages in LDS --
 * [parity][x < 4][q][lane][c] packed pairs (lo, hi) of bias - rb[x][k], register j = 4q + c of the lane holding nodes
 * (2g - 1, 2g) on odd rows (parity 0) and (2g, 2g + 1) on even rows, g = lane * R + j; <pair> != 0: with the virtual node
 * M + 1 (emission 0) of the every-second-row flavour.  Returns the number of 32-bit words of the table (written when
 * cap_words is large enough), or a negative status. */
int64_t p7x_debug_ssv_tables(const p7x_oprofile *om, int pair, int32_t *R, int32_t *pair_slack, uint32_t *tab4q, size_t cap_words);
/* Test / diagnostic seam: process-wide knobs, by name; value -1 = not set (the library decides).  The library itself reads no
 * environment variables.  Kernel families for parity tests: "small_block" (0: never the wave-per-target filters for small
 * blocks), "vit_wave" (1: the wave-per-target Viterbi kernel for every model), "msv_exact" (1: no fast MSV pass),
 * "msv_long_groups" (0: the longest groups stay with the lane kernel), "msv_blocks_per_cu" (cap), "device_clustered" (0 / 1:
 * with host ensembles, where their clustered envelopes are rescored), "env_workspace_gb" (cap of the envelope kernel's
 * workspace), "ssv_kernel" (long-target SSV scan: 3 = the row maximum in every row, 4 = in every second row whatever the
 * model).  Traces on stderr: "trace_finish", "trace_longtarget", "trace_envelope", "host_profile" (1: on). */
int  p7x_debug_set_option(const char *name, int value);
int  p7x_debug_ensemble(const p7x_oprofile *om, const p7x_seqdb *db, int64_t target, int32_t i, int32_t j, uint32_t seed,
                        int use_device, int32_t *ndom, int32_t *dom, int32_t dom_cap, float *n2, int32_t *status);
/* Parity seam of the batched cascade (the multi-profile twin of p7x_filters_batch): runs stage 1 exactly as
 * p7x_search_batch_enqueue queues it -- profiles grouped into kernel classes, the hybrid lane / wave MSV split, the work
 * lists -- and returns what the stages left behind, [nq][ntargets] in caller order of both: xJ (every target; -1
 * overflow), xC of the targets that reached the Viterbi filter (INT32_MIN elsewhere; 32767 overflow) and the last
 * stage each target passed (0 none, 1 MSV, 2 bias, 3 Viterbi, 4 Forward).  Any output may be NULL. */
int  p7x_search_batch_raw(const p7x_pipeline_cfg *cfg, const p7x_oprofile *const *oms, size_t nq, const float *bg_f,
                          const p7x_seqdb *db, int32_t *xJ, int32_t *xC, uint8_t *stage);

/* nhmmer: LongTargetsPipeline.search_hmm / _search_loop_longtargets (plan7.pyx:7272-7418, 7541-7664) for a block of
 * long DNA / RNA targets.  Per target and strand: the SSV scan of p7_Pipeline_LongTarget (p7_pipeline.pxd:131-143) runs
 * on the device over the whole strand (reverse complement formed on the fly); the windows it seeds go through
 * p7_Pipeline_LongTarget's tail on the host (window MSV + bias, long-target Viterbi, Forward / Backward, domain
 * definition with long_target = TRUE, one hit per domain), followed by p7_tophits_ComputeNhmmerEvalues, target lengths,
 * p7_tophits_RemoveDuplicates, sort and threshold (plan7.pyx:7390-7412).  Targets: residues dsq[offsets[t] ..
 * offsets[t] + lengths[t] - 1] (64-bit lengths: chromosomes); cfg.long_targets must be set.
 * Records that lie end to end in dsq with one sentinel (a code outside the alphabet) between them -- the layout of
 * p7x_seqdb_create -- are scanned where they are, all of them with one launch (the sentinel ends every diagonal); any
 * other layout is packed that way first.  The window stages run one device batch per stage for the whole target set. */
int p7x_search_longtargets(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, int device,
                           const uint8_t *dsq, const int64_t *offsets, const int64_t *lengths, size_t n,
                           const char *const *names, const char *const *accs, const char *const *descs, p7x_tophits **out);
/* The device copy that cfg.lt_resident_key keeps between searches is dropped: on <device> (-1: every device), if it carries
 * <key> (0: whatever it carries).  The reference has no counterpart (its targets live in host memory); here the owner of a
 * target image calls this when the image goes (pyhmmer_amd does, from a finalizer of the packed block), so that a genome
 * does not stay pinned in HBM for the life of the process.  A search still scanning the copy keeps it until it is done. */
int p7x_longtargets_release_resident(int device, uint64_t key);
/* SSV window seeds of one strand of one target block as p7_SSVFilter_longtarget emits them (position of the diagonal's
 * first residue, model node of its last cell, diagonal length), for tests against the oracle: the device scan followed
 * by upstream's sequential bookkeeping.  seeds: caller array of cap x 3 int64; returns the number found (may exceed cap)
 * or a negative status. */
int64_t p7x_ssv_longtarget_seeds(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, int device, const uint8_t *dsq, int64_t L,
                                 int complement, int64_t *seeds, size_t cap);
/* CPU test seam: the host tail of the long-target pipeline for seeds found elsewhere (the oracle's scan); no device. */
int p7x_longtarget_from_seeds(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om,
                              const uint8_t *dsq, const int64_t *offsets, const int64_t *lengths, size_t n,
                              const char *const *names, const char *const *accs, const char *const *descs,
                              const int64_t *seed_target, const int64_t *seed_block, const int32_t *seed_strand,
                              const int64_t *seeds, size_t nseeds, p7x_tophits **out);

/* The Forward parser score (nats) of residues dsq[1..L] in the summation order of the reference's striped vectors
 * (impl_sse/fwdback.c forward_engine, do_full = FALSE; multihit, length model of L): what the host stage uses to decide
 * targets whose Forward P-value falls within the guard band around F3 (cfg.f3_guard).  Host code; no device. */
int p7x_forward_parser_exact(const p7x_oprofile *om, const uint8_t *dsq, int32_t L, float *sc);

/* hmmscan orientation (Pipeline.scan_seq / _scan_loop, plan7.pyx:6534-6677; hmmer/_hmmscan.py): search every model
 * against the block of query sequences with cfg.mode = P7X_SCAN_MODELS (one device pass per model, nothing pruned), then
 * transpose: out[s] (caller-provided array of nseqs pointers) receives the hit list of query sequence s, its hits
 * named after the models, reportability tested with the running Z = models seen so far, E-values with Z = nmodels,
 * counters per sequence (nmodels, nnodes, n_past_* = models whose filters the sequence passed). */
int p7x_scan_collect(p7x_tophits *const *per_model, size_t nmodels, const p7x_pipeline_cfg *cfg, size_t nseqs,
                     const char *const *seq_names, const char *const *seq_accs, const char *const *seq_descs,
                     const int32_t *seq_lengths, p7x_tophits **out);
/* The same, incrementally: the per-model results are folded in as the device batches of a scan come back (in model order;
 * the running Z of p7_pli_NewModel is the model's number) and can be destroyed right after, so a scan over a library of
 * any size holds one batch of per-model results at a time, as the reference's loop over the pressed file holds one
 * profile (plan7.pyx:6680-6737).  p7x_scan_accum_finish consumes the accumulator. */
typedef struct p7x_scan_accum p7x_scan_accum;
int  p7x_scan_accum_create(const p7x_pipeline_cfg *cfg, size_t nseqs, const char *const *seq_names, const char *const *seq_accs,
                           const char *const *seq_descs, const int32_t *seq_lengths, p7x_scan_accum **out);
int  p7x_scan_accum_add(p7x_scan_accum *acc, p7x_tophits *const *per_model, size_t nmodels);
/* the same for results that come back in any order (batches of a length-sorted profile database): model_index[i] is the
 * number of per_model[i]'s profile in the database (0-based, each exactly once over the scan); callable from several threads.
 * The running Z of a model is its number + 1 either way, and p7x_scan_accum_finish restores the database's order. */
int  p7x_scan_accum_add_indexed(p7x_scan_accum *acc, p7x_tophits *const *per_model, const int64_t *model_index, size_t nmodels);
/* Test seam: the last filter every target passed (0 none, 1 MSV, 2 bias, 3 Viterbi, 4 Forward), as the device path records it
 * in a scan's per-model results and p7x_scan_accum_* turn into per-sequence accounting; lets the CPU tests attach the
 * oracle's stages to results of p7x_postprocess_targets. */
int  p7x_debug_tophits_set_stages(p7x_tophits *th, const uint8_t *stage, size_t n);
int  p7x_scan_accum_finish(p7x_scan_accum *acc, p7x_tophits **out);
void p7x_scan_accum_destroy(p7x_scan_accum *acc);

/* Host half of p7_Pipeline for targets that already passed the Forward filter: Backward-derived domain
 * definition (p7_domaindef_ByPosteriorHeuristics, p7_domaindef.pxd:69-72), per-sequence / per-domain scores,
 * reporting thresholds, sort.  p7x_search_block calls this internally with the device parsers' output; it is
 * exported so the host logic can be tested without a GPU from any Forward/Backward parser rows.
 * surv[i] = target index; fwdsc[i] nats; fwd_xmx / bck_xmx + xmx_off[i] = (L+1) x [E,N,J,B,C,SCALE] rows
 * (impl_sse/p7_omx.pxd:16-38).  offsets[t] >= 1.  stage_counts = n_past_{msv,bias,vit,fwd} or NULL. */
int p7x_postprocess_targets(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const uint8_t *dsq,
                            const int64_t *offsets, const int32_t *lengths, size_t n, const int32_t *surv,
                            size_t nsurv, const float *fwdsc, const float *fwd_xmx, const float *bck_xmx,
                            const int64_t *xmx_off, const uint64_t *stage_counts, const char *const *names,
                            const char *const *accs, const char *const *descs, p7x_tophits **out);

void     p7x_tophits_destroy(p7x_tophits *th);
/* n results at once (a scan batch is thousands of per-model lists that only pass through the caller); the slots are zeroed */
void     p7x_tophits_destroy_many(p7x_tophits **th, size_t n);
p7x_tophits *p7x_tophits_clone(const p7x_tophits *th);   /* TopHits.copy, plan7.pyx:9150-9170 */
int64_t  p7x_tophits_nhits(const p7x_tophits *th);
int      p7x_tophits_get_counters(const p7x_tophits *th, p7x_counters *c);
int      p7x_tophits_get_cfg(const p7x_tophits *th, p7x_pipeline_cfg *cfg);   /* Z/domZ as finally set */
int      p7x_tophits_get_hit(const p7x_tophits *th, int64_t i, p7x_hit *hit);
int      p7x_tophits_get_domain(const p7x_tophits *th, int64_t i, int32_t d, p7x_domain *dom);
/* p7_tophits_Merge + p7_pipeline_Merge + re-threshold (plan7.pyx:9172-9276): merges src into dst. */
int      p7x_tophits_merge(p7x_tophits *dst, const p7x_tophits *src);
/* The parts of one long-target search (cfg.lt_nparts > 1, one p7x_search_longtargets call per part, typically one per
 * device) -> the hit list of the whole search: p7_tophits_ComputeNhmmerEvalues with the residues of all parts,
 * p7_tophits_RemoveDuplicates across them, sort, threshold (plan7.pyx:7390-7412).  The parts are consumed. */
int      p7x_tophits_merge_longtargets(p7x_tophits **parts, size_t nparts, p7x_tophits **out);
/* Many queries x many shards in one call, threaded over the queries: blobs[q * nparts + r] / sizes[...] = the
 * p7x_tophits_serialize image of query q on shard r (NULL / 0: that shard had nothing to say, e.g. an empty shard that
 * was never searched).  outs[q] = what TopHits.merge gives for the same lists in shard order.  threads <= 0: every usable CPU. */
int      p7x_tophits_merge_many(const void *const *blobs, const size_t *sizes, size_t nq, size_t nparts, int threads, p7x_tophits **outs);
/* Flat byte image of a hit list (the reference pickles TopHits through p7_hit_Serialize, plan7.pyx:8394-8572);
 * used to move per-device / per-process results to the merging host.  serialize returns the size needed. */
int64_t  p7x_tophits_serialize(const p7x_tophits *th, void *buf, size_t cap);
p7x_tophits *p7x_tophits_deserialize(const void *buf, size_t n);
int      p7x_tophits_sort_by_key(p7x_tophits *th);       /* p7_tophits_SortBySortkey, plan7.pyx:8820-8824 */
int      p7x_tophits_threshold(p7x_tophits *th);         /* p7_tophits_Threshold, plan7.pyx:8804-8818 */
int      p7x_tophits_sort_by_seqidx(p7x_tophits *th);    /* p7_tophits_SortBySeqidxAndAlipos, TopHits.sort(by="seqidx") plan7.pyx:9120-9148 */
int      p7x_tophits_is_sorted(const p7x_tophits *th, int by_seqidx);   /* TopHits.is_sorted, plan7.pyx:9078-9118; 1 / 0 */
/* Hit.reported / included / dropped / duplicate setters (plan7.pyx:2125-2235): replace the P7X_IS_* flag word of hit i. */
int      p7x_tophits_set_hit_flags(p7x_tophits *th, int64_t i, uint32_t flags);
/* Hit.name / accession / description setters (plan7.pyx:1960-2050): which = 1 name (not NULL), 2 accession, 4 description
 * (NULL clears); pointers handed out earlier by p7x_tophits_get_hit for this hit are invalidated. */
int      p7x_tophits_set_hit_text(p7x_tophits *th, int64_t i, int which, const char *value);
/* per-stage device timings of the search that produced th, milliseconds (HIP events):
 * [0] msv + P-value pass [1] bias filter [2] viterbi [3] forward [4] forward rows for survivors [5] backward, then
 * overwritten by host domain definition wall time (including [8]) [6] whole call wall time [7] the MSV kernel alone
 * [8] device rescoring of domain envelopes (wall, with transfers) [9] host: multi-domain regions (overlaps [8])
 * [10] wall time of stage 1 (p7x_search_block_begin) [11] of stage 2 (_finish); [6] = [10] + [11].  n <= 12. */
int      p7x_tophits_get_timings(const p7x_tophits *th, double *ms, int n);
/* how often the two guards acted in the search that produced <th>: targets the F3 guard took back out of the device's
 * survivor list, device envelopes the optimal-accuracy near-tie guard repeated on the host (0 after a merge /
 * deserialisation), and the latter by the kind of choice that was close: the predecessor of a match / insert / delete
 * cell, C<-E, J<-E, the end cell, B<-N/J, a printed posterior digit (an envelope can count under several) */
int      p7x_tophits_get_guard_counts(const p7x_tophits *th, int64_t *f3_dropped, int64_t *oa_redone, int64_t oa_why[8]);
/* ABI 8: multi-domain regions whose traceback ensemble the device sampled, regions its near-threshold guard
 * (p7x_pipeline_cfg.ens_guard) flagged and the host stage sampled again in upstream's summation order, and targets whose
 * region scan (rt1 / rt2 / rt3 against posteriors from the device parsers' rows) had a comparison within 2e-5 of its
 * threshold and was repeated by the host stage on parser rows in upstream's order (active with oa_guard > 0) */
int      p7x_tophits_get_ensemble_counts(const p7x_tophits *th, int64_t *sampled_on_device, int64_t *redone_by_host, int64_t *region_scans_redone);

const char *p7x_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* P7X_H */
