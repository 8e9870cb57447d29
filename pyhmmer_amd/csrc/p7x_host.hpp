// p7x_host.hpp -- host-side post-processing of Forward survivors: domain definition and hit lists.
#pragma once
#include "p7x_internal.hpp"
#include <functional>
#include <memory>
#include <string>
#include <vector>

struct p7x_seqdb;

namespace p7x {

// One domain as p7_domaindef + p7_Pipeline leave it (P7_DOMAIN, p7_domain.pxd:10-26) with its
// alignment display (P7_ALIDISPLAY, p7_alidisplay.pxd:26-53).
struct Domain {
  int64_t ienv = 0, jenv = 0, iali = 0, jali = 0;
  float envsc = 0, domcorrection = 0, dombias = 0, oasc = 0, bitscore = 0;
  double lnP = 0;
  bool is_reported = false, is_included = false;
  // alignment display
  int N = 0, hmmfrom = 0, hmmto = 0, M = 0;
  int64_t sqfrom = 0, sqto = 0, L = 0;
  std::string model, mline, aseq, ppline, rfline, mmline, csline;
  int deferred = -1;          // >= 0: placeholder, to be filled from envelope request <deferred> (device rescoring)
  int multi_slot = -1;        // deferred == -2: placeholder of a multi-domain region, domains in DomainDefResult::multi
  int deferred2 = -1;         // >= 0 (inside DomainDefResult::multi): clustered envelope rescored by the device in a second round
};

struct DomainDefResult {      // the P7_DOMAINDEF fields p7_Pipeline reads (p7_domaindef.pxd:23-59)
  std::vector<Domain> dcl;
  std::vector<float> n2sc;    // [L+1]
  std::vector<std::vector<Domain>> multi;   // domains of deferred multi-domain regions (see Domain::multi_slot)
  float nexpected = 0;
  int nregions = 0, nclustered = 0, noverlaps = 0, nenvelopes = 0;
  int nneartie = 0;           // device envelopes repeated by the host twin (optimal-accuracy near-tie guard)
  int neartie_why[8]{};       // ... by the kind of choice that was close (EnvArgs::out_status bits 8-15)
  int nregion_redone = 0;     // 1: the device's region scan was inside its guard band, the regions come from rows in upstream's order
  int nens_device = 0;        // multi-domain regions whose ensemble the device sampled
  int nens_redone = 0;        // ... and those it flagged as too close to call (near-threshold guard): sampled again here, in upstream's order
};

// Rescoring of single-domain envelopes can be handed to the device (p7x_envelope.hip): the first half of domain
// definition then only queues a request per envelope, the second half turns the device's answer into a Domain.
struct EnvelopeRequest { int item; int32_t i, j; };          // item: index into the survivor list; envelope i..j (1-based)
struct EnvelopeResult {
  float envsc = 0, oasc = 0; int status = 0;
  float orig = 0;                                            // long-target envelopes: Forward with the unmodified odds
  float null2[32];                                           // odds of the canonical residues (before esl_abc_FAvgScVec)
  int ntrace = 0; const uint32_t *ta = nullptr; const int32_t *ti = nullptr; const float *tp = nullptr;   // traceback order
};
// One job = the envelopes of one query profile (targets[item] = caller index of the survivor a request belongs to).
struct EnvelopeJob { const p7x_oprofile *om = nullptr; const std::vector<EnvelopeRequest> *req = nullptr; const std::vector<int32_t> *targets = nullptr;
                     // long-target envelopes: match odds of every request in the device's table layout ([nrows][Mpad] each,
                     // lt_stride floats apart); the length model is then the envelope's own length
                     const float *lt_tables = nullptr; size_t lt_stride = 0; };
struct EnvelopeScorer {
  virtual ~EnvelopeScorer() = default;
  // begin() enqueues the jobs of a batch of queries (one launch per model-length class, all profiles of a class in
  // it) and returns; wait() blocks until they are done and fills res[j][r] for every request r of job j.  Result
  // buffers stay valid until the next begin().
  virtual int begin(const std::vector<EnvelopeJob> &jobs) = 0;
  virtual int wait(std::vector<std::vector<EnvelopeResult>> &res) = 0;
};

// The stochastic traceback ensembles of multi-domain regions can be handed to the device as well (p7x_ensemble.hip): a
// request is region i..j of survivor <item> (EnvelopeRequest), the answer the sampled domains' end points and the
// per-residue sums of their null2 odds; clustering and everything after it stays with the host stage.
struct EnsembleRaw { std::vector<int32_t> dom; std::vector<float> n2; };      // test seam: a host-sampled ensemble in the device's format
struct EnsembleResult {
  int status = -1;                      // 0: done; anything else: the host resolves the region itself
  int ndom = 0;                         // sampled domains, all samples together
  const int32_t *dom = nullptr;         // [ndom][5] sample, sqfrom, sqto (1-based inside the region), hmmfrom, hmmto; the domains
                                        // of a sample in traceback order (last domain first)
  const float *n2 = nullptr;            // [Lr + 1] n2[pos]: sum over the samples of the odds ratio of region residue pos
  EnsembleRaw *raw_out = nullptr;       // test seam (status != 0): the host's own ensemble of the region is written here, domains first to last
};
struct EnsembleRunner {
  virtual ~EnsembleRunner() = default;
  // seed_state: the generator's state after esl_randomness_Init (every region starts there: do_reseeding)
  virtual int begin(const std::vector<EnvelopeJob> &jobs, uint32_t seed_state, int nsamples) = 0;
  virtual int wait(std::vector<std::vector<EnsembleResult>> &res) = 0;
};
uint32_t fast_rng_state(uint32_t seed);          // esl_randomness_Init for the LCG
// p7_ForwardParser / p7_BackwardParser special-state rows ((L+1) x [E,N,J,B,C,SCALE], multihit, length model of L) in upstream's
// summation order (the full-matrix engine; the target's DP matrices are scratch): dsq[1..L]
int parser_rows_upstream(const Profile &p, const uint8_t *dsq, int L, std::vector<float> &fx, std::vector<float> &bx);

// p7_domaindef_ByPosteriorHeuristics (p7_domaindef.pxd:69-72).  dsq is 1-indexed (dsq[1..L]);
// fwd_xmx / bck_xmx are the parsers' special-state rows, (L+1) x [E,N,J,B,C,SCALE].
struct Region { int i, j; bool multi; };
// Region scan done on the device (p7x_pipeline.hip regions_kernel): per survivor n[i] regions (or -1: range error),
// regs[(first(i) + r)*3 ..] = first residue, last residue, is_multidomain_region, with first(i) = start[i] (packed), or
// i*cap when <start> is null; nexpected[i] = expected number of domains.
struct DeviceRegions { const int32_t *n = nullptr; const int32_t *regs = nullptr; const float *nexpected = nullptr; int cap = 0; const int32_t *start = nullptr; };
struct MultiRegionState { bool started = false; uint32_t rng_seed = 42, rng_x = 0; };   // RNG carried between the regions of one target
int domaindef_regions(const Profile &p, int L, const float *fwd_xmx, const float *bck_xmx, DomainDefResult &dd, std::vector<Region> &regs);
// <defer2>: the clustered envelopes are queued there (tagged <item>) for a second round of device rescoring instead of being
// rescored on the host; their placeholders carry Domain::deferred2 and are completed by domaindef_finish_deferred().
int domaindef_multi_region(const Profile &p, const uint8_t *dsq, int L, int i, int j, uint32_t seed, bool do_reseeding,
                           MultiRegionState &state, DomainDefResult &dd, std::vector<Domain> &out,
                           std::vector<EnvelopeRequest> *defer2 = nullptr, int item = 0, const EnsembleResult *ens = nullptr);
// <ens>: the device's ensembles of this target's multi-domain regions, in order (nullptr, or an entry with status != 0: the
// host samples the region itself)
int domaindef_finish_multi(const Profile &p, const uint8_t *dsq, int L, uint32_t seed, bool do_reseeding, DomainDefResult &dd,
                           std::vector<EnvelopeRequest> *defer2 = nullptr, int item = 0, const EnsembleResult *const *ens = nullptr);

// With <defer> the single-domain regions are queued there (tagged <item>) instead of being rescored on the host;
// domaindef_finish_deferred() completes them from the device results (res[d.deferred] for placeholder d).
int domaindef_by_posterior_heuristics(const Profile &p, const uint8_t *dsq, int L, const float *fwd_xmx,
                                      const float *bck_xmx, uint32_t seed, bool do_reseeding, DomainDefResult &out,
                                      std::vector<EnvelopeRequest> *defer = nullptr, int item = 0);
// The same from a region list that was found elsewhere (device scan); n2sc etc. are initialised here.
int domaindef_from_regions(const Profile &p, const uint8_t *dsq, int L, float nexpected, const Region *regs, int nregs,
                           uint32_t seed, bool do_reseeding, DomainDefResult &out, std::vector<EnvelopeRequest> *defer, int item);
int domaindef_finish_deferred(const Profile &p, const uint8_t *dsq, int L, const std::vector<EnvelopeResult> &res,
                              const std::vector<int> &req_index, DomainDefResult &dd,
                              const std::vector<EnvelopeResult> *res2 = nullptr, const std::vector<int> *req_index2 = nullptr);

struct Hit {                  // P7_HIT, p7_hit.pxd:27-58
  std::string name, acc, desc;
  bool has_acc = false, has_desc = false;
  int64_t seqidx = 0;
  int32_t window_length = 0;  // long targets: the model's max_length the E-value refers to
  double sortkey = 0;
  float score = 0, pre_score = 0, sum_score = 0;
  double lnP = 0, pre_lnP = 0, sum_lnP = 0;
  float nexpected = 0;
  int nregions = 0, nclustered = 0, noverlaps = 0, nenvelopes = 0, ndom = 0;
  uint32_t flags = 0;
  int nreported = 0, nincluded = 0, best_domain = 0;
  std::vector<Domain> dcl;
};

// The searched block as the host sees it: residues of target t are dsq[off[t] .. off[t]+len[t]-1], off[t] >= 1.
struct HostTargets { int64_t n = 0, nres = 0; const int32_t *len = nullptr; const int64_t *off = nullptr; const uint8_t *dsq = nullptr; };

// counts[4] = n_past_{msv,bias,vit,fwd}; targets[] = indices of the Forward survivors; fwdsc / xmx blocks per survivor.
int host_finish_search(const p7x_pipeline_cfg &cfg, const p7x_oprofile *om, const HostTargets &tg,
                       const char *const *names, const char *const *accs, const char *const *descs,
                       const std::vector<int32_t> &targets, const float *fwdsc,
                       const float *fwd_xmx, const float *bck_xmx, const int64_t *xmx_off,
                       const uint64_t *counts, const double *ms, p7x_tophits **out, EnvelopeScorer *scorer = nullptr,
                       const DeviceRegions *regions = nullptr);
// One query of a batch as the host stage receives it from the device stage.
struct FinishItem {
  const p7x_oprofile *om = nullptr;
  const std::vector<int32_t> *targets = nullptr;      // caller indices of the Forward survivors
  const float *fwdsc = nullptr;                       // per survivor
  const std::vector<char> *near = nullptr;            // which survivors the first stage saw inside the F3 guard band (empty: none;
                                                      // nullptr: not known, the host stage checks every survivor itself)
  const float *fwd_xmx = nullptr, *bck_xmx = nullptr; const int64_t *xmx_off = nullptr;   // parser rows (when regions == nullptr)
  uint64_t counts[4] = { 0, 0, 0, 0 };                // n_past_{msv,bias,vit,fwd}
  const double *ms = nullptr; int nms = 8;            // 0-7 as TopHits.timings_ms; 8-10 (nms >= 11): queries of the batch, lanes and nodes of its largest MSV launch
  const DeviceRegions *regions = nullptr;             // region lists found on the device
  bool device_envelopes = false;                      // single-domain envelopes go to the scorer
};
int host_finish_batch(const p7x_pipeline_cfg &cfg, const std::vector<FinishItem> &items, const HostTargets &tg,
                      const char *const *names, const char *const *accs, const char *const *descs,
                      p7x_tophits **outs, EnvelopeScorer *scorer = nullptr, EnvelopeScorer *scorer2 = nullptr,
                      EnsembleRunner *ensembles = nullptr);
void tophits_set_total_ms(p7x_tophits *th, double stage1_ms, double stage2_ms);
void tophits_sort_by_key(p7x_tophits &th);
void tophits_threshold(p7x_tophits &th);
bool tophits_target_reportable(const p7x_pipeline_cfg &c, float score, double lnP);
int tophits_usable_cpus();
// body(i) for i in [0, n) on the persistent host workers (dynamic schedule; nthreads <= 0: every usable CPU)
void host_parallel_for(int n, int nthreads, const std::function<void(int)> &body);
void tophits_set_stages(p7x_tophits *th, std::vector<uint8_t> &&stage);
float kahan_fsum(const float *v, int n);

// ---- long targets (p7x_longtarget.inc.hpp <-> p7x_longtarget.hip)
struct LongTargetRow { int64_t pos; int k, sc; };            // a row of a strand block that reached the SSV threshold, and the cell upstream picks
struct LongTargetSeed { int64_t target, block_start; int strand; int64_t n; int k; int64_t length; };   // an SSV window seed of one block
// (target, block, strand) units of a long-target search in the order of the reference's loop, and the part that owns one
struct LongTargetUnits {
  int64_t W = 0, C = 0; int strands = 0; uint64_t total = 0; int part = 0, nparts = 1;
  void count(const p7x_pipeline_cfg &cfg, int max_length, const int64_t *lengths, size_t n);
  bool mine(uint64_t u) const { return nparts <= 1 || (int) ((u * (uint64_t) nparts) / (total ? total : 1)) == part; }
};
void longtarget_finalize(p7x_tophits *th, int evalue_window, double res_count);
double longtarget_res_count(const p7x_pipeline_cfg &cfg, uint64_t nres);
int longtarget_setup(const p7x_pipeline_cfg &cfg, const Profile &p, int *max_length, int *sc_thresh, int *xB);
const uint8_t *longtarget_complement(int abc_type);
void longtarget_seeds_from_rows(const Profile &p, const uint8_t *block_dsq, int64_t L, const LongTargetRow *rows, size_t nrows,
                                int sc_thresh, int xB, std::vector<int64_t> &seeds3);
// The filter scores of a target's windows from one device batch (p7x_longtarget.hip): a window = <length> residues of
// <strand> starting at original position <start> (strand 1: running towards lower positions, complemented).
struct LongTargetWindowRef { int64_t start, length; int strand; };
struct LongTargetWindowScore { float usc, bias_filtersc, vfsc; int have_vit; };
struct LongTargetWindowRegions { int n = -2; float nexpected = 0.0f; std::vector<Region> regs; };
// one long-target envelope for the device: window (index into the last regions() call), envelope i..j in the window,
// and its adjusted match odds rf[x * (M + 1) + k] (x < Kp), as reparameterize() makes them
struct LongTargetEnvRequest { int window = 0; int i = 0, j = 0; const float *rf = nullptr; };
struct LongTargetEnvResult { float envsc = 0, oasc = 0, orig = 0; int status = 0; std::vector<uint32_t> ta; std::vector<int32_t> ti; std::vector<float> tp; };
struct LongTargetWindowScorer {
  virtual ~LongTargetWindowScorer() = default;
  // seq1: the target, 1-based; sc[w]: MSV score (nats), bias filter score, and for windows that pass both P <= F1 tests
  // but not P <= F2 the standard Viterbi filter score
  virtual int score(const uint8_t *seq1, int64_t L, const uint8_t *comp, const LongTargetWindowRef *w, size_t nw, double F1, bool do_bias,
                    LongTargetWindowScore *sc) = 0;
  // p7_ViterbiFilter_longtarget over windows which[0..n) of the last score() call, window i with row-score threshold
  // thresh[i]: rec receives (i, row, node) for every seeding cell, sorted by (i, row, node)
  virtual int viterbi(const int *which, const int *thresh, size_t n, std::vector<int> &rec) = 0;
  // p7_ForwardParser scores (nats) of another set of windows of the same target (the Viterbi windows)
  virtual int forward(const uint8_t *seq1, const uint8_t *comp, const LongTargetWindowRef *w, size_t nw, float *fwdsc) = 0;
  // Forward + Backward parsers, posterior decoding of the special states and the region scan (the first step of
  // p7_domaindef_ByPosteriorHeuristics) of the windows that passed the Forward filter, in one device batch:
  // out[w].n regions (-1: p7_DomainDecoding range error, the window is dropped; -2: not available, the host scans)
  virtual int regions(const uint8_t *seq1, const uint8_t *comp, const LongTargetWindowRef *w, size_t nw, std::vector<LongTargetWindowRegions> &out) = 0;
  // rescore_isolated_domain(long_target = TRUE) of envelopes of those windows, one device batch (the envelope kernel's
  // LT instantiation): adjusted and unmodified Forward scores, optimal-accuracy score and trace
  virtual int envelopes(const LongTargetEnvRequest *req, size_t n, std::vector<LongTargetEnvResult> &out) = 0;
};
int longtarget_run_host(const p7x_pipeline_cfg &cfg, const p7x_oprofile *om, const uint8_t *dsq, const int64_t *offsets, const int64_t *lengths,
                        size_t n, const char *const *names, const char *const *accs, const char *const *descs,
                        const std::vector<LongTargetSeed> &seeds, p7x_tophits **out, LongTargetWindowScorer *filters = nullptr);
int host_forward_parser_exact(const Profile &p, const uint8_t *dsq1, int L, float *sc);     // dsq1[1..L]
float host_filter_null_score(const Profile &p, const uint8_t *dsq1, int L, bool do_bias);
void host_prof_dump();

} // namespace p7x

struct p7x_tophits {
  std::vector<p7x::Hit> hits;         // storage order
  std::vector<int> order;             // presentation order (indices into hits)
  p7x_pipeline_cfg cfg{};             // copy of the pipeline configuration, Z/domZ as finally set
  p7x_counters ctr{};
  std::string qname, qacc, qdesc;     // the query (model) the alignment displays refer to
  bool q_has_acc = false, q_has_desc = false;
  int M = 0;
  static constexpr int kMs = 20;
  double ms[kMs]{};           // see TopHits.timings_ms (plan7.py) for the slots
  bool sorted_by_key = false;
  bool scan_collected = false;        // built by p7x_scan_collect(): one query sequence, hits are models
  std::vector<uint8_t> stage;         // scan mode, per-model result: last filter passed by each target (not serialised)
  std::vector<int32_t> guard_dropped; // targets the F3 guard took out of the device's survivor list (not serialised)
  bool lt_unfinished = false;         // one part of a long-target search (cfg.lt_nparts > 1): E-values, duplicates, thresholds still to do
  int lt_evalue_window = 0;           // ... and the window length its E-values refer to
  int64_t oa_redone = 0;              // device envelopes the near-tie guard sent to the host twin (not serialised)
  int64_t oa_why[8]{};                // ... by kind of choice: M, I, D cell, C<-E, J<-E, end cell, B<-N/J, posterior digit
  int64_t region_redone = 0;                // targets whose region scan the host stage repeated on rows in upstream's order (the device scan's guard)
  int64_t ens_device = 0, ens_redone = 0;   // regions sampled on the device / flagged by its near-threshold guard and sampled again by the host stage (not serialised)
  int64_t nreported = 0, nincluded = 0;
};
