// p7x_tophits.cpp -- from Forward survivors to a thresholded, sorted hit list (host side).
//
// Restates the tail of upstream p7_pipeline.c:p7_Pipeline (after the Backward parser; reference
// p7_pipeline.pxd:130) and upstream p7_tophits.c: p7_tophits_SortBySortkey, p7_tophits_Threshold,
// p7_tophits_Merge + p7_pipeline_Merge (reference p7_tophits.pxd:20-72; plan7.pyx:8804-8830, 9172-9276).
#include "p7x_host.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace p7x {

static bool target_reportable(const p7x_pipeline_cfg &c, double Z, float score, double lnP)
{
  if (c.by_E) return std::exp(lnP) * (c.long_targets ? 1.0 : Z) <= c.E;     // long targets: the database size is in the P-value
  return score >= c.T;
}
static bool target_includable(const p7x_pipeline_cfg &c, double Z, float score, double lnP)
{
  if (c.inc_by_E) return std::exp(lnP) * (c.long_targets ? 1.0 : Z) <= c.incE;
  return score >= c.incT;
}
static bool domain_reportable(const p7x_pipeline_cfg &c, double domZ, float score, double lnP)
{
  if (c.dom_by_E) return std::exp(lnP) * (c.long_targets ? 1.0 : domZ) <= c.domE;
  return score >= c.domT;
}
static bool domain_includable(const p7x_pipeline_cfg &c, double domZ, float score, double lnP)
{
  if (c.incdom_by_E) return std::exp(lnP) * (c.long_targets ? 1.0 : domZ) <= c.incdomE;
  return score >= c.incdomT;
}

// the same thresholds, taken over from a configuration they were already applied to (scan mode: one model per result)
static void apply_bit_cutoffs_from(p7x_pipeline_cfg &c, const p7x_pipeline_cfg &model_cfg)
{
  if (!c.use_bit_cutoffs) return;
  c.T = model_cfg.T; c.incT = model_cfg.incT; c.domT = model_cfg.domT; c.incdomT = model_cfg.incdomT;
  c.by_E = c.dom_by_E = c.inc_by_E = c.incdom_by_E = 0;
}

// model-specific thresholds (p7_pli_NewModelThresholds): T/domT/incT/incdomT from GA/TC/NC
static void apply_bit_cutoffs(p7x_pipeline_cfg &c, const Profile &p)
{
  if (!c.use_bit_cutoffs) return;
  const int i = c.use_bit_cutoffs == P7X_BITCUT_GA ? P7X_GA1 : (c.use_bit_cutoffs == P7X_BITCUT_TC ? P7X_TC1 : P7X_NC1);
  c.T = c.incT = p.cutoff[i];
  c.domT = c.incdomT = p.cutoff[i + 1];
  c.by_E = c.dom_by_E = c.inc_by_E = c.incdom_by_E = 0;
}

struct Pending { bool have = false; Hit hit; };

// CPUs this process may actually use: affinity mask, then the cgroup CPU quota (containers on many-core hosts)
static int usable_cpus()
{
  int n = (int) std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (n <= 0 || c < n)) n = c; }
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota|max> <period>"
    char q[64]; long period = 0;
    if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
      const long quota = std::atol(q);
      if (quota > 0) { const int c = (int) ((quota + period - 1) / period); if (c > 0 && c < n) n = c; }
    }
    std::fclose(f);
  } else {
    long quota = -1, period = -1;                                        // cgroup v1
    if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%ld", &quota) != 1) quota = -1; std::fclose(g); }
    if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%ld", &period) != 1) period = -1; std::fclose(g); }
    if (quota > 0 && period > 0) { const int c = (int) ((quota + period - 1) / period); if (c > 0 && c < n) n = c; }
  }
  return n > 0 ? n : 1;
}

// Persistent workers for the host phases of a search (spawning 16 threads three times per query costs ~1 ms).
// run(n, nthreads, body) calls body(i) for every i in [0, n), dynamically scheduled; the caller participates and at most
// nthreads - 1 workers join it.  Several callers may be inside run() at once (the host stages of the batches in flight):
// their regions share the workers, oldest region first, so a region that is down to one long item (a stochastic
// traceback ensemble is a serial walk) does not hold up the short phases of the next batch.
class HostPool {
public:
  static HostPool &get() { static HostPool *pool = new HostPool(); return *pool; }      // never torn down (no joins at process exit)
  void run(int n, int nthreads, const std::function<void(int)> &body)
  {
    if (n <= 0) return;
    if (nthreads > n) nthreads = n;
    flogsum_init();
    if (nthreads <= 1) { for (int i = 0; i < n; ++i) body(i); return; }
    Region rg;
    rg.body = &body; rg.n = n; rg.max_workers = nthreads - 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      while ((int) workers_.size() < nthreads - 1) workers_.emplace_back([this] { loop(); });
      regions_.push_back(&rg);
    }
    cv_.notify_all();
    for (;;) { const int i = rg.next.fetch_add(1); if (i >= n) break; body(i); }
    std::unique_lock<std::mutex> lk(mu_);
    for (size_t z = 0; z < regions_.size(); ++z) if (regions_[z] == &rg) { regions_.erase(regions_.begin() + (long) z); break; }   // no new workers
    rg.done_cv.wait(lk, [&] { return rg.active == 0; });
  }
private:
  struct Region {
    const std::function<void(int)> *body = nullptr;
    int n = 0, max_workers = 0, active = 0;         // active: workers inside this region (under mu_)
    std::atomic<int> next{0};
    std::condition_variable done_cv;
  };
  HostPool() = default;
  void loop()
  {
    flogsum_init();
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      Region *rg = nullptr;
      for (Region *r : regions_) if (r->active < r->max_workers && r->next.load(std::memory_order_relaxed) < r->n) { rg = r; break; }
      if (!rg) { cv_.wait(lk); continue; }
      rg->active++;
      lk.unlock();
      for (;;) { const int i = rg->next.fetch_add(1); if (i >= rg->n) break; (*rg->body)(i); }
      lk.lock();
      if (--rg->active == 0) rg->done_cv.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<std::thread> workers_;
  std::vector<Region *> regions_;
};

float kahan_fsum(const float *v, int n)
{ // Easel esl_vec_FSum: Kahan compensated summation
  float sum = 0.0f, c = 0.0f;
  for (int i = 0; i < n; ++i) { volatile float y = v[i] - c; volatile float t = sum + y; c = (t - sum) - y; sum = t; }
  return sum;
}


// Everything p7_Pipeline does after p7_domaindef_ByPosteriorHeuristics, for one target.
static void finish_one(const p7x_pipeline_cfg &cfg, const Profile &p, int L, float fwdsc, double Z_running,
                       DomainDefResult &dd, Pending &out)
{
  if (dd.nregions == 0 || dd.nenvelopes == 0 || dd.dcl.empty()) return;
  const float nullsc = null1_score(L);
  const float omega = 1.0f / 256.0f;
  float seqbias;
  if (cfg.do_null2) {
    seqbias = kahan_fsum(dd.n2sc.data(), L + 1);                      // esl_vec_FSum (compensated)
    seqbias = flogsum(0.0f, std::log((double) omega) + seqbias);
  } else seqbias = 0.0f;
  float pre_score = (fwdsc - nullsc) / kLog2;
  float seq_score = (fwdsc - (nullsc + seqbias)) / kLog2;
  // reconstruction score: sum of the domains that stay significant after their null2 correction
  float sum_score = 0.0f;
  int Ld = 0;
  seqbias = 0.0f;
  if (cfg.do_null2) {
    for (const Domain &d : dd.dcl)
      if (d.envsc - d.domcorrection > 0.0) { sum_score += d.envsc; Ld += (int) (d.jenv - d.ienv + 1); seqbias += d.domcorrection; }
    seqbias = flogsum(0.0f, std::log((double) omega) + seqbias);
  } else {
    for (const Domain &d : dd.dcl)
      if (d.envsc > 0.0) { sum_score += d.envsc; Ld += (int) (d.jenv - d.ienv + 1); }
    seqbias = 0.0f;
  }
  sum_score += (L - Ld) * std::log((double) ((float) L / (float) (L + 3)));
  const float pre2_score = (sum_score - nullsc) / kLog2;
  sum_score = (sum_score - (nullsc + seqbias)) / kLog2;
  if (Ld > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2_score; }

  const double lnP = exp_logsurv(seq_score, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]);
  // scan mode: the running Z of a query sequence is the number of models seen so far, known only to p7x_scan_collect()
  if (cfg.mode != P7X_SCAN_MODELS && !target_reportable(cfg, Z_running, seq_score, lnP)) return;

  Hit &h = out.hit;
  out.have = true;
  h.ndom = (int) dd.dcl.size();
  h.nexpected = dd.nexpected; h.nregions = dd.nregions; h.nclustered = dd.nclustered;
  h.noverlaps = dd.noverlaps; h.nenvelopes = dd.nenvelopes;
  h.pre_score = pre_score; h.pre_lnP = exp_logsurv(pre_score, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]);
  h.score = seq_score; h.lnP = lnP;
  h.sortkey = cfg.inc_by_E ? -lnP : seq_score;
  h.sum_score = sum_score; h.sum_lnP = exp_logsurv(sum_score, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]);
  h.dcl = std::move(dd.dcl);
  h.best_domain = 0;
  for (int d = 0; d < h.ndom; ++d) {
    Domain &dm = h.dcl[d];
    const int Ldd = (int) (dm.jenv - dm.ienv + 1);
    dm.bitscore = dm.envsc + (L - Ldd) * std::log((double) ((float) L / (float) (L + 3)));
    dm.dombias = cfg.do_null2 ? flogsum(0.0f, std::log((double) omega) + dm.domcorrection) : 0.0f;
    dm.bitscore = (dm.bitscore - (nullsc + dm.dombias)) / kLog2;
    dm.lnP = exp_logsurv(dm.bitscore, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]);
    if (dm.bitscore > h.dcl[h.best_domain].bitscore) h.best_domain = d;
  }
  if (cfg.use_bit_cutoffs) {
    if (target_reportable(cfg, Z_running, h.score, h.lnP)) {
      h.flags |= P7X_IS_REPORTED;
      if (target_includable(cfg, Z_running, h.score, h.lnP)) h.flags |= P7X_IS_INCLUDED;
    }
    for (Domain &dm : h.dcl)
      if (domain_reportable(cfg, cfg.domZ, dm.bitscore, dm.lnP)) {
        dm.is_reported = true;
        if (domain_includable(cfg, cfg.domZ, dm.bitscore, dm.lnP)) dm.is_included = true;
      }
  }
}

static void sort_by_key(p7x_tophits &th)
{
  th.order.resize(th.hits.size());
  for (size_t i = 0; i < th.order.size(); ++i) th.order[i] = (int) i;
  std::stable_sort(th.order.begin(), th.order.end(), [&](int a, int b) {
    const Hit &h1 = th.hits[a], &h2 = th.hits[b];
    if (h1.sortkey != h2.sortkey) return h1.sortkey > h2.sortkey;
    const int c = std::strcmp(h1.name.c_str(), h2.name.c_str());
    if (c != 0) return c < 0;
    const int dir1 = h1.dcl[0].iali < h1.dcl[0].jali ? 1 : -1, dir2 = h2.dcl[0].iali < h2.dcl[0].jali ? 1 : -1;
    if (dir1 != dir2) return dir2 < 0;
    return h1.dcl[0].iali < h2.dcl[0].iali;
  });
  th.sorted_by_key = true;
}

static void threshold(p7x_tophits &th)
{
  p7x_pipeline_cfg &c = th.cfg;
  if (!c.use_bit_cutoffs) {
    for (Hit &h : th.hits) {
      h.flags &= ~(uint32_t) (P7X_IS_REPORTED | P7X_IS_INCLUDED);
      if (!(h.flags & P7X_IS_DUPLICATE) && target_reportable(c, c.Z, h.score, h.lnP)) {
        h.flags |= P7X_IS_REPORTED;
        if (target_includable(c, c.Z, h.score, h.lnP)) h.flags |= P7X_IS_INCLUDED;
      }
    }
  }
  th.nreported = th.nincluded = 0;
  for (const Hit &h : th.hits) { if (h.flags & P7X_IS_REPORTED) th.nreported++; if (h.flags & P7X_IS_INCLUDED) th.nincluded++; }
  if (c.domZ_setby == P7X_ZSETBY_NTARGETS) c.domZ = (double) th.nreported;
  if (!c.use_bit_cutoffs) {
    for (Hit &h : th.hits) {
      for (Domain &d : h.dcl) d.is_reported = d.is_included = false;
      if (h.flags & P7X_IS_REPORTED)
        for (Domain &d : h.dcl) {
          if (domain_reportable(c, c.domZ, d.bitscore, d.lnP)) d.is_reported = true;
          if ((h.flags & P7X_IS_INCLUDED) && domain_includable(c, c.domZ, d.bitscore, d.lnP)) d.is_included = true;
        }
    }
  }
  for (Hit &h : th.hits) {
    h.nreported = h.nincluded = 0;
    for (const Domain &d : h.dcl) { if (d.is_reported) h.nreported++; if (d.is_included) h.nincluded++; }
  }
  // upstream workaround_bug_h74: hide all but one of several envelopes that produced the same alignment
  for (Hit &h : th.hits)
    if (h.noverlaps)
      for (int d1 = 0; d1 < h.ndom; ++d1)
        for (int d2 = d1 + 1; d2 < h.ndom; ++d2)
          if (h.dcl[d1].iali == h.dcl[d2].iali && h.dcl[d1].jali == h.dcl[d2].jali) {
            const int rm = (h.dcl[d1].bitscore >= h.dcl[d2].bitscore) ? d2 : d1;
            if (h.dcl[rm].is_reported) { h.dcl[rm].is_reported = false; h.nreported--; }
            if (h.dcl[rm].is_included) { h.dcl[rm].is_included = false; h.nincluded--; }
          }
}

// The host stage of a batch of queries against one target block.  The per-survivor phases run over the survivors of
// ALL queries at once (one parallel region per phase, not per query), and the single-domain envelopes of all queries
// go to the device in one submission (one launch per model-length class).
int host_finish_batch(const p7x_pipeline_cfg &cfg_in, const std::vector<FinishItem> &items, const HostTargets &tg,
                      const char *const *names, const char *const *accs, const char *const *descs,
                      p7x_tophits **outs, EnvelopeScorer *scorer, EnvelopeScorer *scorer2, EnsembleRunner *ensembles)
{
  const int nq = (int) items.size();
  std::vector<std::unique_ptr<p7x_tophits>> ths((size_t) nq);
  std::vector<int> first((size_t) nq + 1, 0);            // flattened survivor index of every query's first survivor
  for (int q = 0; q < nq; ++q) {
    const FinishItem &it = items[(size_t) q];
    const Profile &p = it.om->p;
    auto th = std::make_unique<p7x_tophits>();
    th->cfg = cfg_in;
    apply_bit_cutoffs(th->cfg, p);
    th->qname = p.name; th->qacc = p.acc; th->qdesc = p.desc; th->q_has_acc = p.has_acc; th->q_has_desc = p.has_desc;
    th->M = p.M;
    th->ctr.nmodels = 1; th->ctr.nnodes = (uint64_t) p.M;
    th->ctr.nseqs = (uint64_t) tg.n; th->ctr.nres = (uint64_t) tg.nres;
    th->ctr.n_past_msv = it.counts[0]; th->ctr.n_past_bias = it.counts[1];
    th->ctr.n_past_vit = it.counts[2]; th->ctr.n_past_fwd = it.counts[3];
    if (it.ms) for (int i = 0; i < 8; ++i) th->ms[i] = it.ms[i];
    if (it.ms && it.nms >= 11) { th->ms[15] = it.ms[8]; th->ms[16] = it.ms[9]; th->ms[17] = it.ms[10]; }
    ths[(size_t) q] = std::move(th);
    first[(size_t) q + 1] = first[(size_t) q] + (int) it.targets->size();
  }
  const int S = first[(size_t) nq];
  const auto t0 = std::chrono::steady_clock::now();
  // option "trace_finish": wall time of the phases of this call on stderr
  const bool debug = debug_opt(OPT_TRACE_FINISH) > 0;
  auto tlast = t0;
  std::string dbg;
  auto tick = [&](const char *what) {
    if (!debug) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[64]; std::snprintf(buf, sizeof buf, " %s %.2f", what, std::chrono::duration<double, std::milli>(now - tlast).count());
    dbg += buf; tlast = now;
  };
  std::vector<Pending> pend((size_t) S);
  std::atomic<int> failed{0};
  int nthreads = cfg_in.host_threads > 0 ? cfg_in.host_threads : usable_cpus();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > (S + 3) / 4) nthreads = (S + 3) / 4;      // at least ~4 targets per worker
  if (nthreads < 1) nthreads = 1;
  auto run_pool = [&](int count, const std::function<void(int)> &body) { HostPool::get().run(count, nthreads, body); };
  // flattened survivor f -> (query q, survivor i of q)
  std::vector<int> q_of((size_t) S);
  for (int q = 0; q < nq; ++q) for (int f = first[(size_t) q]; f < first[(size_t) q + 1]; ++f) q_of[(size_t) f] = q;
  const bool reseed = cfg_in.seed != 0;
  // F3 guard: the device let every target with P <= F3 (1 + g) through.  Those whose P-value (device Forward score) lies
  // above F3 (1 - g) are re-scored in the reference's summation order and F3 is applied to that score; the rest keep
  // the device's score.  <dropped> targets leave the survivor list (and the n_past_fwd count).
  std::vector<float> fwd_use((size_t) S);
  std::vector<char> dropped((size_t) S, 0);
  for (int f = 0; f < S; ++f) { const int q = q_of[(size_t) f]; fwd_use[(size_t) f] = items[(size_t) q].fwdsc[f - first[(size_t) q]]; }
  if (cfg_in.f3_guard > 0.0f && !cfg_in.do_max && !cfg_in.long_targets) {
    const double lo = cfg_in.F3 * (1.0 - (double) cfg_in.f3_guard);
    run_pool(S, [&](int f) {
      const int q = q_of[(size_t) f], i = f - first[(size_t) q];
      const FinishItem &it = items[(size_t) q];
      if (it.near && (it.near->empty() || !(*it.near)[(size_t) i])) return;
      const Profile &p = it.om->p;
      const int t = (*it.targets)[(size_t) i];
      const uint8_t *dsq1 = tg.dsq + tg.off[t] - 1;
      const float filtersc = host_filter_null_score(p, dsq1, tg.len[t], cfg_in.do_biasfilter != 0);
      const float s_dev = (float) ((double) (fwd_use[(size_t) f] - filtersc) / kLog2);
      if (!(exp_surv(s_dev, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]) > lo)) return;          // clearly inside
      float exact = 0.0f;
      if (host_forward_parser_exact(p, dsq1, tg.len[t], &exact) != P7X_OK) return;              // range error: the device's call stands
      const float s_ex = (float) ((double) (exact - filtersc) / kLog2);
      if (exp_surv(s_ex, p.evparam[P7X_FTAU], p.evparam[P7X_FLAMBDA]) > cfg_in.F3) dropped[(size_t) f] = 1;
      else fwd_use[(size_t) f] = exact;
    });
    for (int f = 0; f < S; ++f)
      if (dropped[(size_t) f]) {
        const int q = q_of[(size_t) f];
        ths[(size_t) q]->ctr.n_past_fwd--;
        ths[(size_t) q]->guard_dropped.push_back((*items[(size_t) q].targets)[(size_t) (f - first[(size_t) q])]);
      }
  }
  tick("guard");
  auto finish = [&](int f, DomainDefResult &dd) {
    const int q = q_of[(size_t) f], i = f - first[(size_t) q];
    const FinishItem &it = items[(size_t) q];
    const p7x_pipeline_cfg &cfg = ths[(size_t) q]->cfg;
    const int t = (*it.targets)[(size_t) i];
    const double Zrun = (cfg.Z_setby == P7X_ZSETBY_NTARGETS) ? (double) (t + 1) : cfg.Z;
    finish_one(cfg, it.om->p, tg.len[t], fwd_use[(size_t) f], Zrun, dd, pend[(size_t) f]);
    if (pend[(size_t) f].have) pend[(size_t) f].hit.seqidx = t;
  };
  // regions come from the device scan when there is one, else from the parsers' rows
  auto define = [&](int f, DomainDefResult &dd, std::vector<EnvelopeRequest> *defer) -> int {
    const int q = q_of[(size_t) f], i = f - first[(size_t) q];
    const FinishItem &it = items[(size_t) q];
    const Profile &p = it.om->p;
    const int t = (*it.targets)[(size_t) i];
    const uint8_t *dsq = tg.dsq + tg.off[t] - 1;
    if (!it.regions)
      return domaindef_by_posterior_heuristics(p, dsq, tg.len[t], it.fwd_xmx + it.xmx_off[i], it.bck_xmx + it.xmx_off[i], cfg_in.seed, reseed, dd, defer, i);
    const DeviceRegions *dregs = it.regions;
    const int nr = dregs->n[i];
    if (nr == -3) {
      // the device's region scan met a threshold comparison inside its guard band: this target's parser rows again, in
      // upstream's summation order, and the scan on those (p7x_domaindef.cpp)
      thread_local std::vector<float> ufx, ubx;
      const int pst = parser_rows_upstream(p, dsq, tg.len[t], ufx, ubx);
      if (pst != P7X_OK) return pst;
      dd.nregion_redone = 1;
      return domaindef_by_posterior_heuristics(p, dsq, tg.len[t], ufx.data(), ubx.data(), cfg_in.seed, reseed, dd, defer, i);
    }
    if (nr < 0) return P7X_ERANGE;
    Region regs[256];
    const int32_t *src = dregs->regs + (dregs->start ? (size_t) dregs->start[i] : (size_t) i * dregs->cap) * 3;
    for (int r = 0; r < nr && r < 256; ++r) regs[r] = Region{ src[r * 3], src[r * 3 + 1], src[r * 3 + 2] != 0 };
    return domaindef_from_regions(p, dsq, tg.len[t], dregs->nexpected[i], regs, nr, cfg_in.seed, reseed, dd, defer, i);
  };
  auto on_device = [&](int q) { return scorer != nullptr && items[(size_t) q].device_envelopes; };
  // 1. regions (cheap); single-domain envelopes of the queries that use the device are queued for it, the others
  //    (the CPU test seam, models the envelope kernel does not cover) are rescored right here
  std::vector<DomainDefResult> dds((size_t) S);
  std::vector<std::vector<EnvelopeRequest>> local((size_t) S);
  run_pool(S, [&](int f) {
    if (dropped[(size_t) f]) return;
    const bool dev = on_device(q_of[(size_t) f]);
    const int st = define(f, dds[(size_t) f], dev ? &local[(size_t) f] : nullptr);
    if (st != P7X_OK) { failed.store(st); return; }
    if (!dev) finish(f, dds[(size_t) f]);
  });
  tick("regions");
  double ms_multi = 0.0, ms_env = 0.0, ms_ens_wait = 0.0, ms_env_wait = 0.0;
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  if (failed.load() == 0 && scorer) {
    std::vector<std::vector<EnvelopeRequest>> req((size_t) nq);
    std::vector<std::vector<int>> req_index((size_t) S);
    std::vector<int> heavy;
    for (int f = 0; f < S; ++f) {
      const int q = q_of[(size_t) f];
      if (!on_device(q)) continue;
      for (const EnvelopeRequest &r : local[(size_t) f]) { req_index[(size_t) f].push_back((int) req[(size_t) q].size()); req[(size_t) q].push_back(r); }
      if (!dds[(size_t) f].multi.empty()) heavy.push_back(f);
    }
    // an ensemble is a serial walk (one RNG stream per target): hand the targets out longest first, so that the
    // parallel region does not end on one long target that was started last
    {
      std::vector<double> cost((size_t) S, 0.0);
      for (int f : heavy) {
        double c = 0.0;
        for (const Domain &d : dds[(size_t) f].dcl) if (d.deferred == -2) c += (double) (d.jenv - d.ienv + 1);
        cost[(size_t) f] = c * (double) items[(size_t) q_of[(size_t) f]].om->p.M;
      }
      std::stable_sort(heavy.begin(), heavy.end(), [&](int a, int b) { return cost[(size_t) a] > cost[(size_t) b]; });
    }
    // 2. The stochastic traceback ensembles of the multi-domain regions are the longest chain of this stage: sampled on the
    //    device when there is a runner (every region starts from the re-seeded generator, so regions are independent of each
    //    other), they are queued first; the single-domain envelopes' kernel follows on its own stream.  Without a runner --
    //    or for a region it hands back -- the host workers sample (one serial walk per region, spread target by target).
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<std::vector<EnvelopeRequest>> ereq((size_t) nq);
    std::vector<std::vector<EnsembleResult>> eres;
    std::vector<std::vector<int>> ereq_index((size_t) S);
    const bool dev_ens = ensembles != nullptr && reseed && !cfg_in.long_targets;
    if (dev_ens) {
      for (int f : heavy) {
        const int q = q_of[(size_t) f], i = f - first[(size_t) q];
        for (const Domain &d : dds[(size_t) f].dcl)
          if (d.deferred == -2) { ereq_index[(size_t) f].push_back((int) ereq[(size_t) q].size()); ereq[(size_t) q].push_back(EnvelopeRequest{ i, (int32_t) d.ienv, (int32_t) d.jenv }); }
      }
      std::vector<EnvelopeJob> ejobs((size_t) nq);
      for (int q = 0; q < nq; ++q) ejobs[(size_t) q] = EnvelopeJob{ items[(size_t) q].om, &ereq[(size_t) q], items[(size_t) q].targets };
      const int st = ensembles->begin(ejobs, fast_rng_state(cfg_in.seed), 200);
      if (st != P7X_OK) return st;
      tick("ens_begin");
    }
    std::vector<EnvelopeJob> jobs((size_t) nq);
    bool any = false;
    for (int q = 0; q < nq; ++q) { jobs[(size_t) q] = EnvelopeJob{ items[(size_t) q].om, &req[(size_t) q], items[(size_t) q].targets }; any = any || !req[(size_t) q].empty(); }
    if (any) { const int st = scorer->begin(jobs); if (st != P7X_OK) return st; }
    tick("env_begin");
    if (dev_ens) {
      const auto tw = std::chrono::steady_clock::now();
      const int st = ensembles->wait(eres);
      if (st != P7X_OK) return st;
      ms_ens_wait = since(tw);
      tick("ens_wait");
    }
    const auto t_multi = std::chrono::steady_clock::now();
    // what remains of a region's resolution here is the clustering of the sampled end points; the clustered envelopes go
    // to the device as a second round (scorer2) instead of being rescored here
    std::vector<std::vector<EnvelopeRequest>> local2((size_t) S);
    run_pool((int) heavy.size(), [&](int h) {
      const int f = heavy[(size_t) h], q = q_of[(size_t) f], i = f - first[(size_t) q];
      const int t = (*items[(size_t) q].targets)[(size_t) i];
      std::vector<const EnsembleResult *> mine;
      if (dev_ens) for (int e : ereq_index[(size_t) f]) mine.push_back(&eres[(size_t) q][(size_t) e]);
      const int st = domaindef_finish_multi(items[(size_t) q].om->p, tg.dsq + tg.off[t] - 1, tg.len[t], cfg_in.seed, reseed, dds[(size_t) f],
                                            scorer2 ? &local2[(size_t) f] : nullptr, i, dev_ens ? mine.data() : nullptr);
      if (st != P7X_OK) failed.store(st);
    });
    ms_multi = since(t_multi);                  // the host's own share of the multi-domain regions: clustering, or sampling as well
    tick("multi");
    std::vector<std::vector<EnvelopeRequest>> req2((size_t) nq);
    std::vector<std::vector<int>> req_index2((size_t) S);
    std::vector<EnvelopeJob> jobs2((size_t) nq);
    bool any2 = false;
    if (scorer2 && failed.load() == 0) {
      for (int f : heavy) {
        const int q = q_of[(size_t) f];
        for (const EnvelopeRequest &r : local2[(size_t) f]) { req_index2[(size_t) f].push_back((int) req2[(size_t) q].size()); req2[(size_t) q].push_back(r); }
      }
      for (int q = 0; q < nq; ++q) { jobs2[(size_t) q] = EnvelopeJob{ items[(size_t) q].om, &req2[(size_t) q], items[(size_t) q].targets }; any2 = any2 || !req2[(size_t) q].empty(); }
      if (any2) { const int st = scorer2->begin(jobs2); if (st != P7X_OK) return st; }
    }
    std::vector<std::vector<EnvelopeResult>> res((size_t) nq), res2((size_t) nq);
    if (any) { const auto tw = std::chrono::steady_clock::now(); const int st = scorer->wait(res); if (st != P7X_OK) return st; ms_env_wait += since(tw); }
    tick("env_wait");
    // 3. alignment displays, null2 corrections, per-target scores: the targets without a second-round envelope while that
    //    round runs, the others when it is in
    std::vector<char> second((size_t) S, 0);
    if (any2) for (int f : heavy) if (!local2[(size_t) f].empty()) second[(size_t) f] = 1;
    auto complete = [&](int f) {
      const int q = q_of[(size_t) f];
      if (!on_device(q) || dropped[(size_t) f]) return;
      const int i = f - first[(size_t) q];
      const int t = (*items[(size_t) q].targets)[(size_t) i];
      domaindef_finish_deferred(items[(size_t) q].om->p, tg.dsq + tg.off[t] - 1, tg.len[t], res[(size_t) q], req_index[(size_t) f], dds[(size_t) f],
                                &res2[(size_t) q], &req_index2[(size_t) f]);
      finish(f, dds[(size_t) f]);
    };
    if (!any2) res2.assign((size_t) nq, {});
    if (failed.load() == 0) run_pool(S, [&](int f) { if (!second[(size_t) f]) complete(f); });
    tick("deferred");
    if (any2) {
      const auto tw = std::chrono::steady_clock::now();
      const int st = scorer2->wait(res2); if (st != P7X_OK) return st;
      ms_env_wait += since(tw);
      tick("env2_wait");
      if (failed.load() == 0) run_pool((int) heavy.size(), [&](int h) { const int f = heavy[(size_t) h]; if (second[(size_t) f]) complete(f); });
      tick("deferred2");
    }
    ms_env = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
  }
  for (int f = 0; f < S; ++f) {
    p7x_tophits &th = *ths[(size_t) q_of[(size_t) f]];
    th.oa_redone += dds[(size_t) f].nneartie;
    for (int b = 0; b < 8; ++b) th.oa_why[b] += dds[(size_t) f].neartie_why[b];
    th.ens_device += dds[(size_t) f].nens_device; th.ens_redone += dds[(size_t) f].nens_redone; th.region_redone += dds[(size_t) f].nregion_redone;
  }
  host_prof_dump();
  if (failed.load() != 0) { set_error("domain definition workflow failure"); return failed.load(); }
  const double ms_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // 4. one hit list per query: hits in target order, as the reference's sequential loop would have appended them
  run_pool(nq, [&](int q) {
    const FinishItem &it = items[(size_t) q];
    p7x_tophits &th = *ths[(size_t) q];
    const int n = (int) it.targets->size(), f0 = first[(size_t) q];
    std::vector<int> idx((size_t) n);
    for (int i = 0; i < n; ++i) idx[(size_t) i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return (*it.targets)[(size_t) a] < (*it.targets)[(size_t) b]; });
    for (int i : idx) {
      if (!pend[(size_t) (f0 + i)].have) continue;
      Hit &h = pend[(size_t) (f0 + i)].hit;
      const int t = (*it.targets)[(size_t) i];
      if (names && names[t]) h.name = names[t];
      if (accs && accs[t] && accs[t][0]) { h.acc = accs[t]; h.has_acc = true; }
      if (descs && descs[t] && descs[t][0]) { h.desc = descs[t]; h.has_desc = true; }
      th.hits.push_back(std::move(h));
    }
    th.ms[5] = ms_host; th.ms[8] = ms_env; th.ms[9] = ms_multi;
    th.ms[13] = ms_ens_wait; th.ms[14] = ms_env_wait;
    th.ms[12] = std::max(0.0, ms_host - ms_ens_wait - ms_env_wait);       // the host stage without its waits for the device
    // Z for E-values: number of targets seen (p7_pli_NewSeq), unless set by the caller
    if (th.cfg.Z_setby == P7X_ZSETBY_NTARGETS) th.cfg.Z = (double) tg.n;
    sort_by_key(th);
    threshold(th);
  });
  tick("hitlists");
  if (debug) std::fprintf(stderr, "[finish] nq %d survivors %d threads %d:%s ms\n", nq, S, nthreads, dbg.c_str());
  for (int q = 0; q < nq; ++q) outs[q] = ths[(size_t) q].release();
  return P7X_OK;
}

int host_finish_search(const p7x_pipeline_cfg &cfg_in, const p7x_oprofile *om, const HostTargets &tg,
                       const char *const *names, const char *const *accs, const char *const *descs,
                       const std::vector<int32_t> &tgt, const float *fwdsc,
                       const float *fwd_xmx, const float *bck_xmx, const int64_t *xmx_off,
                       const uint64_t *counts, const double *ms, p7x_tophits **out, EnvelopeScorer *scorer,
                       const DeviceRegions *dregs)
{
  std::vector<FinishItem> items(1);
  FinishItem &it = items[0];
  it.om = om; it.targets = &tgt; it.fwdsc = fwdsc; it.fwd_xmx = fwd_xmx; it.bck_xmx = bck_xmx; it.xmx_off = xmx_off;
  for (int i = 0; i < 4; ++i) it.counts[i] = counts[i];
  it.ms = ms; it.regions = dregs; it.device_envelopes = scorer != nullptr;
  return host_finish_batch(cfg_in, items, tg, names, accs, descs, out, scorer);
}

void tophits_sort_by_key(p7x_tophits &th) { sort_by_key(th); }
void tophits_threshold(p7x_tophits &th) { threshold(th); }
bool tophits_target_reportable(const p7x_pipeline_cfg &c, float score, double lnP) { return target_reportable(c, c.Z, score, lnP); }
int tophits_usable_cpus() { return usable_cpus(); }
void host_parallel_for(int n, int nthreads, const std::function<void(int)> &body)
{
  if (nthreads <= 0) nthreads = usable_cpus();
  HostPool::get().run(n, nthreads, body);
}

void tophits_set_stages(p7x_tophits *th, std::vector<uint8_t> &&stage)
{
  th->stage = std::move(stage);
  for (int32_t t : th->guard_dropped) if ((size_t) t < th->stage.size() && th->stage[(size_t) t] == 4) th->stage[(size_t) t] = 3;
}
void tophits_set_total_ms(p7x_tophits *th, double stage1, double stage2) { th->ms[6] = stage1 + stage2; th->ms[10] = stage1; th->ms[11] = stage2; }

} // namespace p7x

using namespace p7x;

extern "C" {

int p7x_postprocess_targets(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const uint8_t *dsq, const int64_t *offsets,
                            const int32_t *lengths, size_t n, const int32_t *surv, size_t nsurv, const float *fwdsc,
                            const float *fwd_xmx, const float *bck_xmx, const int64_t *xmx_off, const uint64_t *stage_counts,
                            const char *const *names, const char *const *accs, const char *const *descs, p7x_tophits **out)
{
  if (!cfg || !om || !out || (n && (!dsq || !offsets || !lengths)) || (nsurv && (!surv || !fwdsc || !fwd_xmx || !bck_xmx || !xmx_off))) {
    set_error("p7x_postprocess_targets: bad arguments"); return P7X_EINVAL;
  }
  flogsum_init();
  HostTargets tg;
  tg.n = (int64_t) n; tg.len = lengths; tg.off = offsets; tg.dsq = dsq;
  for (size_t t = 0; t < n; ++t) tg.nres += lengths[t];
  std::vector<int32_t> targets(surv, surv + nsurv);
  const uint64_t zero[4] = {0, 0, 0, 0};
  return host_finish_search(*cfg, om, tg, names, accs, descs, targets, fwdsc, fwd_xmx, bck_xmx, xmx_off,
                            stage_counts ? stage_counts : zero, nullptr, out);
}

void p7x_tophits_destroy(p7x_tophits *th) { delete th; }
void p7x_tophits_destroy_many(p7x_tophits **th, size_t n) { if (th) for (size_t i = 0; i < n; ++i) { delete th[i]; th[i] = nullptr; } }
p7x_tophits *p7x_tophits_clone(const p7x_tophits *th) { return th ? new p7x_tophits(*th) : nullptr; }
int64_t p7x_tophits_nhits(const p7x_tophits *th) { return th ? (int64_t) th->hits.size() : -1; }

int p7x_tophits_get_counters(const p7x_tophits *th, p7x_counters *c)
{
  if (!th || !c) return P7X_EINVAL;
  *c = th->ctr;
  c->n_output = 0; c->pos_output = 0;
  return P7X_OK;
}
int p7x_tophits_get_cfg(const p7x_tophits *th, p7x_pipeline_cfg *cfg)
{
  if (!th || !cfg) return P7X_EINVAL;
  *cfg = th->cfg;
  return P7X_OK;
}

static const Hit *hit_at(const p7x_tophits *th, int64_t i)
{
  if (!th || i < 0 || i >= (int64_t) th->hits.size()) return nullptr;
  return &th->hits[th->order.size() == th->hits.size() ? th->order[i] : (int) i];
}

int p7x_tophits_get_hit(const p7x_tophits *th, int64_t i, p7x_hit *o)
{
  const Hit *h = hit_at(th, i);
  if (!h || !o) return P7X_EINVAL;
  std::memset(o, 0, sizeof(*o));
  o->name = h->name.c_str(); o->acc = h->has_acc ? h->acc.c_str() : nullptr; o->desc = h->has_desc ? h->desc.c_str() : nullptr;
  o->seqidx = h->seqidx; o->window_length = h->window_length; o->sortkey = h->sortkey; o->score = h->score; o->pre_score = h->pre_score; o->sum_score = h->sum_score;
  o->lnP = h->lnP; o->pre_lnP = h->pre_lnP; o->sum_lnP = h->sum_lnP; o->nexpected = h->nexpected;
  o->nregions = h->nregions; o->nclustered = h->nclustered; o->noverlaps = h->noverlaps; o->nenvelopes = h->nenvelopes;
  o->ndom = h->ndom; o->flags = h->flags; o->nreported = h->nreported; o->nincluded = h->nincluded; o->best_domain = h->best_domain;
  return P7X_OK;
}

int p7x_tophits_get_domain(const p7x_tophits *th, int64_t i, int32_t d, p7x_domain *o)
{
  const Hit *h = hit_at(th, i);
  if (!h || !o || d < 0 || d >= h->ndom) return P7X_EINVAL;
  const Domain &m = h->dcl[d];
  std::memset(o, 0, sizeof(*o));
  o->ienv = m.ienv; o->jenv = m.jenv; o->iali = m.iali; o->jali = m.jali;
  o->envsc = m.envsc; o->domcorrection = m.domcorrection; o->dombias = m.dombias; o->oasc = m.oasc; o->bitscore = m.bitscore;
  o->lnP = m.lnP; o->is_reported = m.is_reported; o->is_included = m.is_included;
  o->N = m.N; o->hmmfrom = m.hmmfrom; o->hmmto = m.hmmto; o->M = m.M; o->sqfrom = m.sqfrom; o->sqto = m.sqto; o->L = m.L;
  o->model = m.model.c_str(); o->mline = m.mline.c_str(); o->aseq = m.aseq.c_str(); o->ppline = m.ppline.c_str();
  o->rfline = m.rfline.empty() ? nullptr : m.rfline.c_str();
  o->mmline = m.mmline.empty() ? nullptr : m.mmline.c_str();
  o->csline = m.csline.empty() ? nullptr : m.csline.c_str();
  const bool scan = th->cfg.mode == P7X_SCAN_MODELS && th->scan_collected;      // hits are models, the query is the sequence
  const char *qn = th->qname.c_str(), *qa = th->q_has_acc ? th->qacc.c_str() : nullptr, *qd = th->q_has_desc ? th->qdesc.c_str() : nullptr;
  const char *hn = h->name.c_str(), *ha = h->has_acc ? h->acc.c_str() : nullptr, *hd = h->has_desc ? h->desc.c_str() : nullptr;
  o->hmmname = scan ? hn : qn; o->hmmacc = scan ? ha : qa; o->hmmdesc = scan ? hd : qd;
  o->sqname = scan ? qn : hn; o->sqacc = scan ? qa : ha; o->sqdesc = scan ? qd : hd;
  return P7X_OK;
}

// hmmscan orientation (Pipeline._scan_loop, plan7.pyx:6624-6677): per_model[m] is the result of model m against
// all query sequences (searched with cfg.mode = P7X_SCAN_MODELS: nothing pruned); out[s] becomes the hit list of
// sequence s, whose hits are the models.  Reportability is applied in model order with the running Z = number of
// models seen (p7_pli_NewModel), then the usual sort and threshold with Z = nmodels.
// The collection is incremental: the per-model results of one device batch are folded in (in model order) and can be
// released before the next batch is in, so a scan never holds more than a batch of them.
struct p7x_scan_accum {
  std::vector<std::unique_ptr<p7x_tophits>> res;
  size_t nmodels = 0;
  bool in_order = true;            // every model so far was added with the next number: the hit lists are in model order
  std::vector<bool> seen;          // model numbers added so far (a number twice would count its hits and its accounting twice)
  // accounting of the scan, kept flat until the end (20,000 models x 2,100 sequences are 42 M counter updates: through the
  // per-sequence hit lists they cost 76 ms, as four arrays 9): models and nodes are the same for every sequence
  uint64_t nnodes = 0;
  std::vector<uint32_t> past[4];   // [nseqs] models whose MSV / bias / Viterbi / Forward filter the sequence passed
  std::mutex mu;                   // p7x_scan_accum_add_indexed may be called from several threads
};

int p7x_scan_accum_create(const p7x_pipeline_cfg *cfg_in, size_t nseqs, const char *const *seq_names, const char *const *seq_accs,
                          const char *const *seq_descs, const int32_t *seq_lengths, p7x_scan_accum **out)
{
  if (!cfg_in || !out || (nseqs && !seq_lengths)) { set_error("p7x_scan_accum_create: bad arguments"); return P7X_EINVAL; }
  flogsum_init();
  auto acc = std::make_unique<p7x_scan_accum>();
  acc->res.resize(nseqs);
  for (size_t s = 0; s < nseqs; ++s) {
    auto th = std::make_unique<p7x_tophits>();
    th->cfg = *cfg_in; th->cfg.mode = P7X_SCAN_MODELS;
    th->scan_collected = true;
    if (seq_names && seq_names[s]) th->qname = seq_names[s];
    if (seq_accs && seq_accs[s] && seq_accs[s][0]) { th->qacc = seq_accs[s]; th->q_has_acc = true; }
    if (seq_descs && seq_descs[s] && seq_descs[s][0]) { th->qdesc = seq_descs[s]; th->q_has_desc = true; }
    th->ctr.nseqs = 1; th->ctr.nres = (uint64_t) seq_lengths[s];
    acc->res[s] = std::move(th);
  }
  *out = acc.release();
  return P7X_OK;
}

// model_index: the models' numbers in the scan (0-based: the order of the profile database), each exactly once over the
// calls of a scan, in any order and from any thread; NULL: the next nmodels numbers
static int scan_accum_add(p7x_scan_accum *acc, p7x_tophits *const *per_model, const int64_t *model_index, size_t nmodels, const char *who)
{
  if (!acc || (nmodels && !per_model)) { set_error(std::string(who) + ": bad arguments"); return P7X_EINVAL; }
  std::lock_guard<std::mutex> lk(acc->mu);
  const size_t nseqs = acc->res.size();
  for (size_t mm = 0; mm < nmodels; ++mm) {
    if (!per_model[mm]) { set_error(std::string(who) + ": missing per-model result"); return P7X_EINVAL; }
    if (per_model[mm]->ctr.nseqs != nseqs) { set_error(std::string(who) + ": per-model results cover different sequence sets"); return P7X_EINVAL; }
    if (model_index && model_index[mm] < 0) { set_error(std::string(who) + ": negative model number"); return P7X_EINVAL; }
    // model numbers index a bitmap: one that cannot be a position in any profile database is a caller's error, not an allocation
    if (model_index && model_index[mm] >= ((int64_t) 1 << 31)) { set_error(std::string(who) + ": model number out of range"); return P7X_EINVAL; }
  }
  {
    size_t top = 0;
    for (size_t mm = 0; mm < nmodels; ++mm) top = std::max(top, model_index ? (size_t) model_index[mm] : acc->nmodels + mm);
    std::vector<bool> here(nmodels ? top + 1 : 0, false);
    for (size_t mm = 0; mm < nmodels; ++mm) {
      const size_t m = model_index ? (size_t) model_index[mm] : acc->nmodels + mm;
      if (here[m] || (m < acc->seen.size() && acc->seen[m])) { set_error(std::string(who) + ": a model number twice"); return P7X_EINVAL; }
      here[m] = true;
    }
  }
  for (size_t mm = 0; mm < nmodels; ++mm) {
    const p7x_tophits *pm = per_model[mm];
    const size_t m = model_index ? (size_t) model_index[mm] : acc->nmodels + mm;       // the model's number in the scan
    if (m != acc->nmodels + mm) acc->in_order = false;
    if (m >= acc->seen.size()) acc->seen.resize(m + 1, false);
    acc->seen[m] = true;
    acc->nnodes += (uint64_t) pm->M;
    if (pm->stage.size() == nseqs) {
      for (int f = 0; f < 4; ++f) {
        if (acc->past[f].size() != nseqs) acc->past[f].assign(nseqs, 0);
        uint32_t *__restrict dst = acc->past[f].data();
        const auto *__restrict stg = pm->stage.data();
        for (size_t s = 0; s < nseqs; ++s) dst[s] += stg[s] > f;
      }
    }
    for (const Hit &h : pm->hits) {
      if (h.seqidx < 0 || (size_t) h.seqidx >= nseqs) continue;
      p7x_tophits &th = *acc->res[(size_t) h.seqidx];
      p7x_pipeline_cfg rc = th.cfg;
      apply_bit_cutoffs_from(rc, pm->cfg);               // model-specific GA/TC/NC thresholds travel with the per-model result
      const double Zrun = (rc.Z_setby == P7X_ZSETBY_NTARGETS) ? (double) (m + 1) : rc.Z;      // p7_pli_NewModel's running count
      if (!target_reportable(rc, Zrun, h.score, h.lnP)) continue;
      Hit copy = h;
      copy.name = pm->qname; copy.acc = pm->qacc; copy.desc = pm->qdesc; copy.has_acc = pm->q_has_acc; copy.has_desc = pm->q_has_desc;
      copy.seqidx = (int64_t) m;
      th.hits.push_back(std::move(copy));
    }
  }
  acc->nmodels += nmodels;
  return P7X_OK;
}

int p7x_scan_accum_add(p7x_scan_accum *acc, p7x_tophits *const *per_model, size_t nmodels)
{
  return scan_accum_add(acc, per_model, nullptr, nmodels, "p7x_scan_accum_add");
}

int p7x_scan_accum_add_indexed(p7x_scan_accum *acc, p7x_tophits *const *per_model, const int64_t *model_index, size_t nmodels)
{
  if (nmodels && !model_index) { set_error("p7x_scan_accum_add_indexed: no model numbers"); return P7X_EINVAL; }
  return scan_accum_add(acc, per_model, model_index, nmodels, "p7x_scan_accum_add_indexed");
}

int p7x_scan_accum_finish(p7x_scan_accum *acc, p7x_tophits **out)
{ // consumes the accumulator
  if (!acc || !out) { set_error("p7x_scan_accum_finish: bad arguments"); delete acc; return P7X_EINVAL; }
  std::unique_ptr<p7x_scan_accum> owner(acc);
  for (size_t s = 0; s < acc->res.size(); ++s) {
    p7x_tophits &th = *acc->res[s];
    th.ctr.nmodels += acc->nmodels; th.ctr.nnodes += acc->nnodes;
    if (acc->past[0].size() == acc->res.size()) {
      th.ctr.n_past_msv += acc->past[0][s]; th.ctr.n_past_bias += acc->past[1][s]; th.ctr.n_past_vit += acc->past[2][s]; th.ctr.n_past_fwd += acc->past[3][s];
    }
    if (th.cfg.Z_setby == P7X_ZSETBY_NTARGETS) th.cfg.Z = (double) acc->nmodels;
    // results that arrived out of order: back into the order of the models first, which is the order the reference's loop
    // over the profile database appends them in (ties of the key sort keep it)
    if (!acc->in_order) std::stable_sort(th.hits.begin(), th.hits.end(), [](const Hit &a, const Hit &b) { return a.seqidx < b.seqidx; });
    sort_by_key(th);
    threshold(th);
  }
  for (size_t s = 0; s < acc->res.size(); ++s) out[s] = acc->res[s].release();
  return P7X_OK;
}

void p7x_scan_accum_destroy(p7x_scan_accum *acc) { delete acc; }

int p7x_debug_tophits_set_stages(p7x_tophits *th, const uint8_t *stage, size_t n)
{
  if (!th || (n && !stage)) { set_error("p7x_debug_tophits_set_stages: bad arguments"); return P7X_EINVAL; }
  th->stage.assign(stage, stage + n);
  return P7X_OK;
}

int p7x_scan_collect(p7x_tophits *const *per_model, size_t nmodels, const p7x_pipeline_cfg *cfg_in, size_t nseqs,
                     const char *const *seq_names, const char *const *seq_accs, const char *const *seq_descs,
                     const int32_t *seq_lengths, p7x_tophits **out)
{
  if (!per_model || !cfg_in || !out || (nseqs && !seq_lengths)) { set_error("p7x_scan_collect: bad arguments"); return P7X_EINVAL; }
  p7x_scan_accum *acc = nullptr;
  int st = p7x_scan_accum_create(cfg_in, nseqs, seq_names, seq_accs, seq_descs, seq_lengths, &acc);
  if (st != P7X_OK) return st;
  if ((st = p7x_scan_accum_add(acc, per_model, nmodels)) != P7X_OK) { delete acc; return st; }
  return p7x_scan_accum_finish(acc, out);
}

int p7x_tophits_sort_by_key(p7x_tophits *th) { if (!th) return P7X_EINVAL; sort_by_key(*th); return P7X_OK; }
int p7x_tophits_threshold(p7x_tophits *th)   { if (!th) return P7X_EINVAL; threshold(*th); return P7X_OK; }

// p7_tophits_SortBySeqidxAndAlipos: target index, then strand (forward first), then alignment start.
static bool seqidx_less(const Hit &h1, const Hit &h2)
{
  if (h1.seqidx != h2.seqidx) return h1.seqidx < h2.seqidx;
  const int dir1 = h1.dcl[0].iali < h1.dcl[0].jali ? 1 : -1, dir2 = h2.dcl[0].iali < h2.dcl[0].jali ? 1 : -1;
  if (dir1 != dir2) return dir2 < 0;
  return h1.dcl[0].iali < h2.dcl[0].iali;
}
int p7x_tophits_sort_by_seqidx(p7x_tophits *th)
{
  if (!th) return P7X_EINVAL;
  th->order.resize(th->hits.size());
  for (size_t i = 0; i < th->order.size(); ++i) th->order[i] = (int) i;
  std::stable_sort(th->order.begin(), th->order.end(), [&](int a, int b) { return seqidx_less(th->hits[a], th->hits[b]); });
  th->sorted_by_key = false;
  return P7X_OK;
}
int p7x_tophits_is_sorted(const p7x_tophits *th, int by_seqidx)
{
  if (!th) return 0;
  const size_t n = th->hits.size();
  if (th->order.size() != n) return n <= 1;
  for (size_t i = 1; i < n; ++i) {
    const Hit &a = th->hits[th->order[i - 1]], &b = th->hits[th->order[i]];
    if (by_seqidx) { if (seqidx_less(b, a)) return 0; }
    else if (a.sortkey < b.sortkey || (a.sortkey == b.sortkey && std::strcmp(a.name.c_str(), b.name.c_str()) > 0)) return 0;
  }
  return 1;
}
int p7x_tophits_set_hit_text(p7x_tophits *th, int64_t i, int which, const char *value)
{
  if (!th || i < 0 || (size_t) i >= th->hits.size()) return P7X_EINVAL;
  Hit &h = th->order.size() == th->hits.size() ? th->hits[th->order[i]] : th->hits[i];
  switch (which) {
    case 1: if (!value) return P7X_EINVAL; h.name = value; break;
    case 2: h.has_acc = value != nullptr; h.acc = value ? value : ""; break;
    case 4: h.has_desc = value != nullptr; h.desc = value ? value : ""; break;
    default: return P7X_EINVAL;
  }
  return P7X_OK;
}
int p7x_tophits_set_hit_flags(p7x_tophits *th, int64_t i, uint32_t flags)
{
  if (!th || i < 0 || (size_t) i >= th->hits.size()) return P7X_EINVAL;
  Hit &h = th->order.size() == th->hits.size() ? th->hits[th->order[i]] : th->hits[i];
  h.flags = flags;
  return P7X_OK;
}

// TopHits.merge (plan7.pyx:9172-9276): p7_tophits_Merge (concatenate, re-sort), p7_pipeline_Merge (add the
// accounting; Z too when it counts targets), clear REPORTED/INCLUDED unless bit cutoffs, re-threshold.
p7x_tophits *p7x_tophits_deserialize(const void *buf, size_t n);
static int merge_check(const p7x_tophits *dst, const p7x_tophits *src)
{
  if (dst->qname != src->qname) { set_error("Trying to merge `TopHits` obtained from different queries"); return P7X_EINVAL; }
  const p7x_pipeline_cfg &a = dst->cfg, &b = src->cfg;
  if (a.by_E != b.by_E || a.dom_by_E != b.dom_by_E || a.inc_by_E != b.inc_by_E || a.incdom_by_E != b.incdom_by_E ||
      a.use_bit_cutoffs != b.use_bit_cutoffs || a.Z_setby != b.Z_setby || a.domZ_setby != b.domZ_setby ||
      (a.by_E ? a.E != b.E : a.T != b.T) || (a.dom_by_E ? a.domE != b.domE : a.domT != b.domT) ||
      (a.inc_by_E ? a.incE != b.incE : a.incT != b.incT) || (a.incdom_by_E ? a.incdomE != b.incdomE : a.incdomT != b.incdomT) ||
      (a.Z_setby != P7X_ZSETBY_NTARGETS && a.Z != b.Z) || (a.domZ_setby != P7X_ZSETBY_NTARGETS && a.domZ != b.domZ)) {
    set_error("Trying to merge `TopHits` obtained from pipelines configured with different parameters");
    return P7X_EINVAL;
  }
  return P7X_OK;
}
// concatenation and accounting; <src>'s hits are moved when it is not needed afterwards
static void merge_append(p7x_tophits *dst, p7x_tophits *src, bool move_hits)
{
  dst->hits.reserve(dst->hits.size() + src->hits.size());
  if (move_hits) for (Hit &h : src->hits) dst->hits.push_back(std::move(h));
  else for (const Hit &h : src->hits) dst->hits.push_back(h);
  dst->ctr.nseqs += src->ctr.nseqs; dst->ctr.nres += src->ctr.nres;
  dst->ctr.nmodels = std::max(dst->ctr.nmodels, src->ctr.nmodels); dst->ctr.nnodes = std::max(dst->ctr.nnodes, src->ctr.nnodes);
  dst->ctr.n_past_msv += src->ctr.n_past_msv; dst->ctr.n_past_bias += src->ctr.n_past_bias;
  dst->ctr.n_past_vit += src->ctr.n_past_vit; dst->ctr.n_past_fwd += src->ctr.n_past_fwd;
  if (dst->cfg.Z_setby == P7X_ZSETBY_NTARGETS) dst->cfg.Z += src->cfg.Z;
  for (int i = 0; i < p7x_tophits::kMs; ++i) dst->ms[i] += src->ms[i];
}
static void merge_finalize(p7x_tophits *dst)
{
  if (!dst->cfg.use_bit_cutoffs)
    for (Hit &h : dst->hits) {
      h.flags &= ~(uint32_t) (P7X_IS_REPORTED | P7X_IS_INCLUDED);
      h.nreported = h.nincluded = 0;
      for (Domain &d : h.dcl) d.is_reported = d.is_included = false;
    }
  sort_by_key(*dst);
  threshold(*dst);
}

int p7x_tophits_merge(p7x_tophits *dst, const p7x_tophits *src)
{
  if (!dst || !src) return P7X_EINVAL;
  const int st = merge_check(dst, src);
  if (st != P7X_OK) return st;
  merge_append(dst, const_cast<p7x_tophits *>(src), false);
  merge_finalize(dst);
  return P7X_OK;
}

int p7x_tophits_merge_longtargets(p7x_tophits **parts, size_t nparts, p7x_tophits **out)
{
  if (!parts || !out || nparts == 0) { set_error("p7x_tophits_merge_longtargets: bad arguments"); return P7X_EINVAL; }
  *out = nullptr;
  std::vector<std::unique_ptr<p7x_tophits>> own;
  for (size_t r = 0; r < nparts; ++r) { own.emplace_back(parts[r]); parts[r] = nullptr; }      // consumed, also on failure
  for (size_t r = 0; r < nparts; ++r) {
    if (!own[r] || !own[r]->lt_unfinished || !own[r]->cfg.long_targets) { set_error("p7x_tophits_merge_longtargets: not an unfinished part of a long-target search"); return P7X_EINVAL; }
    if (own[r]->qname != own[0]->qname || own[r]->cfg.lt_nparts != (int) nparts || own[r]->ctr.nres != own[0]->ctr.nres ||
        own[r]->lt_evalue_window != own[0]->lt_evalue_window) { set_error("p7x_tophits_merge_longtargets: the parts belong to different searches"); return P7X_EINVAL; }
  }
  std::vector<size_t> by_part(nparts);
  for (size_t r = 0; r < nparts; ++r) by_part[r] = r;
  std::sort(by_part.begin(), by_part.end(), [&](size_t a, size_t b) { return own[a]->cfg.lt_part < own[b]->cfg.lt_part; });
  for (size_t r = 0; r < nparts; ++r) if (own[by_part[r]]->cfg.lt_part != (int) r) { set_error("p7x_tophits_merge_longtargets: a part is missing or given twice"); return P7X_EINVAL; }
  std::unique_ptr<p7x_tophits> dst = std::move(own[by_part[0]]);
  for (size_t r = 1; r < nparts; ++r) {
    p7x_tophits *src = own[by_part[r]].get();
    // every part kept the residue accounting of the WHOLE search (the reference's pli.nres, which the reportability test
    // of a window reads as it goes), so nres / nseqs are not added; the window counters are per part
    for (Hit &h : src->hits) dst->hits.push_back(std::move(h));
    dst->ctr.n_past_msv += src->ctr.n_past_msv; dst->ctr.n_past_bias += src->ctr.n_past_bias;
    dst->ctr.n_past_vit += src->ctr.n_past_vit; dst->ctr.n_past_fwd += src->ctr.n_past_fwd;
    dst->ctr.pos_past_msv += src->ctr.pos_past_msv; dst->ctr.pos_past_bias += src->ctr.pos_past_bias;
    dst->ctr.pos_past_vit += src->ctr.pos_past_vit; dst->ctr.pos_past_fwd += src->ctr.pos_past_fwd;
    for (int i = 0; i < p7x_tophits::kMs; ++i) dst->ms[i] = std::max(dst->ms[i], src->ms[i]);      // the parts ran side by side
  }
  dst->cfg.lt_part = 0; dst->cfg.lt_nparts = 1;
  longtarget_finalize(dst.get(), dst->lt_evalue_window, longtarget_res_count(dst->cfg, dst->ctr.nres));
  *out = dst.release();
  return P7X_OK;
}

// The merging side of a sharded many-query search (rank 0 of `bench.py --gpus N`, the reference's
// _ReverseSEARCHDispatcher collecting its chunks, _hmmsearch.py:259-263): blobs[q * nparts + r] is the serialised hit
// list of query q on shard r.  Per query: deserialise, concatenate in shard order, ONE sort and ONE threshold (the result
// of TopHits.merge over the same lists: sort and threshold depend only on the concatenation), the queries spread over the
// host workers.  Empty shards (size 0 / NULL) are skipped; a query whose shards are all empty is an error.
int p7x_tophits_merge_many(const void *const *blobs, const size_t *sizes, size_t nq, size_t nparts, int threads, p7x_tophits **outs)
{
  if (!blobs || !sizes || !outs || nparts == 0) { set_error("p7x_tophits_merge_many: bad arguments"); return P7X_EINVAL; }
  for (size_t q = 0; q < nq; ++q) outs[q] = nullptr;
  std::atomic<int> failed{ P7X_OK };
  std::mutex err_mu; std::string err;
  host_parallel_for((int) nq, threads, [&](int qi) {
    const size_t q = (size_t) qi;
    std::unique_ptr<p7x_tophits> dst;
    auto fail = [&](int st) { int expect = P7X_OK; if (failed.compare_exchange_strong(expect, st)) { std::lock_guard<std::mutex> lk(err_mu); err = p7x_last_error(); } };
    for (size_t r = 0; r < nparts; ++r) {
      const void *b = blobs[q * nparts + r]; const size_t n = sizes[q * nparts + r];
      if (!b || n == 0) continue;
      std::unique_ptr<p7x_tophits> part(p7x_tophits_deserialize(b, n));
      if (!part) { fail(P7X_EINVAL); return; }
      if (!dst) { dst = std::move(part); continue; }
      const int st = merge_check(dst.get(), part.get());
      if (st != P7X_OK) { fail(st); return; }
      merge_append(dst.get(), part.get(), true);
    }
    if (!dst) { set_error("p7x_tophits_merge_many: a query without any shard result"); fail(P7X_EINVAL); return; }
    merge_finalize(dst.get());
    outs[q] = dst.release();
  });
  if (failed.load() != P7X_OK) {
    for (size_t q = 0; q < nq; ++q) { delete outs[q]; outs[q] = nullptr; }
    set_error(err.c_str());
    return failed.load();
  }
  return P7X_OK;
}

} // extern "C"

// ---------------------------------------------------------------- serialisation (TopHits pickling, plan7.pyx:8394-8572)
namespace {
struct Writer {
  std::vector<uint8_t> b;
  template <class T> void pod(const T &v) { const uint8_t *q = reinterpret_cast<const uint8_t *>(&v); b.insert(b.end(), q, q + sizeof(T)); }
  void str(const std::string &s) { const uint32_t n = (uint32_t) s.size(); pod(n); b.insert(b.end(), s.begin(), s.end()); }
};
struct Reader {
  const uint8_t *p, *e; bool ok = true;
  template <class T> void pod(T &v) { if (p + sizeof(T) > e) { ok = false; return; } std::memcpy(&v, p, sizeof(T)); p += sizeof(T); }
  void str(std::string &s) { uint32_t n = 0; pod(n); if (!ok || p + n > e) { ok = false; return; } s.assign((const char *) p, n); p += n; }
};
constexpr uint32_t kMagic = 0x70377879u;   // "p7xy" (format 6: the ABI version and the size of the configuration record follow the magic)

template <class IO> void io_domain(IO &io, Domain &d)
{
  io.pod(d.ienv); io.pod(d.jenv); io.pod(d.iali); io.pod(d.jali); io.pod(d.envsc); io.pod(d.domcorrection); io.pod(d.dombias);
  io.pod(d.oasc); io.pod(d.bitscore); io.pod(d.lnP); io.pod(d.is_reported); io.pod(d.is_included);
  io.pod(d.N); io.pod(d.hmmfrom); io.pod(d.hmmto); io.pod(d.M); io.pod(d.sqfrom); io.pod(d.sqto); io.pod(d.L);
  io.str(d.model); io.str(d.mline); io.str(d.aseq); io.str(d.ppline); io.str(d.rfline); io.str(d.mmline); io.str(d.csline);
}
template <class IO> void io_hit(IO &io, Hit &h)
{
  io.str(h.name); io.str(h.acc); io.str(h.desc); io.pod(h.has_acc); io.pod(h.has_desc); io.pod(h.seqidx); io.pod(h.window_length); io.pod(h.sortkey);
  io.pod(h.score); io.pod(h.pre_score); io.pod(h.sum_score); io.pod(h.lnP); io.pod(h.pre_lnP); io.pod(h.sum_lnP);
  io.pod(h.nexpected); io.pod(h.nregions); io.pod(h.nclustered); io.pod(h.noverlaps); io.pod(h.nenvelopes); io.pod(h.ndom);
  io.pod(h.flags); io.pod(h.nreported); io.pod(h.nincluded); io.pod(h.best_domain);
}
} // namespace

extern "C" {

int64_t p7x_tophits_serialize(const p7x_tophits *th, void *buf, size_t cap)
{
  if (!th) return -1;
  Writer w;
  uint32_t magic = kMagic; w.pod(magic);
  // the layout follows the configuration record and the timing array, i.e. the ABI: a blob of another ABI version (a pickled
  // TopHits, a rank running an older library) is rejected, not mis-parsed
  uint32_t abi = (uint32_t) P7X_ABI_VERSION, cfg_bytes = (uint32_t) sizeof(p7x_pipeline_cfg); w.pod(abi); w.pod(cfg_bytes);
  w.pod(th->cfg); w.pod(th->ctr);
  w.str(th->qname); w.str(th->qacc); w.str(th->qdesc); w.pod(th->q_has_acc); w.pod(th->q_has_desc); w.pod(th->M); w.pod(th->scan_collected);
  for (int i = 0; i < p7x_tophits::kMs; ++i) w.pod(th->ms[i]);
  const uint64_t n = th->hits.size(); w.pod(n);
  for (const Hit &hc : th->hits) {
    Hit &h = const_cast<Hit &>(hc);
    io_hit(w, h);
    for (Domain &d : h.dcl) io_domain(w, d);
  }
  // presentation state: the order (key or seqidx sort), the counts, and the flags inside the hits are kept as they are
  w.pod(th->sorted_by_key); w.pod(th->nreported); w.pod(th->nincluded);
  const uint64_t no = th->order.size(); w.pod(no);
  for (int o : th->order) w.pod(o);
  if (buf && cap >= w.b.size()) std::memcpy(buf, w.b.data(), w.b.size());
  return (int64_t) w.b.size();
}

p7x_tophits *p7x_tophits_deserialize(const void *buf, size_t n)
{
  if (!buf) return nullptr;
  Reader r{ (const uint8_t *) buf, (const uint8_t *) buf + n };
  uint32_t magic = 0; r.pod(magic);
  if (!r.ok || magic != kMagic) { set_error("not a serialised TopHits"); return nullptr; }
  uint32_t abi = 0, cfg_bytes = 0; r.pod(abi); r.pod(cfg_bytes);
  if (!r.ok || abi != (uint32_t) P7X_ABI_VERSION || cfg_bytes != (uint32_t) sizeof(p7x_pipeline_cfg)) {
    set_error("serialised TopHits of another library version (ABI " + std::to_string(abi) + ", this library " + std::to_string(P7X_ABI_VERSION) + ")");
    return nullptr;
  }
  auto th = std::make_unique<p7x_tophits>();
  r.pod(th->cfg); r.pod(th->ctr);
  r.str(th->qname); r.str(th->qacc); r.str(th->qdesc); r.pod(th->q_has_acc); r.pod(th->q_has_desc); r.pod(th->M); r.pod(th->scan_collected);
  for (int i = 0; i < p7x_tophits::kMs; ++i) r.pod(th->ms[i]);
  uint64_t nh = 0; r.pod(nh);
  if (!r.ok) { set_error("truncated serialised TopHits"); return nullptr; }
  // a hit takes at least its fixed fields (three length words, the scores and counts): a count the buffer cannot hold
  // is corruption, not an allocation request
  constexpr size_t kMinHit = 3 * 4 + 2 + 8 + 4 + 8 + 3 * 4 + 3 * 8 + 4 + 5 * 4 + 4 + 3 * 4;
  if (nh > (uint64_t) (r.e - r.p) / kMinHit) { set_error("corrupt serialised TopHits"); return nullptr; }
  try {
    th->hits.resize(nh);
    for (Hit &h : th->hits) {
      io_hit(r, h);
      if (!r.ok || h.ndom < 0 || (uint64_t) h.ndom > (uint64_t) (r.e - r.p) / 64) { set_error("corrupt serialised TopHits"); return nullptr; }
      h.dcl.resize(h.ndom);
      for (Domain &d : h.dcl) io_domain(r, d);
    }
    uint64_t no = 0;
    r.pod(th->sorted_by_key); r.pod(th->nreported); r.pod(th->nincluded); r.pod(no);
    if (!r.ok || (no != 0 && no != nh)) { set_error("truncated serialised TopHits"); return nullptr; }
    th->order.resize(no);
    for (int &o : th->order) { r.pod(o); if (!r.ok || o < 0 || (uint64_t) o >= nh) { set_error("corrupt serialised TopHits"); return nullptr; } }
  } catch (const std::exception &) {
    set_error("corrupt serialised TopHits"); return nullptr;
  }
  if (!r.ok) { set_error("truncated serialised TopHits"); return nullptr; }
  return th.release();
}

int p7x_tophits_get_guard_counts(const p7x_tophits *th, int64_t *f3_dropped, int64_t *oa_redone, int64_t *oa_why)
{
  if (!th) return P7X_EINVAL;
  if (f3_dropped) *f3_dropped = (int64_t) th->guard_dropped.size();
  if (oa_redone) *oa_redone = th->oa_redone;
  if (oa_why) for (int b = 0; b < 8; ++b) oa_why[b] = th->oa_why[b];
  return P7X_OK;
}

int p7x_tophits_get_ensemble_counts(const p7x_tophits *th, int64_t *sampled_on_device, int64_t *redone_by_host, int64_t *region_scans_redone)
{
  if (!th) return P7X_EINVAL;
  if (region_scans_redone) *region_scans_redone = th->region_redone;
  if (sampled_on_device) *sampled_on_device = th->ens_device;
  if (redone_by_host) *redone_by_host = th->ens_redone;
  return P7X_OK;
}

int p7x_tophits_get_timings(const p7x_tophits *th, double *ms, int n)
{
  if (!th || !ms) return P7X_EINVAL;
  for (int i = 0; i < n && i < p7x_tophits::kMs; ++i) ms[i] = th->ms[i];
  return P7X_OK;
}

} // extern "C"
