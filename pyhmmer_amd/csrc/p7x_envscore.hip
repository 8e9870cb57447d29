// p7x_envscore.hip -- host side of the envelope kernel (p7x_envelope.hip): batches the envelope requests of one search,
// runs env_kernel on a leased stream and hands the results (scores, null2 odds, optimal-accuracy traces) back to the
// host stage (p7x_domaindef.cpp / p7x_tophits.cpp) through the EnvelopeScorer interface.
#include "p7x_wave.hpp"
#include "p7x_host.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>

namespace p7x {

// ---------------------------------------------------------------------------- envelope rescoring on the device
// Device and pinned-host buffers are grow-only, like the cascade workspace; a buffer that has to grow goes back to the
// context's slab pool and the larger one comes from there (hipFree would wait for every stream of the device, i.e. for
// the cascades of the other searches in flight).
struct EnvBuffers {
  int device = -1;
  float *work = nullptr; size_t work_floats = 0, work_bytes = 0;
  unsigned char *d_in = nullptr; size_t d_in_cap = 0;        // env_sq | tr_off | env_len | env_L | EnvArgs records
  unsigned char *h_in = nullptr; size_t h_in_cap = 0;        // pinned mirror of d_in
  unsigned char *d_out = nullptr; size_t d_out_cap = 0;      // out_sc | out_null2 | out_status | tr_n | tr_a | tr_i | tr_pp
  unsigned char *h_out = nullptr; size_t h_out_cap = 0;      // pinned mirror of d_out
  float *lt_dev = nullptr; size_t lt_bytes = 0;              // long-target envelopes: their emission tables
  hipStream_t stream = nullptr;                               // per host thread: concurrent host stages do not wait on each other
  ~EnvBuffers() {
    if (device < 0) return;
    (void) hipSetDevice(device);
    if (stream) (void) hipStreamDestroy(stream);
    DeviceCtx *ctx = nullptr;
    if (get_ctx(device, &ctx) == P7X_OK) { slab_release(ctx, work, work_bytes); slab_release(ctx, d_in, d_in_cap); slab_release(ctx, d_out, d_out_cap); slab_release(ctx, lt_dev, lt_bytes); }
    pinned_release(h_out, h_out_cap);
    pinned_release(h_in, h_in_cap);
  }
};
// leased from a process-wide pool like the cascade workspaces (never destroyed: no device teardown from exiting threads)
struct EnvPool { std::mutex mu; std::vector<EnvBuffers *> all; std::vector<char> busy; };
static EnvPool &env_pool() { static EnvPool *p = new EnvPool(); return *p; }
static void release_env_buffers(EnvBuffers *eb)
{
  if (!eb) return;
  EnvPool &ep = env_pool();
  std::lock_guard<std::mutex> lk(ep.mu);
  for (size_t i = 0; i < ep.all.size(); ++i) if (ep.all[i] == eb) ep.busy[i] = 0;
}

static size_t env_budget_bytes()
{
  size_t gb = 24;
  if (debug_opt(OPT_ENV_WORKSPACE_GB) > 0) gb = (size_t) debug_opt(OPT_ENV_WORKSPACE_GB);      // tests: a workspace too small for every envelope at once
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 2 < gb << 30) return free_b / 2;
  return gb << 30;
}

class DeviceEnvelopeScorer final : public EnvelopeScorer {
public:
  DeviceEnvelopeScorer(DeviceCtx *ctx, const p7x_seqdb *db) : ctx_(ctx), db_(db) {}
  ~DeviceEnvelopeScorer() override { if (lease_) { if (lease_->stream) (void) hipStreamSynchronize(lease_->stream); release_env_buffers(lease_); } }

  int begin(const std::vector<EnvelopeJob> &jobs) override
  {
    const bool debug = debug_opt(OPT_TRACE_FINISH) > 0;
    auto tlast = std::chrono::steady_clock::now();
    std::string dbg;
    auto tick = [&](const char *what) {
      if (!debug) return;
      const auto now = std::chrono::steady_clock::now();
      char buf[64]; std::snprintf(buf, sizeof buf, " %s %.2f", what, std::chrono::duration<double, std::milli>(now - tlast).count());
      dbg += buf; tlast = now;
    };
    jobs_ = jobs;
    meta_.assign(jobs.size(), JobMeta{});
    int64_t nenv_tot = 0;
    for (size_t j = 0; j < jobs.size(); ++j) { meta_[j].first = nenv_tot; meta_[j].nenv = (int) jobs[j].req->size(); nenv_tot += meta_[j].nenv; }
    nenv_ = nenv_tot;
    if (nenv_tot == 0) return P7X_OK;
    P7X_HIP(hipSetDevice(db_->device));
    EnvBuffers *eb = nullptr;
    {
      EnvPool &ep = env_pool();
      std::lock_guard<std::mutex> lk(ep.mu);
      size_t pick = ep.all.size();       // the idle set with the largest workspace: fewer buffers have to grow
      for (size_t i = 0; i < ep.all.size(); ++i)
        if (!ep.busy[i] && ep.all[i]->device == db_->device && (pick == ep.all.size() || ep.all[i]->work_bytes > ep.all[pick]->work_bytes)) pick = i;
      if (pick < ep.all.size()) { ep.busy[pick] = 1; eb = ep.all[pick]; }
      if (!eb) { eb = new EnvBuffers(); eb->device = db_->device; ep.all.push_back(eb); ep.busy.push_back(1); }
    }
    lease_ = eb;
    tick("lease");
    if (!eb->stream) {
      const int cst = create_tail_stream(ctx_, false, &eb->stream); if (cst != P7X_OK) return cst;
    }
    // inputs: [env_sq i64][tr_off i64][env_len i32][env_L i32][order i32] over all envelopes, one queue cursor per job,
    // then one EnvArgs record per job
    const size_t nj = jobs.size();
    const size_t o_cursor = (size_t) nenv_tot * (8 + 8 + 4 + 4 + 4);
    const size_t o_args = (o_cursor + nj * 4 + 255) & ~(size_t) 255;
    const size_t in_bytes = o_args + nj * sizeof(EnvArgs);
    if (in_bytes > eb->h_in_cap) {
      pinned_release(eb->h_in, eb->h_in_cap); eb->h_in = nullptr; eb->h_in_cap = 0;
      void *hp = nullptr; size_t got = 0; const int pst = pinned_acquire(std::max<size_t>(in_bytes * 2, (size_t) 1 << 20), &hp, &got); if (pst != P7X_OK) return pst;
      eb->h_in = static_cast<unsigned char *>(hp); eb->h_in_cap = got;
    }
    if (in_bytes > eb->d_in_cap) {
      slab_release(ctx_, eb->d_in, eb->d_in_cap); eb->d_in = nullptr; eb->d_in_cap = 0;
      void *dp = nullptr; size_t got = 0; const int sst = slab_acquire(ctx_, std::max<size_t>(in_bytes * 2, (size_t) 1 << 20), &dp, &got); if (sst != P7X_OK) return sst;
      eb->d_in = static_cast<unsigned char *>(dp); eb->d_in_cap = got;
    }
    tick("in_bufs");
    int64_t *env_sq = reinterpret_cast<int64_t *>(eb->h_in);
    int64_t *tr_off = env_sq + nenv_tot;
    int32_t *env_len = reinterpret_cast<int32_t *>(tr_off + nenv_tot);
    int32_t *env_L = env_len + nenv_tot;
    int32_t *env_order = env_L + nenv_tot;
    int32_t *h_cursor = reinterpret_cast<int32_t *>(eb->h_in + o_cursor);
    for (size_t j = 0; j < nj; ++j) h_cursor[j] = 0;
    EnvArgs *h_args = reinterpret_cast<EnvArgs *>(eb->h_in + o_args);
    int64_t ntr = 0;
    // jobs of one model-length class (nodes per lane C) go into one launch: order them by class
    order_.resize(nj);
    for (size_t j = 0; j < nj; ++j) order_[j] = (int) j;
    int st = P7X_OK;
    for (size_t j = 0; j < nj; ++j) {
      JobMeta &m = meta_[j];
      if (m.nenv == 0) continue;
      const Profile &p = jobs[j].om->p;
      if ((st = get_dev_profile(jobs[j].om, ctx_, &m.dp)) != P7X_OK) return st;
      m.C = m.dp->vitC; m.Lmax = 1;
      for (int r = 0; r < m.nenv; ++r) {
        const EnvelopeRequest &rq = (*jobs[j].req)[(size_t) r];
        const int t = (*jobs[j].targets)[(size_t) rq.item];
        const int Ld = rq.j - rq.i + 1;
        const int64_t e = m.first + r;
        env_sq[e] = db_->h_off[t] + (rq.i - 1);
        env_len[e] = Ld; env_L[e] = jobs[j].lt_tables ? Ld : db_->h_len[t];      // long targets: the envelope's own length model
        tr_off[e] = ntr; ntr += (int64_t) Ld + p.M + 16;
        m.Lmax = std::max(m.Lmax, Ld);
      }
      // longest first (job-relative indices)
      int32_t *ord = env_order + m.first;
      for (int r = 0; r < m.nenv; ++r) ord[r] = r;
      const int32_t *len = env_len + m.first;
      std::stable_sort(ord, ord + m.nenv, [len](int32_t x, int32_t y) { return len[x] > len[y]; });
    }
    std::stable_sort(order_.begin(), order_.end(), [&](int x, int y) { return meta_[x].C < meta_[y].C; });
    tick("requests");
    // workspace: one slab per wavefront of every job, sized for the longest envelope of the job's class
    std::map<int, int> class_Lmax;
    for (size_t j = 0; j < nj; ++j) if (meta_[j].nenv) { int &v = class_Lmax[meta_[j].C]; v = std::max(v, meta_[j].Lmax); }
    const size_t budget = env_budget_bytes();
    int shrink = 1;
    size_t work_floats = 0;
    for (;; shrink *= 2) {
      work_floats = 0;
      for (size_t j = 0; j < nj; ++j) {
        JobMeta &m = meta_[j];
        if (m.nenv == 0) continue;
        int cap_blocks = 0;
        if ((st = env_max_blocks(m.C, jobs[j].om->p.Kp + 1, ctx_->num_cu, &cap_blocks)) != P7X_OK) return st;
        // the jobs share the resident blocks in proportion to their envelopes; a job with more envelopes than wavefronts
        // hands them out longest first (EnvArgs::order / cursor)
        const int share = (int) std::max<int64_t>(1, (int64_t) cap_blocks * m.nenv / nenv_tot);
        m.nblocks = std::max(1, std::min(share, (m.nenv + env_waves(m.C) - 1) / env_waves(m.C)) / shrink);
        m.stride = env_work_floats(m.C, class_Lmax[m.C]);
        work_floats += (size_t) m.nblocks * env_waves(m.C) * m.stride;
      }
      if (work_floats * 4 <= budget || work_floats <= eb->work_floats) break;
      bool all_one = true;
      for (size_t j = 0; j < nj; ++j) if (meta_[j].nenv && meta_[j].nblocks > 1) all_one = false;
      if (all_one) { set_error("envelope workspace does not fit in device memory"); return P7X_EMEM; }
    }
    if (work_floats > eb->work_floats) {
      slab_release(ctx_, eb->work, eb->work_bytes); eb->work = nullptr; eb->work_floats = 0; eb->work_bytes = 0;
      void *dp = nullptr; size_t got = 0; // 25 % headroom and never less than 1 GiB: a large device allocation stalls this host stage for tens of ms
      const int sst = slab_acquire(ctx_, std::max<size_t>(work_floats * 4 + work_floats, (size_t) 1 << 30), &dp, &got); if (sst != P7X_OK) return sst;
      eb->work = static_cast<float *>(dp); eb->work_bytes = got; eb->work_floats = got / 4;
    }
    tick("work");
    // outputs: [out_sc 2f][null2 32f][status i][tr_n i][orig f] per envelope, then the three trace arrays
    const size_t n = (size_t) nenv_tot;
    const size_t o_sc = 0, o_n2 = o_sc + n * 8, o_st = o_n2 + n * 128, o_n = o_st + n * 4, o_orig = o_n + n * 4;
    const size_t o_ta = o_orig + n * 4, o_ti = o_ta + (size_t) ntr * 4, o_tp = o_ti + (size_t) ntr * 4;
    const size_t out_bytes = o_tp + (size_t) ntr * 4;
    // long-target jobs: the envelopes' own emission tables go up next to the requests
    size_t lt_floats = 0;
    for (size_t j = 0; j < nj; ++j) if (jobs[j].lt_tables) lt_floats += (size_t) meta_[j].nenv * jobs[j].lt_stride;
    if (lt_floats * 4 > eb->lt_bytes) {
      slab_release(ctx_, eb->lt_dev, eb->lt_bytes); eb->lt_dev = nullptr; eb->lt_bytes = 0;
      void *dp = nullptr; size_t got = 0; const int sst = slab_acquire(ctx_, lt_floats * 4 + lt_floats, &dp, &got); if (sst != P7X_OK) return sst;
      eb->lt_dev = static_cast<float *>(dp); eb->lt_bytes = got;
    }
    if (out_bytes > eb->d_out_cap) {
      slab_release(ctx_, eb->d_out, eb->d_out_cap); eb->d_out = nullptr; eb->d_out_cap = 0;
      pinned_release(eb->h_out, eb->h_out_cap); eb->h_out = nullptr; eb->h_out_cap = 0;
      size_t cap = std::max<size_t>(out_bytes + out_bytes / 2, (size_t) 16 << 20);     // growing is a stall (pinned allocation): start generous
      { void *dp = nullptr; size_t got = 0; const int sst = slab_acquire(ctx_, cap, &dp, &got); if (sst != P7X_OK) return sst;
        eb->d_out = static_cast<unsigned char *>(dp); eb->d_out_cap = cap = got; }
      { void *hp = nullptr; size_t got = 0; const int pst = pinned_acquire(cap, &hp, &got); if (pst != P7X_OK) return pst;
        eb->h_out = static_cast<decltype(eb->h_out)>(hp); eb->h_out_cap = got; }
    }
    tick("out_bufs");
    hipStream_t s = eb->stream;
    const int64_t *d_env_sq = reinterpret_cast<const int64_t *>(eb->d_in);
    const int64_t *d_tr_off = d_env_sq + nenv_tot;
    const int32_t *d_env_len = reinterpret_cast<const int32_t *>(d_tr_off + nenv_tot);
    const int32_t *d_env_L = d_env_len + nenv_tot;
    const int32_t *d_env_order = d_env_L + nenv_tot;
    int *d_cursor = reinterpret_cast<int *>(eb->d_in + o_cursor);
    // argument records in launch order (class by class)
    size_t slab_floats = 0, lt_at = 0;
    std::vector<std::pair<int, int>> runs;        // first record, count
    int nrec = 0;
    for (size_t k = 0; k < nj; ++k) {
      const int j = order_[k];
      const JobMeta &m = meta_[(size_t) j];
      if (m.nenv == 0) continue;
      const Profile &p = jobs[(size_t) j].om->p;
      EnvArgs a{};
      a.M = p.M; a.C = m.C; a.K = p.K; a.nrows = p.Kp + 1;
      a.trans = m.dp->fwd_trans; a.emis = m.dp->fwd_emis; a.dsq = db_->d_dsq;
      a.nj = 0.0f; a.xf_e_move = 1.0f; a.xf_e_loop = 0.0f;              // p7_oprofile_ReconfigUnihit
      a.nenv = m.nenv;
      a.env_sq = d_env_sq + m.first; a.tr_off = d_tr_off + m.first; a.env_len = d_env_len + m.first; a.env_L = d_env_L + m.first;
      a.order = d_env_order + m.first; a.cursor = d_cursor + j;
      // slab_base counts slabs of THIS job's stride from the start of its own region of the workspace
      a.work = eb->work + slab_floats; a.work_stride = (int64_t) m.stride; a.Lmax = class_Lmax[m.C];
      a.nblocks = m.nblocks; a.slab_base = 0;
      slab_floats += (size_t) m.nblocks * env_waves(m.C) * m.stride;
      a.out_sc = reinterpret_cast<float *>(eb->d_out + o_sc) + 2 * m.first;
      a.out_null2 = reinterpret_cast<float *>(eb->d_out + o_n2) + 32 * m.first;
      a.out_status = reinterpret_cast<int32_t *>(eb->d_out + o_st) + m.first;
      a.out_orig = reinterpret_cast<float *>(eb->d_out + o_orig) + m.first;
      if (jobs[(size_t) j].lt_tables) {
        const size_t cnt = (size_t) m.nenv * jobs[(size_t) j].lt_stride;
        P7X_HIP(hipMemcpyAsync(eb->lt_dev + lt_at, jobs[(size_t) j].lt_tables, cnt * 4, hipMemcpyHostToDevice, s));
        a.env_emis = eb->lt_dev + lt_at; a.env_emis_stride = (long long) jobs[(size_t) j].lt_stride;
        lt_at += cnt;
      }
      a.oa_guard = oa_guard_;
      a.tr_n = reinterpret_cast<int32_t *>(eb->d_out + o_n) + m.first;
      a.tr_a = reinterpret_cast<uint32_t *>(eb->d_out + o_ta);
      a.tr_i = reinterpret_cast<int32_t *>(eb->d_out + o_ti);
      a.tr_pp = reinterpret_cast<float *>(eb->d_out + o_tp);
      if (!runs.empty() && h_args[runs.back().first].C == a.C && h_args[runs.back().first].nrows == a.nrows &&
          (h_args[runs.back().first].env_emis != nullptr) == (a.env_emis != nullptr)) runs.back().second++;
      else runs.emplace_back(nrec, 1);
      h_args[nrec++] = a;
    }
    tick("args");
    P7X_HIP(hipMemcpyAsync(eb->d_in, eb->h_in, in_bytes, hipMemcpyHostToDevice, s));
    tick("h2d");
    for (const auto &run : runs) {
      ArgRun<EnvArgs> ar;
      ar.host = h_args + run.first; ar.dev = eb->d_in + o_args + (size_t) run.first * sizeof(EnvArgs);
      ar.stride = (uint32_t) sizeof(EnvArgs); ar.n = run.second;
      if ((st = env_launch(ar, s)) != P7X_OK) return st;
    }
    tick("launches");
    P7X_HIP(hipMemcpyAsync(eb->h_out, eb->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    tick("d2h");
    if (debug) std::fprintf(stderr, "[env begin] jobs %zu envelopes %lld runs %zu work %.1f MB out %.1f MB:%s ms\n", nj, (long long) nenv_tot, runs.size(),
                            work_floats * 4 / 1e6, out_bytes / 1e6, dbg.c_str());
    eb_ = eb; o_sc_ = o_sc; o_n2_ = o_n2; o_st_ = o_st; o_n_ = o_n; o_orig_ = o_orig; o_ta_ = o_ta; o_ti_ = o_ti; o_tp_ = o_tp;
    tr_off_.assign(tr_off, tr_off + nenv_tot);
    return P7X_OK;
  }

  int wait(std::vector<std::vector<EnvelopeResult>> &res) override
  {
    res.assign(jobs_.size(), {});
    if (nenv_ == 0) return P7X_OK;
    P7X_HIP(hipSetDevice(db_->device));
    EnvBuffers *eb = eb_;
    P7X_HIP(hipStreamSynchronize(eb->stream));
    const float *h_sc = reinterpret_cast<const float *>(eb->h_out + o_sc_), *h_n2 = reinterpret_cast<const float *>(eb->h_out + o_n2_);
    const int32_t *h_st = reinterpret_cast<const int32_t *>(eb->h_out + o_st_), *h_n = reinterpret_cast<const int32_t *>(eb->h_out + o_n_);
    const uint32_t *h_ta = reinterpret_cast<const uint32_t *>(eb->h_out + o_ta_);
    const int32_t *h_ti = reinterpret_cast<const int32_t *>(eb->h_out + o_ti_);
    const float *h_tp = reinterpret_cast<const float *>(eb->h_out + o_tp_);
    for (size_t j = 0; j < jobs_.size(); ++j) {
      const JobMeta &m = meta_[j];
      res[j].assign((size_t) m.nenv, EnvelopeResult{});
      for (int r = 0; r < m.nenv; ++r) {
        const int64_t g = m.first + r;
        EnvelopeResult &e = res[j][(size_t) r];
        e.envsc = h_sc[2 * g]; e.oasc = h_sc[2 * g + 1]; e.status = h_st[g];
        e.orig = reinterpret_cast<const float *>(eb->h_out + o_orig_)[g];
        std::memcpy(e.null2, h_n2 + (size_t) g * 32, sizeof(e.null2));
        e.ntrace = h_n[g]; e.ta = h_ta + tr_off_[(size_t) g]; e.ti = h_ti + tr_off_[(size_t) g]; e.tp = h_tp + tr_off_[(size_t) g];
      }
    }
    return P7X_OK;
  }

  void set_oa_guard(float g) { oa_guard_ = g > 0.0f ? g : 0.0f; }

private:
  float oa_guard_ = 0.0f;
  struct JobMeta { int64_t first = 0; int nenv = 0, C = 0, Lmax = 1, nblocks = 1; size_t stride = 0; DevProfile *dp = nullptr; };
  DeviceCtx *ctx_; const p7x_seqdb *db_;
  std::vector<EnvelopeJob> jobs_;
  std::vector<JobMeta> meta_;
  std::vector<int> order_;
  std::vector<int64_t> tr_off_;
  int64_t nenv_ = 0;
  EnvBuffers *eb_ = nullptr;
  EnvBuffers *lease_ = nullptr;
  size_t o_sc_ = 0, o_n2_ = 0, o_st_ = 0, o_n_ = 0, o_orig_ = 0, o_ta_ = 0, o_ti_ = 0, o_tp_ = 0;
};

std::unique_ptr<EnvelopeScorer> make_device_envelope_scorer(DeviceCtx *ctx, const p7x_seqdb *db, float oa_guard)
{
  auto s = std::make_unique<DeviceEnvelopeScorer>(ctx, db);
  s->set_oa_guard(oa_guard);
  return s;
}

} // namespace p7x
