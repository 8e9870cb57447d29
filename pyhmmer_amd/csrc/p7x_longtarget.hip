// p7x_longtarget.hip -- device half of the long-target (nhmmer) search: the SSV scan of every target strand
// (p7x_ssvlong.hip) and the hand-over of the rows it reports to the host tail (p7x_longtarget.inc.hpp).
// Reference: LongTargetsPipeline._search_loop_longtargets, plan7.pyx:7541-7664; p7_Pipeline_LongTarget,
// p7_pipeline.pxd:131-143.
#include "p7x_device.hpp"
#include "p7x_kernels.hpp"
#include "p7x_host.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

namespace p7x {

namespace {

// a device buffer from the context's slab pool (hipMalloc / hipFree wait for every stream of the device: see p7x_seqdb_create)
struct DevBuf {
  DeviceCtx *ctx = nullptr; void *p = nullptr; size_t bytes = 0;
  explicit DevBuf(DeviceCtx *c) : ctx(c) {}
  DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) slab_release(ctx, p, bytes); }
  int alloc(size_t want) { if (p) { slab_release(ctx, p, bytes); p = nullptr; bytes = 0; } return slab_acquire(ctx, std::max<size_t>(want, 16), &p, &bytes); }
};
static hipStream_t lt_window_stream(DeviceCtx *ctx)
{
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int k = ctx->lt_next; ctx->lt_next = (k + 1) % DeviceCtx::kLtStreams;
  if (!ctx->lt_stream[k] && hipStreamCreateWithFlags(&ctx->lt_stream[k], hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); return ctx->stream; }
  return ctx->lt_stream[k];
}

struct ScanRow { int64_t pos; int strand, k, sc; };

// The device copy of a target set that its owner promised not to change (cfg.lt_resident_key): one per device, kept
// until a set with another key arrives.  Searches hold a reference while they run, so replacing the copy never pulls it
// from under a search still scanning it.
struct ResidentTargets {
  uint64_t key = 0; int device = -1; void *p = nullptr; size_t bytes = 0;
  ~ResidentTargets() { if (p) { (void) hipSetDevice(device); (void) hipFree(p); } }
};
static std::mutex g_resident_mu;
static std::map<int, std::shared_ptr<ResidentTargets>> &resident_by_device()
{ // never destroyed: at process exit the HIP runtime may be gone before a static destructor could free device memory
  static auto *m = new std::map<int, std::shared_ptr<ResidentTargets>>();
  return *m;
}

// Lets go of the kept copy of <device> (-1: every device) if it carries <key> (0: whatever it carries).  A search still
// scanning it holds its own reference: the memory goes when that search is done.
extern "C" int p7x_longtargets_release_resident(int device, uint64_t key)
{
  std::lock_guard<std::mutex> lk(g_resident_mu);
  auto &m = resident_by_device();
  for (auto it = m.begin(); it != m.end(); ) {
    if ((device < 0 || it->first == device) && (key == 0 || it->second->key == key)) it = m.erase(it);
    else ++it;
  }
  return P7X_OK;
}

// <seq1> (1-based, L residues) on the device: the kept copy when the key and the size match, else a fresh upload --
// kept in its place when there is a key.  *uploaded tells which it was (timing, tests).
static int resident_targets(DeviceCtx *ctx, int device, uint64_t key, const uint8_t *seq1, int64_t L, std::shared_ptr<ResidentTargets> &out, bool *uploaded)
{
  const size_t bytes = (size_t) L + 2;
  if (key != 0) {
    std::lock_guard<std::mutex> lk(g_resident_mu);
    auto it = resident_by_device().find(device);
    if (it != resident_by_device().end() && it->second->key == key && it->second->bytes == bytes) { out = it->second; *uploaded = false; return P7X_OK; }
  }
  auto r = std::make_shared<ResidentTargets>();
  r->key = key; r->device = device; r->bytes = bytes;
  P7X_HIP(hipMalloc(&r->p, std::max<size_t>(bytes, 16)));
  P7X_HIP(hipMemcpyAsync(static_cast<uint8_t *>(r->p) + 1, seq1 + 1, (size_t) L, hipMemcpyHostToDevice, ctx->stream));
  P7X_HIP(hipStreamSynchronize(ctx->stream));            // another search may pick the copy up as soon as it is published
  *uploaded = true;
  if (key != 0) {
    std::lock_guard<std::mutex> lk(g_resident_mu);
    resident_by_device()[device] = r;                      // the previous copy goes when its last search is done
  }
  out = std::move(r);
  return P7X_OK;
}

// The reset-free SSV scan of one target (both strands, or one): every row whose best diagonal reaches the threshold.
// <ms>: kernel time by HIP events.
// <ranges>, when given: only the chunks that touch one of these (strand, first, last) position ranges are scanned
// (a search dealt over several devices: every device scans the blocks of its own units).
struct ScanRange { int strand; int64_t first, last; };
static int scan_target(const p7x_pipeline_cfg &cfg, const Profile &p, DeviceCtx *ctx, const uint8_t *seq1 /* 1-based */, int64_t L,
                       int sc_thresh, int xB, int strands_mask, std::vector<ScanRow> &rows, double *ms, const std::vector<ScanRange> *ranges = nullptr,
                       int device = 0, uint64_t resident_key = 0)
{
  // The row maximum in every second row only (PAIR) tests against a threshold lowered by the most a cell can lose in one
  // row, and every group of rows that reaches the lowered threshold is run again: it pays while that loss is a few score
  // units (9 for bmyD, 12 for RF00001: 2^(loss/3) times as many groups are repeated as reach the threshold itself) and is
  // a loss for models with near-impossible emissions (60-110 units for the tests' Dirichlet models: every block would be
  // repeated).  Option ssv_kernel (tests, A/B): 3 = every row, 4 = every second row whatever the loss.
  constexpr int kMaxPairSlack = 16;
  int opt = debug_opt(OPT_SSV_KERNEL);
  const bool half = !(opt >= 5 && opt <= 7);       // 5, 6, 7: the int16 flavour of the kernel with the library's choice of rows / every row / every second row
  if (!half) opt = opt == 5 ? -1 : (opt == 6 ? 3 : 4);
  bool pair = opt != 3;
  int R = ssvlong_pick_R(p.M, pair);
  if (R < 0) { set_error("model too long for the long-target SSV kernel (M > 6141)"); return P7X_EINVAL; }
  std::vector<uint32_t> tab4q, tab_full;
  int pair_slack = 0;
  ssvlong_build_tables(p, R, pair, tab4q, tab_full, &pair_slack);
  if (pair && opt != 4 && pair_slack > kMaxPairSlack) {
    pair = false;
    R = ssvlong_pick_R(p.M, false);
    ssvlong_build_tables(p, R, false, tab4q, tab_full, &pair_slack);
  }
  std::lock_guard<std::mutex> scan_turn(ctx->lt_scan_mu);      // one scan at a time per device (see DeviceCtx)
  DevBuf d_tab4q{ctx}, d_full{ctx}, d_comp{ctx}, d_nrec{ctx}, d_pos{ctx}, d_strand{ctx}, d_k{ctx}, d_sc{ctx};
  int st;
  if ((st = d_tab4q.alloc(tab4q.size() * 4)) || (st = d_full.alloc(tab_full.size() * 4)) || (st = d_comp.alloc(32)) || (st = d_nrec.alloc(4))) return st;
  hipStream_t s = ctx->stream;
  std::shared_ptr<ResidentTargets> d_seq;
  bool uploaded = false;
  if ((st = resident_targets(ctx, device, resident_key, seq1, L, d_seq, &uploaded)) != P7X_OK) return st;
  if ((debug_opt(OPT_TRACE_LONGTARGET) > 0)) std::fprintf(stderr, "[lt] targets on the device: %s (%lld bytes, key %llu)\n", uploaded ? "uploaded" : "resident", (long long) L, (unsigned long long) resident_key);
  P7X_HIP(hipMemcpyAsync(d_tab4q.p, tab4q.data(), tab4q.size() * 4, hipMemcpyHostToDevice, s));
  P7X_HIP(hipMemcpyAsync(d_full.p, tab_full.data(), tab_full.size() * 4, hipMemcpyHostToDevice, s));
  P7X_HIP(hipMemcpyAsync(d_comp.p, longtarget_complement(p.abc_type), 18, hipMemcpyHostToDevice, s));
  // chunks: long enough that the M warm-up rows are a small overhead.  The kernel's wavefronts all stay on the device and
  // take the chunks in turn, so the number of chunks is made a multiple of the wavefronts (a last round that only some
  // of them take is paid in full: 16,384 chunks on 3,072 wavefronts were six rounds, the last a third full)
  const int nstrands_plan = strands_mask == 3 ? 2 : 1;
  long long waves = 0;
  if ((st = ssvlong_capacity(R, pair, half, ctx->num_cu, &waves)) != P7X_OK) return st;
  const int64_t rounds = std::max<int64_t>(1, (L * nstrands_plan + waves * 49152 - 1) / (waves * 49152));
  const int64_t want_chunks = (waves * rounds + nstrands_plan - 1) / nstrands_plan;             // per strand
  int chunk_len = (int) std::max<int64_t>(8 * (int64_t) p.M, std::min<int64_t>(1 << 16, (L + want_chunks - 1) / want_chunks));
  chunk_len = ((chunk_len + 63) / 64) * 64;
  const int nstrands = strands_mask == 3 ? 2 : 1;
  SsvLongArgs a{};
  a.tab4q = static_cast<const uint32_t *>(d_tab4q.p); a.tab_full = static_cast<const uint32_t *>(d_full.p);
  a.dsq = static_cast<const uint8_t *>(d_seq->p); a.comp = static_cast<const uint8_t *>(d_comp.p);
  a.L = L; a.M = p.M; a.Kp = p.Kp; a.chunk_len = chunk_len;
  a.chunks_per_strand = (L + chunk_len - 1) / chunk_len; a.nchunks = a.chunks_per_strand * nstrands;
  a.thresh_s = sc_thresh - xB - 32768; a.xB = xB; a.Q16 = p.Q16();
  a.nrec = static_cast<int *>(d_nrec.p);
  a.strand0 = strands_mask == 2 ? 1 : 0;
  a.pair_slack = pair_slack;
  DevBuf d_chunks{ctx};
  if (ranges) {
    std::vector<long long> list;
    for (const ScanRange &r : *ranges) {
      const long long s_off = (long long) (r.strand - a.strand0) * a.chunks_per_strand;
      for (long long c = (r.first - 1) / chunk_len; c <= (r.last - 1) / chunk_len && c < a.chunks_per_strand; ++c) list.push_back(s_off + c);
    }
    std::sort(list.begin(), list.end());
    list.erase(std::unique(list.begin(), list.end()), list.end());
    rows.clear();
    if (list.empty()) { if (ms) *ms = 0.0; return P7X_OK; }
    if ((st = d_chunks.alloc(list.size() * 8)) != P7X_OK) return st;
    P7X_HIP(hipMemcpyAsync(d_chunks.p, list.data(), list.size() * 8, hipMemcpyHostToDevice, s));
    P7X_HIP(hipStreamSynchronize(s));                       // <list> leaves scope
    a.chunk_list = static_cast<const long long *>(d_chunks.p); a.nchunks = (long long) list.size();
  }
  struct Events {                     // destroyed on every way out
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() { if (e0) (void) hipEventDestroy(e0); if (e1) (void) hipEventDestroy(e1); }
  } ev;
  P7X_HIP(hipEventCreate(&ev.e0)); P7X_HIP(hipEventCreate(&ev.e1));
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  int cap = (int) std::min<int64_t>(std::max<int64_t>(1 << 16, L / 64), 1 << 28);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if ((st = d_pos.alloc((size_t) cap * 8)) || (st = d_strand.alloc((size_t) cap)) || (st = d_k.alloc((size_t) cap * 4)) || (st = d_sc.alloc((size_t) cap * 4))) return st;
    a.rec_pos = static_cast<long long *>(d_pos.p); a.rec_strand = static_cast<uint8_t *>(d_strand.p);
    a.rec_k = static_cast<int *>(d_k.p); a.rec_sc = static_cast<int *>(d_sc.p); a.rec_cap = cap;
    P7X_HIP(hipMemsetAsync(d_nrec.p, 0, 4, s));
    P7X_HIP(hipEventRecord(e0, s));
    if ((st = ssvlong_launch(R, pair, half, a, ctx->num_cu, s)) != P7X_OK) return st;
    P7X_HIP(hipEventRecord(e1, s));
    int nrec = 0;
    P7X_HIP(hipMemcpyAsync(&nrec, d_nrec.p, 4, hipMemcpyDeviceToHost, s));
    P7X_HIP(hipStreamSynchronize(s));
    float t = 0; (void) hipEventElapsedTime(&t, e0, e1);
    if (ms) *ms = t;
    if (nrec <= cap) {
      std::vector<long long> pos((size_t) nrec); std::vector<uint8_t> strand((size_t) nrec); std::vector<int> k((size_t) nrec), sc((size_t) nrec);
      if (nrec) {
        P7X_HIP(hipMemcpy(pos.data(), d_pos.p, (size_t) nrec * 8, hipMemcpyDeviceToHost));
        P7X_HIP(hipMemcpy(strand.data(), d_strand.p, (size_t) nrec, hipMemcpyDeviceToHost));
        P7X_HIP(hipMemcpy(k.data(), d_k.p, (size_t) nrec * 4, hipMemcpyDeviceToHost));
        P7X_HIP(hipMemcpy(sc.data(), d_sc.p, (size_t) nrec * 4, hipMemcpyDeviceToHost));
      }
      rows.resize((size_t) nrec);
      for (int i = 0; i < nrec; ++i) rows[(size_t) i] = ScanRow{ pos[(size_t) i], strand[(size_t) i], k[(size_t) i], sc[(size_t) i] };
      std::sort(rows.begin(), rows.end(), [](const ScanRow &x, const ScanRow &y) { return x.strand != y.strand ? x.strand < y.strand : x.pos < y.pos; });
      return P7X_OK;
    }
    cap = nrec + 1024;              // more rows than the buffer holds: once more with room for all of them
  }
  set_error("long-target SSV scan: record buffer could not be sized");
  return P7X_EMEM;
}

// seeds of one block of one strand from the rows of the whole-strand scan
// <shift>: what the scan's row positions are ahead of the target's own strand positions (a scan over several targets
// laid end to end)
static void block_seeds(const Profile &p, const uint8_t *seq1, int64_t Lt, int64_t i, int64_t bn, int strand, const std::vector<ScanRow> &rows,
                        int sc_thresh, int xB, std::vector<int64_t> &seeds3, int64_t shift = 0)
{
  // block rows 1..bn: strand 0: original positions i+1 .. i+bn; strand 1: reverse-strand positions (Lt-i-bn)+1 .. (Lt-i-bn)+bn
  const int64_t base = (strand == 0 ? i : Lt - i - bn) + shift;
  std::vector<LongTargetRow> br;
  auto lo = std::lower_bound(rows.begin(), rows.end(), ScanRow{ base + 1, strand, 0, 0 },
                             [](const ScanRow &x, const ScanRow &y) { return x.strand != y.strand ? x.strand < y.strand : x.pos < y.pos; });
  for (auto it = lo; it != rows.end() && it->strand == strand && it->pos <= base + bn; ++it) br.push_back(LongTargetRow{ it->pos - base, it->k, it->sc });
  if (br.empty()) return;
  const uint8_t *comp = longtarget_complement(p.abc_type);
  std::vector<uint8_t> buf((size_t) bn + 2, 255);
  if (strand == 0) std::memcpy(buf.data() + 1, seq1 + i + 1, (size_t) bn);
  else for (int64_t q = 1; q <= bn; ++q) buf[(size_t) q] = comp[seq1[i + bn - q + 1]];
  longtarget_seeds_from_rows(p, buf.data(), bn, br.data(), br.size(), sc_thresh, xB, seeds3);
}

// The windows of a target as one block of "targets" through the batch filter kernels: exact MSV scores (u8), the bias
// filter score, the standard Viterbi filter score (an upper bound for every row of the long-target Viterbi scan), and
// then the long-target Viterbi scan itself (vit_kernel with per-window row thresholds) for the windows that need it.
struct DeviceWindowScorer final : LongTargetWindowScorer {
  const p7x_oprofile *om; int device;
  p7x_seqdb *db = nullptr;
  p7x_seqdb *rdb = nullptr;             // the windows of the last regions() call: envelopes() refers to them
  float oa_guard = 0.0f;
  double ms = 0.0; size_t nwindows = 0;
  hipStream_t stream = nullptr;         // this search's own, for the long-target Viterbi scan of its windows
  DeviceWindowScorer(const p7x_oprofile *o, int d, float guard) : om(o), device(d), oa_guard(guard) {}
  ~DeviceWindowScorer() override { if (db) p7x_seqdb_destroy(db); if (rdb) p7x_seqdb_destroy(rdb); }
  int score(const uint8_t *seq1, int64_t L, const uint8_t *comp, const LongTargetWindowRef *w, size_t nw, double F1, bool do_bias,
            LongTargetWindowScore *sc) override
  {
    (void) L; (void) F1;
    const auto t0 = std::chrono::steady_clock::now();
    const Profile &p = om->p;
    size_t tot = 1;
    for (size_t q = 0; q < nw; ++q) tot += (size_t) w[q].length + 1;
    std::vector<uint8_t> dsq(tot, 255);
    std::vector<int64_t> off(nw); std::vector<int32_t> len(nw);
    size_t pos = 1;
    for (size_t q = 0; q < nw; ++q) {
      off[q] = (int64_t) pos; len[q] = (int32_t) w[q].length;
      if (w[q].strand == 0) std::memcpy(dsq.data() + pos, seq1 + w[q].start, (size_t) w[q].length);
      else for (int64_t r = 0; r < w[q].length; ++r) dsq[pos + (size_t) r] = comp[seq1[w[q].start - r]];
      pos += (size_t) w[q].length + 1;
    }
    if (db) { p7x_seqdb_destroy(db); db = nullptr; }
    int st = p7x_seqdb_create(device, p.abc_type, dsq.data(), off.data(), len.data(), nw, &db);
    if (st != P7X_OK) return st;
    std::vector<int32_t> xJ(nw), xC(nw); std::vector<float> bias(nw);
    st = p7x_filters_batch(om, db, xJ.data(), xC.data(), nullptr, do_bias ? bias.data() : nullptr);
    if (st != P7X_OK) return st;
    for (size_t q = 0; q < nw; ++q) {
      const int Lw = len[q];
      const float pm = logf(3.0f / (float) (Lw + 3));
      float usc;
      if (xJ[q] < 0) usc = INFINITY;
      else { const uint8_t tjb = unbiased_byteify(p.scale_b, pm); usc = ((float) (xJ[q] - tjb) - (float) p.base_b); usc /= p.scale_b; usc -= 3.0f; }
      float vf;
      if (xC[q] >= 32767) vf = INFINITY;
      else if (xC[q] > -32768) { vf = (float) xC[q] + (float) wordify(p.scale_w, pm) - (float) p.base_w; vf /= p.scale_w; vf -= 3.0f; }
      else vf = -INFINITY;
      sc[q].usc = usc; sc[q].bias_filtersc = do_bias ? bias[q] : 0.0f; sc[q].vfsc = vf; sc[q].have_vit = 1;
    }
    ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    nwindows += nw;
    return P7X_OK;
  }

  static void pack_windows(const uint8_t *seq1, const uint8_t *comp, const LongTargetWindowRef *w, size_t nw, std::vector<uint8_t> &dsq,
                           std::vector<int64_t> &off, std::vector<int32_t> &len)
  {
    size_t tot = 1;
    for (size_t q = 0; q < nw; ++q) tot += (size_t) w[q].length + 1;
    dsq.assign(tot, 255); off.resize(nw); len.resize(nw);
    size_t pos = 1;
    for (size_t q = 0; q < nw; ++q) {
      off[q] = (int64_t) pos; len[q] = (int32_t) w[q].length;
      if (w[q].strand == 0) std::memcpy(dsq.data() + pos, seq1 + w[q].start, (size_t) w[q].length);
      else for (int64_t r = 0; r < w[q].length; ++r) dsq[pos + (size_t) r] = comp[seq1[w[q].start - r]];
      pos += (size_t) w[q].length + 1;
    }
  }

  int forward(const uint8_t *seq1, const uint8_t *comp, const LongTargetWindowRef *w, size_t nw, float *fwdsc) override
  {
    if (nw == 0) return P7X_OK;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> dsq; std::vector<int64_t> off; std::vector<int32_t> len;
    pack_windows(seq1, comp, w, nw, dsq, off, len);
    p7x_seqdb *fdb = nullptr;
    int st = p7x_seqdb_create(device, om->p.abc_type, dsq.data(), off.data(), len.data(), nw, &fdb);
    if (st != P7X_OK) return st;
    st = p7x_filters_batch(om, fdb, nullptr, nullptr, fwdsc, nullptr);
    p7x_seqdb_destroy(fdb);
    ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
  }

  int regions(const uint8_t *seq1, const uint8_t *comp, const LongTargetWindowRef *w, size_t nw, std::vector<LongTargetWindowRegions> &out) override
  {
    out.assign(nw, LongTargetWindowRegions{});
    if (nw == 0) return P7X_OK;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> dsq; std::vector<int64_t> off; std::vector<int32_t> len;
    pack_windows(seq1, comp, w, nw, dsq, off, len);
    if (rdb) { p7x_seqdb_destroy(rdb); rdb = nullptr; }
    int st = p7x_seqdb_create(device, om->p.abc_type, dsq.data(), off.data(), len.data(), nw, &rdb);
    if (st != P7X_OK) return st;
    st = device_regions_of_all(om, rdb, out);
    ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
  }

  int envelopes(const LongTargetEnvRequest *req, size_t n, std::vector<LongTargetEnvResult> &out) override
  {
    out.assign(n, LongTargetEnvResult{});
    if (n == 0) return P7X_OK;
    if (!rdb) { set_error("long-target envelopes before the windows' region scan"); return P7X_EINVAL; }
    const auto t0 = std::chrono::steady_clock::now();
    const Profile &p = om->p;
    DeviceCtx *ctx = nullptr; DevProfile *dp = nullptr;
    int st;
    if ((st = get_ctx(device, &ctx)) != P7X_OK || (st = get_dev_profile(om, ctx, &dp)) != P7X_OK) return st;
    const int C = dp->vitC, Mpad = 64 * C, nrows = p.Kp + 1;
    if (C <= 0) { set_error("model too long for the envelope kernel"); return P7X_EINVAL; }
    // the envelopes' match odds in the kernel's table layout: [row x][position of node k], node k of lane z, slot c at c * 64 + z
    const size_t stride = (size_t) nrows * Mpad;
    std::vector<float> tables(n * stride, 0.0f);
    host_parallel_for((int) n, 0, [&](int e) {
      float *t = tables.data() + (size_t) e * stride;
      const float *rf = req[(size_t) e].rf;
      for (int x = 0; x < p.Kp; ++x)
        for (int k = 1; k <= p.M; ++k) t[(size_t) x * Mpad + ((k - 1) % C) * 64 + (k - 1) / C] = rf[(size_t) x * (p.M + 1) + k];
    });
    std::vector<EnvelopeRequest> rq(n);
    std::vector<int32_t> targets((size_t) rdb->n);
    for (int64_t t = 0; t < rdb->n; ++t) targets[(size_t) t] = (int32_t) t;
    for (size_t e = 0; e < n; ++e) rq[e] = EnvelopeRequest{ req[e].window, req[e].i, req[e].j };
    std::unique_ptr<EnvelopeScorer> scorer = make_device_envelope_scorer(ctx, rdb, oa_guard);
    std::vector<EnvelopeJob> jobs(1);
    jobs[0].om = om; jobs[0].req = &rq; jobs[0].targets = &targets; jobs[0].lt_tables = tables.data(); jobs[0].lt_stride = stride;
    if ((st = scorer->begin(jobs)) != P7X_OK) return st;
    std::vector<std::vector<EnvelopeResult>> res;
    if ((st = scorer->wait(res)) != P7X_OK) return st;
    for (size_t e = 0; e < n; ++e) {
      const EnvelopeResult &r = res[0][e];
      LongTargetEnvResult &o = out[e];
      o.envsc = r.envsc; o.oasc = r.oasc; o.orig = r.orig; o.status = r.status;
      if (r.ntrace > 0) { o.ta.assign(r.ta, r.ta + r.ntrace); o.ti.assign(r.ti, r.ti + r.ntrace); o.tp.assign(r.tp, r.tp + r.ntrace); }
    }
    ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return P7X_OK;
  }

  int viterbi(const int *which, const int *thresh, size_t n, std::vector<int> &rec) override
  {
    rec.clear();
    if (n == 0 || !db) return P7X_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const Profile &p = om->p;
    DeviceCtx *ctx = nullptr; DevProfile *dp = nullptr;
    int st;
    if ((st = get_ctx(device, &ctx)) != P7X_OK || (st = get_dev_profile(om, ctx, &dp)) != P7X_OK) return st;
    if (dp->vitC <= 0) { set_error("model too long for the wave-per-target Viterbi kernel"); return P7X_EINVAL; }
    // caller index -> slot of the window block (slots are sorted by decreasing length)
    std::vector<int32_t> slot_of((size_t) db->n, -1);
    for (int64_t sl = 0; sl < db->nslots; ++sl) slot_of[(size_t) db->h_order[(size_t) sl]] = (int32_t) sl;
    std::vector<int32_t> list(n);
    for (size_t i = 0; i < n; ++i) list[i] = slot_of[(size_t) which[i]];
    hipStream_t s = stream ? stream : (stream = lt_window_stream(ctx));
    DevBuf d_list{ctx}, d_thr{ctx}, d_xc{ctx}, d_nrec{ctx}, d_rec{ctx}, d_args{ctx};
    if ((st = d_list.alloc(n * 4)) || (st = d_thr.alloc(n * 4)) || (st = d_xc.alloc(n * 4)) || (st = d_nrec.alloc(4)) || (st = d_args.alloc(sizeof(WaveSeqArgs)))) return st;
    P7X_HIP(hipMemcpyAsync(d_list.p, list.data(), n * 4, hipMemcpyHostToDevice, s));
    P7X_HIP(hipMemcpyAsync(d_thr.p, thresh, n * 4, hipMemcpyHostToDevice, s));
    int cap = (int) std::max<size_t>(1 << 16, 64 * n);
    for (int attempt = 0; attempt < 2; ++attempt) {
      if ((st = d_rec.alloc((size_t) cap * 12)) != P7X_OK) return st;
      WaveSeqArgs a{};
      a.M = p.M; a.C = dp->vitC; a.nrows = p.Kp + 1;
      a.trans = dp->vit_trans; a.emis = dp->vit_emis;
      a.dsq = db->d_dsq; a.slot_off = db->d_slot_off; a.slot_len = db->d_slot_len;
      a.list = static_cast<const int32_t *>(d_list.p); a.nlist = (int) n;
      a.xwmove_tab = ctx->lt.xwmove; a.base_w = p.base_w; a.xw_e = p.xw[XE][MOVE]; a.ddbound = p.ddbound_w;
      a.out_xC = static_cast<int32_t *>(d_xc.p);
      a.lt_thresh = static_cast<const int *>(d_thr.p); a.lt_nrec = static_cast<int *>(d_nrec.p); a.lt_rec = static_cast<int *>(d_rec.p); a.lt_cap = cap;
      P7X_HIP(hipMemcpyAsync(d_args.p, &a, sizeof(a), hipMemcpyHostToDevice, s));
      P7X_HIP(hipMemsetAsync(d_nrec.p, 0, 4, s));
      ArgRun<WaveSeqArgs> run; run.host = &a; run.dev = d_args.p; run.stride = (uint32_t) sizeof(WaveSeqArgs); run.n = 1;
      if ((st = vit_launch(run, ctx->num_cu, s)) != P7X_OK) return st;
      int nrec = 0;
      P7X_HIP(hipMemcpyAsync(&nrec, d_nrec.p, 4, hipMemcpyDeviceToHost, s));
      P7X_HIP(hipStreamSynchronize(s));
      if (nrec <= cap) {
        rec.resize((size_t) nrec * 3);
        if (nrec) P7X_HIP(hipMemcpy(rec.data(), d_rec.p, (size_t) nrec * 12, hipMemcpyDeviceToHost));
        // (item, row, node) ascending, as the host scan emits them
        std::vector<int> ord((size_t) nrec);
        for (int i = 0; i < nrec; ++i) ord[(size_t) i] = i;
        std::sort(ord.begin(), ord.end(), [&](int x, int y) {
          for (int f = 0; f < 3; ++f) if (rec[(size_t) x * 3 + f] != rec[(size_t) y * 3 + f]) return rec[(size_t) x * 3 + f] < rec[(size_t) y * 3 + f];
          return false;
        });
        std::vector<int> sorted((size_t) nrec * 3);
        for (int i = 0; i < nrec; ++i) for (int f = 0; f < 3; ++f) sorted[(size_t) i * 3 + f] = rec[(size_t) ord[(size_t) i] * 3 + f];
        rec.swap(sorted);
        ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return P7X_OK;
      }
      cap = nrec + 1024;
    }
    set_error("long-target Viterbi scan: record buffer could not be sized");
    return P7X_EMEM;
  }
};

} // namespace
} // namespace p7x

using namespace p7x;

extern "C" {

int p7x_search_longtargets(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, int device,
                           const uint8_t *dsq, const int64_t *offsets, const int64_t *lengths, size_t n,
                           const char *const *names, const char *const *accs, const char *const *descs, p7x_tophits **out)
{
  if (!cfg || !om || !out || (n && (!dsq || !offsets || !lengths))) { set_error("p7x_search_longtargets: bad arguments"); return P7X_EINVAL; }
  if (!cfg->long_targets) { set_error("p7x_search_longtargets: cfg.long_targets is not set"); return P7X_EINVAL; }
  if (cfg->lt_nparts > 1 && (cfg->lt_part < 0 || cfg->lt_part >= cfg->lt_nparts)) { set_error("p7x_search_longtargets: cfg.lt_part is not one of the cfg.lt_nparts parts"); return P7X_EINVAL; }
  const Profile &p = om->p;
  int max_length = 0, sc_thresh = 0, xB = 0;
  int st = longtarget_setup(*cfg, p, &max_length, &sc_thresh, &xB);
  if (st != P7X_OK) return st;
  DeviceCtx *ctx = nullptr;
  if ((st = get_ctx(device, &ctx)) != P7X_OK) return st;
  const int mask = cfg->strands == P7X_STRAND_TOPONLY ? 1 : (cfg->strands == P7X_STRAND_BOTTOMONLY ? 2 : 3);
  const int64_t W = cfg->block_length, C = max_length;
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<LongTargetSeed> seeds;
  double scan_ms = 0.0;
  // One scan for the whole target set: the records lie end to end in <dsq> with a sentinel between them, which the scan
  // kernels take as the end of every diagonal, so a file of 1e5 contigs costs one launch, one upload and one set of
  // buffers like a single chromosome does.  Scan position q of strand 0 is dsq[q]; of strand 1 it is dsq[Ltot - q + 1]
  // (the whole buffer read backwards: the last target's reverse strand comes first).
  std::vector<uint8_t> packed;                    // only when the caller's records do not lie end to end
  std::vector<int64_t> poff;
  const int64_t *off = offsets;
  const uint8_t *base = dsq;
  {
    bool contiguous = n == 0 || offsets[0] >= 1;
    for (size_t t = 0; contiguous && t + 1 < n; ++t) contiguous = offsets[t + 1] == offsets[t] + lengths[t] + 1;
    for (size_t t = 0; contiguous && t + 1 < n; ++t) contiguous = dsq[offsets[t] + lengths[t]] >= (uint8_t) p.Kp;
    if (!contiguous) {
      poff.resize(n);
      int64_t at = 1;
      for (size_t t = 0; t < n; ++t) { poff[t] = at; at += std::max<int64_t>(lengths[t], 0) + 1; }
      packed.assign((size_t) at + 1, 255);
      for (size_t t = 0; t < n; ++t) if (lengths[t] > 0) std::memcpy(packed.data() + poff[t], dsq + offsets[t], (size_t) lengths[t]);
      off = poff.data(); base = packed.data();
    }
  }
  const int64_t first_at = n ? off[0] : 1;                                  // the scan starts at the first record
  const int64_t Ltot = n ? off[n - 1] + std::max<int64_t>(lengths[n - 1], 0) - first_at : 0;   // scan positions 1..Ltot
  const uint8_t *scan1 = base + first_at - 1;                               // scan1[q] = position q of strand 0
  LongTargetUnits all_units; all_units.count(*cfg, max_length, lengths, n);
  uint64_t unit = 0;
  // upstream's bookkeeping per (target, block, strand): the units are independent, the host workers take them side by
  // side; with the search dealt over several devices (cfg.lt_nparts) this call only has the units of its part
  struct Unit { size_t t; int64_t i, bn; int strand; int64_t shift; std::vector<int64_t> s3; };
  std::vector<Unit> units;
  std::vector<ScanRange> ranges;
  for (size_t t = 0; t < n; ++t) {
    const int64_t Lt = lengths[t];
    if (Lt <= 0) continue;
    const int64_t at = off[t] - first_at + 1;                 // scan position of the target's first residue on strand 0
    for (int64_t i = 0; i < Lt; i += W - C) {
      const int64_t bc = i == 0 ? 0 : std::min<int64_t>(C, Lt - i);
      const int64_t bw = std::min<int64_t>(W, Lt - i - bc);
      const int64_t bn = bc + bw;
      if (bn <= 0) break;
      for (int strand = 0; strand < 2; ++strand) if (mask & (1 << strand)) {
        if (!all_units.mine(unit++)) continue;
        // the target's strand position ps is scan position ps + shift
        const int64_t shift = strand == 0 ? at - 1 : Ltot - (at + Lt - 1);
        units.push_back(Unit{ t, i, bn, strand, shift, {} });
        const int64_t b0 = (strand == 0 ? i : Lt - i - bn) + shift;  // the block's rows in scan coordinates: b0 + 1 .. b0 + bn
        ranges.push_back(ScanRange{ strand, b0 + 1, b0 + bn });
      }
    }
  }
  if (!units.empty() && Ltot > 0) {
    std::vector<ScanRow> rows;
    double ms = 0.0;
    const auto ts0 = std::chrono::steady_clock::now();
    // the kept copy is of the caller's buffer as it lies: a set that had to be packed first is uploaded per call
    const uint64_t rkey = packed.empty() ? cfg->lt_resident_key : 0;
    if ((st = scan_target(*cfg, p, ctx, scan1, Ltot, sc_thresh, xB, mask, rows, &ms, all_units.nparts > 1 ? &ranges : nullptr, device, rkey)) != P7X_OK) return st;
    scan_ms += ms;
    const auto ts1 = std::chrono::steady_clock::now();
    host_parallel_for((int) units.size(), cfg->host_threads, [&](int u) {
      Unit &un = units[(size_t) u];
      block_seeds(p, base + off[un.t] - 1, lengths[un.t], un.i, un.bn, un.strand, rows, sc_thresh, xB, un.s3, un.shift);
    });
    for (const Unit &un : units)
      for (size_t q = 0; q + 2 < un.s3.size(); q += 3) seeds.push_back(LongTargetSeed{ (int64_t) un.t, un.i, un.strand, un.s3[q], (int) un.s3[q + 1], un.s3[q + 2] });
    if ((debug_opt(OPT_TRACE_LONGTARGET) > 0))
      std::fprintf(stderr, "[lt] %zu targets, %lld positions: scan call %.1f ms (kernel %.1f), %zu rows, %zu seeds, seed bookkeeping %.1f ms\n", n,
                   (long long) Ltot, std::chrono::duration<double, std::milli>(ts1 - ts0).count(), ms, rows.size(), seeds.size(),
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts1).count());
  }
  const auto t_host = std::chrono::steady_clock::now();
  DeviceWindowScorer scorer(om, device, cfg->oa_guard);
  st = longtarget_run_host(*cfg, om, dsq, offsets, lengths, n, names, accs, descs, seeds, out, &scorer);
  if (st == P7X_OK && *out) {
    (*out)->ms[7] = scan_ms;                                   // the SSV scan kernels (HIP events)
    (*out)->ms[0] = std::chrono::duration<double, std::milli>(t_host - t_begin).count();      // scan + seeds, wall
    (*out)->ms[1] = scorer.ms;                                 // window MSV / bias / Viterbi batch on the device, wall
    (*out)->ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count();   // host tail incl. [1]
  }
  return st;
}

int64_t p7x_ssv_longtarget_seeds(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, int device, const uint8_t *dsq, int64_t L,
                                 int complement, int64_t *seeds, size_t cap)
{
  if (!cfg || !om || !dsq || L <= 0 || (cap && !seeds)) { set_error("p7x_ssv_longtarget_seeds: bad arguments"); return -P7X_EINVAL; }
  const Profile &p = om->p;
  int max_length = 0, sc_thresh = 0, xB = 0;
  int st = longtarget_setup(*cfg, p, &max_length, &sc_thresh, &xB);
  if (st != P7X_OK) return -st;
  DeviceCtx *ctx = nullptr;
  if ((st = get_ctx(device, &ctx)) != P7X_OK) return -st;
  std::vector<ScanRow> rows;
  if ((st = scan_target(*cfg, p, ctx, dsq - 1, L, sc_thresh, xB, complement ? 2 : 1, rows, nullptr, nullptr, device, 0)) != P7X_OK) return -st;
  std::vector<int64_t> s3;
  block_seeds(p, dsq - 1, L, 0, L, complement ? 1 : 0, rows, sc_thresh, xB, s3);
  const size_t ns = s3.size() / 3;
  for (size_t q = 0; q < ns && q < cap; ++q) { seeds[3 * q] = s3[3 * q]; seeds[3 * q + 1] = s3[3 * q + 1]; seeds[3 * q + 2] = s3[3 * q + 2]; }
  return (int64_t) ns;
}

int64_t p7x_debug_ssv_tables(const p7x_oprofile *om, int pair, int32_t *R, int32_t *pair_slack, uint32_t *tab4q, size_t cap_words)
{
  if (!om) { set_error("p7x_debug_ssv_tables: no profile"); return -P7X_EINVAL; }
  const Profile &p = om->p;
  const int r = ssvlong_pick_R(p.M, pair != 0);
  if (r < 0) { set_error("model too long for the long-target SSV kernel (M > 6141)"); return -P7X_EINVAL; }
  std::vector<uint32_t> quads, full;
  int slack = 0;
  ssvlong_build_tables(p, r, pair != 0, quads, full, &slack);
  if (R) *R = r;
  if (pair_slack) *pair_slack = slack;
  if (tab4q && cap_words >= quads.size()) std::copy(quads.begin(), quads.end(), tab4q);
  return (int64_t) quads.size();
}

} // extern "C"
