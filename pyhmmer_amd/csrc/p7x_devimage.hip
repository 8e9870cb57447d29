// p7x_devimage.hip -- the device image of a query profile, and the pools it is cut from.
//
// A query's tables (MSV parity tables, wave-per-target MSV emissions, Viterbi / packed Viterbi / Forward transitions
// and emissions, bias-filter odds) are laid out back to back in one device slab and uploaded with one copy.  Slabs,
// staging buffers and pinned blocks come from process-wide pools that are never handed back to the runtime: a scan
// walks through thousands of short-lived profiles, and hipMalloc / hipFree / hipHostFree per profile (or from a
// thread that is winding down) would serialise the host against the whole device.
#include "p7x_device.hpp"
#include "p7x_kernels.hpp"
#include "p7x_host.hpp"
#include <cstring>
#include <memory>

namespace p7x {

// ---------------------------------------------------------------------------- device profile image
static void chunk_transpose_fill(int M, int C, std::vector<int> &pos_of_node)
{ // node k (1..M) -> table position c*64 + z with k = z*C + c + 1
  pos_of_node.assign(M + 1, -1);
  for (int k = 1; k <= M; ++k) { const int z = (k - 1) / C, c = (k - 1) % C; pos_of_node[k] = c * 64 + z; }
}

// Pinned host blocks are recycled through a process-wide pool and never handed back to the runtime: hipHostFree from a
// thread that is winding down (thread_local workspaces) while other threads drive the device is not something the
// runtime tolerates reliably, and the blocks are small.
struct PinnedPool { std::mutex mu; std::multimap<size_t, void *> free; };
static PinnedPool &pinned_pool() { static PinnedPool *p = new PinnedPool(); return *p; }      // never destroyed
int pinned_acquire(size_t bytes, void **out, size_t *got)
{
  size_t want = 256;
  while (want < bytes) want *= 2;
  {
    PinnedPool &pp = pinned_pool();
    std::lock_guard<std::mutex> lk(pp.mu);
    auto it = pp.free.find(want);
    if (it != pp.free.end()) { *out = it->second; *got = want; pp.free.erase(it); return P7X_OK; }
  }
  P7X_HIP(hipHostMalloc(out, want, hipHostMallocDefault));
  *got = want;
  return P7X_OK;
}
void pinned_release(void *p, size_t bytes)
{
  if (!p) return;
  PinnedPool &pp = pinned_pool();
  std::lock_guard<std::mutex> lk(pp.mu);
  pp.free.emplace(bytes, p);
}

int slab_acquire(DeviceCtx *ctx, size_t bytes, void **out, size_t *got)
{
  const size_t want = ((bytes + 65535) / 65536) * 65536;
  {
    std::lock_guard<std::mutex> lk(ctx->slab_mu);
    auto it = ctx->slab_free.lower_bound(want);
    if (it != ctx->slab_free.end() && it->first <= want * 2) {
      *out = it->second; *got = it->first; ctx->slab_free_bytes -= it->first; ctx->slab_free.erase(it);
      return P7X_OK;
    }
  }
  if (hipMalloc(out, want) != hipSuccess) {
    // the device is full while slabs sit parked: hand the parked ones back and try once more (a pool sized for
    // 288 GiB must not run a smaller device -- or one shared with other processes -- out of memory)
    (void) hipGetLastError();
    std::vector<void *> parked;
    {
      std::lock_guard<std::mutex> lk(ctx->slab_mu);
      for (auto &kv : ctx->slab_free) parked.push_back(kv.second);
      ctx->slab_free.clear(); ctx->slab_free_bytes = 0;
    }
    for (void *q : parked) (void) hipFree(q);
    P7X_HIP(hipMalloc(out, want));
  }
  *got = want;
  return P7X_OK;
}

void slab_release(DeviceCtx *ctx, void *p, size_t bytes)
{
  if (!p) return;
  std::lock_guard<std::mutex> lk(ctx->slab_mu);
  // keep at most 32 GiB parked (of 288), and never more than an eighth of the device's memory
  size_t cap = (size_t) 32 << 30, free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 8 < cap) cap = total_b / 8;
  if (ctx->slab_free_bytes + bytes > cap) { (void) hipFree(p); return; }
  ctx->slab_free.emplace(bytes, p); ctx->slab_free_bytes += bytes;
}

SlabRef::~SlabRef() { if (ctx && p) slab_release(ctx, p, bytes); }

void free_dev_profile(DevProfile *d)
{
  delete d;          // the slab goes back to the pool with the last image cut from it (SlabRef)
}

struct DevCache { std::mutex mu; std::vector<DevProfile *> per_device; };

// Host-side staging of a device image: the tables are laid out back to back (256-byte aligned) in one pinned buffer
// and go up with one copy on a stream of the building thread.
struct ImageStage {
  char *pinned = nullptr; size_t cap = 0; hipStream_t stream = nullptr; int device = -1;
  ~ImageStage() { pinned_release(pinned, cap); if (stream) (void) hipStreamDestroy(stream); }
  int reserve(size_t bytes)
  {
    if (bytes <= cap) return P7X_OK;
    pinned_release(pinned, cap);
    pinned = nullptr; cap = 0;
    void *p = nullptr; size_t got = 0;
    const int st = pinned_acquire(std::max<size_t>(bytes, (size_t) 1 << 20), &p, &got);
    if (st != P7X_OK) return st;
    pinned = static_cast<char *>(p); cap = got;
    return P7X_OK;
  }
};
struct StagePool { std::mutex mu; std::vector<ImageStage *> idle; };
static StagePool &stage_pool() { static StagePool *p = new StagePool(); return *p; }      // never destroyed
struct StageLease {
  ImageStage *st = nullptr;
  StageLease()
  {
    StagePool &sp = stage_pool();
    { std::lock_guard<std::mutex> lk(sp.mu); if (!sp.idle.empty()) { st = sp.idle.back(); sp.idle.pop_back(); } }
    if (!st) st = new ImageStage();
  }
  ~StageLease() { std::lock_guard<std::mutex> lk(stage_pool().mu); stage_pool().idle.push_back(st); }
};

// Host side of one image: the tables back to back (256-byte aligned) and, for every table, the DevProfile field that
// will point at it.
struct HostImage {
  std::vector<char> bytes;
  std::vector<std::pair<void **, size_t>> fix;      // pointer field in the DevProfile <- offset in <bytes>
  template <typename T, typename F>
  void add(F **field, const std::vector<T> &v)
  {
    const size_t off = bytes.size(), n = v.size() * sizeof(T);
    bytes.resize(((off + n + 255) / 256) * 256);
    std::memcpy(bytes.data() + off, v.data(), n);
    fix.emplace_back(reinterpret_cast<void **>(field), off);
  }
};

// All the tables of one profile, laid out on the host (pure CPU work: the batch entry point runs it on the workers).
static void build_host_image(const Profile &p, DevProfile *d, HostImage &img)
{
  // MSV parity tables
  d->msvR = msv_pick(p.M, &d->msvK);
  if (d->msvR > 0) {
    d->msvS = msv_stride(d->msvR, d->msvK);
    std::vector<uint32_t> tab;
    msv_build_tables(p, d->msvR, d->msvK, tab);
    img.add(&d->msv_tab, tab);
  }
  // wave-per-sequence tables
  d->vitC = vit_pick_C(p.M);
  if (d->vitC > 0) {
    const int C = d->vitC, Mpad = 64 * C, nrows = p.Kp + 1;
    d->Mpad = Mpad;
    std::vector<int> pos;
    chunk_transpose_fill(p.M, C, pos);
    std::vector<int16_t> vt((size_t) Mpad * 8, -32768), ve((size_t) nrows * Mpad, -32768);
    std::vector<float> ft((size_t) Mpad * 8, 0.0f), fe((size_t) nrows * Mpad, 0.0f);
    for (int k = 1; k <= p.M; ++k) {
      for (int t = 0; t < NTRANS; ++t) {
        vt[(size_t) pos[k] * 8 + t] = p.tw[(size_t) t * (p.M + 1) + k];
        ft[(size_t) pos[k] * 8 + t] = p.tf[(size_t) t * (p.M + 1) + k];
      }
      for (int x = 0; x < p.Kp; ++x) {
        ve[(size_t) x * Mpad + pos[k]] = p.rw[(size_t) x * (p.M + 1) + k];
        fe[(size_t) x * Mpad + pos[k]] = p.rf_[(size_t) x * (p.M + 1) + k];
      }
    }
    {     // emission table of the wave-per-target MSV kernel (long models, small target blocks), same node order as Viterbi's
      std::vector<int16_t> me((size_t) kTabRows * Mpad, (int16_t) kNegPad);
      for (int k = 1; k <= p.M; ++k)
        for (int x = 0; x < p.Kp; ++x) me[(size_t) x * Mpad + pos[k]] = (int16_t) ((int) p.bias_b - (int) p.rb[(size_t) x * (p.M + 1) + k]);
      img.add(&d->msvw_emis, me);
      d->msvwC = C; d->msvw_rows = kTabRows;
      // packed pairs for msv_wavepk_kernel (lane z owns nodes z Cp + 1 .. z Cp + Cp): M > 1021 (no lane kernel, or the eight-lane
      // one: small blocks and the longest targets of large ones still go one per wavefront).  Beyond 2,048 nodes the wavefront
      // kernels' own C (48) would put the table past the LDS; with 36 / 40 nodes per lane and only the rows a target can hold
      // (Kp + 1) it fits up to 2,560 nodes -- these models ran msv_wave_kernel with its table read through L2 until round 6
      // (0.5 TCUPS against ~15).
      int Cp = 0;
      if ((d->msvR <= 0 || d->msvK >= 8) && C % 4 == 0 && C <= 32) Cp = C;
      else if (d->msvR <= 0 && C > 32 && p.M <= 2560 && p.Kp + 1 <= 30) { Cp = p.M <= 2304 ? 36 : 40; d->msvwC = Cp; d->msvw_rows = p.Kp + 1; }
      if (Cp > 0) {
        const int P2 = Cp / 4, rows = d->msvw_rows;
        std::vector<uint32_t> pk((size_t) rows * P2 * 64 * 2);
        auto em = [&](int x, int k) -> uint32_t { return (uint32_t) (uint16_t) (int16_t) ((x < p.Kp && k <= p.M) ? (int) p.bias_b - (int) p.rb[(size_t) x * (p.M + 1) + k] : kNegPad); };
        for (int x = 0; x < rows; ++x)
          for (int j2 = 0; j2 < P2; ++j2)
            for (int z = 0; z < 64; ++z)
              for (int h = 0; h < 2; ++h) {
                const int k = z * Cp + 2 * (2 * j2 + h) + 1;
                pk[(((size_t) x * P2 + j2) * 64 + z) * 2 + h] = em(x, k) | (em(x, k + 1) << 16);
              }
        img.add(&d->msvw_pk, pk);
      }
    }
    img.add(&d->vit_trans, vt);
    img.add(&d->vit_emis, ve);
    img.add(&d->fwd_trans, ft);
    img.add(&d->fwd_emis, fe);
    int gT = 0, gC = 0;
    if (fwdg_pick(p.M, d->vitC, &gT, &gC)) {
      std::vector<float> gt, ge;
      fwdg_build_tables(p, gT, gC, gt, ge);
      d->fwdgT = gT; d->fwdgC = gC;
      img.add(&d->fwdg_trans, gt);
      img.add(&d->fwdg_emis, ge);
    }
  }
  {
    int T = 0, P = 0;
    if (vitpk_pick(p.M, &T, &P)) {
      std::vector<uint32_t> tt, te;
      vitpk_build_tables(p, T, P, tt, te);
      d->vitpkT = T; d->vitpkP = P;
      img.add(&d->vitpk_trans, tt);
      img.add(&d->vitpk_emis, te);
    }
  }
  // bias filter emission odds (esl_hmm_Configure on the 2-state filter HMM, p7_bg_SetFilter)
  {
    const Alphabet &abc = Alphabet::get(p.abc_type);
    std::vector<float> eo((size_t) kTabRows * 2, 1.0f);
    for (int x = 0; x < p.K; ++x) { eo[x * 2 + 0] = p.bgf[x] / p.bgf[x]; eo[x * 2 + 1] = p.compo[x] / p.bgf[x]; }
    for (int x = p.K + 1; x <= p.Kp - 3; ++x)
      for (int s = 0; s < 2; ++s) {
        float e = 0.0f, den = 0.0f;
        for (int y = 0; y < p.K; ++y) if (abc.degen[x][y]) { e += (s == 0 ? p.bgf[y] : p.compo[y]); den += p.bgf[y]; }
        eo[x * 2 + s] = den > 0.0f ? e / den : 0.0f;
      }
    img.add(&d->bias_eo, eo);
  }
}

// Images of the profiles of a batch that have none on ctx's device yet (a scan: all of them): the tables are laid out
// by the host workers, then ONE slab holds them all and ONE copy takes them up -- a device allocation and a stream
// synchronisation per profile cost more than the tables themselves once several batches are in flight.  The images of
// one call share their slab (DevProfile::shared); it goes back to the pool with the last of them.
int get_dev_profiles(const p7x_oprofile *const *oms, int n, DeviceCtx *ctx, DevProfile **out, int nthreads)
{
  std::vector<int> missing;
  for (int i = 0; i < n; ++i) {
    out[i] = nullptr;
    auto *cache = static_cast<DevCache *>(oms[i]->dev_cache);
    std::lock_guard<std::mutex> lk(cache->mu);
    for (DevProfile *d : cache->per_device) if (d->device == ctx->device) { out[i] = d; break; }
    if (!out[i]) missing.push_back(i);
  }
  if (missing.empty()) return P7X_OK;
  const int nm = (int) missing.size();
  std::vector<std::unique_ptr<DevProfile>> fresh((size_t) nm);
  std::vector<HostImage> imgs((size_t) nm);
  auto build = [&](int z) {
    const Profile &p = oms[missing[(size_t) z]]->p;
    auto d = std::make_unique<DevProfile>();
    d->device = ctx->device; d->M = p.M; d->Kp = p.Kp;
    build_host_image(p, d.get(), imgs[(size_t) z]);
    fresh[(size_t) z] = std::move(d);
  };
  if (nm >= 4) host_parallel_for(nm, nthreads, build); else for (int z = 0; z < nm; ++z) build(z);
  std::vector<size_t> off((size_t) nm + 1, 0);
  for (int z = 0; z < nm; ++z) off[(size_t) z + 1] = off[(size_t) z] + imgs[(size_t) z].bytes.size();
  const size_t total = off[(size_t) nm];
  StageLease stage_lease;
  ImageStage &stg = *stage_lease.st;
  int st = P7X_OK;
  if ((st = stg.reserve(total)) != P7X_OK) return st;
  if (stg.stream == nullptr || stg.device != ctx->device) {
    if (stg.stream) (void) hipStreamDestroy(stg.stream);
    P7X_HIP(hipStreamCreateWithFlags(&stg.stream, hipStreamNonBlocking));
    stg.device = ctx->device;
  }
  auto shared = std::make_shared<SlabRef>();
  shared->ctx = ctx;
  if ((st = slab_acquire(ctx, total, &shared->p, &shared->bytes)) != P7X_OK) return st;
  auto place = [&](int z) {
    const HostImage &img = imgs[(size_t) z];
    std::memcpy(stg.pinned + off[(size_t) z], img.bytes.data(), img.bytes.size());
    for (const auto &f : img.fix) *f.first = static_cast<char *>(shared->p) + off[(size_t) z] + f.second;
    fresh[(size_t) z]->shared = shared;
  };
  if (nm >= 4) host_parallel_for(nm, nthreads, place); else for (int z = 0; z < nm; ++z) place(z);
  if (hipMemcpyAsync(shared->p, stg.pinned, total, hipMemcpyHostToDevice, stg.stream) != hipSuccess ||
      hipStreamSynchronize(stg.stream) != hipSuccess) {
    set_error("uploading the profiles' device images failed");
    return P7X_EDEVICE;
  }
  for (int z = 0; z < nm; ++z) {
    const int i = missing[(size_t) z];
    auto *cache = static_cast<DevCache *>(oms[i]->dev_cache);
    std::lock_guard<std::mutex> lk(cache->mu);
    for (DevProfile *d : cache->per_device) if (d->device == ctx->device) { out[i] = d; break; }   // a duplicate in this batch, or another thread
    if (!out[i]) { out[i] = fresh[(size_t) z].get(); cache->per_device.push_back(fresh[(size_t) z].release()); }
  }
  return P7X_OK;
}

int get_dev_profile(const p7x_oprofile *om, DeviceCtx *ctx, DevProfile **out) { return get_dev_profiles(&om, 1, ctx, out, 1); }

} // namespace p7x

using namespace p7x;

extern "C" {

void p7x_oprofile_destroy(p7x_oprofile *om)
{
  if (!om) return;
  if (om->dev_cache) {
    auto *cache = static_cast<DevCache *>(om->dev_cache);
    for (DevProfile *d : cache->per_device) free_dev_profile(d);
    delete cache;
  }
  delete om;
}

} // extern "C"

namespace p7x { void attach_dev_cache(p7x_oprofile *om) { om->dev_cache = new DevCache(); } }
