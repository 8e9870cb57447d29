// p7x_device.hip -- device contexts, the device-resident sequence database and profile images.
#include "p7x_device.hpp"
#include <cstdlib>
#include "p7x_kernels.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

namespace p7x {

// ---------------------------------------------------------------------------- contexts
static std::mutex g_ctx_mu;
static std::map<int, std::unique_ptr<DeviceCtx>> g_ctx;

static int device_count_checked()
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
  return n;
}

// The classes of a batch run on up to eight streams of one workspace, and several workspaces are in flight: ask the
// runtime for more hardware queues than its default of four before it initialises (no effect if the process already
// made a HIP call, or if the user set the variable).
static const int g_hw_queues_set = setenv("GPU_MAX_HW_QUEUES", "8", 0);

int create_tail_stream(DeviceCtx *ctx, bool high_priority, hipStream_t *out)
{ // (Round 5 tried these streams on a subset of the CUs, hipExtStreamCreateWithCUMask with 64 or 128 of 256: the headline fell
  // from 20.0 to 16.5 TCUPS -- such streams lose their priority, and every MSV launch stretched from 17.5 to 22 ms.)
  (void) ctx;
  int least = 0, greatest = 0;
  P7X_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  P7X_HIP(hipStreamCreateWithPriority(out, hipStreamNonBlocking, high_priority ? greatest : least));
  return P7X_OK;
}

int get_ctx(int device, DeviceCtx **out)
{
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  auto it = g_ctx.find(device);
  if (it != g_ctx.end()) { *out = it->second.get(); P7X_HIP(hipSetDevice(device)); return P7X_OK; }
  const int n = device_count_checked();
  if (device < 0 || device >= n) { set_error("no usable HIP device (libp7x has no CPU fallback)"); return P7X_ENODEVICE; }
  P7X_HIP(hipSetDevice(device));
  auto ctx = std::make_unique<DeviceCtx>();
  ctx->device = device;
  {   // cascade stream sets first (see DeviceCtx): set n's main stream goes to queue position off[n] of the round robin
    int least = 0, greatest = 0;
    P7X_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int off[DeviceCtx::kWsSets] = { 0, 4, 2, 6 };
    int created = 0;
    for (int n = 0; n < DeviceCtx::kWsSets; ++n) {
      while (created % 8 != off[n]) { hipStream_t sp = nullptr; P7X_HIP(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, greatest)); ctx->ws_spacers.push_back(sp); ++created; }
      P7X_HIP(hipStreamCreateWithPriority(&ctx->ws_main[n], hipStreamNonBlocking, greatest)); ++created;
      for (auto &q : ctx->ws_side[n]) { P7X_HIP(hipStreamCreateWithPriority(&q, hipStreamNonBlocking, greatest)); ++created; }
    }
  }
  P7X_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  P7X_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  for (auto &e : ctx->msv_done) P7X_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipDeviceProp_t prop;
  P7X_HIP(hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount;
  // exact host arithmetic for everything that depends only on the target length
  std::vector<uint8_t> tjb(kMaxL + 1);
  std::vector<int16_t> xwm(kMaxL + 1);
  std::vector<float> n1(kMaxL + 1);
  const float scale_b = 3.0 / kLog2, scale_w = 500.0 / kLog2;
  for (int L = 0; L <= kMaxL; ++L) {
    tjb[L] = unbiased_byteify(scale_b, logf(3.0f / (float) (L + 3)));
    const float pmove = (2.0f + 1.0f) / ((float) L + 2.0f + 1.0f);
    xwm[L] = wordify(scale_w, logf(pmove));
    n1[L] = null1_score(L);
  }
  P7X_HIP(hipMalloc(&ctx->lt.tjb, tjb.size()));
  P7X_HIP(hipMalloc(&ctx->lt.xwmove, xwm.size() * 2));
  P7X_HIP(hipMalloc(&ctx->lt.null1, n1.size() * 4));
  P7X_HIP(hipMemcpy(ctx->lt.tjb, tjb.data(), tjb.size(), hipMemcpyHostToDevice));
  P7X_HIP(hipMemcpy(ctx->lt.xwmove, xwm.data(), xwm.size() * 2, hipMemcpyHostToDevice));
  P7X_HIP(hipMemcpy(ctx->lt.null1, n1.data(), n1.size() * 4, hipMemcpyHostToDevice));
  {
    std::vector<double> lt(256);
    for (int i = 0; i < 128; ++i) { const double c = 1.0 + ((double) i + 0.5) / 128.0; lt[2 * i] = 1.0 / c; lt[2 * i + 1] = std::log(c); }
    P7X_HIP(hipMalloc(&ctx->lt.logtab, lt.size() * 8));
    P7X_HIP(hipMemcpy(ctx->lt.logtab, lt.data(), lt.size() * 8, hipMemcpyHostToDevice));
  }
  *out = ctx.get();
  g_ctx[device] = std::move(ctx);
  return P7X_OK;
}

// ---------------------------------------------------------------------------- sequence database
// Interleaved tiles: group g = slots [64g, 64g+64) (targets sorted by decreasing length); block b of the
// group holds residues 16b+1..16b+16 of all 64 lanes: tiles[grp_off[g] + b*64 + lane] is one uint4, so a
// wavefront's 16-byte loads are one contiguous 1 KiB segment.  Residues past a lane's end are the pad code Kp.
__global__ void pack_tiles_kernel(const uint8_t *__restrict__ dsq, const int64_t *__restrict__ slot_off,
                                  const int32_t *__restrict__ slot_len, const int64_t *__restrict__ grp_off,
                                  const int32_t *__restrict__ grp_nblk, int64_t nslots, int pad, uint4 *__restrict__ tiles)
{
  const int g = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int64_t slot = (int64_t) g * 64 + lane;
  const int L = slot < nslots ? slot_len[slot] : 0;
  const uint8_t *src = slot < nslots ? dsq + slot_off[slot] : dsq;
  const int nblk = grp_nblk[g];
  uint4 *dst = tiles + grp_off[g] + lane;
  for (int b = wave; b < nblk; b += nwave) {
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t acc = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = b * 16 + q * 4 + r;
        const uint32_t x = (i < L) ? src[i] : (uint32_t) pad;
        acc |= x << (8 * r);
      }
      w[q] = acc;
    }
    dst[(size_t) b * 64] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

} // namespace p7x

using namespace p7x;

extern "C" {

int p7x_device_count(void) { return device_count_checked(); }

int p7x_device_name(int device, char *buf, size_t n)
{
  if (!buf || n == 0) return P7X_EINVAL;
  if (device < 0 || device >= device_count_checked()) { set_error("no such device"); return P7X_ENODEVICE; }
  hipDeviceProp_t prop;
  P7X_HIP(hipGetDeviceProperties(&prop, device));
  std::snprintf(buf, n, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return P7X_OK;
}

int p7x_seqdb_create(int device, int32_t abc_type, const uint8_t *dsq, const int64_t *offsets,
                     const int32_t *lengths, size_t n, p7x_seqdb **out)
{
  if (!out || (n && (!dsq || !offsets || !lengths))) { set_error("p7x_seqdb_create: bad arguments"); return P7X_EINVAL; }
  DeviceCtx *ctx = nullptr;
  int st = get_ctx(device, &ctx);
  if (st != P7X_OK) return st;
  auto db = std::make_unique<p7x_seqdb>();
  db->device = device; db->abc_type = abc_type; db->Kp = Alphabet::get(abc_type).Kp; db->n = (int64_t) n;
  db->h_len.assign(lengths, lengths + n);
  db->h_off.resize(n);
  int maxL = 0; int64_t nres = 0, nslots = 0;
  for (size_t t = 0; t < n; ++t) {
    if (lengths[t] < 0) { set_error("negative target length"); return P7X_EINVAL; }
    if (lengths[t] > kMaxL) { set_error("Target sequence length > 100K, over comparison pipeline limit"); return P7X_ERANGE; }
    nres += lengths[t]; if (lengths[t] > 0) ++nslots; if (lengths[t] > maxL) maxL = lengths[t];
  }
  db->nres = nres; db->nslots = nslots; db->ngroups = (nslots + 63) / 64;
  // sentinel-framed private copy (255 x1..xL 255 ...), as Easel frames ESL_SQ.dsq
  db->h_dsq.assign((size_t) nres + n + 1, 255);
  {
    int64_t pos = 1;
    for (size_t t = 0; t < n; ++t) {
      db->h_off[t] = pos;
      const uint8_t *src = dsq + offsets[t];
      for (int i = 0; i < lengths[t]; ++i) {
        if (src[i] >= db->Kp) { set_error("invalid digital residue code in target"); return P7X_EINVAL; }
        db->h_dsq[pos + i] = src[i];
      }
      pos += lengths[t] + 1;
    }
  }
  // counting sort by decreasing length (stable): slot -> target
  db->h_order.resize(nslots);
  {
    std::vector<int64_t> cnt((size_t) maxL + 2, 0);
    for (size_t t = 0; t < n; ++t) if (lengths[t] > 0) cnt[lengths[t]]++;
    int64_t acc = 0;
    for (int L = maxL; L >= 1; --L) { const int64_t c = cnt[L]; cnt[L] = acc; acc += c; }
    for (size_t t = 0; t < n; ++t) if (lengths[t] > 0) db->h_order[cnt[lengths[t]]++] = (int32_t) t;
  }
  const int64_t G = db->ngroups;
  std::vector<int64_t> slot_off((size_t) std::max<int64_t>(nslots, 1));
  std::vector<int32_t> slot_len((size_t) std::max<int64_t>(G * 64, 1), 0);
  std::vector<int64_t> grp_off((size_t) std::max<int64_t>(G, 1));
  std::vector<int32_t> grp_nblk((size_t) std::max<int64_t>(G, 1));
  int64_t u4 = 0;
  for (int64_t s = 0; s < nslots; ++s) { slot_off[s] = db->h_off[db->h_order[s]]; slot_len[s] = lengths[db->h_order[s]]; }
  for (int64_t g = 0; g < G; ++g) {
    const int Lmax = slot_len[g * 64];     // sorted: first lane is the longest
    grp_nblk[g] = (Lmax + 15) / 16;
    grp_off[g] = u4;
    u4 += (int64_t) grp_nblk[g] * 64;
  }
  db->tile_u4 = u4;
  {   // targets more than three times as long as the median (and longer than 768) are "long" for the packed Viterbi
      // kernel, whose wavefronts run to their longest target: a few per cent of the residues of a proteome.  That rule is
      // for SMALL blocks (the scan orientation, up to 256 groups), whose stages last as long as their longest chain.  Against
      // a large block the device is busy with other batches' kernels while a long chain runs, what counts is the
      // instructions issued, and the wave-per-target kernel issues three times as many per row: only targets beyond
      // 20,000 residues (a 25 ms chain) leave the packed kernel there.  Round 6, 4,000 library profiles x 500,000
      // targets: Viterbi 12.3 -> 9.9 ms per batch, 21.9 -> 22.9 TCUPS (profiles/r06_vit_long_cut.txt).
    const int median = nslots > 0 ? slot_len[nslots / 2] : 0;
    const int cut = debug_opt(OPT_VIT_LONG_CUT) > 0 ? debug_opt(OPT_VIT_LONG_CUT)
                  : (G > 256 ? std::max(20000, 3 * median) : std::max(768, 3 * median));
    int64_t k = 0;
    while (k < nslots && slot_len[k] > cut) ++k;
    db->vit_long_slots = k;
  }
  db->h_grp_len.resize((size_t) G); db->h_grp_suffix.assign((size_t) G + 1, 0);
  for (int64_t g = G - 1; g >= 0; --g) { db->h_grp_len[g] = slot_len[g * 64]; db->h_grp_suffix[g] = db->h_grp_suffix[g + 1] + slot_len[g * 64]; }
  {   // one slab of the context's pool for the six arrays: a block of long-target windows comes and goes with every stage of
      // every nhmmer search, and hipMalloc / hipFree wait for ALL work queued on the device -- the scan and the envelope
      // kernels of the other searches in flight (round 5: with them, searches in flight did not overlap at all)
    auto up = [](size_t v) { return (v + 255) & ~(size_t) 255; };
    const size_t b_dsq = up(db->h_dsq.size()), b_off = up(slot_off.size() * 8), b_len = up(slot_len.size() * 4), b_goff = up(grp_off.size() * 8),
                 b_nblk = up(grp_nblk.size() * 4), b_tiles = up((size_t) std::max<int64_t>(u4, 1) * 16);
    void *base = nullptr;
    if ((st = slab_acquire(ctx, b_dsq + b_off + b_len + b_goff + b_nblk + b_tiles, &base, &db->slab_bytes)) != P7X_OK) return st;
    db->slab = base;
    unsigned char *q = static_cast<unsigned char *>(base);
    db->d_tiles = reinterpret_cast<uint4 *>(q); q += b_tiles;            // 16-byte elements first
    db->d_slot_off = reinterpret_cast<int64_t *>(q); q += b_off;
    db->d_grp_off = reinterpret_cast<int64_t *>(q); q += b_goff;
    db->d_slot_len = reinterpret_cast<int32_t *>(q); q += b_len;
    db->d_grp_nblk = reinterpret_cast<int32_t *>(q); q += b_nblk;
    db->d_dsq = q;
  }
  // an upload that fails hands the slab back with the half-built database (it leaked)
  struct SlabGuard { DeviceCtx *ctx; p7x_seqdb *db; ~SlabGuard() { if (db && db->slab) { slab_release(ctx, db->slab, db->slab_bytes); db->slab = nullptr; } } } guard{ ctx, db.get() };
  P7X_HIP(hipMemcpy(db->d_dsq, db->h_dsq.data(), db->h_dsq.size(), hipMemcpyHostToDevice));
  P7X_HIP(hipMemcpy(db->d_slot_off, slot_off.data(), slot_off.size() * 8, hipMemcpyHostToDevice));
  P7X_HIP(hipMemcpy(db->d_slot_len, slot_len.data(), slot_len.size() * 4, hipMemcpyHostToDevice));
  P7X_HIP(hipMemcpy(db->d_grp_off, grp_off.data(), grp_off.size() * 8, hipMemcpyHostToDevice));
  P7X_HIP(hipMemcpy(db->d_grp_nblk, grp_nblk.data(), grp_nblk.size() * 4, hipMemcpyHostToDevice));
  if (G > 0) {
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned) G), dim3(256), 0, ctx->stream, db->d_dsq, db->d_slot_off,
                       db->d_slot_len, db->d_grp_off, db->d_grp_nblk, nslots, db->Kp, db->d_tiles);
    P7X_HIP(hipGetLastError());
    P7X_HIP(hipStreamSynchronize(ctx->stream));
  }
  guard.db = nullptr;
  *out = db.release();
  return P7X_OK;
}

void p7x_seqdb_destroy(p7x_seqdb *db)
{
  if (!db) return;
  (void) hipSetDevice(db->device);
  DeviceCtx *ctx = nullptr;
  if (db->slab && get_ctx(db->device, &ctx) == P7X_OK) slab_release(ctx, db->slab, db->slab_bytes);      // back to the pool: no device-wide wait
  delete db;
}

int64_t p7x_seqdb_ntargets(const p7x_seqdb *db) { return db ? db->n : -1; }
int64_t p7x_seqdb_nresidues(const p7x_seqdb *db) { return db ? db->nres : -1; }

} // extern "C"
