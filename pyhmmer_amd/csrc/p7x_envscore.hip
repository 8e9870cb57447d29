// p7x_envscore.hip -- host side of the envelope kernel (p7x_envelope.hip): batches the envelope requests of one search,
// runs env_kernel on a leased stream and hands the results (scores, null2 odds, optimal-accuracy traces) back to the
// host stage (p7x_domaindef.cpp / p7x_tophits.cpp) through the EnvelopeScorer interface.
#include "p7x_wave.hpp"
#include "p7x_host.hpp"
#include <cstdlib>
#include <cstring>
#include <memory>

namespace p7x {

// ---------------------------------------------------------------------------- envelope rescoring on the device
// Device and pinned-host buffers live for the thread (grow-only), like the cascade workspace.
struct EnvBuffers {
  int device = -1;
  float *work = nullptr; size_t work_floats = 0;
  unsigned char *d_in = nullptr; size_t d_in_cap = 0;        // env_sq | tr_off | env_len | env_L
  unsigned char *d_out = nullptr; size_t d_out_cap = 0;      // out_sc | out_null2 | out_status | tr_n | tr_a | tr_i | tr_pp
  unsigned char *h_out = nullptr; size_t h_out_cap = 0;      // pinned mirror of d_out
  hipStream_t stream = nullptr;                               // per host thread: concurrent host stages do not wait on each other
  ~EnvBuffers() {
    if (device < 0) return;
    (void) hipSetDevice(device);
    if (stream) (void) hipStreamDestroy(stream);
    (void) hipFree(work); (void) hipFree(d_in); (void) hipFree(d_out);
    pinned_release(h_out, h_out_cap);
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
  if (const char *e = std::getenv("P7X_ENV_WORKSPACE_GB")) { const long v = std::atol(e); if (v > 0) gb = (size_t) v; }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 2 < gb << 30) return free_b / 2;
  return gb << 30;
}

class DeviceEnvelopeScorer final : public EnvelopeScorer {
public:
  DeviceEnvelopeScorer(DeviceCtx *ctx, const DevProfile *dp, const p7x_seqdb *db, const Profile &p) : ctx_(ctx), dp_(dp), db_(db), p_(p) {}
  ~DeviceEnvelopeScorer() override { if (lease_) { if (lease_->stream) (void) hipStreamSynchronize(lease_->stream); release_env_buffers(lease_); } }

  int begin(const std::vector<EnvelopeRequest> &req, const std::vector<int32_t> &targets) override
  {
    const int nenv = (int) req.size();
    nenv_ = nenv;
    if (nenv == 0) return P7X_OK;
    P7X_HIP(hipSetDevice(db_->device));
    EnvBuffers *eb = nullptr;
    {
      EnvPool &ep = env_pool();
      std::lock_guard<std::mutex> lk(ep.mu);
      for (size_t i = 0; i < ep.all.size() && !eb; ++i)
        if (!ep.busy[i] && ep.all[i]->device == db_->device) { ep.busy[i] = 1; eb = ep.all[i]; }
      if (!eb) { eb = new EnvBuffers(); eb->device = db_->device; ep.all.push_back(eb); ep.busy.push_back(1); }
    }
    lease_ = eb;
    if (!eb->stream) {
      int least = 0, greatest = 0;
      P7X_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      P7X_HIP(hipStreamCreateWithPriority(&eb->stream, hipStreamNonBlocking, least));
    }

    // inputs
    const size_t in_bytes = (size_t) nenv * (8 + 8 + 4 + 4);
    h_in_.resize(in_bytes);
    std::vector<unsigned char> &h_in = h_in_;
    int64_t *env_sq = reinterpret_cast<int64_t *>(h_in.data());
    int64_t *tr_off = env_sq + nenv;
    int32_t *env_len = reinterpret_cast<int32_t *>(tr_off + nenv);
    int32_t *env_L = env_len + nenv;
    int Lmax = 1; int64_t ntr = 0;
    for (int r = 0; r < nenv; ++r) {
      const int t = targets[(size_t) req[r].item];
      const int Ld = req[r].j - req[r].i + 1;
      env_sq[r] = db_->h_off[t] + (req[r].i - 1);
      env_len[r] = Ld; env_L[r] = db_->h_len[t];
      tr_off[r] = ntr; ntr += (int64_t) Ld + p_.M + 16;
      Lmax = std::max(Lmax, Ld);
    }
    // workspace: one slab per resident wavefront, sized for the longest envelope of the batch
    const int C = dp_->vitC;
    const size_t stride = env_work_floats(C, Lmax);
    int nblocks = 0;
    int st = env_max_blocks(C, p_.Kp + 1, ctx_->num_cu, &nblocks);
    if (st != P7X_OK) return st;
    nblocks = std::min(nblocks, (nenv + 3) / 4);
    const size_t budget = env_budget_bytes();
    while (nblocks > 1 && (size_t) nblocks * 4 * stride * 4 > budget) nblocks = (nblocks + 1) / 2;
    const size_t work_floats = (size_t) nblocks * 4 * stride;
    if (work_floats * 4 > budget && work_floats > eb->work_floats) {
      set_error("envelope workspace does not fit in device memory (envelope of " + std::to_string(Lmax) + " residues, M = " + std::to_string(p_.M) + ")");
      return P7X_EMEM;
    }
    if (work_floats > eb->work_floats) {
      (void) hipFree(eb->work); eb->work = nullptr; eb->work_floats = 0;
      P7X_HIP(hipMalloc(&eb->work, work_floats * 4)); eb->work_floats = work_floats;
    }
    if (in_bytes > eb->d_in_cap) {
      (void) hipFree(eb->d_in); eb->d_in = nullptr;
      P7X_HIP(hipMalloc(&eb->d_in, in_bytes * 2)); eb->d_in_cap = in_bytes * 2;
    }
    // outputs: [out_sc 2f][null2 32f][status i][tr_n i] per envelope, then the three trace arrays
    const size_t o_sc = 0, o_n2 = o_sc + (size_t) nenv * 8, o_st = o_n2 + (size_t) nenv * 128, o_n = o_st + (size_t) nenv * 4;
    const size_t o_ta = o_n + (size_t) nenv * 4, o_ti = o_ta + (size_t) ntr * 4, o_tp = o_ti + (size_t) ntr * 4;
    const size_t out_bytes = o_tp + (size_t) ntr * 4;
    if (out_bytes > eb->d_out_cap) {
      (void) hipFree(eb->d_out); eb->d_out = nullptr;
      pinned_release(eb->h_out, eb->h_out_cap); eb->h_out = nullptr; eb->h_out_cap = 0;
      const size_t cap = out_bytes + out_bytes / 2;
      P7X_HIP(hipMalloc(&eb->d_out, cap)); eb->d_out_cap = cap;
      { void *hp = nullptr; size_t got = 0; const int pst = pinned_acquire(cap, &hp, &got); if (pst != P7X_OK) return pst;
        eb->h_out = static_cast<decltype(eb->h_out)>(hp); eb->h_out_cap = got; }
    }
    hipStream_t s = eb->stream;
    P7X_HIP(hipMemcpyAsync(eb->d_in, h_in.data(), in_bytes, hipMemcpyHostToDevice, s));
    EnvArgs a{};
    a.M = p_.M; a.C = C; a.K = p_.K; a.nrows = p_.Kp + 1;
    a.trans = dp_->fwd_trans; a.emis = dp_->fwd_emis; a.dsq = db_->d_dsq;
    a.nj = 0.0f; a.xf_e_move = 1.0f; a.xf_e_loop = 0.0f;              // p7_oprofile_ReconfigUnihit
    a.nenv = nenv;
    a.env_sq = reinterpret_cast<const int64_t *>(eb->d_in);
    a.tr_off = a.env_sq + nenv;
    a.env_len = reinterpret_cast<const int32_t *>(a.tr_off + nenv);
    a.env_L = a.env_len + nenv;
    a.work = eb->work; a.work_stride = (int64_t) stride; a.Lmax = Lmax;
    a.out_sc = reinterpret_cast<float *>(eb->d_out + o_sc);
    a.out_null2 = reinterpret_cast<float *>(eb->d_out + o_n2);
    a.out_status = reinterpret_cast<int32_t *>(eb->d_out + o_st);
    a.tr_n = reinterpret_cast<int32_t *>(eb->d_out + o_n);
    a.tr_a = reinterpret_cast<uint32_t *>(eb->d_out + o_ta);
    a.tr_i = reinterpret_cast<int32_t *>(eb->d_out + o_ti);
    a.tr_pp = reinterpret_cast<float *>(eb->d_out + o_tp);
    if ((st = env_launch(a, nblocks, s)) != P7X_OK) return st;
    P7X_HIP(hipMemcpyAsync(eb->h_out, eb->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    eb_ = eb; o_sc_ = o_sc; o_n2_ = o_n2; o_st_ = o_st; o_n_ = o_n; o_ta_ = o_ta; o_ti_ = o_ti; o_tp_ = o_tp;
    return P7X_OK;
  }

  int wait(std::vector<EnvelopeResult> &res) override
  {
    const int nenv = nenv_;
    res.assign((size_t) nenv, EnvelopeResult{});
    if (nenv == 0) return P7X_OK;
    P7X_HIP(hipSetDevice(db_->device));
    EnvBuffers *eb = eb_;
    P7X_HIP(hipStreamSynchronize(eb->stream));
    const size_t o_sc = o_sc_, o_n2 = o_n2_, o_st = o_st_, o_n = o_n_, o_ta = o_ta_, o_ti = o_ti_, o_tp = o_tp_;
    const int64_t *tr_off = reinterpret_cast<const int64_t *>(h_in_.data()) + nenv;
    const float *h_sc = reinterpret_cast<const float *>(eb->h_out + o_sc), *h_n2 = reinterpret_cast<const float *>(eb->h_out + o_n2);
    const int32_t *h_st = reinterpret_cast<const int32_t *>(eb->h_out + o_st), *h_n = reinterpret_cast<const int32_t *>(eb->h_out + o_n);
    const uint32_t *h_ta = reinterpret_cast<const uint32_t *>(eb->h_out + o_ta);
    const int32_t *h_ti = reinterpret_cast<const int32_t *>(eb->h_out + o_ti);
    const float *h_tp = reinterpret_cast<const float *>(eb->h_out + o_tp);
    for (int r = 0; r < nenv; ++r) {
      EnvelopeResult &e = res[(size_t) r];
      e.envsc = h_sc[2 * r]; e.oasc = h_sc[2 * r + 1]; e.status = h_st[r];
      std::memcpy(e.null2, h_n2 + (size_t) r * 32, sizeof(e.null2));
      e.ntrace = h_n[r]; e.ta = h_ta + tr_off[r]; e.ti = h_ti + tr_off[r]; e.tp = h_tp + tr_off[r];
    }
    return P7X_OK;
  }

private:
  DeviceCtx *ctx_; const DevProfile *dp_; const p7x_seqdb *db_; const Profile &p_;
  int nenv_ = 0;
  std::vector<unsigned char> h_in_;
  EnvBuffers *eb_ = nullptr;
  EnvBuffers *lease_ = nullptr;
  size_t o_sc_ = 0, o_n2_ = 0, o_st_ = 0, o_n_ = 0, o_ta_ = 0, o_ti_ = 0, o_tp_ = 0;
};

std::unique_ptr<EnvelopeScorer> make_device_envelope_scorer(DeviceCtx *ctx, const DevProfile *dp, const p7x_seqdb *db, const Profile &p)
{
  return std::make_unique<DeviceEnvelopeScorer>(ctx, dp, db, p);
}

} // namespace p7x
