// p7x_ensemble.hip -- stochastic traceback ensembles of multi-domain regions on CDNA4.
//
// Upstream (p7_domaindef.c region_trace_ensemble, reference include/libhmmer/p7_domaindef.pxd:23-59, p7_spensemble.pxd:3-39;
// re-seeding plan7.pyx:5684-5688) resolves a region that seems to hold several domains by sampling 200 tracebacks from a
// multihit Forward matrix of the region (impl_sse/stotrace.c p7_StochasticTrace), all from ONE generator stream, collecting
// the domains' end points for clustering and the position-specific null2 odds of every sampled domain
// (impl_sse/null2.c p7_Null2_ByTrace).  A region is therefore one serial walk of ~200 x (region length + alignment
// length) dependent choices.  Two kernels:
//
//   ens_forward_kernel<C>   one wavefront per region: p7_Forward, multihit, with the row in registers (EnvForward<C>, the
//                           envelope kernel's recurrence: same operations, same order, same bits as the host twin's
//                           forward_full) -- and, cell by cell while the values are at hand, the INTEGER thresholds of
//                           every choice a traceback can face there (p7x_choice.hpp): all floating-point work of the
//                           ensemble happens here, lane-parallel.
//   ens_walk_kernel         one wavefront per region: the 200 walks.  The state (i, k, state, generator) is
//                           wave-uniform; a step is one 16-byte record and a few integer compares.  The E state's
//                           choice among 2M cells and the null2 vector of a finished domain use the 64 lanes.
//
// Clustering of the sampled end points (a few hundred integers per region) stays on the host (p7x_domaindef.cpp).
#include <cstdio>
#include <cstdlib>
#include "p7x_wave.hpp"
#include "p7x_envfwd.hpp"
#include "p7x_choice.hpp"
#include "p7x_host.hpp"
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace p7x {

namespace {
enum { tM = 1, tD = 2, tI = 3, tS = 4, tN = 5, tB = 6, tE = 7, tC = 8, tT = 9, tJ = 10 };     // p7T_* (p7_trace.pxd)
__device__ __forceinline__ uint32_t fbits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float bitsf(uint32_t v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ void own_stores_visible()
{ // stores of this wavefront are read back by other lanes of it: they have left the wavefront, and the CU's vector cache is
  // coherent for its own stores (work-group scope; see phase_fence in p7x_envelope.hip)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
} // namespace

// ---------------------------------------------------------------------------- Forward fill + choice records
// Tables are read where they lie (L2): a launch has a few hundred regions of assorted profiles at most, one wavefront
// each, so there is nothing to share in LDS and the row's loads are issued together at its top.
template <int C>
__global__ void __launch_bounds__(64) ens_forward_kernel(const EnsArgs a, const int *__restrict__ reg_list)
{
  constexpr int Mpad = 64 * C;
  const int lane = threadIdx.x;
  const EnsRegion reg = a.regions[reg_list[blockIdx.x]];
  const EnsJob job = a.jobs[reg.job];
  const int M = job.M, Mrow = M + 1, Lr = reg.Lr;
  const float4 *tr = reinterpret_cast<const float4 *>(job.trans);
  const float *em = reinterpret_cast<const float *>(job.emis);
  const uint8_t *sq = a.dsq + reg.sq;                              // sq[0] = first residue of the region
  ChoiceCell *cells = a.cells + reg.cell0;
  float2 *md = a.md + reg.cell0;
  ChoiceRow *rows = a.rows + reg.row0;
  const float nj = 1.0f;                                           // p7_oprofile_ReconfigMultihit(om, L)
  const float pmove = (2.0f + nj) / ((float) reg.L + 2.0f + nj), ploop = 1.0f - pmove;
  const float xf_e_move = 0.5f, xf_e_loop = 0.5f;

  EnvForward<C> f;
  f.init(tr, lane, pmove);
  // leaving transitions of the node before this lane's first one (the delete cell's choice looks one node back)
  float p_md0, p_dd0;
  { const F8 t = load_f8(tr, (C - 1) * 64 + lane); p_md0 = dpp_shr1f(t.md, 0.0f); p_dd0 = dpp_shr1f(t.dd, 0.0f); }
  if (lane == 0) {
    ChoiceRow r0{};
    uint32_t T, b;
    choice_pair(f.xN * pmove, f.xJ * pmove, &T, &b);               // B(0): N or J
    r0.x[2] = T; r0.x[3] = b << 4;
    rows[0] = r0;
  }
  float pC = f.xC, pJ = f.xJ, pB = f.xB;
  for (int i0 = 0; i0 < Lr; i0 += 64) {
    const int nrow = min(64, Lr - i0);
    const uint32_t resid = (lane < nrow) ? sq[i0 + lane] : 0;
    for (int r = 0; r < nrow; ++r) {
      const int i = i0 + r + 1;
      float pm[C], pi[C], pd[C];
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) { pm[c] = f.mm[c]; pi[c] = f.im[c]; pd[c] = f.dm[c]; }
      f.row(tr, em, Mpad, lane, __builtin_amdgcn_readlane((int) resid, r), pmove, ploop, xf_e_move, xf_e_loop);
      // the predecessors of this lane's first node live in the lane before (all lanes take part in the moves)
      float mp = dpp_shr1f(pm[C - 1], 0.0f), ip = dpp_shr1f(pi[C - 1], 0.0f), dp = dpp_shr1f(pd[C - 1], 0.0f);
      float cm = dpp_shr1f(f.mm[C - 1], 0.0f), cd = dpp_shr1f(f.dm[C - 1], 0.0f);
      float lmd = p_md0, ldd = p_dd0;
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) {
        const int k = lane * C + c + 1;
        const F8 t = load_f8(tr, c * 64 + lane);
        if (k <= M) {
          ChoiceCell cell;
          choice_cell_m(pB * t.bm, mp * t.mm, ip * t.im, dp * t.dm, cell.m);
          uint32_t bi, bd;
          choice_pair(pm[c] * t.mi, pi[c] * t.ii, &cell.id[0], &bi);
          choice_pair(cm * lmd, cd * ldd, &cell.id[1], &bd);
          cell.id[2] = bi | (bd << 2); cell.id[3] = 0;
          uint4 *dst = reinterpret_cast<uint4 *>(cells + (size_t) i * Mrow + k);
          dst[0] = make_uint4(cell.m[0], cell.m[1], cell.m[2], cell.m[3]);
          dst[1] = make_uint4(cell.id[0], cell.id[1], cell.id[2], cell.id[3]);
          md[(size_t) i * Mrow + k] = make_float2(f.mm[c], f.dm[c]);
        }
        mp = pm[c]; ip = pi[c]; dp = pd[c];
        cm = f.mm[c]; cd = f.dm[c]; lmd = t.md; ldd = t.dd;
      }
      if (lane == 0) {
        ChoiceRow rw{};
        uint32_t bc, bj, bb;
        choice_pair(pC * ploop, f.xE * xf_e_move * f.scale, &rw.x[0], &bc);     // C(i): C(i-1) or E(i)
        choice_pair(pJ * ploop, f.xE * xf_e_loop * f.scale, &rw.x[1], &bj);     // J(i): J(i-1) or E(i)
        choice_pair(f.xN * pmove, f.xJ * pmove, &rw.x[2], &bb);                 // B(i): N(i) or J(i)
        rw.x[3] = bc | (bj << 2) | (bb << 4);
        rw.e[0] = fbits((float) (1.0 / (double) f.xE));
        rows[i] = rw;
      }
      pC = f.xC; pJ = f.xJ; pB = f.xB;
    }
  }
}

// ---------------------------------------------------------------------------- the walks
// LDS: visit counts per node (p7_Null2_ByTrace's usage counts of one domain), the null2 vector of the domain just
// finished, its four per-stripe partial sums, and -- for regions that fit -- the per-residue accumulators.
__global__ void __launch_bounds__(64) ens_walk_kernel(const EnsArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int r = blockIdx.x;
  const EnsRegion reg = a.regions[r];
  const EnsJob job = a.jobs[reg.job];
  const int M = job.M, Mrow = M + 1, Lr = reg.Lr, Q = job.Q, K = job.K, Kp = job.Kp;
  uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);                          // [M + 2]
  float *n2v = reinterpret_cast<float *>(cnt + ((M + 2 + 31) & ~31));          // [32]
  float *accz = n2v + 32;                                                      // [4][32]
  float *n2lds = accz + 128;                                                   // [Lr + 1] when it fits
  const bool n2_in_lds = Lr <= a.n2_lds_cap;
  float *n2g = a.n2acc + reg.row0;
  float *n2 = n2_in_lds ? n2lds : n2g;
  for (int k = lane; k < M + 2; k += 64) cnt[k] = 0;
  for (int pos = lane; pos <= Lr; pos += 64) n2[pos] = 0.0f;
  if (!n2_in_lds) own_stores_visible();
  __syncthreads();
  const ChoiceCell *__restrict__ cells = a.cells + reg.cell0;
  const float2 *__restrict__ md = a.md + reg.cell0;
  const ChoiceRow *__restrict__ rows = a.rows + reg.row0;
  const uint8_t *__restrict__ sq = a.dsq + reg.sq;
  const float *__restrict__ rft = job.rft;
  int32_t *dom = a.dom + reg.dom0 * 5;
  uint32_t x = a.seed_x;
  int ndom = 0, status = 0;
  const int step_cap = 4 * (Lr + M) + 64;

  for (int t = 0; t < a.nsamples && status == 0; ++t) {
    int i = Lr, k = 0, st = tC;
    int hi = Lr;                        // residues hi+1 .. Lr have received this sample's contribution
    int dj = 0, dm_ = 0, di = 0, dk = 0, Ld = 0, klo = 0, khi = 0;
    int steps = 0;
    bool running = true;
    while (running) {
      if (++steps > step_cap) { status |= 1; break; }
      // the walk's state is the same in every lane: keep it in scalar registers (scalar branches, one address per load)
      st = rfl(st); i = rfl(i); k = rfl(k); x = (uint32_t) rfl((int) x);
      switch (st) {
        case tC: {
          if (i < 1) { status |= 2; running = false; break; }
          x = lcg_next(x);
          const uint4 rw = *reinterpret_cast<const uint4 *>(rows[i].x);
          if (choice_pick_pair(rw.x, rw.w & 3u, x) == 0) --i; else st = tE;
          break;
        }
        case tJ: {
          if (i < 1) { status |= 2; running = false; break; }
          x = lcg_next(x);
          const uint4 rw = *reinterpret_cast<const uint4 *>(rows[i].x);
          if (choice_pick_pair(rw.y, (rw.w >> 2) & 3u, x) == 0) --i; else st = tE;
          break;
        }
        case tE: {
          // select_e: the first cell, in the striped visiting order (q outer; four match cells, then four delete cells),
          // whose cumulative share of xE(i) exceeds the deviate; the sum runs in double, as upstream's does
          x = lcg_next(x);
          const double roll = (double) x / 4294967296.0;
          const float norm = bitsf(rows[i].e[0]);
          const float2 *mdr = md + (size_t) i * Mrow;
          double sum = 0.0;
          int found = 0;
          for (int pass = 0; pass < 2 && !found; ++pass) {
            for (int q = 0; q < Q && !found; ++q) {
              float mv[4], dv[4];
#pragma unroll
              for (int z = 0; z < 4; ++z) {
                const int kk = z * Q + q + 1;
                float2 v = make_float2(0.0f, 0.0f);
                if (kk <= M) v = mdr[kk];
                mv[z] = (kk <= M) ? v.x * norm : 0.0f;
                dv[z] = (kk <= M) ? v.y * norm : 0.0f;
              }
#pragma unroll
              for (int z = 0; z < 4; ++z) { sum += (double) mv[z]; if (!found && roll < sum) { found = 1; k = z * Q + q + 1; st = tM; } }
#pragma unroll
              for (int z = 0; z < 4; ++z) { sum += (double) dv[z]; if (!found && roll < sum) { found = 1; k = z * Q + q + 1; st = tD; } }
            }
            if (!found && sum < 0.99) break;
          }
          if (!found) { status |= 4; running = false; break; }
          dj = 0; Ld = 0; khi = k; klo = k;
          break;
        }
        case tM: {
          if (i < 1 || k < 1) { status |= 2; running = false; break; }
          if (lane == 0) cnt[k] += 1;
          ++Ld; klo = k;
          if (dj == 0) { dj = i; dm_ = k; }
          di = i; dk = k;
          x = lcg_next(x);
          const uint4 c4 = *reinterpret_cast<const uint4 *>(cells[(size_t) i * Mrow + k].m);
          const uint32_t cw[4] = { c4.x, c4.y, c4.z, c4.w };
          const int c = choice_pick_m(cw, x);
          --i; --k;
          st = (c == 0) ? tB : (c == 1) ? tM : (c == 2) ? tI : tD;
          break;
        }
        case tI: {
          if (i < 1 || k < 1) { status |= 2; running = false; break; }
          if (lane == 0) cnt[k] += 1;
          ++Ld; klo = k;
          x = lcg_next(x);
          const uint4 c4 = *reinterpret_cast<const uint4 *>(cells[(size_t) i * Mrow + k].id);
          st = (choice_pick_pair(c4.x, c4.z & 3u, x) == 0) ? tM : tI;
          --i;
          break;
        }
        case tD: {
          if (i < 1 || k < 1) { status |= 2; running = false; break; }
          klo = k;
          x = lcg_next(x);
          const uint4 c4 = *reinterpret_cast<const uint4 *>(cells[(size_t) i * Mrow + k].id);
          st = (choice_pick_pair(c4.y, (c4.z >> 2) & 3u, x) == 0) ? tM : tD;
          --k;
          break;
        }
        case tB: {
          // a domain is complete: residues di .. dj of the region, nodes dk .. dm_ (p7_trace_Index); its end points go out for
          // clustering, its null2 odds (p7_Null2_ByTrace over the visit counts) onto the residues di+1 .. dj
          if (dj == 0 || Ld < 1) { status |= 8; running = false; break; }
          if (ndom < a.dom_cap) {
            if (lane == 0) { int32_t *o = dom + (size_t) ndom * 5; o[0] = t; o[1] = di; o[2] = dj; o[3] = dk; o[4] = dm_; }
          } else status |= 16;                                            // more domains than the record holds: the host repeats the region
          ++ndom;
          __syncthreads();                                                // the counts are in
          {
            const float nrm = (float) (1.0 / (double) (float) Ld);
            const int z = lane >> 4, xs = lane & 15;
            float acc0 = 0.0f, acc1 = 0.0f;
            const int q0 = max(0, klo - 1 - z * Q), q1 = min(Q - 1, khi - 1 - z * Q);
            for (int q = q0; q <= q1; ++q) {
              const int kk = q + 1 + z * Q;
              const uint32_t cv = cnt[kk];
              if (cv != 0 && kk <= M) {
                const float w = (float) cv * nrm;
                acc0 = acc0 + w * rft[(size_t) kk * 32 + xs];
                acc1 = acc1 + w * rft[(size_t) kk * 32 + 16 + xs];
              }
            }
            accz[z * 32 + xs] = acc0; accz[z * 32 + 16 + xs] = acc1;
          }
          __syncthreads();
          if (lane < 32) n2v[lane] = ((accz[lane] + accz[32 + lane]) + (accz[64 + lane] + accz[96 + lane])) + 0.0f;      // + xfactor: no N, C, J inside a domain
          __syncthreads();
          {   // esl_abc_FAvgScVec over the degenerate codes; gap, nonresidue and missing-data codes score 1
            float v = 0.0f; bool set = false;
            if (lane > K && lane <= Kp - 3) {
              float res = 0.0f; int n = 0;
              for (int y = 0; y < K; ++y) if (job.degen[lane * 32 + y]) { res += n2v[y]; ++n; }
              v = res / (float) n; set = true;
            } else if (lane == K || lane == Kp - 2 || lane == Kp - 1) { v = 1.0f; set = true; }
            __syncthreads();
            if (set) n2v[lane] = v;
          }
          __syncthreads();
          for (int pos = dj + 1 + lane; pos <= hi; pos += 64) n2[pos] += 1.0f;
          for (int pos = di + 1 + lane; pos <= dj; pos += 64) n2[pos] += n2v[sq[pos - 1]];
          hi = di;
          for (int kk = klo + lane; kk <= khi; kk += 64) cnt[kk] = 0;
          if (!n2_in_lds) own_stores_visible();
          __syncthreads();
          x = lcg_next(x);
          const uint4 rw = *reinterpret_cast<const uint4 *>(rows[i].x);
          if (choice_pick_pair(rw.z, (rw.w >> 4) & 3u, x) == 0) running = false;     // N: the rest of the trace is N ... N S
          else st = tJ;
          break;
        }
        default: status |= 32; running = false; break;
      }
    }
    if (status == 0) {
      for (int pos = 1 + lane; pos <= hi; pos += 64) n2[pos] += 1.0f;
      if (!n2_in_lds) own_stores_visible();
      __syncthreads();
    }
  }
  if (n2_in_lds) for (int pos = lane; pos <= Lr; pos += 64) n2g[pos] = n2[pos];
  if (lane == 0) { a.out_ndom[r] = ndom; a.out_status[r] = status; }
}

// ---------------------------------------------------------------------------- launches
static int ens_forward_launch(int C, const EnsArgs &a, const int *d_reg_list, int n, hipStream_t st)
{
  if (n <= 0) return P7X_OK;
#define P7X_ENS_CASE(CC) case CC: hipLaunchKernelGGL(ens_forward_kernel<CC>, dim3((unsigned) n), dim3(64), 0, st, a, d_reg_list); break;
  switch (C) {
    P7X_ENS_CASE(1) P7X_ENS_CASE(2) P7X_ENS_CASE(3) P7X_ENS_CASE(4) P7X_ENS_CASE(5) P7X_ENS_CASE(6) P7X_ENS_CASE(8) P7X_ENS_CASE(10)
    P7X_ENS_CASE(12) P7X_ENS_CASE(16) P7X_ENS_CASE(20) P7X_ENS_CASE(24) P7X_ENS_CASE(32) P7X_ENS_CASE(48) P7X_ENS_CASE(64)
    P7X_ENS_CASE(96) P7X_ENS_CASE(128)
    default: set_error("model too long for the ensemble kernel"); return P7X_EINVAL;
  }
#undef P7X_ENS_CASE
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

static size_t ens_walk_lds_bytes(int maxM, int n2_cap)
{
  return (size_t) (((maxM + 2 + 31) & ~31) + 32 + 128 + (n2_cap + 1) + 15) * 4;
}

static int ens_walk_launch(const EnsArgs &a, int maxM, hipStream_t st)
{
  if (a.nregions <= 0) return P7X_OK;
  const size_t lds = ens_walk_lds_bytes(maxM, a.n2_lds_cap);
  if (lds > 64 * 1024)
    P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ens_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipLaunchKernelGGL(ens_walk_kernel, dim3((unsigned) a.nregions), dim3(64), lds, st, a);
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

// ---------------------------------------------------------------------------- host side
// Buffers are grow-only and leased from a process-wide pool, like the envelope scorer's (p7x_envscore.hip).
namespace {
struct EnsBuffers {
  int device = -1;
  unsigned char *work = nullptr; size_t work_bytes = 0;       // cells | md | rows
  unsigned char *d_in = nullptr; size_t d_in_cap = 0;
  unsigned char *h_in = nullptr; size_t h_in_cap = 0;
  unsigned char *d_out = nullptr; size_t d_out_cap = 0;
  unsigned char *h_out = nullptr; size_t h_out_cap = 0;
  hipStream_t stream = nullptr;
};
struct EnsPool { std::mutex mu; std::vector<EnsBuffers *> all; std::vector<char> busy; };
EnsPool &ens_pool() { static EnsPool *p = new EnsPool(); return *p; }
void release_ens_buffers(EnsBuffers *eb)
{
  if (!eb) return;
  EnsPool &ep = ens_pool();
  std::lock_guard<std::mutex> lk(ep.mu);
  for (size_t i = 0; i < ep.all.size(); ++i) if (ep.all[i] == eb) ep.busy[i] = 0;
}
int grow_device(DeviceCtx *ctx, unsigned char *&p, size_t &cap, size_t need, size_t floor_bytes)
{
  if (need <= cap) return P7X_OK;
  slab_release(ctx, p, cap); p = nullptr; cap = 0;
  void *dp = nullptr; size_t got = 0;
  const int st = slab_acquire(ctx, std::max(need + need / 4, floor_bytes), &dp, &got);
  if (st != P7X_OK) return st;
  p = static_cast<unsigned char *>(dp); cap = got;
  return P7X_OK;
}
int grow_pinned(unsigned char *&p, size_t &cap, size_t need, size_t floor_bytes)
{
  if (need <= cap) return P7X_OK;
  pinned_release(p, cap); p = nullptr; cap = 0;
  void *hp = nullptr; size_t got = 0;
  const int st = pinned_acquire(std::max(need + need / 4, floor_bytes), &hp, &got);
  if (st != P7X_OK) return st;
  p = static_cast<unsigned char *>(hp); cap = got;
  return P7X_OK;
}
size_t align256(size_t v) { return (v + 255) & ~(size_t) 255; }
} // namespace

class DeviceEnsembleRunner final : public EnsembleRunner {
public:
  DeviceEnsembleRunner(DeviceCtx *ctx, const p7x_seqdb *db) : ctx_(ctx), db_(db) {}
  ~DeviceEnsembleRunner() override { if (lease_) { if (lease_->stream) (void) hipStreamSynchronize(lease_->stream); release_ens_buffers(lease_); } }

  int begin(const std::vector<EnvelopeJob> &jobs, uint32_t seed_state, int nsamples) override
  {
    jobs_ = jobs;
    const size_t nj = jobs.size();
    first_.assign(nj + 1, 0);
    for (size_t j = 0; j < nj; ++j) first_[j + 1] = first_[j] + (int64_t) jobs[j].req->size();
    const int64_t nreg = first_[nj];
    nreg_ = nreg; nlaunched_ = 0;
    launched_.assign((size_t) nreg, 0);
    if (nreg == 0) return P7X_OK;
    P7X_HIP(hipSetDevice(db_->device));
    EnsBuffers *eb = nullptr;
    {
      EnsPool &ep = ens_pool();
      std::lock_guard<std::mutex> lk(ep.mu);
      for (size_t i = 0; i < ep.all.size() && !eb; ++i) if (!ep.busy[i] && ep.all[i]->device == db_->device) { ep.busy[i] = 1; eb = ep.all[i]; }
      if (!eb) { eb = new EnsBuffers(); eb->device = db_->device; ep.all.push_back(eb); ep.busy.push_back(1); }
    }
    lease_ = eb;
    if (!eb->stream) {
      int least = 0, greatest = 0;
      P7X_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      P7X_HIP(hipStreamCreateWithPriority(&eb->stream, hipStreamNonBlocking, least));
    }
    // which regions the device takes: every one whose records fit the workspace budget, and whose model the kernels cover
    size_t free_b = 0, total_b = 0;
    size_t budget = (size_t) 16 << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 4 < budget) budget = std::max(free_b / 4, eb->work_bytes);
    std::vector<DevProfile *> dps(nj, nullptr);
    std::vector<EnsRegion> regs;
    std::vector<int64_t> reg_global;          // launched region -> global request number
    std::vector<EnsJob> ejobs(nj);
    std::map<int, std::vector<int>> by_class;
    int64_t ncells = 0, nrows = 0;
    int maxM = 1;
    size_t rft_floats = 0;
    std::vector<size_t> rft_at(nj, 0);
    const int dom_cap = nsamples * 4;
    for (size_t j = 0; j < nj; ++j) {
      EnsJob &ej = ejobs[j];
      std::memset(&ej, 0, sizeof ej);
      ej.reg_first = (int) regs.size();
      if (jobs[j].req->empty()) continue;
      const Profile &p = jobs[j].om->p;
      int st = get_dev_profile(jobs[j].om, ctx_, &dps[j]);
      if (st != P7X_OK) return st;
      ej.M = p.M; ej.C = dps[j]->vitC; ej.K = p.K; ej.Kp = p.Kp; ej.nrows = p.Kp + 1; ej.Q = p.Q4();
      ej.trans = dps[j]->fwd_trans; ej.emis = dps[j]->fwd_emis;
      if (ej.C <= 0 || p.Kp > 31) continue;                       // left to the host
      for (size_t r = 0; r < jobs[j].req->size(); ++r) {
        const EnvelopeRequest &rq = (*jobs[j].req)[r];
        const int t = (*jobs[j].targets)[(size_t) rq.item];
        const int Lr = rq.j - rq.i + 1;
        const int64_t cells = (int64_t) (Lr + 1) * (p.M + 1);
        if ((size_t) (ncells + cells) * 40 + (size_t) (nrows + Lr + 1) * 32 > budget) continue;
        EnsRegion er;
        er.sq = db_->h_off[t] + (rq.i - 1); er.cell0 = ncells; er.row0 = nrows; er.dom0 = (int64_t) regs.size() * dom_cap;
        er.Lr = Lr; er.L = db_->h_len[t]; er.job = (int) j; er.pad = 0;
        ncells += cells; nrows += Lr + 1;
        by_class[ej.C].push_back((int) regs.size());
        launched_[(size_t) (first_[j] + (int64_t) r)] = 1;
        reg_global.push_back(first_[j] + (int64_t) r);
        regs.push_back(er);
      }
      ej.nreg = (int) regs.size() - ej.reg_first;
      if (ej.nreg > 0) { maxM = std::max(maxM, p.M); rft_at[j] = rft_floats; rft_floats += (size_t) (p.M + 1) * 32; }
    }
    const int nl = (int) regs.size();
    nlaunched_ = nl; reg_global_ = reg_global;
    if (nl == 0) return P7X_OK;
    // workspace
    const size_t o_cells = 0, o_md = align256(o_cells + (size_t) ncells * 32), o_rows = align256(o_md + (size_t) ncells * 8);
    const size_t work_bytes = align256(o_rows + (size_t) nrows * 32);
    int st = grow_device(ctx_, eb->work, eb->work_bytes, work_bytes, (size_t) 256 << 20);
    if (st != P7X_OK) return st;
    // inputs: jobs | regions | class lists | rft tables | degeneracy matrices (one per job that has regions)
    const size_t o_jobs = 0, o_regs = align256(o_jobs + nj * sizeof(EnsJob)), o_lists = align256(o_regs + (size_t) nl * sizeof(EnsRegion));
    const size_t o_rft = align256(o_lists + (size_t) nl * 4), o_degen = align256(o_rft + rft_floats * 4);
    const size_t in_bytes = align256(o_degen + nj * 1024);
    if ((st = grow_pinned(eb->h_in, eb->h_in_cap, in_bytes, (size_t) 1 << 20)) != P7X_OK) return st;
    if ((st = grow_device(ctx_, eb->d_in, eb->d_in_cap, in_bytes, (size_t) 1 << 20)) != P7X_OK) return st;
    // outputs: ndom | status | dom records | null2 accumulators
    o_ndom_ = 0; o_status_ = align256((size_t) nl * 4); o_dom_ = align256(o_status_ + (size_t) nl * 4);
    o_n2_ = align256(o_dom_ + (size_t) nl * dom_cap * 5 * 4);
    const size_t out_bytes = align256(o_n2_ + (size_t) nrows * 4);
    if ((st = grow_device(ctx_, eb->d_out, eb->d_out_cap, out_bytes, (size_t) 4 << 20)) != P7X_OK) return st;
    if ((st = grow_pinned(eb->h_out, eb->h_out_cap, out_bytes, (size_t) 4 << 20)) != P7X_OK) return st;
    float *h_rft = reinterpret_cast<float *>(eb->h_in + o_rft);
    for (size_t j = 0; j < nj; ++j) {
      EnsJob &ej = ejobs[j];
      if (ej.nreg <= 0) continue;
      const Profile &p = jobs[j].om->p;
      float *t = h_rft + rft_at[j];
      std::memset(t, 0, (size_t) (p.M + 1) * 32 * 4);
      for (int x = 0; x < p.K; ++x) { const float *r = p.rf_.data() + (size_t) x * (p.M + 1); for (int k = 1; k <= p.M; ++k) t[(size_t) k * 32 + x] = r[k]; }
      unsigned char *dg = eb->h_in + o_degen + j * 1024;
      std::memset(dg, 0, 1024);
      const Alphabet &abc = Alphabet::get(p.abc_type);
      for (int x = 0; x < p.Kp; ++x) for (int y = 0; y < p.K; ++y) dg[x * 32 + y] = abc.degen[x][y];
      ej.rft = reinterpret_cast<const float *>(eb->d_in + o_rft) + rft_at[j];
      ej.degen = eb->d_in + o_degen + j * 1024;
    }
    std::memcpy(eb->h_in + o_jobs, ejobs.data(), nj * sizeof(EnsJob));
    std::memcpy(eb->h_in + o_regs, regs.data(), (size_t) nl * sizeof(EnsRegion));
    int32_t *h_lists = reinterpret_cast<int32_t *>(eb->h_in + o_lists);
    std::vector<std::pair<int, std::pair<int, int>>> runs;       // class, (first, count)
    { int at = 0; for (auto &kv : by_class) { std::copy(kv.second.begin(), kv.second.end(), h_lists + at); runs.push_back({ kv.first, { at, (int) kv.second.size() } }); at += (int) kv.second.size(); } }
    hipStream_t s = eb->stream;
    P7X_HIP(hipMemcpyAsync(eb->d_in, eb->h_in, in_bytes, hipMemcpyHostToDevice, s));
    EnsArgs a{};
    a.jobs = reinterpret_cast<const EnsJob *>(eb->d_in + o_jobs);
    a.regions = reinterpret_cast<const EnsRegion *>(eb->d_in + o_regs);
    a.nregions = nl; a.dsq = db_->d_dsq;
    a.cells = reinterpret_cast<ChoiceCell *>(eb->work + o_cells);
    a.md = reinterpret_cast<float2 *>(eb->work + o_md);
    a.rows = reinterpret_cast<ChoiceRow *>(eb->work + o_rows);
    a.seed_x = seed_state; a.nsamples = nsamples;
    a.n2acc = reinterpret_cast<float *>(eb->d_out + o_n2_);
    a.dom = reinterpret_cast<int32_t *>(eb->d_out + o_dom_); a.dom_cap = dom_cap;
    a.out_ndom = reinterpret_cast<int32_t *>(eb->d_out + o_ndom_); a.out_status = reinterpret_cast<int32_t *>(eb->d_out + o_status_);
    {   // accumulators in LDS for regions that fit beside the node counts in 60 KiB
      const long words = 60 * 256 - (((maxM + 2 + 31) & ~31) + 32 + 128 + 1 + 15);
      a.n2_lds_cap = (int) std::max(0L, words);
    }
    const int32_t *d_lists = reinterpret_cast<const int32_t *>(eb->d_in + o_lists);
    for (const auto &run : runs)
      if ((st = ens_forward_launch(run.first, a, d_lists + run.second.first, run.second.second, s)) != P7X_OK) return st;
    if ((st = ens_walk_launch(a, maxM, s)) != P7X_OK) return st;
    P7X_HIP(hipMemcpyAsync(eb->h_out, eb->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    regs_ = std::move(regs); dom_cap_ = dom_cap;
    return P7X_OK;
  }

  int wait(std::vector<std::vector<EnsembleResult>> &res) override
  {
    res.assign(jobs_.size(), {});
    for (size_t j = 0; j < jobs_.size(); ++j) res[j].assign(jobs_[j].req->size(), EnsembleResult{});
    if (nlaunched_ == 0) return P7X_OK;
    P7X_HIP(hipSetDevice(db_->device));
    P7X_HIP(hipStreamSynchronize(lease_->stream));
    const unsigned char *h = lease_->h_out;
    const int32_t *ndom = reinterpret_cast<const int32_t *>(h + o_ndom_), *status = reinterpret_cast<const int32_t *>(h + o_status_);
    const int32_t *dom = reinterpret_cast<const int32_t *>(h + o_dom_);
    const float *n2 = reinterpret_cast<const float *>(h + o_n2_);
    for (int l = 0; l < nlaunched_; ++l) {
      const int64_t g = reg_global_[(size_t) l];
      const size_t j = (size_t) (std::upper_bound(first_.begin(), first_.end(), g) - first_.begin()) - 1;
      EnsembleResult &e = res[j][(size_t) (g - first_[j])];
      e.status = status[l]; e.ndom = ndom[l];
      e.dom = dom + (size_t) regs_[(size_t) l].dom0 * 5;
      e.n2 = n2 + regs_[(size_t) l].row0;
    }
    return P7X_OK;
  }

private:
  DeviceCtx *ctx_; const p7x_seqdb *db_;
  std::vector<EnvelopeJob> jobs_;
  std::vector<int64_t> first_, reg_global_;
  std::vector<char> launched_;
  std::vector<EnsRegion> regs_;
  int64_t nreg_ = 0; int nlaunched_ = 0, dom_cap_ = 0;
  EnsBuffers *lease_ = nullptr;
  size_t o_ndom_ = 0, o_status_ = 0, o_dom_ = 0, o_n2_ = 0;
};

std::unique_ptr<EnsembleRunner> make_device_ensemble_runner(DeviceCtx *ctx, const p7x_seqdb *db)
{
  return std::make_unique<DeviceEnsembleRunner>(ctx, db);
}

} // namespace p7x

// Test seam: the ensemble of region i..j of one target, as the device samples it (use_device != 0) or as the host twin does:
// the sampled domains (sample, sqfrom, sqto inside the region, hmmfrom, hmmto; a sample's domains first to last) and the
// per-residue sums of the null2 odds ratios, before the logarithm and the clustering.
extern "C" int p7x_debug_ensemble(const p7x_oprofile *om, const p7x_seqdb *db, int64_t target, int32_t i, int32_t j, uint32_t seed,
                                  int use_device, int32_t *ndom, int32_t *dom, int32_t dom_cap, float *n2, int32_t *status)
{
  using namespace p7x;
  if (!om || !db || !ndom || !dom || !n2 || !status || target < 0 || target >= db->n || i < 1 || j < i || j > db->h_len[(size_t) target]) {
    set_error("p7x_debug_ensemble: bad arguments"); return P7X_EINVAL;
  }
  const int Lr = j - i + 1;
  std::vector<EnvelopeRequest> req{ EnvelopeRequest{ 0, i, j } };
  std::vector<int32_t> targets{ (int32_t) target };
  *ndom = 0; *status = -1;
  if (use_device) {
    DeviceCtx *ctx = nullptr;
    int st = get_ctx(db->device, &ctx);
    if (st != P7X_OK) return st;
    auto runner = make_device_ensemble_runner(ctx, db);
    std::vector<EnvelopeJob> jobs{ EnvelopeJob{ om, &req, &targets } };
    if ((st = runner->begin(jobs, fast_rng_state(seed), 200)) != P7X_OK) return st;
    std::vector<std::vector<EnsembleResult>> res;
    if ((st = runner->wait(res)) != P7X_OK) return st;
    const EnsembleResult &e = res[0][0];
    *status = e.status; *ndom = e.ndom;
    if (e.status != 0 || !e.dom || !e.n2) return P7X_OK;
    int n = 0;
    for (int a = 0; a < e.ndom; ) {
      int b = a;
      while (b < e.ndom && e.dom[(size_t) b * 5] == e.dom[(size_t) a * 5]) ++b;
      for (int d = b - 1; d >= a; --d, ++n) if (n < dom_cap) std::memcpy(dom + (size_t) n * 5, e.dom + (size_t) d * 5, 20);
      a = b;
    }
    std::memcpy(n2, e.n2, (size_t) (Lr + 1) * 4);
    return P7X_OK;
  }
  EnsembleRaw raw;
  EnsembleResult e; e.status = -1; e.raw_out = &raw;
  DomainDefResult dd;
  const int L = db->h_len[(size_t) target];
  dd.n2sc.assign((size_t) L + 1, 0.0f);
  std::vector<Domain> out;
  std::vector<EnvelopeRequest> defer2;
  MultiRegionState state;
  const uint8_t *dsq1 = db->h_dsq.data() + db->h_off[(size_t) target] - 1;
  const int st = domaindef_multi_region(om->p, dsq1, L, i, j, seed, true, state, dd, out, &defer2, 0, &e);
  if (st != P7X_OK) return st;
  *status = 0; *ndom = (int32_t) (raw.dom.size() / 5);
  std::memcpy(dom, raw.dom.data(), std::min(raw.dom.size(), (size_t) dom_cap * 5) * 4);
  std::memcpy(n2, raw.n2.data(), (size_t) (Lr + 1) * 4);
  return P7X_OK;
}
