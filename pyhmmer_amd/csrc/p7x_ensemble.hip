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
  const TransView<false> tr{ reinterpret_cast<const float4 *>(job.trans), 0 };     // read where they lie (L2)
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
  { const F8 t = tr.at((C - 1) * 64 + lane); p_md0 = dpp_shr1f(t.md, 0.0f); p_dd0 = dpp_shr1f(t.dd, 0.0f); }
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
        const F8 t = tr.at(c * 64 + lane);
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
#ifdef P7X_ENS_PROFILE
// build-time experiment (-DP7X_ENS_PROFILE): core-clock cycles the walk kernel's wavefronts spent in [0] C / J runs, [1] select_e,
// [2] the core walk, [3] finishing domains; [4] core steps, [5] samples, [6] record-cache misses, [7] regions
__device__ unsigned long long g_ens_prof[8];
#define P7X_ENS_STAMP(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_ens_prof[slot], now_ - stamp_); stamp_ = now_; } while (0)
#define P7X_ENS_COUNT(slot, n) do { if (lane == 0) atomicAdd(&g_ens_prof[slot], (unsigned long long) (n)); } while (0)
#else
#define P7X_ENS_STAMP(slot) do { } while (0)
#define P7X_ENS_COUNT(slot, n) do { } while (0)
#endif
namespace {
// inclusive prefix sum over the wavefront, in double (two 32-bit DPP moves per step; the scan tree of wave_sum_f32)
__device__ __forceinline__ double dpp_f64(double v, const int ctrl_sel)
{
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  int lo = (int) (unsigned) u, hi = (int) (unsigned) (u >> 32);
  switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false); break;
    case 3: lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, false); break;
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false); break;
  }
  return __builtin_bit_cast(double, ((unsigned long long) (unsigned) hi << 32) | (unsigned long long) (unsigned) lo);
}
__device__ __forceinline__ double wave_scan_f64(double v)
{
  v = v + dpp_f64(v, 0); v = v + dpp_f64(v, 1); v = v + dpp_f64(v, 2); v = v + dpp_f64(v, 3);
  v = v + dpp_f64(v, 4); v = v + dpp_f64(v, 5);
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) u, l), hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}
// Near-threshold guard (status bit 6).  The thresholds come from a Forward matrix summed in the device's lane-chunk order;
// upstream sums in its four stripes, so a threshold may lie a few units of 2^-24 (relative) elsewhere there, and a deviate
// that falls between the two takes another path -- and with it every later sample of the region.  Every comparison the walk
// actually takes that is closer than the guard flags the region, and the host stage samples flagged regions itself from a
// Forward matrix in upstream's order (domaindef_multi_region).  T == 0 stands for "never" or, as a fallback, "always": the
// other order may hold a threshold just inside either end of the deviate's range there.
__device__ __forceinline__ bool near_thr(uint32_t x, uint32_t T, uint32_t G)
{
  if (T == 0u) return x <= G || x >= ~G;
  return (x > T ? x - T : T - x) <= G;
}
// a 16-byte record through the vector memory path (in-order returns: a load issued a step ahead stays in flight while the
// current one is consumed; scalar loads return out of order and would be waited for together)
__device__ __forceinline__ uint4 vload16(const void *base, unsigned byte_off)
{
  asm volatile("" : "+v"(byte_off));
  return *reinterpret_cast<const uint4 *>(static_cast<const unsigned char *>(base) + byte_off);
}

// select_e exactly as the reference sums it: one double accumulator, the cells in the striped visiting order (q outer; four
// match cells, then four delete cells).  Only the rare draw that the lane-parallel version below cannot decide comes here.
__device__ __noinline__ int select_e_serial(const float2 *mdr, int M, int Q, float norm, double roll)
{ // returns node | delete << 30, or -1
  double sum = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int q = 0; q < Q; ++q) {
      for (int z = 0; z < 4; ++z) { const int kk = z * Q + q + 1; sum += (double) (kk <= M ? mdr[kk].x * norm : 0.0f); if (roll < sum) return kk; }
      for (int z = 0; z < 4; ++z) { const int kk = z * Q + q + 1; sum += (double) (kk <= M ? mdr[kk].y * norm : 0.0f); if (roll < sum) return kk | (1 << 30); }
    }
    if (sum < 0.99) return -1;
  }
  return -1;
}
} // namespace

// LDS: visit counts per node (p7_Null2_ByTrace's usage counts of one domain), the null2 vector of the domain just
// finished and its four per-stripe partial sums; then, as far as they fit, the rows' choice records, the region's residues,
// the per-residue accumulators (regions of up to 512 residues keep them in registers), the profile's residue-minor match
// odds, a few Forward rows for select_e, and the record caches of the core walk.
__global__ void __launch_bounds__(64) ens_walk_kernel(const EnsArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // One wavefront walks a region's 200 samples, a serial chain from the first deviate to the last: whenever it can issue
  // it should, ahead of the throughput kernels' wavefronts it shares a SIMD with.
#ifndef P7X_ENS_NO_SETPRIO
  __builtin_amdgcn_s_setprio(3);
#endif
  const int lane = threadIdx.x;
  const int r = blockIdx.x;
  const EnsRegion reg = a.regions[r];
  const EnsJob job = a.jobs[reg.job];
  const int M = job.M, Mrow = M + 1, Lr = reg.Lr, Q = job.Q, K = job.K, Kp = job.Kp;
  uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);                          // [M + 2]
  float *n2v = reinterpret_cast<float *>(cnt + ((M + 2 + 31) & ~31));          // [32]
  float *accz = n2v + 32;                                                      // [4][32]
  int *mdtag = reinterpret_cast<int *>(accz + 128);                            // [8]
  size_t used = (size_t) (((M + 2 + 31) & ~31) + 32 + 128 + 8) * 4;
  const size_t cap = (size_t) a.lds_bytes - 4096;               // the record caches' minimum stays free
  const ChoiceRow *__restrict__ rows = a.rows + reg.row0;
  // rows' records C / J / B (16 bytes each) and select_e's factor 1 / xE(i)
  const bool rows_in_lds = used + (size_t) (Lr + 1) * 20 + 16 <= cap;
  uint4 *rowl = reinterpret_cast<uint4 *>(smem + used);
  float *norml = reinterpret_cast<float *>(rowl + (Lr + 1));
  if (rows_in_lds) used += ((size_t) (Lr + 1) * 20 + 15) & ~(size_t) 15;
  // the region's residues (the finished domains look their odds up by residue)
  const uint8_t *__restrict__ sqg = a.dsq + reg.sq;
  const bool sq_in_lds = used + (size_t) Lr + 16 <= cap;
  uint8_t *sql = smem + used;
  if (sq_in_lds) { for (int q = lane; q < Lr; q += 64) sql[q] = sqg[q]; used += ((size_t) Lr + 15) & ~(size_t) 15; }
  // accumulators: lane l keeps residues l + 1 + 64 j, j < 8, in registers when the region is short enough
  constexpr int NR = 8;
  const bool n2_in_regs = Lr <= 64 * NR && sq_in_lds;
  const bool n2_in_lds = !n2_in_regs && used + (size_t) (Lr + 1) * 4 + 16 <= cap;
  float *n2g = a.n2acc + reg.row0;
  float *n2 = n2_in_lds ? reinterpret_cast<float *>(smem + used) : n2g;
  if (n2_in_lds) used += (size_t) (((Lr + 1) + 3) & ~3) * 4;
  // match odds
  const bool rft_in_lds = used + (size_t) (M + 1) * 128 <= cap;
  const float *rft = rft_in_lds ? reinterpret_cast<const float *>(smem + used) : job.rft;
  if (rft_in_lds) {
    float4 *dst = reinterpret_cast<float4 *>(smem + used);
    const float4 *src = reinterpret_cast<const float4 *>(job.rft);
    for (int q = lane; q < (M + 1) * 8; q += 64) dst[q] = src[q];
    used += (size_t) (M + 1) * 128;
  }
  // eight Forward rows (M and D cells) for select_e: a sample's domains end where the last sample's did, give or take
  const bool md_in_lds = used + (size_t) 8 * Mrow * 8 + 16 <= cap;
  float2 *mdl = reinterpret_cast<float2 *>(smem + used);
  if (md_in_lds) used += ((size_t) 8 * Mrow * 8 + 15) & ~(size_t) 15;
  // the record caches of the core walk (see there): 16-byte entries, powers of two; the match cells' gets about three
  // quarters of what is left (at most 4096 entries), the insert / delete cells' the rest (at most 1024)
  used = (used + 15) & ~(size_t) 15;
  const size_t left = (size_t) a.lds_bytes - used;
  int mbits = 7, idbits = 6;
  while (mbits < 12 && ((size_t) 32 << mbits) <= left - left / 4) ++mbits;
  while (idbits < 10 && ((size_t) 16 << mbits) + ((size_t) 32 << idbits) <= left) ++idbits;
  uint4 *mcache = reinterpret_cast<uint4 *>(smem + used);
  const unsigned idbase = 1u << mbits;                                                 // the insert / delete cache follows the match cache
  for (int q = lane; q < (1 << mbits) + (1 << idbits); q += 64) mcache[q] = make_uint4(0u, 0u, 0u, 0u);
  const bool cache_ok = (size_t) (Lr + 1) * (size_t) Mrow < ((size_t) 1 << 27) - 2;      // cell numbers fit the 27-bit tags
  const bool by_row = Lr < (1 << (mbits - 2));                                         // every row has its own four entries
  if (rows_in_lds) for (int q = lane; q <= Lr; q += 64) { rowl[q] = *reinterpret_cast<const uint4 *>(rows[q].x); norml[q] = bitsf(rows[q].e[0]); }
  for (int k = lane; k < M + 2; k += 64) cnt[k] = 0;
  if (lane < 8) mdtag[lane] = -1;
  if (!n2_in_regs) { for (int pos = lane; pos <= Lr; pos += 64) n2[pos] = 0.0f; if (!n2_in_lds) own_stores_visible(); }
  float acc[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) acc[j] = 0.0f;
  // esl_abc_FAvgScVec: which canonical residues residue code <lane> stands for
  uint32_t degen_mask = 0;
  if (lane > K && lane <= Kp - 3) for (int y = 0; y < K; ++y) if (job.degen[lane * 32 + y]) degen_mask |= 1u << y;
  __syncthreads();
  const ChoiceCell *cells = a.cells + reg.cell0;
  const float2 *__restrict__ md = a.md + reg.cell0;
  int32_t *dom = a.dom + reg.dom0 * 5;
  const int dom_cap = reg.dom_cap;
  uint32_t x = a.seed_x;
  int ndom = 0, status = 0;
  const uint32_t G = a.guard;
  const double Ge = (double) G / 4294967296.0;
  bool near_any = false;
  const int step_cap = 4 * (Lr + M) + 64;
  uint32_t jumpA = 1u, jumpC = 0u;                                 // lane + 1 draws of x <- 69069 x + 1 in one step
  for (int l = 0; l <= lane; ++l) { jumpA *= 69069u; jumpC = jumpC * 69069u + 1u; }
  auto row_rec = [&](int i) -> uint4 { return rows_in_lds ? rowl[i] : *reinterpret_cast<const uint4 *>(rows[i].x); };
  const unsigned rmask = (1u << (mbits - 2)) - 1u, mshift = 32u - (unsigned) mbits, idshift = 32u - (unsigned) idbits;

#ifdef P7X_ENS_PROFILE
  unsigned long long stamp_ = __builtin_readcyclecounter();
  P7X_ENS_COUNT(7, 1);
#endif
  for (int t = 0; t < a.nsamples && status == 0 && !near_any; ++t) {
    P7X_ENS_COUNT(5, 1);
    int i = Lr, k = 0, st = tC;
    int hi = Lr;                        // residues hi+1 .. Lr have received this sample's contribution
    int steps = 0;
    bool running = true;
    while (running) {
      if (++steps > step_cap) { status |= 1; break; }
      // the walk's state is the same in every lane: keep it in scalar registers (scalar branches, one address per load)
      st = rfl(st); i = rfl(i); k = rfl(k); x = (uint32_t) rfl((int) x);
      if (st == tC || st == tJ) {
        // A run of C (or J) states: row i-l decides with the l-th deviate from here, every choice of the run is independent
        // of the others, so 64 rows are looked at together -- lane l jumps the generator l draws ahead (x -> A_l x + C_l)
        // and the first lane that leaves for E ends the run.
        if (i < 1) { status |= 2; break; }
        const int row = i - lane;
        const uint32_t xl = jumpA * x + jumpC;                        // the state after lane + 1 draws
        bool leave = false, nearv = false;
        if (row >= 1) {
          const uint4 rw = row_rec(row);
          const int c = (st == tC) ? choice_pick_pair(rw.x, rw.w & 3u, xl) : choice_pick_pair(rw.y, (rw.w >> 2) & 3u, xl);
          leave = c != 0;
          nearv = G != 0u && near_thr(xl, (st == tC) ? rw.x : rw.y, G);
        }
        const unsigned long long lv = __ballot(leave);
        const int nrows = min(64, i);                                 // rows this look covers
        {   // the rows the run really decides: up to and including the first that leaves
          const int lastl = lv != 0ull ? (int) __builtin_ctzll(lv) : nrows - 1;
          if (__ballot(nearv && lane <= lastl) != 0ull) near_any = true;
        }
        if (lv != 0ull) {
          const int l = (int) __builtin_ctzll(lv);
          x = (uint32_t) __builtin_amdgcn_readlane((int) xl, l);
          i -= l; st = tE;
        } else {
          x = (uint32_t) __builtin_amdgcn_readlane((int) xl, nrows - 1);
          i -= nrows;
          if (i < 1) { status |= 2; break; }                          // the run reached row 0 without an E: no such trace
        }
        P7X_ENS_STAMP(0);
        continue;
      }
      if (st != tE) { status |= 32; break; }
      // ---- E(i): select_e, then the domain's core walk back to its B state
      {
        x = lcg_next(x);
        const double roll = (double) x / 4294967296.0;
        const float norm = rows_in_lds ? norml[i] : bitsf(rows[i].e[0]);
        const float2 *mdg = md + (size_t) i * Mrow;
        const int ms = i & 7;
        if (md_in_lds && rfl(mdtag[ms]) != i) {                     // this row's cells into one of the eight LDS rows
          for (int q = lane; q < Mrow; q += 64) mdl[ms * Mrow + q] = mdg[q];
          if (lane == 0) mdtag[ms] = i;
          __syncthreads();
        }
        // Lane l of block b holds the cell of rank 64 b + l in the visiting order: rank = 8 q + 4 [delete] + z, node z Q + q + 1.
        // Cumulative shares by a wavefront scan per block; the first rank whose cumulative share exceeds the deviate wins.  A
        // scan adds in another order than the reference's single accumulator (differences ~1e-13): a deviate closer than
        // 1e-9 to any cumulative share it was compared with is decided by select_e_serial instead.
        const int nblk = (8 * Q + 63) >> 6;
        double base = 0.0;
        int found = -1, ambiguous = 0;
        for (int b0 = 0; b0 < nblk && found < 0 && !ambiguous; b0 += 8) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int rank = ((b0 + u) << 6) + lane, q = rank >> 3, z = rank & 3, kk = z * Q + q + 1;
            float2 c = make_float2(0.0f, 0.0f);
            if (b0 + u < nblk && q < Q && kk <= M) { if (md_in_lds) c = mdl[ms * Mrow + kk]; else c = mdg[kk]; }
            v[u] = ((rank & 4) ? c.y : c.x) * norm;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (b0 + u < nblk && found < 0 && !ambiguous) {
              const double cum = base + wave_scan_f64((double) v[u]);
              const double dist = __builtin_fabs(cum - roll);
              if (G != 0u && __ballot(dist <= Ge) != 0ull) near_any = true;          // (cells behind the winner lie further above the deviate than it does)
              if (__ballot(dist < 1.0e-9) != 0ull) ambiguous = 1;
              else {
                const unsigned long long hit = __ballot(roll < cum);
                if (hit != 0ull) found = ((b0 + u) << 6) + (int) __builtin_ctzll(hit);
                else base = readlane_f64(cum, 63);
              }
            }
          }
        }
        if (found >= 0 && !ambiguous) { const int q = found >> 3, z = found & 3; k = z * Q + q + 1; st = (found & 4) ? tD : tM; }
        else {                                                            // also the second pass of a first pass that ended short
          const int code = select_e_serial(mdg, M, Q, norm, roll);
          if (code < 0) { status |= 4; break; }
          if (G != 0u && !ambiguous) near_any = true;                      // a first pass that ended short of the deviate: the sums' last bits decide
          k = code & 0x3fffffff; st = (code >> 30) ? tD : tM;
        }
      }
      k = rfl(k); st = rfl(st);
      P7X_ENS_STAMP(1);
      int dj = 0, dm_ = 0, di = 0, dk = 0;
      const int khi = k;
      // The core walk, a RUN at a time.  While the walk stays in one state it moves along a line -- M down the diagonal
      // (i-l, k-l), I up the column (i-l, k), D along the row (i, k-l) -- and the choice at the l-th cell of the line uses the
      // l-th deviate from here: lane l looks at that cell with the generator jumped l draws ahead, and the first lane whose
      // choice leaves the state ends the run.  A domain of a few hundred residues is a dozen runs instead of a few hundred
      // dependent steps.
      // The cells a region's samples visit are a narrow band around its alignments, visited again and again -- but between
      // two visits the filter kernels running beside this one have streamed gigabytes through L2, and a round trip to HBM
      // (and through the TLB: every cell of a line is another row, kilobytes away) is what a record then costs.  So the
      // records are cached in LDS, 16 bytes per entry: the match cells' in a cache indexed by row and node -- row i owns
      // entries 4 (i mod R) .. + 3, node k goes to k mod 4, so a walk that keeps to a band of four diagonals never evicts
      // its own cells (regions longer than R rows: hashed by cell number instead) -- and the insert / delete cells' in a
      // smaller one hashed by cell number.  The first sample fetches its cells from memory (all lanes of a run at once),
      // the other 199 mostly find them here.
      while (st != tB) {
        if (++steps > step_cap) { status |= 1; break; }
        if (i < 1 || k < 1) { status |= 2; break; }
        const bool isM = st == tM, isI = st == tI;
        const int ii = i - (isM || isI ? lane : 0), kk = k - (isI ? 0 : lane);
        const bool inside = ii >= 1 && kk >= 1;
        const unsigned tg = (unsigned) ii * (unsigned) Mrow + (unsigned) kk + 1u;         // cell number + 1
        const uint32_t xl = jumpA * x + jumpC;                                            // the state after lane + 1 draws
        const unsigned hsh = tg * 2654435761u;
        const unsigned slot = isM ? (by_row ? ((((unsigned) ii & rmask) << 2) | ((unsigned) kk & 3u)) : (hsh >> mshift)) : idbase + (hsh >> idshift);
        uint4 rec = mcache[slot];                                                         // (any lane: the index is in range)
        asm volatile("" : "+v"(rec.x), "+v"(rec.y), "+v"(rec.z), "+v"(rec.w));            // the whole entry, one read
        const bool missed = inside && (!cache_ok || (isM ? (rec.w >> 5) : rec.w) != tg);
        if (missed) {                                                                     // from memory (all missing lanes at once)
          const unsigned char *src = reinterpret_cast<const unsigned char *>(cells) + ((size_t) ii * Mrow + kk) * 32 + (isM ? 0 : 16);
          rec = *reinterpret_cast<const uint4 *>(src);
          rec.w = isM ? ((rec.w & 31u) | (tg << 5)) : tg;
        }
        int s1;
        bool nearv = false;
        if (isM) {
          const int fstate = (int) ((0x2316u >> ((rec.w & 3u) * 4u)) & 15u);              // fallback path 0..3 -> B, M, I, D
          s1 = xl < rec.x ? tB : (xl < rec.y ? tM : (xl < rec.z ? tI : fstate));
          nearv = G != 0u && (near_thr(xl, rec.x, G) || near_thr(xl, rec.y, G) || near_thr(xl, rec.z, G));
        } else {
          const uint32_t thr = isI ? rec.x : rec.y;
          const bool stay = isI ? (rec.z & 1u) != 0u : (rec.z & 4u) != 0u;
          s1 = xl < thr ? tM : (stay ? st : tM);
          nearv = G != 0u && near_thr(xl, thr, G);
        }
        if (!inside) s1 = 0;                                                              // off the matrix: leaves, and is an error if reached
        const unsigned long long leave = __ballot(s1 != st);
        const int l = leave ? (int) __builtin_ctzll(leave) : 63;                          // the run's last cell is lane l's
        if (__ballot(nearv && inside && lane <= l) != 0ull) near_any = true;
        if (cache_ok && missed && lane <= l) mcache[slot] = rec;                          // only cells the walk really visited
        P7X_ENS_COUNT(6, __builtin_popcountll(__ballot(missed && lane <= l)));
        if (st != tD && lane <= l && inside) atomicAdd(&cnt[kk], 1u);                     // emitting states: one visit each
        const int sl = __builtin_amdgcn_readlane(s1, l);
        if (leave && sl == 0) { status |= 2; break; }                                     // the walk ran off the matrix
        if (isM) {
          if (dj == 0) { dj = i; dm_ = k; }
          di = i - l; dk = k - l;
        }
        x = (uint32_t) __builtin_amdgcn_readlane((int) xl, l);
        const int adv = l + 1;
        if (st != tD) i -= adv;
        if (st != tI) k -= adv;
        if (leave) st = sl;
        i = rfl(i); k = rfl(k); st = rfl(st);
      }
      if (status) break;
      const int Ld = dj - di + 1;             // every residue di .. dj was emitted by exactly one M or I state of the domain
      P7X_ENS_STAMP(2);
      P7X_ENS_COUNT(4, Ld);
      const int klo = k + 1;                // the lowest node the domain touched: its first match state's
      // ---- B(i): the domain is complete -- residues di .. dj of the region, nodes dk .. dm_ (p7_trace_Index); its end points go
      // out for clustering, its null2 odds (p7_Null2_ByTrace over the visit counts) onto the residues di+1 .. dj
      if (dj == 0 || Ld < 1) { status |= 8; break; }
      if (ndom < dom_cap) {
        if (lane == 0) { int32_t *o = dom + (size_t) ndom * 5; o[0] = t; o[1] = di; o[2] = dj; o[3] = dk; o[4] = dm_; }
      } else status |= 16;                                              // more domains than the record holds: the host repeats the region
      ++ndom;
      __syncthreads();                                                  // the counts are in
      {
        const float nrm = (float) (1.0 / (double) (float) Ld);
        const int z = lane >> 4, xs = lane & 15;
        float acc0 = 0.0f, acc1 = 0.0f;
        const int lo = max(klo, 1);
        const int q0 = max(0, lo - 1 - z * Q), q1 = min(min(Q - 1, khi - 1 - z * Q), M - 1 - z * Q);
        int q = q0;
        for (; q + 3 <= q1; q += 4) {                                   // four nodes' loads in flight, then the chain
          const int kk = q + 1 + z * Q;
          const float w0 = (float) cnt[kk] * nrm, w1 = (float) cnt[kk + 1] * nrm, w2 = (float) cnt[kk + 2] * nrm, w3 = (float) cnt[kk + 3] * nrm;
          const float a0 = rft[(size_t) kk * 32 + xs], a1 = rft[(size_t) (kk + 1) * 32 + xs], a2 = rft[(size_t) (kk + 2) * 32 + xs], a3 = rft[(size_t) (kk + 3) * 32 + xs];
          const float b0 = rft[(size_t) kk * 32 + 16 + xs], b1 = rft[(size_t) (kk + 1) * 32 + 16 + xs], b2 = rft[(size_t) (kk + 2) * 32 + 16 + xs], b3 = rft[(size_t) (kk + 3) * 32 + 16 + xs];
          acc0 = acc0 + w0 * a0; acc1 = acc1 + w0 * b0;                 // an unvisited node adds w = 0: an exact no-op, as skipping it is
          acc0 = acc0 + w1 * a1; acc1 = acc1 + w1 * b1;
          acc0 = acc0 + w2 * a2; acc1 = acc1 + w2 * b2;
          acc0 = acc0 + w3 * a3; acc1 = acc1 + w3 * b3;
        }
        for (; q <= q1; ++q) {
          const int kk = q + 1 + z * Q;
          const float w = (float) cnt[kk] * nrm;
          acc0 = acc0 + w * rft[(size_t) kk * 32 + xs];
          acc1 = acc1 + w * rft[(size_t) kk * 32 + 16 + xs];
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
          for (int y = 0; y < K; ++y) if ((degen_mask >> y) & 1u) { res += n2v[y]; ++n; }
          v = res / (float) n; set = true;
        } else if (lane == K || lane == Kp - 2 || lane == Kp - 1) { v = 1.0f; set = true; }
        __syncthreads();
        if (set) n2v[lane] = v;
      }
      __syncthreads();
      if (n2_in_regs) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int pos = 1 + lane + 64 * j;
          if (pos > dj && pos <= hi) acc[j] += 1.0f;
          else if (pos > di && pos <= dj) acc[j] += n2v[sql[pos - 1]];
        }
      } else {
        for (int pos = dj + 1 + lane; pos <= hi; pos += 64) n2[pos] += 1.0f;
        for (int pos = di + 1 + lane; pos <= dj; pos += 64) n2[pos] += n2v[sq_in_lds ? sql[pos - 1] : sqg[pos - 1]];
        if (!n2_in_lds) own_stores_visible();
      }
      hi = di;
      for (int kk = max(klo, 0) + lane; kk <= khi; kk += 64) cnt[kk] = 0;
      __syncthreads();
      {
        x = lcg_next(x);
        const uint4 rw = row_rec(i);
        if (G != 0u && near_thr(x, rw.z, G)) near_any = true;
        if (choice_pick_pair(rw.z, (rw.w >> 4) & 3u, x) == 0) running = false;       // N: the rest of the trace is N ... N S
        else st = tJ;
      }
      P7X_ENS_STAMP(3);
    }
    if (status == 0) {
      if (n2_in_regs) {
#pragma unroll
        for (int j = 0; j < NR; ++j) if (1 + lane + 64 * j <= hi) acc[j] += 1.0f;
      } else {
        for (int pos = 1 + lane; pos <= hi; pos += 64) n2[pos] += 1.0f;
        if (!n2_in_lds) own_stores_visible();
        __syncthreads();
      }
    }
  }
  if (n2_in_regs) {
#pragma unroll
    for (int j = 0; j < NR; ++j) { const int pos = 1 + lane + 64 * j; if (pos <= Lr) n2g[pos] = acc[j]; }
    if (lane == 0) n2g[0] = 0.0f;
  } else if (n2_in_lds) for (int pos = lane; pos <= Lr; pos += 64) n2g[pos] = n2[pos];
  if (near_any) status |= 64;                    // too close to call: the host stage samples this region in upstream's order
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

static int ens_walk_launch(const EnsArgs &a, hipStream_t st)
{
  if (a.nregions <= 0) return P7X_OK;
  const size_t lds = (size_t) a.lds_bytes;
  if (lds > 64 * 1024) {
    static std::mutex mu; static std::map<int, size_t> granted_by_device;       // a per-device attribute of the kernel
    int dev = 0; P7X_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    size_t &granted = granted_by_device[dev];
    if (lds > granted) { P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ens_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) (128 * 1024))); granted = 128 * 1024; }
  }
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
  DeviceEnsembleRunner(DeviceCtx *ctx, const p7x_seqdb *db, float guard) : ctx_(ctx), db_(db)
  {
    const double g = (double) guard * 4294967296.0;
    guard_ = g <= 0.0 ? 0u : (g >= 1.0e9 ? 1000000000u : (uint32_t) g);
  }
  ~DeviceEnsembleRunner() override { if (lease_) { if (lease_->stream) (void) hipStreamSynchronize(lease_->stream); release_ens_buffers(lease_); } }

  // Any region may be sampled by the host workers instead (EnsembleResult::status != 0): a device-side failure here --
  // the workspace cannot grow because other host stages hold the memory, a launch is refused -- sends every region of
  // this call there instead of failing the batch and with it the whole search.
  int begin(const std::vector<EnvelopeJob> &jobs, uint32_t seed_state, int nsamples) override
  {
    const int st = begin_on_device(jobs, seed_state, nsamples);
    if (st == P7X_OK) return st;
    // Only a lack of memory (or the test seam that plays one) is a reason to sample on the host instead: a launch fault
    // or a sticky device error must surface, not turn into a slower search that looks healthy.
    if (st != P7X_EMEM && debug_opt(OPT_ENS_FAIL) <= 0) return st;
    if (debug_opt(OPT_TRACE_FINISH) > 0) std::fprintf(stderr, "[ens] device ensembles unavailable for this call (%s): its regions go to the host workers\n", p7x_last_error());
    if (lease_ && lease_->stream) (void) hipStreamSynchronize(lease_->stream);      // whatever was queued before the failure
    (void) hipGetLastError();
    std::fill(launched_.begin(), launched_.end(), (char) 0);
    nlaunched_ = 0;
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
  int begin_on_device(const std::vector<EnvelopeJob> &jobs, uint32_t seed_state, int nsamples)
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
      // a few wavefronts, each a long serial chain, and the host stage waits for them: ahead of the filter kernels' queues
      const int cst = create_tail_stream(ctx_, true, &eb->stream); if (cst != P7X_OK) return cst;
    }
    // which regions the device takes: every one whose records fit the workspace budget, and whose model the kernels cover
    // (a quarter of what is free -- up to eight host stages lease a workspace each, next to two envelope scorers apiece --
    // and all leases of the device together at most a quarter of its memory: a lease that would have to grow beyond that
    // keeps its size, and the regions that do not fit sample on the host)
    size_t free_b = 0, total_b = 0;
    size_t budget = (size_t) 16 << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      if (free_b / 4 < budget) budget = std::max(free_b / 4, eb->work_bytes);
      size_t pooled = 0;
      {
        EnsPool &ep = ens_pool();
        std::lock_guard<std::mutex> lk(ep.mu);
        for (const EnsBuffers *o : ep.all) if (o != eb && o->device == db_->device) pooled += o->work_bytes;
      }
      const size_t cap = total_b / 4;
      const size_t room = pooled < cap ? cap - pooled : 0;
      if (budget > std::max(room, eb->work_bytes)) budget = std::max(room, eb->work_bytes);
    }
    std::vector<DevProfile *> dps(nj, nullptr);
    std::vector<EnsRegion> regs;
    std::vector<int64_t> reg_global;          // launched region -> global request number
    std::vector<EnsJob> ejobs(nj);
    std::map<int, std::vector<int>> by_class;
    int64_t ncells = 0, nrows = 0, ndomrec = 0;
    int maxM = 1, maxLr = 1;
    size_t rft_floats = 0;
    std::vector<size_t> rft_at(nj, 0);
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
        er.sq = db_->h_off[t] + (rq.i - 1); er.cell0 = ncells; er.row0 = nrows; er.dom0 = ndomrec;
        er.Lr = Lr; er.L = db_->h_len[t]; er.job = (int) j;
        er.dom_cap = nsamples * std::min(64, 4 + 4 * Lr / (p.M + 16));   // a region whose samples average more domains than that goes back to the host
        ndomrec += er.dom_cap;
        maxLr = std::max(maxLr, Lr);
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
    if (debug_opt(OPT_ENS_FAIL) > 0) { set_error("ensemble workspace: failure requested by the test seam ens_fail"); return P7X_EMEM; }
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
    o_n2_ = align256(o_dom_ + (size_t) ndomrec * 5 * 4);
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
    a.seed_x = seed_state; a.nsamples = nsamples; a.guard = guard_;
    a.n2acc = reinterpret_cast<float *>(eb->d_out + o_n2_);
    a.dom = reinterpret_cast<int32_t *>(eb->d_out + o_dom_);
    a.out_ndom = reinterpret_cast<int32_t *>(eb->d_out + o_ndom_); a.out_status = reinterpret_cast<int32_t *>(eb->d_out + o_status_);
    {   // LDS of the walk kernel: node counts and the null2 scratch always; then, as far as 128 KiB go, the largest region's row
        // records, residues and accumulators, the longest model's odds table and eight of its Forward rows (each region
        // takes what fits, in that order), and the record caches of the core walk in the rest (48 KiB when there is room,
        // never less than 4 KiB)
      const size_t least = (size_t) (((maxM + 2 + 31) & ~31) + 32 + 128 + 8) * 4 + 64 + 4096;
      const size_t want = least + (size_t) (maxLr + 1) * 25 + 128 + (size_t) (maxM + 1) * (128 + 64) + 44 * 1024;
      // ... but no more than 32 KiB by default (option ens_lds_kb): a walk is one wavefront per workgroup, a latency chain that
      // needs no throughput, and what it holds is missing for the filter kernels' blocks -- three MSV blocks of a Pfam-sized
      // model take 100 of a CU's 160 KiB, and with 109 KiB gone to a walk only one of them fits.  Round 5, headline workload:
      // 19.0-20.0 TCUPS with the full 128 KiB, 20.3-21.0 with 16-40 KiB (profiles/r05_ens_lds.txt); the host stage even waits less
      // for the ensembles (23-30 ms instead of 30-36), because more walks are resident at once.
      size_t most = (size_t) 32 * 1024;
      if (debug_opt(OPT_ENS_LDS_KB) > 0) most = (size_t) debug_opt(OPT_ENS_LDS_KB) * 1024;
      a.lds_bytes = (int) std::max(least, std::min(want, most));
    }
    const int32_t *d_lists = reinterpret_cast<const int32_t *>(eb->d_in + o_lists);
    for (const auto &run : runs)
      if ((st = ens_forward_launch(run.first, a, d_lists + run.second.first, run.second.second, s)) != P7X_OK) return st;
    if ((st = ens_walk_launch(a, s)) != P7X_OK) return st;
    P7X_HIP(hipMemcpyAsync(eb->h_out, eb->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    regs_ = std::move(regs);
    return P7X_OK;
  }

  DeviceCtx *ctx_; const p7x_seqdb *db_;
  uint32_t guard_ = 0;
  std::vector<EnvelopeJob> jobs_;
  std::vector<int64_t> first_, reg_global_;
  std::vector<char> launched_;
  std::vector<EnsRegion> regs_;
  int64_t nreg_ = 0; int nlaunched_ = 0;
  EnsBuffers *lease_ = nullptr;
  size_t o_ndom_ = 0, o_status_ = 0, o_dom_ = 0, o_n2_ = 0;
};

std::unique_ptr<EnsembleRunner> make_device_ensemble_runner(DeviceCtx *ctx, const p7x_seqdb *db, float guard)
{
  return std::make_unique<DeviceEnsembleRunner>(ctx, db, guard);
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
    auto runner = make_device_ensemble_runner(ctx, db, 0.0f);        // the seam compares the device's samples with the host twin's in the device's order
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

#ifdef P7X_ENS_PROFILE
extern "C" int p7x_debug_ens_profile(unsigned long long *out8)
{
  unsigned long long zero[8] = { 0 };
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(p7x::g_ens_prof), sizeof zero) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(p7x::g_ens_prof), zero, sizeof zero) != hipSuccess;
}
#endif
