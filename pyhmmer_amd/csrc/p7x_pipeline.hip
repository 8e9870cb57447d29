// p7x_pipeline.hip -- the filter cascade of p7_Pipeline on the device, plus the C-ABI entry points.
//
// Control flow restates upstream p7_pipeline.c:p7_Pipeline (reference p7_pipeline.pxd:130; called from
// Pipeline._search_loop, plan7.pyx:6393-6453) for a whole DigitalSequenceBlock at once:
//   null1 -> MSV -> P<=F1 -> bias filter -> P<=F1 -> [P>F2: Viterbi -> P<=F2] -> Forward -> P<=F3
//   -> Backward -> (host) domain definition -> hit.
// Every stage is a kernel over a device-resident work list; the lists are built with atomics and their
// lengths are read by the next stage from device memory, so the cascade runs without host round trips.
#include "p7x_wave.hpp"
#include "p7x_host.hpp"
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

namespace p7x {

// ---------------------------------------------------------------------------- small per-target kernels
struct StageBufs {            // all indexed by slot unless noted
  int16_t *xJ;                // [ngroups*64]
  float *usc, *filtersc, *vfsc, *fwdsc;
  int32_t *xC;                // [nslots] by Viterbi work-list position
  float *fwd_by_item;         // [nslots] by Forward work-list position
  int32_t *list_bias, *list_vit, *list_fwd, *list_fin;
  uint8_t *stage;             // [nslots] last filter passed: 1 MSV, 2 bias, 3 Viterbi, 4 Forward (scan mode accounting)
  int *counters;              // [0] msv groups [1] n_bias(list_bias) [2] n_vit [3] n_fwd [4] n_fin [5..7] work counters
                              // [8] n_past_bias [9] n_past_vit(reach Forward) [14] Forward survivors inside the F3 guard band
};

struct StageParams {
  double F1, F2, F3;
  double F3_near;             // Forward survivors with P above this are listed for the host stage's F3 guard (list_bias is free by then)
  float mmu, mlambda, vmu, vlambda, ftau, flambda;
  float msv_reject_below;     // bit scores below this cannot pass F1 (the Gumbel tail is monotone): decide_msv skips the double-precision tail for them
  int do_bias;
  int base_b, tjb_unused; float scale_b;
  int base_w; float scale_w;
  int M;
};

__device__ __forceinline__ double d_gumbel_surv(double x, double mu, double lambda)
{
  const double y = lambda * (x - mu), ey = -exp(-y);
  return (fabs(ey) < 5e-9) ? -ey : 1 - exp(ey);
}
__device__ __forceinline__ double d_exp_surv(double x, double mu, double lambda) { return x < mu ? 1.0 : exp(-lambda * (x - mu)); }

// Append <value> to a device work list for every lane with <take>: one atomic per wavefront instead of one per lane
// (thousands of lanes finish the same stage at the same moment; same-address atomics serialise in L2).
// Must be reached by all lanes of the wavefront that are still in the kernel.
__device__ __forceinline__ void wave_append(int *counter, int32_t *list, bool take, int32_t value)
{
  const unsigned long long mask = __ballot(take);
  if (mask == 0) return;
  const int lane = (int) (threadIdx.x & 63);
  const int leader = __ffsll((long long) mask) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(mask));
  base = __shfl(base, leader);
  if (take) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = value;
}
__device__ __forceinline__ void wave_count(int *counter, bool take)
{
  const unsigned long long mask = __ballot(take);
  if (mask != 0 && (int) (threadIdx.x & 63) == __ffsll((long long) mask) - 1) atomicAdd(counter, __popcll(mask));
}

// What the per-target decision kernels of one lane read
struct DecideArgs {
  StageBufs b; StageParams p;
  const int32_t *slot_len; const uint8_t *tjb_tab; const float *null1_tab; const int16_t *xwmove_tab; const double *logtab;
  const uint8_t *dsq; const int64_t *slot_off; const float *eo;
  int64_t nslots;
};

// after MSV: usc, P1; survivors -> list_bias
__global__ void decide_msv_kernel(const ArgRef ref)
{
  const DecideArgs a = load_args<DecideArgs>(ref);
  const StageBufs &b = a.b; const StageParams &p = a.p;
  const int64_t s = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  bool take = false;
  if (s < a.nslots) {
    const int L = a.slot_len[s];
    const int xJ = b.xJ[s];
    float usc;
    if (xJ < 0) usc = __builtin_inff();
    else {
      usc = ((float) (xJ - (int) a.tjb_tab[L]) - (float) p.base_b);
      usc /= p.scale_b;
      usc -= 3.0f;
    }
    b.usc[s] = usc;
    const float seq_score = (float) ((double) (usc - a.null1_tab[L]) / kLog2);
    // 98 % of the targets end here: the survival function falls with the score, so a score a thousandth of a bit below the
    // one at which it equals F1 has P > F1 by a margin ten orders of magnitude above the double arithmetic's error, and the
    // two exponentials in double precision (most of this kernel's 0.66 ms per 7-query launch) are for the others only
    if (!(seq_score < p.msv_reject_below)) {
      const double P = d_gumbel_surv((double) seq_score, (double) p.mmu, (double) p.mlambda);
      take = !(P > p.F1);
    }
    b.stage[s] = take ? 1 : 0;
  }
  wave_append(&b.counters[1], b.list_bias, take, (int32_t) s);
}

// (float) log((double) x) of a positive normal float, as the reference's scaled Forward takes it at every residue:
// x = 2^e m, m = c (1 + r) with c the centre of m's 1/128 interval, log x = e ln 2 + log c + log(1 + r), |r| <= 2^-8,
// five terms of the series in double precision (absolute error 4e-15) at a tenth of the library logarithm's instructions.
// That error matters only where the double lies within it of the midpoint between two floats (six of the 117,440,512
// floats of [2^-7, 2^7) rounded the other way): those arguments -- one in a million -- take the library call, and every
// float of that range then gives the host library's float (tests/test_gpu_filters.py).  tab: [128][2] = 1/c, log c.
__device__ __forceinline__ float log_of_float(float x, const double *tab)
{
  const uint32_t u = __float_as_uint(x);
  if (u - 0x00800000u >= 0x7f000000u) return (float) log((double) x);       // zero, denormal, inf, nan, negative: the library's answer
  const int e = (int) (u >> 23) - 127;
  const uint32_t mant = u & 0x7fffffu;
  const int i = (int) (mant >> 16);
  const double m = (double) __uint_as_float(mant | 0x3f800000u);
  const double r = fma(m, tab[2 * i], -1.0);
  const double q = r * fma(r, fma(r, fma(r, fma(r, 0.2, -0.25), 1.0 / 3.0), -0.5), 1.0);
  const double y = fma((double) e, 0.693147180559945309417232121458, tab[2 * i + 1] + q);
  const float f = (float) y;
  // distance of y from the nearer rounding midpoint next to f, against the series' error bound (relative to |y|, with room)
  const uint32_t fb = __float_as_uint(f) & 0x7f800000u;
  if (fb > (25u << 23)) {
    const double half_ulp = (double) __uint_as_float(fb - (24u << 23));
    if (fabs(fabs(y - (double) f) - half_ulp) < 4e-14 * fabs(y) + 1e-17) return (float) log((double) x);
  }
  return f;
}

// test seam: log_of_float over an array (tests/test_gpu_filters.py sweeps every float of the range the bias filter sees)
__global__ void log_of_float_kernel(const float *in, float *out, size_t n, const double *tab)
{
  __shared__ double s_logtab[256];
  for (int z = (int) threadIdx.x; z < 256; z += (int) blockDim.x) s_logtab[z] = tab[z];
  __syncthreads();
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) out[i] = log_of_float(in[i], s_logtab);
}

// bias filter: esl_hmm_Forward on the 2-state composition HMM (p7_bg_FilterScore); survivors -> list_vit / list_fwd
__global__ void bias_kernel(const ArgRef ref)
{
  const DecideArgs a = load_args<DecideArgs>(ref);
  const StageBufs &b = a.b; const StageParams &p = a.p;
  const uint8_t *dsq = a.dsq; const float *eo = a.eo;
  __shared__ double s_logtab[256];
  __shared__ float s_eo[kTabRows * 2];
  for (int z = (int) threadIdx.x; z < 256; z += (int) blockDim.x) s_logtab[z] = a.logtab[z];
  for (int z = (int) threadIdx.x; z < kTabRows * 2; z += (int) blockDim.x) s_eo[z] = eo[z];
  __syncthreads();
  const int n = b.counters[1];
  for (int it0 = blockIdx.x * blockDim.x; it0 < n; it0 += gridDim.x * blockDim.x) {     // uniform trip count per wavefront
    const int it = it0 + (int) threadIdx.x;
    bool to_vit = false, to_fwd = false;
    int s = 0;
    if (it < n) {
    s = b.list_bias[it];
    const int L = a.slot_len[s];
    const float nullsc = a.null1_tab[L];
    float filtersc = nullsc;
    const float usc = b.usc[s];
    double P;
    bool pass = true;
    if (p.do_bias) {
      const uint8_t *sq = dsq + a.slot_off[s];
      const float p1 = (float) L / (float) (L + 1);
      const float L1 = (float) ((double) (float) p.M / 8.0);
      const float t00 = p1, t01 = 1.0f - p1, t10 = 1.0f / (L1 + 1.0f), t11 = L1 / (L1 + 1.0f);
      const float pi0 = (float) 0.999, pi1 = (float) 0.001;
      int x = sq[0];
      float dp0 = s_eo[x * 2] * pi0, dp1 = s_eo[x * 2 + 1] * pi1;
      float mx = 0.0f; mx = dp0 > mx ? dp0 : mx; mx = dp1 > mx ? dp1 : mx;
      dp0 /= mx; dp1 /= mx;
      float logsc = 0.0f;
      logsc += log_of_float(mx, s_logtab);
      // The recurrence is a short dependent chain per residue; a byte load per step would put a memory round trip
      // on it (~1 us x L).  Residues are fetched a block of sixteen steps ahead of the block being computed (the loads
      // of block b + 1 are in flight during the chain and the logs of block b); emission odds come from LDS.
      int xcur[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) xcur[j] = (1 + j < L) ? (int) sq[1 + j] : 0;
      for (int i0 = 1; i0 < L; i0 += 16) {
        const int nstep = min(16, L - i0);
        int xnext[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) xnext[j] = (i0 + 16 + j < L) ? (int) sq[i0 + 16 + j] : 0;
        float e0[16], e1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { e0[j] = s_eo[xcur[j] * 2]; e1[j] = s_eo[xcur[j] * 2 + 1]; }
        float mxs[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          mxs[j] = 1.0f;
          if (j < nstep) {
            float n0 = 0.0f; n0 += dp0 * t00; n0 += dp1 * t10; n0 *= e0[j];
            float n1 = 0.0f; n1 += dp0 * t01; n1 += dp1 * t11; n1 *= e1[j];
            mx = 0.0f; mx = n0 > mx ? n0 : mx; mx = n1 > mx ? n1 : mx;
            dp0 = n0 / mx; dp1 = n1 / mx;
            mxs[j] = mx;
          }
        }
        // the sixteen logs are independent of each other and of the chain above: the compiler interleaves them.  They
        // are still added to the score in sequence order.
        float lg[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) lg[j] = log_of_float(mxs[j], s_logtab);
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j < nstep) logsc += lg[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) xcur[j] = xnext[j];
      }
      float last = 0.0f; last += dp0 * 1.0f; last += dp1 * 1.0f;
      logsc += log_of_float(last, s_logtab);
      filtersc = logsc + (float) L * logf(p1) + logf((float) (1. - (double) p1));
      const float seq_score = (float) ((double) (usc - filtersc) / kLog2);
      P = d_gumbel_surv((double) seq_score, (double) p.mmu, (double) p.mlambda);
      b.filtersc[s] = filtersc;
      if (P > p.F1) pass = false;
    } else {
      const float seq_score = (float) ((double) (usc - nullsc) / kLog2);
      P = d_gumbel_surv((double) seq_score, (double) p.mmu, (double) p.mlambda);
      b.filtersc[s] = filtersc;
    }
    to_vit = pass && (P > p.F2);
    to_fwd = pass && !(P > p.F2);
    if (pass) b.stage[s] = to_fwd ? 3 : 2;     // P <= F2 already: the Viterbi filter is skipped and counts as passed
    }
    wave_count(&b.counters[8], to_vit || to_fwd);
    // targets for the Viterbi filter are marked stage == 2; their work list is built by vit_compact_* below, in slot
    // order (= by decreasing length), instead of in the order in which the threads here happen to finish
    wave_append(&b.counters[3], b.list_fwd, to_fwd, s);
  }
}

// after Viterbi: vfsc, P2; survivors -> list_fwd
__global__ void decide_vit_kernel(const ArgRef ref)
{
  const DecideArgs a = load_args<DecideArgs>(ref);
  const StageBufs &b = a.b; const StageParams &p = a.p;
  const int n = b.counters[2];
  for (int it0 = blockIdx.x * blockDim.x; it0 < n; it0 += gridDim.x * blockDim.x) {
    const int it = it0 + (int) threadIdx.x;
    bool take = false; int s = 0;
    if (it < n) {
    s = b.list_vit[it];
    const int L = a.slot_len[s];
    const int xC = b.xC[it];
    float vfsc;
    if (xC >= 32767) vfsc = __builtin_inff();
    else if (xC > -32768) {
      vfsc = (float) xC + (float) a.xwmove_tab[L] - (float) p.base_w;
      vfsc /= p.scale_w;
      vfsc -= 3.0f;
    } else vfsc = -__builtin_inff();
    b.vfsc[s] = vfsc;
    const float seq_score = (float) ((double) (vfsc - b.filtersc[s]) / kLog2);
    const double P = d_gumbel_surv((double) seq_score, (double) p.vmu, (double) p.vlambda);
    take = !(P > p.F2);
    if (take) b.stage[s] = 3;
    }
    wave_append(&b.counters[3], b.list_fwd, take, s);
  }
}

// after Forward: P3; survivors -> list_fin
__global__ void decide_fwd_kernel(const ArgRef ref)
{
  const DecideArgs a = load_args<DecideArgs>(ref);
  const StageBufs &b = a.b; const StageParams &p = a.p;
  const int n = b.counters[3];
  for (int it0 = blockIdx.x * blockDim.x; it0 < n; it0 += gridDim.x * blockDim.x) {
    const int it = it0 + (int) threadIdx.x;
    bool take = false, near = false; int s = 0;
    if (it < n) {
      s = b.list_fwd[it];
      const float fwdsc = b.fwd_by_item[it];
      b.fwdsc[s] = fwdsc;
      const float seq_score = (float) ((double) (fwdsc - b.filtersc[s]) / kLog2);
      const double P = d_exp_surv((double) seq_score, (double) p.ftau, (double) p.flambda);
      take = !(P > p.F3);
      near = take && (P > p.F3_near);
      if (take) b.stage[s] = 4;
    }
    wave_append(&b.counters[4], b.list_fin, take, s);
    wave_append(&b.counters[14], b.list_bias, near, s);
  }
}

// ---------------------------------------------------------------------------- Viterbi work list, sorted by length
// Slots are sorted by decreasing target length, so the list of slots with stage == 2 in slot order is the Viterbi
// work list sorted by decreasing length: the 8 (or 4) targets that share a wavefront of the packed kernel then have
// about the same length (a wavefront runs to its longest target), and the longest targets -- the first <long_slots>
// slots -- form a prefix of the list that goes to the wave-per-target kernel instead (counters[13] = its length).
constexpr int kCompactChunk = 4096;       // slots per block
struct CompactArgs {
  const uint8_t *stage; int32_t *list; int *chunk_cnt; int *counters;
  int64_t nslots; int64_t long_slots;
};

__device__ __forceinline__ int count_marked(const uint8_t *stage, int64_t lo, int64_t hi, uint32_t &mask)
{ // this thread's 16 slots lo .. lo+15 (clipped to hi): bit j of mask = slot lo + j is marked
  mask = 0;
  if (lo + 16 <= hi) {
    const uint4 v = *reinterpret_cast<const uint4 *>(stage + lo);
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int j = 0; j < 16; ++j) if (((w[j >> 2] >> (8 * (j & 3))) & 0xffu) == 2u) mask |= 1u << j;
  } else {
    for (int j = 0; j < 16 && lo + j < hi; ++j) if (stage[lo + j] == 2) mask |= 1u << j;
  }
  return __popc(mask);
}

__global__ void __launch_bounds__(256) vit_compact_count_kernel(const ArgRef ref)
{
  const CompactArgs a = load_args<CompactArgs>(ref);
  const int64_t lo = (int64_t) blockIdx.x * kCompactChunk + (int64_t) threadIdx.x * 16;
  if ((int64_t) blockIdx.x * kCompactChunk >= a.nslots) return;
  uint32_t mask;
  int c = count_marked(a.stage, lo, a.nslots, mask);
  c = (int) wave_sum_i32(c);
  __shared__ int part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) a.chunk_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(256) vit_compact_write_kernel(const ArgRef ref)
{
  const CompactArgs a = load_args<CompactArgs>(ref);
  const int64_t chunk_lo = (int64_t) blockIdx.x * kCompactChunk;
  if (chunk_lo >= a.nslots) return;
  __shared__ int scan[256];
  __shared__ int base_s;
  // list position of this chunk's first marked slot: the marked slots of the chunks before it
  if (threadIdx.x < 64) {
    int acc = 0;
    for (int c = (int) threadIdx.x; c < (int) blockIdx.x; c += 64) acc += a.chunk_cnt[c];
    acc = (int) wave_sum_i32(acc);
    if (threadIdx.x == 0) base_s = acc;
  }
  const int64_t lo = chunk_lo + (int64_t) threadIdx.x * 16;
  uint32_t mask;
  const int c = count_marked(a.stage, lo, a.nslots, mask);
  scan[threadIdx.x] = c;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {         // inclusive Hillis-Steele scan over the 256 thread counts
    const int v = (int) threadIdx.x >= d ? scan[threadIdx.x - d] : 0;
    __syncthreads();
    scan[threadIdx.x] += v;
    __syncthreads();
  }
  int pos = base_s + scan[threadIdx.x] - c;
  for (int j = 0; j < 16; ++j) if (mask & (1u << j)) a.list[pos++] = (int32_t) (lo + j);
  // the long prefix ends inside the chunk that holds slot long_slots (or is the whole list)
  if (a.long_slots > chunk_lo && a.long_slots <= chunk_lo + kCompactChunk && a.long_slots < a.nslots) {
    const int64_t cut = a.long_slots - lo;             // slots of this thread below the cut
    int below = cut >= 16 ? c : (cut <= 0 ? 0 : __popc(mask & ((1u << cut) - 1u)));
    below = (int) wave_sum_i32(below);
    if ((threadIdx.x & 63) == 0) atomicAdd(&a.counters[13], below);
    if (threadIdx.x == 0) atomicAdd(&a.counters[13], base_s);
  }
  if (chunk_lo + kCompactChunk >= a.nslots && threadIdx.x == 255) {      // last chunk: the list's length
    const int total = base_s + scan[255];
    a.counters[2] = total;
    if (a.long_slots >= a.nslots) a.counters[13] = total;
  }
}

// ---------------------------------------------------------------------------- layout of the survivors' row blocks
// Exclusive scan of (L+1)*6 floats over a lane's Forward survivors, so that the rows pass, Backward and the region
// scan can follow the filters without the host learning the number of survivors first.  One block per lane; the lanes
// of a batch share one row arena and take their part of it from a cursor (one 64-bit atomic per lane).  flags[0] of
// the lane is raised when it has more survivors than its per-survivor arrays hold, or when its rows do not fit in what
// is left of the arena; the host then repeats the tail for that lane with buffers of the right size.
struct LayoutArgs {
  const int *nfin_ptr; const int32_t *list_fin; const int32_t *slot_len;
  int64_t *xmx_off; int cap_items; long long cap_floats; unsigned long long *cursor; int *flags;
};

__global__ void __launch_bounds__(256) layout_rows_kernel(const ArgRef ref)
{
  const LayoutArgs a = load_args<LayoutArgs>(ref);
  __shared__ long long part[256];
  __shared__ long long base_s;
  const int n = *a.nfin_ptr;
  if (n == 0) return;
  if (n > a.cap_items) { if (threadIdx.x == 0) a.flags[0] = 1; return; }
  const int per = (n + 255) / 256, lo = threadIdx.x * per, hi = min(n, lo + per);
  long long sum = 0;
  for (int i = lo; i < hi; ++i) sum += (long long) (a.slot_len[a.list_fin[i]] + 1) * 6;
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long run = 0; for (int t = 0; t < 256; ++t) { const long long v = part[t]; part[t] = run; run += v; }
    const long long base = (long long) atomicAdd(a.cursor, (unsigned long long) run);
    if (base + run > a.cap_floats) { a.flags[0] = 1; base_s = -1; } else base_s = base;
  }
  __syncthreads();
  if (base_s < 0) return;
  long long run = base_s + part[threadIdx.x];
  for (int i = lo; i < hi; ++i) { a.xmx_off[i] = run; run += (long long) (a.slot_len[a.list_fin[i]] + 1) * 6; }
}

// ---------------------------------------------------------------------------- regions (p7_DomainDecoding + region scan)
// One wavefront per Forward survivor.  The parsers' special-state rows stay on the device: the posterior
// begin/end/occupancy terms are formed lane-parallel, their running sums and the trigger scan run in upstream's
// sequential order (every lane follows the same scalar recurrence; only the prefix arrays needed by the
// multi-domain test are written out), and is_multidomain_region() is a wave-parallel max.  Same arithmetic as
// domaindef_regions() in p7x_domaindef.cpp, which remains the reference for it (tests/test_gpu_envelopes.py).
constexpr int kRegionCap = 128;     // regions kept per target; more than that falls back to the host scan

struct RegionArgs {
  const int *nitems_ptr;      // number of Forward survivors, on the device
  const int *abort_flag;      // raised by layout_rows_kernel when the buffers are too small: do nothing
  const int32_t *list;        // slot of item
  const int32_t *slot_len;
  const float *fx, *bx;       // parser rows, (L+1) x [E,N,J,B,C,SCALE]
  const int64_t *xmx_off;     // per item
  float *scratch;             // per item at xmx_off: terms [3][L+1] then prefix sums [2][L+1]  (5 of the 6 row floats)
  int32_t *out_regs;          // [nitems][kRegionCap][3] = i, j, multi
  int32_t *out_n;             // [nitems] number of regions, or -1 (range error), or -2 (more than kRegionCap), or -3 (a threshold
                              // comparison of the scan within the guard band: the host stage scans this target itself, from parser rows
                              // in upstream's summation order)
  float guard;                // absolute half-width of that band (p7x_pipeline_cfg.oa_guard > 0: 2e-5; else 0 = off)
  float *out_nexpected;       // [nitems]
};

__global__ void __launch_bounds__(256) regions_kernel(const ArgRef ref)
{
  const RegionArgs a = load_args<RegionArgs>(ref);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int nwaves = (int) gridDim.x * 4;
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;                         // p7_domaindef.pxd:39-41
  const int nitems = (a.abort_flag && *a.abort_flag) ? 0 : *a.nitems_ptr;
  for (int it = wave; it < nitems; it += nwaves) {
    const int L = __builtin_amdgcn_readfirstlane(a.slot_len[a.list[it]]);
    const long long off = a.xmx_off[it];
    const float *fx = a.fx + off, *bx = a.bx + off;
    float *tb = a.scratch + off, *te = tb + (L + 1), *tm = te + (L + 1), *pb = tm + (L + 1), *pe = pb + (L + 1);
    const float pmove = (2.0f + 1.0f) / ((float) L + 2.0f + 1.0f), ploop = 1.0f - pmove;      // multihit, full length
    const float scaleproduct = (float) (1.0 / (double) bx[1]);
    for (int i = 1 + lane; i <= L; i += 64) {
      const float *f0 = fx + (size_t) (i - 1) * 6, *f1 = fx + (size_t) i * 6, *b0 = bx + (size_t) (i - 1) * 6, *b1 = bx + (size_t) i * 6;
      tb[i] = (f0[3] * b0[3]) * f0[5] * scaleproduct;
      te[i] = (f1[0] * b1[0]) * f1[5] * scaleproduct;
      float njcp = f0[1] * b1[1] * ploop * scaleproduct;
      njcp += f0[2] * b1[2] * ploop * scaleproduct;
      njcp += f0[4] * b1[4] * ploop * scaleproduct;
      tm[i] = (float) (1. - (double) njcp);
    }
    if (lane == 0) { pb[0] = 0.0f; pe[0] = 0.0f; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // producer and consumer are this wavefront
    int nreg = 0;
    int32_t *regs = a.out_regs + (size_t) it * kRegionCap * 3;
    float b = 0.0f, e = 0.0f;
    int i = -1; bool triggered = false;
    // the rows come from the device parsers (lane-chunk summation order): a comparison closer to its threshold than the guard
    // may fall the other way in upstream's order
    bool near = false;
    const float gd = a.guard;
    for (int j0 = 1; j0 <= L; j0 += 64) {
      const int nj = min(64, L - j0 + 1);
      // this block's terms, one per lane; broadcast in order below
      const float vb = (lane < nj) ? tb[j0 + lane] : 0.0f, ve = (lane < nj) ? te[j0 + lane] : 0.0f, vm = (lane < nj) ? tm[j0 + lane] : 0.0f;
      float keepb = 0.0f, keepe = 0.0f;
      for (int r = 0; r < nj; ++r) {
        const int j = j0 + r;
        const float dbt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vb), r));
        const float det = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ve), r));
        const float mo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vm), r));
        const float bprev = b, eprev = e;
        b = bprev + dbt; e = eprev + det;
        if (lane == r) { keepb = b; keepe = e; }
        if (!triggered) {
          near = near || __builtin_fabsf((mo - (b - bprev)) - rt2) <= gd || __builtin_fabsf(mo - rt1) <= gd;
          if (mo - (b - bprev) < rt2) i = j;
          else if (i == -1) i = j;
          if (mo >= rt1) triggered = true;
        } else if ((near = near || __builtin_fabsf((mo - (e - eprev)) - rt2) <= gd), mo - (e - eprev) < rt2) {
          // region i..j closes here: flush this block's prefix sums, then is_multidomain_region()
          if (lane <= r) { pb[j0 + lane] = keepb; pe[j0 + lane] = keepe; }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // producer and consumer are this wavefront
          const float e0 = pe[i - 1], bj = b;
          float mx = -1.0f;
          for (int z = i + lane; z <= j; z += 64) mx = fmaxf(mx, fminf(pe[z] - e0, bj - pb[z - 1]));
          mx = wave_max_f32(mx);
          near = near || __builtin_fabsf(mx - rt3) <= gd;
          if (nreg < kRegionCap && lane == 0) { regs[nreg * 3 + 0] = i; regs[nreg * 3 + 1] = j; regs[nreg * 3 + 2] = (mx >= rt3) ? 1 : 0; }
          ++nreg;
          i = -1; triggered = false;
        }
      }
      if (lane < nj) { pb[j0 + lane] = keepb; pe[j0 + lane] = keepe; }
    }
    if (lane == 0) {
      a.out_nexpected[it] = b;
      a.out_n[it] = __builtin_isinf(scaleproduct) ? -1 : (nreg > kRegionCap ? -2 : ((near && gd > 0.0f) ? -3 : nreg));
    }
  }
}

// ---------------------------------------------------------------------------- workspace
// Everything the kernels of one lane (one query profile of a batch) read, as one device record.
struct LaneArgs {
  DecideArgs dec;
  MsvArgs msv, msv_amb;       // fast (or exact-over-everything) MSV; exact MSV over the lane's ambiguous groups
  MsvWaveArgs msvw;
  VitPkArgs vitpk;
  CompactArgs cmp;
  WaveSeqArgs vit, fwd, rows, bck;
  LayoutArgs lay;
  RegionArgs reg;
};
constexpr int kLaneCounters = 16;
// counters of a lane: [0] msv groups taken [1] n(list_bias) [2] n(list_vit) [3] n(list_fwd) [4] n(list_fin)
// [5] work counter of the grouped Forward parser [8] n_past_bias [10] ambiguous MSV groups [11] exact-MSV groups taken [12] flag: survivor buffers too small
// [13] length of the prefix of list_vit that holds the longest targets (wave-per-target Viterbi)

// A workspace serves a batch of up to <nlanes> queries against a block of up to <cap_slots> targets: per-lane
// score / list arrays, one row arena shared by the lanes, the lanes' argument records, one stream.
struct Workspace {
  int device = -1; int64_t cap_slots = 0; int nlanes = 0;
  // per lane, lane-major: lane l's part starts at l * cap_slots elements
  int16_t *xJ = nullptr; float *usc = nullptr, *filtersc = nullptr, *vfsc = nullptr, *fwdsc = nullptr;
  int32_t *xC = nullptr; float *fwd_by_item = nullptr;
  int32_t *list_bias = nullptr, *list_vit = nullptr, *list_fwd = nullptr, *list_fin = nullptr;
  uint8_t *stage = nullptr;
  int *chunk_cnt = nullptr; int64_t chunks_per_lane = 0;      // [nlanes][chunks_per_lane] marked slots per compaction chunk
  int *counters = nullptr;                  // [nlanes][kLaneCounters] then the arena cursor (64 bit)
  LaneArgs *d_args = nullptr, *h_args = nullptr; size_t h_args_bytes = 0;     // [nlanes], device / pinned host
  // Forward survivors: row arena (Forward rows, Backward rows, region-scan scratch) and per-lane, per-survivor arrays
  float *xmx_f = nullptr, *xmx_b = nullptr, *xmx_s = nullptr; int64_t xmx_cap = 0; size_t xmx_bytes[3] = { 0, 0, 0 };   // slabs of the context's pool
  int64_t *xmx_off = nullptr; int32_t *reg_out = nullptr; float *bck_sc = nullptr;   // [nlanes][fin_cap (* kRegionCap*3+2)]
  int64_t fin_cap = 0;
  // one lane at a time, when the shared buffers were too small for it (rare)
  int64_t *rt_xmx_off = nullptr; int32_t *rt_reg_out = nullptr; float *rt_bck_sc = nullptr; int64_t rt_cap = 0;
  hipEvent_t ev[8]{};
  hipEvent_t ev_sync = nullptr;
  hipStream_t stream = nullptr;     // one stream per cascade in flight: concurrent searches overlap on the device
  // The lanes of a batch fall into classes that share every kernel instantiation; each class runs its cascade on its
  // own stream (forked from / joined into <stream>), so that the classes' latency-bound launches overlap
  static constexpr int kSide = DeviceCtx::kWsSide;        // the streams belong to the device context (queue placement, see there)
  hipStream_t side[kSide]{};
  hipEvent_t ev_fork = nullptr, ev_join[kSide]{};
  static constexpr int kTierRuns = 8;      // MSV tier launches of a batch (p7x_msv.hip: five tiers; runs of lanes that share one)
  hipEvent_t ev_tier[kTierRuns]{}, ev_tier0[kTierRuns]{};     // end / begin of every tier launch (timed: the bench's roofline is the largest one's)
  int ntier_runs = 0, tier_lanes[kTierRuns]{}; int64_t tier_nodes[kTierRuns]{};     // of the cascade in flight
  int *h_counts = nullptr; size_t h_counts_bytes = 0;   // pinned mirror of counters
  // results of the Forward survivors of all lanes, packed by pack_survivors_kernel for one copy to the host
  unsigned char *pack_dev = nullptr; size_t pack_dev_bytes = 0;      // slab of the context's pool
  unsigned char *pack_host = nullptr; size_t pack_host_bytes = 0;    // pinned
  int64_t early_T_cap = 0;          // > 0: the enqueue half packed and copied the survivors' results itself, for up to this many
  int64_t last_T = 2048;            // survivors of the last batch collected on this workspace (sizes the next early pack)
  bool busy = false;                // between the enqueue and the collect half of a cascade
  int set = -1;                     // the context's stream set this lease runs on (taken when leased, given back on release)
  size_t counters_bytes() const { return (size_t) nlanes * kLaneCounters * 4 + 8; }
  unsigned long long *cursor() const { return reinterpret_cast<unsigned long long *>(counters + (size_t) nlanes * kLaneCounters); }
  ~Workspace() {
    if (device < 0) return;
    (void) hipSetDevice(device);
    (void) hipFree(xJ); (void) hipFree(usc); (void) hipFree(filtersc); (void) hipFree(vfsc); (void) hipFree(fwdsc);
    (void) hipFree(xC); (void) hipFree(fwd_by_item); (void) hipFree(list_bias); (void) hipFree(list_vit);
    (void) hipFree(list_fwd); (void) hipFree(list_fin); (void) hipFree(counters); (void) hipFree(stage); (void) hipFree(d_args);
    (void) hipFree(chunk_cnt);
    (void) hipFree(xmx_f); (void) hipFree(xmx_b); (void) hipFree(xmx_s); (void) hipFree(xmx_off); (void) hipFree(reg_out); (void) hipFree(bck_sc);
    (void) hipFree(rt_xmx_off); (void) hipFree(rt_reg_out); (void) hipFree(rt_bck_sc);
    for (auto &e : ev) if (e) (void) hipEventDestroy(e);
    if (ev_sync) (void) hipEventDestroy(ev_sync);
    if (ev_fork) (void) hipEventDestroy(ev_fork);
    for (auto &e : ev_join) if (e) (void) hipEventDestroy(e);
    for (auto &e : ev_tier) if (e) (void) hipEventDestroy(e);
    for (auto &e : ev_tier0) if (e) (void) hipEventDestroy(e);
    pinned_release(h_counts, h_counts_bytes);
    pinned_release(pack_host, pack_host_bytes); (void) hipFree(pack_dev);
    pinned_release(h_args, h_args_bytes);
  }
  StageBufs lane_bufs(int l) const
  {
    const size_t o = (size_t) l * (size_t) cap_slots;
    StageBufs b{};
    b.xJ = xJ + o; b.usc = usc + o; b.filtersc = filtersc + o; b.vfsc = vfsc + o; b.fwdsc = fwdsc + o;
    b.xC = xC + o; b.fwd_by_item = fwd_by_item + o;
    b.list_bias = list_bias + o; b.list_vit = list_vit + o; b.list_fwd = list_fwd + o; b.list_fin = list_fin + o;
    b.stage = stage + o; b.counters = counters + (size_t) l * kLaneCounters;
    return b;
  }
};

// Workspaces are leased from a process-wide pool and returned to it (release_workspace); the pool is never torn down,
// so no device memory, stream or event is destroyed from a thread that is exiting or at process exit.
struct WorkspacePool { std::mutex mu; std::vector<Workspace *> all; };
static WorkspacePool &ws_pool() { static WorkspacePool *p = new WorkspacePool(); return *p; }

// The stream sets are the context's (created once, placed apart on the hardware queues); a lease runs on the set that the
// fewest cascades are using right now.  (Until round 5 a workspace kept the set it was created with, handed out by a
// creation counter: a workspace that was replaced by a larger one, or more than four live workspaces, could put two
// running cascades on the same eight streams while another set sat idle.)
static int lease_stream_set(Workspace *w)
{
  DeviceCtx *ctx = nullptr;
  const int cst = get_ctx(w->device, &ctx);
  if (cst != P7X_OK) return cst;
  std::lock_guard<std::mutex> lk(ctx->mu);
  int set = 0;
  for (int k = 1; k < DeviceCtx::kWsSets; ++k) if (ctx->ws_set_users[k] < ctx->ws_set_users[set]) set = k;
  ctx->ws_set_users[set]++;
  w->set = set;
  w->stream = ctx->ws_main[set];
  for (int k = 0; k < Workspace::kSide; ++k) w->side[k] = ctx->ws_side[set][k];
  return P7X_OK;
}

static void release_workspace(Workspace *w)
{
  if (!w) return;
  if (w->set >= 0) {
    DeviceCtx *ctx = nullptr;
    if (get_ctx(w->device, &ctx) == P7X_OK) { std::lock_guard<std::mutex> lk(ctx->mu); if (ctx->ws_set_users[w->set] > 0) ctx->ws_set_users[w->set]--; }
    w->set = -1;
  }
  std::lock_guard<std::mutex> lk(ws_pool().mu);
  w->busy = false;
}

static int get_workspace(int device, int64_t nslots, int nlanes, Workspace **out)
{
  Workspace *found = nullptr;
  {
    std::lock_guard<std::mutex> lk(ws_pool().mu);
    Workspace *best = nullptr;
    for (Workspace *w : ws_pool().all)
      if (!w->busy && w->device == device && w->cap_slots >= nslots && w->nlanes >= nlanes &&
          (!best || w->cap_slots * w->nlanes < best->cap_slots * best->nlanes)) best = w;
    if (best) { best->busy = true; found = best; }
  }
  if (found) { *out = found; return lease_stream_set(found); }
  auto w = std::make_unique<Workspace>();
  w->device = device;
  const int64_t cap = std::max<int64_t>(64, ((nslots + 63) / 64) * 64);
  w->cap_slots = cap; w->nlanes = nlanes;
  const size_t tot = (size_t) cap * (size_t) nlanes;
  P7X_HIP(hipMalloc(&w->xJ, tot * 2));
  P7X_HIP(hipMalloc(&w->usc, tot * 4)); P7X_HIP(hipMalloc(&w->filtersc, tot * 4));
  P7X_HIP(hipMalloc(&w->vfsc, tot * 4)); P7X_HIP(hipMalloc(&w->fwdsc, tot * 4));
  P7X_HIP(hipMalloc(&w->xC, tot * 4)); P7X_HIP(hipMalloc(&w->fwd_by_item, tot * 4));
  P7X_HIP(hipMalloc(&w->list_bias, tot * 4)); P7X_HIP(hipMalloc(&w->list_vit, tot * 4));
  P7X_HIP(hipMalloc(&w->list_fwd, tot * 4)); P7X_HIP(hipMalloc(&w->list_fin, tot * 4));
  P7X_HIP(hipMalloc(&w->counters, w->counters_bytes()));
  P7X_HIP(hipMalloc(&w->d_args, (size_t) nlanes * sizeof(LaneArgs)));
  { void *hp = nullptr; const int pst = pinned_acquire(w->counters_bytes(), &hp, &w->h_counts_bytes); if (pst != P7X_OK) return pst; w->h_counts = static_cast<int *>(hp); }
  { void *hp = nullptr; const int pst = pinned_acquire((size_t) nlanes * sizeof(LaneArgs), &hp, &w->h_args_bytes); if (pst != P7X_OK) return pst; w->h_args = static_cast<LaneArgs *>(hp); }
  P7X_HIP(hipMalloc(&w->stage, tot));
  w->chunks_per_lane = (cap + kCompactChunk - 1) / kCompactChunk;
  P7X_HIP(hipMalloc(&w->chunk_cnt, (size_t) nlanes * (size_t) w->chunks_per_lane * 4));
  for (auto &e : w->ev) P7X_HIP(hipEventCreate(&e));
  P7X_HIP(hipEventCreateWithFlags(&w->ev_sync, hipEventDisableTiming));
  {   // the cascade is the critical path of a search (its streams have the highest priority: its wavefronts go first when
      // the envelope kernel of the previous query shares the device); the streams are the context's, dealt out in turn
    const int cst = lease_stream_set(w.get());
    if (cst != P7X_OK) return cst;
    P7X_HIP(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
    for (auto &e : w->ev_join) P7X_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : w->ev_tier) P7X_HIP(hipEventCreate(&e));
    for (auto &e : w->ev_tier0) P7X_HIP(hipEventCreate(&e));
  }
  w->busy = true;
  *out = w.get();
  std::lock_guard<std::mutex> lk(ws_pool().mu);
  ws_pool().all.push_back(w.release());
  return P7X_OK;
}
struct WorkspaceLease {      // for the synchronous entry points
  Workspace *w = nullptr;
  ~WorkspaceLease() { release_workspace(w); }
};

static StageParams make_params(const Profile &p, const p7x_pipeline_cfg &cfg)
{
  StageParams s{};
  s.F1 = cfg.do_max ? 1.0 : cfg.F1; s.F2 = cfg.do_max ? 1.0 : cfg.F2; s.F3 = cfg.do_max ? 1.0 : cfg.F3;
  // the host stage has the last word on targets within the guard band of F3 (p7x_tophits.cpp): let all of them through
  s.F3_near = 2.0;
  if (!cfg.do_max && !cfg.long_targets && cfg.f3_guard > 0.0f) { s.F3 = cfg.F3 * (1.0 + (double) cfg.f3_guard); s.F3_near = cfg.F3 * (1.0 - (double) cfg.f3_guard); }
  s.mmu = p.evparam[P7X_MMU]; s.mlambda = p.evparam[P7X_MLAMBDA]; s.vmu = p.evparam[P7X_VMU];
  s.vlambda = p.evparam[P7X_VLAMBDA]; s.ftau = p.evparam[P7X_FTAU]; s.flambda = p.evparam[P7X_FLAMBDA];
  s.do_bias = cfg.do_max ? 0 : cfg.do_biasfilter;
  // gumbel_surv(x) = F1 at x = mu - log(-log(1 - F1)) / lambda; NaN-proof: any doubt leaves the threshold at -inf
  s.msv_reject_below = -INFINITY;
  if (s.F1 > 0.0 && s.F1 < 1.0 && s.mlambda > 0.0f) {
    const double x1 = (double) s.mmu - std::log(-std::log1p(-s.F1)) / (double) s.mlambda;
    if (std::isfinite(x1)) s.msv_reject_below = (float) (x1 - 1e-3) - 1e-3f;
  }
  s.base_b = p.base_b; s.scale_b = p.scale_b; s.base_w = p.base_w; s.scale_w = p.scale_w; s.M = p.M;
  return s;
}

// Expected lengths of the work lists (grid sizing only; the kernels read the true lengths on the device): the filters
// pass about F1 / F2 / F3 of the comparisons of unrelated sequences, a few times that on homolog-rich databases.
struct ListEstimates { int bias, vit, fwd, fin; };
static ListEstimates estimate_lists(const p7x_pipeline_cfg &cfg, int64_t nslots)
{
  auto est = [&](double f, double slack, int floor_) {
    const double v = cfg.do_max ? (double) nslots : std::min<double>((double) nslots, (double) nslots * f * slack + floor_);
    return (int) std::min<double>(v, (double) INT_MAX);
  };
  ListEstimates e;
  e.bias = est(cfg.F1, 2.0, 1024); e.vit = e.bias;
  e.fwd = est(cfg.F2, 8.0, 512);
  e.fin = est(cfg.F3, 50.0, 1024);
  return e;
}

static WaveSeqArgs ws_args(const Profile &p, const DevProfile *dp, const p7x_seqdb *db, DeviceCtx *ctx)
{
  WaveSeqArgs a{};
  a.M = p.M; a.C = dp->vitC; a.nrows = p.Kp + 1;
  a.dsq = db->d_dsq; a.slot_off = db->d_slot_off; a.slot_len = db->d_slot_len;
  a.xwmove_tab = ctx->lt.xwmove; a.base_w = p.base_w; a.xw_e = p.xw[XE][MOVE]; a.ddbound = p.ddbound_w;
  a.xf_e_move = p.xf[XE][MOVE]; a.xf_e_loop = p.xf[XE][LOOP];
  return a;
}

// kernel families and placements a parity test or an A/B run can force through the test seam p7x_debug_set_option (p7x.h)
static bool msv_exact_only() { return debug_opt(OPT_MSV_EXACT) > 0; }      // only the exact MSV kernel (no fast pass)
static bool vit_wave_only() { return debug_opt(OPT_VIT_WAVE) > 0; }        // the wave-per-target Viterbi kernel for every model
// With host_ensembles set (the stochastic ensembles stay on the host workers) their clustered envelopes -- about two per
// multi-domain region -- can still be rescored by the envelope kernel as a second round: identical results either way.
// The host workers do it where there are many of them and the device where there are few (round 3, 16 threads 19.1 vs
// 18.7-19.2 TCUPS, 8 threads 10.0 vs 11.1); option "device_clustered" forces either.
static bool device_clustered(int host_threads)
{
  const int forced = debug_opt(OPT_DEVICE_CLUSTERED);
  if (forced >= 0) return forced != 0;
  const int threads = host_threads > 0 ? std::min(host_threads, tophits_usable_cpus()) : tophits_usable_cpus();
  return threads < 12;
}

// The lane-per-target MSV and the 8-lanes-per-target Viterbi need many (profile, 64-target group) pairs to fill the
// device and run for as long as the longest member of a group takes one wavefront.  A batch with at most one such
// pair per SIMD (one profile against a scan's query sequences, the fixture proteome) goes one target per wavefront
// through the filters instead.  Option "small_block" = 0 disables the switch (the tests run every filter through both
// families of kernels).
static bool small_block(const p7x_seqdb *db, const DeviceCtx *ctx, int nlanes)
{
  return debug_opt(OPT_SMALL_BLOCK) != 0 && db->ngroups * (int64_t) nlanes <= (int64_t) ctx->num_cu * 4;
}

// The lane-per-target MSV kernel walks a 64-target group for as long as its longest member: one 8000-residue protein
// keeps a wavefront busy for milliseconds while the other groups of the launch finished long ago.  The leading
// (longest) groups are therefore left to the wave-per-target kernel: as many as it takes for the lane kernel's
// longest remaining group to be no longer than its average work per resident wavefront (its makespan is then set by
// throughput, not by the tail).  With many lanes (profiles) per launch the average grows and fewer groups qualify.
static int long_groups(const p7x_seqdb *db, const DeviceCtx *ctx, int nlanes)
{
  if (debug_opt(OPT_MSV_LONG_GROUPS) == 0 || db->ngroups < 2) return 0;
  const int64_t resident = (int64_t) ctx->num_cu * 4 * 3;           // wavefronts the lane kernel keeps in flight
  const int64_t G = db->ngroups;
  int64_t g = 0;
  while (g < G / 2 && (int64_t) db->h_grp_len[(size_t) g] * resident > (int64_t) nlanes * db->h_grp_suffix[(size_t) g]) ++g;
  return (int) g;
}

// ---- argument records of a lane
static void fill_msv_args(LaneArgs &la, const Profile &p, const DevProfile *dp, const p7x_seqdb *db, DeviceCtx *ctx, const StageBufs &b,
                          int nlong = 0)
{
  MsvWaveArgs w{};
  w.C = dp->msvwC; w.nrows = dp->msvw_rows; w.emis = dp->msvw_emis; w.emis_pk = dp->msvw_pk; w.dsq = db->d_dsq; w.slot_off = db->d_slot_off; w.slot_len = db->d_slot_len;
  w.tjb_tab = ctx->lt.tjb; w.nslots = (int) db->nslots; w.base = p.base_b; w.bias = p.bias_b; w.tec = p.tec_b; w.tbm = p.tbm_b;
  w.out_xJ = b.xJ;
  if (nlong > 0) w.nslots = (int) std::min<int64_t>((int64_t) nlong * 64, db->nslots);   // only the long groups (with the lane kernel)
  la.msvw = w;
  MsvArgs a{};
  a.group_first = nlong;
  a.tab = dp->msv_tab; a.tiles = db->d_tiles; a.grp_off = db->d_grp_off; a.grp_nblk = db->d_grp_nblk;
  a.slot_len = db->d_slot_len; a.tjb_tab = ctx->lt.tjb; a.ngroups = (int) db->ngroups;
  a.base = p.base_b; a.bias = p.bias_b; a.tec = p.tec_b; a.tbm = p.tbm_b;
  a.counter = &b.counters[0]; a.out_xJ = b.xJ;
  a.amb_count = &b.counters[10]; a.counter2 = &b.counters[11];
  a.amb_groups = msv_exact_only() ? nullptr : b.list_fin;     // list_fin is free until the Forward stage
  a.R = dp->msvR;
  la.msv = a;
  MsvArgs x = a;
  x.group_list = a.amb_groups; x.group_count = a.amb_count; x.counter = a.counter2; x.amb_groups = nullptr;
  la.msv_amb = x;
}

// <nlong_ptr> (packed kernel only): the first *nlong_ptr items of the list are the longest targets; the wave-per-target
// kernel takes them (<nlong_bound> of them at most), the packed kernel the rest.
static void fill_vit_args(LaneArgs &la, const Profile &p, const DevProfile *dp, const p7x_seqdb *db, DeviceCtx *ctx,
                          const int32_t *list, int nlist, const int *nlist_ptr, int32_t *out_xC,
                          const int *nlong_ptr = nullptr, int nlong_bound = 0)
{
  WaveSeqArgs w = ws_args(p, dp, db, ctx);
  w.trans = dp->vit_trans; w.emis = dp->vit_emis; w.list = list; w.nlist = nlist; w.nlist_ptr = nlist_ptr; w.out_xC = out_xC;
  if (nlong_ptr) { w.nlist_ptr = nlong_ptr; w.nlist = nlong_bound; }
  la.vit = w;
  VitPkArgs a{};
  a.trans = dp->vitpk_trans; a.emis = dp->vitpk_emis; a.dsq = db->d_dsq; a.slot_off = db->d_slot_off; a.slot_len = db->d_slot_len;
  a.list = list; a.nlist = nlist; a.nlist_ptr = nlist_ptr; a.nskip_ptr = nlong_ptr; a.nrows = p.Kp + 1;
  a.xwmove_tab = ctx->lt.xwmove; a.base_w = p.base_w; a.xw_e = p.xw[XE][MOVE]; a.ddbound = p.ddbound_w;
  a.out_xC = out_xC;
  la.vitpk = a;
}

template <class A> static ArgRun<A> lane_run(const Workspace *ws, A LaneArgs::*member, int first, int n)
{
  ArgRun<A> r;
  r.host = &(ws->h_args[first].*member);
  r.dev = reinterpret_cast<const char *>(ws->d_args + first) + (reinterpret_cast<const char *>(r.host) - reinterpret_cast<const char *>(ws->h_args + first));
  r.stride = (uint32_t) sizeof(LaneArgs); r.n = n;
  return r;
}

struct LaneModel { const p7x_oprofile *om = nullptr; DevProfile *dp = nullptr; };

static int upload_args(Workspace *ws, int first, int n, hipStream_t s)
{
  P7X_HIP(hipMemcpyAsync(ws->d_args + first, ws->h_args + first, (size_t) n * sizeof(LaneArgs), hipMemcpyHostToDevice, s));
  return P7X_OK;
}

// Lanes are sorted by model length, so the lanes that share the instantiation of every kernel family (MSV register
// tile or wave kernel, packed or wave Viterbi, nodes per lane of the parsers) are consecutive: a class.
struct LaneClass { int first = 0, n = 0; long msv_key = 0, vit_key = 0; int C = 0; int nlong = 0; int64_t vit_long = 0; };

static long msv_key_of(const DevProfile *dp, bool small)
{ // M > 478, or too few targets for one per lane: wave-per-target kernel (key < 0), else the register tile
  return (dp->msvR <= 0 || small) ? -(long) dp->msvwC : (long) (dp->msvK * 1000 + dp->msvR);
}
static long vit_key_of(const DevProfile *dp, bool small)
{
  return (dp->vitpkT > 0 && !vit_wave_only() && !small) ? (long) (dp->vitpkT * 256 + dp->vitpkP) : -(long) dp->vitC;
}

static int lane_classes(const std::vector<LaneModel> &lm, const p7x_seqdb *db, const DeviceCtx *ctx, std::vector<LaneClass> &out)
{
  const int nl = (int) lm.size();
  const bool small = small_block(db, ctx, nl);
  const int nlong = small ? 0 : long_groups(db, ctx, nl);
  out.clear();
  for (int l = 0; l < nl; ++l) {
    const DevProfile *dp = lm[l].dp;
    if ((dp->msvR <= 0 || small) && !dp->msvw_emis) { set_error("model too long for the MSV kernels (M > 8192)"); return P7X_EINVAL; }
    LaneClass c; c.first = l; c.n = 1; c.msv_key = msv_key_of(dp, small); c.vit_key = vit_key_of(dp, small); c.C = dp->vitC;
    c.nlong = (c.msv_key >= 0 && dp->msvw_emis) ? nlong : 0;
    c.vit_long = c.vit_key >= 0 ? db->vit_long_slots : 0;
    if (!out.empty() && out.back().msv_key == c.msv_key && out.back().vit_key == c.vit_key && out.back().C == c.C) out.back().n++;
    else out.push_back(c);
  }
  return P7X_OK;
}

// MSV over the whole database for the lanes of a class; leaves xJ (slot order) in the lanes' xJ arrays.
static int class_msv(const LaneClass &c, const std::vector<LaneModel> &lm, DeviceCtx *ctx, Workspace *ws, hipStream_t stream)
{
  if (c.msv_key < 0) return msv_wave_launch(lane_run(ws, &LaneArgs::msvw, c.first, c.n), ctx->num_cu, stream);
  if (c.nlong > 0) {      // the longest targets first, one per wavefront
    const int st = msv_wave_launch(lane_run(ws, &LaneArgs::msvw, c.first, c.n), ctx->num_cu, stream);
    if (st != P7X_OK) return st;
  }
  const ArgRun<MsvArgs> amb = lane_run(ws, &LaneArgs::msv_amb, c.first, c.n);
  return msv_launch(lm[c.first].dp->msvR, lm[c.first].dp->msvK, lane_run(ws, &LaneArgs::msv, c.first, c.n), msv_exact_only() ? nullptr : &amb, ctx->num_cu, stream);
}

// What is left of class_msv for a class whose fast kernel ran inside a tier launch (msv_tier_launch): the longest targets'
// wave-per-target pass, and the exact kernel over the groups the fast kernel reported as ambiguous.
static int class_msv_rest(const LaneClass &c, const std::vector<LaneModel> &lm, DeviceCtx *ctx, Workspace *ws, hipStream_t stream)
{
  if (c.nlong > 0) {
    const int st = msv_wave_launch(lane_run(ws, &LaneArgs::msvw, c.first, c.n), ctx->num_cu, stream);
    if (st != P7X_OK) return st;
  }
  return msv_exact_launch(lm[c.first].dp->msvR, lm[c.first].dp->msvK, lane_run(ws, &LaneArgs::msv_amb, c.first, c.n), ctx->num_cu, stream);
}
static bool msv_tiers_enabled() { return !msv_exact_only() && debug_opt(OPT_MSV_F16) != 0 && debug_opt(OPT_MSV_TIERS) != 0; }

// Viterbi filter over the lanes' work lists: the packed kernel when the model fits it, else one target per wavefront.
static int class_viterbi(const LaneClass &c, const std::vector<LaneModel> &lm, DeviceCtx *ctx, Workspace *ws, hipStream_t s)
{
  if (c.vit_key < 0) return vit_launch(lane_run(ws, &LaneArgs::vit, c.first, c.n), ctx->num_cu, s);
  if (c.vit_long > 0) {       // the longest targets (the head of the sorted list) one per wavefront
    const int st = vit_launch(lane_run(ws, &LaneArgs::vit, c.first, c.n), ctx->num_cu, s);
    if (st != P7X_OK) return st;
  }
  return vitpk_launch(lm[c.first].dp->vitpkT, lm[c.first].dp->vitpkP, lane_run(ws, &LaneArgs::vitpk, c.first, c.n), ctx->num_cu, s);
}

static int class_wave(const LaneClass &c, Workspace *ws, WaveSeqArgs LaneArgs::*member, bool backward, DeviceCtx *ctx, hipStream_t s)
{
  return backward ? bck_launch(lane_run(ws, member, c.first, c.n), ctx->num_cu, s) : fwd_launch(lane_run(ws, member, c.first, c.n), ctx->num_cu, s);
}

struct CascadeOut {
  std::vector<int32_t> fin_slots;         // survivors of the Forward filter (slot ids)
  std::vector<float> fwdsc;               // per survivor
  std::vector<float> fwd_xmx, bck_xmx;    // concatenated (L+1)*6 blocks; only fetched when the device region scan overflowed
  std::vector<int64_t> xmx_off;
  std::vector<int32_t> reg_n, regs;       // device region scan: count per survivor (-1 range error), regions (i, j, multi)
  std::vector<int32_t> reg_start;         // first region of every survivor in <regs>; empty: kRegionCap slots per survivor
  std::vector<float> nexpected;
  std::vector<char> near;                 // per survivor: inside the F3 guard band (empty: none)
  bool have_xmx = false;
  std::vector<uint8_t> stage;             // scan mode: last filter passed, per target (caller order)
  int counts[16]{};
  double ms[12]{};                        // 0-5 stage events, 6 wall, 7 the MSV launch, 8 queries of the batch, 9 / 10 lanes and nodes of that launch
};

// One batched cascade in flight: the enqueue half queues every kernel of stage 1 for all lanes on the workspace's
// stream and returns, the collect half waits for them (once), repeats the survivor passes of the lanes whose row
// buffers were too small (rare), and downloads the small result arrays.  A host thread may keep several cascades
// in flight, each on its own leased workspace.
struct CascadeRun {
  p7x_pipeline_cfg cfg{};
  std::vector<const p7x_oprofile *> oms;  // the queries, caller order
  std::vector<int> lane_of;               // caller index -> lane (lanes are sorted by model length)
  std::vector<LaneModel> lm;              // per lane
  std::vector<int> query_of;              // lane -> caller index
  const p7x_seqdb *db = nullptr;
  DeviceCtx *ctx = nullptr;
  Workspace *ws = nullptr;
  int est_bias = 0, est_vit = 0, est_fwd = 0, est_fin = 0;      // expected list lengths (grid sizing)
  bool queued = false, collected = false;
  ~CascadeRun() { if (ws && queued && !collected) { (void) hipStreamSynchronize(ws->stream); release_workspace(ws); } }
};

// The Forward survivors of every lane of a batch, gathered for ONE copy to the host: slot, Forward score, row offset,
// region count, expected number of domains and the regions themselves (the region scan keeps kRegionCap slots per
// survivor, 1.5 KB, of which a handful are used).  Survivor i of lane l goes to position lane_base[l] + i; its regions to
// reg_start (a cursor: any order), -1 when the packed region array is full (the host then reads that survivor's
// regions from the lane's own array).
struct PackArgs {
  const int32_t *list_fin; const float *fwd_by_item; int64_t slot_pitch;         // [lane][slot_pitch]
  const int64_t *xmx_off; const int32_t *reg_out; int64_t cap;                    // [lane][cap], [lane][cap * (kRegionCap*3 + 2)]
  const int *counters;                                                            // [lane][kLaneCounters]
  const int32_t *lane_base;                                                       // [nlanes]
  int32_t *fin; float *fwd; int64_t *off; int32_t *regn; float *nexp; int32_t *reg_start; int32_t *regs; int32_t regs_cap;
  int *cursor;
  int64_t tcap;             // survivors the packed arrays hold (a batch with more is packed again, to measure, by the collect half)
};
__global__ void pack_survivors_kernel(const PackArgs a)
{
  const int l = (int) blockIdx.y;
  const int nfin = a.counters[(size_t) l * kLaneCounters + 4];
  if (a.counters[(size_t) l * kLaneCounters + 12] != 0) return;          // flagged lane: redone on its own
  const int32_t *reg_l = a.reg_out + (size_t) l * (size_t) a.cap * (kRegionCap * 3 + 2);
  for (int i = (int) (blockIdx.x * blockDim.x + threadIdx.x); i < nfin; i += (int) (gridDim.x * blockDim.x)) {
    const int64_t t = (int64_t) a.lane_base[l] + i;
    if (t >= a.tcap) return;
    a.fin[t] = a.list_fin[(size_t) l * a.slot_pitch + i];
    a.fwd[t] = a.fwd_by_item[(size_t) l * a.slot_pitch + i];
    a.off[t] = a.xmx_off[(size_t) l * a.cap + i];
    const int n = reg_l[(size_t) a.cap * kRegionCap * 3 + i];
    a.regn[t] = n;
    a.nexp[t] = reinterpret_cast<const float *>(reg_l)[(size_t) a.cap * (kRegionCap * 3 + 1) + i];
    int start = -1;
    if (n > 0) {
      const int r0 = atomicAdd(a.cursor, n);
      if (r0 + n <= a.regs_cap) {
        start = r0;
        const int32_t *src = reg_l + (size_t) i * kRegionCap * 3;
        for (int z = 0; z < n * 3; ++z) a.regs[(size_t) r0 * 3 + z] = src[z];
      }
    }
    a.reg_start[t] = start;
  }
}

// first survivor of every lane in the packed arrays (exclusive prefix sum of the lanes' survivor counts, flagged lanes
// counting nothing -- what the collect half computes on the host from the counters), and the region cursor's reset
__global__ void __launch_bounds__(1024) lane_base_kernel(const int *counters, int nq, int32_t *lane_base, int *cursor)
{
  __shared__ int part[1024];
  const int t = (int) threadIdx.x, per = (nq + 1023) / 1024;
  auto count_of = [&](int l) { return counters[(size_t) l * kLaneCounters + 12] != 0 ? 0 : counters[(size_t) l * kLaneCounters + 4]; };
  int sum = 0;
  for (int i = 0; i < per; ++i) { const int l = t * per + i; if (l < nq) sum += count_of(l); }
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
    *cursor = 0;
  }
  __syncthreads();
  int run = part[t];
  for (int i = 0; i < per; ++i) { const int l = t * per + i; if (l < nq) { lane_base[l] = run; run += count_of(l); } }
}

// where the packed results of <T> survivors of <nq> lanes lie (device slab and its pinned mirror)
struct PackLayout { size_t o_base, o_cursor, o_off, o_fin, o_fwd, o_regn, o_nexp, o_start, o_regs, o_stage, bytes; int64_t regs_cap; };
static PackLayout pack_layout(int nq, int64_t T, bool scan_mode, int64_t nslots)
{
  PackLayout p{};
  p.regs_cap = std::max<int64_t>(8 * T, 4096);
  p.o_base = 0; p.o_cursor = (size_t) nq * 4; p.o_off = (p.o_cursor + 4 + 7) & ~(size_t) 7;
  p.o_fin = p.o_off + (size_t) T * 8; p.o_fwd = p.o_fin + (size_t) T * 4; p.o_regn = p.o_fwd + (size_t) T * 4; p.o_nexp = p.o_regn + (size_t) T * 4;
  p.o_start = p.o_nexp + (size_t) T * 4; p.o_regs = p.o_start + (size_t) T * 4; p.o_stage = p.o_regs + (size_t) p.regs_cap * 12;
  p.bytes = p.o_stage + (scan_mode ? (size_t) nq * (size_t) nslots : 0);
  return p;
}

// rows of the <count> longest targets (slots are sorted by decreasing length): bounds the rows of any <count> survivors
static int64_t longest_rows(const p7x_seqdb *db, int64_t count)
{
  int64_t rows = 0;
  count = std::min<int64_t>(count, db->nslots);
  for (int64_t sl = 0; sl < count; ++sl) rows += (int64_t) db->h_len[db->h_order[sl]] + 1;
  return rows;
}

// Argument records of the survivor passes (Forward with the special-state rows kept, Backward, region scan) of lanes
// [first, first + n), and the buffers behind them.  <retry>: one lane on the retry buffers, sized for its own nfin.
static int fill_survivor_args(CascadeRun &r, int first, int n, bool retry, int nfin)
{
  const p7x_seqdb *db = r.db; Workspace *ws = r.ws; DeviceCtx *ctx = r.ctx;
  int64_t want_cap = std::min<int64_t>(db->nslots, std::max<int64_t>(4096, db->nslots / 64));
  if (retry) want_cap = std::min<int64_t>(db->nslots, std::max<int64_t>(want_cap, (int64_t) nfin));
  // The lanes of a batch share the row arena (one cursor).  A batch of many profiles against a small block (the scan
  // orientation) can hold families with a dozen hits per profile: give it room for up to eight times the block's
  // longest <want_cap> targets, so that such a batch does not fall back to the one-lane-at-a-time retry below.
  const int64_t share = retry ? 1 : std::max<int64_t>(1, std::min<int64_t>(8, n / 32));
  const int64_t want_floats = longest_rows(db, want_cap) * 6 * share;
  if (want_floats > ws->xmx_cap) {
    // through the slab pool: hipFree would wait for the cascades of the other searches in flight
    slab_release(ctx, ws->xmx_f, ws->xmx_bytes[0]); slab_release(ctx, ws->xmx_b, ws->xmx_bytes[1]); slab_release(ctx, ws->xmx_s, ws->xmx_bytes[2]);
    ws->xmx_f = ws->xmx_b = ws->xmx_s = nullptr; ws->xmx_cap = 0;
    float **dst[3] = { &ws->xmx_f, &ws->xmx_b, &ws->xmx_s };
    for (int z = 0; z < 3; ++z) {
      void *dp = nullptr;
      const int sst = slab_acquire(ctx, (size_t) want_floats * 4, &dp, &ws->xmx_bytes[z]); if (sst != P7X_OK) return sst;
      *dst[z] = static_cast<float *>(dp);
    }
    ws->xmx_cap = want_floats;
  }
  int64_t cap = 0;
  if (!retry) {
    if (want_cap > ws->fin_cap) {
      (void) hipFree(ws->xmx_off); (void) hipFree(ws->reg_out); (void) hipFree(ws->bck_sc);
      ws->xmx_off = nullptr; ws->reg_out = nullptr; ws->bck_sc = nullptr; ws->fin_cap = 0;
      P7X_HIP(hipMalloc(&ws->xmx_off, (size_t) ws->nlanes * want_cap * 8));
      P7X_HIP(hipMalloc(&ws->reg_out, (size_t) ws->nlanes * want_cap * (kRegionCap * 3 + 2) * 4));
      P7X_HIP(hipMalloc(&ws->bck_sc, (size_t) ws->nlanes * want_cap * 4));
      ws->fin_cap = want_cap;
    }
    cap = ws->fin_cap;
  } else {
    if (want_cap > ws->rt_cap) {
      (void) hipFree(ws->rt_xmx_off); (void) hipFree(ws->rt_reg_out); (void) hipFree(ws->rt_bck_sc);
      ws->rt_xmx_off = nullptr; ws->rt_reg_out = nullptr; ws->rt_bck_sc = nullptr; ws->rt_cap = 0;
      P7X_HIP(hipMalloc(&ws->rt_xmx_off, (size_t) want_cap * 8));
      P7X_HIP(hipMalloc(&ws->rt_reg_out, (size_t) want_cap * (kRegionCap * 3 + 2) * 4));
      P7X_HIP(hipMalloc(&ws->rt_bck_sc, (size_t) want_cap * 4));
      ws->rt_cap = want_cap;
    }
    cap = ws->rt_cap;
  }
  for (int l = first; l < first + n; ++l) {
    const Profile &p = r.lm[l].om->p; const DevProfile *dp = r.lm[l].dp;
    LaneArgs &la = ws->h_args[l];
    const StageBufs b = ws->lane_bufs(l);
    int64_t *xmx_off = retry ? ws->rt_xmx_off : ws->xmx_off + (size_t) l * cap;
    int32_t *reg_out = retry ? ws->rt_reg_out : ws->reg_out + (size_t) l * cap * (kRegionCap * 3 + 2);
    float *bck_sc = retry ? ws->rt_bck_sc : ws->bck_sc + (size_t) l * cap;
    LayoutArgs lay{};
    lay.nfin_ptr = &b.counters[4]; lay.list_fin = b.list_fin; lay.slot_len = db->d_slot_len; lay.xmx_off = xmx_off;
    lay.cap_items = (int) std::min<int64_t>(cap, INT_MAX); lay.cap_floats = (long long) ws->xmx_cap; lay.cursor = ws->cursor();
    lay.flags = &b.counters[12];
    la.lay = lay;
    WaveSeqArgs a = ws_args(p, dp, db, ctx);
    a.trans = dp->fwd_trans; a.emis = dp->fwd_emis; a.list = b.list_fin; a.nlist_ptr = &b.counters[4];
    a.nlist = (int) std::min<int64_t>(cap, retry ? std::max(nfin, 1) : std::max(r.est_fin, 1));          // sizes the grid only
    a.out_sc = b.fwd_by_item; a.xmx = ws->xmx_f; a.xmx_off = xmx_off;
    a.abort_flag = &b.counters[12];
    la.rows = a;
    a.out_sc = bck_sc; a.xmx = ws->xmx_b; a.fwd_xmx = ws->xmx_f;
    la.bck = a;
    RegionArgs ra{};
    ra.nitems_ptr = &b.counters[4]; ra.abort_flag = &b.counters[12];
    ra.list = b.list_fin; ra.slot_len = db->d_slot_len; ra.fx = ws->xmx_f; ra.bx = ws->xmx_b;
    ra.xmx_off = xmx_off; ra.scratch = ws->xmx_s;
    ra.out_regs = reg_out; ra.out_n = reg_out + (size_t) cap * kRegionCap * 3;
    ra.out_nexpected = reinterpret_cast<float *>(ra.out_n + cap);
    ra.guard = (r.cfg.oa_guard > 0.0f && !r.cfg.long_targets && debug_opt(OPT_HOST_ORDER) <= 0) ? 2.0e-5f : 0.0f;      // (not when the host code is
                                                                          // asked for the device's order: the seam that validates the kernels against their twin)
    if (debug_opt(OPT_REGION_GUARD_PPM) >= 0 && !r.cfg.long_targets) ra.guard = 1.0e-6f * (float) debug_opt(OPT_REGION_GUARD_PPM);      // test seam: widen it (or 0: off)
    la.reg = ra;
  }
  return P7X_OK;
}

// the survivor passes of one class on stream <s>; the kernels read the survivor counts from device memory
static int launch_survivor_passes(CascadeRun &r, const LaneClass &c, hipStream_t s, bool retry, bool record_events)
{
  Workspace *ws = r.ws; DeviceCtx *ctx = r.ctx;
  const int64_t cap = retry ? ws->rt_cap : ws->fin_cap;
  int st = P7X_OK;
  hipLaunchKernelGGL(layout_rows_kernel, dim3(1, (unsigned) c.n), dim3(256), 0, s, lane_run(ws, &LaneArgs::lay, c.first, c.n).ref());
  P7X_HIP(hipGetLastError());
  if ((st = class_wave(c, ws, &LaneArgs::rows, false, ctx, s)) != P7X_OK) return st;
  if (record_events) P7X_HIP(hipEventRecord(ws->ev[5], s));
  if ((st = class_wave(c, ws, &LaneArgs::bck, true, ctx, s)) != P7X_OK) return st;
  {   // posterior decoding of the special states and the region scan, on the rows where they are
    const int64_t items = std::min<int64_t>(cap, retry ? cap : std::max(r.est_fin, 1));
    const unsigned gx = lane_grid((items + 3) / 4, ctx->num_cu * 4, c.n);
    hipLaunchKernelGGL(regions_kernel, dim3(gx, (unsigned) c.n), dim3(256), 0, s, lane_run(ws, &LaneArgs::reg, c.first, c.n).ref());
    P7X_HIP(hipGetLastError());
  }
  if (record_events) P7X_HIP(hipEventRecord(ws->ev[6], s));
  return P7X_OK;
}

// The Forward parser of the first pass (scores only) through the grouped kernel: against large blocks, where throughput counts
// (a small block's stages last as long as their longest chain: one target per wavefront is the shorter chain there, and the
// stage-by-stage cascade launches fwd_kernel by nodes per lane).  OFF by default (option fwd_grouped = 1 switches it on): the
// kernel issues 2.5 times fewer instructions per target and row (ISA: 334 per row of four targets at T = 16, C = 8 against
// about 260 per row of one) and is bit-for-bit as good within the parser's tolerance (tests/test_gpu_filters.py), but the
// many-profile workload did not get faster with it (22.4-22.6 against 22.6-22.75 TCUPS, profiles/r06_fwd_grouped.txt): in
// the pipeline the parser's wavefronts wait for their own dependent chains, not for issue slots.
static bool fwd_grouped(const p7x_seqdb *db) { return debug_opt(OPT_FWD_GROUPED) > 0 && db->ngroups > 256 && debug_opt(OPT_STAGE_MERGE) <= 0; }

// every kernel of stage 1 for the lanes of one class, on stream <s>
static int class_cascade(CascadeRun &r, const LaneClass &c, hipStream_t s, bool record_events, bool chain_msv, bool msv_by_tier = false)
{
  const p7x_seqdb *db = r.db; Workspace *ws = r.ws; DeviceCtx *ctx = r.ctx;
  int st = P7X_OK;
  if (chain_msv) {
    // MSV launches that fill the device on their own are chained (two of them sharing the CUs only slow each other
    // down); small blocks -- a scan's query sequences -- leave most of the device idle and run side by side instead
    std::lock_guard<std::mutex> lk(ctx->msv_mu);          // enqueue order == chain order
    if (ctx->msv_last >= 0) P7X_HIP(hipStreamWaitEvent(s, ctx->msv_done[ctx->msv_last], 0));
    if (record_events) P7X_HIP(hipEventRecord(ws->ev[0], s));
    if ((st = class_msv(c, r.lm, ctx, ws, s)) != P7X_OK) return st;
    if (record_events) P7X_HIP(hipEventRecord(ws->ev[7], s));
    ctx->msv_last = (ctx->msv_last + 1) & 1;
    P7X_HIP(hipEventRecord(ctx->msv_done[ctx->msv_last], s));
  } else {
    if (record_events) P7X_HIP(hipEventRecord(ws->ev[0], s));
    if ((st = msv_by_tier ? class_msv_rest(c, r.lm, ctx, ws, s) : class_msv(c, r.lm, ctx, ws, s)) != P7X_OK) return st;
    if (record_events) P7X_HIP(hipEventRecord(ws->ev[7], s));
  }
  const ArgRef dec = lane_run(ws, &LaneArgs::dec, c.first, c.n).ref();
  hipLaunchKernelGGL(decide_msv_kernel, dim3((unsigned) ((db->nslots + 255) / 256), (unsigned) c.n), dim3(256), 0, s, dec);
  if (record_events) P7X_HIP(hipEventRecord(ws->ev[1], s));
  hipLaunchKernelGGL(bias_kernel, dim3(lane_grid((r.est_bias + 63) / 64, ctx->num_cu * 4, c.n), (unsigned) c.n), dim3(64), 0, s, dec);
  {   // the Viterbi work list, in slot order
    const unsigned nchunks = (unsigned) ((db->nslots + kCompactChunk - 1) / kCompactChunk);
    const ArgRef cmp = lane_run(ws, &LaneArgs::cmp, c.first, c.n).ref();
    hipLaunchKernelGGL(vit_compact_count_kernel, dim3(nchunks, (unsigned) c.n), dim3(256), 0, s, cmp);
    hipLaunchKernelGGL(vit_compact_write_kernel, dim3(nchunks, (unsigned) c.n), dim3(256), 0, s, cmp);
  }
  if (record_events) P7X_HIP(hipEventRecord(ws->ev[2], s));
  if ((st = class_viterbi(c, r.lm, ctx, ws, s)) != P7X_OK) return st;
  hipLaunchKernelGGL(decide_vit_kernel, dim3(lane_grid((r.est_vit + 255) / 256, ctx->num_cu, c.n), (unsigned) c.n), dim3(256), 0, s, dec);
  if (record_events) P7X_HIP(hipEventRecord(ws->ev[3], s));
  if (fwd_grouped(db) && r.lm[c.first].dp->fwdgT > 0) {       // scores only: several targets per wavefront (p7x_fwdpk.hip)
    if ((st = fwdg_launch(lane_run(ws, &LaneArgs::fwd, c.first, c.n), ctx->num_cu, s)) != P7X_OK) return st;
  } else if ((st = class_wave(c, ws, &LaneArgs::fwd, false, ctx, s)) != P7X_OK) return st;
  hipLaunchKernelGGL(decide_fwd_kernel, dim3(lane_grid((r.est_fwd + 255) / 256, ctx->num_cu, c.n), (unsigned) c.n), dim3(256), 0, s, dec);
  P7X_HIP(hipGetLastError());
  if (record_events) P7X_HIP(hipEventRecord(ws->ev[4], s));
  return launch_survivor_passes(r, c, s, false, record_events);
}

struct CascadeRun;
struct PackLayout;
static int queue_pack(CascadeRun &r, const PackLayout &lay, int64_t T, const int32_t *lane_base, int max_nfin);

// The batch stage by stage instead of class by class (multi-class batches against a small block: the scan orientation).
// The kernels that are the same for every lane -- the MSV decision, the bias filter, the compaction of the Viterbi work
// list, the Viterbi and Forward decisions, the row layout, the region scan -- run ONCE over all lanes of the batch on the
// workspace's stream; only the stages whose kernels are instantiated per model size fan out over the side streams (MSV
// by tier, Viterbi by (kernel, nodes per lane), the parsers by nodes per lane) and join again.  Against a block of a few
// dozen 64-target groups every launch is latency bound (it lasts as long as the longest target takes one wavefront or,
// for the bias filter, one lane), the hardware runs eight launches at a time, and a class-by-class batch of 28 classes is
// 28 x 16 of them; stage by stage it is about 28 + 28 + 12 + 24 + 9.
static int staged_cascade(CascadeRun &r, const std::vector<LaneClass> &classes, hipStream_t s, int nside, const std::vector<int> &run_of)
{
  const p7x_seqdb *db = r.db; Workspace *ws = r.ws; DeviceCtx *ctx = r.ctx;
  const int nq = (int) r.lm.size();
  constexpr int kStreams = Workspace::kSide + 1;
  int st = P7X_OK;
  auto stream_of = [&](size_t n) -> hipStream_t {
    int turn = (int) (n % (size_t) kStreams);
    if ((n / (size_t) kStreams) & 1) turn = kStreams - 1 - turn;
    turn = std::min(turn, nside);
    return turn == 0 ? s : ws->side[turn - 1];
  };
  auto fork = [&]() -> int {
    P7X_HIP(hipEventRecord(ws->ev_fork, s));
    for (int k = 0; k < nside; ++k) P7X_HIP(hipStreamWaitEvent(ws->side[k], ws->ev_fork, 0));
    return P7X_OK;
  };
  auto join = [&]() -> int {
    for (int k = 0; k < nside; ++k) {
      P7X_HIP(hipEventRecord(ws->ev_join[k], ws->side[k]));
      P7X_HIP(hipStreamWaitEvent(s, ws->ev_join[k], 0));
    }
    return P7X_OK;
  };
  // runs of adjacent classes that agree in <same>, as classes of their own (longest models first)
  auto runs_by = [&](auto same) {
    std::vector<LaneClass> out;
    for (const LaneClass &c : classes) {
      if (!out.empty() && same(out.back(), c)) out.back().n += c.n;
      else out.push_back(c);
    }
    std::reverse(out.begin(), out.end());
    return out;
  };
  // stage A: MSV.  The tier launches are already queued (run_of); what is left per class is the exact kernel over the
  // ambiguous groups, or the whole MSV stage of the classes outside the tiers (wave-per-target kernels).
  P7X_HIP(hipEventRecord(ws->ev[0], s));
  for (size_t n = 0; n < classes.size(); ++n) {
    const size_t c = classes.size() - 1 - n;
    hipStream_t cs = stream_of(n);
    if (run_of[c] >= 0) P7X_HIP(hipStreamWaitEvent(cs, ws->ev_tier[run_of[c]], 0));
    if ((st = run_of[c] >= 0 ? class_msv_rest(classes[c], r.lm, ctx, ws, cs) : class_msv(classes[c], r.lm, ctx, ws, cs)) != P7X_OK) return st;
  }
  if ((st = join()) != P7X_OK) return st;
  P7X_HIP(hipEventRecord(ws->ev[7], s));
  const ArgRef dec = lane_run(ws, &LaneArgs::dec, 0, nq).ref();
  hipLaunchKernelGGL(decide_msv_kernel, dim3((unsigned) ((db->nslots + 255) / 256), (unsigned) nq), dim3(256), 0, s, dec);
  P7X_HIP(hipEventRecord(ws->ev[1], s));
  hipLaunchKernelGGL(bias_kernel, dim3(lane_grid((r.est_bias + 63) / 64, ctx->num_cu * 4, nq), (unsigned) nq), dim3(64), 0, s, dec);
  {
    const unsigned nchunks = (unsigned) ((db->nslots + kCompactChunk - 1) / kCompactChunk);
    const ArgRef cmp = lane_run(ws, &LaneArgs::cmp, 0, nq).ref();
    hipLaunchKernelGGL(vit_compact_count_kernel, dim3(nchunks, (unsigned) nq), dim3(256), 0, s, cmp);
    hipLaunchKernelGGL(vit_compact_write_kernel, dim3(nchunks, (unsigned) nq), dim3(256), 0, s, cmp);
  }
  P7X_HIP(hipGetLastError());
  P7X_HIP(hipEventRecord(ws->ev[2], s));
  // stage B: Viterbi, by (kernel instantiation, nodes per lane)
  if ((st = fork()) != P7X_OK) return st;
  {
    const std::vector<LaneClass> runs = runs_by([](const LaneClass &a, const LaneClass &b) { return a.vit_key == b.vit_key && a.C == b.C && a.vit_long == b.vit_long; });
    for (size_t n = 0; n < runs.size(); ++n)
      if ((st = class_viterbi(runs[n], r.lm, ctx, ws, stream_of(n))) != P7X_OK) return st;
  }
  if ((st = join()) != P7X_OK) return st;
  hipLaunchKernelGGL(decide_vit_kernel, dim3(lane_grid((r.est_vit + 255) / 256, ctx->num_cu, nq), (unsigned) nq), dim3(256), 0, s, dec);
  P7X_HIP(hipEventRecord(ws->ev[3], s));
  // stage C: the Forward parser, by nodes per lane
  const std::vector<LaneClass> by_C = runs_by([](const LaneClass &a, const LaneClass &b) { return a.C == b.C; });
  if ((st = fork()) != P7X_OK) return st;
  for (size_t n = 0; n < by_C.size(); ++n)
    if ((st = class_wave(by_C[n], ws, &LaneArgs::fwd, false, ctx, stream_of(n))) != P7X_OK) return st;
  if ((st = join()) != P7X_OK) return st;
  hipLaunchKernelGGL(decide_fwd_kernel, dim3(lane_grid((r.est_fwd + 255) / 256, ctx->num_cu, nq), (unsigned) nq), dim3(256), 0, s, dec);
  P7X_HIP(hipEventRecord(ws->ev[4], s));
  hipLaunchKernelGGL(layout_rows_kernel, dim3(1, (unsigned) nq), dim3(256), 0, s, lane_run(ws, &LaneArgs::lay, 0, nq).ref());
  P7X_HIP(hipGetLastError());
  P7X_HIP(hipEventRecord(ws->ev[5], s));
  // stage D: Forward rows and Backward of the survivors, by nodes per lane; then the region scan of all lanes
  if ((st = fork()) != P7X_OK) return st;
  for (size_t n = 0; n < by_C.size(); ++n) {
    hipStream_t cs = stream_of(n);
    if ((st = class_wave(by_C[n], ws, &LaneArgs::rows, false, ctx, cs)) != P7X_OK) return st;
    if ((st = class_wave(by_C[n], ws, &LaneArgs::bck, true, ctx, cs)) != P7X_OK) return st;
  }
  if ((st = join()) != P7X_OK) return st;
  {
    const int64_t items = std::min<int64_t>(ws->fin_cap, std::max(r.est_fin, 1));
    const unsigned gx = lane_grid((items + 3) / 4, ctx->num_cu * 4, nq);
    hipLaunchKernelGGL(regions_kernel, dim3(gx, (unsigned) nq), dim3(256), 0, s, lane_run(ws, &LaneArgs::reg, 0, nq).ref());
    P7X_HIP(hipGetLastError());
  }
  P7X_HIP(hipEventRecord(ws->ev[6], s));
  return P7X_OK;
}

static int cascade_enqueue(CascadeRun &r)
{
  const p7x_pipeline_cfg &cfg = r.cfg; const p7x_seqdb *db = r.db;
  int st = get_ctx(db->device, &r.ctx);
  if (st != P7X_OK) return st;
  DeviceCtx *ctx = r.ctx;
  const int nq = (int) r.oms.size();
  // lanes in order of model length: the lanes of one kernel instantiation are then consecutive
  r.query_of.resize(nq);
  for (int i = 0; i < nq; ++i) r.query_of[i] = i;
  std::stable_sort(r.query_of.begin(), r.query_of.end(), [&](int a, int b) { return r.oms[a]->p.M < r.oms[b]->p.M; });
  r.lane_of.resize(nq); r.lm.resize(nq);
  for (int l = 0; l < nq; ++l) { r.lane_of[r.query_of[l]] = l; r.lm[l].om = r.oms[r.query_of[l]]; }
  const bool debug = debug_opt(OPT_TRACE_FINISH) > 0;
  auto tlast = std::chrono::steady_clock::now();
  std::string dbg;
  auto tick = [&](const char *what) {
    if (!debug) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[64]; std::snprintf(buf, sizeof buf, " %s %.2f", what, std::chrono::duration<double, std::milli>(now - tlast).count());
    dbg += buf; tlast = now;
  };
  // device images of profiles seen for the first time (a scan: all of them): laid out by the host workers, one slab
  // and one copy for the batch
  {
    std::vector<const p7x_oprofile *> lane_oms((size_t) nq);
    std::vector<DevProfile *> dps((size_t) nq, nullptr);
    for (int l = 0; l < nq; ++l) lane_oms[(size_t) l] = r.lm[l].om;
    if ((st = get_dev_profiles(lane_oms.data(), nq, ctx, dps.data(), cfg.host_threads)) != P7X_OK) return st;
    for (int l = 0; l < nq; ++l) r.lm[l].dp = dps[(size_t) l];
  }
  tick("images");
  if (db->nslots == 0 || nq == 0) return P7X_OK;
  for (int l = 0; l < nq; ++l)
    if (r.lm[l].dp->vitC <= 0) { set_error("model too long for the device kernels: M > 8192 nodes (the reference has no limit, plan7.pyx:6156-6262; the lane-chunk layout of the wave-per-target kernels ends at 128 nodes per lane -- no Pfam-A model comes near it)"); return P7X_EINVAL; }
  std::vector<LaneClass> classes;
  if ((st = lane_classes(r.lm, db, ctx, classes)) != P7X_OK) return st;
  if ((st = get_workspace(db->device, db->nslots, nq, &r.ws)) != P7X_OK) return st;
  tick("classes+workspace");
  Workspace *ws = r.ws;
  hipStream_t s = ws->stream;
  r.queued = true;
  const ListEstimates est = estimate_lists(cfg, db->nslots);
  r.est_bias = est.bias; r.est_vit = est.vit; r.est_fwd = est.fwd; r.est_fin = est.fin;
  std::vector<int> nlong_of((size_t) nq, 0);
  std::vector<int64_t> vit_long_of((size_t) nq, 0);
  for (const LaneClass &c : classes) for (int l = c.first; l < c.first + c.n; ++l) { nlong_of[(size_t) l] = c.nlong; vit_long_of[(size_t) l] = c.vit_long; }
  for (int l = 0; l < nq; ++l) {
    const Profile &p = r.lm[l].om->p; const DevProfile *dp = r.lm[l].dp;
    LaneArgs &la = ws->h_args[l];
    const StageBufs b = ws->lane_bufs(l);
    DecideArgs d{};
    d.b = b; d.p = make_params(p, cfg);
    d.slot_len = db->d_slot_len; d.tjb_tab = ctx->lt.tjb; d.null1_tab = ctx->lt.null1; d.xwmove_tab = ctx->lt.xwmove; d.logtab = ctx->lt.logtab;
    d.dsq = db->d_dsq; d.slot_off = db->d_slot_off; d.eo = dp->bias_eo; d.nslots = db->nslots;
    la.dec = d;
    fill_msv_args(la, p, dp, db, ctx, b, nlong_of[(size_t) l]);
    const int64_t vit_long = vit_long_of[(size_t) l];
    fill_vit_args(la, p, dp, db, ctx, b.list_vit, est.vit, &b.counters[2], b.xC,
                  vit_long > 0 ? &b.counters[13] : nullptr, (int) std::min<int64_t>(vit_long, (int64_t) est.vit));
    CompactArgs ca{};
    ca.stage = b.stage; ca.list = b.list_vit; ca.chunk_cnt = ws->chunk_cnt + (size_t) l * (size_t) ws->chunks_per_lane; ca.counters = b.counters;
    ca.nslots = db->nslots; ca.long_slots = vit_long;
    la.cmp = ca;
    WaveSeqArgs a = ws_args(p, dp, db, ctx);
    a.trans = dp->fwd_trans; a.emis = dp->fwd_emis; a.list = b.list_fwd; a.nlist = est.fwd; a.nlist_ptr = &b.counters[3];
    a.out_sc = b.fwd_by_item;
    if (fwd_grouped(db) && dp->fwdgT > 0) { a.trans = dp->fwdg_trans; a.emis = dp->fwdg_emis; a.C = dp->fwdgT * 256 + dp->fwdgC; a.counter = &b.counters[5]; }
    la.fwd = a;
  }
  tick("args");
  if ((st = fill_survivor_args(r, 0, nq, false, 0)) != P7X_OK) return st;
  tick("survivor_args");
  if ((st = upload_args(ws, 0, nq, s)) != P7X_OK) return st;
  tick("upload");
  P7X_HIP(hipMemsetAsync(ws->counters, 0, ws->counters_bytes(), s));      // lane counters and the arena cursor
  ws->ntier_runs = 0;
  if (classes.size() == 1) {
    const bool fills_device = (db->nslots / 64) * (int64_t) nq >= (int64_t) ctx->num_cu * 8;
    if ((st = class_cascade(r, classes[0], s, true, fills_device)) != P7X_OK) return st;
  } else {
    // the classes in turn over the workspace's stream and its side streams (eight streams = the eight hardware queues:
    // with the workspace's stream taking only the first class its queue idled through most of a 28-class batch of the
    // scan orientation, whose device phase is the longest queue's sum of kernel latencies), forked from and joined into
    // the workspace's stream
    P7X_HIP(hipEventRecord(ws->ev_fork, s));
    const int nside = std::min<int>((int) classes.size() - 1, Workspace::kSide);
    for (int k = 0; k < nside; ++k) P7X_HIP(hipStreamWaitEvent(ws->side[k], ws->ev_fork, 0));
    // The fast MSV kernel of the whole batch first, as one launch per tier of register tiles (p7x_msv.hip); the classes'
    // chains then wait for their tier and start at the exact kernel over the ambiguous groups.
    constexpr int kStreamsAll = Workspace::kSide + 1;
    std::vector<int> run_of(classes.size(), -1);
    if (msv_tiers_enabled()) {
      int nruns = 0, last_tier = -1;
      for (size_t c = 0; c < classes.size(); ++c) {
        const LaneClass &k = classes[c];
        if (k.msv_key < 0) { last_tier = -1; continue; }
        const int tier = msv_tier(r.lm[k.first].dp->msvR, r.lm[k.first].dp->msvK);
        if (tier != last_tier) { if (nruns == Workspace::kTierRuns) break; ++nruns; last_tier = tier; }
        run_of[c] = nruns - 1;
      }
      for (int run = nruns - 1; run >= 0; --run) {
        size_t a = 0; while (run_of[a] != run) ++a;
        size_t b = a; while (b + 1 < classes.size() && run_of[b + 1] == run) ++b;
        const int first = classes[a].first, n = classes[b].first + classes[b].n - first;
        const int turn = (kStreamsAll - 1 - run) % kStreamsAll;       // from the far end: the longest classes' chains are dealt from stream 0
        hipStream_t ts = turn == 0 ? s : ws->side[std::min(turn, nside) - 1];
        const LaneClass &k = classes[a];
        P7X_HIP(hipEventRecord(ws->ev_tier0[run], ts));
        if ((st = msv_tier_launch(msv_tier(r.lm[k.first].dp->msvR, r.lm[k.first].dp->msvK), lane_run(ws, &LaneArgs::msv, first, n), ctx->num_cu, ts)) != P7X_OK) return st;
        P7X_HIP(hipEventRecord(ws->ev_tier[run], ts));
        ws->tier_lanes[run] = n; ws->tier_nodes[run] = 0;
        for (int l = first; l < first + n; ++l) ws->tier_nodes[run] += r.lm[l].om->p.M;
      }
      ws->ntier_runs = nruns;
    }
    // a small block: stage by stage (option stage_merge: 0 never, 1 always, unset: blocks of up to 256 groups)
    const int sm = debug_opt(OPT_STAGE_MERGE);
    const bool staged = sm >= 0 ? sm != 0 : db->ngroups <= 256;
    if (staged) {
      if ((st = staged_cascade(r, classes, s, nside, run_of)) != P7X_OK) return st;
    } else
    for (size_t n = 0; n < classes.size(); ++n) {
      // the classes of the longest models first (their chains are the longest: started last they would end the batch alone),
      // dealt back and forth (in one direction the first stream would get the heaviest class of every round)
      const size_t c = classes.size() - 1 - n;
      constexpr int kStreams = Workspace::kSide + 1;
      int turn = (int) (n % (size_t) kStreams);
      if ((n / (size_t) kStreams) & 1) turn = kStreams - 1 - turn;
      hipStream_t cs = turn == 0 ? s : ws->side[turn - 1];
      if (run_of[c] >= 0) P7X_HIP(hipStreamWaitEvent(cs, ws->ev_tier[run_of[c]], 0));
      if ((st = class_cascade(r, classes[c], cs, n == 0, false, run_of[c] >= 0)) != P7X_OK) return st;
    }
    for (int k = 0; k < nside; ++k) {
      P7X_HIP(hipEventRecord(ws->ev_join[k], ws->side[k]));
      P7X_HIP(hipStreamWaitEvent(s, ws->ev_join[k], 0));
    }
  }
  // The survivors' results travel with the cascade: gathered and copied behind its last kernel for up to kEarlyPack of them
  // (the collect half used to queue this after it had read the counts: a second round trip on a stream that shares its
  // hardware queue with other batches' launches -- 20-90 ms of a scan batch's feeder thread, for a 10 us kernel).
  ws->early_T_cap = 0;
  // Against small blocks only (up to 256 groups: the scan orientation, the long-target pipeline's windows), where the second
  // round trip waited behind other batches' launches; the many-profile search of a 500,000-target block LOST 2.7 % with it
  // (22.1 instead of 22.7 TCUPS, whatever the size of the copy), the headline neither gained nor lost.
  // Option early_pack: 0 off, n > 0 always, with room for n survivors (tests: a batch that does not fit).
  if (debug_opt(OPT_EARLY_PACK) > 0 || (debug_opt(OPT_EARLY_PACK) < 0 && db->ngroups <= 256)) {
    // room for twice the survivors of this workspace's last batch (1,024 ... 16,384): the arrays are copied whole
    int64_t room = 1024;
    while (room < 16384 && room < 2 * ws->last_T) room *= 2;
    const int64_t kEarlyPack = debug_opt(OPT_EARLY_PACK) > 0 ? debug_opt(OPT_EARLY_PACK) : room;
    const PackLayout lay = pack_layout(nq, kEarlyPack, cfg.mode == P7X_SCAN_MODELS, db->nslots);
    if ((st = queue_pack(r, lay, kEarlyPack, nullptr, 2048)) != P7X_OK) return st;
    ws->early_T_cap = kEarlyPack;
  }
  P7X_HIP(hipMemcpyAsync(ws->h_counts, ws->counters, ws->counters_bytes(), hipMemcpyDeviceToHost, s));
  P7X_HIP(hipEventRecord(ws->ev_sync, s));
  tick("launches");
  if (debug) std::fprintf(stderr, "[enqueue] nq %d classes %zu:%s ms\n", nq, classes.size(), dbg.c_str());
  return P7X_OK;
}

// one lane again, after its survivors did not fit the shared buffers
static int retry_survivor_passes(CascadeRun &r, int l, int nfin)
{
  Workspace *ws = r.ws; hipStream_t s = ws->stream;
  int st = fill_survivor_args(r, l, 1, true, nfin);
  if (st != P7X_OK) return st;
  if ((st = upload_args(ws, l, 1, s)) != P7X_OK) return st;
  P7X_HIP(hipMemsetAsync(&ws->lane_bufs(l).counters[12], 0, 4, s));
  P7X_HIP(hipMemsetAsync(ws->cursor(), 0, 8, s));
  LaneClass c; c.first = l; c.n = 1; c.C = r.lm[l].dp->vitC;
  if ((st = launch_survivor_passes(r, c, s, true, false)) != P7X_OK) return st;
  P7X_HIP(hipMemcpyAsync(ws->h_counts, ws->counters, ws->counters_bytes(), hipMemcpyDeviceToHost, s));
  P7X_HIP(hipEventRecord(ws->ev_sync, s));
  return P7X_OK;
}

// download one lane's per-survivor results (queued on the workspace's stream; the caller synchronises)
static int fetch_lane(CascadeRun &r, int l, bool retry, CascadeOut &out)
{
  Workspace *ws = r.ws; hipStream_t s = ws->stream;
  const int nfin = out.counts[4];
  out.fin_slots.resize(nfin); out.fwdsc.resize(nfin); out.xmx_off.resize(nfin);
  if (nfin == 0) return P7X_OK;
  const StageBufs b = ws->lane_bufs(l);
  const int64_t cap = retry ? ws->rt_cap : ws->fin_cap;
  const int64_t *xmx_off = retry ? ws->rt_xmx_off : ws->xmx_off + (size_t) l * cap;
  const int32_t *reg_out = retry ? ws->rt_reg_out : ws->reg_out + (size_t) l * cap * (kRegionCap * 3 + 2);
  out.regs.resize((size_t) nfin * kRegionCap * 3); out.reg_n.resize(nfin); out.nexpected.resize(nfin);
  P7X_HIP(hipMemcpyAsync(out.fin_slots.data(), b.list_fin, (size_t) nfin * 4, hipMemcpyDeviceToHost, s));
  P7X_HIP(hipMemcpyAsync(out.xmx_off.data(), xmx_off, (size_t) nfin * 8, hipMemcpyDeviceToHost, s));
  P7X_HIP(hipMemcpyAsync(out.regs.data(), reg_out, out.regs.size() * 4, hipMemcpyDeviceToHost, s));
  P7X_HIP(hipMemcpyAsync(out.reg_n.data(), reg_out + (size_t) cap * kRegionCap * 3, (size_t) nfin * 4, hipMemcpyDeviceToHost, s));
  P7X_HIP(hipMemcpyAsync(out.nexpected.data(), reg_out + (size_t) cap * (kRegionCap * 3 + 1), (size_t) nfin * 4, hipMemcpyDeviceToHost, s));
  // the rows pass recomputed each survivor's Forward score in list order: one contiguous copy
  P7X_HIP(hipMemcpyAsync(out.fwdsc.data(), b.fwd_by_item, (size_t) nfin * 4, hipMemcpyDeviceToHost, s));
  return P7X_OK;
}

// after the lane's results have arrived: the parsers' rows when the host has to scan them itself
static int fetch_lane_rows(CascadeRun &r, CascadeOut &out)
{
  const p7x_seqdb *db = r.db; Workspace *ws = r.ws;
  const int nfin = out.counts[4];
  if (nfin == 0) return P7X_OK;
  bool overflow = r.cfg.host_regions != 0;
  for (int i = 0; i < nfin; ++i) if (out.reg_n[i] == -2) overflow = true;
  if (!overflow) return P7X_OK;     // a target with more regions than the device keeps (or the A/B switch): the host scans the rows
  // the lane's blocks are contiguous in the arena (one cursor step), in list order
  const int64_t lo = out.xmx_off[0];
  const int64_t hi = out.xmx_off[(size_t) nfin - 1] + (int64_t) (db->h_len[db->h_order[out.fin_slots[(size_t) nfin - 1]]] + 1) * 6;
  out.fwd_xmx.resize((size_t) (hi - lo)); out.bck_xmx.resize((size_t) (hi - lo));
  P7X_HIP(hipMemcpy(out.fwd_xmx.data(), ws->xmx_f + lo, (size_t) (hi - lo) * 4, hipMemcpyDeviceToHost));
  P7X_HIP(hipMemcpy(out.bck_xmx.data(), ws->xmx_b + lo, (size_t) (hi - lo) * 4, hipMemcpyDeviceToHost));
  for (auto &o : out.xmx_off) o -= lo;
  out.have_xmx = true;
  return P7X_OK;
}

// Queue, on the workspace's stream: the gather of the survivors' results into the packed arrays of <lay> and their copy
// to the pinned mirror (scan orientation: with the per-target stage bytes).  <lane_base> == nullptr: the lanes' first
// positions are computed on the device (the enqueue half, which does not know the counts yet), else they are the host's.
static int queue_pack(CascadeRun &r, const PackLayout &lay, int64_t T, const int32_t *lane_base, int max_nfin)
{
  Workspace *ws = r.ws; const p7x_seqdb *db = r.db; hipStream_t s = ws->stream;
  const int nq = (int) r.oms.size();
  const bool scan_mode = r.cfg.mode == P7X_SCAN_MODELS;
  int st = P7X_OK;
  if (lay.bytes > ws->pack_dev_bytes) {
    slab_release(r.ctx, ws->pack_dev, ws->pack_dev_bytes); ws->pack_dev = nullptr; ws->pack_dev_bytes = 0;
    void *dp = nullptr; size_t got = 0;
    if ((st = slab_acquire(r.ctx, lay.bytes + lay.bytes / 2, &dp, &got)) != P7X_OK) return st;
    ws->pack_dev = static_cast<unsigned char *>(dp); ws->pack_dev_bytes = got;
  }
  if (lay.bytes > ws->pack_host_bytes) {
    pinned_release(ws->pack_host, ws->pack_host_bytes); ws->pack_host = nullptr; ws->pack_host_bytes = 0;
    void *hp = nullptr; size_t got = 0;
    if ((st = pinned_acquire(lay.bytes + lay.bytes / 2, &hp, &got)) != P7X_OK) return st;
    ws->pack_host = static_cast<unsigned char *>(hp); ws->pack_host_bytes = got;
  }
  unsigned char *ph = ws->pack_host, *pdv = ws->pack_dev;
  if (T > 0) {
    if (lane_base) {
      std::memcpy(ph + lay.o_base, lane_base, (size_t) nq * 4);
      *reinterpret_cast<int32_t *>(ph + lay.o_cursor) = 0;
      P7X_HIP(hipMemcpyAsync(pdv, ph, lay.o_cursor + 4, hipMemcpyHostToDevice, s));
    } else {
      hipLaunchKernelGGL(lane_base_kernel, dim3(1), dim3(1024), 0, s, ws->counters, nq, reinterpret_cast<int32_t *>(pdv + lay.o_base),
                         reinterpret_cast<int *>(pdv + lay.o_cursor));
    }
    PackArgs pa{};
    pa.list_fin = ws->list_fin; pa.fwd_by_item = ws->fwd_by_item; pa.slot_pitch = ws->cap_slots;
    pa.xmx_off = ws->xmx_off; pa.reg_out = ws->reg_out; pa.cap = ws->fin_cap;
    pa.counters = ws->counters; pa.lane_base = reinterpret_cast<const int32_t *>(pdv + lay.o_base);
    pa.fin = reinterpret_cast<int32_t *>(pdv + lay.o_fin); pa.fwd = reinterpret_cast<float *>(pdv + lay.o_fwd);
    pa.off = reinterpret_cast<int64_t *>(pdv + lay.o_off); pa.regn = reinterpret_cast<int32_t *>(pdv + lay.o_regn);
    pa.nexp = reinterpret_cast<float *>(pdv + lay.o_nexp); pa.reg_start = reinterpret_cast<int32_t *>(pdv + lay.o_start);
    pa.regs = reinterpret_cast<int32_t *>(pdv + lay.o_regs); pa.regs_cap = (int32_t) std::min<int64_t>(lay.regs_cap, INT32_MAX);
    pa.cursor = reinterpret_cast<int *>(pdv + lay.o_cursor);
    pa.tcap = T;
    const unsigned gx = (unsigned) std::max(1, std::min(64, (max_nfin + 255) / 256));
    hipLaunchKernelGGL(pack_survivors_kernel, dim3(gx, (unsigned) nq), dim3(256), 0, s, pa);
    P7X_HIP(hipGetLastError());
    P7X_HIP(hipMemcpyAsync(ph + lay.o_cursor, pdv + lay.o_cursor, lay.o_regs - lay.o_cursor, hipMemcpyDeviceToHost, s));
    // regions: the cursor tells how many there are, but waiting for it costs a round trip; the array is small
    P7X_HIP(hipMemcpyAsync(ph + lay.o_regs, pdv + lay.o_regs, (size_t) lay.regs_cap * 12, hipMemcpyDeviceToHost, s));
  }
  if (scan_mode)           // per-target accounting: which filters every (model, sequence) pair passed
    P7X_HIP(hipMemcpy2DAsync(ph + lay.o_stage, (size_t) db->nslots, ws->stage, (size_t) ws->cap_slots, (size_t) db->nslots, nq, hipMemcpyDeviceToHost, s));
  return P7X_OK;
}

static int cascade_collect(CascadeRun &r, std::vector<CascadeOut> &outs)
{
  const int nq = (int) r.oms.size();
  outs.assign((size_t) nq, CascadeOut{});
  if (!r.queued || r.collected) return P7X_OK;
  const p7x_pipeline_cfg &cfg = r.cfg; const p7x_seqdb *db = r.db; Workspace *ws = r.ws;
  hipStream_t s = ws->stream;
  struct Release { CascadeRun &r; ~Release() { (void) hipStreamSynchronize(r.ws->stream); release_workspace(r.ws); r.collected = true; } } release{ r };
  int st = P7X_OK;
  P7X_HIP(hipSetDevice(db->device));                      // the collecting thread may have driven another device since
  const bool debug = debug_opt(OPT_TRACE_FINISH) > 0;
  const auto tc0 = std::chrono::steady_clock::now();
  P7X_HIP(hipEventSynchronize(ws->ev_sync));              // our work only: other cascades run on other streams
  const auto tc1 = std::chrono::steady_clock::now();
  std::vector<int> flagged;
  std::vector<int32_t> lane_base((size_t) nq, 0);
  int64_t T = 0; int max_nfin = 0;
  for (int l = 0; l < nq; ++l) {
    CascadeOut &out = outs[(size_t) r.query_of[l]];
    std::memcpy(out.counts, ws->h_counts + (size_t) l * kLaneCounters, kLaneCounters * 4);
    lane_base[(size_t) l] = (int32_t) T;
    if (out.counts[12] != 0) flagged.push_back(l);
    else { T += out.counts[4]; max_nfin = std::max(max_nfin, out.counts[4]); }
  }
  // One gather kernel and one copy bring the survivors' results of all lanes to the host (pinned): the region scan's
  // own array has kRegionCap slots per survivor, and lane-by-lane or strided copies of it were most of this call.
  // Normally the enqueue half queued both behind the cascade (queue_pack) and they are here already; a batch with more
  // survivors than it made room for is packed again, to measure.
  const bool scan_mode = cfg.mode == P7X_SCAN_MODELS;
  const bool early = ws->early_T_cap > 0 && T <= ws->early_T_cap;
  const PackLayout lay = pack_layout(nq, early ? ws->early_T_cap : T, scan_mode, db->nslots);
  ws->early_T_cap = 0;
  ws->last_T = T;
  const size_t o_fin = lay.o_fin, o_fwd = lay.o_fwd, o_regn = lay.o_regn, o_nexp = lay.o_nexp, o_start = lay.o_start, o_regs = lay.o_regs;
  const size_t o_off = lay.o_off, o_stage = lay.o_stage;
  if (!early) {
    if ((st = queue_pack(r, lay, T, lane_base.data(), max_nfin)) != P7X_OK) return st;
    P7X_HIP(hipEventRecord(ws->ev_sync, s)); P7X_HIP(hipEventSynchronize(ws->ev_sync));
  }
  unsigned char *ph = ws->pack_host;
  const uint8_t *by_slot = ph + o_stage;
  if (T > 0) {
    const int32_t *h_fin = reinterpret_cast<const int32_t *>(ph + o_fin), *h_regn = reinterpret_cast<const int32_t *>(ph + o_regn);
    const int32_t *h_start = reinterpret_cast<const int32_t *>(ph + o_start), *h_regs = reinterpret_cast<const int32_t *>(ph + o_regs);
    const float *h_fwd = reinterpret_cast<const float *>(ph + o_fwd), *h_nexp = reinterpret_cast<const float *>(ph + o_nexp);
    const int64_t *h_off = reinterpret_cast<const int64_t *>(ph + o_off);
    for (int l = 0; l < nq; ++l) {
      CascadeOut &out = outs[(size_t) r.query_of[l]];
      const int nfin = out.counts[4];
      if (out.counts[12] != 0 || nfin == 0) continue;
      const size_t base = (size_t) lane_base[(size_t) l];
      out.fin_slots.assign(h_fin + base, h_fin + base + nfin);
      out.fwdsc.assign(h_fwd + base, h_fwd + base + nfin);
      out.xmx_off.assign(h_off + base, h_off + base + nfin);
      out.reg_n.assign(h_regn + base, h_regn + base + nfin);
      out.nexpected.assign(h_nexp + base, h_nexp + base + nfin);
      out.reg_start.assign((size_t) nfin, 0);
      out.regs.clear();
      for (int i = 0; i < nfin; ++i) {
        const int n = out.reg_n[(size_t) i];
        if (n <= 0) continue;
        out.reg_start[(size_t) i] = (int32_t) (out.regs.size() / 3);
        const int32_t rs = h_start[base + (size_t) i];
        if (rs >= 0) out.regs.insert(out.regs.end(), h_regs + (size_t) rs * 3, h_regs + (size_t) (rs + n) * 3);
        else {        // the packed array was full: this survivor's regions from the lane's own array
          const size_t at = out.regs.size();
          out.regs.resize(at + (size_t) n * 3);
          const int32_t *src = ws->reg_out + (size_t) l * (size_t) ws->fin_cap * (kRegionCap * 3 + 2) + (size_t) i * kRegionCap * 3;
          P7X_HIP(hipMemcpy(out.regs.data() + at, src, (size_t) n * 12, hipMemcpyDeviceToHost));
        }
      }
    }
  }
  for (int l = 0; l < nq; ++l) {
    CascadeOut &out = outs[(size_t) r.query_of[l]];
    if (out.counts[12] == 0 && (st = fetch_lane_rows(r, out)) != P7X_OK) return st;
  }
  // lanes whose survivors did not fit (more of them than the shared buffers were sized for): one at a time, with
  // buffers sized for the lane's own count
  for (int l : flagged) {
    CascadeOut &out = outs[(size_t) r.query_of[l]];
    if ((st = retry_survivor_passes(r, l, out.counts[4])) != P7X_OK) return st;
    P7X_HIP(hipEventSynchronize(ws->ev_sync));
    std::memcpy(out.counts, ws->h_counts + (size_t) l * kLaneCounters, kLaneCounters * 4);
    if (out.counts[12] != 0) { set_error("row buffers could not be sized for the Forward survivors"); return P7X_EMEM; }
    if ((st = fetch_lane(r, l, true, out)) != P7X_OK) return st;
    P7X_HIP(hipEventRecord(ws->ev_sync, s)); P7X_HIP(hipEventSynchronize(ws->ev_sync));
    if ((st = fetch_lane_rows(r, out)) != P7X_OK) return st;
  }
  // Forward survivors inside the F3 guard band (normally none): their slots, matched against the survivor list
  for (int l = 0; l < nq; ++l) {
    CascadeOut &out = outs[(size_t) r.query_of[l]];
    const int nnear = out.counts[14];
    if (nnear <= 0 || out.fin_slots.empty()) continue;
    std::vector<int32_t> slots((size_t) nnear);
    P7X_HIP(hipMemcpy(slots.data(), ws->lane_bufs(l).list_bias, (size_t) nnear * 4, hipMemcpyDeviceToHost));
    std::sort(slots.begin(), slots.end());
    out.near.assign(out.fin_slots.size(), 0);
    for (size_t i = 0; i < out.fin_slots.size(); ++i) out.near[i] = std::binary_search(slots.begin(), slots.end(), out.fin_slots[i]) ? 1 : 0;
  }
  double ms[12]{};
  for (int i = 0; i < 6; ++i) { float t = 0; (void) hipEventElapsedTime(&t, ws->ev[i], ws->ev[i + 1]); ms[i] = t; }
  { float t = 0; (void) hipEventElapsedTime(&t, ws->ev[0], ws->ev[7]); ms[7] = t; }
  ms[8] = nq; ms[9] = nq; ms[10] = 0.0;
  for (int l = 0; l < nq; ++l) ms[10] += r.lm[l].om->p.M;
  if (ws->ntier_runs > 0) {
    // a batch of several classes: slot 7 is the batch's LARGEST fast-MSV launch (a tier of register tiles, p7x_msv.hip), by its own events
    int big = 0;
    for (int k = 1; k < ws->ntier_runs; ++k) if (ws->tier_nodes[k] > ws->tier_nodes[big]) big = k;
    float t = 0;
    if (hipEventElapsedTime(&t, ws->ev_tier0[big], ws->ev_tier[big]) == hipSuccess) { ms[7] = t; ms[9] = ws->tier_lanes[big]; ms[10] = (double) ws->tier_nodes[big]; }
  }
  if (debug) {
    const auto tc2 = std::chrono::steady_clock::now();
    long long nfin_tot = 0; for (const CascadeOut &o : outs) nfin_tot += o.counts[4];
    std::fprintf(stderr, "[collect] nq %d survivors %lld flagged %zu: device wait %.2f fetch %.2f ms; events msv %.2f bias %.2f vit %.2f fwd %.2f rows %.2f bck+regions %.2f (msv kernel %.2f)\n",
                 nq, nfin_tot, flagged.size(), std::chrono::duration<double, std::milli>(tc1 - tc0).count(),
                 std::chrono::duration<double, std::milli>(tc2 - tc1).count(), ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[7]);
  }
  for (int l = 0; l < nq; ++l) {
    CascadeOut &out = outs[(size_t) r.query_of[l]];
    std::memcpy(out.ms, ms, sizeof(ms));
    if (cfg.mode == P7X_SCAN_MODELS) {
      out.stage.assign((size_t) db->n, 0);
      const uint8_t *row = by_slot + (size_t) l * (size_t) db->nslots;
      for (int64_t sl = 0; sl < db->nslots; ++sl) out.stage[(size_t) db->h_order[sl]] = row[(size_t) sl];
    }
  }
  return P7X_OK;
}

} // namespace p7x

using namespace p7x;


extern "C" {

// ---------------------------------------------------------------------------- batched raw filter outputs
int p7x_filters_batch(const p7x_oprofile *om, const p7x_seqdb *db, int32_t *xJ, int32_t *xC, float *fwd, float *bias_filtersc)
{
  if (!om || !db) { set_error("p7x_filters_batch: bad arguments"); return P7X_EINVAL; }
  DeviceCtx *ctx = nullptr;
  int st = get_ctx(db->device, &ctx);
  if (st != P7X_OK) return st;
  const Profile &p = om->p;
  std::vector<LaneModel> lm(1);
  lm[0].om = om;
  if ((st = get_dev_profile(om, ctx, &lm[0].dp)) != P7X_OK) return st;
  DevProfile *dp = lm[0].dp;
  const int64_t ns = db->nslots;
  LaneClass cls;
  {
    const bool small = small_block(db, ctx, 1);
    cls.first = 0; cls.n = 1; cls.msv_key = msv_key_of(dp, small); cls.vit_key = vit_key_of(dp, small); cls.C = dp->vitC;
    cls.nlong = (!small && cls.msv_key >= 0 && dp->msvw_emis) ? long_groups(db, ctx, 1) : 0;
    if (xJ && cls.msv_key < 0 && !dp->msvw_emis) { set_error("model too long for the MSV kernels (M > 8192)"); return P7X_EINVAL; }
  }
  for (int64_t t = 0; t < db->n; ++t) {
    if (xJ) xJ[t] = 0;
    if (xC) xC[t] = -32768;
    if (fwd) fwd[t] = -INFINITY;
    if (bias_filtersc) bias_filtersc[t] = 0.0f;
  }
  if (ns == 0) return P7X_OK;
  Workspace *ws = nullptr;
  if ((st = get_workspace(db->device, ns, 1, &ws)) != P7X_OK) return st;
  WorkspaceLease lease{ ws };
  hipStream_t s = ctx->stream;
  const StageBufs b = ws->lane_bufs(0);
  LaneArgs &la = ws->h_args[0];
  p7x_pipeline_cfg cfg; p7x_pipeline_cfg_default(&cfg);
  {
    DecideArgs d{};
    d.b = b; d.p = make_params(p, cfg);
    d.p.F1 = 2.0; d.p.F2 = 2.0;              // the bias pass below scores every target
    d.slot_len = db->d_slot_len; d.tjb_tab = ctx->lt.tjb; d.null1_tab = ctx->lt.null1; d.xwmove_tab = ctx->lt.xwmove; d.logtab = ctx->lt.logtab;
    d.dsq = db->d_dsq; d.slot_off = db->d_slot_off; d.eo = dp->bias_eo; d.nslots = ns;
    la.dec = d;
    fill_msv_args(la, p, dp, db, ctx, b, cls.nlong);
    if (dp->vitC > 0) {
      fill_vit_args(la, p, dp, db, ctx, nullptr, (int) ns, nullptr, b.xC);
      WaveSeqArgs a = ws_args(p, dp, db, ctx);
      a.trans = dp->fwd_trans; a.emis = dp->fwd_emis; a.list = nullptr; a.nlist = (int) ns; a.nlist_ptr = nullptr; a.out_sc = b.fwd_by_item;
      la.fwd = a;
    }
  }
  if ((st = upload_args(ws, 0, 1, s)) != P7X_OK) return st;
  P7X_HIP(hipMemsetAsync(ws->counters, 0, ws->counters_bytes(), s));
  if (xJ) {
    if ((st = class_msv(cls, lm, ctx, ws, s)) != P7X_OK) return st;
    std::vector<int16_t> h((size_t) ns);
    P7X_HIP(hipMemcpyAsync(h.data(), b.xJ, (size_t) ns * 2, hipMemcpyDeviceToHost, s));
    P7X_HIP(hipStreamSynchronize(s));
    for (int64_t sl = 0; sl < ns; ++sl) xJ[db->h_order[sl]] = h[sl];
  }
  if (xC || fwd) {
    if (dp->vitC <= 0) { set_error("model too long for the wave-per-sequence kernels"); return P7X_EINVAL; }
    if (xC) {
      if ((st = class_viterbi(cls, lm, ctx, ws, s)) != P7X_OK) return st;
      std::vector<int32_t> h((size_t) ns);
      P7X_HIP(hipMemcpyAsync(h.data(), b.xC, (size_t) ns * 4, hipMemcpyDeviceToHost, s));
      P7X_HIP(hipStreamSynchronize(s));
      for (int64_t sl = 0; sl < ns; ++sl) xC[db->h_order[sl]] = h[sl];
    }
    if (fwd) {
      if ((st = class_wave(cls, ws, &LaneArgs::fwd, false, ctx, s)) != P7X_OK) return st;
      std::vector<float> h((size_t) ns);
      P7X_HIP(hipMemcpyAsync(h.data(), b.fwd_by_item, (size_t) ns * 4, hipMemcpyDeviceToHost, s));
      P7X_HIP(hipStreamSynchronize(s));
      for (int64_t sl = 0; sl < ns; ++sl) fwd[db->h_order[sl]] = h[sl];
    }
  }
  if (bias_filtersc) {
    // run the bias kernel over every target: list_bias = identity
    std::vector<int32_t> ident((size_t) ns);
    for (int64_t i = 0; i < ns; ++i) ident[i] = (int32_t) i;
    P7X_HIP(hipMemcpyAsync(b.list_bias, ident.data(), (size_t) ns * 4, hipMemcpyHostToDevice, s));
    int cnt = (int) ns;
    P7X_HIP(hipMemcpyAsync(&b.counters[1], &cnt, 4, hipMemcpyHostToDevice, s));
    std::vector<float> zero((size_t) ns, 0.0f);
    P7X_HIP(hipMemcpyAsync(b.usc, zero.data(), (size_t) ns * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(bias_kernel, dim3(ctx->num_cu * 4), dim3(64), 0, s, lane_run(ws, &LaneArgs::dec, 0, 1).ref());
    std::vector<float> h((size_t) ns);
    P7X_HIP(hipMemcpyAsync(h.data(), b.filtersc, (size_t) ns * 4, hipMemcpyDeviceToHost, s));
    P7X_HIP(hipStreamSynchronize(s));
    for (int64_t sl = 0; sl < ns; ++sl) bias_filtersc[db->h_order[sl]] = h[sl];
  }
  return P7X_OK;
}

// ---------------------------------------------------------------------------- single-sequence seam
static int run_cascade(const p7x_pipeline_cfg &cfg, const p7x_oprofile *om, const p7x_seqdb *db, CascadeOut &out)
{
  CascadeRun r;
  r.cfg = cfg; r.oms.assign(1, om); r.db = db;
  int st = cascade_enqueue(r);
  if (st != P7X_OK) return st;
  std::vector<CascadeOut> outs;
  if ((st = cascade_collect(r, outs)) != P7X_OK) return st;
  out = std::move(outs[0]);
  return P7X_OK;
}

static int one_seq(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, int which, float *sc)
{
  if (!om || !dsq || !sc || L < 1) { set_error("bad arguments"); return P7X_EINVAL; }
  const int64_t off = 0; const int32_t len = L;
  p7x_seqdb *db = nullptr;
  int st = p7x_seqdb_create(device, om->p.abc_type, dsq, &off, &len, 1, &db);
  if (st != P7X_OK) return st;
  int32_t xJ = 0, xC = 0; float f = 0;
  const Profile &p = om->p;
  DeviceCtx *ctx = nullptr;
  get_ctx(device, &ctx);
  if (which == 0) {
    st = p7x_filters_batch(om, db, &xJ, nullptr, nullptr, nullptr);
    if (st == P7X_OK) {
      if (xJ < 0) { *sc = INFINITY; st = P7X_ERANGE; }
      else {
        const uint8_t tjb = unbiased_byteify(p.scale_b, logf(3.0f / (float) (L + 3)));
        float v = ((float) (xJ - tjb) - (float) p.base_b); v /= p.scale_b; v -= 3.0; *sc = v;
      }
    }
  } else if (which == 1) {
    st = p7x_filters_batch(om, db, nullptr, &xC, nullptr, nullptr);
    if (st == P7X_OK) {
      if (xC >= 32767) { *sc = INFINITY; st = P7X_ERANGE; }
      else if (xC > -32768) {
        const float pmove = 3.0f / ((float) L + 3.0f);
        float v = (float) xC + (float) wordify(p.scale_w, logf(pmove)) - (float) p.base_w; v /= p.scale_w; v -= 3.0; *sc = v;
      } else *sc = -INFINITY;
    }
  } else if (which == 2) {
    st = p7x_filters_batch(om, db, nullptr, nullptr, &f, nullptr);
    *sc = f;
    if (st == P7X_OK && (std::isnan(f) || std::isinf(f))) st = P7X_ERANGE;
  } else {
    // Backward: run Forward with rows, then Backward
    p7x_pipeline_cfg cfg; p7x_pipeline_cfg_default(&cfg);
    cfg.do_max = 1;
    cfg.host_regions = 1;                 // fetch the parsers' rows
    CascadeOut out;
    st = run_cascade(cfg, om, db, out);
    if (st == P7X_OK) {
      if (out.fin_slots.size() != 1) { set_error("backward: sequence did not reach the Backward stage"); st = P7X_EINVAL; }
      else {
        // score = totscale + log(xN(0)): recompute from the stored rows
        double tot = 0.0;
        for (int i = 1; i <= L; ++i) { const float s = out.bck_xmx[(size_t) i * 6 + 5]; if (s > 1.0f) tot += std::log((double) s); }
        *sc = (float) (tot + std::log((double) out.bck_xmx[1]));
      }
    }
  }
  p7x_seqdb_destroy(db);
  return st;
}

int p7x_msv_filter(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc) { return one_seq(om, device, dsq, L, 0, sc); }
int p7x_vit_filter(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc) { return one_seq(om, device, dsq, L, 1, sc); }
int p7x_fwd_parser(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc) { return one_seq(om, device, dsq, L, 2, sc); }
int p7x_bck_parser(const p7x_oprofile *om, int device, const uint8_t *dsq, int32_t L, float *sc) { return one_seq(om, device, dsq, L, 3, sc); }

// ---------------------------------------------------------------------------- the search
// Two stages, so that a caller with many queries can overlap them (hmmer.hmmsearch does): enqueue / wait = the filter
// cascade and the parsers on the device for a batch of query profiles against one resident target block, finish =
// domain definition on the host (+ the envelope kernel on its own stream), one hit list per query.  One query alone
// (p7x_search_block_*) is a batch of one.
struct p7x_pending {
  p7x_pipeline_cfg cfg{};
  const p7x_seqdb *db = nullptr;
  CascadeRun run;
  std::vector<CascadeOut> co;             // caller order
  bool waited = false;
  std::chrono::steady_clock::time_point t0;
};

int p7x_search_batch_enqueue(const p7x_pipeline_cfg *cfg, const p7x_oprofile *const *oms, size_t nq, const float *bg_f,
                             const p7x_seqdb *db, p7x_pending **out)
{
  if (!cfg || !oms || !db || !out || nq == 0 || nq > 4096) { set_error("p7x_search_batch_enqueue: bad arguments"); return P7X_EINVAL; }
  *out = nullptr;
  auto pd = std::make_unique<p7x_pending>();
  pd->t0 = std::chrono::steady_clock::now();
  for (size_t q = 0; q < nq; ++q) {
    if (!oms[q]) { set_error("p7x_search_batch_enqueue: null profile"); return P7X_EINVAL; }
    const Profile &p = oms[q]->p;
    if (p.abc_type != db->abc_type) { set_error("profile and target block have different alphabets"); return P7X_EINVAL; }
    // The bias filter's composition odds and the null2 background are baked into the optimized profile when it is
    // converted (p7_bg_SetFilter / p7_ProfileConfig with the Background given then).  A pipeline whose background
    // differs from that one would silently score against the wrong null model: refuse it instead.
    if (bg_f) for (int x = 0; x < p.K; ++x) if (bg_f[x] != p.bgf[x]) {
      set_error("the pipeline's background differs from the one the optimized profile was configured with"); return P7X_EINVAL; }
    if (cfg->use_bit_cutoffs) {   // p7_pli_NewModelThresholds: eslEINVAL when the model lacks the cutoffs
      const int i = cfg->use_bit_cutoffs == P7X_BITCUT_GA ? P7X_GA1 : (cfg->use_bit_cutoffs == P7X_BITCUT_TC ? P7X_TC1 : P7X_NC1);
      if (p.cutoff[i] == P7X_CUTOFF_UNSET || p.cutoff[i + 1] == P7X_CUTOFF_UNSET) { set_error("model is missing the requested bit score cutoffs"); return P7X_EINVAL; }
    }
  }
  pd->cfg = *cfg; pd->db = db;
  pd->run.cfg = *cfg; pd->run.oms.assign(oms, oms + nq); pd->run.db = db;
  const int st = cascade_enqueue(pd->run);
  if (st != P7X_OK) return st;
  *out = pd.release();
  return P7X_OK;
}

int p7x_debug_log_of_float(int device, const float *in, float *out, size_t n)
{
  if (!in || !out) { set_error("p7x_debug_log_of_float: bad arguments"); return P7X_EINVAL; }
  if (n == 0) return P7X_OK;
  DeviceCtx *ctx = nullptr;
  const int st = get_ctx(device, &ctx);
  if (st != P7X_OK) return st;
  struct Dev { void *p = nullptr; ~Dev() { if (p) (void) hipFree(p); } } b_in, b_out;       // released on every return path
  P7X_HIP(hipMalloc(&b_in.p, n * 4)); P7X_HIP(hipMalloc(&b_out.p, n * 4));
  float *d_in = static_cast<float *>(b_in.p), *d_out = static_cast<float *>(b_out.p);
  P7X_HIP(hipMemcpy(d_in, in, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(log_of_float_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_in, d_out, n, ctx->lt.logtab);
  P7X_HIP(hipGetLastError());
  P7X_HIP(hipStreamSynchronize(ctx->stream));
  P7X_HIP(hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost));
  return P7X_OK;
}

int p7x_search_batch_raw(const p7x_pipeline_cfg *cfg, const p7x_oprofile *const *oms, size_t nq, const float *bg_f,
                         const p7x_seqdb *db, int32_t *xJ, int32_t *xC, uint8_t *stage)
{
  p7x_pending *pd = nullptr;
  int st = p7x_search_batch_enqueue(cfg, oms, nq, bg_f, db, &pd);
  if (st != P7X_OK) return st;
  std::unique_ptr<p7x_pending> owner(pd);
  CascadeRun &r = pd->run;
  const int64_t n = db->n, ns = db->nslots;
  if (xJ) std::fill(xJ, xJ + nq * (size_t) n, 0);
  if (xC) std::fill(xC, xC + nq * (size_t) n, INT32_MIN);
  if (stage) std::fill(stage, stage + nq * (size_t) n, (uint8_t) 0);
  if (!r.queued || ns == 0) return P7X_OK;
  Workspace *ws = r.ws;
  P7X_HIP(hipStreamSynchronize(ws->stream));          // the batch as the search runs it, kernels chosen per class
  std::vector<int16_t> hj((size_t) ns); std::vector<int32_t> hl((size_t) ns), hc((size_t) ns); std::vector<uint8_t> hs((size_t) ns);
  for (size_t q = 0; q < nq; ++q) {
    const int l = r.lane_of[q];
    const StageBufs b = ws->lane_bufs(l);
    int counters[kLaneCounters];
    P7X_HIP(hipMemcpy(counters, b.counters, sizeof(counters), hipMemcpyDeviceToHost));
    if (xJ) {
      P7X_HIP(hipMemcpy(hj.data(), b.xJ, (size_t) ns * 2, hipMemcpyDeviceToHost));
      for (int64_t sl = 0; sl < ns; ++sl) xJ[q * (size_t) n + (size_t) db->h_order[sl]] = hj[(size_t) sl];
    }
    if (xC) {
      const int nv = counters[2];
      P7X_HIP(hipMemcpy(hl.data(), b.list_vit, (size_t) nv * 4, hipMemcpyDeviceToHost));
      P7X_HIP(hipMemcpy(hc.data(), b.xC, (size_t) nv * 4, hipMemcpyDeviceToHost));
      for (int it = 0; it < nv; ++it) xC[q * (size_t) n + (size_t) db->h_order[hl[(size_t) it]]] = hc[(size_t) it];
    }
    if (stage) {
      P7X_HIP(hipMemcpy(hs.data(), b.stage, (size_t) ns, hipMemcpyDeviceToHost));
      for (int64_t sl = 0; sl < ns; ++sl) stage[q * (size_t) n + (size_t) db->h_order[sl]] = hs[(size_t) sl];
    }
  }
  return p7x_search_block_wait(pd);                   // collect (and release the workspace) as a search would
}

int p7x_search_block_enqueue(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const float *bg_f, const p7x_seqdb *db,
                             p7x_pending **out)
{
  if (!cfg || !om || !db || !out) { set_error("p7x_search_block_enqueue: bad arguments"); return P7X_EINVAL; }
  return p7x_search_batch_enqueue(cfg, &om, 1, bg_f, db, out);
}

int p7x_search_block_wait(p7x_pending *pd)
{
  if (!pd) { set_error("p7x_search_block_wait: bad arguments"); return P7X_EINVAL; }
  if (pd->waited) return P7X_OK;
  const int st = cascade_collect(pd->run, pd->co);
  if (st != P7X_OK) return st;
  pd->waited = true;
  const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pd->t0).count();   // stage 1 wall
  for (auto &c : pd->co) c.ms[6] = wall;
  return P7X_OK;
}

size_t p7x_pending_nqueries(const p7x_pending *pd) { return pd ? pd->run.oms.size() : 0; }

int p7x_search_block_begin(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const float *bg_f, const p7x_seqdb *db,
                           p7x_pending **out)
{
  if (!out) { set_error("p7x_search_block_begin: bad arguments"); return P7X_EINVAL; }
  p7x_pending *pd = nullptr;
  int st = p7x_search_block_enqueue(cfg, om, bg_f, db, &pd);
  if (st != P7X_OK) return st;
  if ((st = p7x_search_block_wait(pd)) != P7X_OK) { delete pd; *out = nullptr; return st; }
  *out = pd;
  return P7X_OK;
}

int p7x_search_batch_finish(p7x_pending *pd, const char *const *names, const char *const *accs, const char *const *descs,
                            p7x_tophits **outs)
{
  if (!pd || !outs) { set_error("p7x_search_batch_finish: bad arguments"); return P7X_EINVAL; }
  std::unique_ptr<p7x_pending> owner(pd);                // consumed, also on failure
  const size_t nq = pd->run.oms.size();
  for (size_t q = 0; q < nq; ++q) outs[q] = nullptr;
  if (!pd->waited) { const int wst = p7x_search_block_wait(pd); if (wst != P7X_OK) return wst; }
  const p7x_seqdb *db = pd->db;
  HostTargets tg;
  tg.n = db->n; tg.nres = db->nres; tg.len = db->h_len.data(); tg.off = db->h_off.data(); tg.dsq = db->h_dsq.data();
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<std::vector<int32_t>> targets(nq);
  std::vector<DeviceRegions> dr(nq);
  std::vector<FinishItem> items(nq);
  bool any_device = false;
  for (size_t q = 0; q < nq; ++q) {
    const p7x_oprofile *om = pd->run.oms[q]; CascadeOut &co = pd->co[q];
    targets[q].resize(co.fin_slots.size());
    for (size_t i = 0; i < targets[q].size(); ++i) targets[q][i] = db->h_order[co.fin_slots[i]];
    FinishItem &it = items[q];
    it.om = om; it.targets = &targets[q]; it.fwdsc = co.fwdsc.data();
    it.near = &co.near;
    it.fwd_xmx = co.fwd_xmx.data(); it.bck_xmx = co.bck_xmx.data(); it.xmx_off = co.xmx_off.data();
    it.counts[0] = (uint64_t) co.counts[1]; it.counts[1] = (uint64_t) co.counts[8]; it.counts[2] = (uint64_t) co.counts[3]; it.counts[3] = (uint64_t) co.counts[4];
    it.ms = co.ms; it.nms = 12;
    if (!co.have_xmx && !targets[q].empty()) { dr[q].n = co.reg_n.data(); dr[q].regs = co.regs.data(); dr[q].nexpected = co.nexpected.data(); dr[q].cap = kRegionCap;
      dr[q].start = co.reg_start.empty() ? nullptr : co.reg_start.data(); it.regions = &dr[q]; }
    it.device_envelopes = !pd->cfg.host_envelopes && !targets[q].empty();
    any_device = any_device || it.device_envelopes;
  }
  int st = P7X_OK;
  std::unique_ptr<EnvelopeScorer> scorer, scorer2;      // single-domain envelopes; second round: the ensembles' clustered envelopes
  std::unique_ptr<EnsembleRunner> ensembles;
  if (any_device) {
    DeviceCtx *ctx = nullptr;
    if ((st = get_ctx(db->device, &ctx)) != P7X_OK) return st;
    scorer = make_device_envelope_scorer(ctx, db, pd->cfg.oa_guard);
    if (!pd->cfg.host_ensembles) {          // the multi-domain regions' ensembles and their clustered envelopes on the device as well
      ensembles = make_device_ensemble_runner(ctx, db, pd->cfg.ens_guard);
      scorer2 = make_device_envelope_scorer(ctx, db, pd->cfg.oa_guard);
    } else if (device_clustered(pd->cfg.host_threads)) scorer2 = make_device_envelope_scorer(ctx, db, pd->cfg.oa_guard);
  }
  const auto t2 = std::chrono::steady_clock::now();
  if ((st = host_finish_batch(pd->cfg, items, tg, names, accs, descs, outs, scorer.get(), scorer2.get(), ensembles.get())) != P7X_OK) return st;
  // work time of this batch (stage 1 + stage 2), not the time it spent queued between the stages
  const auto t3 = std::chrono::steady_clock::now();
  const double stage2 = std::chrono::duration<double, std::milli>(t3 - t1).count();
  for (size_t q = 0; q < nq; ++q) {
    CascadeOut &co = pd->co[q];
    if (!co.stage.empty()) tophits_set_stages(outs[q], std::move(co.stage));
    tophits_set_total_ms(outs[q], co.ms[6], stage2);
  }
  if (debug_opt(OPT_TRACE_FINISH) > 0) {       // where the call spends its time, the teardown of its helpers included
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t4 = std::chrono::steady_clock::now();
    ensembles.reset(); scorer2.reset(); scorer.reset();
    const auto t5 = std::chrono::steady_clock::now();
    owner.reset();
    const auto t6 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[batch_finish] nq %zu: items %.2f helpers %.2f host_finish_batch %.2f stages %.2f helpers' teardown %.2f pending's teardown %.2f ms\n",
                 nq, ms(t1, t2) - 0.0, 0.0, ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, t6));
  }
  return P7X_OK;
}

int p7x_search_block_finish(p7x_pending *pd, const char *const *names, const char *const *accs, const char *const *descs,
                            p7x_tophits **out)
{
  if (!pd || !out) { set_error("p7x_search_block_finish: bad arguments"); return P7X_EINVAL; }
  if (pd->run.oms.size() != 1) { delete pd; set_error("p7x_search_block_finish: the handle holds a batch (use p7x_search_batch_finish)"); return P7X_EINVAL; }
  return p7x_search_batch_finish(pd, names, accs, descs, out);
}

void p7x_pending_destroy(p7x_pending *pd) { delete pd; }

int p7x_search_block(const p7x_pipeline_cfg *cfg, const p7x_oprofile *om, const float *bg_f, const p7x_seqdb *db,
                     const char *const *names, const char *const *accs, const char *const *descs, p7x_tophits **out)
{
  if (!cfg || !om || !db || !out) { set_error("p7x_search_block: bad arguments"); return P7X_EINVAL; }
  p7x_pending *pd = nullptr;
  const int st = p7x_search_block_begin(cfg, om, bg_f, db, &pd);
  if (st != P7X_OK) return st;
  return p7x_search_block_finish(pd, names, accs, descs, out);
}

} // extern "C"

namespace p7x {
// Forward / Backward parsers + region scan of EVERY target of <db> (the long-target pipeline's Forward survivors, packed
// as a block of windows): the survivor passes of the cascade with all filters open.  out[t] in the caller's order.
int device_regions_of_all(const p7x_oprofile *om, const p7x_seqdb *db, std::vector<LongTargetWindowRegions> &out)
{
  out.assign((size_t) db->n, LongTargetWindowRegions{});
  if (db->n == 0) return P7X_OK;
  p7x_pipeline_cfg cfg; p7x_pipeline_cfg_default(&cfg);
  cfg.do_max = 1;
  CascadeOut co;
  const int st = run_cascade(cfg, om, db, co);
  if (st != P7X_OK) return st;
  if (co.have_xmx) return P7X_OK;                      // a window with more regions than the device keeps: the host scans (n stays -2)
  for (size_t i = 0; i < co.fin_slots.size(); ++i) {
    LongTargetWindowRegions &w = out[(size_t) db->h_order[(size_t) co.fin_slots[i]]];
    const int n = co.reg_n[i];
    w.n = n < 0 ? (n == -1 ? -1 : -2) : n;
    w.nexpected = co.nexpected[i];
    if (n <= 0) continue;
    const size_t start = co.reg_start.empty() ? i * (size_t) kRegionCap : (size_t) co.reg_start[i];
    w.regs.resize((size_t) n);
    for (int z = 0; z < n; ++z) w.regs[(size_t) z] = Region{ co.regs[(start + (size_t) z) * 3], co.regs[(start + (size_t) z) * 3 + 1], co.regs[(start + (size_t) z) * 3 + 2] != 0 };
  }
  return P7X_OK;
}
}
