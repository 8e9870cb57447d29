// p7x_ssvlong.hip -- p7_SSVFilter_longtarget on CDNA4: the SSV scan of nhmmer over a chromosome, both strands.
//
// Upstream (impl_sse/msvfilter.c: p7_SSVFilter_longtarget; reference p7_pipeline.pxd:131-143, called from
// p7_Pipeline_LongTarget) walks the target once with the single-segment recurrence
//     M_i[k] = sat( max(M_{i-1}[k-1], xB) + bias - rbv[x_i][k] ),    xB = base - tjb - tbm  (constant: no J state),
// and whenever a cell reaches the score threshold that corresponds to P = F1 it emits the diagonal through that cell as
// a window seed, zeroes the row and skips ahead.  The scan is >99 % of nhmmer's work: L x M cells per strand.
//
// Here the scan is RESET-FREE and parallel: the strand is cut into chunks, one wavefront per chunk, each with M rows
// of warm-up so that every diagonal that can reach a row of the chunk is complete.  A reset can only lower scores, so
// the rows that reach the threshold upstream are a subset of the rows reported here; the host replays upstream's
// sequential bookkeeping (choice of the seed cell, diagonal recovery and extension, skip-ahead) on the reported rows
// (p7x_longtarget.inc.hpp).
//
// Layout: the model is split across the 64 lanes, R packed int16 pairs per lane.  Global register g = lane*R + j holds
// cells (2g-1, 2g) on odd rows and (2g, 2g+1) on even rows (the MSV kernel's parity trick: the diagonal move costs
// nothing inside a lane, and one DPP move per odd row carries the last register of a lane to the next lane).  Cells
// are stored relative to the constant begin score, s = v - xB - 32768, so v_pk_add_i16 clamp computes
// max(M[k-1], xB) + e in one instruction; with the row-maximum update that is 2 packed ops per 2 cells.  The emission
// scores of A, C, G, T for both parities sit in LDS ([parity][x][j][lane], one conflict-free ds_read_b32 per register);
// degenerate residues (rare in chromosomes) take a slow path through the full table in global memory.  u8 saturation at
// 255 is not reproduced: it can only keep a cell at or above a threshold it has already reached.
#include "p7x_wave.hpp"
#include <mutex>

namespace p7x {

namespace {

typedef short s2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_adds_u(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s2w, a), __builtin_bit_cast(s2w, b))); }
__device__ __forceinline__ uint32_t pk_max_u(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2w, a), __builtin_bit_cast(s2w, b))); }
constexpr uint32_t kFloor2 = 0x80008000u;

}  // namespace

// one chunk of one strand per wavefront
template <int R>
__global__ void __launch_bounds__(256) ssvlong_kernel(const SsvLongArgs a)
{
  // LDS: emission pairs of the four canonical residues, [parity][x][j][lane]
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < 2 * 4 * R * 64; i += 256) lds[i] = a.tab4[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int nwaves = (int) gridDim.x * 4;

  for (long long chi = wave; chi < a.nchunks; chi += nwaves) {
    const long long ch = a.chunk_list ? a.chunk_list[chi] : chi;
    // chunk ch of strand s: rows first .. last (1-based positions on that strand), preceded by up to M warm-up rows
    const int strand = a.strand0 + (int) (ch / a.chunks_per_strand);  // 0: as given, 1: reverse complement
    const long long c0 = (ch % a.chunks_per_strand) * (long long) a.chunk_len;
    const long long first = c0 + 1, last = min(a.L, c0 + a.chunk_len);
    const long long warm = max(1LL, first - a.M);
    uint32_t v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = kFloor2;
    for (long long i0 = warm; i0 <= last; i0 += 64) {
      const int nrow = (int) min(64LL, last - i0 + 1);
      // residues of the next 64 rows, one per lane: strand 1 reads the target backwards and complements
      uint32_t res = 0;
      if (lane < nrow) {
        const long long pos = i0 + lane;                              // position on this strand
        const long long src = strand == 0 ? pos : a.L - pos + 1;      // position in the stored sequence
        const uint32_t x = a.dsq[src];
        res = strand == 0 || x >= (uint32_t) a.Kp ? x : (uint32_t) a.comp[x];       // codes outside the alphabet separate targets
      }
      for (int r = 0; r < nrow; ++r) {
        const long long i = i0 + r;
        const int x = __builtin_amdgcn_readlane((int) res, r);
        const bool odd = ((i - warm) & 1) == 0;                       // the first row of a chunk is an "odd" row
        uint32_t acc = kFloor2;
        if (x < 4) {
          const uint32_t *e = lds + ((size_t) ((odd ? 0 : 4) + x) * R) * 64 + lane;
          if (odd) {       // register g <- f(register g-1): walk downwards, the lane's first register comes from the lane before
            const uint32_t carry = (uint32_t) dpp_shr1((int) v[R - 1], (int) kFloor2);
#pragma unroll
            for (int j = R - 1; j >= 1; --j) { v[j] = pk_adds_u(v[j - 1], e[j * 64]); acc = pk_max_u(acc, v[j]); }
            v[0] = pk_adds_u(carry, e[0]); acc = pk_max_u(acc, v[0]);
          } else {
#pragma unroll
            for (int j = 0; j < R; ++j) { v[j] = pk_adds_u(v[j], e[j * 64]); acc = pk_max_u(acc, v[j]); }
          }
        } else if (x >= a.Kp) {     // between two targets of a concatenated scan: every diagonal ends here
#pragma unroll
          for (int j = 0; j < R; ++j) v[j] = kFloor2;
        } else {           // degenerate residue: emissions from the full table in global memory [parity][Kp][R][64]
          const uint32_t *e = a.tab_full + ((size_t) ((odd ? 0 : a.Kp) + x) * R) * 64 + lane;
          if (odd) {
            const uint32_t carry = (uint32_t) dpp_shr1((int) v[R - 1], (int) kFloor2);
#pragma unroll
            for (int j = R - 1; j >= 1; --j) { v[j] = pk_adds_u(v[j - 1], e[j * 64]); acc = pk_max_u(acc, v[j]); }
            v[0] = pk_adds_u(carry, e[0]); acc = pk_max_u(acc, v[0]);
          } else {
#pragma unroll
            for (int j = 0; j < R; ++j) { v[j] = pk_adds_u(v[j], e[j * 64]); acc = pk_max_u(acc, v[j]); }
          }
        }
        // any cell of this row at or above the threshold?  (rare: the wavefront leaves the fast path together)
        const int hi = (int) (short) (acc >> 16), lo = (int) (short) (acc & 0xffffu);
        const bool hit = max(hi, lo) >= a.thresh_s;
        if (i >= first && __any(hit)) {
          // the cell upstream would pick: the highest byte score (scores saturate at 255), the first such cell in the
          // order in which p7_SSVFilter_longtarget unstripes the row (vector q outer, byte z inner; k = q + Q z + 1)
          int best = INT_MIN, bestkey = INT_MAX;
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int g = lane * R + j, c0 = odd ? 2 * g - 1 : 2 * g;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int k = c0 + h;
              const int sv = (int) (short) (h ? (v[j] >> 16) : (v[j] & 0xffffu));
              if (k >= 1 && k <= a.M) {
                const int val = min(255, sv + 32768 + a.xB);
                const int key = ((k - 1) % a.Q16) * 16 + (k - 1) / a.Q16;
                if (val > best || (val == best && key < bestkey)) { best = val; bestkey = key; }
              }
            }
          }
          const int smax = wave_max_i32(best);
          const int kmin = -wave_max_i32(best == smax ? -bestkey : INT_MIN);
          if (lane == 0) {
            const int slot = atomicAdd(a.nrec, 1);
            if (slot < a.rec_cap) {
              a.rec_pos[slot] = i; a.rec_strand[slot] = (uint8_t) strand;
              a.rec_k[slot] = (kmin / 16) + a.Q16 * (kmin % 16) + 1; a.rec_sc[slot] = smax;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------- emission pairs in registers
// Every lane of a wavefront works on the same residue, so a row needs the emission pairs of ONE residue: with the pairs
// of A, C, G, T for both parities held in registers (8 R of them) a row is R saturating adds and R maxima and no LDS
// traffic at all (the LDS kernel above issues one ds_read_b32 per register and row, which at 64 lanes x 4 bytes occupies
// the LDS pipe for as long as the two packed operations occupy the SIMD).  The residue selects the register set through
// a wave-uniform branch; rows run in (odd, even) pairs so that the parity is static.  The threshold test is deferred: the
// row maxima of eight rows are folded into one register and compared once; only a block in which some row reached the
// threshold (rare) is run again from its saved first row with the test in every row.  The residues of the next 64 rows
// are fetched while the current ones are processed.  Models up to 3,069 nodes (R <= 24: 192 table registers); longer
// ones keep the LDS kernel.  250 Mbp x 2 strands x M = 1203: 37.7 ms against 44.1 ms (profiles/r03_ssv_kernels.txt lists
// the variants that were measured, among them two that halve the maxima and were no faster: the compiler's handling of
// the wave-uniform branches, not the arithmetic, sets the pace).
template <int R>
__global__ void __launch_bounds__(256) ssvlong_reg_kernel(const SsvLongArgs a)
{
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int nwaves = (int) gridDim.x * 4;
  uint32_t T[2][4][R];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int j = 0; j < R; ++j) T[par][x][j] = a.tab4[((size_t) (par * 4 + x) * R + j) * 64 + lane];

  // one row with the emission pairs <e>; ODD: register g <- f(register g-1), the lane's first register from the lane before
  auto row = [&](uint32_t (&v)[R], const uint32_t (&e)[R], bool odd) -> uint32_t {
    uint32_t acc0 = kFloor2, acc1 = kFloor2;
    if (odd) {
      const uint32_t carry = (uint32_t) dpp_shr1((int) v[R - 1], (int) kFloor2);
#pragma unroll
      for (int j = R - 1; j >= 1; --j) { v[j] = pk_adds_u(v[j - 1], e[j]); if (j & 1) acc1 = pk_max_u(acc1, v[j]); else acc0 = pk_max_u(acc0, v[j]); }
      v[0] = pk_adds_u(carry, e[0]); acc0 = pk_max_u(acc0, v[0]);
    } else {
#pragma unroll
      for (int j = 0; j < R; ++j) { v[j] = pk_adds_u(v[j], e[j]); if (j & 1) acc1 = pk_max_u(acc1, v[j]); else acc0 = pk_max_u(acc0, v[j]); }
    }
    return pk_max_u(acc0, acc1);
  };
  auto row_any = [&](uint32_t (&v)[R], int x, bool odd) -> uint32_t {
    if (x < 4) {
      const int par = odd ? 0 : 1;
      switch (x) {                                  // wave-uniform: one of four register sets
        case 0: return row(v, T[par][0], odd);
        case 1: return row(v, T[par][1], odd);
        case 2: return row(v, T[par][2], odd);
        default: return row(v, T[par][3], odd);
      }
    }
    if (x >= a.Kp) {              // between two targets of a concatenated scan: every diagonal ends here
#pragma unroll
      for (int j = 0; j < R; ++j) v[j] = kFloor2;
      return kFloor2;
    }
    // degenerate residue: emissions from the full table in global memory [parity][Kp][R][64]
    const uint32_t *eg = a.tab_full + ((size_t) ((odd ? 0 : a.Kp) + x) * R) * 64 + lane;
    uint32_t e[R];
#pragma unroll
    for (int j = 0; j < R; ++j) e[j] = eg[j * 64];
    return row(v, e, odd);
  };
  // the report of a row that reached the threshold, exactly as the LDS kernel makes it
  auto report = [&](const uint32_t (&v)[R], bool odd, long long i, int strand) {
    int best = INT_MIN, bestkey = INT_MAX;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int g = lane * R + j, c0 = odd ? 2 * g - 1 : 2 * g;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = c0 + h;
        const int sv = (int) (short) (h ? (v[j] >> 16) : (v[j] & 0xffffu));
        if (k >= 1 && k <= a.M) {
          const int val = min(255, sv + 32768 + a.xB);
          const int key = ((k - 1) % a.Q16) * 16 + (k - 1) / a.Q16;
          if (val > best || (val == best && key < bestkey)) { best = val; bestkey = key; }
        }
      }
    }
    const int smax = wave_max_i32(best);
    const int kmin = -wave_max_i32(best == smax ? -bestkey : INT_MIN);
    if (lane == 0) {
      const int slot = atomicAdd(a.nrec, 1);
      if (slot < a.rec_cap) {
        a.rec_pos[slot] = i; a.rec_strand[slot] = (uint8_t) strand;
        a.rec_k[slot] = (kmin / 16) + a.Q16 * (kmin % 16) + 1; a.rec_sc[slot] = smax;
      }
    }
  };
  auto reached = [&](uint32_t acc) -> bool {
    const int hi = (int) (short) (acc >> 16), lo = (int) (short) (acc & 0xffffu);
    return __any(max(hi, lo) >= a.thresh_s) != 0;
  };

  for (long long chi = wave; chi < a.nchunks; chi += nwaves) {
    const long long ch = a.chunk_list ? a.chunk_list[chi] : chi;
    const int strand = a.strand0 + (int) (ch / a.chunks_per_strand);
    const long long c0 = (ch % a.chunks_per_strand) * (long long) a.chunk_len;
    const long long first = c0 + 1, last = min(a.L, c0 + a.chunk_len);
    const long long warm = max(1LL, first - a.M);
    uint32_t v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = kFloor2;
    // residues of 64 rows, one per lane: strand 1 reads the target backwards and complements (canonical residues by
    // arithmetic: the table lookup would be a second dependent load); fetched one block ahead of its use
    // The fetch is the byte load and nothing else: whatever consumes the byte (the complement) would make the compiler
    // wait for it on the spot, and the prefetch would be a synchronous load.
    auto fetch = [&](long long i0) -> uint32_t {
      const long long pos = i0 + lane;
      if (pos > last) return 0u;
      return a.dsq[strand == 0 ? pos : a.L - pos + 1];
    };
    auto on_strand = [&](uint32_t x) -> uint32_t {
      return strand == 0 ? x : (x < 4 ? 3u - x : (x >= (uint32_t) a.Kp ? x : (uint32_t) a.comp[x]));
    };
    uint32_t raw_next = fetch(warm);
    for (long long i0 = warm; i0 <= last; i0 += 64) {     // i0 - warm is a multiple of 64: row r of a block is odd iff r is even
      const int nrow = (int) min(64LL, last - i0 + 1);
      // The bytes fetched during the previous 64 rows are waited for HERE, before the next fetch is issued: left to the
      // compiler the wait lands at the top of the 8-row loop as vmcnt(0) (the counter retires in order), where it also
      // waits for the fetch just issued -- one exposed memory round trip per 64 rows.
      __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0); expcnt and lgkmcnt unconstrained
      const uint32_t res = on_strand(raw_next);
      raw_next = fetch(i0 + 64);
      for (int r0 = 0; r0 < nrow; r0 += 8) {
        const int nb = min(8, nrow - r0);
        uint32_t saved[R];
#pragma unroll
        for (int j = 0; j < R; ++j) saved[j] = v[j];
        uint32_t blk = kFloor2;
        if (nb == 8) {
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) blk = pk_max_u(blk, row_any(v, __builtin_amdgcn_readlane((int) res, r0 + rr), (rr & 1) == 0));
        } else {
          for (int rr = 0; rr < nb; ++rr) blk = pk_max_u(blk, row_any(v, __builtin_amdgcn_readlane((int) res, r0 + rr), (rr & 1) == 0));
        }
        if (i0 + r0 + nb - 1 < first || !reached(blk)) continue;       // warm-up rows report nothing
        // some row of the block reached the threshold: the block again, row by row (same arithmetic, same values)
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = saved[j];
        for (int rr = 0; rr < nb; ++rr) {
          const bool odd = (rr & 1) == 0;
          const uint32_t acc = row_any(v, __builtin_amdgcn_readlane((int) res, r0 + rr), odd);
          const long long i = i0 + r0 + rr;
          if (i >= first && reached(acc)) report(v, odd, i, strand);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------- host side
static const int kSsvR[] = { 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 24, 32, 48 };      // <= 24: emission pairs in registers

int ssvlong_pick_R(int M)
{
  const int need = (M + 1) / 2 + 1;                 // global registers: cells up to M in both parities
  for (int r : kSsvR) if (r * 64 >= need) return r;
  return -1;
}

// tables: pairs (lo, hi) of signed emission scores s[x][k] = bias - rb[x][k] in byte units, kNegPad outside 1..M.
// odd rows: register g = cells (2g-1, 2g); even rows: (2g, 2g+1); g = lane*R + j, stored [parity][x][j][lane].
// <virtual_node> (register kernel): node M+1 exists with emission 0 for every residue, so that a score that reached the
// last node is still there one row later (ssvlong_reg_kernel tests once per pair of rows); it never counts as a cell.
void ssvlong_build_tables(const Profile &p, int R, bool virtual_node, std::vector<uint32_t> &tab4, std::vector<uint32_t> &tab_full, int *pair_slack)
{
  auto sval = [&](int x, int k) -> int {
    if (virtual_node && k == p.M + 1 && x < p.Kp) return 0;
    if (x >= p.Kp || k < 1 || k > p.M) return kNegPad;
    return (int) p.bias_b - (int) p.rb[(size_t) x * (p.M + 1) + k];
  };
  auto pack = [](int lo, int hi) -> uint32_t { return ((uint32_t) (uint16_t) (int16_t) lo) | ((uint32_t) (uint16_t) (int16_t) hi << 16); };
  tab4.assign((size_t) 2 * 4 * R * 64, 0);
  tab_full.assign((size_t) 2 * p.Kp * R * 64, 0);
  for (int par = 0; par < 2; ++par)
    for (int x = 0; x < p.Kp; ++x)
      for (int j = 0; j < R; ++j)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane * R + j;
          const uint32_t w = par == 0 ? pack(sval(x, 2 * g - 1), sval(x, 2 * g)) : pack(sval(x, 2 * g), sval(x, 2 * g + 1));
          tab_full[(((size_t) par * p.Kp + x) * R + j) * 64 + lane] = w;
          if (x < 4) tab4[(((size_t) par * 4 + x) * R + j) * 64 + lane] = w;
        }
  if (pair_slack) {
    int worst = 0;
    for (int x = 0; x < 4 && x < p.Kp; ++x) for (int k = 1; k <= p.M; ++k) worst = std::max(worst, -sval(x, k));
    *pair_slack = worst;
  }
}

template <int R>
static int launch_ssv(const SsvLongArgs &a, int num_cu, hipStream_t st)
{
  const size_t lds = (size_t) 2 * 4 * R * 64 * 4;
  auto kern = ssvlong_kernel<R>;
  static int per_cu_cached = 0;
  static std::mutex mu;
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (per_cu_cached == 0) {
      if (lds > 64 * 1024) P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_cached, kern, 256, lds));
      if (per_cu_cached < 1) per_cu_cached = 1;
    }
    per_cu = per_cu_cached;
  }
  long long grid = std::min<long long>((a.nchunks + 3) / 4, (long long) num_cu * per_cu);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(256), lds, st, a);
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

template <int R>
static int launch_ssv_reg(const SsvLongArgs &a, int num_cu, hipStream_t st)
{
  auto kern = ssvlong_reg_kernel<R>;
  static int per_cu_cached = 0;
  static std::mutex mu;
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (per_cu_cached == 0) {
      P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_cached, kern, 256, 0));
      if (per_cu_cached < 1) per_cu_cached = 1;
    }
    per_cu = per_cu_cached;
  }
  long long grid = std::min<long long>((a.nchunks + 3) / 4, (long long) num_cu * per_cu);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(256), 0, st, a);
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

int ssvlong_launch(int R, const SsvLongArgs &a, int num_cu, hipStream_t st)
{
  switch (R) {
    case 3: return launch_ssv_reg<3>(a, num_cu, st);
    case 5: return launch_ssv_reg<5>(a, num_cu, st);
    case 7: return launch_ssv_reg<7>(a, num_cu, st);
    case 9: return launch_ssv_reg<9>(a, num_cu, st);
    case 10: return launch_ssv_reg<10>(a, num_cu, st);
    case 11: return launch_ssv_reg<11>(a, num_cu, st);
    case 14: return launch_ssv_reg<14>(a, num_cu, st);
    case 20: return launch_ssv_reg<20>(a, num_cu, st);
    case 2: return a.use_lds ? launch_ssv<2>(a, num_cu, st) : launch_ssv_reg<2>(a, num_cu, st);
    case 4: return a.use_lds ? launch_ssv<4>(a, num_cu, st) : launch_ssv_reg<4>(a, num_cu, st);
    case 6: return a.use_lds ? launch_ssv<6>(a, num_cu, st) : launch_ssv_reg<6>(a, num_cu, st);
    case 8: return a.use_lds ? launch_ssv<8>(a, num_cu, st) : launch_ssv_reg<8>(a, num_cu, st);
    case 12: return a.use_lds ? launch_ssv<12>(a, num_cu, st) : launch_ssv_reg<12>(a, num_cu, st);
    case 16: return a.use_lds ? launch_ssv<16>(a, num_cu, st) : launch_ssv_reg<16>(a, num_cu, st);
    case 24: return a.use_lds ? launch_ssv<24>(a, num_cu, st) : launch_ssv_reg<24>(a, num_cu, st);
    case 32: return launch_ssv<32>(a, num_cu, st);
    case 48: return launch_ssv<48>(a, num_cu, st);
    default: set_error("model too long for the long-target SSV kernel (M > 6141)"); return P7X_EINVAL;
  }
}

} // namespace p7x
