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
// scores of A, C, G, T for both parities sit in LDS as quads of registers; degenerate residues (rare in chromosomes)
// take a slow path through the full table in global memory.  u8 saturation at 255 is not reproduced: it can only keep a
// cell at or above a threshold it has already reached.
#include <map>
#include "p7x_wave.hpp"
#include <mutex>

namespace p7x {

namespace {

typedef short s2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_adds_u(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s2w, a), __builtin_bit_cast(s2w, b))); }
__device__ __forceinline__ uint32_t pk_max_u(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2w, a), __builtin_bit_cast(s2w, b))); }
constexpr uint32_t kFloor2 = 0x80008000u;

// The binary16 flavour (round 6; what p7x_msv.hip's fast kernel does since round 5): a cell holds (v - xB) / 256, a multiple
// of 2^-8 in [0, 1] -- every sum of a cell and an emission is exact --, v_pk_add_f16 clamp is the add with its floor at the
// begin score (the ceiling xB + 256 lies above every byte score: a clamped cell has reached any threshold, as the int16
// flavour's unsaturated one has), and v_pk_maximum3_f16 folds TWO registers into the running maximum: with the maximum
// taken on every second row that is 2.5 packed operations per register and row pair instead of 3.
typedef _Float16 h2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t h_adds_u(uint32_t a, uint32_t b)            // v_pk_add_f16 clamp
{
  const h2w z = { (_Float16) 0.0f, (_Float16) 0.0f }, o = { (_Float16) 1.0f, (_Float16) 1.0f };
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(h2w, a) + __builtin_bit_cast(h2w, b), z), o));
}
__device__ __forceinline__ uint32_t h_max_u(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_bit_cast(h2w, a), __builtin_bit_cast(h2w, b))); }
__device__ __forceinline__ uint32_t h_max3_u(uint32_t a, uint32_t b, uint32_t c)  // v_pk_maximum3_f16
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(h2w, a), __builtin_bit_cast(h2w, b)), __builtin_bit_cast(h2w, c))); }
// a dword of the integer tables (two emissions, int16) as two halves: e / 256, pad entries (kNegPad) -> -2.0
__device__ __forceinline__ uint32_t h_of_i16_pair_u(uint32_t w)
{
  const int lo = max((int) (short) (w & 0xffffu), -512), hi = max((int) (short) (w >> 16), -512);
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz((float) lo * (1.0f / 256.0f), (float) hi * (1.0f / 256.0f)));
}
// the larger cell of a register, in score units above the begin score
template <bool H> __device__ __forceinline__ int rel_max(uint32_t w)
{
  if constexpr (H) { const h2w m = __builtin_bit_cast(h2w, w); return (int) (256.0f * fmaxf((float) m.x, (float) m.y)); }
  else return max((int) (short) (w >> 16), (int) (short) (w & 0xffffu)) + 32768;
}
template <bool H> __device__ __forceinline__ int rel_half(uint32_t w, int h)
{
  if constexpr (H) { const h2w m = __builtin_bit_cast(h2w, w); return (int) (256.0f * (float) (h ? m.y : m.x)); }
  else return (int) (short) (h ? (w >> 16) : (w & 0xffffu)) + 32768;
}

}  // namespace

// ---------------------------------------------------------------------------------------- emission quads from LDS
// The row loop has no branch on the residue.  Its predecessors did: rounds 2-3 kept the emission pairs of A, C, G, T in
// 8 R registers and selected the set through a four-way wave-uniform branch per row -- 15.9 TCUPS = 0.40 of the
// packed-op roof, held there not by arithmetic but by the code around it (5-8 scalar branches and mask moves per row as
// the compiler lowers the switch, 136 VGPRs = three wavefronts per SIMD, a vmcnt(0) at every join that also waits for the
// residues fetched ahead); round 2 read them from LDS one ds_read_b32 per register, which occupies the LDS pipe for as
// long as the two packed operations occupy the SIMD (13.6 TCUPS).  Here the table is [parity][x][q][lane] uint4 (quad
// q = registers 4q .. 4q+3 of the lane), one ds_read_b128 per quad whose address is the lane's slot plus a wave-uniform
// offset x * R4 * 1 KiB.  A ds_read_b128 moves 256 B per LDS clock (a ds_read_b32 128 B), so a register costs the LDS
// one clock per wavefront against eight SIMD clocks of packed arithmetic: with four SIMDs per CU the LDS pipe is half
// used.  The 16-byte slots of consecutive lanes are consecutive, which is conflict-free for the b128 service groups.
// 77-80 VGPRs at R = 10: six wavefronts per SIMD.  Measured, bmyD (M = 1203) x 250 Mbp x 2 strands: 25.6 ms = 23.5 TCUPS
// with the maximum in every row, 21.7 ms = 27.7 TCUPS = 0.70 of the roof with PAIR (round 3: 37.7 ms).
//   * 64 rows at a time.  A block whose residues are all canonical (ballot, once per block) runs as straight-line
//     code, 16 rows per loop trip, the quads of row r+1 in flight while row r is computed (two register sets), the
//     row maxima folded into two accumulators that are compared with the threshold once per 16 rows.  If a group
//     reaches it (rare) -- or the block holds a degenerate residue, a separator, or is the ragged end of a chunk -- the
//     block is run again from its saved first row by the row-by-row loop, which tests every row and reports.
//   * PAIR: the maximum is only taken on every second row (the second, fourth, ... of a block, so that the successor of
//     every skipped row lies in the same block), against a threshold lowered by the most a cell can lose in one row
//     (pair_slack): a cell at or above the threshold on a skipped row is at or above the lowered one a row later, one
//     node further -- for the last node that is the virtual node M + 1 of the tables, whose emission is 0.  The exact
//     test of the repeat decides; the virtual node never counts as a cell there.
//   * chunks are sized by the caller so that their number is a multiple of the resident wavefronts (ssvlong_capacity).
template <bool H> __device__ __forceinline__ bool ssv_reached(uint32_t acc, int thr_rel)        // thr_rel: above the begin score
{
  return __any(rel_max<H>(acc) >= thr_rel) != 0;
}

// one row; ODD: register g <- f(register g-1), the lane's first register from the lane before.  DOMAX: fold into acc0 / acc1
template <int R, bool ODD, bool DOMAX, int RE, bool H>
__device__ __forceinline__ void ssv_row(uint32_t (&v)[R], const uint32_t (&e)[RE], uint32_t &acc0, uint32_t &acc1)
{
  static_assert(RE >= R, "emission registers");
  if constexpr (H) {          // binary16 cells: one maximum3 per two registers, placed behind the second of them
    if constexpr (ODD) {
      const uint32_t carry = (uint32_t) dpp_shr1((int) v[R - 1], 0);
#pragma unroll
      for (int j = R - 1; j >= 0; --j) {
        v[j] = h_adds_u(j >= 1 ? v[j - 1] : carry, e[j]);
        if constexpr (DOMAX) {
          if ((j & 1) == 0 && j + 1 < R) { if (j & 2) acc1 = h_max3_u(acc1, v[j], v[j + 1]); else acc0 = h_max3_u(acc0, v[j], v[j + 1]); }
          else if ((j & 1) == 0) acc0 = h_max_u(acc0, v[j]);      // R odd: the top register has no partner
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        v[j] = h_adds_u(v[j], e[j]);
        if constexpr (DOMAX) {
          if (j & 1) { if (j & 2) acc1 = h_max3_u(acc1, v[j - 1], v[j]); else acc0 = h_max3_u(acc0, v[j - 1], v[j]); }
          else if (j + 1 == R) acc0 = h_max_u(acc0, v[j]);
        }
      }
    }
  } else if constexpr (ODD) {
    const uint32_t carry = (uint32_t) dpp_shr1((int) v[R - 1], (int) kFloor2);
#pragma unroll
    for (int j = R - 1; j >= 1; --j) {
      v[j] = pk_adds_u(v[j - 1], e[j]);
      if constexpr (DOMAX) { if (j & 1) acc1 = pk_max_u(acc1, v[j]); else acc0 = pk_max_u(acc0, v[j]); }
    }
    v[0] = pk_adds_u(carry, e[0]);
    if constexpr (DOMAX) acc0 = pk_max_u(acc0, v[0]);
  } else {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      v[j] = pk_adds_u(v[j], e[j]);
      if constexpr (DOMAX) { if (j & 1) acc1 = pk_max_u(acc1, v[j]); else acc0 = pk_max_u(acc0, v[j]); }
    }
  }
}

template <int R, bool PAIR, bool H>
__global__ void __launch_bounds__(256) ssvlong_quad_kernel(const SsvLongArgs a)
{
  constexpr uint32_t kFloorV = H ? 0u : kFloor2;
  constexpr int R4 = (R + 3) / 4;
  constexpr int RE = 4 * R4;                       // emission registers of a row (the last quad may be partly unused)
  constexpr int XS = R4 * 64;                      // uint4 per (parity, residue)
  extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];
  {
    const uint4 *g = reinterpret_cast<const uint4 *>(a.tab4q);
    for (int i = threadIdx.x; i < 8 * XS; i += 256) {
      uint4 w = g[i];
      if constexpr (H) { w.x = h_of_i16_pair_u(w.x); w.y = h_of_i16_pair_u(w.y); w.z = h_of_i16_pair_u(w.z); w.w = h_of_i16_pair_u(w.w); }
      ldsq[i] = w;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = rfl((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int nwaves = (int) gridDim.x * 4;
  const uint4 *lq = ldsq + lane;
  const int thr_exact = a.thresh_s + 32768;                 // the threshold above the begin score (a.thresh_s: in the int16 flavour's offset)
  const int thr_fast = PAIR ? thr_exact - a.pair_slack : thr_exact;
  const int sc_thresh = a.thresh_s + a.xB + 32768;          // the threshold in byte units

  // the quads of residue x (xs = x * XS, wave-uniform) and parity <par> (0: odd rows)
  auto load_quads = [&](uint32_t (&e)[RE], uint32_t xs, int par) {
    const uint4 *p = lq + xs + par * 4 * XS;
#pragma unroll
    for (int q = 0; q < R4; ++q) { const uint4 t = p[q * 64]; e[4 * q] = t.x; e[4 * q + 1] = t.y; e[4 * q + 2] = t.z; e[4 * q + 3] = t.w; }
  };
  // the report of a row whose best cell reaches the threshold: the cell upstream would pick -- the highest byte score
  // (scores saturate at 255), the first such cell in the order in which p7_SSVFilter_longtarget unstripes the row
  // (vector q outer, byte z inner; k = q + Q z + 1).  Cells outside 1..M (padding, the virtual node) do not count.
  auto report = [&](const uint32_t (&v)[R], bool odd, long long i, int strand) {
    int best = INT_MIN, bestkey = INT_MAX;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int g = lane * R + j, c0 = odd ? 2 * g - 1 : 2 * g;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = c0 + h;
        const int sv = rel_half<H>(v[j], h);
        if (k >= 1 && k <= a.M) {
          const int val = min(255, sv + a.xB);
          const int key = ((k - 1) % a.Q16) * 16 + (k - 1) / a.Q16;
          if (val > best || (val == best && key < bestkey)) { best = val; bestkey = key; }
        }
      }
    }
    const int smax = wave_max_i32(best);
    if (smax < sc_thresh) return;
    const int kmin = -wave_max_i32(best == smax ? -bestkey : INT_MIN);
    if (lane == 0) {
      const int slot = atomicAdd(a.nrec, 1);
      if (slot < a.rec_cap) {
        a.rec_pos[slot] = i; a.rec_strand[slot] = (uint8_t) strand;
        a.rec_k[slot] = (kmin / 16) + a.Q16 * (kmin % 16) + 1; a.rec_sc[slot] = smax;
      }
    }
  };
  // 64 canonical rows; xsv = the lanes' residues times XS.  false: some group of 16 rows may have reached the threshold
  auto fast_block = [&](uint32_t (&v)[R], uint32_t xsv) -> bool {
    uint32_t eA[RE], eB[RE];
    load_quads(eA, (uint32_t) __builtin_amdgcn_readlane((int) xsv, 0), 0);
    for (int r0 = 0; r0 < 64; r0 += 16) {
      uint32_t acc0 = kFloorV, acc1 = kFloorV;
#pragma unroll
      for (int rr = 0; rr < 16; rr += 2) {
        // the scheduler stays inside a row: left alone it gathers the quads of many rows at the top of the loop and
        // pays for them with half the wavefronts per SIMD (136 VGPRs instead of 77 at R = 10)
        load_quads(eB, (uint32_t) __builtin_amdgcn_readlane((int) xsv, r0 + rr + 1), 1);
        ssv_row<R, true, !PAIR, RE, H>(v, eA, acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        load_quads(eA, (uint32_t) __builtin_amdgcn_readlane((int) xsv, (r0 + rr + 2) & 63), 0);      // the last one of a block is not used
        ssv_row<R, false, true, RE, H>(v, eB, acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ssv_reached<H>(H ? h_max_u(acc0, acc1) : pk_max_u(acc0, acc1), thr_fast)) return false;
    }
    return true;
  };

  for (long long chi = wave; chi < a.nchunks; chi += nwaves) {
    const long long ch = a.chunk_list ? a.chunk_list[chi] : chi;
    const int strand = a.strand0 + (int) (ch / a.chunks_per_strand);
    const long long c0 = (ch % a.chunks_per_strand) * (long long) a.chunk_len;
    const long long first = c0 + 1, last = min(a.L, c0 + a.chunk_len);
    const long long warm = max(1LL, first - a.M);
    uint32_t v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = kFloorV;
    // residues of 64 rows, one per lane: strand 1 reads the target backwards and complements (canonical residues by
    // arithmetic: the table lookup would be a second dependent load); fetched one block ahead of its use.  The fetch is
    // the byte load and nothing else: whatever consumes the byte would make the compiler wait for it on the spot.
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
      // the bytes fetched during the previous block are waited for HERE, before the next fetch is issued: left to the
      // compiler the wait would also cover the fetch just issued -- one exposed memory round trip per 64 rows
      __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0); expcnt and lgkmcnt unconstrained
      const uint32_t res = on_strand(raw_next);
      raw_next = fetch(i0 + 64);
      if (nrow == 64 && __all(res < 4u)) {
        uint32_t saved[R];
#pragma unroll
        for (int j = 0; j < R; ++j) saved[j] = v[j];
        if (fast_block(v, res * (uint32_t) XS)) continue;
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = saved[j];
      }
      // row by row: every kind of residue, the exact test in every row
      for (int r = 0; r < nrow; r += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (r + h >= nrow) break;
          const int x = __builtin_amdgcn_readlane((int) res, r + h);
          uint32_t acc0 = kFloorV, acc1 = kFloorV;
          if (x < 4) {
            uint32_t e[RE];
            load_quads(e, (uint32_t) x * XS, h);
            if (h == 0) ssv_row<R, true, true, RE, H>(v, e, acc0, acc1); else ssv_row<R, false, true, RE, H>(v, e, acc0, acc1);
          } else if (x >= a.Kp) {     // between two targets of a concatenated scan: every diagonal ends here
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = kFloorV;
          } else {                    // degenerate residue: emissions from the full table in global memory [parity][Kp][R][64]
            const uint32_t *eg = a.tab_full + ((size_t) ((h == 0 ? 0 : a.Kp) + x) * R) * 64 + lane;
            uint32_t e[R];
#pragma unroll
            for (int j = 0; j < R; ++j) e[j] = H ? h_of_i16_pair_u(eg[j * 64]) : eg[j * 64];
            if (h == 0) ssv_row<R, true, true, R, H>(v, e, acc0, acc1); else ssv_row<R, false, true, R, H>(v, e, acc0, acc1);
          }
          const long long i = i0 + r + h;
          if (i >= first && ssv_reached<H>(H ? h_max_u(acc0, acc1) : pk_max_u(acc0, acc1), thr_exact)) report(v, h == 0, i, strand);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------- host side
static const int kSsvR[] = { 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 24, 32, 48 };      // packed registers per lane

int ssvlong_pick_R(int M, bool virtual_node)
{
  const int top = M + (virtual_node ? 1 : 0);
  const int need = (top + 1) / 2 + 1;               // global registers: cells up to <top> in both parities
  for (int r : kSsvR) if (r * 64 >= need) return r;
  return -1;
}

// tables: pairs (lo, hi) of signed emission scores s[x][k] = bias - rb[x][k] in byte units, kNegPad outside 1..M.
// odd rows: register g = cells (2g-1, 2g); even rows: (2g, 2g+1); g = lane*R + j.
//   tab4q    [parity][x < 4][q][lane] uint4: registers 4q .. 4q+3 of the lane for A, C, G, T (0 beyond R)
//   tab_full [parity][Kp][R][64] uint32: every residue code (degenerate residues, read from global memory)
// <virtual_node> (PAIR): node M+1 exists with emission 0 for every residue, so that a score that reached the last node
// is still there one row later; it never counts as a cell.  pair_slack: the most a cell can lose in one canonical row.
void ssvlong_build_tables(const Profile &p, int R, bool virtual_node, std::vector<uint32_t> &tab4q, std::vector<uint32_t> &tab_full, int *pair_slack)
{
  auto sval = [&](int x, int k) -> int {
    if (virtual_node && k == p.M + 1 && x < p.Kp) return 0;
    if (x >= p.Kp || k < 1 || k > p.M) return kNegPad;
    return (int) p.bias_b - (int) p.rb[(size_t) x * (p.M + 1) + k];
  };
  auto pack = [](int lo, int hi) -> uint32_t { return ((uint32_t) (uint16_t) (int16_t) lo) | ((uint32_t) (uint16_t) (int16_t) hi << 16); };
  const int R4 = (R + 3) / 4;
  tab4q.assign((size_t) 2 * 4 * R4 * 64 * 4, 0);
  tab_full.assign((size_t) 2 * p.Kp * R * 64, 0);
  for (int par = 0; par < 2; ++par)
    for (int x = 0; x < p.Kp; ++x)
      for (int j = 0; j < R; ++j)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane * R + j;
          const uint32_t w = par == 0 ? pack(sval(x, 2 * g - 1), sval(x, 2 * g)) : pack(sval(x, 2 * g), sval(x, 2 * g + 1));
          tab_full[(((size_t) par * p.Kp + x) * R + j) * 64 + lane] = w;
          if (x < 4) tab4q[((((size_t) par * 4 + x) * R4 + j / 4) * 64 + lane) * 4 + j % 4] = w;
        }
  if (pair_slack) {
    int worst = 0;
    for (int x = 0; x < 4 && x < p.Kp; ++x) for (int k = 1; k <= p.M; ++k) worst = std::max(worst, -sval(x, k));
    *pair_slack = worst;
  }
}

// cap_waves != NULL: no launch, only the number of wavefronts the device holds at once
template <int R, bool PAIR, bool H>
static int launch_ssv_quad(const SsvLongArgs &a, int num_cu, hipStream_t st, long long *cap_waves)
{
  const size_t lds = (size_t) 2 * 4 * ((R + 3) / 4) * 64 * 16;
  auto kern = ssvlong_quad_kernel<R, PAIR, H>;
  static std::map<int, int> per_cu_by_device;       // the LDS opt-in is a per-device attribute of the kernel: looked up once per device
  static std::mutex mu;
  int per_cu = 0;
  {
    int dev = 0; P7X_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    int &cached = per_cu_by_device[dev];
    if (cached == 0) {
      if (lds > 64 * 1024) P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&cached, kern, 256, lds));
      if (cached < 1) cached = 1;
    }
    per_cu = cached;
  }
  if (cap_waves) { *cap_waves = (long long) num_cu * per_cu * 4; return P7X_OK; }
  long long grid = std::min<long long>((a.nchunks + 3) / 4, (long long) num_cu * per_cu);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(256), lds, st, a);
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

static int ssvlong_quad_dispatch(int R, bool pair, bool half, const SsvLongArgs &a, int num_cu, hipStream_t st, long long *cap_waves)
{
#define P7X_SQ(RR) case RR: return half ? (pair ? launch_ssv_quad<RR, true, true>(a, num_cu, st, cap_waves) : launch_ssv_quad<RR, false, true>(a, num_cu, st, cap_waves)) \
                                       : (pair ? launch_ssv_quad<RR, true, false>(a, num_cu, st, cap_waves) : launch_ssv_quad<RR, false, false>(a, num_cu, st, cap_waves));
  switch (R) {
    P7X_SQ(2) P7X_SQ(3) P7X_SQ(4) P7X_SQ(5) P7X_SQ(6) P7X_SQ(7) P7X_SQ(8) P7X_SQ(9) P7X_SQ(10) P7X_SQ(11) P7X_SQ(12) P7X_SQ(14) P7X_SQ(16)
    P7X_SQ(20) P7X_SQ(24) P7X_SQ(32) P7X_SQ(48)
    default: set_error("model too long for the long-target SSV kernel (M > 6141)"); return P7X_EINVAL;
  }
#undef P7X_SQ
}

// wavefronts of the quad kernel the device holds at once: the caller cuts the strands into a multiple of that many chunks
int ssvlong_capacity(int R, bool pair, bool half, int num_cu, long long *waves)
{
  SsvLongArgs none{};
  return ssvlong_quad_dispatch(R, pair, half, none, num_cu, nullptr, waves);
}

int ssvlong_launch(int R, bool pair, bool half, const SsvLongArgs &a, int num_cu, hipStream_t st)
{
  return ssvlong_quad_dispatch(R, pair, half, a, num_cu, st, nullptr);
}

} // namespace p7x
