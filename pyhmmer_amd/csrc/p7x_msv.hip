// p7x_msv.hip -- MSV filter on CDNA4 (gfx950): one target sequence per LANE, DP row in VGPRs.
//
// Computes exactly p7_MSVFilter's integer end state xJ (upstream impl_sse/msvfilter.c; reference
// impl_sse/__init__.pxd:12, plan7.pyx:5005) for every target of a packed sequence database.
//
// Why lane-per-sequence and not block-per-comparison: the MSV recurrence
//     M_i[k] = sat_u8( sat_u8( max(M_{i-1}[k-1], xB) + bias ) - rbv[x_i][k] ),   xE = max_k M_i[k]
// needs xE -> xJ -> xB before the next row can start.  Spreading one comparison over a wavefront
// puts a 64-lane max-reduction + broadcast on that serial chain for every residue; keeping the whole
// row inside one lane removes every cross-lane operation and leaves 1.5 VALU ops per DP cell:
//   * values live in packed signed 16-bit pairs (v_pk_max_i16 / v_pk_add_u16): two cells per VGPR;
//   * the diagonal move k-1 -> k is free: odd rows keep register j = cells (2j-1, 2j), even rows
//     cells (2j, 2j+1); an odd row reads the register below it, an even row updates in place, and each
//     alignment has its own emission table in LDS ("odd"/"even" parity tables);
//   * unsigned-byte saturation is reproduced exactly without clamps: floor-at-0 is redundant because
//     the next row takes max(., xB) with xB >= 0 and xE is floored once per row; clip-at-255 can only
//     happen after a row whose xE >= 255 - bias, i.e. after p7_MSVFilter has already returned eslERANGE
//     -- we track the running row maximum and report overflow (xJ = -1) for exactly those targets;
//   * emission lookups are per-lane LDS gathers with ds_read_b64; the table row stride S (dwords)
//     has S/2 odd so the <=32 residue rows fall on distinct bank pairs: conflict-free;
//   * residues arrive as coalesced 16-byte loads from 64-sequence interleaved tiles (p7x_seqdb).
#include <cstdlib>
#include <iterator>
#include <map>
#include <mutex>
#include "p7x_device.hpp"
#include "p7x_kernels.hpp"

namespace p7x {

typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s2 as_s2(uint32_t u) { return __builtin_bit_cast(s2, u); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s2 pk_max(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 pk_adds(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }   // v_pk_add_i16 clamp
__device__ __forceinline__ s2 pk_subs(s2 a, s2 b) { return __builtin_elementwise_sub_sat(a, b); }   // v_pk_sub_i16 clamp
__device__ __forceinline__ s2 splat(int v) { s2 r; r.x = (short) v; r.y = (short) v; return r; }
__device__ __forceinline__ s2 swap_halves(s2 v) { return as_s2(__builtin_amdgcn_alignbit(as_u32(v), as_u32(v), 16)); }

// The same two cells per register as IEEE half floats (the fast kernel's second flavour, see msv_fast_kernel): a cell holds
// (v - xB) / 256, a multiple of 2^-8 in [0, 1] -- every sum of a cell and an emission is exact in binary16 -- and the packed
// add's clamp modifier ([0, 1]) is the floor at the begin score.  What the flavour buys is v_pk_maximum3_f16 (new on gfx950):
// ONE instruction folds two registers into the running maximum, where the integer flavour needs two v_pk_max_i16.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 as_h2(s2 v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ s2 as_s2(h2 v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ s2 h_adds(s2 a, s2 b)            // v_pk_add_f16 clamp
{
  const h2 z = { (_Float16) 0.0f, (_Float16) 0.0f }, o = { (_Float16) 1.0f, (_Float16) 1.0f };
  return as_s2(__builtin_elementwise_min(__builtin_elementwise_max(as_h2(a) + as_h2(b), z), o));
}
__device__ __forceinline__ s2 h_subs(s2 a, s2 b)            // v_pk_add_f16 neg clamp
{
  const h2 z = { (_Float16) 0.0f, (_Float16) 0.0f }, o = { (_Float16) 1.0f, (_Float16) 1.0f };
  return as_s2(__builtin_elementwise_min(__builtin_elementwise_max(as_h2(a) - as_h2(b), z), o));
}
__device__ __forceinline__ s2 h_max(s2 a, s2 b) { return as_s2(__builtin_elementwise_maximum(as_h2(a), as_h2(b))); }
__device__ __forceinline__ s2 h_max3(s2 a, s2 b, s2 c)      // v_pk_maximum3_f16
{ return as_s2(__builtin_elementwise_maximum(__builtin_elementwise_maximum(as_h2(a), as_h2(b)), as_h2(c))); }
__device__ __forceinline__ s2 h_splat(int units)            // <units> / 256 in both halves; exact for |units| <= 2048
{ const _Float16 x = (_Float16) ((float) units * (1.0f / 256.0f)); h2 r; r.x = x; r.y = x; return as_s2(r); }
// a dword of the integer parity tables (two emissions, int16) as two halves: e / 256, pad entries (kNegPad and below) -> -2.0
__device__ __forceinline__ uint32_t h_of_i16_pair(uint32_t w)
{
  const int lo = max((int) (short) (w & 0xffffu), -512), hi = max((int) (short) (w >> 16), -512);
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz((float) lo * (1.0f / 256.0f), (float) hi * (1.0f / 256.0f)));
}


constexpr int msv_stride_c(int R)
{ // smallest S >= R with S/2 odd: the row -> bank-pair map 2*(x*(S/2) mod 32) is then injective for x < 32
  int S = R + (R & 1);
  if (((S / 2) & 1) == 0) S += 2;
  return S;
}
// K > 1 lanes per target (models whose row does not fit one lane's registers): lane h of a target holds registers
// h*R .. h*R + R - 1 of the row and reads them at column h*Rs of the table row.  With Rs = 2 (mod 64) a lane's bank pair is
// (x*K + h + j/2) mod 32: the K lanes of a target never collide, and two targets only when their residues agree mod 32/K.
constexpr int msv_lane_stride_c(int R, int K)
{
  if (K == 1) return msv_stride_c(R);
  int s = R + (R & 1);
  while (s % 64 != 2) s += 2;
  return s;
}
constexpr int msv_row_stride_c(int R, int K) { return K * msv_lane_stride_c(R, K); }      // dwords per table row
// K = 8 (models of 1,022 - 2,048 nodes): two parity tables of such a model do not fit a CU's LDS, so these tiles keep ONE
// table and ONE register alignment -- register j = cells (2j + 1, 2j + 2) in every row -- and move the row down the
// diagonal with a v_alignbit_b32 per register instead (the "uniform" row, RowChunks<..., U = true>): 2.5 operations per
// register where the parity trick spends 1.5, and a 32-bit one among them (issued at twice the packed rate) -- against
// the four packed operations per register, the wavefront reduction per block of rows and the one target per wavefront
// of msv_wavepk_kernel, which took these models until round 6 (2.8 % of a Pfam-shaped library's cells, 18 % of its MSV
// kernel time: profiles/r06_msv_k8.txt).
constexpr bool msv_uniform_c(int K) { return K >= 8; }
constexpr int msv_parities_c(int K) { return msv_uniform_c(K) ? 1 : 2; }
// the even-parity table starts kTabRows rows behind the odd one: as an instruction immediate while that fits 16 bits (K = 1),
// else as part of the lane's address
constexpr int msv_even_imm_c(int R, int K) { return K == 1 ? kTabRows * msv_row_stride_c(R, K) * 4 : 0; }
constexpr int msv_even_add_c(int R, int K) { return (K == 1 || msv_uniform_c(K)) ? 0 : kTabRows * msv_row_stride_c(R, K) * 4; }
constexpr int msv_block_c(int K) { return K == 1 ? 256 : 512; }                            // threads per block: K > 1 tables are big, more
                                                                                           // wavefronts share one copy
// maximum over the K lanes of a target (K = 1, 2, 4, 8; lanes of a target are consecutive)
template <int K, bool H = false> __device__ __forceinline__ s2 target_max(s2 m)
{
  auto mx = [](s2 a, s2 b) { if constexpr (H) return h_max(a, b); else return pk_max(a, b); };
  if constexpr (K >= 2) m = mx(m, as_s2((uint32_t) __builtin_amdgcn_update_dpp(0, (int) as_u32(m), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
  if constexpr (K >= 4) m = mx(m, as_s2((uint32_t) __builtin_amdgcn_update_dpp(0, (int) as_u32(m), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
  if constexpr (K >= 8) m = mx(m, as_s2((uint32_t) __builtin_amdgcn_update_dpp(0, (int) as_u32(m), 0x141, 0xf, 0xf, false)));  // row_half_mirror
  return m;
}
// the register below a lane's first one: the last register of the lane before it (same target), or <edge> in a target's first lane
template <int K> __device__ __forceinline__ s2 carry_in(s2 last, s2 edge, int h)
{
  if constexpr (K == 1) return edge;
  const s2 up = as_s2((uint32_t) __builtin_amdgcn_update_dpp(0, (int) as_u32(last), 0x111, 0xf, 0xf, false));      // row_shr:1
  return h == 0 ? edge : up;
}

// LDS emission fetch.  hipcc fuses adjacent 8-byte LDS loads into ds_read2_b64, which is serviced in
// 16-lane groups at half the rate of ds_read_b64 and breaks the conflict-free bank map above, so the
// loads are issued by hand: four ds_read_b64 per statement, completion counted with lgkmcnt.
struct Chunk { uint2 e0, e1, e2, e3; };

template <int OFF>
__device__ __forceinline__ void lds_issue4(uint32_t addr, Chunk &c)
{
  asm volatile("ds_read_b64 %0, %4 offset:%5\n\t"
               "ds_read_b64 %1, %4 offset:%6\n\t"
               "ds_read_b64 %2, %4 offset:%7\n\t"
               "ds_read_b64 %3, %4 offset:%8"
               : "=&v"(c.e0), "=&v"(c.e1), "=&v"(c.e2), "=&v"(c.e3)   // early-clobber: results may land before the last issue
               : "v"(addr), "i"(OFF), "i"(OFF + 8), "i"(OFF + 16), "i"(OFF + 24));
}
template <int OFF>
__device__ __forceinline__ void lds_issue2(uint32_t addr, Chunk &c)
{
  asm volatile("ds_read_b64 %0, %2 offset:%3\n\t"
               "ds_read_b64 %1, %2 offset:%4"
               : "=&v"(c.e0), "=&v"(c.e1)
               : "v"(addr), "i"(OFF), "i"(OFF + 8));
}
template <int PENDING>
__device__ __forceinline__ void lds_wait4(Chunk &c)
{
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(c.e0), "+v"(c.e1), "+v"(c.e2), "+v"(c.e3) : "i"(PENDING));
}
template <int PENDING>
__device__ __forceinline__ void lds_wait2(Chunk &c)
{
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(c.e0), "+v"(c.e1) : "i"(PENDING));
}

// One DP row over the register file: chunks of four register pairs (8 registers, 16 cells); when R is not a multiple
// of 8 the top chunk holds two pairs.
template <int R, int TB, bool ODD, int T, int FAST = 0, bool U = false>   // TB = byte offset of the row's parity table that goes into the
struct RowChunks {                                        // instructions' immediate; T = chunk index being computed;
                                                          // FAST: 0 = the recurrence as written, 1 = floored int16, 2 = floored half
                                                          // U: the uniform row (one alignment, one table; walked downwards: ODD = true)
  static_assert(!U || ODD, "the uniform row reads the register below: it is walked from the top register down");
  static_assert(R % 4 == 0, "a row is walked in chunks of 8 registers and a last one of 4: the register count must be a multiple of 4");
  static constexpr int NT = (R + 7) / 8;
  static constexpr int TBASE = TB;
  static constexpr int pairs(int t) { return (t == NT - 1 && (R % 8) != 0) ? 2 : 4; }

  template <int TT>
  static __device__ __forceinline__ void issue(uint32_t addr, Chunk &c)
  {
    if constexpr (pairs(TT) == 4) lds_issue4<TBASE + TT * 32>(addr, c); else lds_issue2<TBASE + TT * 32>(addr, c);
  }

  // <carry>: what sits below register 0 (odd rows): the begin score / floor, or the neighbour lane's last register
  template <int JJ>
  static __device__ __forceinline__ void pair(s2 (&v)[R], const uint2 e, const s2 xB, const s2 carry, s2 &accA, s2 &accB)
  {
    if constexpr (U) {          // register j <- (high cell of register j - 1, low cell of register j) of the previous row, + emissions
      s2 pred = carry;
      if constexpr (JJ > 0) pred = v[2 * JJ - 1];
      const s2 d1 = as_s2(__builtin_amdgcn_alignbit(as_u32(v[2 * JJ + 1]), as_u32(v[2 * JJ]), 16));
      const s2 d0 = as_s2(__builtin_amdgcn_alignbit(as_u32(v[2 * JJ]), as_u32(pred), 16));
      if constexpr (FAST == 2) {
        v[2 * JJ + 1] = h_adds(d1, as_s2(e.y));
        v[2 * JJ] = h_adds(d0, as_s2(e.x));
        if constexpr (JJ & 1) { accB = h_max3(accB, v[2 * JJ], v[2 * JJ + 1]); asm volatile("" : "+v"(accB)); }
        else { accA = h_max3(accA, v[2 * JJ], v[2 * JJ + 1]); asm volatile("" : "+v"(accA)); }
      } else if constexpr (FAST == 1) {
        v[2 * JJ + 1] = pk_adds(d1, as_s2(e.y));
        v[2 * JJ] = pk_adds(d0, as_s2(e.x));
        accA = pk_max(accA, v[2 * JJ]);
        accB = pk_max(accB, v[2 * JJ + 1]);
        asm volatile("" : "+v"(accA), "+v"(accB));
      } else {
        v[2 * JJ + 1] = pk_max(d1, xB) + as_s2(e.y);
        accA = pk_max(accA, v[2 * JJ + 1]);
        v[2 * JJ] = pk_max(d0, xB) + as_s2(e.x);
        accB = pk_max(accB, v[2 * JJ]);
      }
    } else if constexpr (FAST == 2) {  // floored halves: as below, and one maximum3 takes both registers of the pair
      if constexpr (ODD) {
        v[2 * JJ + 1] = h_adds(v[2 * JJ], as_s2(e.y));
        s2 pred = carry;
        if constexpr (JJ > 0) pred = v[2 * JJ - 1];
        v[2 * JJ] = h_adds(pred, as_s2(e.x));
      } else {
        v[2 * JJ] = h_adds(v[2 * JJ], as_s2(e.x));
        v[2 * JJ + 1] = h_adds(v[2 * JJ + 1], as_s2(e.y));
      }
      if constexpr (JJ & 1) { accB = h_max3(accB, v[2 * JJ], v[2 * JJ + 1]); asm volatile("" : "+v"(accB)); }
      else { accA = h_max3(accA, v[2 * JJ], v[2 * JJ + 1]); asm volatile("" : "+v"(accA)); }
    } else if constexpr (FAST == 1) {       // floored representation: the saturating add *is* max(., xB) (see msv_fast_kernel)
      if constexpr (ODD) {
        v[2 * JJ + 1] = pk_adds(v[2 * JJ], as_s2(e.y));
        s2 pred = carry;
        if constexpr (JJ > 0) pred = v[2 * JJ - 1];
        v[2 * JJ] = pk_adds(pred, as_s2(e.x));
      } else {
        v[2 * JJ] = pk_adds(v[2 * JJ], as_s2(e.x));
        v[2 * JJ + 1] = pk_adds(v[2 * JJ + 1], as_s2(e.y));
      }
      accA = pk_max(accA, v[2 * JJ]);
      accB = pk_max(accB, v[2 * JJ + 1]);
      // Pin the two accumulator updates here (empty asm = opaque use).  Otherwise LLVM sinks the whole max chain to
      // the end of the row, where back-to-back dependent VOP3P ops each need a wait state on gfx950 (s_nop per op).
      asm volatile("" : "+v"(accA), "+v"(accB));
    } else if constexpr (ODD) {        // register j <- cells (2j-1, 2j), read from register j-1 of the previous (even) row
      const s2 t1 = pk_max(v[2 * JJ], xB);
      v[2 * JJ + 1] = t1 + as_s2(e.y);
      accA = pk_max(accA, v[2 * JJ + 1]);
      s2 t0 = pk_max(carry, xB);
      if constexpr (JJ > 0) t0 = pk_max(v[2 * JJ - 1], xB);
      v[2 * JJ] = t0 + as_s2(e.x);
      accB = pk_max(accB, v[2 * JJ]);
    } else {                    // register j <- cells (2j, 2j+1), in place
      v[2 * JJ] = pk_max(v[2 * JJ], xB) + as_s2(e.x);
      accA = pk_max(accA, v[2 * JJ]);
      v[2 * JJ + 1] = pk_max(v[2 * JJ + 1], xB) + as_s2(e.y);
      accB = pk_max(accB, v[2 * JJ + 1]);
    }
  }

  // cur holds chunk T (already issued); nxt is free.  ODD rows walk chunks downwards, EVEN rows upwards.
  static __device__ __forceinline__ void run(s2 (&v)[R], uint32_t addr, Chunk &cur, Chunk &nxt, const s2 xB, const s2 carry, s2 &accA, s2 &accB)
  {
    constexpr bool last = ODD ? (T == 0) : (T == NT - 1);
    constexpr int TN = ODD ? T - 1 : T + 1;
    constexpr int np = pairs(T);
    if constexpr (!last) {
      issue<TN>(addr, nxt);
      if constexpr (np == 4) lds_wait4<pairs(TN)>(cur); else lds_wait2<pairs(TN)>(cur);
    } else {
      if constexpr (np == 4) lds_wait4<0>(cur); else lds_wait2<0>(cur);
    }
    if constexpr (ODD) {
      if constexpr (np == 4) {
        pair<4 * T + 3>(v, cur.e3, xB, carry, accA, accB);
        pair<4 * T + 2>(v, cur.e2, xB, carry, accA, accB);
      }
      pair<4 * T + 1>(v, cur.e1, xB, carry, accA, accB);
      pair<4 * T + 0>(v, cur.e0, xB, carry, accA, accB);
    } else {
      pair<4 * T + 0>(v, cur.e0, xB, carry, accA, accB);
      pair<4 * T + 1>(v, cur.e1, xB, carry, accA, accB);
      if constexpr (np == 4) {
        pair<4 * T + 2>(v, cur.e2, xB, carry, accA, accB);
        pair<4 * T + 3>(v, cur.e3, xB, carry, accA, accB);
      }
    }
    if constexpr (!last) RowChunks<R, TB, ODD, TN, FAST, U>::run(v, addr, nxt, cur, xB, carry, accA, accB);
  }
};

template <int R, int K, bool ODD>
__device__ __forceinline__ void msv_row(s2 (&v)[R], uint32_t addr, int h, s2 &xB, s2 &xJ, s2 &xEmax,
                                        const s2 basev, const s2 tecv, const s2 tjbmv, const s2 zerov)
{
  constexpr int NT = (R + 7) / 8;
  constexpr bool U = msv_uniform_c(K), DOWN = ODD || U;
  constexpr int T0 = DOWN ? NT - 1 : 0;
  constexpr int TB = DOWN ? 0 : msv_even_imm_c(R, K);
  const s2 carry = DOWN ? carry_in<K>(v[R - 1], xB, h) : xB;
  Chunk ca, cb;
  RowChunks<R, TB, DOWN, T0, 0, U>::template issue<T0>(addr, ca);
  s2 accA = splat(kNegPad), accB = accA;
  RowChunks<R, TB, DOWN, T0, 0, U>::run(v, addr, ca, cb, xB, carry, accA, accB);
  s2 m = pk_max(accA, accB);
  m = pk_max(m, swap_halves(m));
  m = target_max<K>(m);
  xEmax = pk_max(xEmax, m);          // running row maximum: overflow <=> some xE >= 255 - bias
  m = m - tecv;
  xJ = pk_max(xJ, m);
  xB = pk_max(pk_max(basev, xJ) - tjbmv, zerov);
}

// A work item is a (64-target group, part) pair: with K lanes per target a wavefront holds 64 / K targets, part p of a
// group are its targets p * 64 / K ...; lane l of the wavefront is lane h = l % K of target l / K of the part.
template <int K> struct MsvItem {
  int g, t, h;
  __device__ __forceinline__ MsvItem(int group, int part, int lane) : g(group), t(part * (64 / K) + lane / K), h(lane % K) {}
};

template <int R, int K>
__global__ void __launch_bounds__(msv_block_c(K)) msv_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  constexpr int S = msv_row_stride_c(R, K), Rs = msv_lane_stride_c(R, K), BLK = msv_block_c(K);
  const MsvArgs a = load_args<MsvArgs>(ref);
  const int nitems = (a.group_list ? *a.group_count : a.ngroups - a.group_first) * K;
  if (nitems <= 0 || *a.counter >= nitems) return;          // nothing (left) for this lane: skip the table load
  {
    constexpr int n4 = (msv_parities_c(K) * kTabRows * S) / 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.tab);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (int i = threadIdx.x; i < n4; i += BLK) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const s2 basev = splat(a.base), tecv = splat(a.tec), zerov = splat(0);

  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(a.counter, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= nitems) break;
    const int gi = item / K;
    const MsvItem<K> it(a.group_list ? __builtin_amdgcn_readfirstlane(a.group_list[gi]) : gi + a.group_first, item % K, lane);

    const int slot = it.g * 64 + it.t;
    const int L = a.slot_len[slot];
    const int nblk = a.grp_nblk[it.g];
    const uint4 *tp = a.tiles + a.grp_off[it.g] + it.t;
    const s2 tjbmv = splat((int) a.tjb_tab[L] + a.tbm);
    const uint32_t lane_col = (uint32_t) it.h * (uint32_t) (Rs * 4);

    s2 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = zerov;
    s2 xJ = zerov, xEmax = zerov;
    s2 xB = pk_max(basev - tjbmv, zerov);

    uint4 cur = tp[0];
    for (int b = 0; b < nblk; ++b) {
      const uint4 nxt = (b + 1 < nblk) ? tp[(size_t) (b + 1) * 64] : cur;   // prefetch the next 16 residues
      uint32_t w0 = cur.x, w1 = cur.y, w2 = cur.z, w3 = cur.w;
#pragma unroll 1
      for (int c = 0; c < 8; ++c) {
        const uint32_t x0 = w0 & 0xffu, x1 = (w0 >> 8) & 0xffu;
        w0 = __builtin_amdgcn_alignbit(w1, w0, 16); w1 = __builtin_amdgcn_alignbit(w2, w1, 16);
        w2 = __builtin_amdgcn_alignbit(w3, w2, 16); w3 >>= 16;
        msv_row<R, K, true >(v, x0 * (uint32_t) (S * 4) + lane_col, it.h, xB, xJ, xEmax, basev, tecv, tjbmv, zerov);
        msv_row<R, K, false>(v, x1 * (uint32_t) (S * 4) + lane_col + (uint32_t) msv_even_add_c(R, K), it.h, xB, xJ, xEmax, basev, tecv, tjbmv, zerov);
      }
      cur = nxt;
    }
    if (L > 0 && it.h == 0) {
      const int xe = xEmax.x;
      a.out_xJ[slot] = (xe >= 255 - a.bias) ? (int16_t) -1 : (int16_t) xJ.x;
    }
  }
}

// ---------------------------------------------------------------------------- the fast variant: 1.0 op per cell
// Cells are stored relative to the current begin score:  s = v - xB - 32768.  Then the signed saturating add
// floors at -32768 == xB, i.e. one v_pk_add_i16 clamp computes  max(M[k-1], xB) + e  *and* the next row's floor;
// with the row-maximum update that is 2 packed ops per two cells.  Differences to the exact kernel:
//   * xE is seen as max(xE, xB).  This never changes xB (J only matters above base) and changes the final xJ
//     only if *every* row maximum stayed below its begin score, in which case xJ comes out as the constant
//     F0 = base - tjb - tbm - tec.  Exactly those targets (xJ == F0) are ambiguous: their 64-target group is
//     appended to a list and recomputed by the exact kernel above.  All other xJ are exact.
//   * the begin score moves when xJ rises above base: with xB = max(max(base, xJ) - tjbm, 0) as the invariant, a
//     row moves it iff its (relative) maximum m exceeds the per-target constant T = tjbm + tec - 32768.  The
//     maximum is therefore accumulated over an EPOCH of rows (the accumulators are not reset per row), a row costs
//     three ops beyond its cells (merge, max with T, compare), and xJ / the overflow watermark are folded from the
//     epoch maximum when an epoch ends: at the (rare, wave-uniform branch) row that moves some lane's begin score
//     -- every register is then re-biased by the increment (v_pk_sub_i16 clamp) -- and after the last row.
// K > 1: the row of a long model is spread over K consecutive lanes (blocks of R registers).  Per row that adds the
// neighbour's last register as the carry into an odd row (one DPP move) and the maximum over the K lanes (log2 K DPP
// steps); everything else, including the begin-score bookkeeping, runs identically in the K lanes.
// H: the half-float flavour (default).  Cells hold (v - xB) / 256 as binary16, the emission tables are converted while they
// are staged into LDS, v_pk_add_f16 clamp is the add with its floor (0 == xB; the ceiling 1.0 == xB + 256 lies above the
// overflow watermark 255 - bias, so a clamped cell belongs to a target that is reported as overflow either way), and
// v_pk_maximum3_f16 takes two registers into the epoch's maximum at once: 1.5 packed ops per cell pair instead of 2.
// Every value is a multiple of 2^-8 below 4: binary16 arithmetic on them is exact, the integers that leave the kernel
// are the same as the integer flavour's (option msv_f16 = 0 selects that one; tests compare both with the oracle).
template <int R, int K, bool H>
__device__ __forceinline__ void msv_fast_body(const MsvArgs &a, uint32_t *lds)
{
  constexpr int S = msv_row_stride_c(R, K), Rs = msv_lane_stride_c(R, K), BLK = msv_block_c(K);
  constexpr int MODE = H ? 2 : 1;
  const int nitems = (a.ngroups - a.group_first) * K;
  if (*a.counter >= nitems) return;            // this lane's groups are all taken: skip the table load
  {
    constexpr int n4 = (msv_parities_c(K) * kTabRows * S) / 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.tab);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (int i = threadIdx.x; i < n4; i += BLK) {
      uint4 w = src[i];
      if constexpr (H) { w.x = h_of_i16_pair(w.x); w.y = h_of_i16_pair(w.y); w.z = h_of_i16_pair(w.z); w.w = h_of_i16_pair(w.w); }
      dst[i] = w;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const s2 floorv = H ? as_s2(0u) : splat(-32768);
  constexpr int NT = (R + 7) / 8;
  auto mx = [](s2 x, s2 y) { if constexpr (H) return h_max(x, y); else return pk_max(x, y); };

  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(a.counter, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= nitems) break;
    const MsvItem<K> it(item / K + a.group_first, item % K, lane);

    const int slot = it.g * 64 + it.t;
    const int L = a.slot_len[slot];
    const int nblk = a.grp_nblk[it.g];
    const uint2 *tp = reinterpret_cast<const uint2 *>(a.tiles + a.grp_off[it.g] + it.t);   // this target's 16 residues of a tile: tp[0], tp[1]
    const int tjbm = (int) a.tjb_tab[L] + a.tbm;
    const uint32_t lane_col = (uint32_t) it.h * (uint32_t) (Rs * 4);

    s2 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = floorv;
    int xJ = 0, xEmax = 0;
    int xB = max(a.base - tjbm, 0);
    const int F0 = xB - a.tec;
    const s2 Tv = H ? h_splat(tjbm + a.tec) : splat(tjbm + a.tec - 32768);
    s2 accA = floorv, accB = floorv;             // the epoch's running maximum
    // fold the epoch maximum into xJ / the overflow watermark; returns the begin score that follows
    auto fold = [&](const s2 m2) -> int {
      int rel;
      if constexpr (H) { const h2 m = as_h2(m2); rel = (int) (256.0f * fmaxf((float) m.x, (float) m.y)); }
      else rel = max((int) m2.x, (int) m2.y) + 32768;
      const int xE = rel + xB;
      xEmax = max(xEmax, xE);
      xJ = max(xJ, xE - a.tec);
      return max(max(a.base, xJ) - tjbm, 0);
    };

    uint2 cur = tp[0];
    for (int hb = 0; hb < 2 * nblk; ++hb) {          // 8 residues at a time
      const uint2 nxt = (hb + 1 < 2 * nblk) ? tp[(size_t) ((hb + 1) >> 1) * 128 + ((hb + 1) & 1)] : cur;
      uint32_t w0 = cur.x, w1 = cur.y;
#pragma unroll 1
      for (int c = 0; c < 4; ++c) {
        const uint32_t x0 = w0 & 0xffu, x1 = (w0 >> 8) & 0xffu;
        w0 = __builtin_amdgcn_alignbit(w1, w0, 16); w1 >>= 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          Chunk ca, cb;
          if constexpr (msv_uniform_c(K)) {      // one kind of row
            const uint32_t addr = (half == 0 ? x0 : x1) * (uint32_t) (S * 4) + lane_col;
            const s2 carry = carry_in<K>(v[R - 1], floorv, it.h);
            RowChunks<R, 0, true, NT - 1, MODE, true>::template issue<NT - 1>(addr, ca);
            RowChunks<R, 0, true, NT - 1, MODE, true>::run(v, addr, ca, cb, floorv, carry, accA, accB);
          } else if (half == 0) {        // odd row
            const uint32_t addr = x0 * (uint32_t) (S * 4) + lane_col;
            const s2 carry = carry_in<K>(v[R - 1], floorv, it.h);
            RowChunks<R, 0, true, NT - 1, MODE>::template issue<NT - 1>(addr, ca);
            RowChunks<R, 0, true, NT - 1, MODE>::run(v, addr, ca, cb, floorv, carry, accA, accB);
          } else {                // even row
            const uint32_t addr = x1 * (uint32_t) (S * 4) + lane_col + (uint32_t) msv_even_add_c(R, K);
            RowChunks<R, msv_even_imm_c(R, K), false, 0, MODE>::template issue<0>(addr, ca);
            RowChunks<R, msv_even_imm_c(R, K), false, 0, MODE>::run(v, addr, ca, cb, floorv, floorv, accA, accB);
          }
          const s2 m2 = target_max<K, H>(mx(accA, accB));
          if (__builtin_expect(__any(as_u32(mx(m2, Tv)) != as_u32(Tv)), 0)) {
            asm volatile("; re-bias: the begin score moved" ::: "memory");   // keeps this a real (rare) branch: no if-conversion
            const int xBn = fold(m2);
            const s2 dv = H ? h_splat(xBn - xB) : splat(xBn - xB);        // zero in the lanes whose own maximum stayed at or below T
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = H ? h_subs(v[j], dv) : pk_subs(v[j], dv);
            xB = xBn;
            accA = floorv; accB = floorv;
          }
        }
      }
      cur = nxt;
    }
    fold(target_max<K, H>(mx(accA, accB)));
    const bool ambiguous = (L > 0) && (xJ == F0) && !(xEmax >= 255 - a.bias);
    if (L > 0 && it.h == 0) a.out_xJ[slot] = (xEmax >= 255 - a.bias) ? (int16_t) -1 : (int16_t) xJ;
    if (__any(ambiguous) && lane == 0) { const int idx = atomicAdd(a.amb_count, 1); a.amb_groups[idx] = it.g; }
  }
}

constexpr int msv_min_blocks_c(int R, int K) { return K > 1 ? 2 : (R <= 92 ? 4 : (R <= 136 ? 3 : 2)); }

template <int R, int K, bool H>
__global__ void __launch_bounds__(msv_block_c(K), msv_min_blocks_c(R, K)) msv_fast_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const MsvArgs a = load_args<MsvArgs>(ref);
  msv_fast_body<R, K, H>(a, lds);
}

// One launch for the lanes of SEVERAL register tiles (half-float flavour): a workgroup serves one lane (blockIdx.y) and
// branches, uniformly, to that lane's instantiation of the row loop.  A batch of the scan orientation holds up to 28 tiles
// with 1-900 profiles each against a block of a few dozen 64-target groups: one launch per tile is a few hundred to a
// few thousand wavefronts that last as long as the block's longest group, eight of them in flight on the hardware
// queues beside the other stages' kernels -- the device ran the MSV stage at a quarter of its rate.  A tier -- the tiles
// that share a block size and an occupancy -- is one launch that fills the device.  The dynamic LDS of the launch is that
// of its largest tile; the occupancy that the tier promises (4 / 3 / 2 blocks per CU) holds for it.
//   tier 0: K = 1, R <= 92;  1: K = 1, R <= 136;  2: K = 1, R <= 224;  3: K = 2;  4: K = 4;  5: K = 8 (uniform rows)
// The register tiles, tier by tier: ONE list for the tier kernels' dispatch, the per-tile launches and msv_pick (a tile that
// is in one of them and not in another would leave scores unwritten without an error: ADVICE r05).
#define P7X_TILES_TIER0(X) X(8, 1) X(12, 1) X(16, 1) X(20, 1) X(24, 1) X(28, 1) X(32, 1) X(36, 1) X(40, 1) X(44, 1) X(48, 1) X(52, 1) \
                           X(56, 1) X(60, 1) X(64, 1) X(68, 1) X(72, 1) X(76, 1) X(80, 1) X(84, 1) X(88, 1) X(92, 1)
#define P7X_TILES_TIER1(X) X(96, 1) X(100, 1) X(104, 1) X(108, 1) X(112, 1) X(116, 1) X(120, 1) X(124, 1) X(128, 1) X(132, 1) X(136, 1)
#define P7X_TILES_TIER2(X) X(140, 1) X(144, 1) X(148, 1) X(152, 1) X(156, 1) X(160, 1) X(176, 1) X(192, 1) X(208, 1) X(224, 1)
#define P7X_TILES_TIER3(X) X(120, 2) X(128, 2) X(136, 2) X(144, 2) X(152, 2) X(160, 2) X(168, 2) X(176, 2) X(184, 2) X(192, 2) X(200, 2) \
                           X(208, 2) X(216, 2) X(224, 2)
#define P7X_TILES_TIER4(X) X(120, 4) X(128, 4)
#define P7X_TILES_TIER5(X) X(64, 8) X(72, 8) X(80, 8) X(88, 8) X(96, 8) X(104, 8) X(112, 8) X(120, 8) X(128, 8)
constexpr int msv_tier_block_c(int T) { return T <= 2 ? 256 : 512; }
constexpr int msv_tier_min_blocks_c(int T) { return T == 0 ? 4 : (T == 1 ? 3 : 2); }      // (waves per SIMD the register budget must allow)

template <int TIER>
__global__ void __launch_bounds__(msv_tier_block_c(TIER), msv_tier_min_blocks_c(TIER)) msv_tier_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const MsvArgs a = load_args<MsvArgs>(ref);
  const int R = __builtin_amdgcn_readfirstlane(a.R);
#define P7X_TILE(r, k) case r: msv_fast_body<r, k, true>(a, lds); break;
  if constexpr (TIER == 0) { switch (R) { P7X_TILES_TIER0(P7X_TILE) default: break; } }
  else if constexpr (TIER == 1) { switch (R) { P7X_TILES_TIER1(P7X_TILE) default: break; } }
  else if constexpr (TIER == 2) { switch (R) { P7X_TILES_TIER2(P7X_TILE) default: break; } }
  else if constexpr (TIER == 3) { switch (R) { P7X_TILES_TIER3(P7X_TILE) default: break; } }
  else if constexpr (TIER == 4) { switch (R) { P7X_TILES_TIER4(P7X_TILE) default: break; } }
  else { switch (R) { P7X_TILES_TIER5(P7X_TILE) default: break; } }        // (launch_tier refuses a tile that is not in its tier's list)
#undef P7X_TILE
}

// ---------------------------------------------------------------------------- host side

// One lane per target up to 224 row registers (M <= 445); two lanes (M <= 893) and four lanes (M <= 1021) beyond, while
// the parity tables of the K lanes (256 * K * Rs bytes) fit the CU's LDS.  Longer models: wave-per-target kernel.
#define P7X_R_ONLY(r, k) r,
static const int kRList[] = { P7X_TILES_TIER0(P7X_R_ONLY) P7X_TILES_TIER1(P7X_R_ONLY) P7X_TILES_TIER2(P7X_R_ONLY) };
static const int kRList2[] = { P7X_TILES_TIER3(P7X_R_ONLY) };
static const int kRList4[] = { P7X_TILES_TIER4(P7X_R_ONLY) };
static const int kRList8[] = { P7X_TILES_TIER5(P7X_R_ONLY) };      // uniform rows: register j = cells (2j + 1, 2j + 2)
#undef P7X_R_ONLY
static bool msv_tile_exists(int R, int K)
{
  auto in = [R](const int *b, const int *e) { for (; b != e; ++b) if (*b == R) return true; return false; };
  switch (K) {
    case 1: return in(std::begin(kRList), std::end(kRList));
    case 2: return in(std::begin(kRList2), std::end(kRList2));
    case 4: return in(std::begin(kRList4), std::end(kRList4));
    case 8: return in(std::begin(kRList8), std::end(kRList8));
    default: return false;
  }
}

int msv_pick(int M, int *K)
{
  const int need = (M + 1) / 2 + 1;   // odd rows hold cells (2j-1, 2j): node M needs register (M+1)/2
  *K = 1;
  for (int r : kRList) if (r >= need) return r;
  *K = 2;
  for (int r : kRList2) if (2 * r >= need) return r;
  *K = 4;
  for (int r : kRList4) if (4 * r >= need) return r;
  *K = 8;
  if (debug_opt(OPT_MSV_K8) != 0)           // (0: A/B and test seam -- these models one target per wavefront, as until round 6)
    for (int r : kRList8) if (8 * r >= (M + 1) / 2) return r;
  *K = 0;
  return -1;
}

static int row_stride(int R, int K)
{
  int rs = R + (R & 1);
  if (K == 1) { if (((rs / 2) & 1) == 0) rs += 2; return rs; }
  while (rs % 64 != 2) rs += 2;
  return K * rs;
}
int msv_stride(int R, int K) { return row_stride(R, K); }
int msv_table_dwords(int R, int K) { return msv_parities_c(K) * kTabRows * row_stride(R, K); }

// Build the two parity tables from un-striped biased costs rb[x][k] (k = 1..M):
//   signed emission s[x][k] = bias - rb[x][k]  (the value sbv holds, impl_sse/p7_oprofile.pxd: sbv)
//   odd  table, register j: (s[2j-1], s[2j]);  even table, register j: (s[2j], s[2j+1])
// nodes < 1 or > M, and the pad residue row, get kNegPad.  K lanes per target: register j = h * R + r of the row is
// register r of lane h and sits at column h * (S / K) + r of the table row.
void msv_build_tables(const Profile &p, int R, int K, std::vector<uint32_t> &out)
{
  const int S = row_stride(R, K), Rs = S / K;
  out.assign((size_t) msv_table_dwords(R, K), 0);
  auto sval = [&](int x, int k) -> int {
    if (x >= p.Kp || k < 1 || k > p.M) return kNegPad;
    return (int) p.bias_b - (int) p.rb[(size_t) x * (p.M + 1) + k];
  };
  auto pack = [](int lo, int hi) -> uint32_t { return ((uint32_t) (uint16_t) (int16_t) lo) | ((uint32_t) (uint16_t) (int16_t) hi << 16); };
  if (msv_uniform_c(K)) {         // one table, register j: (s[2j+1], s[2j+2])
    for (int x = 0; x < kTabRows; ++x)
      for (int h = 0; h < K; ++h)
        for (int r = 0; r < Rs; ++r) {
          const int j = h * R + r;
          out[(size_t) x * S + (size_t) h * Rs + r] = r < R ? pack(sval(x, 2 * j + 1), sval(x, 2 * j + 2)) : pack(kNegPad, kNegPad);
        }
    return;
  }
  for (int x = 0; x < kTabRows; ++x)
    for (int h = 0; h < K; ++h)
      for (int r = 0; r < Rs; ++r) {
        const bool live = r < R;
        const int j = h * R + r;
        out[((size_t) 0 * kTabRows + x) * S + (size_t) h * Rs + r] = live ? pack(sval(x, 2 * j - 1), sval(x, 2 * j)) : pack(kNegPad, kNegPad);
        out[((size_t) 1 * kTabRows + x) * S + (size_t) h * Rs + r] = live ? pack(sval(x, 2 * j), sval(x, 2 * j + 1)) : pack(kNegPad, kNegPad);
      }
}

template <int R, int K>
static int launch_RK(const ArgRun<MsvArgs> &main, const ArgRun<MsvArgs> *amb, int num_cu, hipStream_t st, bool amb_only = false)
{
  constexpr int BLK = msv_block_c(K);
  const size_t lds_bytes = (size_t) msv_parities_c(K) * kTabRows * msv_row_stride_c(R, K) * 4;
  // occupancy and the LDS opt-in are per kernel instantiation: looked up once
  struct Info { int per_cu_exact = 0, per_cu_fast = 0, per_cu_half = 0; bool ok = false; };
  static std::map<int, Info> info_by_device;       // ... and per device (the opt-in is an attribute of the kernel ON a device)
  static std::mutex mu;
  Info info;
  {
    int dev = 0; P7X_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    Info &slot = info_by_device[dev];
    if (!slot.ok) {
      Info &info = slot;
      if (lds_bytes > 64 * 1024) {
        P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&msv_kernel<R, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
        P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&msv_fast_kernel<R, K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
        P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&msv_fast_kernel<R, K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
      }
      P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&info.per_cu_exact, msv_kernel<R, K>, BLK, lds_bytes));
      P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&info.per_cu_fast, msv_fast_kernel<R, K, false>, BLK, lds_bytes));
      P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&info.per_cu_half, msv_fast_kernel<R, K, true>, BLK, lds_bytes));
      if (info.per_cu_exact < 1) info.per_cu_exact = 1;
      if (info.per_cu_fast < 1) info.per_cu_fast = 1;
      if (info.per_cu_half < 1) info.per_cu_half = 1;
      info.ok = true;
    }
    info = slot;
  }
  if (amb_only) {                        // the fast kernel ran as part of a tier launch: only the lists of ambiguous groups are left
    hipLaunchKernelGGL((msv_kernel<R, K>), dim3(lane_grid(32, 32, amb->n), (unsigned) amb->n), dim3(BLK), lds_bytes, st, amb->ref());
    P7X_HIP(hipGetLastError());
    return P7X_OK;
  }
  constexpr int wpb = BLK / 64;          // wavefronts (work items in flight) per block
  long want = 1;
  for (int i = 0; i < main.n; ++i) want = std::max<long>(want, ((long) (main.at(i).ngroups - main.at(i).group_first) * K + wpb - 1) / wpb);
  if (amb == nullptr) {           // exact kernel over every group
    const unsigned gx = lane_grid_pull(want, (long) num_cu * info.per_cu_exact, main.n, debug_opt(OPT_MSV_LANE_BLOCKS) == 0);
    hipLaunchKernelGGL((msv_kernel<R, K>), dim3(gx, (unsigned) main.n), dim3(BLK), lds_bytes, st, main.ref());
    P7X_HIP(hipGetLastError());
    return P7X_OK;
  }
  // fast kernel over every group, then the exact kernel over the (normally empty) lists of ambiguous groups
  const bool half = debug_opt(OPT_MSV_F16) != 0;          // the half-float flavour unless a test asks for the integer one
  int per_cu2 = half ? info.per_cu_half : info.per_cu_fast;
  // A/B switch: cap the resident blocks per CU so that other kernels' wavefronts fit beside the MSV row registers
  const int cap = debug_opt(OPT_MSV_BLOCKS_PER_CU);
  if (cap > 0 && per_cu2 > cap) per_cu2 = cap;
  const unsigned gx2 = lane_grid_pull(want, (long) num_cu * per_cu2, main.n, debug_opt(OPT_MSV_LANE_BLOCKS) == 0);
  if (half) hipLaunchKernelGGL((msv_fast_kernel<R, K, true>), dim3(gx2, (unsigned) main.n), dim3(BLK), lds_bytes, st, main.ref());
  else hipLaunchKernelGGL((msv_fast_kernel<R, K, false>), dim3(gx2, (unsigned) main.n), dim3(BLK), lds_bytes, st, main.ref());
  P7X_HIP(hipGetLastError());
  hipLaunchKernelGGL((msv_kernel<R, K>), dim3(lane_grid(32, 32, amb->n), (unsigned) amb->n), dim3(BLK), lds_bytes, st, amb->ref());
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

static int msv_launch_impl(int R, int K, const ArgRun<MsvArgs> &main, const ArgRun<MsvArgs> *amb, int num_cu, hipStream_t st, bool amb_only);

int msv_launch(int R, int K, const ArgRun<MsvArgs> &main, const ArgRun<MsvArgs> *amb, int num_cu, hipStream_t st)
{ return msv_launch_impl(R, K, main, amb, num_cu, st, false); }

int msv_exact_launch(int R, int K, const ArgRun<MsvArgs> &amb, int num_cu, hipStream_t st)
{ return msv_launch_impl(R, K, amb, &amb, num_cu, st, true); }

int msv_tier(int R, int K)
{
  if (K == 1) return R <= 92 ? 0 : (R <= 136 ? 1 : 2);
  return K == 2 ? 3 : (K == 4 ? 4 : 5);
}

template <int TIER>
static int launch_tier(const ArgRun<MsvArgs> &main, int num_cu, hipStream_t st)
{
  constexpr int BLK = msv_tier_block_c(TIER), K = TIER <= 2 ? 1 : (TIER == 3 ? 2 : (TIER == 4 ? 4 : 8));
  int Rmax = 0;
  long want = 1;
  constexpr int wpb = BLK / 64;
  for (int i = 0; i < main.n; ++i) {
    const MsvArgs &a = main.at(i);
    if (msv_tier(a.R, K) != TIER || !msv_tile_exists(a.R, K)) { set_error("msv_tier_launch: a lane of another tier, or a register tile without a kernel"); return P7X_EINVAL; }
    Rmax = std::max(Rmax, a.R);
    want = std::max<long>(want, ((long) (a.ngroups - a.group_first) * K + wpb - 1) / wpb);
  }
  const size_t lds_bytes = (size_t) msv_table_dwords(Rmax, K) * 4;
  // occupancy by (device, LDS bytes); the LDS opt-in once per device, for the tier's largest tile
  static std::map<std::pair<int, size_t>, int> per_cu_of;
  static std::map<int, bool> opted;
  static std::mutex mu;
  int per_cu = 0;
  {
    int dev = 0; P7X_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (!opted[dev]) {
      constexpr int Rtop = TIER == 0 ? 92 : (TIER == 1 ? 136 : (TIER >= 4 ? 128 : 224));
      const size_t most = (size_t) msv_table_dwords(Rtop, K) * 4;
      if (most > 64 * 1024)
        P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&msv_tier_kernel<TIER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) most));
      opted[dev] = true;
    }
    auto it = per_cu_of.find({ dev, lds_bytes });
    if (it == per_cu_of.end()) {
      int v = 0;
      P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, msv_tier_kernel<TIER>, BLK, lds_bytes));
      it = per_cu_of.emplace(std::make_pair(dev, lds_bytes), std::max(v, 1)).first;
    }
    per_cu = it->second;
  }
  const unsigned gx = lane_grid_pull(want, (long) num_cu * per_cu, main.n, debug_opt(OPT_MSV_LANE_BLOCKS) == 0);
  hipLaunchKernelGGL((msv_tier_kernel<TIER>), dim3(gx, (unsigned) main.n), dim3(BLK), lds_bytes, st, main.ref());
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

int msv_tier_launch(int tier, const ArgRun<MsvArgs> &main, int num_cu, hipStream_t st)
{
  if (main.n <= 0) return P7X_OK;
  switch (tier) {
    case 0: return launch_tier<0>(main, num_cu, st);
    case 1: return launch_tier<1>(main, num_cu, st);
    case 2: return launch_tier<2>(main, num_cu, st);
    case 3: return launch_tier<3>(main, num_cu, st);
    case 4: return launch_tier<4>(main, num_cu, st);
    case 5: return launch_tier<5>(main, num_cu, st);
    default: set_error("msv_tier_launch: no such tier"); return P7X_EINVAL;
  }
}

static int msv_launch_impl(int R, int K, const ArgRun<MsvArgs> &main, const ArgRun<MsvArgs> *amb, int num_cu, hipStream_t st, bool amb_only)
{
  if (main.n <= 0) return P7X_OK;
  switch (K * 1000 + R) {
#define P7X_CASE(r, k) case k * 1000 + r: return launch_RK<r, k>(main, amb, num_cu, st, amb_only);
    P7X_TILES_TIER0(P7X_CASE) P7X_TILES_TIER1(P7X_CASE) P7X_TILES_TIER2(P7X_CASE) P7X_TILES_TIER3(P7X_CASE) P7X_TILES_TIER4(P7X_CASE) P7X_TILES_TIER5(P7X_CASE)
#undef P7X_CASE
    default: set_error("msv_launch: unsupported register tile"); return P7X_EINVAL;
  }
}

} // namespace p7x
