// p7x_vitfwd.hip -- Viterbi filter, Forward parser and Backward parser on CDNA4: one target per WAVEFRONT.
//
// These stages only see the survivors of the MSV filter (~2% / ~0.1% of the targets), too few to fill the
// chip with one sequence per lane, so a wavefront shares one (profile, sequence) comparison: lane z owns
// the C consecutive nodes k = z*C+1 .. z*C+C (this is Farrar striping with 64 stripes: the k-1 neighbour is
// the previous register, except at chunk boundaries where one DPP wave_shr moves it across lanes).
//   * transitions: one ds_read_b128 per node per row (8 x int16) or two (8 x f32), stored [c*64 + lane];
//   * emissions:   [residue][c*64 + lane];
//   * xE (row maximum / row sum) and the special states: DPP row_shr/row_bcast reduction -> SGPR scalars;
//   * D->D: serial inside the lane; across lanes the Viterbi filter uses HMMER's lazy-F test
//     (Dmax + ddbound_w > xB) and then relaxes chunk carries until no lane improves -- a fixed point of
//     max-plus, hence bit-identical to the serial evaluation; Forward/Backward use a 6-step affine scan.
// Integer semantics follow upstream impl_sse/vitfilter.c exactly (signed saturating 16-bit adds via
// v_add_i16 clamp); float semantics follow impl_sse/fwdback.c up to the association order of sums.
#include "p7x_wave.hpp"

namespace p7x {

// ======================================================================================= Viterbi filter
template <int C>
__global__ void __launch_bounds__(kWsBlock) vit_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Mpad = 64 * C;
  const WaveSeqArgs a = load_args<WaveSeqArgs>(ref);
  const int nlist = (a.abort_flag && *a.abort_flag) ? 0 : (a.nlist_ptr ? *a.nlist_ptr : a.nlist);
  if ((int) (blockIdx.x * (kWsBlock / 64)) >= nlist) return;         // no item for this block: skip the table load
  constexpr bool EG = C > 32;         // long models (M > 2048): the emission table is read where it lies (L2), only the transitions are staged
  uint4 *tr = reinterpret_cast<uint4 *>(smem);                       // [Mpad]
  const short *em = EG ? reinterpret_cast<const short *>(a.emis) : reinterpret_cast<const short *>(smem + (size_t) Mpad * 16);  // [kTabRows][Mpad]
  {
    const uint4 *gt = reinterpret_cast<const uint4 *>(a.trans);
    for (int i = threadIdx.x; i < Mpad; i += kWsBlock) tr[i] = gt[i];
    if constexpr (!EG) {
      const uint4 *ge = reinterpret_cast<const uint4 *>(a.emis);
      uint4 *le = reinterpret_cast<uint4 *>(smem + (size_t) Mpad * 16);
      for (int i = threadIdx.x; i < a.nrows * Mpad / 8; i += kWsBlock) le[i] = ge[i];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const short NEG = (short) -32768;

  P7X_WAVE_ITEMS(it) {
    const Item item = load_item(a, it);
    const int L = item.L;
    const uint8_t *sq = item.sq;
    const int xwm = rfl((int) a.xwmove_tab[L]);

    short mm[C], im[C], dm[C], tdd[C];
#pragma unroll unroll_c(C)
    for (int c = 0; c < C; ++c) { mm[c] = im[c] = dm[c] = NEG; tdd[c] = NEG; }
    int xN = a.base_w, xB = xN + xwm, xJ = -32768, xC = -32768;
    bool overflow = false;
    const int lt_thr = a.lt_thresh ? rfl(a.lt_thresh[it]) : INT_MAX;

    for (int i0 = 0; i0 < L && !overflow; i0 += 64) {
      const int nrow = min(64, L - i0);
      const uint32_t resid = (lane < nrow) ? sq[i0 + lane] : 0;
      for (int r = 0; r < nrow && !overflow; ++r) {
        const int x = __builtin_amdgcn_readlane((int) resid, r);
        const short *er = em + x * Mpad + lane;
        const short xBs = (short) xB;
        short mp = (short) dpp_shr1(mm[C - 1], NEG);
        short ip = (short) dpp_shr1(im[C - 1], NEG);
        short dp = (short) dpp_shr1(dm[C - 1], NEG);
        short rowmax = NEG, dmax = NEG, dcarry = NEG;
#pragma unroll unroll_c(C)
        for (int c = 0; c < C; ++c) {
          const uint4 t = tr[c * 64 + lane];
          short sv = adds16(xBs, lo16(t.x));
          sv = max16(sv, adds16(mp, hi16(t.x)));
          sv = max16(sv, adds16(ip, lo16(t.y)));
          sv = max16(sv, adds16(dp, hi16(t.y)));
          sv = adds16(sv, er[c * 64]);
          rowmax = max16(rowmax, sv);
          mp = mm[c]; ip = im[c]; dp = dm[c];
          im[c] = max16(adds16(mp, hi16(t.z)), adds16(ip, lo16(t.w)));
          mm[c] = sv;
          dm[c] = dcarry;                       // M(i,k-1) -> D(i,k); node c=0 is patched below
          dcarry = adds16(sv, lo16(t.z));
          dmax = max16(dmax, dcarry);
          tdd[c] = hi16(t.w);
        }
        dm[0] = (short) dpp_shr1(dcarry, NEG);

        const int xE = wave_max_i32((int) rowmax);
        if (xE >= lt_thr) {
          // long-target scan: every match cell that holds the row maximum seeds a window; the rows start afresh and the
          // special states keep their values (upstream p7_ViterbiFilter_longtarget)
#pragma unroll unroll_c(C)
          for (int c = 0; c < C; ++c) {
            const int k = lane * C + c + 1;
            if ((int) mm[c] == xE && k <= a.M) {
              const int slot = atomicAdd(a.lt_nrec, 1);
              if (slot < a.lt_cap) { a.lt_rec[3 * slot] = it; a.lt_rec[3 * slot + 1] = i0 + r + 1; a.lt_rec[3 * slot + 2] = k; }
            }
            mm[c] = im[c] = dm[c] = NEG;
          }
          continue;
        }
        if (xE >= 32767) overflow = true;
        xC = max(xC, xE + a.xw_e);              // xw[C][LOOP] = xw[J][LOOP] = xw[N][LOOP] = 0
        xJ = max(xJ, xE + a.xw_e);
        xB = max(xJ + xwm, xN + xwm);

        const int Dmax = wave_max_i32((int) dmax);
        if (Dmax + a.ddbound > xB) {            // lazy F: only now can a D->D path beat B->M on the next row
#pragma unroll unroll_c(C)
          for (int c = 1; c < C; ++c) dm[c] = max16(dm[c], adds16(dm[c - 1], tdd[c - 1]));
          for (int pass = 0; pass < 64; ++pass) { // a carry can cross at most 63 lane boundaries
            const short ddout = adds16(dm[C - 1], tdd[C - 1]);
            const short cand = (short) dpp_shr1(ddout, NEG);
            const int improved = wave_max_i32((cand > dm[0]) ? 1 : 0);
            if (improved == 0) break;
            dm[0] = max16(dm[0], cand);
#pragma unroll unroll_c(C)
            for (int c = 1; c < C; ++c) dm[c] = max16(dm[c], adds16(dm[c - 1], tdd[c - 1]));
          }
        }
      }
    }
    if (lane == 0) a.out_xC[it] = overflow ? 32767 : xC;
  }
}

// ======================================================================================= MSV, one target per wavefront
// The lane-per-target kernels of p7x_msv.hip keep a whole DP row in registers and stop at M = 478, and they need
// tens of thousands of targets to fill the device (64 per wavefront, a group lasts as long as its longest member).
// Longer models (a few per cent of Pfam) and small target blocks (hmmscan's query sequences: a few thousand
// wavefronts of work, where the latency of the longest sequence is what counts) take this wave-per-target form of
// the same recurrence: lane z owns nodes zC+1..zC+C,
// int arithmetic with the u8 semantics of impl_sse/msvfilter.c reproduced as in msv_kernel (floor at 0 is implied by
// the next row's max(., xB); the 255 clip can only follow a row that already reported overflow).  ~4x the
// instructions per cell of the fast kernel: a functional fallback, bit-exact (tests/test_gpu_filters.py).
template <int C>
__global__ void __launch_bounds__(kWsBlock) msv_wave_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Mpad = 64 * C;
  const MsvWaveArgs a = load_args<MsvWaveArgs>(ref);
  const int nlist = a.nslots;
  if ((int) (blockIdx.x * (kWsBlock / 64)) >= nlist) return;
  constexpr bool EG = C > 32;         // long models (M > 2048): the table is read where it lies (L2)
  const short *em = EG ? reinterpret_cast<const short *>(a.emis) : reinterpret_cast<const short *>(smem);      // [nrows][Mpad] bias - cost, kNegPad outside the model
  if constexpr (!EG) {
    const uint4 *ge = reinterpret_cast<const uint4 *>(a.emis);
    uint4 *le = reinterpret_cast<uint4 *>(smem);
    for (int i = threadIdx.x; i < a.nrows * Mpad / 8; i += kWsBlock) le[i] = ge[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  P7X_WAVE_ITEMS(it) {
    const int slot = it;
    const int L = rfl(a.slot_len[slot]);
    const unsigned long long off = (unsigned long long) a.slot_off[slot];
    const unsigned lo = (unsigned) rfl((int) (unsigned) off), hi = (unsigned) rfl((int) (unsigned) (off >> 32));
    const uint8_t *sq = a.dsq + (((unsigned long long) hi << 32) | lo);
    const int tjbm = rfl((int) a.tjb_tab[L]) + a.tbm;
    int mm[C];
#pragma unroll unroll_c(C)
    for (int c = 0; c < C; ++c) mm[c] = 0;
    int xJ = 0, xEmax = 0;
    int xB = max(a.base - tjbm, 0);
    // The begin score xB moves only when a row's maximum lifts xJ above the base, i.e. on a real hit.  Rows are therefore
    // run in blocks of kBlk with xB held: no wavefront reduction (and nothing else across the lanes but the one-lane
    // shift) sits on the row-to-row chain, and one reduction of the block's maximum tells whether the assumption held.
    // Where it did not (xE - tec above max(base, xJ) somewhere in the block), the block is repeated from its saved
    // first row with the reduction in every row, as the recurrence is written.  Same integers either way.
    constexpr int kBlk = C <= 8 ? 16 : 1;              // long models: a row is dozens of cells per lane, the reduction a small part of it
    for (int i0 = 0; i0 < L; i0 += 64) {
      const int nrow = min(64, L - i0);
      const uint32_t resid = (lane < nrow) ? sq[i0 + lane] : 0;
      for (int r0 = 0; r0 < nrow; r0 += kBlk) {
        if (kBlk == 1) {                               // the recurrence as written, one reduction per row
          const int x = __builtin_amdgcn_readlane((int) resid, r0);
          const short *er = em + x * Mpad + lane;
          int mp = dpp_shr1(mm[C - 1], 0);
          int rowmax = kNegPad;
#pragma unroll unroll_c(C)
          for (int c = 0; c < C; ++c) {
            const int sv = max(mp, xB) + (int) er[c * 64];
            mp = mm[c];
            mm[c] = sv;
            rowmax = max(rowmax, sv);
          }
          const int xE = wave_max_i32(rowmax);
          xEmax = max(xEmax, xE);
          xJ = max(xJ, xE - a.tec);
          xB = max(max(a.base, xJ) - tjbm, 0);
          continue;
        }
        const int nb = min(kBlk, nrow - r0);
        int saved[C];
#pragma unroll unroll_c(C)
        for (int c = 0; c < C; ++c) saved[c] = mm[c];
        int blkmax = kNegPad;
        auto row_held = [&](int r) {
          const int x = __builtin_amdgcn_readlane((int) resid, r);
          const short *er = em + x * Mpad + lane;
          int mp = dpp_shr1(mm[C - 1], 0);
#pragma unroll unroll_c(C)
          for (int c = 0; c < C; ++c) {
            const int sv = max(mp, xB) + (int) er[c * 64];
            mp = mm[c];
            mm[c] = sv;
            blkmax = max(blkmax, sv);
          }
        };
        if (C <= 4 && nb == kBlk) {                    // short models: the rows unrolled (their LDS reads issue ahead of the chain)
#pragma unroll
          for (int rr = 0; rr < kBlk; ++rr) row_held(r0 + rr);
        } else if (C <= 12) {
#pragma unroll 4
          for (int rr = 0; rr < nb; ++rr) row_held(r0 + rr);
        } else {
#pragma unroll 2
          for (int rr = 0; rr < nb; ++rr) row_held(r0 + rr);
        }
        const int xEb = wave_max_i32(blkmax);
        if (xEb - a.tec <= max(a.base, xJ)) {          // xB was right for every row of the block
          xEmax = max(xEmax, xEb);
          xJ = max(xJ, xEb - a.tec);
          continue;
        }
#pragma unroll unroll_c(C)
        for (int c = 0; c < C; ++c) mm[c] = saved[c];
        for (int r = r0; r < r0 + nb; ++r) {
          const int x = __builtin_amdgcn_readlane((int) resid, r);
          const short *er = em + x * Mpad + lane;
          int mp = dpp_shr1(mm[C - 1], 0);
          int rowmax = kNegPad;
#pragma unroll unroll_c(C)
          for (int c = 0; c < C; ++c) {
            const int sv = max(mp, xB) + (int) er[c * 64];
            mp = mm[c];
            mm[c] = sv;
            rowmax = max(rowmax, sv);
          }
          const int xE = wave_max_i32(rowmax);
          xEmax = max(xEmax, xE);
          xJ = max(xJ, xE - a.tec);
          xB = max(max(a.base, xJ) - tjbm, 0);
        }
      }
    }
    if (lane == 0 && L > 0) a.out_xJ[slot] = (xEmax >= 255 - a.bias) ? (int16_t) -1 : (int16_t) xJ;
  }
}

// ---------------------------------------------------------------------------------------- packed form, long models
// Models beyond the register-resident lane kernels (M > 1021; a few per cent of Pfam, a fifth of its MSV cells) went
// through msv_wave_kernel at five instructions per cell and one wavefront per SIMD (the emission table of such a model
// fills most of the LDS, and a block was four wavefronts).  Here the same recurrence runs on packed 16-bit pairs --
// two cells per register, values offset by -32768 so that the saturating add is the floor at zero, the one-node shift
// of a row done with v_alignbit across the pair boundary (and one DPP move across the lane boundary): alignbit, max
// with the begin score, saturating add of the emission pair, max into the row maximum = four operations per two cells
// -- and sixteen wavefronts share one copy of the table.  Emission pairs come with one ds_read_b64 per four cells from
// a table laid out [residue][pair of pairs][lane].  Bit-identical to msv_wave_kernel (tests/test_gpu_filters.py).
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v u2s(uint32_t u) { return __builtin_bit_cast(s2v, u); }
__device__ __forceinline__ uint32_t s2u(s2v v) { return __builtin_bit_cast(uint32_t, v); }
constexpr int kPkWaves = 16;
template <int C>
__global__ void __launch_bounds__(kPkWaves * 64) msv_wavepk_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(C % 4 == 0, "pairs of pairs");
  constexpr int P = C / 2, P2 = C / 4;
  constexpr uint32_t kFloor = 0x80008000u;
  const MsvWaveArgs a = load_args<MsvWaveArgs>(ref);
  const int nlist = a.nslots;
  if ((int) (blockIdx.x * kPkWaves) >= nlist) return;
  {
    const uint4 *g = reinterpret_cast<const uint4 *>(a.emis_pk);
    uint4 *l = reinterpret_cast<uint4 *>(smem);
    for (int i = threadIdx.x; i < a.nrows * P * 64 / 4; i += kPkWaves * 64) l[i] = g[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint2 *tab = reinterpret_cast<const uint2 *>(smem) + lane;          // [residue][P2][64] b64
  const int wave0 = rfl((int) (blockIdx.x * kPkWaves + (threadIdx.x >> 6)));
  const int nwaves = (int) (gridDim.x * kPkWaves);
  for (int it = wave0; it < nlist; it += nwaves) {
    const int slot = it;
    const int L = rfl(a.slot_len[slot]);
    const unsigned long long off = (unsigned long long) a.slot_off[slot];
    const unsigned lo = (unsigned) rfl((int) (unsigned) off), hi = (unsigned) rfl((int) (unsigned) (off >> 32));
    const uint8_t *sq = a.dsq + (((unsigned long long) hi << 32) | lo);
    const int tjbm = rfl((int) a.tjb_tab[L]) + a.tbm;
    uint32_t mm[P];
#pragma unroll
    for (int j = 0; j < P; ++j) mm[j] = kFloor;
    int xJ = 0, xEmax = 0;
    int xB = max(a.base - tjbm, 0);
    // one row: every cell's predecessor is the node before it in the previous row (alignbit across the pair boundary,
    // <left> across the register and the lane boundary)
    auto row = [&](const uint2 (&ee)[P2], const s2v xBs, s2v &acc) {
      uint32_t left = (uint32_t) dpp_shr1((int) mm[P - 1], (int) kFloor);
#pragma unroll
      for (int j2 = 0; j2 < P2; ++j2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * j2 + h;
          const uint32_t cur = mm[j];
          const uint32_t mp = __builtin_amdgcn_alignbit(cur, left, 16);      // (node before the pair's first, the pair's first)
          const s2v sv = __builtin_elementwise_add_sat(__builtin_elementwise_max(u2s(mp), xBs), u2s(h == 0 ? ee[j2].x : ee[j2].y));
          mm[j] = s2u(sv);
          acc = __builtin_elementwise_max(acc, sv);
          left = cur;
        }
      }
    };
    constexpr int kBlk = 8;
    for (int i0 = 0; i0 < L; i0 += 64) {
      const int nrow = min(64, L - i0);
      const uint32_t resid = (lane < nrow) ? sq[i0 + lane] : 0;
      auto load_e = [&](uint2 (&dst)[P2], int r) {
        const int x = __builtin_amdgcn_readlane((int) resid, r);
#pragma unroll
        for (int j2 = 0; j2 < P2; ++j2) dst[j2] = tab[(x * P2 + j2) * 64];
      };
      for (int r0 = 0; r0 < nrow; r0 += kBlk) {
        const int nb = min(kBlk, nrow - r0);
        const s2v xBs = u2s((uint32_t) ((xB - 32768) & 0xffff) * 0x00010001u);
        if (nb == kBlk) {
          // The begin score moves only when a row's maximum lifts xJ above the base (a real hit): eight rows with xB held,
          // their emissions fetched a row ahead, one reduction of their common maximum.  If it shows that xB would have
          // moved, the rows are repeated from the saved first one with the reduction in every row.
          uint32_t saved[P];
#pragma unroll
          for (int j = 0; j < P; ++j) saved[j] = mm[j];
          s2v acc = u2s(kFloor);
          uint2 ea[P2], eb[P2];
          load_e(ea, r0);
#pragma unroll
          for (int rr = 0; rr < kBlk; rr += 2) {
            load_e(eb, r0 + rr + 1);
            row(ea, xBs, acc);
            if (rr + 2 < kBlk) load_e(ea, r0 + rr + 2);
            row(eb, xBs, acc);
          }
          const int xEb = wave_max_i32(max((int) acc.x, (int) acc.y)) + 32768;
          if (xEb - a.tec <= max(a.base, xJ)) {
            xEmax = max(xEmax, xEb);
            xJ = max(xJ, xEb - a.tec);
            continue;
          }
#pragma unroll
          for (int j = 0; j < P; ++j) mm[j] = saved[j];
        }
        for (int r = r0; r < r0 + nb; ++r) {
          uint2 e1[P2];
          load_e(e1, r);
          s2v rowmax = u2s(kFloor);
          row(e1, u2s((uint32_t) ((xB - 32768) & 0xffff) * 0x00010001u), rowmax);
          const int xE = wave_max_i32(max((int) rowmax.x, (int) rowmax.y)) + 32768;
          xEmax = max(xEmax, xE);
          xJ = max(xJ, xE - a.tec);
          xB = max(max(a.base, xJ) - tjbm, 0);
        }
      }
    }
    if (lane == 0 && L > 0) a.out_xJ[slot] = (xEmax >= 255 - a.bias) ? (int16_t) -1 : (int16_t) xJ;
  }
}

// ======================================================================================= Forward parser
// EG: the emission table does not fit in LDS next to the transitions (M > 1024) and is read from global memory
// (it stays L2-resident: 30 rows x Mpad floats).
template <int C, bool EG = false>
__global__ void __launch_bounds__(kWsBlock) fwd_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Mpad = 64 * C;
  const WaveSeqArgs a = load_args<WaveSeqArgs>(ref);
  const int nlist = (a.abort_flag && *a.abort_flag) ? 0 : (a.nlist_ptr ? *a.nlist_ptr : a.nlist);
  if ((int) (blockIdx.x * (kWsBlock / 64)) >= nlist) return;         // no item for this block: skip the table load
  constexpr bool TG = C > 64;         // M > 4096: the transitions no longer fit the LDS either and are read through L2
  const float4 *tr = TG ? reinterpret_cast<const float4 *>(a.trans) : reinterpret_cast<const float4 *>(smem);      // [2*Mpad]
  const float *em = EG ? reinterpret_cast<const float *>(a.emis) : reinterpret_cast<const float *>(smem + (size_t) Mpad * 32);      // [kTabRows][Mpad]
  {
    if constexpr (!TG) {
      const float4 *gt = reinterpret_cast<const float4 *>(a.trans);
      float4 *lt = reinterpret_cast<float4 *>(smem);
      for (int i = threadIdx.x; i < 2 * Mpad; i += kWsBlock) lt[i] = gt[i];
    }
    if constexpr (!EG) {
      const float4 *ge = reinterpret_cast<const float4 *>(a.emis);
      float4 *le = reinterpret_cast<float4 *>(smem + (size_t) Mpad * 32);
      for (int i = threadIdx.x; i < a.nrows * Mpad / 4; i += kWsBlock) le[i] = ge[i];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;

  P7X_WAVE_ITEMS(it) {
    const Item item = load_item(a, it);
    const int L = item.L;
    const uint8_t *sq = item.sq;
    float *xo = a.xmx ? a.xmx + a.xmx_off[it] : nullptr;
    const float pmove = (2.0f + 1.0f) / ((float) L + 2.0f + 1.0f), ploop = 1.0f - pmove;

    float mm[C], im[C], dm[C];
#pragma unroll unroll_c(C)
    for (int c = 0; c < C; ++c) mm[c] = im[c] = dm[c] = 0.0f;
    // product of this lane's D->D probabilities: the multiplier of an incoming D carry
    float ddprod = 1.0f;
#pragma unroll unroll_c(C)
    for (int c = 0; c < C; ++c) ddprod *= tr[2 * (c * 64 + lane) + 1].w;

    float xN = 1.0f, xB = pmove, xJ = 0.0f, xC = 0.0f, xE = 0.0f, totscale = 0.0f;
    if (xo && lane == 0) { xo[0] = 0.0f; xo[1] = 1.0f; xo[2] = 0.0f; xo[3] = xB; xo[4] = 0.0f; xo[5] = 1.0f; }
    for (int i0 = 0; i0 < L; i0 += 64) {
     const int nrow = min(64, L - i0);
     const uint32_t resid = (lane < nrow) ? sq[i0 + lane] : 0;
     for (int r = 0; r < nrow; ++r) {
      const int i = i0 + r;
      const int x = __builtin_amdgcn_readlane((int) resid, r);
      const float *er = em + x * Mpad + lane;
      float mp = dpp_shr1f(mm[C - 1], 0.0f), ip = dpp_shr1f(im[C - 1], 0.0f), dp = dpp_shr1f(dm[C - 1], 0.0f);
      float esum = 0.0f, dcarry = 0.0f;
      float tdd[C], tmd[C];
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) {
        const F8 t = load_f8(tr, c * 64 + lane);
        float sv = xB * t.bm;
        sv = sv + mp * t.mm;
        sv = sv + ip * t.im;
        sv = sv + dp * t.dm;
        sv = sv * er[c * 64];
        esum = esum + sv;
        mp = mm[c]; ip = im[c]; dp = dm[c];
        im[c] = mp * t.mi + ip * t.ii;
        mm[c] = sv;
        tdd[c] = t.dd; tmd[c] = t.md;
      }
      // D(i,k) = M(i,k-1) tMD(k-1) + D(i,k-1) tDD(k-1): serial inside the lane, affine scan across lanes
      float A = 0.0f;                                   // this lane's outgoing carry for a zero incoming carry
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) { dm[c] = A; A = mm[c] * tmd[c] + A * tdd[c]; }
      float sa = A, sp = ddprod;                        // inclusive scan of (carry, multiplier)
      affine_scan_up(sa, sp);
      dcarry = dpp_shr1f(sa, 0.0f);                     // exclusive: the carry entering this lane
      {
        float w = dcarry;
#pragma unroll unroll_c(C)
        for (int c = 0; c < C; ++c) { dm[c] = dm[c] + w; esum = esum + dm[c]; w = w * tdd[c]; }
      }
      xE = wave_sum_f32(esum);
      xN = xN * ploop;
      xC = (xC * ploop) + (xE * a.xf_e_move);
      xJ = (xJ * ploop) + (xE * a.xf_e_loop);
      xB = (xJ * pmove) + (xN * pmove);
      float scale = 1.0f;
      if (xE > 1.0e4f) {
        xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
        const float inv = (float) (1.0 / (double) xE);
#pragma unroll unroll_c(C)
        for (int c = 0; c < C; ++c) { mm[c] *= inv; dm[c] *= inv; im[c] *= inv; }
        scale = xE;
        totscale = (float) ((double) totscale + log((double) xE));
        xE = 1.0f;
      }
      if (xo && lane == 0) {
        float *row = xo + (size_t) (i + 1) * 6;
        row[0] = xE; row[1] = xN; row[2] = xJ; row[3] = xB; row[4] = xC; row[5] = scale;
      }
     }
    }
    if (lane == 0) {
      float sc;
      if (xC != xC) sc = __builtin_nanf("");
      else if ((L > 0 && xC == 0.0f) || __builtin_isinf(xC)) sc = __builtin_inff();
      else sc = (float) ((double) totscale + log((double) (xC * pmove)));
      a.out_sc[it] = sc;
    }
  }
}

// ======================================================================================= Backward parser
// Mirror of the Forward parser, re-using Forward's per-row scale factors (upstream backward_engine).
template <int C, bool EG = false>
__global__ void __launch_bounds__(kWsBlock) bck_kernel(const ArgRef ref)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Mpad = 64 * C;
  const WaveSeqArgs a = load_args<WaveSeqArgs>(ref);
  const int nlist = (a.abort_flag && *a.abort_flag) ? 0 : (a.nlist_ptr ? *a.nlist_ptr : a.nlist);
  if ((int) (blockIdx.x * (kWsBlock / 64)) >= nlist) return;         // no item for this block: skip the table load
  constexpr bool TG = C > 64;         // M > 4096: the transitions no longer fit the LDS either and are read through L2
  const float4 *tr = TG ? reinterpret_cast<const float4 *>(a.trans) : reinterpret_cast<const float4 *>(smem);      // [2*Mpad]
  const float *em = EG ? reinterpret_cast<const float *>(a.emis) : reinterpret_cast<const float *>(smem + (size_t) Mpad * 32);      // [kTabRows][Mpad]
  {
    if constexpr (!TG) {
      const float4 *gt = reinterpret_cast<const float4 *>(a.trans);
      float4 *lt = reinterpret_cast<float4 *>(smem);
      for (int i = threadIdx.x; i < 2 * Mpad; i += kWsBlock) lt[i] = gt[i];
    }
    if constexpr (!EG) {
      const float4 *ge = reinterpret_cast<const float4 *>(a.emis);
      float4 *le = reinterpret_cast<float4 *>(smem + (size_t) Mpad * 32);
      for (int i = threadIdx.x; i < a.nrows * Mpad / 4; i += kWsBlock) le[i] = ge[i];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;

  P7X_WAVE_ITEMS(it) {
    const Item item = load_item(a, it);
    const int L = item.L;
    const uint8_t *sq = item.sq;
    const float *fx = a.fwd_xmx + a.xmx_off[it];
    float *xo = a.xmx + a.xmx_off[it];
    const float pmove = (2.0f + 1.0f) / ((float) L + 2.0f + 1.0f), ploop = 1.0f - pmove;

    // per-lane constants: transitions leaving node (c) and entering node (c+1) [next node]
    float tmd[C], tdd[C], tmi[C], tii[C];       // leaving node k
    float nmm[C], nim[C], ndm[C];               // entering node k+1 (B->M uses node k itself: bm[c])
    float bm[C];
    float ddprod = 1.0f;
#pragma unroll unroll_c(C)
    for (int c = 0; c < C; ++c) {
      const F8 t = load_f8(tr, c * 64 + lane);
      tmd[c] = t.md; tdd[c] = t.dd; tmi[c] = t.mi; tii[c] = t.ii; bm[c] = t.bm;
      ddprod *= t.dd;
    }
#pragma unroll unroll_c(C)
    for (int c = 0; c < C; ++c) {
      float mmn, imn, dmn;
      if (c + 1 < C) { const F8 t = load_f8(tr, (c + 1) * 64 + lane); mmn = t.mm; imn = t.im; dmn = t.dm; }
      else {
        const F8 t = load_f8(tr, 0 * 64 + lane);              // first node of the next lane
        mmn = dpp_shl1f(t.mm, 0.0f); imn = dpp_shl1f(t.im, 0.0f); dmn = dpp_shl1f(t.dm, 0.0f);
      }
      nmm[c] = mmn; nim[c] = imn; ndm[c] = dmn;
    }

    float mm[C], im[C], dm[C];
    float xJ = 0.0f, xB = 0.0f, xN = 0.0f;
    float xC = pmove;
    float xE = xC * a.xf_e_move;
    float totscale = 0.0f;
    bool own_scales = false;

    // D chain helper: D(k) = base(k) + D(k+1) tDD(k), running from the last node to the first; affine scan
    // across lanes in reverse lane order.  On entry dm[c] holds base(k); on exit the full D(i,k).
    auto d_chain = [&](float (&d)[C]) {
      float A = 0.0f;                                   // outgoing carry (towards lower k) for zero incoming
#pragma unroll unroll_c(C)
      for (int c = C - 1; c >= 0; --c) { A = d[c] + A * tdd[c]; }
      float sa = A, sp = ddprod;
      affine_scan_down(sa, sp, lane);
      float w = dpp_shl1f(sa, 0.0f);                    // D of the first node of the next lane
#pragma unroll unroll_c(C)
      for (int c = C - 1; c >= 0; --c) { d[c] = d[c] + w * tdd[c]; w = d[c]; }
    };

    // row L
#pragma unroll unroll_c(C)
    for (int c = 0; c < C; ++c) { mm[c] = xE; dm[c] = xE; im[c] = 0.0f; }
    d_chain(dm);
    {
      float dn = dpp_shl1f(dm[0], 0.0f);
#pragma unroll unroll_c(C)
      for (int c = C - 1; c >= 0; --c) { mm[c] = mm[c] + dn * tmd[c]; dn = dm[c]; }
    }
    float sc = __builtin_bit_cast(float, rfl(__builtin_bit_cast(int, fx[(size_t) L * 6 + 5])));
    if (sc > 1.0f) {
      xE = xE / sc; xN = xN / sc; xC = xC / sc; xJ = xJ / sc; xB = xB / sc;
      const float inv = (float) (1.0 / (double) sc);
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) { mm[c] *= inv; dm[c] *= inv; im[c] *= inv; }
    }
    totscale = (float) log((double) sc);
    if (lane == 0) { float *r = xo + (size_t) L * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = sc; }

    // Residue x_{i+1} (0-based index i) and Forward's scale factor of row i come 64 rows at a time, one per lane: a load
    // inside the row loop is waited for with vmcnt(0), which also covers the row stores issued before it (the counter
    // retires in order) -- two memory round trips per row.
    for (int ib = L - 1; ib >= 1; ib -= 64) {
    const int nblk = min(64, ib);
    uint32_t res_b = 0; float fsc_b = 0.0f;
    if (lane < nblk) { res_b = sq[ib - lane]; fsc_b = fx[(size_t) (ib - lane) * 6 + 5]; }
    for (int l = 0; l < nblk; ++l) {
      const int i = ib - l;
      const int x = __builtin_amdgcn_readlane((int) res_b, l);
      const float *er = em + x * Mpad + lane;
      // mp(k) = M(i+1,k+1) e(x_{i+1},k+1): value of the NEXT node
      float me[C];
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) me[c] = mm[c] * er[c * 64];
      float bsum = 0.0f;
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) bsum = bsum + me[c] * bm[c];
      const float me_next0 = dpp_shl1f(me[0], 0.0f);
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) {
        const float mp = (c + 1 < C) ? me[c + 1] : me_next0;
        const float ipv = im[c];
        im[c] = ipv * tii[c] + mp * nim[c];
        dm[c] = mp * ndm[c];
        mm[c] = ipv * tmi[c] + mp * nmm[c];
      }
      xB = wave_sum_f32(bsum);
      xC = xC * ploop;
      xJ = (xB * pmove) + (xJ * ploop);
      xN = (xB * pmove) + (xN * ploop);
      xE = (xC * a.xf_e_move) + (xJ * a.xf_e_loop);
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) { dm[c] = dm[c] + xE; mm[c] = mm[c] + xE; }
      d_chain(dm);
      {
        float dn = dpp_shl1f(dm[0], 0.0f);
#pragma unroll unroll_c(C)
        for (int c = C - 1; c >= 0; --c) { mm[c] = mm[c] + dn * tmd[c]; dn = dm[c]; }
      }
      if (xB > 1.0e16f) own_scales = true;
      sc = own_scales ? ((xB > 1.0e4f) ? xB : 1.0f) : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fsc_b), l));
      if (sc > 1.0f) {
        xE /= sc; xN /= sc; xJ /= sc; xB /= sc; xC /= sc;
        const float inv = (float) (1.0 / (double) sc);
#pragma unroll unroll_c(C)
        for (int c = 0; c < C; ++c) { mm[c] *= inv; dm[c] *= inv; im[c] *= inv; }
        totscale = (float) ((double) totscale + log((double) sc));
      }
      if (lane == 0) { float *r = xo + (size_t) i * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = sc; }
    }
    }
    // row 0
    {
      const int x = rfl((int) sq[0]);
      const float *er = em + x * Mpad + lane;
      float bsum = 0.0f;
#pragma unroll unroll_c(C)
      for (int c = 0; c < C; ++c) bsum = bsum + (mm[c] * er[c * 64]) * bm[c];
      xB = wave_sum_f32(bsum);
      xN = (xB * pmove) + (xN * ploop);
      if (lane == 0) {
        float *r = xo; r[0] = 0.0f; r[1] = xN; r[2] = 0.0f; r[3] = xB; r[4] = 0.0f; r[5] = 1.0f;
        float scv;
        if (xN != xN || (L > 0 && xN == 0.0f) || __builtin_isinf(xN)) scv = __builtin_inff();
        else scv = (float) ((double) totscale + log((double) xN));
        a.out_sc[it] = scv;
      }
    }
  }
}

// ---------------------------------------------------------------------------- host side
static const int kCList[] = { 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64, 96, 128 };      // M <= 8192

int vit_pick_C(int M)
{
  const int need = (M + 63) / 64;
  for (int c : kCList) if (c >= need) return c;
  return -1;
}

// occupancy and the LDS opt-in of one kernel instantiation, looked up once
struct KernelInfo { std::mutex mu; std::map<std::pair<int, const void *>, int> per_cu; };      // by (device, kernel): the opt-in is per device
static KernelInfo &kernel_info() { static KernelInfo *k = new KernelInfo(); return *k; }
template <typename K>
static int blocks_per_cu(K kernel, size_t lds_bytes, int *out)
{
  KernelInfo &ki = kernel_info();
  std::lock_guard<std::mutex> lk(ki.mu);
  int dev = 0; P7X_HIP(hipGetDevice(&dev));
  const void *fn = reinterpret_cast<const void *>(kernel);
  const std::pair<int, const void *> key(dev, fn);
  auto it = ki.per_cu.find(key);
  if (it != ki.per_cu.end()) { *out = it->second; return P7X_OK; }
  if (lds_bytes > 64 * 1024)
    P7X_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
  int per_cu = 0;
  P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWsBlock, lds_bytes));
  if (per_cu < 1) per_cu = 1;
  ki.per_cu[key] = per_cu;
  *out = per_cu;
  return P7X_OK;
}

template <typename K>
static int launch_ws(K kernel, const ArgRun<WaveSeqArgs> &a, size_t lds_bytes, int num_cu, hipStream_t st)
{
  long want = 0;
  for (int i = 0; i < a.n; ++i) want = std::max<long>(want, ((long) a.at(i).nlist + 3) / 4);      // nlist bounds the device-side count
  if (want <= 0) return P7X_OK;
  int per_cu = 0;
  const int pst = blocks_per_cu(kernel, lds_bytes, &per_cu);
  if (pst != P7X_OK) return pst;
  hipLaunchKernelGGL(kernel, dim3(lane_grid(want, (long) num_cu * per_cu, a.n), (unsigned) a.n), dim3(kWsBlock), lds_bytes, st, a.ref());
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

#define P7X_C_SWITCH(KERNEL, BYTES_PER_NODE, EBYTES)                                                        \
  switch (C) {                                                                                              \
    case 1:  return launch_ws(KERNEL<1>,  a, (size_t) 64 * 1  * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 2:  return launch_ws(KERNEL<2>,  a, (size_t) 64 * 2  * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 3:  return launch_ws(KERNEL<3>,  a, (size_t) 64 * 3  * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 4:  return launch_ws(KERNEL<4>,  a, (size_t) 64 * 4  * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 5:  return launch_ws(KERNEL<5>,  a, (size_t) 64 * 5  * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 6:  return launch_ws(KERNEL<6>,  a, (size_t) 64 * 6  * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 8:  return launch_ws(KERNEL<8>,  a, (size_t) 64 * 8  * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 10: return launch_ws(KERNEL<10>, a, (size_t) 64 * 10 * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 12: return launch_ws(KERNEL<12>, a, (size_t) 64 * 12 * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 16: return launch_ws(KERNEL<16>, a, (size_t) 64 * 16 * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 20: return launch_ws(KERNEL<20>, a, (size_t) 64 * 20 * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 24: return launch_ws(KERNEL<24>, a, (size_t) 64 * 24 * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 32: return launch_ws(KERNEL<32>, a, (size_t) 64 * 32 * (BYTES_PER_NODE + nrows * EBYTES), num_cu, st); \
    case 48: return launch_ws(KERNEL<48>, a, (size_t) 64 * 48 * BYTES_PER_NODE, num_cu, st);                \
    case 64: return launch_ws(KERNEL<64>, a, (size_t) 64 * 64 * BYTES_PER_NODE, num_cu, st);                \
    case 96: return launch_ws(KERNEL<96>, a, (size_t) 64 * 96 * BYTES_PER_NODE, num_cu, st);                \
    case 128: return launch_ws(KERNEL<128>, a, (size_t) 64 * 128 * BYTES_PER_NODE, num_cu, st);             \
    default: set_error("model too long for the wave-per-sequence kernels (M > 8192)"); return P7X_EINVAL;  \
  }

int vit_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st)
{
  if (a.n <= 0) return P7X_OK;
  const int C = a.at(0).C, nrows = a.at(0).nrows;
  P7X_C_SWITCH(vit_kernel, 16, 2)
}

#define P7X_CF_SWITCH(KERNEL)                                                                               \
  switch (C) {                                                                                              \
    case 1:  return launch_ws(KERNEL<1>,  a, (size_t) 64 * 1  * (32 + nrows * 4), num_cu, st);           \
    case 2:  return launch_ws(KERNEL<2>,  a, (size_t) 64 * 2  * (32 + nrows * 4), num_cu, st);           \
    case 3:  return launch_ws(KERNEL<3>,  a, (size_t) 64 * 3  * (32 + nrows * 4), num_cu, st);           \
    case 4:  return launch_ws(KERNEL<4>,  a, (size_t) 64 * 4  * (32 + nrows * 4), num_cu, st);           \
    case 5:  return launch_ws(KERNEL<5>,  a, (size_t) 64 * 5  * (32 + nrows * 4), num_cu, st);           \
    case 6:  return launch_ws(KERNEL<6>,  a, (size_t) 64 * 6  * (32 + nrows * 4), num_cu, st);           \
    case 8:  return launch_ws(KERNEL<8>,  a, (size_t) 64 * 8  * (32 + nrows * 4), num_cu, st);           \
    case 10: return launch_ws(KERNEL<10>, a, (size_t) 64 * 10 * (32 + nrows * 4), num_cu, st);           \
    case 12: return launch_ws(KERNEL<12>, a, (size_t) 64 * 12 * (32 + nrows * 4), num_cu, st);           \
    case 16: return launch_ws(KERNEL<16>, a, (size_t) 64 * 16 * (32 + nrows * 4), num_cu, st);           \
    case 20: return launch_ws(KERNEL<20, true>, a, (size_t) 64 * 20 * 32, num_cu, st);                     \
    case 24: return launch_ws(KERNEL<24, true>, a, (size_t) 64 * 24 * 32, num_cu, st);                     \
    case 32: return launch_ws(KERNEL<32, true>, a, (size_t) 64 * 32 * 32, num_cu, st);                     \
    case 48: return launch_ws(KERNEL<48, true>, a, (size_t) 64 * 48 * 32, num_cu, st);                     \
    case 64: return launch_ws(KERNEL<64, true>, a, (size_t) 64 * 64 * 32, num_cu, st);                     \
    case 96: return launch_ws(KERNEL<96, true>, a, (size_t) 256, num_cu, st);                              \
    case 128: return launch_ws(KERNEL<128, true>, a, (size_t) 256, num_cu, st);                            \
    default: set_error("model too long for the Forward/Backward kernels (M > 8192)"); return P7X_EINVAL;   \
  }

int msv_wave_launch(const ArgRun<MsvWaveArgs> &a, int num_cu, hipStream_t st)
{
  long want = 0;
  for (int i = 0; i < a.n; ++i) want = std::max<long>(want, ((long) a.at(i).nslots + 3) / 4);
  if (want <= 0) return P7X_OK;
  const int C = a.at(0).C, nrows = a.at(0).nrows;
  if (a.at(0).emis_pk && C >= 20) {          // long models: packed pairs, sixteen wavefronts per copy of the table
    long wantpk = 0;
    for (int i = 0; i < a.n; ++i) wantpk = std::max<long>(wantpk, ((long) a.at(i).nslots + kPkWaves - 1) / kPkWaves);
    auto gopk = [&](auto kernel) -> int {
      const size_t lds = (size_t) 64 * C * nrows * 2;
      P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      hipLaunchKernelGGL(kernel, dim3(lane_grid(wantpk, (long) num_cu, a.n), (unsigned) a.n), dim3(kPkWaves * 64), lds, st, a.ref());
      P7X_HIP(hipGetLastError());
      return P7X_OK;
    };
    switch (C) {
      case 20: return gopk(msv_wavepk_kernel<20>);
      case 24: return gopk(msv_wavepk_kernel<24>);
      case 32: return gopk(msv_wavepk_kernel<32>);
      case 36: return gopk(msv_wavepk_kernel<36>);      // 2,048 < M <= 2,304 (the other wavefront kernels of such a model: C = 48)
      case 40: return gopk(msv_wavepk_kernel<40>);      // ... <= 2,560
      default: break;
    }
  }
  auto go = [&](auto kernel) -> int {
    const size_t lds = C > 32 ? (size_t) 256 : (size_t) 64 * C * nrows * 2;
    int per_cu = 0;
    const int pst = blocks_per_cu(kernel, lds, &per_cu);
    if (pst != P7X_OK) return pst;
    hipLaunchKernelGGL(kernel, dim3(lane_grid(want, (long) num_cu * per_cu, a.n), (unsigned) a.n), dim3(kWsBlock), lds, st, a.ref());
    P7X_HIP(hipGetLastError());
    return P7X_OK;
  };
  switch (C) {
    case 1:  return go(msv_wave_kernel<1>);
    case 2:  return go(msv_wave_kernel<2>);
    case 3:  return go(msv_wave_kernel<3>);
    case 4:  return go(msv_wave_kernel<4>);
    case 5:  return go(msv_wave_kernel<5>);
    case 6:  return go(msv_wave_kernel<6>);
    case 8:  return go(msv_wave_kernel<8>);
    case 10: return go(msv_wave_kernel<10>);
    case 12: return go(msv_wave_kernel<12>);
    case 16: return go(msv_wave_kernel<16>);
    case 20: return go(msv_wave_kernel<20>);
    case 24: return go(msv_wave_kernel<24>);
    case 32: return go(msv_wave_kernel<32>);
    case 48: return go(msv_wave_kernel<48>);
    case 64: return go(msv_wave_kernel<64>);
    case 96: return go(msv_wave_kernel<96>);
    case 128: return go(msv_wave_kernel<128>);
    default: set_error("no wave-per-target MSV kernel for this model length"); return P7X_EINVAL;
  }
}

int fwd_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st)
{
  if (a.n <= 0) return P7X_OK;
  const int C = a.at(0).C, nrows = a.at(0).nrows;
  P7X_CF_SWITCH(fwd_kernel)
}
int bck_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st)
{
  if (a.n <= 0) return P7X_OK;
  const int C = a.at(0).C, nrows = a.at(0).nrows;
  P7X_CF_SWITCH(bck_kernel)
}

} // namespace p7x
