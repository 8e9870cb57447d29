// p7x_vitpk.hip -- Viterbi filter, packed: T lanes per target (T = 8 or 16), 64/T targets per wavefront.
//
// The wave-per-target kernel in p7x_vitfwd.hip spends most of a row on work that does not shrink with the model:
// wave-wide reductions, the scalar special-state chain, and 16-bit values in 32-bit lanes.  For the common
// Pfam-sized models (M <= 640) this kernel shares those costs between several targets:
//   * the nodes are striped over the 2T half-lanes of a group (Farrar): register q of lane s holds node q+1+s*P in
//     its low half and node q+1+(s+T)*P in its high half, so the k-1 neighbour of a register is the register
//     before it and only register 0 takes values from another lane (v_pk_add_i16 clamp / v_pk_max_i16 reproduce
//     _mm_adds_epi16 / _mm_max_epi16 of impl_sse/vitfilter.c exactly);
//   * transitions (8 x int16 per node: BM MM IM DM MD MI II DD) and emissions come from LDS as ds_read_b128,
//     the DP rows (M, I, D) stay in 3P registers, two sets of them (a row reads one and writes the other);
//   * xE / Dmax are 3- or 4-step DPP butterflies inside the group, the special states are per-lane integers that
//     are uniform within a group; targets that have ended are masked, the row loop runs to the longest target of
//     the wavefront;
//   * lazy F follows upstream: D gets only M->D unless Dmax + ddbound > xB for that target; then the D->D closure
//     is evaluated in full: two packed ops per register walk every stripe, the carry into the next stripe is a
//     stripe shift, passes repeat until nothing improves (two on average).
// Results are bit-identical to p7x_vitfwd.hip::vit_kernel and to the oracle (tests/test_gpu_filters.py).
#include <map>
#include "p7x_wave.hpp"
#include <mutex>

namespace p7x {

namespace {

typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v as_s2v(uint32_t u) { return __builtin_bit_cast(s2v, u); }
__device__ __forceinline__ uint32_t as_u(s2v v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_add_sat(as_s2v(a), as_s2v(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_s2v(a), as_s2v(b))); }
__device__ __forceinline__ uint32_t splat16(int v) { const uint32_t w = (uint32_t) v & 0xffffu; return w | (w << 16); }
__device__ __forceinline__ int lo_of(uint32_t w) { return (int) (short) (w & 0xffffu); }
__device__ __forceinline__ int hi_of(uint32_t w) { return (int) (short) (w >> 16); }

constexpr uint32_t kNeg2 = 0x80008000u;      // (-32768, -32768)
// LDS image of the emission rows.  A row is (P+3)/4 chunks, a chunk the uint4 of every lane of a group (pairs 4q .. 4q+3).
// The groups of a wavefront read the rows of different residues, and a ds_read_b128 is served 16 lanes at a time
// (lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... of the 64): which 16-byte slots of the 256-byte bank row those
// lanes hit depends on the residues.
//   T = 16: a chunk is a whole bank row; with a row stride that is a multiple of 16 slots the 16 lanes of a service
//           group always cover 16 different slots, whatever the residues (the odd stride of rounds 1-3 did not).
//   T = 8:  a chunk is half a bank row and a service group takes four lanes of each of four groups (four residues), two of
//           them on slots 0-3 and two on 4-7 of their chunks.  With a stride of 8 mod 16 slots a row starts on either half
//           of a bank row, and a pair conflicts only when its two residues differ and have the same parity: 0.7 extra LDS
//           clocks per service group on average against 1.3 with the odd stride (random residues give random 4-slot
//           windows there, which overlap most of the time).  Measured, KR (M = 262, <8, 17>), 7 queries x 21,113
//           survivors: 3.43 ms against 3.61.  Keeping every chunk twice (slots 0-7 and 8-15, the groups whose second
//           bit is set reading the second copy) removes the conflicts altogether and was slower, 4.07 ms: 40 KB of
//           emissions to stage per workgroup, and the kernel is bound by its packed arithmetic (80 % VALU), not by LDS.
constexpr int vitpk_rowq(int T, int P)
{
  const int n = ((P + 3) / 4) * T;
  return T == 8 ? ((n & 15) == 8 ? n : n + 8) : n;
}

#define P7X_DPP_U(v, ctrl) ((uint32_t) __builtin_amdgcn_update_dpp((int) kNeg2, (int) (v), (ctrl), 0xf, 0xf, false))

// maximum over the T lanes of a group, every lane receives it.  Packed pairs in, one int out.
template <int T>
__device__ __forceinline__ int group_max(uint32_t v)
{
  v = pk_max(v, P7X_DPP_U(v, 0xB1));      // quad_perm [1,0,3,2]
  v = pk_max(v, P7X_DPP_U(v, 0x4E));      // quad_perm [2,3,0,1]
  v = pk_max(v, P7X_DPP_U(v, 0x141));     // row_half_mirror: 8 lanes
  if constexpr (T == 16) v = pk_max(v, P7X_DPP_U(v, 0x140));   // row_mirror: 16 lanes
  return max(lo_of(v), hi_of(v));
}

} // namespace

struct RowState { int xN, xB, xJ, xC; bool overflow; };

// Value of the previous stripe for every stripe of a register: lane z-1's register for lanes z >= 1; the first lane of a
// group takes (-32768, lo half of the group's last lane): stripe 0 has no predecessor, stripe T follows stripe T-1.
template <int T>
__device__ __forceinline__ uint32_t stripe_shift(uint32_t r, bool first)
{
  const uint32_t prev = P7X_DPP_U(r, 0x138);                          // wave_shr:1
  const uint32_t last = P7X_DPP_U(r, T == 8 ? 0x141 : 0x140);         // row_half_mirror / row_mirror: lane 0 <- lane T-1
  return first ? ((last << 16) | 0x8000u) : prev;
}

// One DP row for the G targets of a wavefront: reads the previous row from (mi, ii, di), writes (mo_, io_, do_).
// Register q of a lane holds the nodes q + 1 + s*P of its two stripes s = lane (low half) and lane + T (high half),
// so the k-1 neighbour of a register is the register before it and only register 0 needs values from another lane.
template <int T, int P>
__device__ __forceinline__ void vit_row(const VitPkArgs &a, const uint4 *tral, const uint4 *trbl, const uint4 *er, bool first,
                                        bool active, int xwm, const uint32_t (&mi)[P], const uint32_t (&ii)[P],
                                        const uint32_t (&di)[P], uint32_t (&mo_)[P], uint32_t (&io_)[P], uint32_t (&do_)[P],
                                        const uint32_t (&tdd)[P], RowState &rs)
{
  constexpr int PS = (P + 3) & ~3;
  const uint32_t xBv = splat16(rs.xB);
  uint32_t mp = stripe_shift<T>(mi[P - 1], first), ip = stripe_shift<T>(ii[P - 1], first), dp = stripe_shift<T>(di[P - 1], first);
  uint32_t xEv = kNeg2, dmaxv = kNeg2, dcv = kNeg2;
#pragma unroll
  for (int j4 = 0; j4 < PS; j4 += 4) {
    const uint4 e4 = er[(j4 / 4) * T];
    const uint32_t ev[4] = { e4.x, e4.y, e4.z, e4.w };
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int q = j4 + jj;
      if (q < P) {
        const uint4 ta = tral[q * T], tb = trbl[q * T];          // BM MM IM DM | MD MI II DD
        // four independent adds, then a max tree: the same values as the serial chain, half its depth
        const uint32_t a0 = pk_adds(xBv, ta.x), a1 = pk_adds(mp, ta.y), a2 = pk_adds(ip, ta.z), a3 = pk_adds(dp, ta.w);
        uint32_t sv = pk_max(pk_max(a0, a1), pk_max(a2, a3));
        sv = pk_adds(sv, ev[jj]);
        xEv = pk_max(xEv, sv);
        mo_[q] = sv;
        if (q > 0) do_[q] = dcv;                                 // M(i,k-1) -> D(i,k), D->D follows in the closure
        dcv = pk_adds(sv, tb.x);
        dmaxv = pk_max(dmaxv, dcv);
        io_[q] = pk_max(pk_adds(mi[q], tb.y), pk_adds(ii[q], tb.z));
        mp = mi[q]; ip = ii[q]; dp = di[q];
      }
    }
    // keep the LDS loads of later registers behind this point: hoisting all 2P transition loads costs 8 VGPRs each
    asm volatile("" ::: "memory");
  }
  do_[0] = stripe_shift<T>(dcv, first);                          // the first node of a stripe follows the last of the one before
  const int xE = group_max<T>(xEv);
  const int Dmax = group_max<T>(dmaxv);
  if (active) {
    if (xE >= 32767) rs.overflow = true;
    rs.xC = max(rs.xC, xE + a.xw_e);               // xw[C][LOOP] = xw[J][LOOP] = xw[N][LOOP] = 0
    rs.xJ = max(rs.xJ, xE + a.xw_e);
    rs.xB = max(rs.xJ + xwm, rs.xN + xwm);
  }
  const bool trig = active && (Dmax + a.ddbound > rs.xB);          // lazy F, per target
#ifdef P7X_VITPK_NO_CLOSURE      // build-time experiment (timing only, wrong scores): what the D->D closure costs
  if (false) {
#else
  if (__any(trig)) {
#endif
    // The registers of every target of the wavefront are walked, also of those that did not ask for the closure:
    // relaxing D->D edges of such a target cannot reach the next row's M (that is what the lazy-F bound says), so its
    // score is the same with or without them.  Only the carry into the next stripe is restricted to the targets that
    // asked, so that the others cannot prolong the loop.  Passes repeat until no stripe improves (2T at most).
    int pass = 0;
    bool more;
    do {
#pragma unroll
      for (int q = 1; q < P; ++q) do_[q] = pk_max(do_[q], pk_adds(do_[q - 1], tdd[q - 1]));
      uint32_t c = stripe_shift<T>(pk_adds(do_[P - 1], tdd[P - 1]), first);
      if (!trig) c = kNeg2;
      const uint32_t d0 = pk_max(do_[0], c);
      const bool changed = d0 != do_[0];
      do_[0] = d0;
      more = __any(changed) && ++pass < 2 * T;
    } while (more);
  }
}

template <int T, int P>
__global__ void __launch_bounds__(256) vitpk_kernel(const ArgRef ref)
{
  constexpr int G = 64 / T;                // targets per wavefront
  const VitPkArgs a = load_args<VitPkArgs>(ref);
  const int nlist = a.nlist_ptr ? *a.nlist_ptr : a.nlist;
  const int nskip = a.nskip_ptr ? *a.nskip_ptr : 0;         // the longest targets (a prefix of the list) go elsewhere
  if (nskip + (int) (blockIdx.x * 4 * G) >= nlist) return;  // no target for this block: skip the table load
  constexpr int PS = (P + 3) & ~3;         // table stride per lane, in pairs
  constexpr int ROWQ = vitpk_rowq(T, P);   // uint4 per emission row (see vitpk_rowq)
  // LDS layouts are lane-minor, so that the 16-byte reads of the T lanes of a group (and of the groups of a
  // ds_read_b128 service group) fall on distinct 4-bank slots:
  //   tra / trb [P][T] uint4   (BM MM IM DM) / (MD MI II DD) of pair j, lane s (every group reads the same address)
  //   em        [nrows][ROWQ]  uint4 q*T + s = pairs 4q..4q+3 of lane s
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *tra = reinterpret_cast<uint4 *>(smem);
  uint4 *trb = tra + P * T;
  uint4 *em = trb + P * T;
  {
    const uint4 *gt = reinterpret_cast<const uint4 *>(a.trans);
    for (int i = threadIdx.x; i < 2 * P * T; i += 256) tra[i] = gt[i];
    const uint4 *ge = reinterpret_cast<const uint4 *>(a.emis);
    for (int i = threadIdx.x; i < a.nrows * ROWQ; i += 256) em[i] = ge[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int s = lane % T, g = lane / T;
  const bool first = (s == 0);
  const int wave0 = rfl((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int nwaves = (int) gridDim.x * 4;
  const uint4 *tral = tra + s, *trbl = trb + s, *eml = em + s;
  uint32_t tdd[P];                 // D->D transitions of this lane's nodes: row-invariant, kept in registers for the closure
#pragma unroll
  for (int q = 0; q < P; ++q) tdd[q] = trbl[q * T].w;

  for (int it0 = nskip + wave0 * G; it0 < nlist; it0 += nwaves * G) {
    const int it = it0 + g;
    const bool have = it < nlist;
    const int slot = have ? (a.list ? a.list[it] : it) : 0;
    const int L = have ? a.slot_len[slot] : 0;
    const uint8_t *sq = a.dsq + (have ? a.slot_off[slot] : 0);
    const int xwm = (int) a.xwmove_tab[L];
    const int Lmax = wave_max_i32(L);

    // Two register sets for the rows: a row reads one and writes the other, so the old M / I / D of a pair stay
    // where they are while the new ones are produced (one set would cost a register copy per pair and array per row)
    uint32_t mA[P], iA[P], dA[P], mB[P], iB[P], dB[P];
#pragma unroll
    for (int j = 0; j < P; ++j) mA[j] = iA[j] = dA[j] = kNeg2;
    RowState rs;
    rs.xN = a.base_w; rs.xB = rs.xN + xwm; rs.xJ = -32768; rs.xC = -32768; rs.overflow = false;

    for (int i0 = 0; i0 < Lmax; i0 += 4 * T) {
      // residues of the next 4T rows: lane s holds rows i0+4s .. i0+4s+3 of its target (clamped reads; rows past the
      // end of a target are masked below)
      uint32_t word = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int r = i0 + 4 * s + b;
        const uint32_t x = (r < L) ? (uint32_t) sq[r] : 0u;
        word |= x << (8 * b);
      }
      const int nrow = min(4 * T, Lmax - i0);
      auto residue = [&](int r) -> uint32_t {
        const uint32_t w = (uint32_t) __shfl((int) word, g * T + (r >> 2));
        return (w >> (8 * (r & 3))) & 0xffu;
      };
      int r = 0;
      for (; r + 1 < nrow; r += 2) {
        vit_row<T, P>(a, tral, trbl, eml + residue(r) * ROWQ, first, i0 + r < L, xwm, mA, iA, dA, mB, iB, dB, tdd, rs);
        vit_row<T, P>(a, tral, trbl, eml + residue(r + 1) * ROWQ, first, i0 + r + 1 < L, xwm, mB, iB, dB, mA, iA, dA, tdd, rs);
      }
      if (r < nrow) {     // odd row count (only the last block of a group): one more row, then back into set A
        vit_row<T, P>(a, tral, trbl, eml + residue(r) * ROWQ, first, i0 + r < L, xwm, mA, iA, dA, mB, iB, dB, tdd, rs);
#pragma unroll
        for (int j = 0; j < P; ++j) { mA[j] = mB[j]; iA[j] = iB[j]; dA[j] = dB[j]; }
      }
    }
    const bool overflow = rs.overflow; const int xC = rs.xC;
    if (have && s == 0) a.out_xC[it] = overflow ? 32767 : xC;
  }
}

// ---------------------------------------------------------------------------- host side
bool vitpk_pick(int M, int *T, int *P)
{
  static const int p8[] = { 2, 4, 6, 8, 10, 12, 14, 15, 16, 17, 18, 19, 20 };
  static const int p16[] = { 11, 12, 13, 14, 15, 16, 17, 18, 19, 20 };
  for (int p : p8) if (M <= 8 * 2 * p) { *T = 8; *P = p; return true; }
  for (int p : p16) if (M <= 16 * 2 * p) { *T = 16; *P = p; return true; }
  return false;
}

// trans: uint4 [2][P][T] = (BM MM IM DM) then (MD MI II DD) of register j of lane s, each dword (node lo, node hi)
// = nodes j + 1 + s*P and j + 1 + (s+T)*P (Farrar striping over the 2T half-lanes of a group);
// emis: uint4 [nrows][ROWQ], q*T + s = pairs 4q .. 4q+3 of lane s.  Nodes beyond M and unused pairs hold -32768.
void vitpk_build_tables(const Profile &p, int T, int P, std::vector<uint32_t> &trans, std::vector<uint32_t> &emis)
{
  const int rowq = vitpk_rowq(T, P), nrows = p.Kp + 1;
  auto pack = [](int lo, int hi) { return ((uint32_t) lo & 0xffffu) | (((uint32_t) hi & 0xffffu) << 16); };
  trans.assign((size_t) 2 * P * T * 4, kNeg2);
  emis.assign((size_t) nrows * rowq * 4, kNeg2);
  for (int s = 0; s < T; ++s)
    for (int j = 0; j < P; ++j) {
      const int k0 = s * P + j + 1, k1 = (s + T) * P + j + 1;      // stripes s (low half) and s + T (high half)
      for (int t = 0; t < NTRANS; ++t) {
        const int lo = (k0 <= p.M) ? p.tw[(size_t) t * (p.M + 1) + k0] : -32768;
        const int hi = (k1 <= p.M) ? p.tw[(size_t) t * (p.M + 1) + k1] : -32768;
        trans[((size_t) (t / 4) * P * T + (size_t) j * T + s) * 4 + (t % 4)] = pack(lo, hi);
      }
      for (int x = 0; x < p.Kp; ++x) {
        const int lo = (k0 <= p.M) ? p.rw[(size_t) x * (p.M + 1) + k0] : -32768;
        const int hi = (k1 <= p.M) ? p.rw[(size_t) x * (p.M + 1) + k1] : -32768;
        emis[((size_t) x * rowq + (size_t) (j / 4) * T + s) * 4 + (j % 4)] = pack(lo, hi);
      }
    }
}

template <int T, int P>
static int launch_pk(const ArgRun<VitPkArgs> &a, int num_cu, hipStream_t st)
{
  const size_t lds = ((size_t) 2 * P * T + (size_t) a.at(0).nrows * vitpk_rowq(T, P)) * 16;
  auto kern = vitpk_kernel<T, P>;
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
  long want = 0;
  for (int i = 0; i < a.n; ++i) want = std::max<long>(want, ((long) a.at(i).nlist + 4 * (64 / T) - 1) / (4 * (64 / T)));   // nlist bounds the list
  if (want <= 0) return P7X_OK;
  hipLaunchKernelGGL(kern, dim3(lane_grid(want, (long) num_cu * per_cu, a.n), (unsigned) a.n), dim3(256), lds, st, a.ref());
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

int vitpk_launch(int T, int P, const ArgRun<VitPkArgs> &a, int num_cu, hipStream_t st)
{
  if (a.n <= 0) return P7X_OK;
#define P7X_PK(TT, PP) if (T == TT && P == PP) return launch_pk<TT, PP>(a, num_cu, st);
  P7X_PK(8, 2) P7X_PK(8, 4) P7X_PK(8, 6) P7X_PK(8, 8) P7X_PK(8, 10) P7X_PK(8, 12) P7X_PK(8, 14) P7X_PK(8, 15) P7X_PK(8, 16)
  P7X_PK(8, 17) P7X_PK(8, 18) P7X_PK(8, 19) P7X_PK(8, 20)
  P7X_PK(16, 11) P7X_PK(16, 12) P7X_PK(16, 13) P7X_PK(16, 14) P7X_PK(16, 15) P7X_PK(16, 16) P7X_PK(16, 17) P7X_PK(16, 18)
  P7X_PK(16, 19) P7X_PK(16, 20)
#undef P7X_PK
  set_error("no packed Viterbi kernel for this model length");
  return P7X_EINVAL;
}

} // namespace p7x
