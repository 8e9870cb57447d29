// p7x_wave.hpp -- device helpers shared by the wave-per-sequence kernels (p7x_vitfwd.hip, p7x_envelope.hip):
// DPP shifts and reductions over a 64-lane wavefront, scalar-uniform work-item loading, f32 transition records.
#pragma once
#include "p7x_device.hpp"
#include "p7x_kernels.hpp"

namespace p7x {

constexpr int kWsBlock = 256;     // 4 wavefronts per workgroup, each walking its own targets

__device__ __forceinline__ int   dpp_shr1(int v, int fill)   { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int   dpp_shl1(int v, int fill)   { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float dpp_shr1f(float v, float fill) { return __builtin_bit_cast(float, dpp_shr1(__builtin_bit_cast(int, v), __builtin_bit_cast(int, fill))); }
__device__ __forceinline__ float dpp_shl1f(float v, float fill) { return __builtin_bit_cast(float, dpp_shl1(__builtin_bit_cast(int, v), __builtin_bit_cast(int, fill))); }

#define P7X_DPP_STEP_I(v, ident, ctrl, rmask) __builtin_amdgcn_update_dpp((ident), (v), (ctrl), (rmask), 0xf, false)

__device__ __forceinline__ int wave_max_i32(int v)
{
  const int id = INT_MIN;
  v = max(v, P7X_DPP_STEP_I(v, id, 0x111, 0xf));
  v = max(v, P7X_DPP_STEP_I(v, id, 0x112, 0xf));
  v = max(v, P7X_DPP_STEP_I(v, id, 0x114, 0xf));
  v = max(v, P7X_DPP_STEP_I(v, id, 0x118, 0xf));
  v = max(v, P7X_DPP_STEP_I(v, id, 0x142, 0xa));
  v = max(v, P7X_DPP_STEP_I(v, id, 0x143, 0xc));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
  v = v + P7X_DPP_STEP_I(v, 0, 0x111, 0xf);
  v = v + P7X_DPP_STEP_I(v, 0, 0x112, 0xf);
  v = v + P7X_DPP_STEP_I(v, 0, 0x114, 0xf);
  v = v + P7X_DPP_STEP_I(v, 0, 0x118, 0xf);
  v = v + P7X_DPP_STEP_I(v, 0, 0x142, 0xa);
  v = v + P7X_DPP_STEP_I(v, 0, 0x143, 0xc);
  return __builtin_amdgcn_readlane(v, 63);
}
#define P7X_DPP_STEP_F(v, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ float wave_sum_f32(float v)
{
  v = v + P7X_DPP_STEP_F(v, 0x111, 0xf);
  v = v + P7X_DPP_STEP_F(v, 0x112, 0xf);
  v = v + P7X_DPP_STEP_F(v, 0x114, 0xf);
  v = v + P7X_DPP_STEP_F(v, 0x118, 0xf);
  v = v + P7X_DPP_STEP_F(v, 0x142, 0xa);
  v = v + P7X_DPP_STEP_F(v, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- scans of the delete-state chain across the 64 lanes.  Lane l holds the affine map x -> a_l + p_l * x of its own
// nodes; the inclusive scan composes the maps of lanes 0..l (up) or l..63 (down).  DPP moves inside the four 16-lane
// rows (Kogge-Stone with row_shr / row_shl: a lane whose source falls outside its row keeps the identity), then the
// row totals: row_bcast:15 / row_bcast:31 upwards; v_readlane of the first lane of the next row(s) downwards.  A DPP
// step is one VALU issue; the ds_bpermute it replaces (__shfl_up) is an LDS round trip on the row's critical path.
#define P7X_DPPF(v, old, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float) (old)), __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))
#define P7X_AFFINE_STEP(ctrl, rmask) { const float pa_ = P7X_DPPF(sa, 0.0f, ctrl, rmask), pp_ = P7X_DPPF(sp, 1.0f, ctrl, rmask); sa = sa + pa_ * sp; sp = sp * pp_; }
__device__ __forceinline__ float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

__device__ __forceinline__ void affine_scan_up(float &sa, float &sp)
{
  P7X_AFFINE_STEP(0x111, 0xf) P7X_AFFINE_STEP(0x112, 0xf) P7X_AFFINE_STEP(0x114, 0xf) P7X_AFFINE_STEP(0x118, 0xf)
  P7X_AFFINE_STEP(0x142, 0xa) P7X_AFFINE_STEP(0x143, 0xc)
}
__device__ __forceinline__ void affine_scan_down(float &sa, float &sp, int lane)
{
  P7X_AFFINE_STEP(0x101, 0xf) P7X_AFFINE_STEP(0x102, 0xf) P7X_AFFINE_STEP(0x104, 0xf) P7X_AFFINE_STEP(0x108, 0xf)
  const int row = lane >> 4;
  {   // rows 2 <- 3 and 0 <- 1
    const float a48 = readlane_f(sa, 48), p48 = readlane_f(sp, 48), a16 = readlane_f(sa, 16), p16 = readlane_f(sp, 16);
    const float pa_ = (row == 2) ? a48 : ((row == 0) ? a16 : 0.0f), pp_ = (row == 2) ? p48 : ((row == 0) ? p16 : 1.0f);
    sa = sa + pa_ * sp; sp = sp * pp_;
  }
  {   // rows 0, 1 <- rows 2-3
    const float a32 = readlane_f(sa, 32), p32 = readlane_f(sp, 32);
    const float pa_ = (row < 2) ? a32 : 0.0f, pp_ = (row < 2) ? p32 : 1.0f;
    sa = sa + pa_ * sp; sp = sp * pp_;
  }
}
#undef P7X_AFFINE_STEP
// the same upwards for the optimal-accuracy delete chain: x -> max(a, open ? x : -inf), open = every D->D of the lane is open
__device__ __forceinline__ void gated_max_scan_up(float &sa, int &sp)
{
#define P7X_GMAX_STEP(ctrl, rmask) { const float pa_ = P7X_DPPF(sa, -__builtin_inff(), ctrl, rmask);                                  \
    const int pp_ = __builtin_amdgcn_update_dpp(1, sp, (ctrl), (rmask), 0xf, false);                                                  \
    const float cand_ = sp ? pa_ : -__builtin_inff(); sa = sa > cand_ ? sa : cand_; sp = sp & pp_; }
  P7X_GMAX_STEP(0x111, 0xf) P7X_GMAX_STEP(0x112, 0xf) P7X_GMAX_STEP(0x114, 0xf) P7X_GMAX_STEP(0x118, 0xf)
  P7X_GMAX_STEP(0x142, 0xa) P7X_GMAX_STEP(0x143, 0xc)
#undef P7X_GMAX_STEP
}

__device__ __forceinline__ short adds16(short a, short b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ short max16(short a, short b) { return a > b ? a : b; }
__device__ __forceinline__ short lo16(uint32_t w) { return (short) (w & 0xffffu); }
__device__ __forceinline__ short hi16(uint32_t w) { return (short) (w >> 16); }

// Work distribution: wave w of the grid takes items w, w + nwaves, ...  Everything that steers control flow
// (item index, slot, length, residue pointer) is forced into SGPRs with readfirstlane so that the row loops are
// plain scalar loops: the DPP / readlane steps below must never run under a partial EXEC mask.
struct Item { int slot, L; const uint8_t *sq; };

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ Item load_item(const WaveSeqArgs &a, int it)
{
  Item o;
  o.slot = rfl(a.list ? a.list[it] : it);
  o.L = rfl(a.slot_len[o.slot]);
  const unsigned long long off = (unsigned long long) a.slot_off[o.slot];
  const unsigned lo = (unsigned) rfl((int) (unsigned) off), hi = (unsigned) rfl((int) (unsigned) (off >> 32));
  o.sq = a.dsq + (((unsigned long long) hi << 32) | lo);
  return o;
}

#define P7X_WAVE_ITEMS(it)                                                                                   \
  const int wave0_ = rfl((int) (blockIdx.x * (kWsBlock / 64) + (threadIdx.x >> 6)));                        \
  const int nwaves_ = (int) (gridDim.x * (kWsBlock / 64));                                                  \
  for (int it = wave0_; it < nlist; it += nwaves_)

// Loops over the C nodes of a lane are unrolled (the row state then lives in registers) up to 32 nodes per lane; the
// long-model instantiations beyond that (M > 2048) keep them rolled: the row state goes to scratch memory, the code
// stays small, and such models -- a handful in any profile library -- run at a fraction of the speed, but they run.
constexpr int unroll_c(int C) { return C <= 32 ? C : 1; }

struct F8 { float bm, mm, im, dm, md, mi, ii, dd; };
__device__ __forceinline__ F8 load_f8(const float4 *t, int idx)
{
  const float4 a = t[2 * idx], b = t[2 * idx + 1];
  return F8{ a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
}
// The transitions of a node are two float4 (BM MM IM DM | MD MI II DD).  In global memory the two lie side by side
// ([2 idx], [2 idx + 1]).  Read like that from LDS by the 64 lanes of a wavefront (idx = c * 64 + lane), the 16 lanes a
// ds_read_b128 is served for at a time touch only every second 16-byte slot of the bank row -- a two-way conflict on
// every such read (the envelope kernel re-reads them in every row: 2.8 conflict cycles per LDS instruction, VERDICT r03
// weak #3).  An LDS copy therefore keeps the halves in two planes ([idx], [plane + idx]): consecutive lanes, consecutive
// slots.  PLANES = false: the interleaved image where it lies in global memory.
template <bool PLANES>
struct TransView {
  const float4 *p; int plane;
  __device__ __forceinline__ F8 at(int idx) const
  {
    const float4 a = PLANES ? p[idx] : p[2 * idx], b = PLANES ? p[plane + idx] : p[2 * idx + 1];
    return F8{ a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
  }
  __device__ __forceinline__ float dd(int idx) const { return (PLANES ? p[plane + idx] : p[2 * idx + 1]).w; }
};

__device__ __forceinline__ float wave_max_f32(float v)
{
  const int id = __builtin_bit_cast(int, -__builtin_inff());
#define P7X_DPP_STEP_FM(v, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(id, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))
  v = fmaxf(v, P7X_DPP_STEP_FM(v, 0x111, 0xf));
  v = fmaxf(v, P7X_DPP_STEP_FM(v, 0x112, 0xf));
  v = fmaxf(v, P7X_DPP_STEP_FM(v, 0x114, 0xf));
  v = fmaxf(v, P7X_DPP_STEP_FM(v, 0x118, 0xf));
  v = fmaxf(v, P7X_DPP_STEP_FM(v, 0x142, 0xa));
  v = fmaxf(v, P7X_DPP_STEP_FM(v, 0x143, 0xc));
#undef P7X_DPP_STEP_FM
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

} // namespace p7x
