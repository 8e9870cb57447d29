// p7x_envelope.hip -- rescoring of one domain envelope on CDNA4: one envelope per WAVEFRONT, all four steps of
// upstream's rescore_isolated_domain() (p7_domaindef.c) fused into one kernel:
//
//   1. Forward over the envelope, unihit, full target length model     (impl_sse/fwdback.c  p7_Forward)
//   2. Backward, re-using Forward's scale factors                       (impl_sse/fwdback.c  p7_Backward)
//   3. posterior decoding + null2 expectation + optimal-accuracy DP     (decoding.c, null2.c, optacc.c)
//   4. optimal-accuracy traceback                                       (optacc.c  p7_OATrace)
//
// Layout is the one of the parsers in p7x_vitfwd.hip: lane z owns nodes z*C+1 .. z*C+C, device tables are
// [c*64 + lane].  Step 1 keeps only the envelope score and the per-row scale factors; step 2 parks Backward's M and I
// rows in a per-wavefront HBM workspace (D is not needed: posterior decoding leaves D at zero); step 3 streams them back
// once, row by row, while it runs Forward AGAIN next to them (same code as step 1: bit-identical values), keeps the OA
// row in registers and writes 16 bits per cell: the back-pointers and the posterior digits of the M and I cell (all the
// alignment display needs of the posteriors); step 4 is a serial walk over those by lane 0, followed by a lane-parallel
// pass that attaches the posterior digit of every emitted residue.  8 + 8 + 2 bytes of HBM traffic per cell (round 1:
// 16 + 16 + 1); lanes whose nodes are all padding move nothing.
// The host (p7x_domaindef.cpp) turns the trace into the alignment display and applies the null2 correction.
#include <cstdlib>
#include <cstdio>
#include <cstdlib>
#include "p7x_wave.hpp"
#include "p7x_envfwd.hpp"

namespace p7x {

namespace {



constexpr float kNegInf = -__builtin_inff();

// p7T_* state codes (p7_trace.pxd), as the host uses them
enum { tM = 1, tD = 2, tI = 3, tS = 4, tN = 5, tB = 6, tE = 7, tC = 8, tT = 9, tJ = 10 };

__device__ __forceinline__ float gate(float t, float v) { return t > 0.0f ? v : 0.0f; }          // and(cmpgt(t, 0), v)
__device__ __forceinline__ float block(float t, float v) { return t > 0.0f ? v : kNegInf; }      // traceback: t == 0 ? -inf : v
__device__ __forceinline__ float vmax(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float rflf(float v) { return __builtin_bit_cast(float, rfl(__builtin_bit_cast(int, v))); }

__device__ __forceinline__ void phase_fence()
{ // Rows written by this wavefront are read back by it (possibly by other lanes, and the workspace is re-used for the
  // next envelope).  Producer and consumer are the same wavefront, so work-group scope is all that is needed: the
  // stores have left the wavefront (vmcnt(0)) and the CU's vector cache is coherent for its own stores.  Agent scope
  // would write back and invalidate the XCD's whole L2 (buffer_wbl2 / buffer_inv sc1) four times per envelope and
  // wavefront -- taking the lines of every other wavefront and of the filter kernels running beside this one with it.
  // (This relies on the wavefront's producer and consumer lanes sharing one CU's vector cache: the kernels of this file must
  // not be built for tgsplit mode, where a work-group may straddle CUs.)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

} // namespace

#ifdef P7X_ENV_PROFILE
// build-time experiment (-DP7X_ENV_PROFILE): core-clock cycles every wavefront spent in the phases of env_kernel, summed
// over the wavefronts of all launches since the last read: [0..3] phases 1-4, [4] rows, [5] envelopes
__device__ unsigned long long g_env_prof[8];
#define P7X_ENV_STAMP(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_env_prof[slot], now_ - stamp_); stamp_ = now_; } while (0)
#else
#define P7X_ENV_STAMP(slot) do { } while (0)
#endif

// One block per CU: its wavefronts (env_waves(C): 8, or 4 for models of more than 448 nodes) share one copy of the
// profile tables in LDS and each walks its own envelopes.
// posterior probability -> the digit of the alignment's posterior line, exactly as the host prints it
// (p7_alidisplay: (p + 0.05 >= 1.0) ? '*' : '0' + (int) ((p + 0.05) * 10.0), in double): 0..9, 10 = '*'
__device__ __forceinline__ unsigned pp_code(float p)
{
  const double v = (double) p + 0.05;
  return v >= 1.0 ? 10u : (unsigned) (int) (v * 10.0);
}
// Near-tie guard (status bit 6).  The optimal-accuracy recursion sums posteriors that differ from the host twin's by a
// few units in the last place (another summation order in Forward / Backward), so a traceback choice between two
// candidates that lie within a few ulps of each other -- or a posterior within that distance of the next printed digit
// -- can fall the other way than in the reference's order of operations.  Every such choice ON THE TRACE flags the
// envelope, and the host stage repeats flagged envelopes with the host twin (domaindef_finish_deferred), which performs
// the reference's operations in the reference's order.  guard: relative half-width (cfg.oa_guard); 0 switches it off.
__device__ __forceinline__ float guard_band(float v, float guard) { return __builtin_fabsf(v) * guard + guard; }
__device__ __forceinline__ int near_tie(float x, float y, float guard)
{ // the band is the winner's: one candidate at -inf (a closed transition, the first row) is an infinite distance away;
  // both at -inf: the difference is NaN and the test is false (such a cell is unreachable anyway)
  return (__builtin_fabsf(x - y) <= guard_band(vmax(x, y), guard)) ? 1 : 0;
}
__device__ __forceinline__ int pp_near(float p, float guard)
{
  // float is enough here: the band is an order of magnitude wider than the rounding of this expression
  const float v = (p + 0.05f) * 10.0f;
  return (__builtin_fabsf(v - __builtin_rintf(v)) < 4.0f * guard && v > 0.75f) ? 1 : 0;
}
// The same two tests as they run inside the decoding row (every cell of every row): the winner <hi> is known there and
// optimal-accuracy values are sums of probabilities (>= 0, or -inf where nothing leads), so  hi - lo <= hi g + g  is
// lo >= fma(hi, 1 - g, -g): one fused multiply-add and one comparison.  (An unreachable cell, hi = -inf, tests true; no
// trace passes through one.)
__device__ __forceinline__ int near_below(float hi, float lo, float g1, float g) { return lo >= __builtin_fmaf(hi, g1, -g) ? 1 : 0; }
// ... and the printed digit with its distance from the next one, in float under the guard: v = 10 p + 0.5 is off by an ulp
// or two of what the host computes in double, the band of 4 guards on either side of a digit boundary is ten times wider
__device__ __forceinline__ unsigned pp_code_guarded(float p, float band, int &near)
{
  const float v = __builtin_fmaf(p, 10.0f, 0.5f);
  const float f = v - __builtin_floorf(v);
  near |= (__builtin_fabsf(f - 0.5f) > band) ? 1 : 0;             // band = 0.5 - 4 guard
  const int d = (int) v;
  return (unsigned) (d > 10 ? 10 : d);
}
// and back to a float that prints as that digit (the host stage formats the line from floats)
__device__ __forceinline__ float pp_from_code(unsigned code) { return code >= 10u ? 1.0f : (float) (((double) code + 0.5) / 10.0 - 0.05); }

// G: with the near-tie guard (a.oa_guard > 0).  Without it the kernel carries none of the guard's arithmetic.
// LT: a long-target (nhmmer) envelope -- upstream rescore_isolated_domain(long_target = TRUE): the match odds come from a
// table of the ENVELOPE's own (re-derived by the host for the background mixed with the envelope's composition,
// a.env_emis; the length model is the envelope's own length through env_L), and Forward runs once more with the profile's
// unmodified odds: that score is the envelope's, the difference the bias (a.out_orig).
template <int C, bool G, bool LT = false>
__global__ void __launch_bounds__(env_waves(C) * 64, env_waves(C) / 4) env_kernel(const ArgRef ref)
{
  constexpr int kEnvBlock = env_waves(C) * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Mpad = 64 * C;
  const EnvArgs a = load_args<EnvArgs>(ref);
  if ((int) blockIdx.x >= a.nblocks) return;        // this job has fewer blocks than the widest job of the launch
  constexpr bool TG = C > 64;         // M > 4096: the transitions are read through L2 as well
  // transitions: from LDS in two planes (see TransView), or where they lie
  const TransView<!TG> tr{ TG ? reinterpret_cast<const float4 *>(a.trans) : reinterpret_cast<const float4 *>(smem), Mpad };
  // emission odds [nrows][Mpad]: staged in LDS while they fit beside the transitions (M <= 1024), else read where they
  // lie (one coalesced 256-byte row segment per chunk and row: L2-resident, like the parsers' long-model variant)
  constexpr bool kEmisInLds = C <= 16 && !LT;
  const float *em_profile = kEmisInLds ? reinterpret_cast<const float *>(smem + (size_t) Mpad * 32) : reinterpret_cast<const float *>(a.emis);
  {
    if constexpr (!TG) {
      const float4 *gt = reinterpret_cast<const float4 *>(a.trans);
      float4 *lt = reinterpret_cast<float4 *>(smem);
      for (int i = threadIdx.x; i < 2 * Mpad; i += kEnvBlock) lt[(i & 1) * Mpad + (i >> 1)] = gt[i];
    }
    if constexpr (kEmisInLds) {
      const float4 *ge = reinterpret_cast<const float4 *>(a.emis);
      float4 *le = reinterpret_cast<float4 *>(smem + (size_t) Mpad * 32);
      for (int i = threadIdx.x; i < a.nrows * Mpad / 4; i += kEnvBlock) le[i] = ge[i];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int nlist = a.nenv;
  const int wave_in_job = rfl((int) (blockIdx.x * (kEnvBlock / 64) + (threadIdx.x >> 6)));
  const int wave_id = a.slab_base + wave_in_job;     // per-wavefront workspace slabs are numbered across the jobs of a launch

  // per-wavefront workspace
  float *wsf = a.work + (size_t) wave_id * (size_t) a.work_stride;
  const size_t rows = (size_t) a.Lmax + 1;
  float *bM = wsf, *bI = bM + rows * Mpad;  // Backward's M and I rows (Forward's are recomputed in phase 3)
  float *fx = bI + rows * Mpad;             // [rows][6]  E N J B C SCALE
  float *bx = fx + rows * 6;                // [rows][6]
  float *ox = bx + rows * 6;                // [rows][5]  OA specials E N J B C
  float *px = ox + rows * 5;                // [rows][3]  posterior N J C
  float *totr_row = px + rows * 3;          // [rows]
  unsigned short *bp = reinterpret_cast<unsigned short *>(totr_row + rows); // [rows][Mpad] back-pointers (bits 0-3) and the posterior
                                                                            // digits of the M (4-7) and I (8-11) cells
  const bool lane_live = lane * C < a.M;    // lanes whose nodes are all padding neither store nor load rows
  int erank[C];                             // rank of this lane's nodes in the striped visiting order of select_e (q outer, stripe inner)
  {
    const int Qe = max(2, (a.M - 1) / 4 + 1);
#pragma unroll unroll_env(C)
    for (int c = 0; c < C; ++c) { const int k = lane * C + c; erank[c] = (k % Qe) * 4 + k / Qe; }
  }

  for (;;) {
    // envelopes are taken longest first from the job's queue: a wavefront that drew a short one comes back for more
    int q = 0;
    if (lane == 0) q = atomicAdd(a.cursor, 1);
    q = rfl(q);
    if (q >= nlist) break;
    const int it = rfl(a.order[q]);
    const int Ld = rfl(a.env_len[it]);
    const int Lfull = rfl(a.env_L[it]);
    const unsigned long long off = (unsigned long long) a.env_sq[it];
    const unsigned olo = (unsigned) rfl((int) (unsigned) off), ohi = (unsigned) rfl((int) (unsigned) (off >> 32));
    const uint8_t *sq = a.dsq + (((unsigned long long) ohi << 32) | olo);      // sq[0] = first residue of the envelope
    const float pmove = (2.0f + a.nj) / ((float) Lfull + 2.0f + a.nj), ploop = 1.0f - pmove;
    int status = 0;
    const float *em = LT ? a.env_emis + (size_t) it * (size_t) a.env_emis_stride : em_profile;      // [nrows][Mpad]

#ifdef P7X_ENV_PROFILE
    unsigned long long stamp_ = __builtin_readcyclecounter();
    if (lane == 0) { atomicAdd(&g_env_prof[4], (unsigned long long) Ld); atomicAdd(&g_env_prof[5], 1ull); }
#endif
    // ------------------------------------------------------------------ 1. Forward (score and scale factors)
    float envsc;
    {
      EnvForward<C> f;
      f.init(tr, lane, pmove);
      if (lane == 0) { fx[0] = 0.0f; fx[1] = 1.0f; fx[2] = 0.0f; fx[3] = f.xB; fx[4] = 0.0f; fx[5] = 1.0f; }
      for (int i0 = 0; i0 < Ld; i0 += 64) {
        const int nrow = min(64, Ld - i0);
        const uint32_t resid = (lane < nrow) ? sq[i0 + lane] : 0;
        for (int r = 0; r < nrow; ++r) {
          const int i = i0 + r;
          f.row(tr, em, Mpad, lane, __builtin_amdgcn_readlane((int) resid, r), pmove, ploop, a.xf_e_move, a.xf_e_loop);
          if (lane == 0) {
            float *row = fx + (size_t) (i + 1) * 6;
            row[0] = f.xE; row[1] = f.xN; row[2] = f.xJ; row[3] = f.xB; row[4] = f.xC; row[5] = f.scale;
          }
        }
      }
      if (f.xC != f.xC || (Ld > 0 && f.xC == 0.0f) || __builtin_isinf(f.xC)) { envsc = __builtin_inff(); status |= 1; }
      else envsc = (float) ((double) f.totscale + log((double) (f.xC * pmove)));
    }
    if constexpr (LT) {            // Forward with the profile's own odds: the envelope's score proper.  (Run row by row next to
                                   // the first recurrence it was slower: 66 against 58 ms for the benchmark's two rounds.)
      EnvForward<C> g;
      g.init(tr, lane, pmove);
      for (int i0 = 0; i0 < Ld; i0 += 64) {
        const int nrow = min(64, Ld - i0);
        const uint32_t resid = (lane < nrow) ? sq[i0 + lane] : 0;
        for (int r = 0; r < nrow; ++r)
          g.row(tr, em_profile, Mpad, lane, __builtin_amdgcn_readlane((int) resid, r), pmove, ploop, a.xf_e_move, a.xf_e_loop);
      }
      float orig;
      if (g.xC != g.xC || (Ld > 0 && g.xC == 0.0f) || __builtin_isinf(g.xC)) { orig = __builtin_inff(); status |= 1; }
      else orig = (float) ((double) g.totscale + log((double) (g.xC * pmove)));
      if (lane == 0) a.out_orig[it] = orig;
    }
    phase_fence();
    P7X_ENV_STAMP(0);

    // ------------------------------------------------------------------ 2. Backward
    bool own_scales = false;
    float bck_xN0;
    {
      float t_md[C], t_dd[C], t_mi[C], t_ii[C], t_bm[C], n_mm[C], n_im[C], n_dm[C];
      float ddprod = 1.0f;
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) {
        const F8 t = tr.at(c * 64 + lane);
        t_md[c] = t.md; t_dd[c] = t.dd; t_mi[c] = t.mi; t_ii[c] = t.ii; t_bm[c] = t.bm;
        ddprod *= t.dd;
      }
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) {          // transitions entering the NEXT node
        float mmn, imn, dmn;
        if (c + 1 < C) { const F8 t = tr.at((c + 1) * 64 + lane); mmn = t.mm; imn = t.im; dmn = t.dm; }
        else { const F8 t = tr.at(lane); mmn = dpp_shl1f(t.mm, 0.0f); imn = dpp_shl1f(t.im, 0.0f); dmn = dpp_shl1f(t.dm, 0.0f); }
        n_mm[c] = mmn; n_im[c] = imn; n_dm[c] = dmn;
      }
      float mm[C], im[C], dm[C];
      float xJ = 0.0f, xB = 0.0f, xN = 0.0f;
      float xC = pmove;
      float xE = xC * a.xf_e_move;
      auto d_chain = [&](float (&d)[C]) {
        float A = 0.0f;
#pragma unroll unroll_env(C)
        for (int c = C - 1; c >= 0; --c) { A = d[c] + A * t_dd[c]; }
        float sa = A, sp = ddprod;
        affine_scan_down(sa, sp, lane);
        float w = dpp_shl1f(sa, 0.0f);
#pragma unroll unroll_env(C)
        for (int c = C - 1; c >= 0; --c) { d[c] = d[c] + w * t_dd[c]; w = d[c]; }
      };
      auto store_row = [&](int r) {
        float *rm = bM + (size_t) r * Mpad + lane, *ri = bI + (size_t) r * Mpad + lane;
        if (lane_live) {
#pragma unroll unroll_env(C)
          for (int c = 0; c < C; ++c) { rm[c * 64] = mm[c]; ri[c * 64] = im[c]; }
        }
      };
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) { mm[c] = xE; dm[c] = xE; im[c] = 0.0f; }
      d_chain(dm);
      {
        float dn = dpp_shl1f(dm[0], 0.0f);
#pragma unroll unroll_env(C)
        for (int c = C - 1; c >= 0; --c) { mm[c] = mm[c] + dn * t_md[c]; dn = dm[c]; }
      }
      float sc = rflf(fx[(size_t) Ld * 6 + 5]);
      if (sc > 1.0f) {
        xE = xE / sc; xN = xN / sc; xC = xC / sc; xJ = xJ / sc; xB = xB / sc;
        const float inv = (float) (1.0 / (double) sc);
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) { mm[c] *= inv; dm[c] *= inv; im[c] *= inv; }
      }
      store_row(Ld);
      if (lane == 0) { float *r = bx + (size_t) Ld * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = sc; }

      // Residue x_{i+1} and Forward's scale factor of row i: 64 rows at a time, one per lane, so that the row loop itself
      // has stores only.  (A load inside the row loop is waited for with vmcnt(0) -- the counter retires in order -- and
      // that wait also covers every row store issued before it: one HBM round trip per row.)
      for (int ib = Ld - 1; ib >= 1; ib -= 64) {
      const int nblk = min(64, ib);
      uint32_t res_b = 0; float fsc_b = 0.0f;
      if (lane < nblk) { res_b = sq[ib - lane]; fsc_b = fx[(size_t) (ib - lane) * 6 + 5]; }
      for (int l = 0; l < nblk; ++l) {
        const int i = ib - l;
        const int x = __builtin_amdgcn_readlane((int) res_b, l);
        const float fsc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fsc_b), l));
        const float *er = em + x * Mpad + lane;
        float me[C];
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) me[c] = mm[c] * er[c * 64];
        float bsum = 0.0f;
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) bsum = bsum + me[c] * t_bm[c];
        const float me_next0 = dpp_shl1f(me[0], 0.0f);
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) {
          const float mp = (c + 1 < C) ? me[c + 1] : me_next0;
          const float ipv = im[c];
          im[c] = ipv * t_ii[c] + mp * n_im[c];
          dm[c] = mp * n_dm[c];
          mm[c] = ipv * t_mi[c] + mp * n_mm[c];
        }
        xB = wave_sum_f32(bsum);
        xC = xC * ploop;
        xJ = (xB * pmove) + (xJ * ploop);
        xN = (xB * pmove) + (xN * ploop);
        xE = (xC * a.xf_e_move) + (xJ * a.xf_e_loop);
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) { dm[c] = dm[c] + xE; mm[c] = mm[c] + xE; }
        d_chain(dm);
        {
          float dn = dpp_shl1f(dm[0], 0.0f);
#pragma unroll unroll_env(C)
          for (int c = C - 1; c >= 0; --c) { mm[c] = mm[c] + dn * t_md[c]; dn = dm[c]; }
        }
        if (xB > 1.0e16f) own_scales = true;
        sc = own_scales ? ((xB > 1.0e4f) ? xB : 1.0f) : fsc;
        if (sc > 1.0f) {
          xE /= sc; xN /= sc; xJ /= sc; xB /= sc; xC /= sc;
          const float inv = (float) (1.0 / (double) sc);
#pragma unroll unroll_env(C)
          for (int c = 0; c < C; ++c) { mm[c] *= inv; dm[c] *= inv; im[c] *= inv; }
        }
        store_row(i);
        if (lane == 0) { float *r = bx + (size_t) i * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = sc; }
      }
      }
      {
        const int x = rfl((int) sq[0]);
        const float *er = em + x * Mpad + lane;
        float bsum = 0.0f;
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) bsum = bsum + (mm[c] * er[c * 64]) * t_bm[c];
        xB = wave_sum_f32(bsum);
        xN = (xB * pmove) + (xN * ploop);
        bck_xN0 = xN;
      }
    }
    phase_fence();
    P7X_ENV_STAMP(1);

    // ------------------------------------------------------------------ 3. decoding, null2 sums, optimal accuracy
    float oasc;
    int e_row = -1, e_k = 0, e_s = 0, e_near = 0, c_near_row = -1;
    {
      float scaleproduct = (float) (1.0 / (double) bck_xN0);
      bool ddpass = true;                                            // every D->D transition of this lane is open
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) ddpass = ddpass && (tr.dd(c * 64 + lane) > 0.0f);
      float p_md0, p_dd0;                                            // leaving transitions of the previous lane's last node
      { const F8 t = tr.at((C - 1) * 64 + lane); p_md0 = dpp_shr1f(t.md, 0.0f); p_dd0 = dpp_shr1f(t.dd, 0.0f); }
      float om_[C], oi_[C], od_[C], msum[C], isum[C];
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) { om_[c] = oi_[c] = od_[c] = kNegInf; msum[c] = isum[c] = 0.0f; }
      float oE = kNegInf, oN = 0.0f, oJ = kNegInf, oB = 0.0f, oC = kNegInf;
      float eN = 0.0f, eJ = 0.0f, eC = 0.0f;
      const int Q = max(2, (a.M - 1) / 4 + 1);                         // p7O_NQF(M): the striped visiting order of select_e
      const float g1 = 1.0f - a.oa_guard, ppband = 0.5f - __builtin_fmaxf(4.0f * a.oa_guard, 2.0e-6f);     // (2e-6: what float costs the digit)
      const bool loopJ = ploop != 0.0f, loopE = a.xf_e_loop != 0.0f, moveE = a.xf_e_move != 0.0f, moveNJ = pmove != 0.0f;
      // Row r+1 is fetched while row r is processed: four vector rows and the twelve special-state values (one load,
      // lane l < 6 takes Forward's, lane 8 + l Backward's), so that no memory round trip sits on the row's critical path.
      float nbm[C], nbi[C];
      auto fetch_row = [&](int r, float (&c2)[C], float (&d)[C]) {
        const float *rbm = bM + (size_t) r * Mpad + lane, *rbi = bI + (size_t) r * Mpad + lane;
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) { c2[c] = rbm[c * 64]; d[c] = rbi[c * 64]; }      // every lane loads (its own columns: valid
      };                                                                                 // memory); dead lanes are zeroed at the use
      EnvForward<C> f;                 // Forward again, row by row, next to the decoding
      f.init(tr, lane, pmove);
      auto fetch_x = [&](int r) -> float {       // unconditional: lanes past 16 repeat the pattern, nobody reads them
        const int l = min(lane & 7, 5);
        const float *src = (lane & 8) ? bx + (size_t) r * 6 : fx + (size_t) r * 6;
        return src[l];
      };
      auto xval = [&](float v, int idx) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), idx)); };
      fetch_row(1, nbm, nbi);
      float xprev = fetch_x(0), xcur = fetch_x(1);
      // The row loop has no conditional load and none that is consumed in the iteration that issues it: the in-order
      // vmcnt counter then lets the wait for row r's values leave the loads of row r + 1 and the stores in flight.  The
      // residues come 64 rows at a time in the outer loop.
      for (int r0 = 1; r0 <= Ld; r0 += 64) {
      const int nblk = min(64, Ld - r0 + 1);
      const uint32_t resid3 = (lane < nblk) ? sq[(r0 - 1) + lane] : 0;
      for (int l = 0; l < nblk; ++l) {
        const int r = r0 + l;
        float cbm[C], cbi[C];
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) { cbm[c] = lane_live ? nbm[c] : 0.0f; cbi[c] = lane_live ? nbi[c] : 0.0f; }
        const float xthis = xcur;
        const int rn = (r < Ld) ? r + 1 : r;                 // the last iteration re-reads its own row (harmless)
        fetch_row(rn, nbm, nbi);
        xcur = fetch_x(rn);
        f.row(tr, em, Mpad, lane, __builtin_amdgcn_readlane((int) resid3, l), pmove, ploop, a.xf_e_move, a.xf_e_loop);
        const float (&cfm)[C] = f.mm;
        const float (&cfi)[C] = f.im;
        const float fS = xval(xthis, 5), bS = xval(xthis, 8 + 5);
        const float totr = scaleproduct * fS;
        float ppm[C], ppi[C];
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) {
          ppm[c] = (cfm[c] * cbm[c]) * totr;
          ppi[c] = (cfi[c] * cbi[c]) * totr;
          msum[c] = ppm[c] + msum[c];
          isum[c] = ppi[c] + isum[c];
        }
        const float ppN = xval(xprev, 1) * xval(xthis, 8 + 1) * ploop * scaleproduct;
        const float ppJ = xval(xprev, 2) * xval(xthis, 8 + 2) * ploop * scaleproduct;
        const float ppC = xval(xprev, 4) * xval(xthis, 8 + 4) * ploop * scaleproduct;
        xprev = xthis;
        eN += ppN; eJ += ppJ; eC += ppC;
        if (own_scales) scaleproduct *= fS / bS;

        // OA row.  DP values use gate() (0 when a transition is closed); the traceback rule uses -inf (block()).
        const float xBp = oB;
        float mp = dpp_shr1f(om_[C - 1], kNegInf), ip = dpp_shr1f(oi_[C - 1], kNegInf), dp = dpp_shr1f(od_[C - 1], kNegInf);
        unsigned short code[C];
        float t_md[C], t_dd[C];
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) {
          const F8 t = tr.at(c * 64 + lane);
          t_md[c] = t.md; t_dd[c] = t.dd;
          float sv = gate(t.bm, xBp);
          sv = vmax(sv, gate(t.mm, mp));
          sv = vmax(sv, gate(t.im, ip));
          sv = vmax(sv, gate(t.dm, dp));
          // winner by upstream's rule (strict >, in this order) and the runner-up for the near-tie guard (near_tie() above)
          int best = 0; float bv = block(t.mm, mp), second = kNegInf;
          { const float p1 = block(t.im, ip); if (p1 > bv) { best = 1; second = bv; bv = p1; } else second = vmax(second, p1); }
          { const float p2 = block(t.dm, dp); if (p2 > bv) { best = 2; second = bv; bv = p2; } else second = vmax(second, p2); }
          { const float p3 = block(t.bm, xBp); if (p3 > bv) { best = 3; second = bv; bv = p3; } else second = vmax(second, p3); }
          int near_m = 0;
          if constexpr (G) near_m = near_below(bv, second, g1, a.oa_guard);
          const float mcur = om_[c], icur = oi_[c];
          float iv = gate(t.mi, mcur);
          iv = vmax(iv, gate(t.ii, icur));
          const float q0 = block(t.mi, mcur), q1 = block(t.ii, icur);
          const int ichoice = (q0 >= q1) ? 0 : 1;
          int near_i = 0;
          if constexpr (G) near_i = near_below(vmax(q0, q1), __builtin_fminf(q0, q1), g1, a.oa_guard);
          mp = mcur; ip = icur; dp = od_[c];
          om_[c] = sv + ppm[c];
          oi_[c] = iv + ppi[c];
          int near_pp = 0;
          unsigned cm, ci;
          // (long-target envelopes keep the digits in double: with the float form the <20, true, true> instantiation -- 256 VGPRs,
          // 420 spilled SGPRs, scratch -- faulted on the device, round 6; their posterior digits are tested against the band alone)
          if constexpr (G && !LT) { cm = pp_code_guarded(ppm[c], ppband, near_pp); ci = pp_code_guarded(ppi[c], ppband, near_pp); }
          else {
            cm = pp_code(ppm[c]); ci = pp_code(ppi[c]);
            if constexpr (G) near_pp = pp_near(ppm[c], a.oa_guard) | pp_near(ppi[c], a.oa_guard);
          }
          code[c] = (unsigned short) (best | (ichoice << 2) | (cm << 4) | (ci << 8) | (near_m << 12) | (near_i << 13) | (near_pp << 15));
        }
        // D(r,k) = max(gate(tMD(k-1), M(r,k-1)), tDD(k-1) > 0 ? D(r,k-1) : 0), D(r,1) = -inf: a segmented max-scan
        {
          float w = kNegInf;
#pragma unroll unroll_env(C)
          for (int c = 0; c < C; ++c) w = vmax(gate(t_md[c], om_[c]), t_dd[c] > 0.0f ? w : 0.0f);
          float sa = w; int sp = ddpass ? 1 : 0;
          gated_max_scan_up(sa, sp);
          w = dpp_shr1f(sa, kNegInf);
#pragma unroll unroll_env(C)
          for (int c = 0; c < C; ++c) { od_[c] = w; w = vmax(gate(t_md[c], om_[c]), t_dd[c] > 0.0f ? w : 0.0f); }
        }
        {
          float pm = dpp_shr1f(om_[C - 1], kNegInf), pd = dpp_shr1f(od_[C - 1], kNegInf);
          float pmd = p_md0, pdd = p_dd0;
#pragma unroll unroll_env(C)
          for (int c = 0; c < C; ++c) {
            const float d0 = block(pmd, pm), d1 = block(pdd, pd);
            const int dchoice = (d0 >= d1) ? 0 : 1;
            int near_d = 0;
            if constexpr (G) near_d = near_below(vmax(d0, d1), __builtin_fminf(d0, d1), g1, a.oa_guard);
            code[c] |= (unsigned short) ((dchoice << 3) | (near_d << 14));
            pm = om_[c]; pd = od_[c]; pmd = t_md[c]; pdd = t_dd[c];
          }
        }
        // Row r + 1's loads have had this row's arithmetic to arrive; waiting for them HERE, before this row's stores are
        // issued, keeps those stores out of the wait (vmcnt retires in order: a wait placed after the stores -- at the
        // next row's first use, where the compiler would put it -- covers the stores' round trip as well).
        __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0), expcnt and lgkmcnt unconstrained
        {
          unsigned short *rb = bp + (size_t) r * Mpad + lane;
          if (lane_live) {
#pragma unroll unroll_env(C)
            for (int c = 0; c < C; ++c) rb[c * 64] = code[c];
          }
        }
        float rowmax = kNegInf;
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) if (lane * C + c + 1 <= a.M) rowmax = vmax(rowmax, vmax(om_[c], od_[c]));
        oE = wave_max_f32(rowmax);
        float t1, t2;
        t1 = !loopJ ? 0.0f : oJ + ppJ;
        t2 = !loopE ? 0.0f : oE;
        oJ = fmaxf(t1, t2);
        t1 = !loopJ ? 0.0f : oC + ppC;                 // C, J and N share one loop probability
        t2 = !moveE ? 0.0f : oE;
        const int c_from_e = rfl((int) !(t1 > t2));    // what select_c will decide at this row (wave-uniform)
        if constexpr (G) { if (near_tie(t1, t2, a.oa_guard)) c_near_row = r; }     // ... and whether that decision was a close one
        oC = fmaxf(t1, t2);
        oN = !loopJ ? 0.0f : oN + ppN;
        t1 = !moveNJ ? 0.0f : oN;
        t2 = !moveNJ ? 0.0f : oJ;
        oB = fmaxf(t1, t2);
        if (c_from_e) {
          // select_e for this row, should the traceback enter E here: upstream scans the striped layout, q outer,
          // M cells with >=, D cells with >.  Net effect: the LAST M cell (striped order) that equals the row
          // maximum wins; without one, the FIRST D cell that does.
          int keyM = 0, keyD = 0, nearM = 0;
          const float ethr = oE - guard_band(oE, a.oa_guard);
#pragma unroll unroll_env(C)
          for (int c = 0; c < C; ++c) {
            const int k = lane * C + c + 1;
            if (k <= a.M) {
              const int rank = erank[c];
              if (om_[c] == oE) keyM = max(keyM, rank + 1);
              if (od_[c] == oE) keyD = max(keyD, (1 << 24) - rank);
              if constexpr (G) nearM += om_[c] >= ethr;
            }
          }
          keyM = wave_max_i32(keyM);
          // D cells copy the M cell they derive from (a structural tie, the same on any device): only a second MATCH cell
          // inside the guard band -- or an end in a delete state -- makes the choice of the end cell a near-tie
          if constexpr (G) e_near = (wave_max_i32(nearM) > 1 || __builtin_popcountll(__ballot(nearM > 0)) > 1 || keyM == 0) ? 1 : 0;
          if (keyM > 0) { const int rank = keyM - 1; e_k = (rank % 4) * Q + rank / 4 + 1; e_s = tM; }
          else {
            keyD = wave_max_i32(keyD);
            if (keyD > 0) { const int rank = (1 << 24) - keyD; e_k = (rank % 4) * Q + rank / 4 + 1; e_s = tD; }
            else { e_k = 0; e_s = -1; }
          }
          e_row = r;
        }
        if (lane == 0) {
          float *o = ox + (size_t) r * 5; o[0] = oE; o[1] = oN; o[2] = oJ; o[3] = oB; o[4] = oC;
          float *q = px + (size_t) r * 3; q[0] = ppN; q[1] = ppJ; q[2] = ppC;
        }
      }
      }
      if (lane == 0) { float *o = ox; o[0] = kNegInf; o[1] = 0.0f; o[2] = kNegInf; o[3] = 0.0f; o[4] = kNegInf; }
      oasc = oC;
      if (__builtin_isinf(scaleproduct)) status |= 2;            // p7_Decoding: eslERANGE, the envelope is dropped

      // null2 by expectation: state occupancies -> residue odds
      const float norm = (float) (1.0 / (double) (float) Ld);
      const float xfactor = (eN * norm + eC * norm) + eJ * norm;
      float *n2 = a.out_null2 + (size_t) it * 32;
      for (int x = 0; x < a.K; ++x) {
        const float *er = em + x * Mpad + lane;
        float s = 0.0f;
#pragma unroll unroll_env(C)
        for (int c = 0; c < C; ++c) { s = s + (msum[c] * norm) * er[c * 64]; s = s + isum[c] * norm; }
        s = wave_sum_f32(s);
        if (lane == 0) n2[x] = s + xfactor;
      }
    }
    phase_fence();
    P7X_ENV_STAMP(2);

    // ------------------------------------------------------------------ 4. traceback (p7_OATrace)
    // The walk is serial, but most of it needs no decision at all: the C states from the last row down to the row where
    // C took E (known from phase 3) and the N states from the first aligned row up are runs that all lanes write side by
    // side; in between, the M / I / D steps read one 16-bit code each -- a dependent memory access per step when done
    // naively.  Here all lanes walk together (the state is wave-uniform) and fetch the codes of the 64 cells DOWN THE
    // DIAGONAL from the current one in one go: a match-to-match step finds its code in a register, only an insert or a
    // delete (or 64 matches) makes a new fetch.
    uint32_t *ta = a.tr_a + a.tr_off[it];
    int32_t *ti = a.tr_i + a.tr_off[it];
    float *tp = a.tr_pp + a.tr_off[it];
    int n = 0;
    {
      const int cap = Ld + a.M + 16;
      int i = Ld, k = 0, s0 = tC;
      if (lane == 0) { ta[0] = tT; ti[0] = i; ta[1] = tC; ti[1] = i; }
      n = 2;
      const float t1c = (ploop == 0.0f) ? 0.0f : 1.0f, t2e_move = (a.xf_e_move == 0.0f) ? 0.0f : 1.0f;
      const float t2e_loop = (a.xf_e_loop == 0.0f) ? 0.0f : 1.0f, tmove = (pmove == 0.0f) ? 0.0f : 1.0f;
      if (e_row >= 1 && e_s >= 0 && t1c != 0.0f && t2e_move != 0.0f) {
        // C <- C at rows Ld .. e_row + 1 (phase 3 saw C take E for the last time at e_row), then C <- E at e_row
        const int nc = Ld - e_row;
        for (int z = lane; z < nc; z += 64) { ta[2 + z] = (uint32_t) tC | 0x80000000u; ti[2 + z] = Ld - z; }
        n += nc;
        if (G && c_near_row >= e_row) status |= 64 | (1 << 11);
        i = e_row;
        if (lane == 0) { ta[n] = (uint32_t) tE; ti[n] = i; }
        ++n;
        s0 = tE;
      }
      int di = -1, dk = -1;                  // <diag> of lane l holds the code of cell (di - l, dk - l)
      uint32_t diag = 0;
      auto code_at = [&](int ci, int ck) -> unsigned {
        int l = di - ci;
        if (!(l >= 0 && l < 64 && dk - ck == l)) {
          di = ci; dk = ck; l = 0;
          const int ii = ci - lane, kk = ck - lane;
          diag = (ii >= 1 && kk >= 1) ? (uint32_t) bp[(size_t) ii * Mpad + ((kk - 1) % C) * 64 + (kk - 1) / C] : 0u;
        }
        return (unsigned) __builtin_amdgcn_readlane((int) diag, l);
      };
      while (s0 != tS && n < cap) {
        int s1 = -1;
        switch (s0) {
          case tM: {
            if (i < 1 || k < 1) { status |= 4; break; }
            const unsigned w16 = code_at(i, k);
            const int code = w16 & 3;
            if (w16 & (1u << 12)) status |= 64 | (1 << 8);
            s1 = (code == 0) ? tM : (code == 1) ? tI : (code == 2) ? tD : tB;
            --k; --i;
            break;
          }
          case tD: {
            if (i < 1 || k < 1) { status |= 4; break; }
            const unsigned w16 = code_at(i, k);
            const int code = (w16 >> 3) & 1;
            if (w16 & (1u << 14)) status |= 64 | (1 << 10);
            s1 = code ? tD : tM; --k;
            break;
          }
          case tI: {
            if (i < 1 || k < 1) { status |= 4; break; }
            const unsigned w16 = code_at(i, k);
            const int code = (w16 >> 2) & 1;
            if (w16 & (1u << 13)) status |= 64 | (1 << 9);
            s1 = code ? tI : tM; --i;
            break;
          }
          case tN: {
            // N <- N at rows i .. 1, then S at row 0: i + 1 entries, written side by side
            const int room = cap - n, want = i + 1, cnt = want < room ? want : room;
            for (int z = lane; z < cnt; z += 64) {
              const bool last = z == i;
              ta[n + z] = (uint32_t) (last ? tS : tN) | ((uint32_t) k << 8) | (last ? 0u : 0x80000000u);
              ti[n + z] = last ? 0 : i - z;
            }
            n += cnt;
            s0 = cnt == want ? tS : tN;
            i = 0;
            continue;
          }
          case tC: {
            if (i < 1) { status |= 4; break; }
            const float p0 = t1c * (ox[(size_t) (i - 1) * 5 + 4] + px[(size_t) i * 3 + 2]), p1 = t2e_move * ox[(size_t) i * 5 + 0];
            if (G && near_tie(p0, p1, a.oa_guard)) status |= 64 | (1 << 11);
            s1 = (p0 > p1) ? tC : tE;
            break;
          }
          case tJ: {
            if (i < 1) { status |= 4; break; }
            const float p0 = t1c * (ox[(size_t) (i - 1) * 5 + 2] + px[(size_t) i * 3 + 1]), p1 = t2e_loop * ox[(size_t) i * 5 + 0];
            if (G && near_tie(p0, p1, a.oa_guard)) status |= 64 | (1 << 12);
            s1 = (p0 > p1) ? tJ : tE;
            break;
          }
          case tE:
            if (i != e_row || e_s < 0) { status |= 8; break; }   // only the last C<-E row was resolved (unihit envelopes)
            if (e_near) status |= 64 | (1 << 13);
            k = e_k; s1 = e_s;
            break;
          case tB:
            if (G && near_tie(tmove * ox[(size_t) i * 5 + 1], tmove * ox[(size_t) i * 5 + 2], a.oa_guard)) status |= 64 | (1 << 14);
            s1 = (tmove * ox[(size_t) i * 5 + 1] > tmove * ox[(size_t) i * 5 + 2]) ? tN : tJ;
            break;
          default: break;
        }
        if (s1 == -1) { status |= 16; break; }
        if (lane == 0) { ta[n] = (uint32_t) s1 | ((uint32_t) k << 8) | ((s1 == s0) ? 0x80000000u : 0u); ti[n] = i; }
        ++n;
        if ((s1 == tN || s1 == tJ || s1 == tC) && s1 == s0) --i;
        s0 = s1;
      }
      if (s0 != tS) status |= 32;
    }
    n = rfl(n);
    phase_fence();
    // posterior probability of each trace step (get_postprob), all lanes
    int pp_flag = 0;
    for (int z = lane; z < n; z += 64) {
      const uint32_t w = ta[z];
      const int s = (int) (w & 0xffu), k = (int) ((w >> 8) & 0xffffu), i = ti[z];
      const bool same = (w & 0x80000000u) != 0;
      float pp = 0.0f;
      if ((s == tM || s == tI) && i >= 1 && k >= 1) {
        const unsigned w16 = bp[(size_t) i * Mpad + ((k - 1) % C) * 64 + (k - 1) / C];
        pp = pp_from_code((s == tM) ? ((w16 >> 4) & 15u) : ((w16 >> 8) & 15u));
        pp_flag |= (int) ((w16 >> 15) & 1u);
      } else if (same && i >= 1) {
        if (s == tN) pp = px[(size_t) i * 3 + 0];
        else if (s == tJ) pp = px[(size_t) i * 3 + 1];
        else if (s == tC) pp = px[(size_t) i * 3 + 2];
      }
      tp[z] = pp;
      ta[z] = w & 0x7fffffffu;
    }
    if (G && __ballot(pp_flag != 0) != 0ull) status |= 64 | (1 << 15);        // a printed posterior digit within the guard band of the next one
    if (lane == 0) {
      a.out_sc[(size_t) it * 2 + 0] = envsc;
      a.out_sc[(size_t) it * 2 + 1] = oasc;
      a.out_status[it] = status;
      a.tr_n[it] = n;
    }
    phase_fence();      // the workspace is about to be overwritten by this wavefront's next envelope
    P7X_ENV_STAMP(3);
  }
}

// ---------------------------------------------------------------------------- host side
size_t env_work_floats(int C, int Lmax)
{ // floats per wavefront; keep in step with the carving at the top of env_kernel
  const size_t rows = (size_t) Lmax + 1, Mpad = (size_t) 64 * C;
  size_t f = 2 * rows * Mpad + rows * (6 + 6 + 5 + 3 + 1);
  f += (rows * Mpad + 1) / 2;          // back-pointers + posterior digits, 16 bits per cell
  return (f + 63) & ~(size_t) 63;
}

template <typename K>
static int launch_env(K kernel, const ArgRun<EnvArgs> &a, size_t lds_bytes, hipStream_t st)
{
  if (lds_bytes > 64 * 1024)
    P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
  int gx = 1;
  for (int i = 0; i < a.n; ++i) gx = std::max(gx, a.at(i).nblocks);
  hipLaunchKernelGGL(kernel, dim3((unsigned) gx, (unsigned) a.n), dim3((unsigned) env_waves(a.at(0).C) * 64), lds_bytes, st, a.ref());
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

template <typename K>
static int occupancy_env(K kernel, int kEnvBlock, size_t lds_bytes, int *per_cu)
{
  if (lds_bytes > 64 * 1024)
    P7X_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
  P7X_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kernel, kEnvBlock, lds_bytes));
  if (debug_opt(OPT_TRACE_ENVELOPE) > 0) {
    hipFuncAttributes fa; (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kernel));
    std::fprintf(stderr, "[env] occupancy %d blocks/CU of %d threads, lds %zu, regs %d, static lds %zu, maxthreads %d\n", *per_cu, kEnvBlock, lds_bytes,
                 fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
  }
  if (*per_cu < 1) *per_cu = 1;
  return P7X_OK;
}

#define P7X_ENV_SWITCH(EXPR)                                                                                  \
  switch (C) {                                                                                                \
    case 1: { if (LT_) { auto kern = env_kernel<1, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<1, true>; return EXPR; } else { auto kern = env_kernel<1, false>; return EXPR; } }                                                     \
    case 2: { if (LT_) { auto kern = env_kernel<2, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<2, true>; return EXPR; } else { auto kern = env_kernel<2, false>; return EXPR; } }                                                     \
    case 3: { if (LT_) { auto kern = env_kernel<3, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<3, true>; return EXPR; } else { auto kern = env_kernel<3, false>; return EXPR; } }                                                     \
    case 4: { if (LT_) { auto kern = env_kernel<4, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<4, true>; return EXPR; } else { auto kern = env_kernel<4, false>; return EXPR; } }                                                     \
    case 5: { if (LT_) { auto kern = env_kernel<5, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<5, true>; return EXPR; } else { auto kern = env_kernel<5, false>; return EXPR; } }                                                     \
    case 6: { if (LT_) { auto kern = env_kernel<6, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<6, true>; return EXPR; } else { auto kern = env_kernel<6, false>; return EXPR; } }                                                     \
    case 8: { if (LT_) { auto kern = env_kernel<8, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<8, true>; return EXPR; } else { auto kern = env_kernel<8, false>; return EXPR; } }                                                     \
    case 10: { if (LT_) { auto kern = env_kernel<10, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<10, true>; return EXPR; } else { auto kern = env_kernel<10, false>; return EXPR; } }                                                     \
    case 12: { if (LT_) { auto kern = env_kernel<12, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<12, true>; return EXPR; } else { auto kern = env_kernel<12, false>; return EXPR; } }                                                     \
    case 16: { if (LT_) { auto kern = env_kernel<16, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<16, true>; return EXPR; } else { auto kern = env_kernel<16, false>; return EXPR; } }                                                     \
    case 20: { if (LT_) { auto kern = env_kernel<20, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<20, true>; return EXPR; } else { auto kern = env_kernel<20, false>; return EXPR; } }                                                     \
    case 24: { if (LT_) { auto kern = env_kernel<24, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<24, true>; return EXPR; } else { auto kern = env_kernel<24, false>; return EXPR; } }                                                     \
    case 32: { if (LT_) { auto kern = env_kernel<32, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<32, true>; return EXPR; } else { auto kern = env_kernel<32, false>; return EXPR; } } \
    case 48: { if (LT_) { auto kern = env_kernel<48, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<48, true>; return EXPR; } else { auto kern = env_kernel<48, false>; return EXPR; } } \
    case 64: { if (LT_) { auto kern = env_kernel<64, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<64, true>; return EXPR; } else { auto kern = env_kernel<64, false>; return EXPR; } } \
    case 96: { if (LT_) { auto kern = env_kernel<96, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<96, true>; return EXPR; } else { auto kern = env_kernel<96, false>; return EXPR; } } \
    case 128: { if (LT_) { auto kern = env_kernel<128, true, true>; return EXPR; } else if (G_) { auto kern = env_kernel<128, true>; return EXPR; } else { auto kern = env_kernel<128, false>; return EXPR; } }                                                     \
    default: set_error("model too long for the envelope kernel"); return P7X_EINVAL;                         \
  }

static size_t env_lds_bytes(int C, int nrows) { return C > 64 ? (size_t) 256 : (size_t) 64 * C * (32 + (C <= 16 ? (size_t) nrows * 4 : 0)); }

int env_max_blocks(int C, int nrows, int num_cu, int *nblocks)
{
  const size_t lds = env_lds_bytes(C, nrows);
  int per_cu = 1;
  auto finish = [&](int st) { if (st == P7X_OK) *nblocks = num_cu * per_cu; return st; };
  const bool G_ = true, LT_ = false;            // the guarded kernel is never the smaller one
  P7X_ENV_SWITCH(finish(occupancy_env(kern, env_waves(C) * 64, lds, &per_cu)))
}

int env_launch(const ArgRun<EnvArgs> &a, hipStream_t st)
{
  if (a.n <= 0) return P7X_OK;
  const int C = a.at(0).C;
  const bool LT_ = a.at(0).env_emis != nullptr;
  const size_t lds = LT_ ? (C > 64 ? (size_t) 256 : (size_t) 64 * C * 32) : env_lds_bytes(C, a.at(0).nrows);
  const bool G_ = a.at(0).oa_guard > 0.0f;
  P7X_ENV_SWITCH(launch_env(kern, a, lds, st))
}

} // namespace p7x

#ifdef P7X_ENV_PROFILE
extern "C" int p7x_debug_env_profile(unsigned long long *out8)
{
  unsigned long long zero[8] = { 0 };
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(p7x::g_env_prof), sizeof zero) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(p7x::g_env_prof), zero, sizeof zero) != hipSuccess;
}
#endif
