// p7x_envfwd.hpp -- one row of the full Forward recurrence with the row state in registers (upstream impl_sse/fwdback.c
// p7_Forward), shared by the envelope kernel (p7x_envelope.hip: unihit, an envelope) and the ensemble kernel
// (p7x_ensemble.hip: multihit, a multi-domain region).  The host twin's forward_full() (p7x_domaindef.cpp) performs the
// same operations in the same order; what is decided from these values is decided alike on both sides.
#pragma once
#include "p7x_wave.hpp"

namespace p7x {

// node loops of the envelope kernel: unrolled (row state in registers) up to this many nodes per lane, rolled beyond
#ifndef P7X_ENV_UNROLL_MAX
#define P7X_ENV_UNROLL_MAX 32
#endif
constexpr int unroll_env(int C) { return C <= P7X_ENV_UNROLL_MAX ? C : 1; }


// One row of the unihit Forward recurrence on the envelope, state in registers.  Phase 1 runs it for the envelope score
// and the per-row scale factors; phase 3 runs it AGAIN next to the decoding (same code, same operations in the same order:
// bit-identical values), which is what lets the kernel park only Backward's rows in HBM.
template <int C>
struct EnvForward {
  float mm[C], im[C], dm[C];
  float ddprod;
  float xN, xB, xJ, xC, xE, scale, totscale;
  template <class TV>
  __device__ __forceinline__ void init(const TV &tr, int lane, float pmove)
  {
#pragma unroll unroll_env(C)
    for (int c = 0; c < C; ++c) mm[c] = im[c] = dm[c] = 0.0f;
    ddprod = 1.0f;
#pragma unroll unroll_env(C)
    for (int c = 0; c < C; ++c) ddprod *= tr.dd(c * 64 + lane);
    xN = 1.0f; xB = pmove; xJ = 0.0f; xC = 0.0f; xE = 0.0f; scale = 1.0f; totscale = 0.0f;
  }
  template <class TV>
  __device__ __forceinline__ void row(const TV &tr, const float *em, int Mpad, int lane, int x, float pmove, float ploop,
                                      float xf_e_move, float xf_e_loop)
  {
    const float *er = em + x * Mpad + lane;
    float mp = dpp_shr1f(mm[C - 1], 0.0f), ip = dpp_shr1f(im[C - 1], 0.0f), dp = dpp_shr1f(dm[C - 1], 0.0f);
    float esum = 0.0f;
    float t_dd[C], t_md[C];
#pragma unroll unroll_env(C)
    for (int c = 0; c < C; ++c) {
      const F8 t = tr.at(c * 64 + lane);
      float sv = xB * t.bm;
      sv = sv + mp * t.mm;
      sv = sv + ip * t.im;
      sv = sv + dp * t.dm;
      sv = sv * er[c * 64];
      esum = esum + sv;
      mp = mm[c]; ip = im[c]; dp = dm[c];
      im[c] = mp * t.mi + ip * t.ii;
      mm[c] = sv;
      t_dd[c] = t.dd; t_md[c] = t.md;
    }
    float A = 0.0f;
#pragma unroll unroll_env(C)
    for (int c = 0; c < C; ++c) { dm[c] = A; A = mm[c] * t_md[c] + A * t_dd[c]; }
    float sa = A, sp = ddprod;
    affine_scan_up(sa, sp);
    {
      float w = dpp_shr1f(sa, 0.0f);
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) { dm[c] = dm[c] + w; esum = esum + dm[c]; w = w * t_dd[c]; }
    }
    xE = wave_sum_f32(esum);
    xN = xN * ploop;
    xC = (xC * ploop) + (xE * xf_e_move);
    xJ = (xJ * ploop) + (xE * xf_e_loop);
    xB = (xJ * pmove) + (xN * pmove);
    scale = 1.0f;
    if (xE > 1.0e4f) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      const float inv = (float) (1.0 / (double) xE);
#pragma unroll unroll_env(C)
      for (int c = 0; c < C; ++c) { mm[c] *= inv; dm[c] *= inv; im[c] *= inv; }
      scale = xE;
      totscale = (float) ((double) totscale + log((double) xE));        // float += double, as upstream (and the host twin) has it
      xE = 1.0f;
    }
  }
};

} // namespace p7x
