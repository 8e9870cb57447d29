// p7x_choice.hpp -- the random choices of a stochastic traceback (upstream impl_sse/stotrace.c, Easel esl_rnd_FChoose on
// the "fast" generator), written once for the host twin (p7x_domaindef.cpp) and for the device (p7x_ensemble.hip).
//
// Every choice point of p7_StochasticTrace normalises a short vector of path weights (esl_vec_FNorm), draws one deviate
// roll = x / 2^32 from the LCG x <- 69069 x + 1 and takes the first path i with roll < (p_0 + .. + p_i) / norm, the sums
// in double, norm the Kahan sum of the normalised weights -- falling back to the last path of positive weight when
// rounding leaves the last cumulative ratio below roll.  Because roll is an integer over 2^32, "roll < q" is the integer
// test x < ceil(q 2^32) exactly (q 2^32 is q with another exponent): a choice point is fully described by its integer
// thresholds and its fallback path, which depend on the Forward matrix alone and not on the deviate.  The device computes
// them for every cell while it fills the matrix (all float work in the fill, at full occupancy) and the serial walk that
// consumes the generator is integer compares only; the host twin computes them where its walk passes.  Same functions,
// same operations, same order: the same path is taken.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define P7X_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define P7X_HD inline
#endif

namespace p7x {

P7X_HD uint32_t lcg_next(uint32_t x) { return x * 69069u + 1u; }      // esl_random(), fast type: the deviate is x / 2^32

// esl_vec_FSum: Kahan summation (no contraction, no reassociation: this header is compiled with -ffp-contract=off)
P7X_HD float choice_fsum(const float *v, int n)
{
  float sum = 0.0f, c = 0.0f;
  for (int i = 0; i < n; ++i) { const float y = v[i] - c; const float t = sum + y; c = (t - sum) - y; sum = t; }
  return sum;
}

// Thresholds of one choice point with n <= 4 paths.  p[]: the raw path weights (normalised in place, as esl_vec_FNorm
// does).  Path i < n - 1 is taken when x < T[i] and no earlier path was; *fallback is the path taken when none of those
// is.  A cumulative ratio that reaches 1 (its threshold would be 2^32: always taken) ends the list there: that path
// becomes the fallback and the thresholds from it on are 0 (never) -- the ratios only grow, so nothing behind it could be
// reached anyway, and no threshold needs a 33rd bit.
P7X_HD void choice_thresholds(float *p, int n, uint32_t *T, uint32_t *fallback)
{
  const float s = choice_fsum(p, n);
  if (s != 0.0f) { for (int i = 0; i < n; ++i) p[i] /= s; }
  else           { for (int i = 0; i < n; ++i) p[i] = (float) (1. / (double) (float) n); }
  const double norm = (double) choice_fsum(p, n);
  double sum = 0.0;
  uint32_t fb = 0;
  int always = -1;                                                        // first path whose cumulative ratio is >= 1
  for (int i = 0; i < n; ++i) {
    sum += (double) p[i];
    if (p[i] > 0.0f) fb = (uint32_t) i;
    if (i < n - 1) {
      const double t = __builtin_ceil((sum / norm) * 4294967296.0);
      uint32_t ti = 0;
      if (t >= 4294967296.0) { if (always < 0) always = i; }
      else if (t > 0.0 && always < 0) ti = (uint32_t) t;                  // NaN or zero: never taken
      T[i] = ti;
    }
  }
  *fallback = always >= 0 ? (uint32_t) always : fb;
}
P7X_HD int choice_pick(const uint32_t *T, uint32_t fallback, int n, uint32_t x)
{
  for (int i = 0; i < n - 1; ++i) if (x < T[i]) return i;
  return (int) fallback;
}

// ---- what the device stores per Forward cell (i, k) and per row i (p7x_ensemble.hip); the host twin fills the same
// records from its own matrix where its walk passes, and the debug seams compare them word for word.
// cell, first 16 bytes -- the match cell's choice among B, M, I, D of (i-1, k-1):  T0 T1 T2 | fallback (bits 0-1)
// cell, second 16 bytes -- insert: M or I of (i-1, k); delete: M or D of (i, k-1):  TI TD | fI (bit 0) fD (bit 2) | 0
// row, first 16 bytes -- C: C(i-1) or E(i); J: J(i-1) or E(i); B: N(i) or J(i):     TC TJ TB | fC (bit 0) fJ (bit 2) fB (bit 4)
// row, second 16 bytes -- (float) (1 / xE(i)), the factor of select_e's cumulative sum; three spare words
struct ChoiceCell { uint32_t m[4]; uint32_t id[4]; };
struct ChoiceRow  { uint32_t x[4]; uint32_t e[4]; };

P7X_HD void choice_cell_m(float b, float m, float i, float d, uint32_t *out4)
{ // path[0..3] = B(i-1) bm(k), M(i-1,k-1) tMM(k), I(i-1,k-1) tIM(k), D(i-1,k-1) tDM(k): the products, formed by the caller
  float p[4] = { b, m, i, d };
  uint32_t fb;
  choice_thresholds(p, 4, out4, &fb);
  out4[3] = fb;
}
P7X_HD void choice_pair(float a, float b, uint32_t *T, uint32_t *fallback)
{
  float p[2] = { a, b };
  choice_thresholds(p, 2, T, fallback);
}
P7X_HD int choice_pick_m(const uint32_t *c4, uint32_t x) { return x < c4[0] ? 0 : (x < c4[1] ? 1 : (x < c4[2] ? 2 : (int) (c4[3] & 3u))); }
P7X_HD int choice_pick_pair(uint32_t T, uint32_t fallback, uint32_t x) { return x < T ? 0 : (int) (fallback & 1u); }

} // namespace p7x
