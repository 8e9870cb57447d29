// p7x_internal.hpp -- shared declarations of libp7x (product code; never includes oracle/).
#pragma once
#include "../../include/p7x.h"
#include <cstdint>
#include <string>
#include <vector>

namespace p7x {

constexpr int MAXK  = 20;
constexpr int MAXKP = 29;
constexpr double kLog2  = 0.69314718055994529;
constexpr double kLog2R = 1.44269504088896341;

// special-state / transition indices, same numbering as impl_sse/p7_oprofile.pxd:29-49
enum { XE = 0, XN = 1, XJ = 2, XC = 3 };
enum { MOVE = 0, LOOP = 1 };
enum { tBM = 0, tMM, tIM, tDM, tMD, tMI, tII, tDD, NTRANS };
// generic profile transition order (upstream p7_profile.h)
enum { gMM = 0, gIM, gDM, gBM, gMD, gDD, gMI, gII };

struct Alphabet {
  int type, K, Kp;
  const char *sym;
  unsigned char degen[MAXKP][MAXK];
  static const Alphabet &get(int type);
};

// Everything the path needs about one query, stored UN-striped: index k = 1..M is the node.
// Transition tables follow the oprofile convention (impl_sse/p7_oprofile.pxd:41-49):
//   tw/tf[tBM|tMM|tIM|tDM][k] = score of entering node k (from B / M,I,D of node k-1)
//   tw/tf[tMD|tMI|tII|tDD][k] = score of leaving node k.
struct Profile {
  int M = 0, K = 0, Kp = 0, abc_type = 0, L = 0, max_length = -1, mode = 0;
  float nj = 1.0f;
  std::string name, acc, desc, consensus, rf, mm, cs;
  bool has_acc = false, has_desc = false;
  float evparam[6], cutoff[6], compo[MAXK], bgf[MAXK];
  // generic log-odds profile (p7_ProfileConfig)
  std::vector<float> tsc;        // [(M+1)*8] generic order gMM..gII
  std::vector<float> msc;        // [Kp][M+1]
  float xsc[4][2];
  // MSV (mf_conversion)
  std::vector<uint8_t> rb;       // [Kp][M+1] biased costs, rb[x][0] unused (255)
  uint8_t tbm_b, tec_b, tjb_b, base_b, bias_b;
  float scale_b;
  // Viterbi (vf_conversion)
  std::vector<int16_t> rw;       // [Kp][M+1]
  std::vector<int16_t> tw;       // [NTRANS][M+1]
  int16_t xw[4][2];
  float scale_w; int16_t base_w, ddbound_w; float ncj_roundoff;
  // Forward/Backward (fb_conversion): odds ratios / probabilities
  std::vector<float> rf_;        // [Kp][M+1]
  std::vector<float> tf;         // [NTRANS][M+1]
  float xf[4][2];

  int Q16() const { int q = (M - 1) / 16 + 1; return q < 2 ? 2 : q; }
  int Q8()  const { int q = (M - 1) / 8  + 1; return q < 2 ? 2 : q; }
  int Q4()  const { int q = (M - 1) / 4  + 1; return q < 2 ? 2 : q; }
};

// host math shared by the pipeline (p7x_stats.cpp)
float  sse_expf(float x);                               // Easel esl_sse_expf, one lane
uint8_t unbiased_byteify(float scale_b, float sc);
int16_t wordify(float scale_w, float sc);
double gumbel_surv(double x, double mu, double lambda);
double exp_surv(double x, double mu, double lambda);
double exp_logsurv(double x, double mu, double lambda);
float  null1_score(int L);
void   flogsum_init();
float  flogsum(float a, float b);

void set_error(const std::string &msg);

// Test / diagnostic seam (p7x_debug_set_option, include/p7x.h): process-wide knobs that pick a kernel family for a parity test, bound
// a workspace, or switch a trace on.  -1 = not set (the library decides).  The library reads no environment variables.
enum { OPT_SMALL_BLOCK = 0, OPT_VIT_WAVE, OPT_MSV_EXACT, OPT_MSV_LONG_GROUPS, OPT_MSV_BLOCKS_PER_CU, OPT_ENV_WORKSPACE_GB, OPT_DEVICE_CLUSTERED,
       OPT_TRACE_FINISH, OPT_TRACE_LONGTARGET, OPT_TRACE_ENVELOPE, OPT_HOST_PROFILE, OPT_SSV_KERNEL, OPT_MSV_F16, OPT_ENS_LDS_KB, OPT_ENS_FAIL, OPT_MSV_TIERS, OPT_STAGE_MERGE, OPT_EARLY_PACK, OPT_HOST_ORDER, OPT_VIT_LONG_CUT, OPT_FWD_GROUPED, OPT_REGION_GUARD_PPM, OPT_MSV_K8, OPT_MSV_LANE_BLOCKS, OPT_COUNT };
int debug_opt(int which);

}
struct p7x_oprofile;
namespace p7x {
void attach_dev_cache(p7x_oprofile *om);

} // namespace p7x

struct p7x_oprofile { p7x::Profile p; void *dev_cache = nullptr; };
