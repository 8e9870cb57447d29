// p7x_fwdpk.hip -- Forward parser (scores only), grouped: T lanes per target (T = 16 or 32), 64 / T targets per wavefront.
//
// p7_ForwardParser runs on every Viterbi survivor (reference p7_pipeline.pxd:130); only its score is
// needed there (the special-state rows are formed again, by fwd_kernel, for the few targets that pass F3).  The
// wave-per-target fwd_kernel (p7x_vitfwd.hip) spends most of a row on work that does not shrink with the model -- the
// six-step affine scan of the D->D chain, the six-step row sum, 64 lanes for a model of 100 nodes -- so a library of
// Pfam-sized models (median 120 nodes) pays 80 lane-operations per cell.  Here a group of T lanes owns a target: the scan
// and the row sum are four (T = 16) or five (T = 32) steps inside the group, every lane carries nodes, and the special
// states are per-lane floats that are uniform within a group (the packed Viterbi kernel's arrangement, p7x_vitpk.hip).
//   lane s of a group owns nodes s C + 1 .. s C + C (M <= T C); tables in LDS, index c T + s
//   a group takes its next target from the lane's work counter as soon as it has finished one; a group without a target
//   computes on (what its lanes then produce is never read: no masking inside the row)
// Same recurrence and rescaling as fwd_kernel (odds space, row divided by xE when xE > 1e4); the additions of a row sum
// and of the D->D chain associate differently, so scores differ from fwd_kernel's in the last bits -- as fwd_kernel's do
// from upstream's; F3 decisions inside the guard band are re-taken on the host in upstream's order either way.
#include "p7x_wave.hpp"
#include <map>
#include <mutex>

namespace p7x {

namespace {

#define P7X_G_DPPF(v, old, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float) (old)), __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))

// inclusive scan of the affine maps x -> a + p x over the lanes of a group (Kogge-Stone inside the 16-lane rows; T = 32: rows
// 1 and 3 then take the last lane of the row before them)
template <int T>
__device__ __forceinline__ void group_affine_scan(float &sa, float &sp)
{
#define P7X_G_STEP(ctrl, rmask) { const float pa_ = P7X_G_DPPF(sa, 0.0f, ctrl, rmask), pp_ = P7X_G_DPPF(sp, 1.0f, ctrl, rmask); sa = sa + pa_ * sp; sp = sp * pp_; }
  P7X_G_STEP(0x111, 0xf) P7X_G_STEP(0x112, 0xf) P7X_G_STEP(0x114, 0xf) P7X_G_STEP(0x118, 0xf)
  if constexpr (T == 32) P7X_G_STEP(0x142, 0xa)
#undef P7X_G_STEP
}
// sum over the lanes of a group, every lane receives it
template <int T>
__device__ __forceinline__ float group_sum(float v)
{
  v = v + P7X_G_DPPF(v, 0.0f, 0xB1, 0xf);       // quad_perm [1,0,3,2]
  v = v + P7X_G_DPPF(v, 0.0f, 0x4E, 0xf);       // quad_perm [2,3,0,1]
  v = v + P7X_G_DPPF(v, 0.0f, 0x141, 0xf);      // row_half_mirror
  v = v + P7X_G_DPPF(v, 0.0f, 0x140, 0xf);      // row_mirror
  if constexpr (T == 32) v = v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
  return v;
}

} // namespace

// The groups of a wavefront take their targets from the lane's work counter one at a time (a.counter, zeroed with the
// batch's counters): a group that finishes a short target fetches the next while its neighbours are still inside a long
// one, so a wavefront never idles three groups behind its longest target -- the Forward work list is not sorted by length.
// Fused multiply-adds throughout: this pass only feeds the F3 decision (guarded, see above), not the rows the host twin
// mirrors; the sums differ from fwd_kernel's in the last bits either way.
template <int T, int C>
__global__ void __launch_bounds__(256) fwdg_kernel(const ArgRef ref)
{
  constexpr int G = 64 / T, N = T * C;        // targets per wavefront, node slots of a group
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WaveSeqArgs a = load_args<WaveSeqArgs>(ref);
  const int nlist = (a.abort_flag && *a.abort_flag) ? 0 : (a.nlist_ptr ? *a.nlist_ptr : a.nlist);
  if ((int) (blockIdx.x * 4 * G) >= nlist) return;                  // no item for this block: skip the table load
  float4 *tra = reinterpret_cast<float4 *>(smem);                   // [N] BM MM IM DM
  float4 *trb = tra + N;                                            // [N] MD MI II DD
  float *em = reinterpret_cast<float *>(trb + N);                   // [nrows][N]
  {
    const float4 *gt = reinterpret_cast<const float4 *>(a.trans);
    for (int i = threadIdx.x; i < 2 * N; i += 256) tra[i] = gt[i];
    const float4 *ge = reinterpret_cast<const float4 *>(a.emis);
    float4 *le = reinterpret_cast<float4 *>(em);
    for (int i = threadIdx.x; i < a.nrows * N / 4; i += 256) le[i] = ge[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int s = lane % T, g = lane / T;
  const bool first = s == 0;
  float ddprod = 1.0f;
  float tdd[C], tmd[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { const float4 b = trb[c * T + s]; tmd[c] = b.x; tdd[c] = b.w; ddprod *= b.w; }

  // per-lane state, uniform within a group
  bool active = false, done = false;
  int it = 0, L = 0, i = 0;
  const uint8_t *sq = a.dsq;
  uint32_t word = 0;
  float pmove = 1.0f, ploop = 0.0f;
  float mm[C], im[C], dm[C];
#pragma unroll
  for (int c = 0; c < C; ++c) mm[c] = im[c] = dm[c] = 0.0f;
  float xN = 1.0f, xB = 1.0f, xJ = 0.0f, xC = 0.0f, totscale = 0.0f;

  for (;;) {
    const bool need = !active && !done;
    if (__any(need)) {
      int nxt = 0;
      if (need && first) nxt = atomicAdd(a.counter, 1);
      nxt = __shfl(nxt, g * T);
      if (need) {
        if (nxt < nlist) {
          it = nxt;
          const int slot = a.list ? a.list[it] : it;
          L = a.slot_len[slot];
          sq = a.dsq + a.slot_off[slot];
          i = 0;
          pmove = (2.0f + 1.0f) / ((float) L + 2.0f + 1.0f); ploop = 1.0f - pmove;
#pragma unroll
          for (int c = 0; c < C; ++c) mm[c] = im[c] = dm[c] = 0.0f;
          xN = 1.0f; xB = pmove; xJ = 0.0f; xC = 0.0f; totscale = 0.0f;
          active = L > 0;
          if (!active && first) a.out_sc[it] = __builtin_inff();
        } else done = true;
      }
    }
    if (!__any(active)) break;
    // residues: lane s holds rows b .. b + 3 of its target's current chunk of 4T rows, b = chunk start + 4 s
    const int inchunk = i & (4 * T - 1);
    if (active && inchunk == 0) {
      word = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int r = i + 4 * s + b;
        const uint32_t x = (r < L) ? (uint32_t) sq[r] : 0u;
        word |= x << (8 * b);
      }
    }
    const uint32_t wv = (uint32_t) __shfl((int) word, g * T + (inchunk >> 2));
    const uint32_t x = (wv >> (8 * (inchunk & 3))) & 0xffu;
    const float *er = em + x * N + s;
    // the node before this lane's first one lives in the lane before; a group's first lane has none (the moves are made by
    // every lane: a DPP step must not run under a partial EXEC mask)
    const float mq = dpp_shr1f(mm[C - 1], 0.0f), iq = dpp_shr1f(im[C - 1], 0.0f), dq = dpp_shr1f(dm[C - 1], 0.0f);
    float mp = first ? 0.0f : mq, ip = first ? 0.0f : iq, dp = first ? 0.0f : dq;
    float esum = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float4 t = tra[c * T + s], u = trb[c * T + s];
      float sv = xB * t.x;
      sv = __builtin_fmaf(mp, t.y, sv);
      sv = __builtin_fmaf(ip, t.z, sv);
      sv = __builtin_fmaf(dp, t.w, sv);
      sv = sv * er[c * T];
      esum = esum + sv;
      mp = mm[c]; ip = im[c]; dp = dm[c];
      im[c] = __builtin_fmaf(ip, u.z, mp * u.y);
      mm[c] = sv;
    }
    // D(i,k) = M(i,k-1) tMD(k-1) + D(i,k-1) tDD(k-1): serial inside the lane, affine scan across the group's lanes
    float A = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) { dm[c] = A; A = __builtin_fmaf(A, tdd[c], mm[c] * tmd[c]); }
    float sa = A, sp = ddprod;
    group_affine_scan<T>(sa, sp);
    {
      const float wq = dpp_shr1f(sa, 0.0f);
      float w = first ? 0.0f : wq;                                  // exclusive: the carry entering this lane
#pragma unroll
      for (int c = 0; c < C; ++c) { dm[c] = dm[c] + w; esum = esum + dm[c]; w = w * tdd[c]; }
    }
    const float xE = group_sum<T>(esum);
    xN = xN * ploop;
    xC = __builtin_fmaf(xC, ploop, xE * a.xf_e_move);
    xJ = __builtin_fmaf(xJ, ploop, xE * a.xf_e_loop);
    xB = (xJ + xN) * pmove;
    const bool big = xE > 1.0e4f;
    if (__any(big)) {
      const float inv = big ? (float) (1.0 / (double) xE) : 1.0f;
      xN *= inv; xC *= inv; xJ *= inv; xB *= inv;
#pragma unroll
      for (int c = 0; c < C; ++c) { mm[c] *= inv; dm[c] *= inv; im[c] *= inv; }
      if (big) totscale = (float) ((double) totscale + log((double) xE));
    }
    if (active) {
      if (i == L - 1) {                                             // this target's last row: its score
        float score;
        if (xC != xC) score = __builtin_nanf("");
        else if (xC == 0.0f || __builtin_isinf(xC)) score = __builtin_inff();
        else score = (float) ((double) totscale + log((double) (xC * pmove)));
        if (first) a.out_sc[it] = score;
        active = false;
      } else ++i;
    }
  }
}

// ---------------------------------------------------------------------------- host side
// Which (T, C) serves a model: the wave-per-target kernels' nodes-per-lane classes 1 .. 4 (M <= 256) map onto one grouped
// instantiation each (and 5, 6 -- M <= 384 -- onto T = 32 with ten and twelve nodes per lane), so the lanes of a kernel class
// (LaneClass::C) share it.
bool fwdg_pick(int M, int vitC, int *T, int *C)
{
  (void) M;
  switch (vitC) {
    case 1: *T = 16; *C = 4; return true;      // M <= 64
    case 2: *T = 16; *C = 8; return true;      // M <= 128
    case 3: *T = 32; *C = 6; return true;      // M <= 192
    case 4: *T = 32; *C = 8; return true;      // M <= 256
    case 5: *T = 32; *C = 10; return true;     // M <= 320
    case 6: *T = 32; *C = 12; return true;     // M <= 384
    default: return false;
  }
}

// trans: float4 [2][T C] (BM MM IM DM plane, then MD MI II DD), index c T + s = node s C + c + 1; emis: float [nrows][T C]
void fwdg_build_tables(const Profile &p, int T, int C, std::vector<float> &trans, std::vector<float> &emis)
{
  const int N = T * C, nrows = p.Kp + 1;
  trans.assign((size_t) 2 * N * 4, 0.0f);
  emis.assign((size_t) nrows * N, 0.0f);
  for (int s = 0; s < T; ++s)
    for (int c = 0; c < C; ++c) {
      const int k = s * C + c + 1;
      if (k > p.M) continue;
      const size_t idx = (size_t) c * T + s;
      for (int t = 0; t < 4; ++t) trans[idx * 4 + t] = p.tf[(size_t) t * (p.M + 1) + k];
      for (int t = 4; t < 8; ++t) trans[((size_t) N + idx) * 4 + (t - 4)] = p.tf[(size_t) t * (p.M + 1) + k];
      for (int x = 0; x < p.Kp; ++x) emis[(size_t) x * N + idx] = p.rf_[(size_t) x * (p.M + 1) + k];
    }
}

template <int T, int C>
static int launch_fwdg(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st)
{
  const size_t lds = ((size_t) 2 * T * C * 16) + (size_t) a.at(0).nrows * T * C * 4;
  auto kern = fwdg_kernel<T, C>;
  static std::map<int, int> per_cu_by_device;
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
  constexpr int G = 64 / T;
  long want = 0;      // a group per expected target (the groups fetch their work themselves: fewer blocks only make the chains longer)
  for (int i = 0; i < a.n; ++i) want = std::max<long>(want, ((long) a.at(i).nlist + 4 * G - 1) / (4 * G));
  if (want <= 0) return P7X_OK;
  hipLaunchKernelGGL(kern, dim3(lane_grid(want, (long) num_cu * per_cu, a.n), (unsigned) a.n), dim3(256), lds, st, a.ref());
  P7X_HIP(hipGetLastError());
  return P7X_OK;
}

// every record of the run has the same (T, C): WaveSeqArgs::C = T * 256 + C
int fwdg_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st)
{
  if (a.n <= 0) return P7X_OK;
  const int T = a.at(0).C / 256, C = a.at(0).C % 256;
  if (T == 16 && C == 4) return launch_fwdg<16, 4>(a, num_cu, st);
  if (T == 16 && C == 8) return launch_fwdg<16, 8>(a, num_cu, st);
  if (T == 32 && C == 6) return launch_fwdg<32, 6>(a, num_cu, st);
  if (T == 32 && C == 8) return launch_fwdg<32, 8>(a, num_cu, st);
  if (T == 32 && C == 10) return launch_fwdg<32, 10>(a, num_cu, st);
  if (T == 32 && C == 12) return launch_fwdg<32, 12>(a, num_cu, st);
  set_error("no grouped Forward kernel for this model length");
  return P7X_EINVAL;
}

} // namespace p7x
