// pk_issue.hip -- issue rate of the packed 16-bit integer ops the MSV / Viterbi kernels are built from, on gfx950.
//
// Question: does v_pk_add_i16 / v_pk_max_i16 issue a wave64 instruction in 4 cycles (16 lanes
// per clock per SIMD, the roof bench.py assumed) or faster?  Each kernel runs NCHAIN independent dependency chains of
// ITER x 64 ops per lane; wavefronts per SIMD and chains are swept.  Reported per configuration:
//   cycles per wave-instruction per SIMD  =  elapsed shader cycles x (SIMDs with work) / wave-instructions issued
// with elapsed cycles taken both from s_memtime inside the wave (first wave of the grid) and from HIP events x the
// clock reported by the device.  Plain v_add_u32 (known full rate) and v_fma_f32 run as controls.
//
// build: hipcc -O3 --offload-arch=gfx950 -o pk_issue pk_issue.hip        run: ./pk_issue > profiles/...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s2 __attribute__((ext_vector_type(2)));

enum Op { PK_ADD_SAT = 0, PK_MAX = 1, PK_ADD_MAX_MIX = 2, ADD_U32 = 3, FMA_F32 = 4, PK_ADD_WRAP = 5,
          MAX3_I16 = 6, MAX_I16 = 7, ADD_I16_SAT = 8, MAX3_I32 = 9, MAX_I32 = 10, ADD_I32_SAT = 11, PK_ADD_MAX3_MIX = 12, MED3_I16 = 13,
          FMA_F32_2V = 14, FMA_F32_K = 15, PK_ADD_F16 = 16, PK_MAX_F16 = 17, PK_MAX3_F16 = 18, PK_F16_MIX = 19, OR_B32 = 20, NOPS };
static const char *kOpName[NOPS] = { "v_pk_add_i16 clamp", "v_pk_max_i16", "pk_add clamp + pk_max (MSV mix)", "v_add_u32", "v_fma_f32", "v_pk_add_u16",
                                     "v_max3_i16 op_sel (lo,lo,hi)", "v_max_i16 (VOP2)", "v_add_i16 clamp (VOP3)", "v_max3_i32", "v_max_i32 (VOP2)",
                                     "v_add_i32 clamp (VOP3)", "pk_add clamp + max3_i16 (candidate)", "v_med3_i16",
                                     "v_fma_f32 c, c, e, c (2 VGPRs read)", "v_fma_f32 c, c, 1.0, e (2 VGPRs read)",
                                     "v_pk_add_f16 clamp", "v_pk_max_f16", "v_pk_maximum3_f16 (3 VGPRs read)", "2 pk_add_f16 clamp + pk_maximum3_f16 (round-5 MSV mix)",
                                     "v_or_b32" };

// One asm statement holds the whole unrolled body (.rept): between separate asm statements the compiler puts a
// conservative s_nop, which would be measured too.
#define P7X_BODY1(INS) ".rept 64\n\t" INS(0) ".endr"
#define P7X_BODY4(INS) ".rept 64\n\t" INS(0) INS(1) INS(2) INS(3) ".endr"
#define P7X_BODY8(INS) ".rept 64\n\t" INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) ".endr"
#define I_PK_ADD_SAT(c)  "v_pk_add_i16 %" #c ", %" #c ", %16 clamp\n\t"
#define I_PK_ADD_WRAP(c) "v_pk_add_u16 %" #c ", %" #c ", %16\n\t"
#define I_PK_MAX(c)      "v_pk_max_i16 %" #c ", %" #c ", %16\n\t"
#define I_ADD_U32(c)     "v_add_u32 %" #c ", %" #c ", %16\n\t"
#define I_FMA_F32(c)     "v_fma_f32 %" #c ", %" #c ", %16, %16\n\t"
// the control above reads three VGPR operands (c, e, e); these two read two: is the 4-cycle figure the operand fetch?
#define I_FMA_F32_2V(c)  "v_fma_f32 %" #c ", %" #c ", %16, %" #c "\n\t"
#define I_FMA_F32_K(c)   "v_fma_f32 %" #c ", %" #c ", 1.0, %16\n\t"
#define I_MAX3_I16(c)    "v_max3_i16 %" #c ", %" #c ", %16, %16 op_sel:[0,0,1,0]\n\t"
#define I_MED3_I16(c)    "v_med3_i16 %" #c ", %" #c ", %16, %16 op_sel:[0,0,1,0]\n\t"
#define I_MAX_I16(c)     "v_max_i16 %" #c ", %" #c ", %16\n\t"
#define I_ADD_I16_SAT(c) "v_add_i16 %" #c ", %" #c ", %16 clamp\n\t"
#define I_MAX3_I32(c)    "v_max3_i32 %" #c ", %" #c ", %16, %16\n\t"
#define I_MAX_I32(c)     "v_max_i32 %" #c ", %" #c ", %16\n\t"
#define I_ADD_I32_SAT(c) "v_add_i32 %" #c ", %" #c ", %16 clamp\n\t"
#define I_PK_ADD_F16(c)  "v_pk_add_f16 %" #c ", %" #c ", %16 clamp\n\t"
#define I_PK_MAX_F16(c)  "v_pk_max_f16 %" #c ", %" #c ", %16\n\t"
#define I_OR_B32(c)      "v_or_b32 %" #c ", %" #c ", %16\n\t"
// the accumulator %8+c takes its own chain's register and the seed: three distinct VGPRs per instruction
#define I_PK_MAX3_F16_0 "v_pk_maximum3_f16 %8, %8, %0, %16\n\t"
#define I_PK_MAX3_F16_1 "v_pk_maximum3_f16 %9, %9, %1, %16\n\t"
#define I_PK_MAX3_F16_2 "v_pk_maximum3_f16 %10, %10, %2, %16\n\t"
#define I_PK_MAX3_F16_3 "v_pk_maximum3_f16 %11, %11, %3, %16\n\t"
#define I_PK_MAX3_F16_4 "v_pk_maximum3_f16 %12, %12, %4, %16\n\t"
#define I_PK_MAX3_F16_5 "v_pk_maximum3_f16 %13, %13, %5, %16\n\t"
#define I_PK_MAX3_F16_6 "v_pk_maximum3_f16 %14, %14, %6, %16\n\t"
#define I_PK_MAX3_F16_7 "v_pk_maximum3_f16 %15, %15, %7, %16\n\t"
// the half-float MSV mix: two adds, one maximum3 over both results (three instructions per two registers)
#define I_HMIX01 "v_pk_add_f16 %0, %0, %16 clamp\n\tv_pk_add_f16 %1, %1, %16 clamp\n\tv_pk_maximum3_f16 %8, %8, %0, %1\n\t"
#define I_HMIX23 "v_pk_add_f16 %2, %2, %16 clamp\n\tv_pk_add_f16 %3, %3, %16 clamp\n\tv_pk_maximum3_f16 %9, %9, %2, %3\n\t"
#define I_HMIX45 "v_pk_add_f16 %4, %4, %16 clamp\n\tv_pk_add_f16 %5, %5, %16 clamp\n\tv_pk_maximum3_f16 %10, %10, %4, %5\n\t"
#define I_HMIX67 "v_pk_add_f16 %6, %6, %16 clamp\n\tv_pk_add_f16 %7, %7, %16 clamp\n\tv_pk_maximum3_f16 %11, %11, %6, %7\n\t"
#define I_MX0 "v_pk_add_i16 %0, %0, %16 clamp\n\tv_max3_i16 %8, %8, %0, %0 op_sel:[0,0,1,0]\n\t"
#define I_MX1 "v_pk_add_i16 %1, %1, %16 clamp\n\tv_max3_i16 %9, %9, %1, %1 op_sel:[0,0,1,0]\n\t"
#define I_MX2 "v_pk_add_i16 %2, %2, %16 clamp\n\tv_max3_i16 %10, %10, %2, %2 op_sel:[0,0,1,0]\n\t"
#define I_MX3 "v_pk_add_i16 %3, %3, %16 clamp\n\tv_max3_i16 %11, %11, %3, %3 op_sel:[0,0,1,0]\n\t"
#define I_MX4 "v_pk_add_i16 %4, %4, %16 clamp\n\tv_max3_i16 %12, %12, %4, %4 op_sel:[0,0,1,0]\n\t"
#define I_MX5 "v_pk_add_i16 %5, %5, %16 clamp\n\tv_max3_i16 %13, %13, %5, %5 op_sel:[0,0,1,0]\n\t"
#define I_MX6 "v_pk_add_i16 %6, %6, %16 clamp\n\tv_max3_i16 %14, %14, %6, %6 op_sel:[0,0,1,0]\n\t"
#define I_MX7 "v_pk_add_i16 %7, %7, %16 clamp\n\tv_max3_i16 %15, %15, %7, %7 op_sel:[0,0,1,0]\n\t"
// the accumulator operands of the mix are %8..%15: spell them out (asm operand numbers cannot be computed)
#define I_MIX0 "v_pk_add_i16 %0, %0, %16 clamp\n\tv_pk_max_i16 %8, %8, %0\n\t"
#define I_MIX1 "v_pk_add_i16 %1, %1, %16 clamp\n\tv_pk_max_i16 %9, %9, %1\n\t"
#define I_MIX2 "v_pk_add_i16 %2, %2, %16 clamp\n\tv_pk_max_i16 %10, %10, %2\n\t"
#define I_MIX3 "v_pk_add_i16 %3, %3, %16 clamp\n\tv_pk_max_i16 %11, %11, %3\n\t"
#define I_MIX4 "v_pk_add_i16 %4, %4, %16 clamp\n\tv_pk_max_i16 %12, %12, %4\n\t"
#define I_MIX5 "v_pk_add_i16 %5, %5, %16 clamp\n\tv_pk_max_i16 %13, %13, %5\n\t"
#define I_MIX6 "v_pk_add_i16 %6, %6, %16 clamp\n\tv_pk_max_i16 %14, %14, %6\n\t"
#define I_MIX7 "v_pk_add_i16 %7, %7, %16 clamp\n\tv_pk_max_i16 %15, %15, %7\n\t"
#define P7X_OPERANDS : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), \
                       "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : "v"(e)

template <int OP, int NCHAIN>
__device__ __forceinline__ void body(uint32_t (&v)[8], uint32_t (&acc)[8], uint32_t e)
{
#define P7X_EMIT(INS)                                                        \
  if constexpr (NCHAIN == 1) asm volatile(P7X_BODY1(INS) P7X_OPERANDS);       \
  else if constexpr (NCHAIN == 4) asm volatile(P7X_BODY4(INS) P7X_OPERANDS);  \
  else asm volatile(P7X_BODY8(INS) P7X_OPERANDS);
  if constexpr (OP == PK_ADD_SAT) { P7X_EMIT(I_PK_ADD_SAT) }
  else if constexpr (OP == PK_ADD_WRAP) { P7X_EMIT(I_PK_ADD_WRAP) }
  else if constexpr (OP == PK_MAX) { P7X_EMIT(I_PK_MAX) }
  else if constexpr (OP == ADD_U32) { P7X_EMIT(I_ADD_U32) }
  else if constexpr (OP == FMA_F32) { P7X_EMIT(I_FMA_F32) }
  else if constexpr (OP == FMA_F32_2V) { P7X_EMIT(I_FMA_F32_2V) }
  else if constexpr (OP == FMA_F32_K) { P7X_EMIT(I_FMA_F32_K) }
  else if constexpr (OP == MAX3_I16) { P7X_EMIT(I_MAX3_I16) }
  else if constexpr (OP == MED3_I16) { P7X_EMIT(I_MED3_I16) }
  else if constexpr (OP == MAX_I16) { P7X_EMIT(I_MAX_I16) }
  else if constexpr (OP == ADD_I16_SAT) { P7X_EMIT(I_ADD_I16_SAT) }
  else if constexpr (OP == MAX3_I32) { P7X_EMIT(I_MAX3_I32) }
  else if constexpr (OP == MAX_I32) { P7X_EMIT(I_MAX_I32) }
  else if constexpr (OP == ADD_I32_SAT) { P7X_EMIT(I_ADD_I32_SAT) }
  else if constexpr (OP == PK_ADD_F16) { P7X_EMIT(I_PK_ADD_F16) }
  else if constexpr (OP == PK_MAX_F16) { P7X_EMIT(I_PK_MAX_F16) }
  else if constexpr (OP == OR_B32) { P7X_EMIT(I_OR_B32) }
  else if constexpr (OP == PK_MAX3_F16) {
    if constexpr (NCHAIN == 1) asm volatile(".rept 64\n\t" I_PK_MAX3_F16_0 ".endr" P7X_OPERANDS);
    else if constexpr (NCHAIN == 4) asm volatile(".rept 64\n\t" I_PK_MAX3_F16_0 I_PK_MAX3_F16_1 I_PK_MAX3_F16_2 I_PK_MAX3_F16_3 ".endr" P7X_OPERANDS);
    else asm volatile(".rept 64\n\t" I_PK_MAX3_F16_0 I_PK_MAX3_F16_1 I_PK_MAX3_F16_2 I_PK_MAX3_F16_3 I_PK_MAX3_F16_4 I_PK_MAX3_F16_5 I_PK_MAX3_F16_6 I_PK_MAX3_F16_7 ".endr" P7X_OPERANDS);
  }
  else if constexpr (OP == PK_F16_MIX) {        // NCHAIN counts registers: 1 -> one pair, 4 -> two pairs, 8 -> four pairs
    if constexpr (NCHAIN == 1) asm volatile(".rept 64\n\t" I_HMIX01 ".endr" P7X_OPERANDS);
    else if constexpr (NCHAIN == 4) asm volatile(".rept 64\n\t" I_HMIX01 I_HMIX23 ".endr" P7X_OPERANDS);
    else asm volatile(".rept 64\n\t" I_HMIX01 I_HMIX23 I_HMIX45 I_HMIX67 ".endr" P7X_OPERANDS);
  }
  else if constexpr (OP == PK_ADD_MAX3_MIX) {
    if constexpr (NCHAIN == 1) asm volatile(".rept 64\n\t" I_MX0 ".endr" P7X_OPERANDS);
    else if constexpr (NCHAIN == 4) asm volatile(".rept 64\n\t" I_MX0 I_MX1 I_MX2 I_MX3 ".endr" P7X_OPERANDS);
    else asm volatile(".rept 64\n\t" I_MX0 I_MX1 I_MX2 I_MX3 I_MX4 I_MX5 I_MX6 I_MX7 ".endr" P7X_OPERANDS);
  }
  else {
    if constexpr (NCHAIN == 1) asm volatile(".rept 64\n\t" I_MIX0 ".endr" P7X_OPERANDS);
    else if constexpr (NCHAIN == 4) asm volatile(".rept 64\n\t" I_MIX0 I_MIX1 I_MIX2 I_MIX3 ".endr" P7X_OPERANDS);
    else asm volatile(".rept 64\n\t" I_MIX0 I_MIX1 I_MIX2 I_MIX3 I_MIX4 I_MIX5 I_MIX6 I_MIX7 ".endr" P7X_OPERANDS);
  }
}

template <int OP, int NCHAIN>
__global__ void __launch_bounds__(64) issue_kernel(uint32_t *out, unsigned long long *cyc, int iters, uint32_t seed)
{
  uint32_t v[8], acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { v[c] = seed + threadIdx.x * 7u + c; acc[c] = 0x80008000u; }
  const uint32_t e = seed | 1u;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) body<OP, NCHAIN>(v, acc, e);
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s ^= v[c] ^ acc[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP, int NCHAIN>
static void run(int waves_per_simd, int num_cu, double clk_hz, uint32_t *d_out, unsigned long long *d_cyc)
{
  const int iters = 2000;
  const int nblocks = num_cu * 4 * waves_per_simd;         // one wave per block: the dispatcher spreads them over SIMDs
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((issue_kernel<OP, NCHAIN>), dim3(nblocks), dim3(64), 0, 0, d_out, d_cyc, 10, 1u);   // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((issue_kernel<OP, NCHAIN>), dim3(nblocks), dim3(64), 0, 0, d_out, d_cyc, iters, 3u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(nblocks);
  CK(hipMemcpy(cyc.data(), d_cyc, nblocks * 8, hipMemcpyDeviceToHost));
  double mean = 0; unsigned long long mx = 0;
  for (auto c : cyc) { mean += (double) c; if (c > mx) mx = c; }
  mean /= nblocks;
  const double per_wave_insts = OP == PK_F16_MIX ? (double) iters * 64 * 3 * (NCHAIN == 1 ? 1 : NCHAIN / 2)
                                                 : (double) iters * 64 * NCHAIN * ((OP == PK_ADD_MAX_MIX || OP == PK_ADD_MAX3_MIX) ? 2 : 1);
  // every SIMD holds waves_per_simd waves when the dispatcher balances them: instructions issued per SIMD
  const double per_simd_insts = per_wave_insts * waves_per_simd;
  const double cyc_counter = mean / per_simd_insts;                        // s_memtime ticks per wave-instruction per SIMD
  const double cyc_events = (ms * 1e-3 * clk_hz) / per_simd_insts;         // event time x nominal clock
  printf("| %-32s | %d | %d | %8.3f | %8.3f | %8.3f | %7.1f |\n", kOpName[OP], NCHAIN, waves_per_simd, ms, cyc_counter, cyc_events,
         64.0 / cyc_events);
}

template <int OP>
static void sweep(int num_cu, double clk_hz, uint32_t *d_out, unsigned long long *d_cyc)
{
  for (int w : { 1, 2, 3, 4 }) {
    run<OP, 1>(w, num_cu, clk_hz, d_out, d_cyc);
    run<OP, 4>(w, num_cu, clk_hz, d_out, d_cyc);
    run<OP, 8>(w, num_cu, clk_hz, d_out, d_cyc);
  }
}

int main(int argc, char **argv)
{
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int num_cu = prop.multiProcessorCount;
  const double clk_hz = (double) prop.clockRate * 1e3;
  printf("# pk_issue: %s, %d CUs, clockRate %.0f MHz (nominal; event-derived cycles assume it)\n\n", prop.name, num_cu, clk_hz / 1e6);
  printf("One wave per block, `waves/SIMD` x 4 x CUs blocks; `cyc/inst (counter)` = s_memtime ticks of a wave / instructions its SIMD issued\n"
         "(s_memtime may tick at a fixed reference clock, see the v_add_u32 control); `cyc/inst (events)` = HIP event time x nominal clock /\n"
         "instructions per SIMD; `lanes/clk` = 64 / the latter.\n\n");
  printf("| op | chains | waves/SIMD | ms | cyc/inst (counter) | cyc/inst (events) | lanes/clk/SIMD |\n|---|---|---|---|---|---|---|\n");
  uint32_t *d_out; unsigned long long *d_cyc;
  CK(hipMalloc(&d_out, (size_t) num_cu * 4 * 8 * 64 * 4)); CK(hipMalloc(&d_cyc, (size_t) num_cu * 4 * 8 * 8));
  if (argc > 1 && argv[1][0] == 'h') {                 // "h": round 5, the half-float MSV flavour's instructions
    sweep<ADD_U32>(num_cu, clk_hz, d_out, d_cyc);
    sweep<OR_B32>(num_cu, clk_hz, d_out, d_cyc);
    sweep<PK_ADD_SAT>(num_cu, clk_hz, d_out, d_cyc);
    sweep<PK_ADD_F16>(num_cu, clk_hz, d_out, d_cyc);
    sweep<PK_MAX_F16>(num_cu, clk_hz, d_out, d_cyc);
    sweep<PK_MAX3_F16>(num_cu, clk_hz, d_out, d_cyc);
    sweep<PK_ADD_MAX_MIX>(num_cu, clk_hz, d_out, d_cyc);
    sweep<PK_F16_MIX>(num_cu, clk_hz, d_out, d_cyc);
    return 0;
  }
  sweep<ADD_U32>(num_cu, clk_hz, d_out, d_cyc);
  sweep<FMA_F32>(num_cu, clk_hz, d_out, d_cyc);
  sweep<FMA_F32_2V>(num_cu, clk_hz, d_out, d_cyc);
  sweep<FMA_F32_K>(num_cu, clk_hz, d_out, d_cyc);
  if (argc > 1 && argv[1][0] == 'f') return 0;          // "f": the f32 controls only
  sweep<PK_ADD_SAT>(num_cu, clk_hz, d_out, d_cyc);
  sweep<PK_ADD_WRAP>(num_cu, clk_hz, d_out, d_cyc);
  sweep<PK_MAX>(num_cu, clk_hz, d_out, d_cyc);
  sweep<PK_ADD_MAX_MIX>(num_cu, clk_hz, d_out, d_cyc);
  sweep<MAX3_I16>(num_cu, clk_hz, d_out, d_cyc);
  sweep<MED3_I16>(num_cu, clk_hz, d_out, d_cyc);
  sweep<MAX_I16>(num_cu, clk_hz, d_out, d_cyc);
  sweep<ADD_I16_SAT>(num_cu, clk_hz, d_out, d_cyc);
  sweep<MAX3_I32>(num_cu, clk_hz, d_out, d_cyc);
  sweep<MAX_I32>(num_cu, clk_hz, d_out, d_cyc);
  sweep<ADD_I32_SAT>(num_cu, clk_hz, d_out, d_cyc);
  sweep<PK_ADD_MAX3_MIX>(num_cu, clk_hz, d_out, d_cyc);
  return 0;
}
