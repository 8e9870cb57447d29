// sustained.hip -- what the chip sustains, in wall-clock terms, for the MSV kernel's instruction mix on gfx950.
//
// pk_issue.hip measures the issue rate of the packed ops in cycles on short launches.  The MSV kernel keeps every SIMD
// busy for milliseconds, and its achieved rate in wall-clock terms also depends on the clock the part holds under that
// load.  This bench runs, with the MSV kernel's launch shape (256-thread blocks, 3 per CU, ~170 VGPRs per lane held by
// a register-resident row), for about 10 ms each:
//   A. the pure VALU mix: v_pk_add_i16 clamp + v_pk_max_i16 per register, nothing else;
//   B. the same with the kernel's LDS traffic: one ds_read_b64 per two register updates (4 ops), per-lane table rows,
//      conflict-free stride, double-buffered with lgkmcnt like p7x_msv.hip;
// and reports packed lane-ops per second (64 lanes x 2 cells per op counted as 1 "op per cell": the unit of
// bench.py's roofline.valu), i.e. the peak GCUPS of a 1.0-op-per-cell kernel whose only work is that mix.
//
// build: hipcc -O3 --offload-arch=gfx950 -o sustained sustained.hip        run: ./sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(uint32_t u) { return __builtin_bit_cast(s2, u); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s2 pk_max(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 pk_adds(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }
// round 5: the half-float flavour of the MSV kernel -- v_pk_add_f16 clamp per register, ONE v_pk_maximum3_f16 per two
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 as_h2(s2 v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ s2 h_adds(s2 a, s2 b)
{
  const h2 z = { (_Float16) 0.0f, (_Float16) 0.0f }, o = { (_Float16) 1.0f, (_Float16) 1.0f };
  return __builtin_bit_cast(s2, __builtin_elementwise_min(__builtin_elementwise_max(as_h2(a) + as_h2(b), z), o));
}
__device__ __forceinline__ s2 h_max3(s2 a, s2 b, s2 c)
{ return __builtin_bit_cast(s2, __builtin_elementwise_maximum(__builtin_elementwise_maximum(as_h2(a), as_h2(b)), as_h2(c))); }

constexpr int R = 128;          // row registers (a multiple of 8)
constexpr int S = 134;          // table row stride in dwords, S/2 odd: conflict-free for <= 32 rows
constexpr int ROWS = 32;

struct Chunk { uint2 e0, e1, e2, e3; };
template <int OFF>
__device__ __forceinline__ void lds_issue4(uint32_t addr, Chunk &c)
{
  asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\tds_read_b64 %3, %4 offset:%8"
               : "=&v"(c.e0), "=&v"(c.e1), "=&v"(c.e2), "=&v"(c.e3) : "v"(addr), "i"(OFF), "i"(OFF + 8), "i"(OFF + 16), "i"(OFF + 24));
}
template <int PENDING>
__device__ __forceinline__ void lds_wait(Chunk &c) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(c.e0), "+v"(c.e1), "+v"(c.e2), "+v"(c.e3) : "i"(PENDING)); }

template <int T, int LDS, int HALF = 0>
struct Row {
  static __device__ __forceinline__ void run(s2 (&v)[R], uint32_t addr, Chunk &cur, Chunk &nxt, s2 &accA, s2 &accB, const s2 e)
  {
    constexpr bool last = T == R / 8 - 1;
    if constexpr (LDS) {
      if constexpr (!last) lds_issue4<(T + 1) * 32>(addr, nxt);
      lds_wait<last ? 0 : 4>(cur);
    }
#define PAIR(JJ, E)                                                                         \
    if constexpr (HALF) {                                                                   \
      v[2 * (JJ)] = h_adds(v[2 * (JJ)], LDS ? as_s2((E).x) : e);                            \
      v[2 * (JJ) + 1] = h_adds(v[2 * (JJ) + 1], LDS ? as_s2((E).y) : e);                    \
      if constexpr ((JJ) & 1) { accB = h_max3(accB, v[2 * (JJ)], v[2 * (JJ) + 1]); asm volatile("" : "+v"(accB)); } \
      else { accA = h_max3(accA, v[2 * (JJ)], v[2 * (JJ) + 1]); asm volatile("" : "+v"(accA)); } \
    } else {                                                                                \
    v[2 * (JJ)] = pk_adds(v[2 * (JJ)], LDS ? as_s2((E).x) : e);                             \
    v[2 * (JJ) + 1] = pk_adds(v[2 * (JJ) + 1], LDS ? as_s2((E).y) : e);                     \
    accA = pk_max(accA, v[2 * (JJ)]); accB = pk_max(accB, v[2 * (JJ) + 1]);                 \
    asm volatile("" : "+v"(accA), "+v"(accB)); }
    PAIR(4 * T + 0, cur.e0) PAIR(4 * T + 1, cur.e1) PAIR(4 * T + 2, cur.e2) PAIR(4 * T + 3, cur.e3)
#undef PAIR
    if constexpr (!last) Row<T + 1, LDS, HALF>::run(v, addr, nxt, cur, accA, accB, e);
  }
};

template <int LDS, int HALF = 0>
__global__ void __launch_bounds__(256, 3) mix_kernel(uint32_t *out, int rows, uint32_t seed)
{
  __shared__ __attribute__((aligned(16))) uint32_t tab[ROWS * S];
  for (int i = threadIdx.x; i < ROWS * S; i += 256) tab[i] = HALF ? 0x9c009c00u : 0xfffdfffeu + (i & 1);      // small negative emissions (half: -1/256)
  __syncthreads();
  s2 v[R];
#pragma unroll
  for (int j = 0; j < R; ++j) v[j] = HALF ? as_s2(0x38003800u) : as_s2(0x80008000u + (uint32_t) (threadIdx.x + j));
  s2 accA = HALF ? as_s2(0u) : as_s2(0x80008000u), accB = accA;
  const s2 e = HALF ? as_s2(0x9c009c00u) : as_s2(seed | 0xfffe0000u);
  uint32_t x = (threadIdx.x * 7u + seed) % 20u;
  for (int r = 0; r < rows; ++r) {
    Chunk ca, cb;
    ca.e0 = ca.e1 = ca.e2 = ca.e3 = cb.e0 = cb.e1 = cb.e2 = cb.e3 = make_uint2(0, 0);
    const uint32_t addr = x * (uint32_t) (S * 4);
    if constexpr (LDS) lds_issue4<0>(addr, ca);
    Row<0, LDS, HALF>::run(v, addr, ca, cb, accA, accB, e);
    x = (x * 5u + 3u) % 20u;                      // the next residue
  }
  uint32_t s = as_u32(accA) ^ as_u32(accB);
#pragma unroll
  for (int j = 0; j < R; ++j) s ^= as_u32(v[j]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS, int HALF = 0>
static void run(const char *name, int num_cu, int rows, uint32_t *d_out)
{
  const int nblocks = num_cu * 3;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mix_kernel<LDS, HALF>), dim3(nblocks), dim3(256), 0, 0, d_out, 64, 1u);
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<LDS, HALF>), dim3(nblocks), dim3(256), 0, 0, d_out, rows, 3u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double cells = (double) nblocks * 256.0 * rows * R * 2.0;          // two cells per register update
    printf("| %-28s | %6d | %8.3f | %9.1f | %6.2f |\n", name, rows, ms, cells / (ms * 1e-3) / 1e9,
           cells / (ms * 1e-3) / ((double) num_cu * 4 * 16) / 1e9);
  }
}

int main()
{
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int num_cu = prop.multiProcessorCount;
  printf("# sustained: %s, %d CUs; 3 blocks of 256 threads per CU, R = %d row registers per lane\n\n", prop.name, num_cu, R);
  printf("`GCUPS` = cells per second of a kernel that spends exactly one packed add and one packed max per two cells (the MSV kernel's\n"
         "two ops per register update); `GHz-equivalent` = that rate / (SIMDs x 16 lanes per clock): the clock a 4-cycle-per-instruction\n"
         "issue would need, i.e. the sustained shader clock if the VALU never idles.\n\n");
  printf("| mix | rows | ms | GCUPS | GHz-equivalent |\n|---|---|---|---|---|\n");
  uint32_t *d_out; CK(hipMalloc(&d_out, (size_t) num_cu * 3 * 256 * 4));
  run<0>("pk_add + pk_max only", num_cu, 20000, d_out);
  run<1>("+ ds_read_b64 per 4 ops", num_cu, 20000, d_out);
  run<0>("pk_add + pk_max only", num_cu, 100000, d_out);
  run<1>("+ ds_read_b64 per 4 ops", num_cu, 100000, d_out);
  // round 5: 2 x v_pk_add_f16 clamp + 1 x v_pk_maximum3_f16 per register pair (0.75 op per cell)
  run<0, 1>("half: 2 add + max3", num_cu, 20000, d_out);
  run<1, 1>("half: + ds_read_b64 per 3 ops", num_cu, 20000, d_out);
  run<0, 1>("half: 2 add + max3", num_cu, 100000, d_out);
  run<1, 1>("half: + ds_read_b64 per 3 ops", num_cu, 100000, d_out);
  return 0;
}
