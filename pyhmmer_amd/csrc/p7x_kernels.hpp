// p7x_kernels.hpp -- launch interfaces of the HIP kernels (one translation unit per kernel family).
#pragma once
#include "p7x_device.hpp"

namespace p7x {

// ---- MSV (p7x_msv.hip)
struct MsvArgs {
  const uint32_t *tab;      // [2][kTabRows][S] dwords: parity 0 = "odd" alignment, parity 1 = "even"
  const uint4 *tiles;
  const int64_t *grp_off;
  const int32_t *grp_nblk;
  const int32_t *slot_len;
  const uint8_t *tjb_tab;
  int ngroups;
  int base, bias, tec, tbm;
  int *counter;
  int16_t *out_xJ;          // [ngroups*64] slot order; -1 = overflow
  // fast variant: groups whose result is ambiguous are appended here and redone by the exact kernel
  int *amb_count; int *amb_groups; int *counter2;
  const int *group_list; const int *group_count;   // exact kernel: optional list of groups to process
};
int  msv_pick_R(int M);
int  msv_stride(int R);
void msv_build_tables(const Profile &p, int R, int S, std::vector<uint32_t> &out);
int  msv_launch(int R, const MsvArgs &a, int num_cu, hipStream_t st);

// ---- wave-per-sequence stages (p7x_vitfwd.hip): Viterbi filter, Forward / Backward parsers
// Node k = z*C + c + 1 lives in lane z, chunk position c; device tables are stored [c*64 + z].
struct WaveSeqArgs {
  int M, C;                 // C = nodes per lane, Mpad = 64*C
  const void *trans;        // Viterbi: uint4[Mpad] (8 x int16); Forward: float4[2*Mpad] (8 x f32)
  const void *emis;         // Viterbi: int16[kTabRows][Mpad]; Forward: float[kTabRows][Mpad]
  const uint8_t *dsq;       // sentinel-framed residues
  const int64_t *slot_off;  // [nslots] offset of x1
  const int32_t *slot_len;
  const int32_t *list;      // slots to process (NULL: 0..nlist-1)
  int nlist;
  const int *nlist_ptr;     // if non-NULL the list length is read from device memory (no host sync between stages)
  int nrows;                // residue rows in the emission table (Kp + 1)
  int *counter;
  // Viterbi
  const int16_t *xwmove_tab; int base_w, xw_e, ddbound;
  int32_t *out_xC;          // [nlist]
  // Forward / Backward
  float xf_e_move, xf_e_loop;
  float *out_sc;            // [nlist] nats
  float *xmx;               // optional [sum (L+1)*6]; rows E,N,J,B,C,SCALE
  const int64_t *xmx_off;   // [nlist] float offset of each target's block
  const float *fwd_xmx;     // Backward only: Forward's blocks (for the scale factors)
};
int vit_pick_C(int M);
int vit_launch(const WaveSeqArgs &a, int num_cu, hipStream_t st);
int fwd_launch(const WaveSeqArgs &a, int num_cu, hipStream_t st);
int bck_launch(const WaveSeqArgs &a, int num_cu, hipStream_t st);

// ---- thread-per-sequence small stages (p7x_pipeline.hip)
} // namespace p7x
