// p7x_kernels.hpp -- launch interfaces of the HIP kernels (one translation unit per kernel family).
#pragma once
#include "p7x_device.hpp"

namespace p7x {

// ---- batched launches
// Every kernel of the cascade takes its arguments from DEVICE memory, one record per profile ("lane") of a batch of
// queries that run against the same resident target block: blockIdx.y selects the record, blockIdx.x walks that lane's
// work exactly as it did when a launch served one profile.  Records are <stride> bytes apart, so they can be members
// of a larger per-lane struct (LaneArgs in p7x_pipeline.hip).  One query alone is a batch of one.
struct ArgRef { const void *base; uint32_t stride; };
#if defined(__HIPCC__)
template <class A> __device__ __forceinline__ A load_args(const ArgRef r)
{ // constant address space: the record is not written while the kernel runs, so it is fetched with scalar loads
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) A *cptr;
  return *(cptr) (unsigned long long) (static_cast<const char *>(r.base) + (size_t) blockIdx.y * r.stride);
#else
  return *reinterpret_cast<const A *>(static_cast<const char *>(r.base) + (size_t) blockIdx.y * r.stride);   // host pass: never run
#endif
}
#endif
// A run of consecutive lanes that share one kernel instantiation: host copies of the records (for grid sizing) and
// the device copies the kernel reads.
template <class A> struct ArgRun {
  const A *host = nullptr; const void *dev = nullptr; uint32_t stride = 0; int n = 0;
  const A &at(int i) const { return *reinterpret_cast<const A *>(reinterpret_cast<const char *>(host) + (size_t) i * stride); }
  ArgRef ref() const { return ArgRef{ dev, stride }; }
};
// blocks per lane: <want> of them are expected to have work (every kernel walks its list with a grid-stride loop, so
// an estimate that is too low only costs balance, and blocks without work leave before they load their tables); at
// most what is resident at once per lane -- the lanes' blocks queue up behind each other
inline unsigned lane_grid(long want, long resident, int nlanes)
{
  (void) nlanes;
  if (want > resident) want = resident;
  return (unsigned) (want < 1 ? 1 : want);
}

// ... for a kernel whose wavefronts PULL their work items (the lane-per-target MSV kernels: an atomic counter per lane, longest
// groups first): with many lanes in a launch -- a scan's thousands of profiles against a few dozen 64-target groups -- a lane
// gets only as many blocks as it takes to keep the device twice subscribed, down to ONE: its wavefronts then balance the items
// among themselves (the longest group beside the sum of the short ones), where nine blocks of four wavefronts for 33 items give
// every wavefront one item and hold a CU for the longest of them (round 6: profiles/r06_scan_lane_blocks.txt).
inline unsigned lane_grid_pull(long want, long resident, int nlanes, bool old_rule)
{
  if (old_rule || nlanes < 32) return lane_grid(want, resident, nlanes);      // a few lanes with thousands of groups each (a search): as many blocks as are resident
  long b = (2 * resident + nlanes - 1) / nlanes;
  if (b > resident) b = resident;
  if (b > want) b = want;
  return (unsigned) (b < 1 ? 1 : b);
}

// ---- MSV (p7x_msv.hip)
struct MsvArgs {
  const uint32_t *tab;      // [2][kTabRows][S] dwords: parity 0 = "odd" alignment, parity 1 = "even"; K = 8: [kTabRows][S], one alignment
  const uint4 *tiles;
  const int64_t *grp_off;
  const int32_t *grp_nblk;
  const int32_t *slot_len;
  const uint8_t *tjb_tab;
  int ngroups;
  int group_first;          // groups below this one are left to the wave-per-target kernel (the longest targets)
  int base, bias, tec, tbm;
  int *counter;
  int16_t *out_xJ;          // [ngroups*64] slot order; -1 = overflow
  // fast variant: groups whose result is ambiguous are appended here and redone by the exact kernel
  int *amb_count; int *amb_groups; int *counter2;
  const int *group_list; const int *group_count;   // exact kernel: optional list of groups to process
  int R;                    // the lane's register tile (read by the tier kernels, which serve several tiles in one launch)
};
int  msv_pick(int M, int *K);       // row registers per lane and lanes per target (K = 1, 2, 4, 8) of the lane kernels; -1: none fits
int  msv_stride(int R, int K);     // dwords per table row
void msv_build_tables(const Profile &p, int R, int K, std::vector<uint32_t> &out);
// fast kernel over <main> followed by the exact kernel over each lane's list of ambiguous groups (<amb>: records with
// group_list / group_count set); amb == nullptr: the exact kernel over every group of <main>
int  msv_launch(int R, int K, const ArgRun<MsvArgs> &main, const ArgRun<MsvArgs> *amb, int num_cu, hipStream_t st);
// The fast kernel for the lanes of several register tiles in one launch (half-float flavour), and the exact kernel over the
// ambiguous groups of one tile's lanes afterwards.  msv_tier: the tier a tile belongs to (0..kMsvTiers-1); lanes of a
// launch must share it.
constexpr int kMsvTiers = 6;
int  msv_tier(int R, int K);
int  msv_tier_launch(int tier, const ArgRun<MsvArgs> &main, int num_cu, hipStream_t st);
int  msv_exact_launch(int R, int K, const ArgRun<MsvArgs> &amb, int num_cu, hipStream_t st);

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
  const int *abort_flag;    // optional: non-zero on the device = skip all work (buffers not sized for this list)
  int nrows;                // residue rows in the emission table (Kp + 1)
  int *counter;
  // Viterbi
  const int16_t *xwmove_tab; int base_w, xw_e, ddbound;
  int32_t *out_xC;          // [nlist]
  // Viterbi, long-target variant (p7_ViterbiFilter_longtarget): a row whose best match cell reaches the item's score
  // threshold is recorded (item, row, node: one record per cell holding that score) and the DP rows are cleared
  const int *lt_thresh;     // [nlist] per item; NULL: the standard filter
  int *lt_nrec; int *lt_rec; int lt_cap;      // records [cap][3]
  // Forward / Backward
  float xf_e_move, xf_e_loop;
  float *out_sc;            // [nlist] nats
  float *xmx;               // optional [sum (L+1)*6]; rows E,N,J,B,C,SCALE
  const int64_t *xmx_off;   // [nlist] float offset of each target's block
  const float *fwd_xmx;     // Backward only: Forward's blocks (for the scale factors)
};
int vit_pick_C(int M);
// every record of a run has the same C (and nrows); nlist of the host copies bounds the lists and sizes the grid
int vit_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st);
int fwd_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st);
// grouped Forward parser, scores only (p7x_fwdpk.hip): WaveSeqArgs::C = T * 256 + C, trans / emis in its own layout
bool fwdg_pick(int M, int vitC, int *T, int *C);
int fwdg_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st);
void fwdg_build_tables(const Profile &p, int T, int C, std::vector<float> &trans, std::vector<float> &emis);
int bck_launch(const ArgRun<WaveSeqArgs> &a, int num_cu, hipStream_t st);

// ---- MSV for models beyond the register-resident kernels (p7x_vitfwd.hip::msv_wave_kernel), M <= 8192
struct MsvWaveArgs {
  int C, nrows;
  const void *emis;         // int16 [nrows][64*C], (bias - cost) in lane-chunk order, kNegPad outside the model / pad row
  const void *emis_pk;      // long models (C >= 20): the same as packed pairs, [nrows][C/4][64] x (pair, pair) for msv_wavepk_kernel
  const uint8_t *dsq; const int64_t *slot_off; const int32_t *slot_len; const uint8_t *tjb_tab;
  int nslots, base, bias, tec, tbm;
  int16_t *out_xJ;
};
int msv_wave_launch(const ArgRun<MsvWaveArgs> &a, int num_cu, hipStream_t st);

// ---- packed Viterbi filter (p7x_vitpk.hip): T lanes per target, 2P nodes per lane, for M <= 640
struct VitPkArgs {
  const void *trans, *emis;     // vitpk_build_tables()
  const uint8_t *dsq; const int64_t *slot_off; const int32_t *slot_len;
  const int32_t *list; int nlist; const int *nlist_ptr;     // as WaveSeqArgs
  const int *nskip_ptr;     // optional: the first *nskip_ptr list items are left to the wave-per-target kernel
  int nrows;
  const int16_t *xwmove_tab; int base_w, xw_e, ddbound;
  int32_t *out_xC;
};
bool vitpk_pick(int M, int *T, int *P);
void vitpk_build_tables(const Profile &p, int T, int P, std::vector<uint32_t> &trans, std::vector<uint32_t> &emis);
int  vitpk_launch(int T, int P, const ArgRun<VitPkArgs> &a, int num_cu, hipStream_t st);

// ---- envelope rescoring (p7x_envelope.hip): Forward + Backward + decoding/null2/optimal accuracy + traceback,
// one domain envelope per wavefront.  Trace steps come back in traceback order (T first): tr_a = state | k << 8,
// tr_i = residue index inside the envelope (1..Ld, as the traceback saw it), tr_pp = posterior of that step.
struct EnvArgs {
  int M, C, K, nrows;       // K: canonical residues (null2 vector length); nrows: residue rows of the emission table
  const void *trans;        // Forward tables of the profile (float4[2*Mpad], float[nrows][Mpad])
  const void *emis;
  const uint8_t *dsq;
  float nj, xf_e_move, xf_e_loop;   // unihit: 0, 1, 0
  int nenv;
  const int64_t *env_sq;    // [nenv] offset in dsq of the first residue of the envelope
  const int32_t *env_len;   // [nenv] envelope length Ld
  const int32_t *env_L;     // [nenv] full target length (the length model is not re-configured per envelope)
  const int32_t *order;     // [nenv] envelopes by decreasing length: the order in which wavefronts draw them
  int *cursor;              // the job's queue position (zero at launch)
  float *work; int64_t work_stride; int Lmax;    // per-wavefront workspace (env_work_floats), rows 0..Lmax
  int nblocks, slab_base;   // blocks of this job (blockIdx.x beyond them exit); first workspace slab of the job
  float *out_sc;            // [nenv][2] envelope Forward score (nats), optimal accuracy score
  const float *env_emis;    // long-target envelopes: per-envelope match odds [nenv][nrows][Mpad] (NULL: the profile's own table)
  long long env_emis_stride;  // floats per envelope table
  float *out_orig;          // long-target envelopes: [nenv] Forward score with the profile's unmodified odds
  float oa_guard;           // near-tie guard of the optimal-accuracy traceback (p7x_pipeline_cfg.oa_guard)
  int32_t *out_status;      // [nenv] bit 0 Forward range, 1 decoding range (envelope dropped), 2-5 traceback failures,
                            // 6 a near-tie on the trace: the host stage repeats the envelope in the reference's order
                            // (bits 8-15 say where: M / I / D cell choice, C<-E, J<-E, end cell, B<-N/J, posterior digit)
  float *out_null2;         // [nenv][32] null2 odds of the canonical residues
  const int64_t *tr_off;    // [nenv] first trace element; capacity Ld + M + 16 each
  uint32_t *tr_a; int32_t *tr_i; float *tr_pp;
  int32_t *tr_n;            // [nenv] trace length
};
// wavefronts (= envelopes in flight) per block of env_kernel: eight (two per SIMD, 256 registers each); from eight
// nodes per lane on, four (one per SIMD, 512 registers): the row state of two would spill to scratch, which made the
// envelopes of long models 1.4-3.6x slower than half the occupancy does (scripts/env_by_length.py)
constexpr int env_waves(int C) { return C >= 8 ? 4 : 8; }
size_t env_work_floats(int C, int Lmax);
int env_max_blocks(int C, int nrows, int num_cu, int *nblocks);
// every record of the run has the same C and nrows; grid.x = the widest job's nblocks
int env_launch(const ArgRun<EnvArgs> &a, hipStream_t st);

// ---- stochastic traceback ensembles of multi-domain regions (p7x_ensemble.hip; upstream p7_domaindef.c
// region_trace_ensemble): a multihit Forward fill of the region that leaves, per cell and per row, the integer
// thresholds of every choice a stochastic traceback can face there (p7x_choice.hpp), then one wavefront per region that
// walks the region's tracebacks through those records, consuming the region's single generator stream.
struct ChoiceCell; struct ChoiceRow;
struct EnsRegion {          // one multi-domain region, as both kernels see it
  int64_t sq;               // offset in dsq of the region's first residue
  int64_t cell0;            // first cell record: (Lr + 1) rows of M + 1 cells, row-major, node-minor (node 0 unused)
  int64_t row0;             // first row record; rows 0 .. Lr.  The null2 accumulators use the same offsets
  int64_t dom0;             // first domain record of the region's output
  int32_t Lr, L;            // region length; full target length (the length model)
  int32_t job, dom_cap;     // the profile; room for this many domain records at dom0
};
struct EnsJob {             // per query profile
  int M, C, K, Kp, nrows, Q;          // Q = p7O_NQF(M): select_e walks the reference's striped order
  const void *trans, *emis;           // Forward tables of the device image
  const float *rft;                   // [M + 1][32] match odds, residue-minor (p7_Null2_ByTrace)
  const uint8_t *degen;               // [32][32] degeneracy matrix of the alphabet (esl_abc_FAvgScVec)
  int reg_first, nreg;                // the job's regions in EnsArgs::regions
};
struct EnsArgs {
  const EnsJob *jobs; const EnsRegion *regions; int nregions;
  const uint8_t *dsq;
  ChoiceCell *cells; float2 *md; ChoiceRow *rows;      // md: Forward's M and D cell (select_e)
  uint32_t seed_x;          // the generator's state after esl_randomness_Init(seed): every region starts there
  int nsamples;
  float *n2acc;             // [rows] per residue: sum over the samples of the null2 odds ratio (1 outside domains)
  int32_t *dom;             // [5] per domain: sample, sqfrom, sqto (1-based inside the region), hmmfrom, hmmto
  int32_t *out_ndom; int32_t *out_status;               // per region
  int lds_bytes;            // dynamic LDS of the walk kernel: what fits of a region's row records, accumulators and odds table lives there
  uint32_t guard;           // near-threshold guard of the walk, in units of the generator's 2^-32 (p7x_pipeline_cfg.ens_guard x 2^32); 0: off
};

// ---- long-target SSV scan (p7x_ssvlong.hip): one chunk of one strand per wavefront, model split across the lanes
struct SsvLongArgs {
  const uint32_t *tab4q;      // [2 parities][4][(R+3)/4][64] uint4: quads of packed emission pairs of A, C, G, T (staged in LDS)
  const uint32_t *tab_full;   // [2][Kp][R][64] the pairs of every residue code (degenerate residues, read from global memory)
  const uint8_t *dsq;         // the target, 1-based (dsq[0] is a sentinel)
  const uint8_t *comp;        // [Kp] complement of every residue code
  int pair_slack;             // the most a cell can lose in one row with a canonical residue (byte units)
  const long long *chunk_list;  // NULL: every chunk; else the nchunks chunk numbers to scan (a part of a search dealt over devices)
  long long L;                // target length
  int M, Kp;
  int chunk_len;              // rows per chunk
  long long chunks_per_strand, nchunks;     // nchunks = strands x chunks_per_strand; chunk c of strand strand0 + c / chunks_per_strand
  int strand0;                // 1: only the reverse-complement strand is scanned
  int thresh_s;               // score threshold, relative to the begin score and offset by -32768 (the cells' representation)
  int xB, Q16;                // the begin score in byte units; vectors per row of the reference's striped layout (tie-breaks)
  // rows that reach the threshold: position on the strand, strand, and the cell upstream would pick (node, byte score)
  int *nrec; long long *rec_pos; uint8_t *rec_strand; int *rec_k; int *rec_sc; int rec_cap;
};
// <pair>: the row maximum only on every second row, against a threshold lowered by pair_slack, tables with the virtual node
int  ssvlong_pick_R(int M, bool pair);
void ssvlong_build_tables(const Profile &p, int R, bool pair, std::vector<uint32_t> &tab4q, std::vector<uint32_t> &tab_full, int *pair_slack);
int  ssvlong_capacity(int R, bool pair, bool half, int num_cu, long long *waves);      // half: binary16 cells (the default), else int16
int  ssvlong_launch(int R, bool pair, bool half, const SsvLongArgs &a, int num_cu, hipStream_t st);

// ---- thread-per-sequence small stages (p7x_pipeline.hip)
} // namespace p7x
