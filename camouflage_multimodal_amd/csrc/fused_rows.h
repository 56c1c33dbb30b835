// Fused row-tile kernels of the bf16 schedule at the reference configuration (hidden 256, 8 heads of 32, input dims 128,
// Nk <= 16): launchers and argument blocks (fused_rows.hip).  All launchers return hipError_t as int.
//
// A "tile" is 32 rows of one stream's packed activations; a block of 4 waves walks a tile through a chain of linear
// layers whose weights it streams from L2 in MFMA-fragment order (weight "shadows", bf16) while every intermediate stays
// in LDS / registers.  Forward = 2 launches for BOTH streams:
//   front: x -> proj -> [q | k' | v'] in-projections                       (RG rows and KG rows, 32-row tiles of the packed rows)
//   back : attention -> out-projection + residual -> LayerNorm -> FFN layer 0 (+ReLU, dropout) -> pooled sums
//          (RG: 32-row tiles of one sample against its 13 KG keys; KG: one block per sample against its Nr RG keys)
#pragma once
#include "common.h"

typedef unsigned short us16;

// ---- weight shadows: W [N][K] fp32 -> bf16 in the order a block's 4 waves stream it:  [wave][k step of 16][tile of 32 rows][lane][8]
// with row n = 32 (wave * N/128 + tile) + (lane & 31), k = 16 kstep + 8 (lane >> 5) + j  (tests/test_fragment_maps.py::frag_order).
// A logical W is up to 4 source blocks stacked along N (transposed == 0: W[n][k] = src_i[(n - n0_i) * ld_i + k]) or, for the
// backward's dy.W products, stacked along K and read transposed (transposed == 1: W[n][k] = src_i[(k - k0_i) * ld_i + n]).
struct ShadowJob { us16* dst; int N, K, transposed, nsrc; const float* src[4]; int rows[4]; int ld[4]; int chunk_begin; int lo; };   // lo: the residual plane bf16(W - bf16(W))
#define SHADOW_MAXJ 24
#define SHADOW_MAXZ 24
struct ShadowBatch { ShadowJob j[SHADOW_MAXJ]; int n; void* zero_ptr[SHADOW_MAXZ]; size_t zero_bytes[SHADOW_MAXZ]; int nzero; };
int launch_weight_shadows(ShadowBatch& sb, hipStream_t stream);

// ---- forward, front half
struct FrontStream {
  const float* X; int M;                       // fp32 input rows [M][128]
  const us16* W0; const float* b0;             // projection: shadow of [256 x 128], bias [256]
  const us16* W1; const float* bq; const float* bkv;   // in-projections: shadow of [768 x 256] = [Wq; Wk'; Wv'], biases [256], [512]
  us16* X16;                                   // [M][128]  bf16 copy of the input (weight-gradient operand; written when save != 0)
  us16* R16; us16* Q16; us16* KV16;            // [M][256] projection, [M][256] queries (pre-scaled by 1/sqrt(32)), [M][512] keys|values
  int tile_begin;
};
#define FUSED_FRONT_MAXZ 4
#define FUSED_FRONT_MAXX 20
#define FUSED_BWD1_MAXZ 16
struct FrontArgs {
  FrontStream s[2]; float qscale; int save; unsigned long long* stamps; int exp;   // exp: developer experiments (timing only), 0 in product calls
  // clears that ride at the end of every block when no shadow launch precedes this one (the step's atomics block, d(mean H)):
  // 16-byte aligned, sizes multiples of 16; spread over the grid
  void* zero_ptr[FUSED_FRONT_MAXZ]; unsigned zero_bytes[FUSED_FRONT_MAXZ]; int nzero;
  // wide tiles only: weight-shadow jobs that ride in extra blocks behind the tile blocks (the per-sample tail's hi / lo planes,
  // tail_wide.h: rebuilt by every call that takes the wide tail, so they need no validity tracking)
  int split3;                                  // wide tiles only: three blocks per tile, one in-projection pass each
  ShadowJob xjob[FUSED_FRONT_MAXX]; int nxjob; int xchunks; int xblock0;       // (xchunks, xblock0: filled by the launcher)
};
int launch_fused_front(FrontArgs& a, int variant, hipStream_t stream);   // variant: 0 <depth 12>, 1 <depth 12, rotated k order> (default), 2 <depth 16, rotated>

// ---- forward, back half
struct BackStream {
  const us16* Wo; const float* bo; const us16* W1; const float* b1; const float* ln_g; const float* ln_b;
  const us16* R16;                             // residual rows (the front half's projection)
  us16* O16; us16* Y16; us16* XH16; float* rstd; uint32_t* mask;   // saved for backward when save != 0: attention output, LN output,
                                                                   // normalised LN input, 1/std per row, ReLU/dropout bit mask [rows][16 words]
  float* Ymean; float* Hmean;                  // [B][256], [B][512]: per-sample means, accumulated with atomics into zeroed buffers
  uint32_t site_ffn;
};
struct BackArgs {
  BackStream s[2];                             // 0: RG rows, 1: KG rows
  const us16* Q16; const us16* KV16;           // RG queries [T][256]; KG keys|values [B*Nk][512]
  const us16* Q2_16; const us16* KV2_16;       // KG queries [B*Nk][256]; RG keys|values [T][512]
  const int* off; const int* tile_off; const float* inv_nr;
  const int4* tile_desc;                       // per 32-row RG tile {sample, first packed row, rows, 1 / Nr as bits}; sample = -1 past the last tile
  float* lse2;                                 // [B][8][16][2]: max and sum of the KG->RG softmax (saved for backward)
  int lead_tiles;                              // (filled by the launcher) RG tile blocks dispatched in front of the KG split blocks
  float* part; int* tickets; int max_splits;   // KG->RG attention runs as ceil(Nr / 64) split blocks per sample (max_splits = the
                                               // largest count: grid sizing): partials [rg_tiles_max][8][FUSED_PART_FLOATS] indexed by
                                               // the split's first 32-row tile, and one ZEROED arrival counter per sample
  int B, Nk, rg_tiles_max, rows_rg;             // rows_rg = T (launch-timing bookkeeping only)
  DropCfg drop; int save; int exp;
  unsigned long long* stamps;                  // developer timeline (null in product calls)
};
extern thread_local int g_back_lead_mode;                   // developer A/B: 0 = KG split blocks always first
int launch_fused_back(BackArgs& a, int variant, hipStream_t stream);

// ---- backward, first half (see fused_rows.hip)
struct Bwd1Stream {
  const us16* W1T; const us16* WoT;            // transposed shadows: [256 x 512] (dY = dH . W1), [256 x 256] (dO = dU . Wo)
  const uint32_t* mask; const us16* XH16; const float* rstd; const float* ln_g;      // saved by the forward
  const float* dHm; int ld_dHm;                // d(mean H)  [B][ld]: gradient w.r.t. the pooled FFN activation
  const float* dcomb; int ld_dcomb;            // d(mean Z)  [B][ld] (+ this stream's column offset): pooled residual-path gradient
  us16* dH16; us16* dU16;                      // out: [rows][512] FFN pre-activation gradient, [rows][256] LayerNorm input gradient
  float* dgamma; float* dbeta;                 // += (atomics)
};
struct Bwd1Args {
  Bwd1Stream s[2];                             // 0: RG rows, 1: KG rows
  const us16* Q16; const us16* KV16;           // RG queries (pre-scaled) [T][256]; KG keys|values [B*Nk][512]
  us16* dQKV16; float* dKV;                    // out: dQ into columns 0..255 of [T][768]; dK|dV [B*Nk][512] += (atomics, zeroed by the caller)
  const us16* O2_16; us16* dO2_16; float* delta2;   // KG: attention output in, its gradient out [B*Nk][256], row-dots out [B][8][16]
  const int* off; const int* tile_off; const float* inv_nr; const int4* tile_desc;
  const int* row_sample;                       // RG row -> sample (the batch descriptor's table)
  int B, Nk, rg_tiles_max, rows_rg; float qscale; DropCfg drop; unsigned long long* stamps;
  int exp;                                     // developer A/B (bwd_wide2.hip: weight fragments in flight); product calls pass 0
  int writer_blocks, writer_first_row;         // (filled by the launcher) dH16 writer blocks and the first row of [RG rows | KG rows] they cover
  // clears for the weight-gradient launch (pad rows of its operands) when no shadow launch did them: one extra block each
  void* zero_ptr[FUSED_BWD1_MAXZ]; unsigned zero_bytes[FUSED_BWD1_MAXZ]; int nzero;
};
int launch_fused_bwd1(Bwd1Args& a, int variant, hipStream_t stream, int kg_only = 0);   // kg_only: the KG rows' blocks alone (launch_wide2_bwd1)
// the RG rows on 64-row half-blocks (bwd_wide2.hip) + the KG rows through launch_fused_bwd1(kg_only); same arguments, outputs and saved set
int launch_wide2_bwd1(Bwd1Args& a, int variant, hipStream_t stream);
size_t fused_bwd1_lds();

// ---- backward, second half (see fused_rows.hip)
struct Bwd2Args {
  const us16* Q2_16; const us16* dO2_16; const float* lse2; const float* delta2;   // KG side, per sample: queries (pre-scaled), d(attention
                                                                                    // output) [B*Nk][256], softmax {max, sum} [B][8][16][2], row-dots [B][8][16]
  const us16* KV2_16;                          // RG keys|values [T][512]
  us16* dQKV16;                                // [T][768]: dQ in (columns 0..255, first half), dK2|dV2 out (columns 256..767)
  const us16* dU16; const us16* WcRgT; us16* dR16;    // dR = dU + [dQ|dK2|dV2] . [Wq1; Wk2; Wv2]: shadow [256 x 768] (transposed job), out [T][256]
  float* dQ2acc;                               // [B*Nk][256] += (atomics; zeroed by the caller)
  const float* dKV;                            // [B*Nk][512] dK|dV sums of the first half
  const us16* dU2_16; const us16* WcKgT; us16* dQKVkg16; us16* dG16;              // KG rows: [B*Nk][768] out (weight-gradient operand), [B*Nk][256] out
  float* dGpart;                               // [B*Nk][256] fp32 scratch: the part of dG an early block computes while the RG tiles run
  int* tickets;                                // [B] zeroed arrival counters (one per sample)
  const int* off; const int* tile_off; const int4* tile_desc;
  int B, Nk, rg_tiles_max, rows_rg; float qscale; DropCfg drop; unsigned long long* stamps;
  int param_space;                             // 1: no dR / dG products (bwd2p_kernel): the caller takes the projections' weight gradients in parameter space
  int split_finish;                            // (param_space only) 1: no arrival protocol -- the KG rows' dQ2 sums become bf16 in a second, B-block launch
};
int launch_fused_bwd2(Bwd2Args& a, int variant, hipStream_t stream);
// param_space calls of large batches: the RG rows on 64-row blocks, wave = head, no barriers (bwd_wide2.hip); same outputs
int launch_wide2_bwd2(Bwd2Args& a, hipStream_t stream);
size_t fused_bwd2_lds();

#define FUSED_PART_FLOATS 544
#define FUSED_MAX_SPLITS 64
size_t fused_front_lds();
size_t fused_back_lds();

// ---- wide row tiles for large batches (fused_wide.hip): the same argument blocks, weight shadows and tile table; a block of 8
// waves owns rt (1, 2 or 4) consecutive 32-row tiles and feeds rt MFMAs from every weight fragment it loads.  The back kernel
// computes the KG->RG attention partials inside the RG blocks (one per run of sub-tiles of a sample: partial slot = the run's
// first tile), so BackArgs::max_splits / lead_tiles are unused there; a sample may span at most FUSED_WIDE_MAXSEG blocks.
#define FUSED_WIDE_MAXSEG 48
int launch_wide_front(FrontArgs& a, int rt, hipStream_t stream, int kg_only = 0);     // kg_only: stream 1 (the KG rows) alone
int launch_wide_rgfwd(const FrontStream& f, float qscale, BackArgs& b, int rt, int max_nr, hipStream_t stream);   // RG rows: front + back in one launch
int launch_wide_back(BackArgs& a, int rt, int max_nr, hipStream_t stream);
int wide_max_rows(int rt);

// ---- second wide design (fused_wide2.hip): the RG rows' whole forward in one launch on 64-row half-blocks of 4 waves, two
// independent blocks per CU, + the KG rows' launch behind it.  Same arguments as launch_wide_rgfwd.  Inference calls (b.save == 0, no
// dropout) read the folded in-projection [q | k2 | v2] = x Wf^T + bf (Wf = [Wq1; Wk2; Wv2] Wrg as a [768 x 128] shadow, bf [768] fp32,
// built by launch_fold_rg whenever the parameters changed) and not f.W1 / f.bq / f.bkv; saving / dropout calls read the unfolded
// weights and write the backward's saved set (Wf / bf unused).
int launch_fold_rg(const float* Wq, const float* Wkv, const float* bq, const float* bkv, const float* Wrg, const float* brg, us16* Wf, float* bf, hipStream_t stream);
// save_r16 (saving calls): 0 when the backward takes the projections' weight gradients in parameter space (nothing reads R16 then)
int launch_wide2_rgfwd(const FrontStream& f, const us16* Wf, const float* bf, float qscale, BackArgs& b, int max_nr, int save_r16, hipStream_t stream);
int wide2_max_rows();
