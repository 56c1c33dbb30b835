// Launchers of the bandwidth-bound kernels (misc.hip).  All return hipError_t as int.
#pragma once
#include "common.h"

struct LnSeg { const float* U; float* Y; float* stats; const float* gamma; const float* beta; int rows;
               unsigned short* Y16; /* optional bf16 copy of Y (bf16 schedule) */
               /* optional fused mean pool: mean[sample(row)][c] += Y[row][c] * inv_n (atomics into a zeroed buffer);
                  sample(row) = row_sample[row] (inv_n[sample]) or row / uniform_n when row_sample is null */
               float* mean; const int* row_sample; const float* inv_n; int uniform_n; };
struct LnBwdSeg { const float* U; const float* dY; const float* stats; const float* gamma; float* dU;
                  float* dgamma; float* dbeta; int rows;
                  unsigned short* dU16; /* optional bf16 copy of dU */ };
// per-sample column means: rows of sample b are offs[b]..offs[b+1]-1, or b*uniform_n.. when offs == null
struct SegMean { const float* X; int ld, C; const int* offs; int uniform_n; float* out; int ldo; };
struct BcastSeg { const float* act; const float* v; int ldv; const int* row_sample; const float* inv_n;
                  int uniform_n; float* dst; int rows;
                  unsigned short* dst16; /* when set the result is written as bf16 here INSTEAD of dst */
                  const unsigned short* act16; /* when set the activation is read from this bf16 tensor instead of act */ };

// One launch of small data-movement jobs in front of the bf16 schedule (misc.hip, prep_kernel):
//   PREP_ZERO  : n bytes at dst := 0                                   (n, dst 16-byte multiples)
//   PREP_CAST  : dst[i] = bf16(src[i]), i < n                          (n % 4 == 0)
//   PREP_CAST_T: dst[c * ld_dst + col_off + r] = bf16(src[r * cols + c]), r < rows, c < cols   (transposed copy)
enum : int { PREP_ZERO = 0, PREP_CAST = 1, PREP_CAST_T = 2 };
struct PrepJob { const float* src; void* dst; size_t n; int type, rows, cols, ld_dst, col_off, blk_begin; };
#define PREP_MAXJ 48
struct PrepBatch { PrepJob j[PREP_MAXJ]; int n; };
int launch_prep(PrepBatch& pb, hipStream_t stream);

int ln_supported(int H);
int launch_rowmap(const int* offs, int* row_sample, float* inv_nr, int* tile_off, int4* tile_desc, int B, int ntile_max, int max_nr, hipStream_t stream);

// ---- parameter-space weight gradients of an input projection + the in-projections that read it (fused backward, fusion_abi.hip):
//   dW_in[i][c] += sum_k M[i][k] W_p[c][k] + db[i] b_p[c]      (i < 3H: rows 0..H-1 -> Gq, H..3H-1 -> Gkv)
//   dW_p[c][k]  += sum_i W_in[i][c] M[i][k],   db_p[c] += sum_i W_in[i][c] db[i]      (W_in rows 0..H-1 = Wq, the rest = Wkv)
// with M = dQKV^T x [3H][D] and db = colsum(dQKV) [3H]; H = 256, D = 128; exact fp32, one launch for both streams.
struct UnfoldStream { const float *M, *db, *Wq, *Wkv, *Wp, *bp; float *Gq, *Gkv, *Gp, *Gbp; };
int launch_unfold(const UnfoldStream& rg, const UnfoldStream& kg, hipStream_t stream);
int launch_gather_batch(const float* rg_all, const long long* sample_off, const float* kg_all, const long long* y_all, const float* e_all, const float* s_all,
                        const long long* idx, int B, int T, int D, int KG, float* rg_out, float* kg_out, int* off_out, long long* y_out, float* e_out,
                        float* s_out, float noise_std, unsigned long long seed, hipStream_t stream);
int launch_ln_fwd(const LnSeg& s0, const LnSeg& s1, int H, hipStream_t stream);
int launch_ln_bwd(const LnBwdSeg& s0, const LnBwdSeg& s1, int H, hipStream_t stream);
int launch_seg_mean(const SegMean* segs, int nseg, int B, int max_rows, hipStream_t stream);
int launch_relu_bcast_bwd(const BcastSeg& s0, const BcastSeg& s1, int C, float scale, hipStream_t stream);
int launch_head_out_grad(const float* outs, const float* d_outs, float* d_logits, int B, int W, hipStream_t stream);
int launch_loss(const float* outs, const long long* y, const float* e, const float* s, int B, int C,
                float* terms, float* d_outs, float* d_pre, int* pred, hipStream_t stream);
// head output layer + loss + its backward in one kernel (native training call); needs heads_loss_ok(B, C)
#define HEADS_MAXW 130   /* 2 * num_classes + 2 with num_classes <= 64 (check_dims) */
struct HeadsOut { const float* W[4]; const float* b[4]; float* gW[4]; float* gb[4]; };
int heads_loss_ok(int B, int C);
int launch_heads_loss(const float* hid, const HeadsOut& hp, const long long* y, const float* e, const float* s, int B, int C,
                      int Fh, float scale, float* outs, float* terms, int* pred, float* dhid, hipStream_t stream, float* dlog = nullptr);
int launch_sumsq(const float* g, size_t n, float* out, hipStream_t stream);
int launch_clip_adamw(float* p, float* g, float* m, float* v, size_t n, float* sumsq, float max_norm,
                      float lr, float b1, float b2, float eps, float wd, int step, int zero_grads, hipStream_t stream);

// The whole per-sample tail (pooled FFN second layers, fusion MLP, four heads; with mode 1 also the loss and everything
// backwards) as ONE launch of 64 co-resident blocks: misc.hip, tail_fused_kernel.  B <= 16, hidden 256.
struct TailFusedArgs {
  const float *Ymean, *H1mean, *Y2mean, *H2mean;            // pooled node-level outputs [B][256], [B][512] x 2 streams
  const float *W13, *b13, *W23, *b23, *Wfu0, *bfu0, *Wfu3, *bfu3; const float* Wh0[4]; const float* bh0[4]; const float* Wh3[4]; const float* bh3[4];
  float *gW13, *gb13, *gW23, *gb23, *gWfu0, *gbfu0, *gWfu3, *gbfu3; float* gWh0[4]; float* gbh0[4]; float* gWh3[4]; float* gbh3[4];   // += (mode 1)
  const long long* y; const float* e; const float* s;       // labels (mode 1)
  float* outs; float* terms; int* pred;                     // [B][2C+2]; mode 1: [B][4], [B] (may be null)
  float *F1sum, *hidsum, *dF1sum; unsigned int* counters;   // ZEROED exchange buffers [B][256], [B][512], [B][256]; 4 words (3 arrival counters + timeout flag)
  float *dcomb, *dHm1, *dHm2;                               // mode 1 out: d(mean Z) [B][512]; d(mean H) [B][512] x 2, ZEROED (accumulated with atomics)
  // B > 16 (more than one group of samples), mode 1: the four big weight gradients are sums over ALL groups -- left to ONE batched
  // GEMM launch behind this one (fusion_abi.hip, tail17), which reads these copies of the per-sample operands
  float *comb_out, *F1_out, *fused_out, *dhid_out, *dfused_out, *dF1_out;       // [B][512], [B][256], [B][256], [B][512], [B][256], [B][256]
  int B, C, mode; DropCfg drop;
  unsigned long long* stamps;                               // developer timeline (null in product calls)
  int debug_skip;                                           // (launcher) developer hook: block id + 1 that skips its first arrival
};
extern thread_local int g_tail_debug_skip;      // (set per call from the caller's camo_options_t: fusion_abi.hip, OptScope)
int tail_fused_ok(int B, int C);
int launch_tail_fused(const TailFusedArgs& a, hipStream_t stream);
int tail_timeouts(unsigned int* out);
int launch_tail_poison_to_grads(float* g, hipStream_t stream);      // data parallel: a pending tail timeout -> NaN in g[0] (misc.hip)

// ---- AdamW that also emits the bf16 weight shadows of the fused row-tile schedule (fused_rows.h), so that the next training
// step needs no shadow launch.  A "row block" is a run of rows of one parameter matrix; it feeds one plain shadow (rows pn0.. of a
// logical [pN][cols] matrix) and optionally one transposed shadow (its rows are k' = tk0.. of a logical [tN = cols][tK]).
struct AdamShadowBlock {
  size_t off;                       // element offset of the row block in the flat parameter / gradient / moment buffers
  int rows, cols;                   // rows % 32 == 0, cols % 64 == 0, row-major with ld = cols
  unsigned short* plain; int pN, pn0;
  unsigned short* trans; int tK, tk0;       // trans == nullptr: none
  int tile_begin;                   // (filled by the launcher) first 32 x 64 tile of this block
};
#define ADAM_SHADOW_MAXB 12
#define ADAM_SHADOW_MAXR 16
struct AdamShadowArgs {
  AdamShadowBlock blk[ADAM_SHADOW_MAXB]; int nblk;
  size_t range_begin[ADAM_SHADOW_MAXR], range_len[ADAM_SHADOW_MAXR]; int nrange;   // the rest of the flat buffer, elementwise (16-byte aligned runs)
  int tiles, nbig;                  // (launcher) tile blocks; ranges [0, nbig) are the long ones
};
int launch_clip_adamw_shadows(float* p, float* g, float* m, float* v, float* sumsq, float max_norm, float lr, float b1, float b2,
                              float eps, float wd, int step, int zero_grads, AdamShadowArgs& a, hipStream_t stream);
