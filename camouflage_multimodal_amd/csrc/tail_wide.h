// The per-sample tail for large batches (tail_wide.hip): pooled FFN second layers, fusion MLP and the four heads of 32 samples
// per block in ONE launch (forward).  Arithmetic: bf16 MFMA on two-plane operands (x ~ x_hi + x_lo, W ~ W_hi + W_lo, three
// products per term: relative error ~2^-16, i.e. fp32-grade), fp32 accumulation; the head output layer in fp32 on the VALU.
#pragma once
#include "common.h"
#include "fused_rows.h"

struct TailWideArgs {
  const float *Ymean, *H1mean, *Y2mean, *H2mean;            // pooled node-level outputs [B][256], [B][512] x 2 streams
  // hi / lo weight shadows in fragment order (fused_rows.h, ShadowJob): W13, W23, Wfu0 [256 x 512]; Wfu3 [256 x 256];
  // the four heads' first layers stacked [512 x 256]
  const us16 *T13h, *T13l, *T23h, *T23l, *Tfu0h, *Tfu0l, *Tfu3h, *Tfu3l, *Th0h, *Th0l;
  const float *b13, *b23, *bfu0, *bfu3; const float* bh0[4]; const float* Wh3[4]; const float* bh3[4];
  float* outs;                                              // [B][2C+2]
  // training calls: fp32 copies of what the backward launches read -- comb [B][512], F1 [B][256] (after ReLU and dropout), fused [B][256],
  // hid [B][512] (after ReLU and dropout); all four or none.  With them the head output layer is left to the loss launch (outs may be null).
  float *comb_out, *F1_out, *fused_out, *hid_out;
  int B, C; DropCfg drop;
};
#define TAILW_MAXC 8
int tail_wide_ok(int B, int C);
int launch_tail_wide(const TailWideArgs& a, hipStream_t stream);

// ---- the same for the tail's input-gradient chain (training calls): from the gradient of the four heads' hidden pre-activations
//   dfused = dhid . [Wh0_0; ..; Wh0_3]      dF1 = (F1 > 0) * scale * (dfused . Wfu3)      dcomb = dF1 . Wfu0
//   d(mean H1d) = dcomb[:, :256] . W13      d(mean H2d) = dcomb[:, 256:] . W23
// for 32 samples per block, two-plane operands; the weight planes are the TRANSPOSED matrices' shadows (the contraction runs over a
// weight's rows).  The weight gradients (sums over the batch) are one batched GEMM launch behind it, fed by the fp32 outputs here.
struct TailWideBwdArgs {
  const float* dhid;                                        // [B][512] gradient of the hidden layers' pre-activations (loss launch)
  const float* F1;                                          // [B][256] fusion layer 0 activations (after ReLU and dropout)
  const us16 *Th0h, *Th0l, *Tfu3h, *Tfu3l, *Tfu0h, *Tfu0l, *T13h, *T13l, *T23h, *T23l;   // shadows of Wh0s^T [256 x 512], Wfu3^T [256 x 256],
                                                                                         // Wfu0^T, W13^T, W23^T [512 x 256]
  float *dfused, *dF1, *dcomb, *dHm1, *dHm2;                // [B][256], [B][256], [B][512], [B][512], [B][512]
  int B; float scale;                                       // scale: the dropout keep scale 1 / (1 - p) (1 in eval mode)
};
int launch_tail_wide_bwd(const TailWideBwdArgs& a, hipStream_t stream);
