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
