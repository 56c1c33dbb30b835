// Fused per-row-tile kernels (fused_fwd.hip): launchers and argument blocks.
#pragma once
#include "common.h"

struct RgFwdArgs {
  const float* X; int D;                 // packed RG inputs [T, D]
  const int* offs;                       // [B+1]
  // bf16 shadow copies of the weights, pre-tiled [N/256][K/32][256][32] (launch_cast_tiled_bf16)
  const unsigned short *Wrg, *Wq, *Wkv2, *Wo, *W1;
  const float *brg, *bq, *bkv2, *bo, *b1, *ln_g, *ln_b;
  const float* KV;                       // [B*Nk, 2H] K|V of the KG stream (fp32)
  float *R, *Q, *KV2, *P, *O, *U, *stats, *Y, *H1, *Ymean, *H1mean;
  int Nk, nh; float scale; DropCfg drop;
  int debug_stop;                        // developer aid: return after stage N (0 = run everything)
};

int rg_fused_supported(int D, int H, int nh, int Nk);
size_t rg_fused_lds_bytes();
int launch_cast_tiled_bf16(const float* const* src, unsigned short* const* dst, const int* N, const int* K, int count,
                           hipStream_t stream);
int launch_rg_forward_fused(const RgFwdArgs& a, int B, int max_nr, hipStream_t stream);
