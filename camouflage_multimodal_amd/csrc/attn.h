// Launchers of the cross-attention cores (attn.hip).  All return hipError_t as int.
#pragma once
#include "common.h"

int attn_supported(int H, int nh, int Nk);

// Optional bf16 destination of an attention output (the bf16 schedule): when p != null the values go there
// (row pitch ld elements, column offset folded into p) INSTEAD of the fp32 tensor.
struct Bf16Dst { unsigned short* p; int ld; };

int launch_attn_rg2kg_fwd(const float* Q, const float* KV, const int* offs, float* P, float* O, float* attn_avg,
                          int B, int T, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_attn_rg2kg_bwd(const float* Q, const float* KV, const float* P, const float* dO, const int* offs,
                          float* dQ, float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop,
                          hipStream_t stream);
int launch_attn_kg2rg_fwd(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2,
                          int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
// O2: the forward's attention output of this direction ([B*Nk, H], fp32); the MFMA kernel takes its softmax row-dots
// from dO2 . O2 instead of a pass over the keys
int launch_attn_kg2rg_bwd(const float* Q2, const float* KV2, const float* P2, const float* dO2, const float* O2, const int* offs,
                          float* dQ2, float* dKV2, float* dS2, int B, int max_nr, int H, int nh, int Nk, DropCfg drop,
                          hipStream_t stream);
int launch_attn_avg(const float* P2, float* out, int T, int nh, int Nk, DropCfg drop, hipStream_t stream);

// head_dim == 32 fast paths (attn_fast.hip); the launchers above dispatch to them when attn_fast_ok().
int attn_fast_ok(int H, int nh, int Nk, int max_nr, bool kg2rg, bool bwd);
int launch_rg2kg_fwd32(const float* Q, const float* KV, const int* offs, float* P, float* O, float* attn_avg,
                       int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_rg2kg_bwd32(const float* Q, const float* KV, const float* P, const float* dO, const int* offs, float* dQ,
                       float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_kg2rg_fwd32(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2, int B, int max_nr,
                       int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_kg2rg_bwd32(const float* Q2, const float* KV2, const float* P2, const float* dO2, const int* offs, float* dQ2,
                       float* dKV2, int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);

// exact-fp32 MFMA versions (attn_mfma.hip): head_dim == 32, Nk <= 16 (kg2rg: Nr <= 768)
int attn_mfma_ok(int H, int nh, int Nk, int max_nr, bool kg2rg);
int launch_rg2kg_fwd_mfma(const float* Q, const float* KV, const int* offs, float* P, float* O, float* attn_avg,
                          int B, int T, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream,
                          Bf16Dst o16 = Bf16Dst{nullptr, 0});
int launch_rg2kg_bwd_mfma(const float* Q, const float* KV, const float* P, const float* dO, const int* offs, float* dQ,
                          float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream,
                          Bf16Dst dq16 = Bf16Dst{nullptr, 0});
int launch_kg2rg_fwd_mfma(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2, int B, int H, int nh,
                          int Nk, DropCfg drop, hipStream_t stream, Bf16Dst o16 = Bf16Dst{nullptr, 0});
// dq2_16 / dkv2_16: bf16 destinations of dQ2 and dK2|dV2 (V at column offset H); dkv_16: also convert the
// finished fp32 dK|dV of the other block (dKV_done, [B*Nk, 2H]) to bf16
int launch_kg2rg_bwd_mfma(const float* Q2, const float* KV2, const float* P2, const float* dO2, const float* O2, const int* offs,
                          float* dQ2, float* dKV2, int B, int H, int nh, int Nk, DropCfg drop, hipStream_t stream,
                          Bf16Dst dq2_16 = Bf16Dst{nullptr, 0}, Bf16Dst dkv2_16 = Bf16Dst{nullptr, 0},
                          const float* dKV_done = nullptr, Bf16Dst dkv_16 = Bf16Dst{nullptr, 0});
// both directions in one launch (bf16 schedule; needs attn_mfma_ok for both): forward writes P, P2 and the bf16
// outputs; backward writes every gradient as the bf16 GEMM operand (dK|dV at column offset 0 / H of dkv16, dkv2_16)
int launch_attn_fwd_pair(const float* Q, const float* KV, const float* Q2, const float* KV2, const int* offs, float* P,
                         float* P2, Bf16Dst o16, Bf16Dst o2_16, float* O2, int B, int max_nr, int H, int nh, int Nk, DropCfg drop,
                         hipStream_t stream);
int launch_attn_bwd_pair(const float* Q, const float* KV, const float* P, const float* dO, const float* Q2, const float* KV2,
                         const float* P2, const float* dO2, const int* offs, Bf16Dst dq16, Bf16Dst dkv16, Bf16Dst dq2_16,
                         Bf16Dst dkv2_16, const float* O2, int B, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_attn_avg_site(const float* Pm, float* out, int T, int nh, int Nk, uint32_t site, DropCfg drop, hipStream_t stream);
