// Launchers of the cross-attention cores (attn.hip).  All return hipError_t as int.
#pragma once
#include "common.h"

int attn_supported(int H, int nh, int Nk);

int launch_attn_rg2kg_fwd(const float* Q, const float* KV, const int* offs, float* P, float* O, float* attn_avg,
                          int B, int T, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_attn_rg2kg_bwd(const float* Q, const float* KV, const float* P, const float* dO, const int* offs,
                          float* dQ, float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop,
                          hipStream_t stream);
int launch_attn_kg2rg_fwd(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2,
                          int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_attn_kg2rg_bwd(const float* Q2, const float* KV2, const float* P2, const float* dO2, const int* offs,
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
                          int B, int T, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_rg2kg_bwd_mfma(const float* Q, const float* KV, const float* P, const float* dO, const int* offs, float* dQ,
                          float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_kg2rg_fwd_mfma(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2, int B, int H, int nh,
                          int Nk, DropCfg drop, hipStream_t stream);
int launch_kg2rg_bwd_mfma(const float* Q2, const float* KV2, const float* P2, const float* dO2, const int* offs, float* dQ2,
                          float* dKV2, int B, int H, int nh, int Nk, DropCfg drop, hipStream_t stream);
int launch_attn_avg_site(const float* Pm, float* out, int T, int nh, int Nk, uint32_t site, DropCfg drop, hipStream_t stream);
