// Grouped GEMM on bf16-RESIDENT operands (gfx950): the node-level GEMMs of the bf16 schedule.
//
// gemm.hip reads fp32 operands and rounds them to bf16 while staging; on MI355X its K loop is bound by
// that VALU work (mask + convert + pack: 100-150 vector instructions per 32-deep step per wave at 4
// cycles each against 4 MFMAs).  Here every operand already sits in HBM as bf16 -- producers emit a bf16
// copy of what a GEMM will read (fusion_abi.hip, "schedule16") -- so a K step is loads, LDS traffic and
// MFMAs only, and the operand bytes read from HBM halve.
//
// Supported per problem (checked by the launcher):
//   NT : C[M,N] = epi(A[M,K] . B[N,K]^T)        A, B row-major bf16, K % 64 == 0        (x.W^T, dy.(W^T)^T)
//   TN : C[M,N] += A[K,M]^T . B[K,N]  (fp32 atomics, split-K)   A, B row-major bf16 with K rows; rows up to
//        round_up(K, 128) must be readable, and zero in one operand / finite in the other (the workspace
//        pads and clears them); M % 8 == 0, N % 8 == 0                                    (dW += dy^T.x)
// N % 4 == 0, 16-byte aligned pointers and leading dimensions throughout.
#pragma once
#include "common.h"
#include "gemm.h"   // GF_* flags

#define GEMM16_MAXP 12

struct Gemm16Prob {
  const unsigned short* A; const unsigned short* B;   // bf16 bit patterns
  float* C;                 // fp32 output (or null)
  unsigned short* C16;      // bf16 copy of the output (or null); NT only
  const float* bias;        // [N] or null
  const float* res;         // fp32 residual / aux (ld = ldr) or null
  float* bias_grad;         // TN: [M] += sum_k A(k, m)
  float* bias_grad2;        // TN: a second destination of the same sums (or null)
  int M, N, K;
  int lda, ldb, ldc, ldc16, ldr;
  int flags;                // GF_RELU | GF_DROPOUT | GF_RELU_BWD | GF_RES_BCAST; GF_A_KMAJOR|GF_B_KMAJOR together = TN
  uint32_t drop_site;
  float aux_scale;
  const int* row_sample; const float* inv_nr; int uniform_n;
  // NT only: colmean[sample(row)][col] += out(row, col) * inv_n(sample(row))  -- the per-sample mean pool of the
  // result, accumulated with atomics into a zeroed [samples][ldm] buffer (sample(): as for GF_RES_BCAST)
  float* colmean; int ldm;
  // NT only, N == 256: a LayerNorm over the output rows fused into the epilogue (the block owns whole rows).
  //   ln_mode 1 (forward):  u = A.B^T + bias + res  -> C (fp32);  y = LN(u) * ln_gamma + ln_beta -> C16 (bf16), its
  //                         per-sample mean pool -> colmean, (mean, rstd) of every row -> ln_stats
  //   ln_mode 2 (backward): dy = A.B^T (+ broadcast residual); with x = ln_x (the forward's u) and ln_stats:
  //                         du = LN_backward(dy) -> C (fp32) and C16 (bf16); ln_dgamma += sum dy*xhat, ln_dbeta += sum dy
  int ln_mode;
  const float* ln_gamma; const float* ln_beta; float* ln_stats; const float* ln_x; float* ln_dgamma; float* ln_dbeta;
  // GF_A_VIRT (whole-row NT and TN problems of a launch that contains whole-row problems): the A operand is not read
  // but built while staging:  A[t][c] = act[t][c] > 0 ? virt_g[sample(t)][c] * inv_n(sample(t)) * aux_scale : 0,
  // act = the bf16 tensor passed as A (the forward's post-ReLU/dropout activation, rows padded and cleared like any
  // TN operand), rounded to bf16 exactly as relu_bcast_bwd would have written it.  sample(): as for GF_RES_BCAST.
  const float* virt_g; int ldg;
  // filled by the launcher
  int tiles_n, ksplit, kchunk, tile_begin;
};

struct Gemm16Batch {
  Gemm16Prob p[GEMM16_MAXP];
  int n;
  DropCfg drop;
  int exp;           // developer experiments (0 in product calls)
  int n_heavy;       // (filled by the launcher) > 0: tiles [0, n_heavy) are split-K blocks with long K loops, the rest short ones
};

// Returns hipError_t as int; hipErrorInvalidValue for an unsupported problem.
extern thread_local int g_gemm16_exp;
extern thread_local int g_gemm16_tn_big;          // developer A/B: -1 by size, 0 never, 1 always (weight-gradient-only launches)
extern thread_local int g_gemm16_balance;         // developer A/B: 0 = plain contiguous XCD remap
extern thread_local int g_gemm16_tn_kcap;         // developer A/B: > 0 pins the split-K depth (64-row tiles per block) of weight-gradient problems
int launch_gemm16_batch(Gemm16Batch& gb, hipStream_t stream);
