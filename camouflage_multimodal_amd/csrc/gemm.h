// Grouped GEMM with fused epilogues: one launch runs up to GEMM_MAXP independent
// problems (different pointers / shapes / operand layouts / epilogues).
#pragma once
#include "common.h"

#define GEMM_MAXP 12

enum : int {
  GF_RELU = 1,        // v = max(v, 0)                       (after bias)
  GF_DROPOUT = 2,     // v *= keep(site, row*N+col)/(1-p)
  GF_ATOMIC = 4,      // C += v with atomicAdd (split-K partials, shared accumulators)
  GF_RELU_BWD = 8,    // v *= (aux[row,col] > 0 ? aux_scale : 0); aux = `res` pointer
  GF_RES_BCAST = 16,  // residual row = res[sample(row), col] * inv_n(sample(row)); sample(): see GemmProb
  GF_SIGMOID = 32,    // v = 1/(1+exp(-v))                   (last)
  GF_A_KMAJOR = 64,   // A element (m,k) at A[k*lda + m] instead of A[m*lda + k]
  GF_B_KMAJOR = 128,  // B element (k,n) at B[k*ldb + n] instead of B[n*ldb + k]
  GF_A_VIRT = 256     // gemm16 only: A is the ReLU/dropout mask-broadcast gradient built on the fly (Gemm16Prob::virt_g)
};

// C[m,n] (+)= epi( sum_k A(m,k) * B(k,n) + bias[n] ) (+ res[m,n])
struct GemmProb {
  const float* A; const float* B; float* C;
  const float* bias;      // [N] or null
  const float* res;       // residual / aux, ld = ldr, or null
  float* bias_grad;       // [M] += sum_k A(m,k)   (A k-major only; the bias gradient of a dW GEMM) or null
  int M, N, K;
  int lda, ldb, ldc, ldr;
  int flags;
  uint32_t drop_site;
  float aux_scale;
  // GF_RES_BCAST: sample(row) = row_sample[row], inv_n = inv_nr[sample]; or, when row_sample is
  // null, sample(row) = row / uniform_n and inv_n = 1/uniform_n
  const int* row_sample; const float* inv_nr; int uniform_n;
  // filled by the launcher
  int tiles_n, ksplit, kchunk, tile_begin;
};

struct GemmBatch {
  GemmProb p[GEMM_MAXP];
  int n;
  DropCfg drop;
  int kcap;          // > 0: split-K depth (32-deep tiles per block) of the dW problems of this launch, whatever their K (0: 12 tiles, only past 16)
};

// precision: CAMO_PREC_F32 / CAMO_PREC_BF16.  Returns hipError_t as int.
int launch_gemm_batch(GemmBatch& gb, int precision, hipStream_t stream);

// ---- opt-in launch timing (bench.py's roofline leg): HIP events recorded on the launch stream
// around every grouped-GEMM launch between prof_begin and prof_end.  Single-threaded use only.
int gemm_prof_begin(int max_launches);
int gemm_prof_end(double* total_ms, int* launches, double* total_flops);
// used by the launchers of both grouped-GEMM kernels: returns a slot (or -1 when timing is off)
enum : int { PROF_GEMM = 0, PROF_FRONT = 1, PROF_BACK = 2, PROF_BWD1 = 3, PROF_BWD2 = 4, PROF_TAIL = 5, PROF_OPT = 6, PROF_SHADOW = 7,
             PROF_ATTN = 8, PROF_OTHER = 9, PROF_KINDS = 10 };
int gemm_prof_open(hipStream_t stream, double flops, int kind = PROF_GEMM);
int gemm_prof_kind(int kind, double* ms, int* launches, double* flops);
void gemm_prof_close(int slot, hipStream_t stream);
