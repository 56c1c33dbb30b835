// head_dim = 32 specialisations of the cross-attention cores (the reference configuration:
// hidden 256 / 8 heads).  Same math and layouts as attn.hip; what changes is how the work is
// fed: compile-time 32-wide inner loops held in registers, 16-B LDS reads (ds_read_b128) of K/V
// rows whose 144-B pitch makes a 16-lane read group hit 16 distinct 16-B slots, probabilities
// padded to 16 floats so a row is four b128 reads, and no runtime-length dependent loops on the
// critical path (those were latency-bound at 1-2 waves per SIMD).
//
// Conditions (checked by the launchers): head_dim == 32, Nk <= 16, num_heads a power of two <= 64.
#include "attn.h"

namespace {

constexpr int DH = 32, PITCH = 36, NKP = 16;

// The j-loops below are fully unrolled so that the per-lane arrays stay in registers; left alone,
// hipcc then hoists all 16 x 8 LDS reads of a loop to its top (> 256 VGPRs, spills).  A compiler
// memory fence every second iteration bounds the reads in flight to 16 x ds_read_b128.
#define J_FENCE(j) do { if ((j) & 1) asm volatile("" ::: "memory"); } while (0)

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
  acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
  return acc;
}
__device__ __forceinline__ void axpy4(float4& y, const float a, const float4 x) {
  y.x = fmaf(a, x.x, y.x); y.y = fmaf(a, x.y, y.y); y.z = fmaf(a, x.z, y.z); y.w = fmaf(a, x.w, y.w);
}

// stage K|V of sample b into LDS: Ks/Vs[(j*nh + h)*PITCH + d]
__device__ __forceinline__ void stage_kv(const float* __restrict__ KV, int b, int H, int nh, int Nk, float* Ks, float* Vs) {
  const int q4 = H >> 2;   // float4 per row half
  for (int i = threadIdx.x; i < Nk * q4; i += 256) {
    const int j = i / q4, c = (i - j * q4) * 4;
    const float* g = KV + (size_t)(b * Nk + j) * (2 * H) + c;
    const int o = (j * nh + (c >> 5)) * PITCH + (c & 31);
    *reinterpret_cast<float4*>(Ks + o) = *reinterpret_cast<const float4*>(g);
    *reinterpret_cast<float4*>(Vs + o) = *reinterpret_cast<const float4*>(g + H);
  }
}

// ------------------------------------------------------------------ rg2kg forward
__global__ __launch_bounds__(256) void rg2kg_fwd32_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const int* __restrict__ offs,
    float* __restrict__ P, float* __restrict__ O, float* __restrict__ attn_avg,
    int H, int nh, int Nk, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int RB = 256 / nh;
  const int row0 = blockIdx.x * RB;
  if (row0 >= nr) return;
  float* Ks = sm;
  float* Vs = Ks + Nk * nh * PITCH;
  stage_kv(KV, b, H, nh, Nk, Ks, Vs);
  const int tid = threadIdx.x;
  const int r = tid / nh, hh = tid & (nh - 1);
  const bool active = row0 + r < nr;
  const int t = r0 + min(row0 + r, nr - 1);
  float4 q[8];
  {
    const float4* qp = reinterpret_cast<const float4*>(Q + (size_t)t * H + hh * DH);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = qp[i];
  }
  __syncthreads();
  float s[NKP];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NKP; ++j) {
    s[j] = -INFINITY;
    if (j < Nk) {
      const float4* k = reinterpret_cast<const float4*>(Ks + (j * nh + hh) * PITCH);
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) a = dot4(q[i], k[i], a);
      s[j] = a * scale;
      mx = fmaxf(mx, s[j]);
    }
    J_FENCE(j);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NKP; ++j) { s[j] = (j < Nk) ? __expf(s[j] - mx) : 0.f; sum += s[j]; }
  const float inv = 1.0f / sum;
  const size_t pbase = ((size_t)t * nh + hh) * Nk;
#pragma unroll
  for (int j = 0; j < NKP; ++j) {
    if (j < Nk) {
      const float p = s[j] * inv;
      if (active) P[pbase + j] = p;
      s[j] = drop.p > 0.f ? p * drop_mult(drop, SITE_ATTN_RG2KG, (uint32_t)(pbase + j)) : p;
    }
  }
  if (attn_avg) {   // mean over the nh consecutive lanes that hold one node's heads
    const float invh = 1.0f / (float)nh;
#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      if (j < Nk) {
        float a = s[j];
        for (int o = 1; o < nh; o <<= 1) a += __shfl_xor(a, o, 64);
        if (active && hh == 0) attn_avg[(size_t)t * Nk + j] = a * invh;
      }
    }
  }
  float4 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NKP; ++j) {
    if (j < Nk) {
      const float4* v = reinterpret_cast<const float4*>(Vs + (j * nh + hh) * PITCH);
#pragma unroll
      for (int i = 0; i < 8; ++i) axpy4(o[i], s[j], v[i]);
    }
    J_FENCE(j);
  }
  if (active) {
    float4* op = reinterpret_cast<float4*>(O + (size_t)t * H + hh * DH);
#pragma unroll
    for (int i = 0; i < 8; ++i) op[i] = o[i];
  }
}

// ------------------------------------------------------------------ rg2kg backward
// grid (chunks, B): a block walks row chunks c = blockIdx.x, += gridDim.x, keeps the dK/dV column
// partials of all its chunks in registers and flushes them with one atomicAdd per element.
__global__ __launch_bounds__(256) void rg2kg_bwd32_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const float* __restrict__ P,
    const float* __restrict__ dO, const int* __restrict__ offs,
    float* __restrict__ dQ, float* __restrict__ dKV,
    int H, int nh, int Nk, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int RB = 256 / nh;
  if ((int)blockIdx.x * RB >= nr) return;
  float* Ks = sm;
  float* Vs = Ks + Nk * nh * PITCH;
  float* dSs = Vs + Nk * nh * PITCH;     // [256][16]  dS*scale of (row, head) pairs
  float* Pds = dSs + 256 * NKP;          // [256][16]  dropped probabilities
  stage_kv(KV, b, H, nh, Nk, Ks, Vs);
  const int tid = threadIdx.x;
  const int r = tid / nh, hh = tid & (nh - 1);
  float accK[NKP], accV[NKP];
#pragma unroll
  for (int j = 0; j < NKP; ++j) { accK[j] = 0.f; accV[j] = 0.f; }
  __syncthreads();
  for (int row0 = blockIdx.x * RB; row0 < nr; row0 += gridDim.x * RB) {
    const int rows = min(RB, nr - row0);
    const bool active = r < rows;
    const int t = r0 + min(row0 + r, nr - 1);
    // ---- phase 1: one lane per (row, head)
    {
      float4 g[8];
      const float4* gp = reinterpret_cast<const float4*>(dO + (size_t)t * H + hh * DH);
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = gp[i];
      const size_t pbase = ((size_t)t * nh + hh) * Nk;
      float p[NKP], ds[NKP], pd[NKP];
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < NKP; ++j) {
        p[j] = 0.f; ds[j] = 0.f; pd[j] = 0.f;
        if (j < Nk) {
          const float4* v = reinterpret_cast<const float4*>(Vs + (j * nh + hh) * PITCH);
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) a = dot4(g[i], v[i], a);
          p[j] = P[pbase + j];
          const float m = drop.p > 0.f ? drop_mult(drop, SITE_ATTN_RG2KG, (uint32_t)(pbase + j)) : 1.0f;
          pd[j] = p[j] * m;
          ds[j] = a * m;
          dot = fmaf(p[j], ds[j], dot);
        }
        J_FENCE(j);
      }
#pragma unroll
      for (int j = 0; j < NKP; ++j) {
        ds[j] = (j < Nk && active) ? p[j] * (ds[j] - dot) * scale : 0.f;
        if (!active) pd[j] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float4*>(dSs + tid * NKP + 4 * i) = make_float4(ds[4 * i], ds[4 * i + 1], ds[4 * i + 2], ds[4 * i + 3]);
        *reinterpret_cast<float4*>(Pds + tid * NKP + 4 * i) = make_float4(pd[4 * i], pd[4 * i + 1], pd[4 * i + 2], pd[4 * i + 3]);
      }
      float4 o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < NKP; ++j) {
        if (j < Nk) {
          const float4* k = reinterpret_cast<const float4*>(Ks + (j * nh + hh) * PITCH);
#pragma unroll
          for (int i = 0; i < 8; ++i) axpy4(o[i], ds[j], k[i]);
        }
        J_FENCE(j);
      }
      if (active) {
        float4* qp = reinterpret_cast<float4*>(dQ + (size_t)t * H + hh * DH);
#pragma unroll
        for (int i = 0; i < 8; ++i) qp[i] = o[i];
      }
    }
    __syncthreads();
    // ---- phase 2: one thread per column c = (h, d): dK/dV partials over the chunk's rows
    for (int c = tid; c < H; c += 256) {      // H <= 256 in practice: one pass, accK/accV stay per column
      const int hc = c >> 5;
      const float* qcol = Q + (size_t)(r0 + row0) * H + c;
      const float* gcol = dO + (size_t)(r0 + row0) * H + c;
#pragma unroll 4
      for (int rr = 0; rr < rows; ++rr) {
        const float qv = qcol[(size_t)rr * H], gv = gcol[(size_t)rr * H];
        const float4* d4 = reinterpret_cast<const float4*>(dSs + (rr * nh + hc) * NKP);
        const float4* p4 = reinterpret_cast<const float4*>(Pds + (rr * nh + hc) * NKP);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 dv = d4[i], pv = p4[i];
          accK[4 * i] = fmaf(dv.x, qv, accK[4 * i]); accK[4 * i + 1] = fmaf(dv.y, qv, accK[4 * i + 1]);
          accK[4 * i + 2] = fmaf(dv.z, qv, accK[4 * i + 2]); accK[4 * i + 3] = fmaf(dv.w, qv, accK[4 * i + 3]);
          accV[4 * i] = fmaf(pv.x, gv, accV[4 * i]); accV[4 * i + 1] = fmaf(pv.y, gv, accV[4 * i + 1]);
          accV[4 * i + 2] = fmaf(pv.z, gv, accV[4 * i + 2]); accV[4 * i + 3] = fmaf(pv.w, gv, accV[4 * i + 3]);
        }
      }
    }
    __syncthreads();
  }
  if (tid < H) {
#pragma unroll
    for (int j = 0; j < NKP; ++j)
      if (j < Nk) {
        float* dst = dKV + (size_t)(b * Nk + j) * (2 * H) + tid;
        atomicAdd(dst, accK[j]);
        atomicAdd(dst + H, accV[j]);
      }
  }
}

// block-wide reductions of NKP values held per thread (256 threads = 4 waves)
template <bool IS_MAX>
__device__ __forceinline__ void block_reduce16(float (&v)[NKP], int Nk, float* red /*[4][NKP]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NKP; ++j)
    if (j < Nk) {
      const float w = IS_MAX ? wave_max(v[j]) : wave_sum(v[j]);
      if (lane == 0) red[wave * NKP + j] = w;
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NKP; ++j)
    if (j < Nk) {
      const float a = red[j], b = red[NKP + j], c = red[2 * NKP + j], d = red[3 * NKP + j];
      v[j] = IS_MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
    }
  __syncthreads();
}

// [Nk x 32] = sum over keys t of w[t][0..15] (LDS rows) times X[t][col0 + d] (global, row stride ldx):
// thread = (key group g of 8, d); partial sums reduced through `part`.  out[j*ldo + d].
__device__ __forceinline__ void keys_contract(const float* __restrict__ wrows, const float* __restrict__ X, size_t ldx,
                                              int nr, int Nk, float* part /*[8][16][32]*/, float* __restrict__ out, int ldo) {
  const int tid = threadIdx.x, g = tid >> 5, d = tid & 31;
  float acc[NKP];
#pragma unroll
  for (int j = 0; j < NKP; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int t = g; t < nr; t += 8) {
    const float x = X[(size_t)t * ldx + d];
    const float4* w4 = reinterpret_cast<const float4*>(wrows + t * NKP);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 w = w4[i];
      acc[4 * i] = fmaf(w.x, x, acc[4 * i]); acc[4 * i + 1] = fmaf(w.y, x, acc[4 * i + 1]);
      acc[4 * i + 2] = fmaf(w.z, x, acc[4 * i + 2]); acc[4 * i + 3] = fmaf(w.w, x, acc[4 * i + 3]);
    }
  }
#pragma unroll
  for (int j = 0; j < NKP; ++j) part[(g * NKP + j) * DH + d] = acc[j];
  __syncthreads();
  for (int i = tid; i < Nk * DH; i += 256) {
    const int j = i >> 5, dd = i & 31;
    float a = 0.f;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) a += part[(gg * NKP + j) * DH + dd];
    out[(size_t)j * ldo + dd] = a;
  }
}

// ------------------------------------------------------------------ kg2rg forward, grid (nh, B)
// LDS: qs [16][32] | red [4][16] | part [8][16][32] | Ss [nr][16]
__global__ __launch_bounds__(256) void kg2rg_fwd32_kernel(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const int* __restrict__ offs,
    float* __restrict__ P2, float* __restrict__ O2, int H, int nh, int Nk, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int hh = blockIdx.x, b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  float* qs = sm;
  float* red = qs + NKP * DH;
  float* part = red + 4 * NKP;
  float* Ss = part + 8 * NKP * DH;
  const int tid = threadIdx.x;
  for (int i = tid; i < NKP * DH; i += 256) {
    const int j = i >> 5, d = i & 31;
    qs[i] = j < Nk ? Q2[(size_t)(b * Nk + j) * H + hh * DH + d] * scale : 0.f;
  }
  __syncthreads();
  float mx[NKP];
#pragma unroll
  for (int j = 0; j < NKP; ++j) mx[j] = -INFINITY;
  for (int t = tid; t < nr; t += 256) {
    float4 k[8];
    const float4* kp = reinterpret_cast<const float4*>(KV2 + (size_t)(r0 + t) * (2 * H) + hh * DH);
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kp[i];
    float s[NKP];
#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      const float4* q = reinterpret_cast<const float4*>(qs + j * DH);
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) a = dot4(k[i], q[i], a);
      s[j] = a;
      mx[j] = fmaxf(mx[j], a);
      J_FENCE(j);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(Ss + t * NKP + 4 * i) = make_float4(s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
  }
  block_reduce16<true>(mx, Nk, red);
  float sum[NKP];
#pragma unroll
  for (int j = 0; j < NKP; ++j) sum[j] = 0.f;
  for (int t = tid; t < nr; t += 256) {
#pragma unroll
    for (int j = 0; j < NKP; ++j)
      if (j < Nk) { const float e = __expf(Ss[t * NKP + j] - mx[j]); Ss[t * NKP + j] = e; sum[j] += e; }
  }
  block_reduce16<false>(sum, Nk, red);
  for (int t = tid; t < nr; t += 256) {
    const size_t pbase = ((size_t)(r0 + t) * nh + hh) * Nk;
#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      float pd = 0.f;
      if (j < Nk) {
        const float p = Ss[t * NKP + j] / sum[j];
        P2[pbase + j] = p;
        pd = drop.p > 0.f ? p * drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)(pbase + j)) : p;
      }
      Ss[t * NKP + j] = pd;
    }
  }
  __syncthreads();
  keys_contract(Ss, KV2 + (size_t)r0 * (2 * H) + H + hh * DH, (size_t)2 * H, nr, Nk, part,
                O2 + (size_t)b * Nk * H + hh * DH, H);
}

// ------------------------------------------------------------------ kg2rg backward, grid (nh, B)
// LDS: qs [16][32] | gs [16][32] | red | part | Ds [nr][16] | Ps [nr][16]
__global__ __launch_bounds__(256) void kg2rg_bwd32_kernel(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const float* __restrict__ P2,
    const float* __restrict__ dO2, const int* __restrict__ offs,
    float* __restrict__ dQ2, float* __restrict__ dKV2, int H, int nh, int Nk, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int hh = blockIdx.x, b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  float* qs = sm;
  float* gs = qs + NKP * DH;
  float* red = gs + NKP * DH;
  float* part = red + 4 * NKP;
  float* Ds = part + 8 * NKP * DH;
  float* Ps = Ds + (size_t)nr * NKP;
  const int tid = threadIdx.x;
  for (int i = tid; i < NKP * DH; i += 256) {
    const int j = i >> 5, d = i & 31;
    const size_t o = (size_t)(b * Nk + j) * H + hh * DH + d;
    qs[i] = j < Nk ? Q2[o] : 0.f;
    gs[i] = j < Nk ? dO2[o] : 0.f;
  }
  __syncthreads();
  float dot[NKP];
#pragma unroll
  for (int j = 0; j < NKP; ++j) dot[j] = 0.f;
  for (int t = tid; t < nr; t += 256) {
    float4 v[8];
    const float4* vp = reinterpret_cast<const float4*>(KV2 + (size_t)(r0 + t) * (2 * H) + H + hh * DH);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = vp[i];
    const size_t pbase = ((size_t)(r0 + t) * nh + hh) * Nk;
    float dp[NKP], pd[NKP], p[NKP];
#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      dp[j] = 0.f; pd[j] = 0.f; p[j] = 0.f;
      if (j < Nk) {
        const float4* g = reinterpret_cast<const float4*>(gs + j * DH);
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a = dot4(v[i], g[i], a);
        p[j] = P2[pbase + j];
        const float m = drop.p > 0.f ? drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)(pbase + j)) : 1.0f;
        dp[j] = a * m; pd[j] = p[j] * m;
        dot[j] = fmaf(p[j], dp[j], dot[j]);
      }
      J_FENCE(j);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(Ds + t * NKP + 4 * i) = make_float4(dp[4 * i], dp[4 * i + 1], dp[4 * i + 2], dp[4 * i + 3]);
      *reinterpret_cast<float4*>(Ps + t * NKP + 4 * i) = make_float4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
    }
    // dV2[t] = sum_j Pd[t][j] dO2[j]
    float4 o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      const float4* g = reinterpret_cast<const float4*>(gs + j * DH);
#pragma unroll
      for (int i = 0; i < 8; ++i) axpy4(o[i], pd[j], g[i]);
      J_FENCE(j);
    }
    float4* dv = reinterpret_cast<float4*>(dKV2 + (size_t)(r0 + t) * (2 * H) + H + hh * DH);
#pragma unroll
    for (int i = 0; i < 8; ++i) dv[i] = o[i];
  }
  block_reduce16<false>(dot, Nk, red);
  for (int t = tid; t < nr; t += 256) {
    float ds[NKP];
#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      ds[j] = (j < Nk) ? Ps[t * NKP + j] * (Ds[t * NKP + j] - dot[j]) * scale : 0.f;
      Ds[t * NKP + j] = ds[j];
    }
    // dK2[t] = sum_j dS[t][j] q[j]
    float4 o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      const float4* q = reinterpret_cast<const float4*>(qs + j * DH);
#pragma unroll
      for (int i = 0; i < 8; ++i) axpy4(o[i], ds[j], q[i]);
      J_FENCE(j);
    }
    float4* dk = reinterpret_cast<float4*>(dKV2 + (size_t)(r0 + t) * (2 * H) + hh * DH);
#pragma unroll
    for (int i = 0; i < 8; ++i) dk[i] = o[i];
  }
  __syncthreads();
  keys_contract(Ds, KV2 + (size_t)r0 * (2 * H) + hh * DH, (size_t)2 * H, nr, Nk, part,
                dQ2 + (size_t)b * Nk * H + hh * DH, H);
}

}  // namespace

// ------------------------------------------------------------------ launchers
static inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
static inline size_t rg2kg32_lds(int nh, int Nk, bool bwd) {
  return sizeof(float) * ((size_t)2 * Nk * nh * PITCH + (bwd ? (size_t)2 * 256 * NKP : 0));
}
static inline size_t kg2rg32_lds(int max_nr, bool bwd) {
  return sizeof(float) * ((size_t)(bwd ? 2 : 1) * NKP * DH + 4 * NKP + (size_t)8 * NKP * DH + (size_t)(bwd ? 2 : 1) * max_nr * NKP);
}
constexpr size_t FAST_LDS_BUDGET = 150 * 1024;

int attn_fast_ok(int H, int nh, int Nk, int max_nr, bool kg2rg, bool bwd) {
  if (nh < 1 || H != nh * DH || Nk < 1 || Nk > NKP || !pow2(nh) || nh > 64 || (H & 3)) return 0;
  if (kg2rg) return kg2rg32_lds(max_nr, bwd) <= FAST_LDS_BUDGET;
  if (bwd && H > 256) return 0;   // the dK/dV column pass keeps one column per thread
  return rg2kg32_lds(nh, Nk, bwd) <= FAST_LDS_BUDGET;
}

#define SET_LDS(KERN, LDS) \
  do { if ((LDS) > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); } while (0)

int launch_rg2kg_fwd32(const float* Q, const float* KV, const int* offs, float* P, float* O, float* attn_avg,
                       int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  const int RB = 256 / nh;
  const size_t lds = rg2kg32_lds(nh, Nk, false);
  SET_LDS(rg2kg_fwd32_kernel, lds);
  hipLaunchKernelGGL(rg2kg_fwd32_kernel, dim3((max_nr + RB - 1) / RB, B), dim3(256), lds, stream, Q, KV, offs, P, O, attn_avg,
                     H, nh, Nk, 1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

int launch_rg2kg_bwd32(const float* Q, const float* KV, const float* P, const float* dO, const int* offs, float* dQ,
                       float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  const int RB = 256 / nh;
  const int chunks = (max_nr + RB - 1) / RB;
  // two chunks per block halve the dK/dV atomics while still filling the chip at B >= 16
  const int per = (B * chunks >= 512) ? 2 : 1;
  const size_t lds = rg2kg32_lds(nh, Nk, true);
  SET_LDS(rg2kg_bwd32_kernel, lds);
  hipLaunchKernelGGL(rg2kg_bwd32_kernel, dim3((chunks + per - 1) / per, B), dim3(256), lds, stream, Q, KV, P, dO, offs, dQ, dKV,
                     H, nh, Nk, 1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

int launch_kg2rg_fwd32(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2, int B, int max_nr,
                       int H, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  const size_t lds = kg2rg32_lds(max_nr, false);
  SET_LDS(kg2rg_fwd32_kernel, lds);
  hipLaunchKernelGGL(kg2rg_fwd32_kernel, dim3(nh, B), dim3(256), lds, stream, Q2, KV2, offs, P2, O2, H, nh, Nk,
                     1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

int launch_kg2rg_bwd32(const float* Q2, const float* KV2, const float* P2, const float* dO2, const int* offs, float* dQ2,
                       float* dKV2, int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  const size_t lds = kg2rg32_lds(max_nr, true);
  SET_LDS(kg2rg_bwd32_kernel, lds);
  hipLaunchKernelGGL(kg2rg_bwd32_kernel, dim3(nh, B), dim3(256), lds, stream, Q2, KV2, P2, dO2, offs, dQ2, dKV2, H, nh, Nk,
                     1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}
