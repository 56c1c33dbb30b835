// Cross-attention cores for tiny key counts (gfx950).
//
//   rg2kg: every RG node (query) attends to the Nk (13) KG rows of its sample.
//          A row's softmax is Nk wide, so one LANE owns one (node, head) pair: the
//          scores live in registers, no cross-lane traffic, no MFMA (S=13 is far too
//          small for a matrix tile).  K/V of the sample and the block's Q rows are staged
//          in LDS with a (dh+1)-float pitch per (row, head) so that the 64 lanes of a
//          wave -- consecutive (node, head) pairs -- hit 64 different banks.
//   kg2rg: the Nk KG rows (queries) attend to the Nr RG nodes of the sample; the softmax
//          runs over Nr (303..530, any length supported).  One workgroup per (head, sample):
//          lanes stride over keys, block-wide max / sum reductions, then a [Nk x dh] = P^T.V
//          contraction with lanes on the dh axis.
//
// Layouts (all fp32): Q [T,H]; KV [B*Nk, 2H] (K | V); P [T, nh, Nk] (softmax output BEFORE
// dropout); O [T,H].  For kg2rg: Q2 [B*Nk,H]; KV2 [T,2H]; P2 [T, nh, Nk] with P2[t,h,j] =
// weight of query j on key t; O2 [B*Nk,H].  Dropout on the probabilities is regenerated from
// the counter hash (element index (t*nh+h)*Nk+j), never stored.
#include "attn.h"

namespace {


// ------------------------------------------------------------------ rg2kg forward
// grid (ceil(max_nr / RB), B); RB <= 256 / nh rows per block (launcher shrinks it to fit LDS).
template <int NKMAX>
__global__ __launch_bounds__(256) void attn_rg2kg_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const int* __restrict__ offs,
    float* __restrict__ P, float* __restrict__ O, float* __restrict__ attn_avg,
    int H, int nh, int Nk, int RB, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int dh = H / nh, pitch = dh + 1;
  const int b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int row0 = blockIdx.x * RB;
  if (row0 >= nr) return;
  const int rows = min(RB, nr - row0);
  float* Ks = sm;                          // [Nk*nh][pitch]
  float* Vs = Ks + Nk * nh * pitch;        // [Nk*nh][pitch]
  float* Qs = Vs + Nk * nh * pitch;        // [RB*nh][pitch]  (reused for O)
  float* Av = Qs + RB * nh * pitch;        // [RB][Nk] head-average accumulator
  const int tid = threadIdx.x;
  for (int i = tid; i < Nk * H; i += 256) {
    const int j = i / H, c = i - j * H;
    const size_t g = (size_t)(b * Nk + j) * (2 * H) + c;
    const int d = (j * nh + c / dh) * pitch + (c % dh);
    Ks[d] = KV[g]; Vs[d] = KV[g + H];
  }
  for (int i = tid; i < rows * H; i += 256) {
    const int r = i / H, c = i - r * H;
    Qs[(r * nh + c / dh) * pitch + (c % dh)] = Q[(size_t)(r0 + row0 + r) * H + c];
  }
  if (attn_avg) for (int i = tid; i < RB * Nk; i += 256) Av[i] = 0.f;
  __syncthreads();
  const int pair = tid;                    // (r, h), h fastest
  const int r = pair / nh, hh = pair - r * nh;
  if (pair < RB * nh && r < rows) {
    const float* q = Qs + pair * pitch;
    float s[NKMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) {
      s[j] = -INFINITY;
      if (j < Nk) {
        const float* k = Ks + (j * nh + hh) * pitch;
        float a = 0.f;
        for (int d = 0; d < dh; ++d) a = fmaf(q[d], k[d], a);
        s[j] = a * scale;
        mx = fmaxf(mx, s[j]);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) { s[j] = (j < Nk) ? __expf(s[j] - mx) : 0.f; sum += s[j]; }
    const float inv = 1.0f / sum;
    const int t = r0 + row0 + r;
    const size_t pbase = ((size_t)t * nh + hh) * Nk;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) {
      if (j < Nk) {
        const float p = s[j] * inv;
        P[pbase + j] = p;
        s[j] = drop.p > 0.f ? p * drop_mult(drop, SITE_ATTN_RG2KG, (uint32_t)(pbase + j)) : p;
        if (attn_avg) atomicAdd(&Av[r * Nk + j], s[j] / (float)nh);
      }
    }
    float* o = Qs + pair * pitch;          // this lane's own slot: safe to overwrite
    for (int d = 0; d < dh; ++d) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) a = fmaf(s[j], Vs[(j * nh + hh) * pitch + d], a);
      o[d] = a;
    }
  }
  __syncthreads();
  for (int i = tid; i < rows * H; i += 256) {
    const int rr = i / H, c = i - rr * H;
    O[(size_t)(r0 + row0 + rr) * H + c] = Qs[(rr * nh + c / dh) * pitch + (c % dh)];
  }
  if (attn_avg)
    for (int i = tid; i < rows * Nk; i += 256) attn_avg[(size_t)(r0 + row0) * Nk + i] = Av[i];
}

// ------------------------------------------------------------------ rg2kg backward
// dO [T,H] -> dQ [T,H]; dKV [B*Nk,2H] += (atomic; zeroed by the caller)
template <int NKMAX>
__global__ __launch_bounds__(256) void attn_rg2kg_bwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const float* __restrict__ P,
    const float* __restrict__ dO, const int* __restrict__ offs,
    float* __restrict__ dQ, float* __restrict__ dKV,
    int H, int nh, int Nk, int RB, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int dh = H / nh, pitch = dh + 1;
  const int b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int row0 = blockIdx.x * RB;
  if (row0 >= nr) return;
  const int rows = min(RB, nr - row0);
  float* Ks = sm;
  float* Vs = Ks + Nk * nh * pitch;
  float* Qs = Vs + Nk * nh * pitch;        // [RB*nh][pitch]
  float* Gs = Qs + RB * nh * pitch;        // dO, later dQ
  float* dSs = Gs + RB * nh * pitch;       // [RB*nh][Nk]  dS*scale
  float* Pds = dSs + RB * nh * Nk;         // [RB*nh][Nk]  dropped probabilities
  const int tid = threadIdx.x;
  for (int i = tid; i < Nk * H; i += 256) {
    const int j = i / H, c = i - j * H;
    const size_t g = (size_t)(b * Nk + j) * (2 * H) + c;
    const int d = (j * nh + c / dh) * pitch + (c % dh);
    Ks[d] = KV[g]; Vs[d] = KV[g + H];
  }
  for (int i = tid; i < RB * H; i += 256) {
    const int r = i / H, c = i - r * H;
    const int d = (r * nh + c / dh) * pitch + (c % dh);
    const bool ok = r < rows;
    Qs[d] = ok ? Q[(size_t)(r0 + row0 + r) * H + c] : 0.f;
    Gs[d] = ok ? dO[(size_t)(r0 + row0 + r) * H + c] : 0.f;
  }
  __syncthreads();
  const int pair = tid;
  const int r = pair / nh, hh = pair - r * nh;
  if (pair < RB * nh) {
    float ds[NKMAX];
    if (r < rows) {
      const float* g = Gs + pair * pitch;
      const int t = r0 + row0 + r;
      const size_t pbase = ((size_t)t * nh + hh) * Nk;
      float p[NKMAX];
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < NKMAX; ++j) {
        p[j] = 0.f; ds[j] = 0.f;
        if (j < Nk) {
          const float* v = Vs + (j * nh + hh) * pitch;
          float a = 0.f;
          for (int d = 0; d < dh; ++d) a = fmaf(g[d], v[d], a);
          p[j] = P[pbase + j];
          const float m = drop.p > 0.f ? drop_mult(drop, SITE_ATTN_RG2KG, (uint32_t)(pbase + j)) : 1.0f;
          Pds[pair * Nk + j] = p[j] * m;
          ds[j] = a * m;                    // dP
          dot = fmaf(p[j], ds[j], dot);
        }
      }
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) { ds[j] = p[j] * (ds[j] - dot) * scale; dSs[pair * Nk + j] = ds[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) { ds[j] = 0.f; dSs[pair * Nk + j] = 0.f; Pds[pair * Nk + j] = 0.f; }
    }
  }
  __syncthreads();   // every lane has finished reading Gs as dO ... but dV below still needs dO
  // ---- dK, dV: column c = (h,d) per thread, reduce over the block's rows
  for (int c = tid; c < H; c += 256) {
    const int hc = c / dh, d = c - hc * dh;
    for (int j = 0; j < Nk; ++j) {
      float ak = 0.f, av = 0.f;
      for (int rr = 0; rr < rows; ++rr) {
        const int pr = rr * nh + hc;
        ak = fmaf(dSs[pr * Nk + j], Qs[pr * pitch + d], ak);
        av = fmaf(Pds[pr * Nk + j], Gs[pr * pitch + d], av);
      }
      float* dst = dKV + (size_t)(b * Nk + j) * (2 * H) + c;
      atomicAdd(dst, ak);
      atomicAdd(dst + H, av);
    }
  }
  __syncthreads();
  // ---- dQ = dS*scale . K  (into this lane's Gs slot, then a coalesced write)
  if (pair < RB * nh && r < rows) {
    float ds[NKMAX];
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) ds[j] = (j < Nk) ? dSs[pair * Nk + j] : 0.f;
    float* g = Gs + pair * pitch;
    for (int d = 0; d < dh; ++d) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) a = fmaf(ds[j], Ks[(j * nh + hh) * pitch + d], a);
      g[d] = a;
    }
  }
  __syncthreads();
  for (int i = tid; i < rows * H; i += 256) {
    const int rr = i / H, c = i - rr * H;
    dQ[(size_t)(r0 + row0 + rr) * H + c] = Gs[(rr * nh + c / dh) * pitch + (c % dh)];
  }
}

// block-wide reductions of NKMAX values held per thread (256 threads = 4 waves)
template <int NKMAX, bool IS_MAX>
__device__ __forceinline__ void block_reduce(float (&v)[NKMAX], int Nk, float* red /*[4][NKMAX]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NKMAX; ++j)
    if (j < Nk) {
      const float w = IS_MAX ? wave_max(v[j]) : wave_sum(v[j]);
      if (lane == 0) red[wave * NKMAX + j] = w;
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NKMAX; ++j)
    if (j < Nk) {
      const float a = red[j], b = red[NKMAX + j], c = red[2 * NKMAX + j], d = red[3 * NKMAX + j];
      v[j] = IS_MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
    }
  __syncthreads();
}

// ------------------------------------------------------------------ kg2rg forward
// grid (nh, B)
template <int NKMAX>
__global__ __launch_bounds__(256) void attn_kg2rg_fwd_kernel(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const int* __restrict__ offs,
    float* __restrict__ P2, float* __restrict__ O2,
    int H, int nh, int Nk, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int dh = H / nh;
  const int hh = blockIdx.x, b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  float* qs = sm;                       // [Nk][dh]
  float* red = qs + Nk * dh;            // [4][NKMAX]
  float* acc_s = red + 4 * NKMAX;       // [G][Nk][dh] partial O2, G = 256/dh groups (dh<=256)
  const int tid = threadIdx.x;
  for (int i = tid; i < Nk * dh; i += 256) {
    const int j = i / dh, d = i - j * dh;
    qs[i] = Q2[(size_t)(b * Nk + j) * H + hh * dh + d];
  }
  __syncthreads();
  // pass 1: raw scores -> P2 (scratch use), running max
  float mx[NKMAX];
#pragma unroll
  for (int j = 0; j < NKMAX; ++j) mx[j] = -INFINITY;
  for (int t = tid; t < nr; t += 256) {
    const float* k = KV2 + (size_t)(r0 + t) * (2 * H) + hh * dh;
    float s[NKMAX];
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) s[j] = 0.f;
    for (int d = 0; d < dh; ++d) {
      const float kd = k[d];
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) s[j] = fmaf(kd, qs[j * dh + d], s[j]);
    }
    float* pp = P2 + ((size_t)(r0 + t) * nh + hh) * Nk;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j)
      if (j < Nk) { s[j] *= scale; pp[j] = s[j]; mx[j] = fmaxf(mx[j], s[j]); }
  }
  block_reduce<NKMAX, true>(mx, Nk, red);
  // pass 2: exponentials (kept in P2), sums
  float sum[NKMAX];
#pragma unroll
  for (int j = 0; j < NKMAX; ++j) sum[j] = 0.f;
  for (int t = tid; t < nr; t += 256) {
    float* pp = P2 + ((size_t)(r0 + t) * nh + hh) * Nk;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j)
      if (j < Nk) { const float e = __expf(pp[j] - mx[j]); pp[j] = e; sum[j] += e; }
  }
  block_reduce<NKMAX, false>(sum, Nk, red);
  // pass 3: normalise
  for (int t = tid; t < nr; t += 256) {
    float* pp = P2 + ((size_t)(r0 + t) * nh + hh) * Nk;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j)
      if (j < Nk) pp[j] = pp[j] / sum[j];
  }
  __syncthreads();   // P2 rows of this (b,h) are now visible to the whole block
  // pass 4: O2[j][d] = sum_t Pd[t][j] * V[t][d]; thread = (group g, d)
  const int G = 256 / dh;
  const int g = tid / dh, d = tid - g * dh;
  float acc[NKMAX];
#pragma unroll
  for (int j = 0; j < NKMAX; ++j) acc[j] = 0.f;
  if (g < G) {
    for (int t = g; t < nr; t += G) {
      const float v = KV2[(size_t)(r0 + t) * (2 * H) + H + hh * dh + d];
      const size_t pbase = ((size_t)(r0 + t) * nh + hh) * Nk;
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) {
          float p = P2[pbase + j];
          if (drop.p > 0.f) p *= drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)(pbase + j));
          acc[j] = fmaf(p, v, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NKMAX; ++j)
      if (j < Nk) acc_s[(g * Nk + j) * dh + d] = acc[j];
  }
  __syncthreads();
  for (int i = tid; i < Nk * dh; i += 256) {
    float a = 0.f;
    for (int gg = 0; gg < G; ++gg) a += acc_s[gg * Nk * dh + i];
    const int j = i / dh, dd = i - j * dh;
    O2[(size_t)(b * Nk + j) * H + hh * dh + dd] = a;
  }
}

// ------------------------------------------------------------------ kg2rg backward
// dO2 [B*Nk,H] -> dQ2 [B*Nk,H], dKV2 [T,2H] (plain stores: each element has one owner).
// dS2 [T,nh,Nk] is scratch.
template <int NKMAX>
__global__ __launch_bounds__(256) void attn_kg2rg_bwd_kernel(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const float* __restrict__ P2,
    const float* __restrict__ dO2, const int* __restrict__ offs,
    float* __restrict__ dQ2, float* __restrict__ dKV2, float* __restrict__ dS2,
    int H, int nh, int Nk, float scale, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int dh = H / nh;
  const int hh = blockIdx.x, b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  float* qs = sm;                       // [Nk][dh]
  float* gs = qs + Nk * dh;             // [Nk][dh]  dO2
  float* red = gs + Nk * dh;            // [4][NKMAX]
  float* acc_s = red + 4 * NKMAX;       // [G][Nk][dh]
  const int tid = threadIdx.x;
  for (int i = tid; i < Nk * dh; i += 256) {
    const int j = i / dh, d = i - j * dh;
    const size_t o = (size_t)(b * Nk + j) * H + hh * dh + d;
    qs[i] = Q2[o]; gs[i] = dO2[o];
  }
  __syncthreads();
  // pass 1: dP[t][j] = mask * dO2[j].V[t]; rowdot[j] = sum_t P*dP; dV2[t] = sum_j Pd[t][j] dO2[j]
  float dot[NKMAX];
#pragma unroll
  for (int j = 0; j < NKMAX; ++j) dot[j] = 0.f;
  for (int t = tid; t < nr; t += 256) {
    const float* v = KV2 + (size_t)(r0 + t) * (2 * H) + H + hh * dh;
    float* dv = dKV2 + (size_t)(r0 + t) * (2 * H) + H + hh * dh;
    const size_t pbase = ((size_t)(r0 + t) * nh + hh) * Nk;
    float dp[NKMAX], pd[NKMAX];
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) { dp[j] = 0.f; pd[j] = 0.f; }
    for (int d = 0; d < dh; ++d) {
      const float vd = v[d];
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) dp[j] = fmaf(vd, gs[j * dh + d], dp[j]);
    }
#pragma unroll
    for (int j = 0; j < NKMAX; ++j)
      if (j < Nk) {
        const float p = P2[pbase + j];
        const float m = drop.p > 0.f ? drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)(pbase + j)) : 1.0f;
        dp[j] *= m; pd[j] = p * m;
        dot[j] = fmaf(p, dp[j], dot[j]);
        dS2[pbase + j] = dp[j];            // dP for pass 2
      }
    for (int d = 0; d < dh; ++d) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) a = fmaf(pd[j], gs[j * dh + d], a);
      dv[d] = a;
    }
  }
  block_reduce<NKMAX, false>(dot, Nk, red);
  // pass 2: dS = P*(dP - rowdot)*scale ; dK2[t] = sum_j dS[t][j] q[j]
  for (int t = tid; t < nr; t += 256) {
    float* dk = dKV2 + (size_t)(r0 + t) * (2 * H) + hh * dh;
    const size_t pbase = ((size_t)(r0 + t) * nh + hh) * Nk;
    float ds[NKMAX];
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) {
      ds[j] = 0.f;
      if (j < Nk) { ds[j] = P2[pbase + j] * (dS2[pbase + j] - dot[j]) * scale; dS2[pbase + j] = ds[j]; }
    }
    for (int d = 0; d < dh; ++d) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) a = fmaf(ds[j], qs[j * dh + d], a);
      dk[d] = a;
    }
  }
  __syncthreads();
  // pass 3: dQ2[j][d] = sum_t dS[t][j] K2[t][d]
  const int G = 256 / dh;
  const int g = tid / dh, d = tid - g * dh;
  float acc[NKMAX];
#pragma unroll
  for (int j = 0; j < NKMAX; ++j) acc[j] = 0.f;
  if (g < G) {
    for (int t = g; t < nr; t += G) {
      const float k = KV2[(size_t)(r0 + t) * (2 * H) + hh * dh + d];
      const size_t pbase = ((size_t)(r0 + t) * nh + hh) * Nk;
#pragma unroll
      for (int j = 0; j < NKMAX; ++j)
        if (j < Nk) acc[j] = fmaf(dS2[pbase + j], k, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NKMAX; ++j)
      if (j < Nk) acc_s[(g * Nk + j) * dh + d] = acc[j];
  }
  __syncthreads();
  for (int i = tid; i < Nk * dh; i += 256) {
    float a = 0.f;
    for (int gg = 0; gg < G; ++gg) a += acc_s[gg * Nk * dh + i];
    const int j = i / dh, dd = i - j * dh;
    dQ2[(size_t)(b * Nk + j) * H + hh * dh + dd] = a;
  }
}

// head-average of the (dropped) kg2rg probabilities: out[t][j] = mean_h Pd[t,h,j]
__global__ void attn_avg_kernel(const float* __restrict__ P2, float* __restrict__ out,
                                int T, int nh, int Nk, DropCfg drop) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * Nk) return;
  const int t = i / Nk, j = i - t * Nk;
  float a = 0.f;
  for (int h = 0; h < nh; ++h) {
    const size_t idx = ((size_t)t * nh + h) * Nk + j;
    float p = P2[idx];
    if (drop.p > 0.f) p *= drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)idx);
    a += p;
  }
  out[i] = a / (float)nh;
}

}  // namespace

// ------------------------------------------------------------------ launchers
static inline size_t rg2kg_fwd_lds(int H, int nh, int Nk, int RB) {
  const int dh = H / nh, pitch = dh + 1;
  return sizeof(float) * ((size_t)2 * Nk * nh * pitch + (size_t)RB * nh * pitch + (size_t)RB * Nk);
}
static inline size_t rg2kg_bwd_lds(int H, int nh, int Nk, int RB) {
  const int dh = H / nh, pitch = dh + 1;
  return sizeof(float) * ((size_t)2 * Nk * nh * pitch + (size_t)2 * RB * nh * pitch + (size_t)2 * RB * nh * Nk);
}
constexpr size_t LDS_BUDGET = 150 * 1024;
// largest rows-per-block (<= 256/nh) whose LDS image fits; 0 if even one row does not
static inline int rg2kg_rows(int H, int nh, int Nk, bool bwd) {
  int RB = 256 / nh;
  while (RB >= 1 && (bwd ? rg2kg_bwd_lds(H, nh, Nk, RB) : rg2kg_fwd_lds(H, nh, Nk, RB)) > LDS_BUDGET) RB >>= 1;
  return RB;
}
static inline size_t kg2rg_lds(int H, int nh, int Nk, int nkmax, bool bwd) {
  const int dh = H / nh, G = 256 / dh;
  return sizeof(float) * ((size_t)(bwd ? 2 : 1) * Nk * dh + 4 * nkmax + (size_t)G * Nk * dh);
}

int attn_supported(int H, int nh, int Nk) {
  if (nh <= 0 || H % nh) return 0;
  const int dh = H / nh;
  if (nh > 256 || dh > 256 || Nk < 1 || Nk > 64) return 0;
  const int nkmax = Nk <= 16 ? 16 : 64;
  if (rg2kg_rows(H, nh, Nk, true) < 1 || kg2rg_lds(H, nh, Nk, nkmax, true) > LDS_BUDGET) return 0;
  return 1;
}

#define DISPATCH_NK(KERN, GRID, LDS, ...)                                                        \
  do {                                                                                           \
    if (Nk <= 16) {                                                                              \
      if ((LDS) > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERN<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); \
      hipLaunchKernelGGL(KERN<16>, GRID, dim3(256), LDS, stream, __VA_ARGS__);                   \
    } else {                                                                                     \
      if ((LDS) > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERN<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); \
      hipLaunchKernelGGL(KERN<64>, GRID, dim3(256), LDS, stream, __VA_ARGS__);                   \
    }                                                                                            \
  } while (0)

int launch_attn_rg2kg_fwd(const float* Q, const float* KV, const int* offs, float* P, float* O, float* attn_avg,
                          int B, int T, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  if (attn_mfma_ok(H, nh, Nk, max_nr, false))
    return launch_rg2kg_fwd_mfma(Q, KV, offs, P, O, attn_avg, B, T, max_nr, H, nh, Nk, drop, stream);
  if (attn_fast_ok(H, nh, Nk, max_nr, false, false))
    return launch_rg2kg_fwd32(Q, KV, offs, P, O, attn_avg, B, max_nr, H, nh, Nk, drop, stream);
  const int RB = rg2kg_rows(H, nh, Nk, false);
  if (RB < 1) return (int)hipErrorInvalidValue;
  const dim3 grid((max_nr + RB - 1) / RB, B);
  const float scale = 1.0f / sqrtf((float)(H / nh));
  const size_t lds = rg2kg_fwd_lds(H, nh, Nk, RB);
  DISPATCH_NK(attn_rg2kg_fwd_kernel, grid, lds, Q, KV, offs, P, O, attn_avg, H, nh, Nk, RB, scale, drop);
  return (int)hipGetLastError();
}

int launch_attn_rg2kg_bwd(const float* Q, const float* KV, const float* P, const float* dO, const int* offs,
                          float* dQ, float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop,
                          hipStream_t stream) {
  if (attn_mfma_ok(H, nh, Nk, max_nr, false))
    return launch_rg2kg_bwd_mfma(Q, KV, P, dO, offs, dQ, dKV, B, max_nr, H, nh, Nk, drop, stream);
  if (attn_fast_ok(H, nh, Nk, max_nr, false, true))
    return launch_rg2kg_bwd32(Q, KV, P, dO, offs, dQ, dKV, B, max_nr, H, nh, Nk, drop, stream);
  const int RB = rg2kg_rows(H, nh, Nk, true);
  if (RB < 1) return (int)hipErrorInvalidValue;
  const dim3 grid((max_nr + RB - 1) / RB, B);
  const float scale = 1.0f / sqrtf((float)(H / nh));
  const size_t lds = rg2kg_bwd_lds(H, nh, Nk, RB);
  DISPATCH_NK(attn_rg2kg_bwd_kernel, grid, lds, Q, KV, P, dO, offs, dQ, dKV, H, nh, Nk, RB, scale, drop);
  return (int)hipGetLastError();
}

int launch_attn_kg2rg_fwd(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2,
                          int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  if (attn_mfma_ok(H, nh, Nk, max_nr, true))
    return launch_kg2rg_fwd_mfma(Q2, KV2, offs, P2, O2, B, H, nh, Nk, drop, stream);
  if (attn_fast_ok(H, nh, Nk, max_nr, true, false))
    return launch_kg2rg_fwd32(Q2, KV2, offs, P2, O2, B, max_nr, H, nh, Nk, drop, stream);
  const dim3 grid(nh, B);
  const float scale = 1.0f / sqrtf((float)(H / nh));
  const int nkmax = Nk <= 16 ? 16 : 64;
  const size_t lds = kg2rg_lds(H, nh, Nk, nkmax, false);
  DISPATCH_NK(attn_kg2rg_fwd_kernel, grid, lds, Q2, KV2, offs, P2, O2, H, nh, Nk, scale, drop);
  return (int)hipGetLastError();
}

int launch_attn_kg2rg_bwd(const float* Q2, const float* KV2, const float* P2, const float* dO2, const float* O2, const int* offs,
                          float* dQ2, float* dKV2, float* dS2, int B, int max_nr, int H, int nh, int Nk, DropCfg drop,
                          hipStream_t stream) {
  if (attn_mfma_ok(H, nh, Nk, max_nr, true))
    return launch_kg2rg_bwd_mfma(Q2, KV2, P2, dO2, O2, offs, dQ2, dKV2, B, H, nh, Nk, drop, stream);
  if (attn_fast_ok(H, nh, Nk, max_nr, true, true))
    return launch_kg2rg_bwd32(Q2, KV2, P2, dO2, offs, dQ2, dKV2, B, max_nr, H, nh, Nk, drop, stream);
  const dim3 grid(nh, B);
  const float scale = 1.0f / sqrtf((float)(H / nh));
  const int nkmax = Nk <= 16 ? 16 : 64;
  const size_t lds = kg2rg_lds(H, nh, Nk, nkmax, true);
  DISPATCH_NK(attn_kg2rg_bwd_kernel, grid, lds, Q2, KV2, P2, dO2, offs, dQ2, dKV2, dS2, H, nh, Nk, scale, drop);
  return (int)hipGetLastError();
}

int launch_attn_avg(const float* P2, float* out, int T, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  const int n = T * Nk;
  hipLaunchKernelGGL(attn_avg_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, P2, out, T, nh, Nk, drop);
  return (int)hipGetLastError();
}
