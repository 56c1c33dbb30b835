// Cross-attention cores on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32), head_dim = 32.
//
// Both attention directions have one tiny dimension (Nk = 13 KG rows, padded to one 16-wide
// MFMA tile) and one long one (the RG nodes of the sample), so every product is a chain of
// 16x16 tiles over 16-node slices.  Lane coordinates: x = lane & 15, q = lane >> 4.
//
//   * "row fragment" of a row-major matrix X: lane (x,q) holds X[x][8q .. 8q+7] (two 16-B loads).
//     As the A operand it supplies rows of A, as the B operand rows of B^T; MFMA step e of the
//     8-step K=32 chain feeds k = 8q + e from BOTH fragments, so C = A.B^T needs no shuffles.
//   * an accumulator tile D (col = x, rows 4q..4q+3 in the 4 registers) is, unchanged, the A
//     operand of the next product that contracts over D's ROW index: step e feeds k = 4q + e
//     and the B operand supplies B[4q+e][col].  So scores are always produced with the index to
//     be contracted next on the rows:  S^T = K.Q^T (keys x nodes) feeds O = P.V and dQ = dS.K;
//     the other orientation (nodes x keys) is recomputed with one more 8-MFMA chain where a
//     gradient contracts over nodes (dK = dS^T.Q, dV = P^T.dO) -- cheaper than a transpose.
//   * softmax reductions run over the 4 registers and the 4 lane groups (xor 16, 32) only.
//
// Numerics: fp32 FMA chains in k order, same math as attn.hip; dropout masks from the shared
// counter hash.  Conditions: head_dim == 32, Nk <= 16; kg2rg additionally Nr <= 16*4*MAXT.
#include "attn.h"


namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int DH = 32;
constexpr int NW = 8;      // waves per block in the kg2rg kernels
constexpr int MAXT = 6;    // key tiles per wave there (8 waves x 6 tiles x 16 keys => Nr <= 768)

struct Frag8 { float4 lo, hi; };

__device__ __forceinline__ Frag8 load_row8(const float* __restrict__ row, int q, bool valid) {
  Frag8 f;
  f.lo = *reinterpret_cast<const float4*>(row + 8 * q);
  f.hi = *reinterpret_cast<const float4*>(row + 8 * q + 4);
  if (!valid) { f.lo = make_float4(0.f, 0.f, 0.f, 0.f); f.hi = f.lo; }
  return f;
}
__device__ __forceinline__ Frag8 scale8(Frag8 f, float s) {
  f.lo.x *= s; f.lo.y *= s; f.lo.z *= s; f.lo.w *= s; f.hi.x *= s; f.hi.y *= s; f.hi.z *= s; f.hi.w *= s;
  return f;
}

// C += A.B^T, K = 32: a = row fragment of A (lane x = row of C), b = row fragment of B (lane x = col of C)
__device__ __forceinline__ f4 mma_nt32(const Frag8& a, const Frag8& b, f4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, c, 0, 0, 0);
  return c;
}
// C += A.B, K = 16: a = an accumulator tile whose ROWS are the contraction index (a[e] <-> k = 4q+e),
// b[e] = B[4q+e][col x]
__device__ __forceinline__ f4 mma_acc16(const f4& a, const f4& b, f4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  return c;
}
// b[e] = X[(row0 + 4q + e) * ld + col] for e = 0..3, rows clamped to [.., row_max] (finite filler;
// the matching A entries are zero there)
__device__ __forceinline__ f4 load_col4(const float* __restrict__ X, size_t ld, int row0, int q, int row_max, int col) {
  f4 b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = X[(size_t)min(row0 + 4 * q + e, row_max) * ld + col];
  return b;
}
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// ------------------------------------------------------------------ rg2kg forward
// one wave = one (16-node tile, head) pair of sample blockIdx.y: tile-major, head fastest, so the 4 waves
// of a block share Q rows.  The head-averaged map (inference only) is a separate pass over P.
__device__ __forceinline__ void rg2kg_fwd_body(
    const float* __restrict__ Q, const float* __restrict__ KV, const int* __restrict__ offs,
    float* __restrict__ P, float* __restrict__ O, Bf16Dst o16, int H, int nh, int Nk, float scale, const DropCfg& drop,
    int bx, int b) {
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int lane = threadIdx.x & 63, x = lane & 15, q = lane >> 4;
  const int task = bx * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int h = task % nh, t0 = (task / nh) * 16;
  if (t0 >= nr) return;
  const int node = r0 + min(t0 + x, nr - 1);            // this lane's node as a COLUMN of S^T
  const bool node_ok = t0 + x < nr;
  const float* kvb = KV + (size_t)b * Nk * 2 * H;
  const Frag8 qf = load_row8(Q + (size_t)node * H + h * DH, q, true);
  const Frag8 kf = load_row8(kvb + (size_t)min(x, Nk - 1) * 2 * H + h * DH, q, x < Nk);
  f4 vb[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) vb[n] = load_col4(kvb + H + h * DH + 16 * n, (size_t)2 * H, 0, q, Nk - 1, x);
  f4 s = mma_nt32(kf, qf, f4{0.f, 0.f, 0.f, 0.f});       // S^T: rows = keys 4q+r, col = node x
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) { s[r] = (4 * q + r < Nk) ? s[r] * scale : -INFINITY; m = fmaxf(m, s[r]); }
  m = group_max(m);
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - m); sum += s[r]; }
  sum = group_sum(sum);
  const float inv = 1.0f / sum;
  const size_t pbase = ((size_t)node * nh + h) * Nk;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = 4 * q + r;
    float p = s[r] * inv;
    if (key < Nk) {
      if (node_ok) P[pbase + key] = p;
      if (drop.p > 0.f) p *= drop_mult(drop, SITE_ATTN_RG2KG, (uint32_t)(pbase + key));
    }
    s[r] = p;
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const f4 o = mma_acc16(s, vb[n], f4{0.f, 0.f, 0.f, 0.f});   // rows = nodes 4q+r, col = 16n + x
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (t0 + 4 * q + r < nr) {
        const size_t rw = (size_t)(r0 + t0 + 4 * q + r); const int cl = h * DH + 16 * n + x;
        if (o16.p) o16.p[rw * o16.ld + cl] = f2bf(o[r]); else O[rw * H + cl] = o[r];
      }
  }
}

__global__ __launch_bounds__(256) void rg2kg_fwd_mfma_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const int* __restrict__ offs,
    float* __restrict__ P, float* __restrict__ O, Bf16Dst o16, int H, int nh, int Nk, float scale, DropCfg drop) {
  rg2kg_fwd_body(Q, KV, offs, P, O, o16, H, nh, Nk, scale, drop, blockIdx.x, blockIdx.y);
}

// head-average of the (dropped) probabilities: out[t][j] = mean_h drop(P[t,h,j])
__global__ void attn_avg_site_kernel(const float* __restrict__ Pm, float* __restrict__ out, int T, int nh, int Nk,
                                     uint32_t site, DropCfg drop) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * Nk) return;
  const int t = i / Nk, j = i - t * Nk;
  float a = 0.f;
  for (int h = 0; h < nh; ++h) {
    const size_t idx = ((size_t)t * nh + h) * Nk + j;
    float p = Pm[idx];
    if (drop.p > 0.f) p *= drop_mult(drop, site, (uint32_t)idx);
    a += p;
  }
  out[i] = a / (float)nh;
}

// ------------------------------------------------------------------ rg2kg backward
// grid (chunks, nh, B); one wave = TPW consecutive 16-node tiles of (sample, head); dK/dV partials in
// accumulator tiles, one atomicAdd per element per wave.
constexpr int TPW = 2;
// TPW 16-node tiles (first nodes t0s[], any order) of (sample, head): dQ rows are stored, the dK / dV partial sums
// of the tiles are added to dKa / dVa (rows = keys 4q+r, col = 16n + x).
__device__ __forceinline__ void rg2kg_bwd_tiles(
    const float* __restrict__ Q, const float* __restrict__ P, const float* __restrict__ dO,
    float* __restrict__ dQ, Bf16Dst dq16, int H, int nh, int Nk, float scale, const DropCfg& drop,
    int r0, int nr, int h, const int (&t0s)[TPW], const Frag8& vf, const f4 (&kb)[2], f4 (&dKa)[2], f4 (&dVa)[2]) {
  const int lane = threadIdx.x & 63, x = lane & 15, q = lane >> 4;
  // all loads of the tiles first, unconditionally, from rows clamped into the sample (see kg2rg forward)
  Frag8 gfv[TPW];
  f4 pTv[TPW], pNv[TPW], qbv[TPW][2], gbv[TPW][2];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int t0 = min(t0s[tt], nr - 1);
    const int node = r0 + min(t0 + x, nr - 1);
    gfv[tt] = load_row8(dO + (size_t)node * H + h * DH, q, true);
    const size_t pbase = ((size_t)node * nh + h) * Nk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pTv[tt][r] = P[pbase + min(4 * q + r, Nk - 1)];
      pNv[tt][r] = P[((size_t)(r0 + min(t0 + 4 * q + r, nr - 1)) * nh + h) * Nk + min(x, Nk - 1)];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      qbv[tt][n] = load_col4(Q + (size_t)r0 * H + h * DH + 16 * n, (size_t)H, t0, q, nr - 1, x);
      gbv[tt][n] = load_col4(dO + (size_t)r0 * H + h * DH + 16 * n, (size_t)H, t0, q, nr - 1, x);
    }
  }
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int t0 = t0s[tt];
    if (t0 >= nr) continue;                                  // wave-uniform; nothing below loads
    const int node = r0 + min(t0 + x, nr - 1);
    const bool node_ok = t0 + x < nr;
    const Frag8 gf = gfv[tt];
    // ---- orientation T: rows = keys 4q+r, col = node x
    f4 dpT = mma_nt32(vf, gf, f4{0.f, 0.f, 0.f, 0.f});
    const size_t pbase = ((size_t)node * nh + h) * Nk;
    f4 pT;
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 4 * q + r;
      const bool ok = key < Nk && node_ok;
      pT[r] = ok ? pTv[tt][r] : 0.f;
      const float m = (ok && drop.p > 0.f) ? drop_mult(drop, SITE_ATTN_RG2KG, (uint32_t)(pbase + key)) : 1.0f;
      dpT[r] *= m;
      dot = fmaf(pT[r], dpT[r], dot);
    }
    dot = group_sum(dot);                                  // row-dot of node x
    f4 dsT;
#pragma unroll
    for (int r = 0; r < 4; ++r) dsT[r] = pT[r] * (dpT[r] - dot) * scale;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f4 dq = mma_acc16(dsT, kb[n], f4{0.f, 0.f, 0.f, 0.f});     // rows = nodes 4q+r, col = 16n + x
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (t0 + 4 * q + r < nr) {
          const size_t rw = (size_t)(r0 + t0 + 4 * q + r); const int cl = h * DH + 16 * n + x;
          if (dq16.p) dq16.p[rw * dq16.ld + cl] = f2bf(dq[r]); else dQ[rw * H + cl] = dq[r];
        }
    }
    // ---- orientation N: rows = nodes 4q+r, col = key x
    f4 dpN = mma_nt32(gf, vf, f4{0.f, 0.f, 0.f, 0.f});
    f4 dsN, pdN;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nd = t0 + 4 * q + r;
      const bool ok = x < Nk && nd < nr;
      const size_t pb = ((size_t)(r0 + min(nd, nr - 1)) * nh + h) * Nk;
      const float p = ok ? pNv[tt][r] : 0.f;
      const float m = (ok && drop.p > 0.f) ? drop_mult(drop, SITE_ATTN_RG2KG, (uint32_t)(pb + x)) : 1.0f;
      const float dotn = __shfl(dot, 4 * q + r, 64);       // row-dot of node 4q+r lives in lane x' = 4q+r
      dsN[r] = p * (dpN[r] * m - dotn) * scale;
      pdN[r] = p * m;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      dKa[n] = mma_acc16(dsN, qbv[tt][n], dKa[n]);         // rows = keys 4q+r, col = 16n + x
      dVa[n] = mma_acc16(pdN, gbv[tt][n], dVa[n]);
    }
  }
}

// grid (chunks, nh, B); one wave = TPW consecutive 16-node tiles of (sample, head); the block's dK/dV partials are
// combined in LDS, one atomicAdd per element per block.
__global__ __launch_bounds__(256) void rg2kg_bwd_mfma_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const float* __restrict__ P,
    const float* __restrict__ dO, const int* __restrict__ offs,
    float* __restrict__ dQ, float* __restrict__ dKV, Bf16Dst dq16,
    int H, int nh, int Nk, float scale, DropCfg drop) {
  __shared__ float comb[3][16][64];
  const int b = blockIdx.z, h = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
  if ((int)blockIdx.x * 4 * TPW * 16 >= nr) return;      // whole block past the sample (uniform)
  const int tfirst = (blockIdx.x * 4 + wave) * TPW;
  const float* kvb = KV + (size_t)b * Nk * 2 * H;
  const Frag8 vf = load_row8(kvb + (size_t)min(x, Nk - 1) * 2 * H + H + h * DH, q, x < Nk);
  f4 kb[2], dKa[2], dVa[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    kb[n] = load_col4(kvb + h * DH + 16 * n, (size_t)2 * H, 0, q, Nk - 1, x);   // K_h[key 4q+e][16n + x]
    dKa[n] = f4{0.f, 0.f, 0.f, 0.f}; dVa[n] = f4{0.f, 0.f, 0.f, 0.f};
  }
  const int t0s[TPW] = {tfirst * 16, (tfirst + 1) * 16};
  rg2kg_bwd_tiles(Q, P, dO, dQ, dq16, H, nh, Nk, scale, drop, r0, nr, h, t0s, vf, kb, dKa, dVa);
  // combine the block's four partial tiles in LDS, then one atomicAdd per element per block
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) { comb[wave - 1][n * 4 + r][lane] = dKa[n][r]; comb[wave - 1][8 + n * 4 + r][lane] = dVa[n][r]; }
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 4 * q + r;
      if (key < Nk) {
        const int i = n * 4 + r;
        float* dst = dKV + (size_t)(b * Nk + key) * 2 * H + h * DH + 16 * n + x;
        atomicAdd(dst, dKa[n][r] + (comb[0][i][lane] + comb[1][i][lane]) + comb[2][i][lane]);
        atomicAdd(dst + H, dVa[n][r] + (comb[0][8 + i][lane] + comb[1][8 + i][lane]) + comb[2][8 + i][lane]);
      }
    }
}

// ------------------------------------------------------------------ kg2rg forward, grid (nh, B)
// wave w owns key tiles w, w+4, ...; scores S2^T (rows = keys 4q+r, col = query x) stay in registers.
__device__ __forceinline__ void kg2rg_fwd_body(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const int* __restrict__ offs,
    float* __restrict__ P2, float* __restrict__ O2, Bf16Dst o16, int H, int nh, int Nk, float scale, const DropCfg& drop,
    int h, int b, float (*red)[16], float (*ored)[2][4][64]) {
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
  const int ntiles = (nr + 15) >> 4;
  const float* kv = KV2 + (size_t)r0 * 2 * H;
  const Frag8 q2f = scale8(load_row8(Q2 + (size_t)(b * Nk + min(x, Nk - 1)) * H + h * DH, q, x < Nk), scale);
  // Every load of the kernel is issued up front, unconditionally, from row indices clamped into the sample
  // (tiles past the end re-read its last row and are masked below): with the loads inside "if (tile valid)" the
  // compiler kept each tile's load -> wait -> MFMA chain separate, one memory round trip per tile.
  Frag8 kf[MAXT];
  f4 vb[MAXT][2];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t0 = (wave + NW * i) * 16;
    kf[i] = load_row8(kv + (size_t)min(t0 + x, nr - 1) * 2 * H + h * DH, q, true);
  }
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t0 = (wave + NW * i) * 16;
#pragma unroll
    for (int n = 0; n < 2; ++n) vb[i][n] = load_col4(kv + H + h * DH + 16 * n, (size_t)2 * H, min(t0, nr - 1), q, nr - 1, x);
  }
  f4 s[MAXT];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t0 = (wave + NW * i) * 16;
    s[i] = mma_nt32(kf[i], q2f, f4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int r = 0; r < 4; ++r) { if (t0 + 4 * q + r >= nr) s[i][r] = -INFINITY; m = fmaxf(m, s[i][r]); }
  }
  m = group_max(m);
  if (q == 0) red[wave][x] = m;
  __syncthreads();
  { float mm = red[0][x];
#pragma unroll
    for (int w = 1; w < NW; ++w) mm = fmaxf(mm, red[w][x]);
    m = mm; }
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[i][r] = __expf(s[i][r] - m); sum += s[i][r]; }
  sum = group_sum(sum);
  if (q == 0) red[wave][x] = sum;
  __syncthreads();
  float tot = red[0][x];
#pragma unroll
  for (int w = 1; w < NW; ++w) tot += red[w][x];
  const float inv = 1.0f / tot;
  f4 o[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t0 = (wave + NW * i) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + 4 * q + r;
      float p = s[i][r] * inv;
      if (t < nr && x < Nk) {
        const size_t idx = ((size_t)(r0 + t) * nh + h) * Nk + x;
        P2[idx] = p;
        if (drop.p > 0.f) p *= drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)idx);
      } else {
        p = 0.f;
      }
      s[i][r] = p;
    }
    if (wave + NW * i < ntiles) {                           // (wave-uniform; skips only MFMAs on zeros)
#pragma unroll
      for (int n = 0; n < 2; ++n) o[n] = mma_acc16(s[i], vb[i][n], o[n]);   // rows = queries 4q+r, col = 16n + x
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) ored[wave - 1][n][r][lane] = o[n][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = o[n][r];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) v += ored[w][n][r][lane];
        if (4 * q + r < Nk) {
          const size_t rw = (size_t)(b * Nk + 4 * q + r); const int cl = h * DH + 16 * n + x;
          if (o16.p) o16.p[rw * o16.ld + cl] = f2bf(v);
          if (O2) O2[rw * H + cl] = v;             // (fp32 copy: the backward takes its row-dots from dO2 . O2)
        }
      }
  }
}

__global__ __launch_bounds__(64 * NW) void kg2rg_fwd_mfma_kernel(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const int* __restrict__ offs,
    float* __restrict__ P2, float* __restrict__ O2, Bf16Dst o16, int H, int nh, int Nk, float scale, DropCfg drop) {
  __shared__ float red[NW][16];
  __shared__ float ored[NW - 1][2][4][64];
  kg2rg_fwd_body(Q2, KV2, offs, P2, O2, o16, H, nh, Nk, scale, drop, blockIdx.x, blockIdx.y, red, ored);
}

// ------------------------------------------------------------------ kg2rg backward, grid (nh, B)
__device__ __forceinline__ void kg2rg_bwd_body(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const float* __restrict__ P2,
    const float* __restrict__ dO2, const int* __restrict__ offs,
    float* __restrict__ dQ2, float* __restrict__ dKV2, Bf16Dst dq2_16, Bf16Dst dkv2_16,
    const float* __restrict__ dKV_done, Bf16Dst dkv_16, const float* __restrict__ O2, int H, int nh, int Nk, float scale,
    const DropCfg& drop, int h, int b, float (*red)[16], float (*ored)[2][4][64]) {
  // bf16 schedule: dK|dV of the OTHER attention block (complete: its kernel ran before this one) is the next
  // GEMM's operand; this block converts the slice of its (sample, head)
  if (dkv_16.p) {
    for (int i = threadIdx.x; i < Nk * 2 * DH; i += 64 * NW) {
      const int j = i / (2 * DH), c = i - j * 2 * DH;
      const int col = c < DH ? h * DH + c : H + h * DH + (c - DH);
      dkv_16.p[(size_t)(b * Nk + j) * dkv_16.ld + col] = f2bf(dKV_done[(size_t)(b * Nk + j) * 2 * H + col]);
    }
  }
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
  const int ntiles = (nr + 15) >> 4;
  const float* kv = KV2 + (size_t)r0 * 2 * H;
  const float* q2p = Q2 + (size_t)b * Nk * H + h * DH;
  const float* g2p = dO2 + (size_t)b * Nk * H + h * DH;
  const Frag8 g2f = load_row8(g2p + (size_t)min(x, Nk - 1) * H, q, x < Nk);
  // row-dot of query x, sum_t P.dP over ALL keys, without a pass over the keys: sum_t Pd[x,t] (dO2[x] . V[t]) =
  // dO2[x] . O2[x] (O2 = the forward's attention output of this head, dropout included)
  float dot;
  {
    const Frag8 o2f = load_row8(O2 + (size_t)(b * Nk + min(x, Nk - 1)) * H + h * DH, q, x < Nk);
    dot = (g2f.lo.x * o2f.lo.x + g2f.lo.y * o2f.lo.y) + (g2f.lo.z * o2f.lo.z + g2f.lo.w * o2f.lo.w) +
          (g2f.hi.x * o2f.hi.x + g2f.hi.y * o2f.hi.y) + (g2f.hi.z * o2f.hi.z + g2f.hi.w * o2f.hi.w);
    dot = group_sum(dot);
  }
  // ---- orientation T (rows = keys 4q+r, col = query x): dP, dS, dQ2.  All loads of the kernel are issued up front,
  // unconditionally, from clamped rows (see kg2rg forward)
  Frag8 vf[MAXT];
  f4 pld[MAXT], pl2[MAXT], kb[MAXT][2];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t0 = (wave + NW * i) * 16;
    const int key = min(t0 + x, nr - 1);
    vf[i] = load_row8(kv + (size_t)key * 2 * H + H + h * DH, q, true);
    const size_t pb = ((size_t)(r0 + key) * nh + h) * Nk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + 4 * q + r;
      pld[i][r] = P2[((size_t)(r0 + min(t, nr - 1)) * nh + h) * Nk + min(x, Nk - 1)];
      pl2[i][r] = P2[pb + min(4 * q + r, Nk - 1)];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) kb[i][n] = load_col4(kv + h * DH + 16 * n, (size_t)2 * H, min(t0, nr - 1), q, nr - 1, x);
  }
  f4 dq[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t0 = (wave + NW * i) * 16;
    if (wave + NW * i >= ntiles) continue;                   // wave-uniform; nothing below loads
    f4 ds = mma_nt32(vf[i], g2f, f4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + 4 * q + r;
      const bool ok = t < nr && x < Nk;
      const size_t idx = ((size_t)(r0 + min(t, nr - 1)) * nh + h) * Nk + min(x, Nk - 1);
      const float p = ok ? pld[i][r] : 0.f;
      const float mk = (ok && drop.p > 0.f) ? drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)idx) : 1.0f;
      ds[r] = p * (ds[r] * mk - dot) * scale;                // 0 on masked entries (p = 0)
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) dq[n] = mma_acc16(ds, kb[i][n], dq[n]);                // rows = queries 4q+r, col = 16n + x
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) ored[wave - 1][n][r][lane] = dq[n][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = dq[n][r];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) v += ored[w][n][r][lane];
        if (4 * q + r < Nk) {
          const size_t rw = (size_t)(b * Nk + 4 * q + r); const int cl = h * DH + 16 * n + x;
          if (dq2_16.p) dq2_16.p[rw * dq2_16.ld + cl] = f2bf(v); else dQ2[rw * H + cl] = v;
        }
      }
  }
  // ---- phase 2, orientation N (rows = queries 4q+r, col = key x): dK2, dV2 of every key tile
  f4 q2b[2], g2b[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    q2b[n] = load_col4(q2p + 16 * n, (size_t)H, 0, q, Nk - 1, x);   // Q2_h[query 4q+e][16n + x]
    g2b[n] = load_col4(g2p + 16 * n, (size_t)H, 0, q, Nk - 1, x);
  }
  float dotq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dotq[r] = __shfl(dot, 4 * q + r, 64);  // row-dot of query 4q+r
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t0 = (wave + NW * i) * 16;
    if (wave + NW * i >= ntiles) continue;                   // wave-uniform; nothing below loads
    const int key = min(t0 + x, nr - 1);
    const bool key_ok = t0 + x < nr;
    const f4 dpN = mma_nt32(g2f, vf[i], f4{0.f, 0.f, 0.f, 0.f});
    const size_t pb = ((size_t)(r0 + key) * nh + h) * Nk;
    f4 dsN, pdN;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = 4 * q + r;
      const bool ok = key_ok && qi < Nk;
      const float p = ok ? pl2[i][r] : 0.f;
      const float mk = (ok && drop.p > 0.f) ? drop_mult(drop, SITE_ATTN_KG2RG, (uint32_t)(pb + qi)) : 1.0f;
      dsN[r] = p * (dpN[r] * mk - dotq[r]) * scale;
      pdN[r] = p * mk;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f4 dk = mma_acc16(dsN, q2b[n], f4{0.f, 0.f, 0.f, 0.f});  // rows = keys 4q+r, col = 16n + x
      const f4 dv = mma_acc16(pdN, g2b[n], f4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = t0 + 4 * q + r;
        if (t < nr) {
          if (dkv2_16.p) {
            unsigned short* dst = dkv2_16.p + (size_t)(r0 + t) * dkv2_16.ld + h * DH + 16 * n + x;
            dst[0] = f2bf(dk[r]);
            dst[H] = f2bf(dv[r]);
          } else {
            float* dst = dKV2 + (size_t)(r0 + t) * 2 * H + h * DH + 16 * n + x;
            dst[0] = dk[r];
            dst[H] = dv[r];
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(64 * NW) void kg2rg_bwd_mfma_kernel(
    const float* __restrict__ Q2, const float* __restrict__ KV2, const float* __restrict__ P2,
    const float* __restrict__ dO2, const int* __restrict__ offs,
    float* __restrict__ dQ2, float* __restrict__ dKV2, Bf16Dst dq2_16, Bf16Dst dkv2_16,
    const float* __restrict__ dKV_done, Bf16Dst dkv_16, const float* __restrict__ O2, int H, int nh, int Nk, float scale,
    DropCfg drop) {
  __shared__ float red[NW][16];
  __shared__ float ored[NW - 1][2][4][64];
  kg2rg_bwd_body(Q2, KV2, P2, dO2, offs, dQ2, dKV2, dq2_16, dkv2_16, dKV_done, dkv_16, O2, H, nh, Nk, scale, drop,
                 blockIdx.x, blockIdx.y, red, ored);
}

// ------------------------------------------------------------------ rg2kg backward, one block per (head, sample)
// The "owner" form used by the paired launch below: the block's NW waves walk all 16-node tiles of the sample
// (tile t -> wave t % NW), so dK / dV of the (head, sample) are complete inside the block: no atomics, no zeroed
// accumulator, and the result is written straight as the bf16 GEMM operand.
__device__ __forceinline__ void rg2kg_bwd_owner_body(
    const float* __restrict__ Q, const float* __restrict__ KV, const float* __restrict__ P,
    const float* __restrict__ dO, const int* __restrict__ offs, Bf16Dst dq16, Bf16Dst dkv16,
    int H, int nh, int Nk, float scale, const DropCfg& drop, int h, int b, float (*comb)[16][64]) {
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, q = lane >> 4;
  const float* kvb = KV + (size_t)b * Nk * 2 * H;
  const Frag8 vf = load_row8(kvb + (size_t)min(x, Nk - 1) * 2 * H + H + h * DH, q, x < Nk);
  f4 kb[2], dKa[2], dVa[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    kb[n] = load_col4(kvb + h * DH + 16 * n, (size_t)2 * H, 0, q, Nk - 1, x);   // K_h[key 4q+e][16n + x]
    dKa[n] = f4{0.f, 0.f, 0.f, 0.f}; dVa[n] = f4{0.f, 0.f, 0.f, 0.f};
  }
  static_assert(MAXT % TPW == 0, "tile groups");
#pragma unroll 1
  for (int g = 0; g < MAXT / TPW; ++g) {
    const int t0s[TPW] = {(wave + NW * (TPW * g)) * 16, (wave + NW * (TPW * g + 1)) * 16};
    if (t0s[0] >= nr) break;                                // wave-uniform
    rg2kg_bwd_tiles(Q, P, dO, nullptr, dq16, H, nh, Nk, scale, drop, r0, nr, h, t0s, vf, kb, dKa, dVa);
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) { comb[wave - 1][n * 4 + r][lane] = dKa[n][r]; comb[wave - 1][8 + n * 4 + r][lane] = dVa[n][r]; }
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 4 * q + r;
      if (key < Nk) {
        const int i = n * 4 + r;
        float dk = dKa[n][r], dv = dVa[n][r];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) { dk += comb[w][i][lane]; dv += comb[w][8 + i][lane]; }
        unsigned short* dst = dkv16.p + (size_t)(b * Nk + key) * dkv16.ld + h * DH + 16 * n + x;
        dst[0] = f2bf(dk);
        dst[H] = f2bf(dv);
      }
    }
}

// ------------------------------------------------------------------ paired launches (bf16 schedule)
// The two attention directions are independent of each other and each is a small-grid, latency-bound kernel, so
// they share one launch: blocks [0, nh*B) run kg2rg for (head, sample), the rest run rg2kg.  Forward: rg2kg takes
// NW (tile, head) tasks per block.  Backward: rg2kg in its owner form, nh*B blocks -- 2*nh*B blocks of NW waves in
// all, one per CU at the benchmark batch.
union PairLds {
  struct { float red[NW][16]; float ored[NW - 1][2][4][64]; } k;
  float comb[NW - 1][16][64];
};

__global__ __launch_bounds__(64 * NW) void attn_fwd_pair_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const float* __restrict__ Q2, const float* __restrict__ KV2,
    const int* __restrict__ offs, float* __restrict__ P, float* __restrict__ P2, float* __restrict__ O, float* __restrict__ O2,
    Bf16Dst o16, Bf16Dst o2_16, int H, int nh, int Nk, int B, int nbx, float scale, DropCfg drop) {
  __shared__ PairLds lds;
  const int bid = blockIdx.x;
  if (bid < nh * B) {
    kg2rg_fwd_body(Q2, KV2, offs, P2, O2, o2_16, H, nh, Nk, scale, drop, bid % nh, bid / nh, lds.k.red, lds.k.ored);
  } else {
    const int r = bid - nh * B;
    rg2kg_fwd_body(Q, KV, offs, P, O, o16, H, nh, Nk, scale, drop, r % nbx, r / nbx);
  }
}

__global__ __launch_bounds__(64 * NW) void attn_bwd_pair_kernel(
    const float* __restrict__ Q, const float* __restrict__ KV, const float* __restrict__ P, const float* __restrict__ dO,
    const float* __restrict__ Q2, const float* __restrict__ KV2, const float* __restrict__ P2, const float* __restrict__ dO2,
    const int* __restrict__ offs, Bf16Dst dq16, Bf16Dst dkv16, Bf16Dst dq2_16, Bf16Dst dkv2_16, const float* __restrict__ O2,
    int H, int nh, int Nk, int B, float scale, DropCfg drop) {
  __shared__ PairLds lds;
  const int bid = blockIdx.x;
  if (bid < nh * B) {
    kg2rg_bwd_body(Q2, KV2, P2, dO2, offs, nullptr, nullptr, dq2_16, dkv2_16, nullptr, Bf16Dst{nullptr, 0}, O2, H, nh, Nk, scale, drop,
                   bid % nh, bid / nh, lds.k.red, lds.k.ored);
  } else {
    const int r = bid - nh * B;
    rg2kg_bwd_owner_body(Q, KV, P, dO, offs, dq16, dkv16, H, nh, Nk, scale, drop, r % nh, r / nh, lds.comb);
  }
}

}  // namespace

int launch_attn_fwd_pair(const float* Q, const float* KV, const float* Q2, const float* KV2, const int* offs, float* P,
                         float* P2, Bf16Dst o16, Bf16Dst o2_16, float* O2, int B, int max_nr, int H, int nh, int Nk, DropCfg drop,
                         hipStream_t stream) {
  const int tasks = ((max_nr + 15) / 16) * nh, nbx = (tasks + NW - 1) / NW;
  hipLaunchKernelGGL(attn_fwd_pair_kernel, dim3(nh * B + nbx * B), dim3(64 * NW), 0, stream, Q, KV, Q2, KV2, offs, P, P2,
                     (float*)nullptr, O2, o16, o2_16, H, nh, Nk, B, nbx, 1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

int launch_attn_bwd_pair(const float* Q, const float* KV, const float* P, const float* dO, const float* Q2, const float* KV2,
                         const float* P2, const float* dO2, const int* offs, Bf16Dst dq16, Bf16Dst dkv16, Bf16Dst dq2_16,
                         Bf16Dst dkv2_16, const float* O2, int B, int H, int nh, int Nk, DropCfg drop, hipStream_t stream) {
  hipLaunchKernelGGL(attn_bwd_pair_kernel, dim3(2 * nh * B), dim3(64 * NW), 0, stream, Q, KV, P, dO, Q2, KV2, P2, dO2, offs,
                     dq16, dkv16, dq2_16, dkv2_16, O2, H, nh, Nk, B, 1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------ launchers
int attn_mfma_ok(int H, int nh, int Nk, int max_nr, bool kg2rg) {
  if (nh < 1 || H != nh * DH || Nk < 1 || Nk > 16) return 0;
  if (kg2rg && max_nr > 16 * NW * MAXT) return 0;
  return 1;
}

int launch_rg2kg_fwd_mfma(const float* Q, const float* KV, const int* offs, float* P, float* O, float* attn_avg,
                          int B, int T, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream, Bf16Dst o16) {
  const int tasks = ((max_nr + 15) / 16) * nh;
  hipLaunchKernelGGL(rg2kg_fwd_mfma_kernel, dim3((tasks + 3) / 4, B), dim3(256), 0, stream, Q, KV, offs, P, O,
                     o16, H, nh, Nk, 1.0f / sqrtf((float)DH), drop);
  if (attn_avg) {
    const int n = T * Nk;
    hipLaunchKernelGGL(attn_avg_site_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, P, attn_avg, T, nh, Nk,
                       (uint32_t)SITE_ATTN_RG2KG, drop);
  }
  return (int)hipGetLastError();
}

int launch_rg2kg_bwd_mfma(const float* Q, const float* KV, const float* P, const float* dO, const int* offs, float* dQ,
                          float* dKV, int B, int max_nr, int H, int nh, int Nk, DropCfg drop, hipStream_t stream, Bf16Dst dq16) {
  const int tiles = (max_nr + 15) / 16;
  hipLaunchKernelGGL(rg2kg_bwd_mfma_kernel, dim3((tiles + 4 * TPW - 1) / (4 * TPW), nh, B), dim3(256), 0, stream, Q, KV, P, dO,
                     offs, dQ, dKV, dq16, H, nh, Nk, 1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

int launch_kg2rg_fwd_mfma(const float* Q2, const float* KV2, const int* offs, float* P2, float* O2, int B, int H, int nh,
                          int Nk, DropCfg drop, hipStream_t stream, Bf16Dst o16) {
  hipLaunchKernelGGL(kg2rg_fwd_mfma_kernel, dim3(nh, B), dim3(64 * NW), 0, stream, Q2, KV2, offs, P2, O2, o16, H, nh, Nk,
                     1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

int launch_kg2rg_bwd_mfma(const float* Q2, const float* KV2, const float* P2, const float* dO2, const float* O2, const int* offs,
                          float* dQ2, float* dKV2, int B, int H, int nh, int Nk, DropCfg drop, hipStream_t stream, Bf16Dst dq2_16,
                          Bf16Dst dkv2_16, const float* dKV_done, Bf16Dst dkv_16) {
  hipLaunchKernelGGL(kg2rg_bwd_mfma_kernel, dim3(nh, B), dim3(64 * NW), 0, stream, Q2, KV2, P2, dO2, offs, dQ2, dKV2,
                     dq2_16, dkv2_16, dKV_done, dkv_16, O2, H, nh, Nk,
                     1.0f / sqrtf((float)DH), drop);
  return (int)hipGetLastError();
}

int launch_attn_avg_site(const float* Pm, float* out, int T, int nh, int Nk, uint32_t site, DropCfg drop, hipStream_t stream) {
  const int n = T * Nk;
  hipLaunchKernelGGL(attn_avg_site_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, Pm, out, T, nh, Nk, site, drop);
  return (int)hipGetLastError();
}
