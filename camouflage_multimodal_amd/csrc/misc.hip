// Bandwidth-bound pieces of the fusion path (gfx950): row maps, LayerNorm fwd/bwd,
// per-sample column means, ReLU-mask broadcast backward, the four-term loss, and the
// clip + AdamW optimizer step.  One wave (64 lanes) per row wherever a row reduction is
// needed; everything streams coalesced fp32.
#include "misc.h"
#include "gemm.h"      // launch timing hooks


namespace {

// ---------------------------------------------------------------- batch descriptor
// row -> sample map, 1 / Nr per sample, first 32-row tile of every sample (fused_rows.hip tiles a sample's rows in 32s: exclusive scan
// of ceil(Nr / 32)) and, per 32-row tile, {sample, first packed row, rows in the tile, 1 / Nr}; sample = -1 for the unused tail of
// the table (the launches size their tile range by the bound T / 32 + B).
// One launch: block k covers packed rows [256 k, 256 k + 256) and tiles [8 k', ...).  The scan of the B
// tile counts is repeated by every block (B values, 256 threads: cheaper than a second launch behind a single-block scan).
__global__ __launch_bounds__(256) void batchdesc_kernel(const int* __restrict__ offs, int* __restrict__ row_sample, float* __restrict__ inv_nr,
                                                        int* __restrict__ tile_off, int4* __restrict__ tile_desc, int B, int ntile_max) {
  __shared__ int part[256];
  __shared__ int total_s;
  const int t = threadIdx.x, per = (B + 255) / 256, b0 = min(B, t * per), b1 = min(B, b0 + per);
  const int T = offs[B];
  int s = 0;
  for (int b = b0; b < b1; ++b) s += (offs[b + 1] - offs[b] + 31) >> 5;
  // exclusive scan of the 256 chunk sums: in-wave by DPP-free shuffles, the four wave totals through LDS (a serial loop in one
  // thread was 256 dependent LDS round trips: most of this launch's 13 us)
  {
    const int lane = t & 63, w = t >> 6;
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) part[w] = incl;
    __syncthreads();
    const int w0 = part[0], w1 = part[1], w2 = part[2], w3 = part[3];
    __syncthreads();
    const int base = w == 0 ? 0 : (w == 1 ? w0 : (w == 2 ? w0 + w1 : w0 + w1 + w2));
    part[t] = base + incl - s;
    if (t == 0) total_s = w0 + w1 + w2 + w3;
  }
  __syncthreads();
  const int ntiles = total_s;
  if (blockIdx.x == 0) {
    int acc = part[t];
    for (int b = b0; b < b1; ++b) { tile_off[b] = acc; acc += (offs[b + 1] - offs[b] + 31) >> 5; }
    if (t == 0) tile_off[B] = ntiles;
    for (int b = t; b < B; b += 256) inv_nr[b] = 1.0f / (float)(offs[b + 1] - offs[b]);
  }
  // rows: sample of packed row r by binary search over the offsets
  const int r = blockIdx.x * 256 + t;
  if (r < T) {
    int lo = 0, hi = B - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offs[mid] <= r) lo = mid; else hi = mid - 1; }
    row_sample[r] = lo;
  }
  // tiles: the grid has >= ntile_max / 8 blocks (32 ntile_max rows), so 8 tiles per block cover the table; thread t < 8 takes one
  if (t < 8) {
    const int tl = blockIdx.x * 8 + t;
    if (tl < ntile_max) {
      int4 d = make_int4(-1, 0, 0, 0);
      if (tl < ntiles) {
        // sample of tile tl: the scan gives tile offsets per thread chunk; search the chunk starts, then walk the chunk
        int c = 0, hi = (B + per - 1) / per - 1;                 // last non-empty chunk whose first tile is <= tl (part[] is non-decreasing)
        while (c < hi) { const int mid = (c + hi + 1) >> 1; if (part[mid] <= tl) c = mid; else hi = mid - 1; }
        int b = min(B - 1, c * per), acc = part[c];
        while (b + 1 < B) { const int n = (offs[b + 1] - offs[b] + 31) >> 5; if (acc + n > tl) break; acc += n; ++b; }
        const int r0 = offs[b] + 32 * (tl - acc), nr = offs[b + 1] - offs[b];
        d = make_int4(b, r0, min(32, offs[b + 1] - r0), __float_as_int(1.0f / (float)nr));
      }
      tile_desc[tl] = d;
    }
  }
}

// ---------------------------------------------------------------- minibatch gather out of a device-resident dataset
// The reference's DataLoader step (train_multimodal.py:385-395: WeightedRandomSampler indices -> samples) plus its training-time
// augmentation (:173-175: with probability 1/2 per sample, N(0, 0.01^2) noise on both streams) as ONE launch: the packed row
// offsets of the minibatch, its RG rows, KG rows and labels.  Row blocks take 64 packed rows each (every block scans the B row
// counts itself); one further block per sample copies the KG rows and the labels.
constexpr int GATHER_ROWS = 64;
constexpr int GATHER_MAXB = 4096;
__device__ __forceinline__ float2 gather_noise(uint32_t a, uint32_t b, float std) {       // two N(0, std^2) values (Box-Muller)
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float r = std * sqrtf(-2.0f * __logf(u1));
  float sn, cs;
  __sincosf(6.28318530718f * u2, &sn, &cs);
  return make_float2(r * cs, r * sn);
}
__global__ __launch_bounds__(256) void gather_batch_kernel(const float* __restrict__ rg_all, const long long* __restrict__ sample_off, const float* __restrict__ kg_all,
                                                          const long long* __restrict__ y_all, const float* __restrict__ e_all, const float* __restrict__ s_all,
                                                          const long long* __restrict__ idx, int B, int T, int D, int KG, int row_blocks,
                                                          float* __restrict__ rg_out, float* __restrict__ kg_out, int* __restrict__ off_out,
                                                          long long* __restrict__ y_out, float* __restrict__ e_out, float* __restrict__ s_out,
                                                          float noise_std, uint32_t seed_lo, uint32_t seed_hi) {
  __shared__ int off[GATHER_MAXB + 1];
  __shared__ int part[256];
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= row_blocks) {                       // one block per sample: KG rows, labels
    const int b = (int)blockIdx.x - row_blocks;
    const long long src = idx[b];
    const bool aug = noise_std > 0.f && (fmix32((uint32_t)b * 0x9E3779B9u + seed_lo) ^ seed_hi) >> 31;
    for (int c = t; c < KG / 2; c += 256) {
      float2 v = *reinterpret_cast<const float2*>(kg_all + src * KG + 2 * c);
      if (aug) { const float2 n = gather_noise(fmix32(0x51ED270Bu + (uint32_t)(b * KG + 2 * c) + seed_lo) ^ seed_hi, fmix32(0x2545F491u + (uint32_t)(b * KG + 2 * c) + seed_hi) ^ seed_lo, noise_std); v.x += n.x; v.y += n.y; }
      *reinterpret_cast<float2*>(kg_out + (size_t)b * KG + 2 * c) = v;
    }
    if (t == 0) { y_out[b] = y_all[src]; e_out[b] = e_all[src]; s_out[b] = s_all[src]; }
    return;
  }
  // packed offsets: exclusive scan of the samples' row counts
  const int per = (B + 255) / 256, b0 = min(B, t * per), b1 = min(B, b0 + per);
  int sum = 0;
  for (int b = b0; b < b1; ++b) { const long long i = idx[b]; sum += (int)(sample_off[i + 1] - sample_off[i]); }
  part[t] = sum;
  __syncthreads();
  if (t == 0) { int acc = 0; for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = acc; acc += v; } off[B] = acc; }
  __syncthreads();
  {
    int acc = part[t];
    for (int b = b0; b < b1; ++b) { off[b] = acc; const long long i = idx[b]; acc += (int)(sample_off[i + 1] - sample_off[i]); }
  }
  __syncthreads();
  if (blockIdx.x == 0) for (int b = t; b <= B; b += 256) off_out[b] = off[b];
  const int lanes = D / 4, rpp = 256 / lanes;                // float4 lanes per row, rows per pass
  const int lr = t / lanes, lc = t % lanes;
  if (lr >= rpp) return;
  for (int r0 = 0; r0 < GATHER_ROWS; r0 += rpp) {
    const int r = (int)blockIdx.x * GATHER_ROWS + r0 + lr;
    if (r >= T || r >= off[B]) break;
    int lo = 0, hi = B - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= r) lo = mid; else hi = mid - 1; }
    const long long srow = sample_off[idx[lo]] + (r - off[lo]);
    float4 v = *reinterpret_cast<const float4*>(rg_all + srow * D + 4 * lc);
    if (noise_std > 0.f && ((fmix32((uint32_t)lo * 0x9E3779B9u + seed_lo) ^ seed_hi) >> 31)) {
      const uint32_t e0 = (uint32_t)r * (uint32_t)D + 4u * (uint32_t)lc;
      const float2 n0 = gather_noise(fmix32(e0 + seed_lo) ^ seed_hi, fmix32(e0 + 0x68E31DA4u + seed_hi) ^ seed_lo, noise_std);
      const float2 n1 = gather_noise(fmix32(e0 + 2u + seed_lo) ^ seed_hi, fmix32(e0 + 2u + 0x68E31DA4u + seed_hi) ^ seed_lo, noise_std);
      v.x += n0.x; v.y += n0.y; v.z += n1.x; v.w += n1.y;
    }
    *reinterpret_cast<float4*>(rg_out + (size_t)r * D + 4 * lc) = v;
  }
}

// ---------------------------------------------------------------- LayerNorm forward
// y = (u - mean) * rstd * gamma + beta, eps 1e-5 (nn.LayerNorm default, fusion_model.py:49-50)
// A wave owns LNF_R consecutive rows.  Optional fused mean pool (S.mean): the wave keeps per-lane column sums
// of its rows' outputs, flushed with atomics when the sample changes; at the end the block's four partial sums
// are combined in LDS when they belong to one sample (the usual case) -> H atomics per 16 rows.
constexpr int LNF_R = 4;
constexpr int LNF_W = 4;      // waves per block (8 waves or 8 rows per wave both measured slower)
template <int HPL>
__global__ __launch_bounds__(64 * LNF_W) void ln_fwd_kernel(LnSeg s0, LnSeg s1, int H, int nb0) {
  __shared__ float part[LNF_W][64 * HPL];
  __shared__ int scur[LNF_W];
  const bool first = (int)blockIdx.x < nb0;
  const LnSeg& S = first ? s0 : s1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rbase = ((first ? blockIdx.x : blockIdx.x - nb0) * LNF_W + wave) * LNF_R;
  const bool pool = S.mean != nullptr;
  float gam[HPL], bet[HPL], cs[HPL];
#pragma unroll
  for (int i = 0; i < HPL; ++i) {
    const int c = lane + 64 * i;
    gam[i] = c < H ? S.gamma[c] : 0.f; bet[i] = c < H ? S.beta[c] : 0.f; cs[i] = 0.f;
  }
  auto flush = [&](int sm) {
    const float inv = S.row_sample ? S.inv_n[sm] : 1.0f / (float)S.uniform_n;
#pragma unroll
    for (int i = 0; i < HPL; ++i) {
      const int c = lane + 64 * i;
      if (c < H) atomicAdd(S.mean + (size_t)sm * H + c, cs[i] * inv);
      cs[i] = 0.f;
    }
  };
  int cur = -1;
  // all rows' loads first (independent), then the per-row reductions
  float x[LNF_R][HPL];
#pragma unroll
  for (int r = 0; r < LNF_R; ++r) {
    const int row = min(rbase + r, S.rows - 1);
    const float* u = S.U + (size_t)row * H;
#pragma unroll
    for (int i = 0; i < HPL; ++i) { const int c = lane + 64 * i; x[r][i] = c < H ? u[c] : 0.f; }
  }
#pragma unroll
  for (int r = 0; r < LNF_R; ++r) {
    const int row = rbase + r;
    if (row >= S.rows) break;                                // wave-uniform
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < HPL; ++i) sum += x[r][i];
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < HPL; ++i) {
      const int c = lane + 64 * i;
      const float d = c < H ? x[r][i] - mean : 0.f;
      sq = fmaf(d, d, sq);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + 1e-5f);
    if (pool) {
      const int sm = S.row_sample ? S.row_sample[row] : row / S.uniform_n;
      if (sm != cur) { if (cur >= 0) flush(cur); cur = sm; }
    }
    float* y = S.Y + (size_t)row * H;
#pragma unroll
    for (int i = 0; i < HPL; ++i) {
      const int c = lane + 64 * i;
      if (c < H) {
        const float v = (x[r][i] - mean) * rstd * gam[i] + bet[i];
        y[c] = v;
        if (S.Y16) S.Y16[(size_t)row * H + c] = f2bf(v);
        cs[i] += v;
      }
    }
    if (lane == 0) { S.stats[2 * row] = mean; S.stats[2 * row + 1] = rstd; }
  }
  if (pool) {                                                // (block-uniform)
#pragma unroll
    for (int i = 0; i < HPL; ++i) part[wave][lane + 64 * i] = cs[i];
    if (lane == 0) scur[wave] = cur;
    __syncthreads();
    const int c0_ = scur[0];
    bool same = c0_ >= 0;
#pragma unroll
    for (int w = 1; w < LNF_W; ++w) same = same && (scur[w] == c0_ || scur[w] < 0);
    if (same) {
      const float inv = S.row_sample ? S.inv_n[c0_] : 1.0f / (float)S.uniform_n;
      for (int c = threadIdx.x; c < H; c += 64 * LNF_W) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < LNF_W; ++w) v += part[w][c];
        atomicAdd(S.mean + (size_t)c0_ * H + c, v * inv);
      }
    } else if (cur >= 0) {
      flush(cur);
    }
  }
}

// ---------------------------------------------------------------- LayerNorm backward
// dU = (g - mean(g) - xh*mean(g*xh)) * rstd, g = dY*gamma; dgamma += sum dY*xh; dbeta += sum dY
// A wave owns rows blk*4+wave, +4*nblk, ...; the next row's operands are in flight while the current
// row reduces (the rows come from HBM / Infinity Cache, ~2 us away).  VEC: H % 4 == 0, lane owns
// columns 4*lane..4*lane+3 (+256 per further quad) and moves them with 16-byte accesses.
template <int HPL, bool VEC>
struct LnCols {
  static __device__ __forceinline__ int col(int lane, int i) { return VEC ? 4 * lane + (i & 3) + 256 * (i >> 2) : lane + 64 * i; }
  static __device__ __forceinline__ void load(const float* __restrict__ p, int lane, int H, float (&x)[HPL]) {
    if constexpr (VEC) {
#pragma unroll
      for (int q = 0; q < HPL / 4; ++q) {
        const int c = 4 * lane + 256 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < H) v = *reinterpret_cast<const float4*>(p + c);
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < HPL; ++i) { const int c = lane + 64 * i; x[i] = c < H ? p[c] : 0.f; }
    }
  }
  static __device__ __forceinline__ void store16(unsigned short* __restrict__ p, int lane, int H, const float (&x)[HPL]) {
    if constexpr (VEC) {
#pragma unroll
      for (int q = 0; q < HPL / 4; ++q) {
        const int c = 4 * lane + 256 * q;
        if (c < H) *reinterpret_cast<uint2*>(p + c) = make_uint2(pack2(x[4 * q], x[4 * q + 1]), pack2(x[4 * q + 2], x[4 * q + 3]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < HPL; ++i) { const int c = lane + 64 * i; if (c < H) p[c] = f2bf(x[i]); }
    }
  }
  static __device__ __forceinline__ void store(float* __restrict__ p, int lane, int H, const float (&x)[HPL]) {
    if constexpr (VEC) {
#pragma unroll
      for (int q = 0; q < HPL / 4; ++q) {
        const int c = 4 * lane + 256 * q;
        if (c < H) *reinterpret_cast<float4*>(p + c) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < HPL; ++i) { const int c = lane + 64 * i; if (c < H) p[c] = x[i]; }
    }
  }
};

// LNB_W waves per block: the dgamma/dbeta atomics (one per column per block, all blocks on the same 2H
// addresses) cost more than the row pass itself -- 19 us with them, 8.6 us without at 4 waves x 503 blocks --
// so the block is made as fat as the LDS combine allows without stretching a wave's serial row loop.
constexpr int LNB_W = 8;
template <int HPL, bool VEC>
__global__ __launch_bounds__(64 * LNB_W) void ln_bwd_kernel(LnBwdSeg s0, LnBwdSeg s1, int H, int nb0) {
  using LC = LnCols<HPL, VEC>;
  __shared__ float red[2][LNB_W][64 * HPL];
  const bool first = (int)blockIdx.x < nb0;
  const LnBwdSeg& S = first ? s0 : s1;
  const int blk = first ? blockIdx.x : blockIdx.x - nb0;
  const int nblk = first ? nb0 : (int)gridDim.x - nb0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[HPL], db[HPL], gam[HPL];
  LC::load(S.gamma, lane, H, gam);
#pragma unroll
  for (int i = 0; i < HPL; ++i) { dg[i] = 0.f; db[i] = 0.f; }
  const int stride = nblk * LNB_W;
  int row = blk * LNB_W + wave;
  float un[HPL], dn[HPL], mean_n = 0.f, rstd_n = 0.f;
  if (row < S.rows) {
    LC::load(S.U + (size_t)row * H, lane, H, un);
    LC::load(S.dY + (size_t)row * H, lane, H, dn);
    mean_n = S.stats[2 * row]; rstd_n = S.stats[2 * row + 1];
  }
  for (; row < S.rows; row += stride) {
    float u[HPL], d[HPL];
#pragma unroll
    for (int i = 0; i < HPL; ++i) { u[i] = un[i]; d[i] = dn[i]; }
    const float mean = mean_n, rstd = rstd_n;
    const int nx = min(row + stride, S.rows - 1);          // clamped: a harmless re-read on the last trip
    LC::load(S.U + (size_t)nx * H, lane, H, un);
    LC::load(S.dY + (size_t)nx * H, lane, H, dn);
    mean_n = S.stats[2 * nx]; rstd_n = S.stats[2 * nx + 1];
    float xh[HPL], g[HPL];
    float s1_ = 0.f, s2_ = 0.f;
#pragma unroll
    for (int i = 0; i < HPL; ++i) {
      const bool ok = LC::col(lane, i) < H;
      xh[i] = ok ? (u[i] - mean) * rstd : 0.f;
      g[i] = d[i] * gam[i];
      s1_ += g[i];
      s2_ = fmaf(g[i], xh[i], s2_);
      dg[i] = fmaf(d[i], xh[i], dg[i]);
      db[i] += d[i];
    }
    const float m1 = wave_sum(s1_) / (float)H, m2 = wave_sum(s2_) / (float)H;
    float o[HPL];
#pragma unroll
    for (int i = 0; i < HPL; ++i) o[i] = (g[i] - m1 - xh[i] * m2) * rstd;
    LC::store(S.dU + (size_t)row * H, lane, H, o);
    if (S.dU16) LC::store16(S.dU16 + (size_t)row * H, lane, H, o);
  }
#pragma unroll
  for (int i = 0; i < HPL; ++i) {
    const int c = LC::col(lane, i);                        // < 64*HPL by construction
    red[0][wave][c] = dg[i]; red[1][wave][c] = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 64 * LNB_W) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_W; ++w) { a += red[0][w][c]; b += red[1][w][c]; }
    if (S.dgamma) {
      atomicAdd(S.dgamma + c, a);
      atomicAdd(S.dbeta + c, b);
    }
  }
}

// ---------------------------------------------------------------- per-sample column means
// out[b][c] += (1/n_b) * sum_{rows of b in this block's chunk} X[row][c]   (out zeroed by the caller)
// grid (chunks, B, nseg).  VEC (C, ld % 4 == 0): a thread owns a column quad and every `groups`-th row of
// the chunk, all of its 16-byte loads independent; the row groups combine through LDS -> C atomics a block.
template <bool VEC>
__global__ __launch_bounds__(256) void seg_mean_kernel(SegMean a, SegMean b2, SegMean c3, SegMean d4, int rows_per_block) {
  __shared__ float4 red[256];
  const SegMean& S = blockIdx.z == 0 ? a : (blockIdx.z == 1 ? b2 : (blockIdx.z == 2 ? c3 : d4));
  if (!S.X) return;
  const int b = blockIdx.y;
  int r0, nr;
  if (S.offs) { r0 = S.offs[b]; nr = S.offs[b + 1] - r0; } else { r0 = b * S.uniform_n; nr = S.uniform_n; }
  const int c0 = blockIdx.x * rows_per_block;
  if (c0 >= nr) return;
  const int c1 = min(nr, c0 + rows_per_block);
  const float inv = 1.0f / (float)nr;
  if constexpr (VEC) {
    const int C4 = S.C >> 2;
    const int lanes = C4 < 256 ? C4 : 256, groups = 256 / lanes;
    const int cq = threadIdx.x % lanes, grp = threadIdx.x / lanes;
    for (int q0 = 0; q0 < C4; q0 += lanes) {               // (one trip unless C > 1024)
      const int q = q0 + cq;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grp < groups && q < C4) {
        const float* base = S.X + (size_t)r0 * S.ld + 4 * q;
#pragma unroll 8
        for (int r = c0 + grp; r < c1; r += groups) {
          const float4 v = *reinterpret_cast<const float4*>(base + (size_t)r * S.ld);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
      if (groups > 1) {
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (grp == 0) {
          for (int gi = 1; gi < groups; ++gi) {
            const float4 v = red[gi * lanes + cq];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          }
        }
      }
      if (grp == 0 && q < C4) {
        float* o = S.out + (size_t)b * S.ldo + 4 * q;
        atomicAdd(o, acc.x * inv); atomicAdd(o + 1, acc.y * inv); atomicAdd(o + 2, acc.z * inv); atomicAdd(o + 3, acc.w * inv);
      }
    }
  } else {
    for (int c = threadIdx.x; c < S.C; c += 256) {
      float acc = 0.f;
      for (int r = c0; r < c1; ++r) acc += S.X[(size_t)(r0 + r) * S.ld + c];
      atomicAdd(S.out + (size_t)b * S.ldo + c, acc * inv);
    }
  }
}

// ---------------------------------------------------------------- ReLU-mask broadcast backward
// dst[t][c] = v[sample(t)][c] * inv_n[sample(t)] * (act[t][c] > 0 ? scale : 0)
// W = 4: C % 4 == 0, one 16-byte quad per thread.
template <int W>
__global__ __launch_bounds__(256) void relu_bcast_bwd_kernel(BcastSeg s0, BcastSeg s1, int C, long n0, float scale) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * W;
  const bool first = i < n0;
  const BcastSeg& S = first ? s0 : s1;
  const long k = first ? i : i - n0;
  if (k >= (long)S.rows * C) return;
  const int t = (int)(k / C), c = (int)(k - (long)t * C);
  int sb; float inv;
  if (S.row_sample) { sb = S.row_sample[t]; inv = S.inv_n[sb]; } else { sb = t / S.uniform_n; inv = 1.0f / (float)S.uniform_n; }
  inv *= scale;
  if constexpr (W == 4) {
    float4 a;
    if (S.act16) {                                             // bf16 activation: sign and zero survive the rounding
      const uint2 r = *reinterpret_cast<const uint2*>(S.act16 + k);
      a = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
    } else {
      a = *reinterpret_cast<const float4*>(S.act + k);
    }
    const float4 v = *reinterpret_cast<const float4*>(S.v + (size_t)sb * S.ldv + c);
    float4 o;
    o.x = a.x > 0.f ? v.x * inv : 0.f; o.y = a.y > 0.f ? v.y * inv : 0.f;
    o.z = a.z > 0.f ? v.z * inv : 0.f; o.w = a.w > 0.f ? v.w * inv : 0.f;
    if (S.dst16) *reinterpret_cast<uint2*>(S.dst16 + k) = make_uint2(pack2(o.x, o.y), pack2(o.z, o.w));
    else *reinterpret_cast<float4*>(S.dst + k) = o;
  } else {
    const float a = S.act16 ? __uint_as_float((uint32_t)S.act16[k] << 16) : S.act[k];
    const float o = a > 0.f ? S.v[(size_t)sb * S.ldv + c] * inv : 0.f;
    if (S.dst16) S.dst16[k] = f2bf(o); else S.dst[k] = o;
  }
}

// ---------------------------------------------------------------- d(outs) -> d(pre-activation)
__global__ void head_out_grad_kernel(const float* __restrict__ outs, const float* __restrict__ d_outs,
                                     float* __restrict__ d_logits, int B, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const int c = i % W;
  float g = d_outs[i];
  if (c == W - 1) { const float s = outs[i]; g *= s * (1.0f - s); }   // score head's Sigmoid (fusion_model.py:234)
  d_logits[i] = g;
}

// ---------------------------------------------------------------- loss
// per sample: 3*focal(mask,y) + CE(inst,y) + .5*BCEWithLogits(edge,e) + .3*MSE(score,s)
// (train_multimodal.py:29-57, 256-268), each at batch size 1.
// o: the sample's 2C+2 outputs (score column post-sigmoid).  d_outs / d_pre rows may be null; they must not alias o.
// FAST: hardware exp / log / reciprocal (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ulp each) instead of the correctly rounded library
// calls -- a third of the instructions; used by the one-launch tail, where sixteen lanes run this on every block's critical path.
template <bool FAST = false>
__device__ __forceinline__ void loss_sample(const float* __restrict__ o, int yb, float eb, float sb, int C,
                                            float* __restrict__ terms4, float* __restrict__ d_outs, float* __restrict__ d_pre,
                                            int* __restrict__ pred) {
  const int W = 2 * C + 2;
  const float sc = o[W - 1];
  auto ex = [](float x) { return FAST ? __expf(x) : expf(x); };
  auto lg = [](float x) { return FAST ? __logf(x) : logf(x); };
  auto rc = [](float x) { return FAST ? __frcp_rn(x) : 1.0f / x; };
  auto put = [&](int k, float val) {        // gradient w.r.t. output k; d_pre: score column w.r.t. the pre-sigmoid value
    if (d_outs) d_outs[k] = val;
    if (d_pre) d_pre[k] = (k == W - 1) ? val * sc * (1.0f - sc) : val;
  };
  // focal on mask logits.  A label outside [0, C) (torch raises on one) poisons this sample's loss and gradients
  // with NaN instead of reading out of bounds; the host wrappers validate labels before the call.
  const bool ybad = yb < 0 || yb >= C;
  if (ybad) yb = 0;
  const float poison = ybad ? __builtin_nanf("") : 0.f;
  {
    float mx = -INFINITY; int am = 0;
    for (int k = 0; k < C; ++k) if (o[k] > mx) { mx = o[k]; am = k; }
    float z = 0.f;
    for (int k = 0; k < C; ++k) z += ex(o[k] - mx);
    const float iz = rc(z);
    // ce from the log-sum-exp (finite when pt underflows, like torch's log_softmax); the gradient is written
    // without the division by pt:  d l / d logit_k = at * (-3 om^2 ce pt - om^3) * (delta_ky - p_k)
    const float ce = -(o[yb] - mx - lg(z));
    const float pt = FAST ? ex(o[yb] - mx) * iz : expf(o[yb] - mx) / z;
    const float at = yb == 1 ? 0.75f : 0.25f;
    const float om = 1.0f - pt;
    terms4[0] = 3.0f * at * om * om * om * ce + poison;
    const float dl = at * (-3.0f * om * om * ce * pt - om * om * om);
    for (int k = 0; k < C; ++k) {
      const float pk = FAST ? ex(o[k] - mx) * iz : expf(o[k] - mx) / z;
      put(k, 3.0f * dl * ((k == yb ? 1.0f : 0.0f) - pk) + poison);
    }
    if (pred) *pred = am;
  }
  // cross entropy on instance logits
  {
    const float* oi = o + C;
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) mx = fmaxf(mx, oi[k]);
    float z = 0.f;
    for (int k = 0; k < C; ++k) z += ex(oi[k] - mx);
    const float iz = rc(z);
    terms4[1] = -(oi[yb] - mx - lg(z));
    for (int k = 0; k < C; ++k) put(C + k, (FAST ? ex(oi[k] - mx) * iz : expf(oi[k] - mx) / z) - (k == yb ? 1.0f : 0.0f));
  }
  // BCE with logits on edge
  {
    const float x = o[2 * C], t = eb;
    terms4[2] = 0.5f * (fmaxf(x, 0.f) - x * t + (FAST ? __logf(1.0f + __expf(-fabsf(x))) : log1pf(expf(-fabsf(x)))));
    put(2 * C, 0.5f * ((FAST ? rc(1.0f + ex(-x)) : 1.0f / (1.0f + expf(-x))) - t));
  }
  // MSE on the (post-sigmoid) score
  {
    const float x = o[2 * C + 1], t = sb;
    terms4[3] = 0.3f * (x - t) * (x - t);
    put(2 * C + 1, 0.6f * (x - t));
  }
}

__global__ void loss_kernel(const float* __restrict__ outs, const long long* __restrict__ y,
                            const float* __restrict__ e, const float* __restrict__ s, int B, int C,
                            float* __restrict__ terms, float* __restrict__ d_outs, float* __restrict__ d_pre,
                            int* __restrict__ pred) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int W = 2 * C + 2;
  loss_sample(outs + (size_t)b * W, (int)y[b], e[b], s[b], C, terms + 4 * b, d_outs ? d_outs + (size_t)b * W : nullptr,
              d_pre ? d_pre + (size_t)b * W : nullptr, pred ? pred + b : nullptr);
}

// ---------------------------------------------------------------- head output layer + loss + its backward
// The last forward launch (four [B, Fh] x [nout, Fh]^T products, nout in {C, C, 1, 1}), the loss and the first
// backward launch (d hidden = (d logits . W) * relu/dropout mask, dW = d logits^T . hidden, db) are 100 kFLOP in
// all; as three launches they cost three launch floors.  One block per sample, wave x owns head x; only the
// output-layer weight gradients cross samples (atomics).
// S consecutive samples per block (large batches): the output layers' weight-gradient contributions are summed over the block's
// samples in registers (ACC: nout <= 8) and leave as ONE atomic per element and block -- with one sample per block a batch of 256 is
// 256-way contention on the same 770 addresses (35 us at B = 256).
template <bool ACC>
__global__ __launch_bounds__(256) void heads_loss_kernel(const float* __restrict__ hid, HeadsOut hp,
                                                         const long long* __restrict__ y, const float* __restrict__ e,
                                                         const float* __restrict__ s, int B, int S, int C, int Fh, float scale,
                                                         float* __restrict__ outs, float* __restrict__ terms,
                                                         int* __restrict__ pred, float* __restrict__ dhid, float* __restrict__ dlog) {
  // dlog != null: the output layers' weight and bias gradients are left to a GEMM launch of the caller, which gets the gradient of
  // the 2C + 2 pre-activations here ([B][2C+2]) -- N blocks adding to the same 770 addresses serialise at the memory side
  __shared__ float so[HEADS_MAXW], sd[HEADS_MAXW];
  const int x = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int W = 2 * C + 2, nout = x < 2 ? C : 1, coff = x == 0 ? 0 : (x == 1 ? C : (x == 2 ? 2 * C : 2 * C + 1));
  // (static indices + selects: indexing the by-value argument struct with the runtime wave id would move it to scratch)
  const float* Wx = x == 0 ? hp.W[0] : (x == 1 ? hp.W[1] : (x == 2 ? hp.W[2] : hp.W[3]));
  const float* bx = x == 0 ? hp.b[0] : (x == 1 ? hp.b[1] : (x == 2 ? hp.b[2] : hp.b[3]));
  float* gWx = x == 0 ? hp.gW[0] : (x == 1 ? hp.gW[1] : (x == 2 ? hp.gW[2] : hp.gW[3]));
  float* gbx = x == 0 ? hp.gb[0] : (x == 1 ? hp.gb[1] : (x == 2 ? hp.gb[2] : hp.gb[3]));
  const float bias0 = bx[0];
  constexpr int HC = 4;                                       // hidden values cached per lane (Fh <= 256), else re-read
  constexpr int AO = ACC ? 8 : 1;
  float gacc[AO][HC], gbacc = 0.f;
#pragma unroll
  for (int o = 0; o < AO; ++o)
#pragma unroll
    for (int i = 0; i < HC; ++i) gacc[o][i] = 0.f;
  for (int si = 0; si < S; ++si) {
    const int b = blockIdx.x * S + si;
    if (b >= B) break;                                        // (block-uniform)
    const float* h = hid + (size_t)b * 4 * Fh + x * Fh;
    // labels first: their load latency hides under the dot products instead of following the barrier
    const int yb = (int)y[b]; const float eb = e[b], sb = s[b];
    float hv[HC];
#pragma unroll
    for (int i = 0; i < HC; ++i) { const int c = lane + 64 * i; hv[i] = c < Fh ? h[c] : 0.f; }
    for (int o = 0; o < nout; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < HC; ++i) { const int c = lane + 64 * i; if (c < Fh) acc = fmaf(hv[i], Wx[(size_t)o * Fh + c], acc); }
      for (int c = lane + 64 * HC; c < Fh; c += 64) acc = fmaf(h[c], Wx[(size_t)o * Fh + c], acc);
      acc = wave_sum(acc);
      if (lane == 0) {
        float v = acc + (o == 0 ? bias0 : bx[o]);
        if (x == 3) v = 1.0f / (1.0f + __expf(-v));
        so[coff + o] = v;
        outs[(size_t)b * W + coff + o] = v;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) loss_sample(so, yb, eb, sb, C, terms + 4 * b, nullptr, sd, pred ? pred + b : nullptr);
    __syncthreads();
    if (dlog && threadIdx.x < W) dlog[(size_t)b * W + threadIdx.x] = sd[threadIdx.x];
    auto back = [&](int c, float hval, int i) {
      float dh = 0.f;
      if (dlog) {
        for (int o = 0; o < nout; ++o) dh = fmaf(sd[coff + o], Wx[(size_t)o * Fh + c], dh);
      } else if constexpr (ACC) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          if (o < nout) {
            const float d = sd[coff + o];
            dh = fmaf(d, Wx[(size_t)o * Fh + c], dh);
            if (i >= 0) gacc[o][i >= 0 ? i : 0] = fmaf(d, hval, gacc[o][i >= 0 ? i : 0]); else atomicAdd(gWx + (size_t)o * Fh + c, d * hval);
          }
        }
      } else {
        for (int o = 0; o < nout; ++o) {
          const float d = sd[coff + o];
          dh = fmaf(d, Wx[(size_t)o * Fh + c], dh);
          atomicAdd(gWx + (size_t)o * Fh + c, d * hval);
        }
      }
      dhid[(size_t)b * 4 * Fh + x * Fh + c] = hval > 0.f ? dh * scale : 0.f;
    };
#pragma unroll
    for (int i = 0; i < HC; ++i) { const int c = lane + 64 * i; if (c < Fh) back(c, hv[i], i); }     // (static register indices)
    for (int c = lane + 64 * HC; c < Fh; c += 64) back(c, h[c], -1);
    if (!dlog && lane < nout) gbacc += sd[coff + lane];
    __syncthreads();                                          // (so / sd are rewritten by the next sample)
  }
  if (dlog) return;
  if constexpr (ACC) {
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int i = 0; i < HC; ++i) { const int c = lane + 64 * i; if (o < nout && c < Fh) atomicAdd(gWx + (size_t)o * Fh + c, gacc[o][i]); }
  }
  if (lane < nout) atomicAdd(gbx + lane, gbacc);
}


// ---------------------------------------------------------------- the whole per-sample tail in ONE launch
// fusion_model.py:134-139 (pooled FFN second layers, fusion MLP), :208-235 (four heads), the 4-term loss
// (train_multimodal.py:256-268) and all of it backwards, for B <= 16 samples at hidden 256.
//
// As separate launches the tail is ten dependent steps of a few 16x16 tiles each: ~80 us of a 210 us training step at
// B = 16, almost all of it launch boundaries and cold weight fetches.  Per-CU bandwidth rules out one block per sample
// (2.4 MB of fp32 weights each).  Here TG = 64 co-resident blocks split every layer's weights between them Megatron
// style, so that consecutive linear layers need ONE exchange, not two:
//   block g owns comb columns [8g, 8g+8) (of the 512 = RG | KG pooled features) and fused columns [4g, 4g+4);
//   L1 (pooled FFN layer) is split by OUTPUT column, L2 (fusion layer 0) by INPUT column -> partial sums of its 256
//   outputs, all-reduced with fp32 atomics;  L3 (fusion layer 3) by output column, L4 (heads' hidden layers) by input
//   column -> partial sums of their 512 outputs, all-reduced;  the head output layers and the loss are small enough
//   to be recomputed by every block.  Backward mirrors it: one all-reduce (d F1), and the last partial sums
//   (d mean H) are simply complete when the kernel ends.  Every weight gradient element has exactly one writer.
// An all-reduce = every block's atomics acknowledged, an arrival counter, a BOUNDED spin on it by one lane per block (the
// grid is 64 blocks of 256 threads: co-resident on any MI355X partition of >= 64 CUs; on timeout the kernel raises
// counters[3] and finishes with wrong numbers instead of hanging), an agent-scope acquire, then plain loads: float atomics
// execute at the memory side and leave no line in any L2.
constexpr int TG = 64;
// sticky health counter: arrival waits of the one-launch tail that gave up (a block of the launch never arrived within the
// spin bound -- it cannot happen while the launch's 64 blocks are co-resident, i.e. one process per GPU; a step that hit it
// has wrong results).  Read by camo_tail_timeouts().
__device__ unsigned int g_tail_timeouts = 0;
__device__ __forceinline__ void tstamp(unsigned long long* stamps, int k) {
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 32 + k] = __builtin_amdgcn_s_memrealtime();
}
// A wait that gives up must not become a parameter update: besides raising the call's flag and the sticky counter it sets
// g_tail_poison, which the step's norm launch (sumsq_kernel, next on the stream) turns into a NaN gradient norm -- and the optimizer
// kernels skip a step whose norm is not finite (clip_adamw_kernel / adamw_shadow_kernel).  The block that writes the outputs and
// the loss terms writes NaN there when its own wait gave up (lds_flag), so the caller sees the step in band as well.
__device__ unsigned int g_tail_poison = 0;      // set by a wait that gave up, consumed by the next sumsq_kernel (same stream: the step's norm launch)
__device__ __forceinline__ void tail_arrive_wait(unsigned int* counter, unsigned int* timeout, int* lds_flag, bool skip_arrival = false) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's atomics are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!skip_arrival) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)TG) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 21)) {
        __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&g_tail_timeouts, 1u);
        __hip_atomic_store(&g_tail_poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_flag = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

constexpr int TF_MAXB = 16;           // samples per group
constexpr int TF_MAXGROUPS = 3;       // 64 co-resident blocks each.  Measured (MI355X, training step): B = 24 -7 %, 32 -5 %, 48 -2 %, 64 +-0 against
                                      // the separate launches -- every group adds its 64-way atomic all-reduces to the same L2 atomic units
constexpr int TF_MAXW = 18;           // 2 * num_classes + 2 (num_classes <= 8)
constexpr int TF_LDS_FLOATS = 2 * TF_MAXB * 512 + 2 * TF_MAXB * 256 + 2048 + 2048 + 3 * 128 + 64 + 64 + 2 * TF_MAXB * TF_MAXW + TF_MAXW * 129 + 2 +
                             8 * 512 + 4 * 256;      // (148 KB)

// The kernel runs once per step on 64 CUs that have just run other code, one block per CU: what it costs is latency, not
// throughput.  So: 1024 threads -- four "quarters" of 256 threads, quarter q walking samples 4q..4q+3 through every
// per-sample phase (four independent chains per thread hide the LDS latency, four waves per SIMD hide each other's), and
// the sums over the batch (weight gradients) split by output column between the quarters instead; every global read whose
// address is known early is issued early (one memory round trip at the start, a second one hidden behind the first
// all-reduce), and the passes that fetch an all-reduce's result have all their loads in flight together.  The LDS images
// are padded with zero rows to a multiple of 4 samples.
__device__ __forceinline__ void wave_sum4(float (&acc)[4]) {           // totals valid in lane 63
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = wave_sum_lane63(acc[j]);
}

constexpr int TF_THREADS = 1024;

__global__ __launch_bounds__(TF_THREADS) void tail_fused_kernel(const TailFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int s_timed_out;                                // a wait of this block gave up (tail_arrive_wait)
  if (threadIdx.x == 0) s_timed_out = 0;                     // (ordered before its first use by the barriers of the first wait)
  // Sample groups: block (g, sg) = column slice g of samples [16 sg, 16 sg + 16).  Groups never wait for each other (their own
  // arrival counters), so a batch of up to 16 * groups samples costs one group's latency while the chip has 64 CUs per group;
  // blocks are numbered group by group, so the blocks a group waits for are never queued behind blocks that wait for them.
  // With more than one group the four big weight gradients (sums over every group's samples) are left to one batched GEMM launch
  // behind this kernel, which gets the per-sample operands from the *_out copies; the small output-layer gradients use atomics.
  const bool wg = a.mode && gridDim.x > TG;               // (write the copies, skip the big weight gradients)
  const int tid = threadIdx.x, lane = tid & 63, g = blockIdx.x & (TG - 1), sg = blockIdx.x >> 6, C = a.C, Wd = 2 * C + 2;
  const bool multi = gridDim.x > TG;
  const int sb = TF_MAXB * sg, B = min(TF_MAXB, a.B - sb);
  // (the group's views of the per-sample arrays: locals -- a modified copy of the argument struct would leave the scalar registers)
  const float* const gYmean = a.Ymean + (size_t)sb * 256; const float* const gY2mean = a.Y2mean + (size_t)sb * 256;
  const float* const gH1mean = a.H1mean + (size_t)sb * 512; const float* const gH2mean = a.H2mean + (size_t)sb * 512;
  float* const gF1sum = a.F1sum + (size_t)sb * 256; float* const ghidsum = a.hidsum + (size_t)sb * 512; float* const gdF1sum = a.dF1sum + (size_t)sb * 256;
  float* const gouts = a.outs + (size_t)sb * Wd;
  const bool md = a.mode != 0;
  const long long* const gy = md ? a.y + sb : nullptr; const float* const ge = md ? a.e + sb : nullptr; const float* const gs = md ? a.s + sb : nullptr;
  float* const gterms = md ? a.terms + 4 * sb : nullptr; int* const gpred = md && a.pred ? a.pred + sb : nullptr;
  float* const gdcomb = md ? a.dcomb + (size_t)sb * 512 : nullptr;
  float* const gdHm1 = md ? a.dHm1 + (size_t)sb * 512 : nullptr; float* const gdHm2 = md ? a.dHm2 + (size_t)sb * 512 : nullptr;
  float* const gcomb_out = wg ? a.comb_out + (size_t)sb * 512 : nullptr; float* const gF1_out = wg ? a.F1_out + (size_t)sb * 256 : nullptr;
  float* const gfused_out = wg ? a.fused_out + (size_t)sb * 256 : nullptr; float* const gdhid_out = wg ? a.dhid_out + (size_t)sb * 512 : nullptr;
  float* const gdfused_out = wg ? a.dfused_out + (size_t)sb * 256 : nullptr; float* const gdF1_out = wg ? a.dF1_out + (size_t)sb * 256 : nullptr;
  unsigned int* const cnt = a.counters + 4 * sg;            // arrivals: words 4 sg .. 4 sg + 2 (word 3: the launch's gave-up flag)
  unsigned int* const gaveup = a.counters + 3;
  auto put = [&](float* p, float v) { if (multi) atomicAdd(p, v); else *p = v; };
  auto put2 = [&](float2* p, float2 v) { if (multi) { atomicAdd(&p->x, v.x); atomicAdd(&p->y, v.y); } else *p = v; };
  const int q = __builtin_amdgcn_readfirstlane(tid >> 8), t = tid & 255, wv = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int b0 = 4 * q, B4 = (B + 3) & ~3;
  const bool active = b0 < B;                               // (wave-uniform)
  float* R1 = sm;                       // [16][512] pooled FFN activations of this block's stream (kept to the end)
  float* R2 = R1 + TF_MAXB * 512;       // [16][512] the heads' hidden activations, then (in place) the gradient of their pre-activations
  float* F1 = R2 + TF_MAXB * 512;       // [16][256] fusion layer 0 activations (after ReLU and dropout)
  float* DF = F1 + TF_MAXB * 256;       // [16][256] gradient of fusion layer 0's pre-activations
  float* wA = DF + TF_MAXB * 256;       // [512][4]  this block's columns of the heads' hidden-layer weights
  float* wB = wA + 2048;                // [256][8]  this block's columns of fusion layer 0's weights
  float* combS = wB + 2048; float* dcombS = combS + 128; float* ymS = dcombS + 128; float* fusedS = ymS + 128; float* dfusedS = fusedS + 64;
  float* outsS = dfusedS + 64; float* dpre = outsS + TF_MAXB * TF_MAXW;
  float* w3s = dpre + TF_MAXB * TF_MAXW;                     // [Wd][129]: output-layer weight rows + bias, in output-column order
  float* W3L = w3s + ((TF_MAXW * 129 + 3) & ~3);             // [8][512] this block's rows of its stream's pooled FFN layer (L1, and d(mean H))
  float* WfL = W3L + 8 * 512;                                // [4][256] this block's rows of fusion layer 3 (L3, and d F1)
  const bool kgs = g >= 32;                                 // comb columns 256.. are the KG stream's
  const int c0 = 8 * (g & 31), ccol = (kgs ? 256 : 0) + c0;
  const float* Hmean = kgs ? gH2mean : gH1mean; const float* Ymean = kgs ? gY2mean : gYmean;
  const float* W3s = kgs ? a.W23 : a.W13; const float* b3s = kgs ? a.b23 : a.b13;
  float* gW3s = kgs ? a.gW23 : a.gW13; float* gb3s = kgs ? a.gb23 : a.gb13; float* dHm = kgs ? gdHm2 : gdHm1;
  const bool dodrop = a.drop.p > 0.f;
  const float dscale = a.drop.scale;
  auto headW0 = [&](int x) { return x == 0 ? a.Wh0[0] : (x == 1 ? a.Wh0[1] : (x == 2 ? a.Wh0[2] : a.Wh0[3])); };
  auto headB0 = [&](int x) { return x == 0 ? a.bh0[0] : (x == 1 ? a.bh0[1] : (x == 2 ? a.bh0[2] : a.bh0[3])); };
  auto headW3 = [&](int x) { return x == 0 ? a.Wh3[0] : (x == 1 ? a.Wh3[1] : (x == 2 ? a.Wh3[2] : a.Wh3[3])); };
  auto headB3 = [&](int x) { return x == 0 ? a.bh3[0] : (x == 1 ? a.bh3[1] : (x == 2 ? a.bh3[2] : a.bh3[3])); };
  auto gheadW0 = [&](int x) { return x == 0 ? a.gWh0[0] : (x == 1 ? a.gWh0[1] : (x == 2 ? a.gWh0[2] : a.gWh0[3])); };
  auto gheadB0 = [&](int x) { return x == 0 ? a.gbh0[0] : (x == 1 ? a.gbh0[1] : (x == 2 ? a.gbh0[2] : a.gbh0[3])); };
  auto gheadW3 = [&](int x) { return x == 0 ? a.gWh3[0] : (x == 1 ? a.gWh3[1] : (x == 2 ? a.gWh3[2] : a.gWh3[3])); };
  auto gheadB3 = [&](int x) { return x == 0 ? a.gbh3[0] : (x == 1 ? a.gbh3[1] : (x == 2 ? a.gbh3[2] : a.gbh3[3])); };
  auto head_of = [&](int o, int& x, int& oo) {              // output column o of [mask C | instance C | edge | score]
    if (o < C) { x = 0; oo = o; } else if (o < 2 * C) { x = 1; oo = o - C; } else if (o == 2 * C) { x = 2; oo = 0; } else { x = 3; oo = 0; }
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  tstamp(a.stamps, 0);
  // ---- every global read of the forward that depends on nothing: issued now, one memory round trip for all of them; what
  // the first layer needs comes first (loads return in order), the rest lands while that layer runs
  // (weight rows go through LDS: one 16-byte load per thread instead of 40 dword loads that four quarters would repeat)
  const float4 w3l = *reinterpret_cast<const float4*>(W3s + (size_t)(c0 + (tid >> 7)) * 512 + 4 * (tid & 127));
  float4 hv[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) { const int i = tid + TF_THREADS * it; hv[it] = i < B * 128 ? reinterpret_cast<const float4*>(Hmean)[i] : z4; }
  const float ymv = tid < B * 8 ? Ymean[(tid >> 3) * 256 + c0 + (tid & 7)] + b3s[c0 + (tid & 7)] : 0.f;
  const float4 wfl = tid < 256 ? *reinterpret_cast<const float4*>(a.Wfu3 + (size_t)(4 * g + (tid >> 6)) * 256 + 4 * (tid & 63)) : z4;
  float4 wb0 = z4, wb1 = z4, wav = z4;
  if (q == 0) {
    const float4* p = reinterpret_cast<const float4*>(a.Wfu0 + (size_t)t * 512 + ccol);
    wb0 = p[0]; wb1 = p[1];
  } else if (q < 3) {
    const int m = tid - 256;
    wav = *reinterpret_cast<const float4*>(headW0(m >> 7) + (size_t)(m & 127) * 256 + 4 * g);
  }
  float w3v[3];                                              // (Wd * 129 <= 2322 elements: at most 3 per thread)
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = tid + TF_THREADS * it;
    w3v[it] = 0.f;
    if (p < Wd * 129) {
      const int o = p / 129, k = p - 129 * o;
      int x, oo; head_of(o, x, oo);
      w3v[it] = k < 128 ? headW3(x)[(size_t)oo * 128 + k] : headB3(x)[oo];
    }
  }
  const float bias2 = g == 0 ? a.bfu0[t] : 0.f;
  const float bias3 = a.bfu3[4 * g + wv];
  const float bias4a = g == 0 ? headB0(t >> 7)[t & 127] : 0.f, bias4b = g == 0 ? headB0(2 + (t >> 7))[t & 127] : 0.f;
  // operands of the backward and the old values of every gradient this block adds to: issued behind the forward's
  // reads (loads return in order: the forward never waits for these), so that no += later waits for its read.  The sums over the batch are split between the quarters by column:
  //   hidden-layer weights [512 m][4 c]: m = t + 256 (q & 1), columns 2 (q >> 1), +1;  fusion layer 3 [256 n][4 c]: column q;
  //   fusion layer 0 [256 n][8 c]: columns 2q, 2q + 1;  pooled FFN layer [512 k][8 c]: k = t + 256 (q & 1), columns 4 (q >> 1)..+3
  const int qm = t + 256 * (q & 1), qh = q >> 1;
  const int hx = (8 * g) >> 7, hmm0 = (8 * g) & 127, hnout = hx < 2 ? C : 1, hcoff = hx == 0 ? 0 : (hx == 1 ? C : (hx == 2 ? 2 * C : 2 * C + 1));
  float* dW3h = gheadW3(hx) + (size_t)(tid >> 3) * 128 + hmm0 + (tid & 7);
  float2* dstW0 = reinterpret_cast<float2*>(gheadW0(qm >> 7) + (size_t)(qm & 127) * 256 + 4 * g + 2 * qh);
  float* dstfu3 = a.gWfu3 + (size_t)(4 * g + q) * 256 + t;
  float2* dstfu0 = reinterpret_cast<float2*>(a.gWfu0 + (size_t)t * 512 + ccol + 2 * q);
  float oW3h = 0.f, ob3h = 0.f, ob0h = 0.f, obfu3 = 0.f, obfu0 = 0.f, ob3s = 0.f, ofu3 = 0.f, ogw3[4] = {0.f, 0.f, 0.f, 0.f};
  float2 oW0 = make_float2(0.f, 0.f), ofu0 = make_float2(0.f, 0.f);
  if (a.mode && !multi) {
    if (tid < hnout * 8) oW3h = *dW3h;
    if (g == 0 && tid >= 128 && tid < 128 + Wd) { int xx, oo; head_of(tid - 128, xx, oo); ob3h = gheadB3(xx)[oo]; }
    oW0 = *dstW0; ofu3 = *dstfu3; ofu0 = *dstfu0;
#pragma unroll
    for (int c = 0; c < 4; ++c) ogw3[c] = gW3s[(size_t)(c0 + 4 * qh + c) * 512 + qm];
    if (tid < 8) { ob0h = gheadB0((8 * g + tid) >> 7)[(8 * g + tid) & 127]; ob3s = gb3s[c0 + tid]; }
    if (tid < 4) { obfu3 = a.gbfu3[4 * g + tid]; obfu0 = a.gbfu0[4 * g + tid]; }
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) { const int i = tid + TF_THREADS * it; if (i < B4 * 128) reinterpret_cast<float4*>(R1)[i] = hv[it]; }
  reinterpret_cast<float4*>(W3L)[tid] = w3l;
  if (tid < 128) ymS[tid] = ymv;
  __syncthreads();
  tstamp(a.stamps, 1);
  // ---- L1 (by output column): comb[b][c] = mean Y + (mean H) . W3^T + b3, c in this block's 8 columns; wave wv: columns 2wv, 2wv+1
  if (active) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(W3L[(2 * wv + cc) * 512 + lane + 64 * i], R1[(b0 + j) * 512 + lane + 64 * i], acc[j]);
      wave_sum4(acc);
      if (lane == 63) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float cv = acc[j] + ymS[(b0 + j) * 8 + 2 * wv + cc];
          combS[(b0 + j) * 8 + 2 * wv + cc] = cv;
          if (wg && b0 + j < B) gcomb_out[(size_t)(b0 + j) * 512 + ccol + 2 * wv + cc] = cv;
        }
      }
    }
  }
  // the operands of the later layers have landed meanwhile
  if (q == 0) { reinterpret_cast<float4*>(wB)[2 * t] = wb0; reinterpret_cast<float4*>(wB)[2 * t + 1] = wb1; }
  else if (q < 3) reinterpret_cast<float4*>(wA)[tid - 256] = wav;
#pragma unroll
  for (int it = 0; it < 3; ++it) { const int p = tid + TF_THREADS * it; if (p < Wd * 129) w3s[p] = w3v[it]; }
  if (tid < 256) reinterpret_cast<float4*>(WfL)[tid] = wfl;
  __syncthreads();
  tstamp(a.stamps, 2);
  // ---- L2 (by input column): partial sums of fusion layer 0's 256 outputs; thread t owns output t
  if (active) {
    float w8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) w8[c] = wB[t * 8 + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 ca = reinterpret_cast<const float4*>(combS)[(b0 + j) * 2], cb = reinterpret_cast<const float4*>(combS)[(b0 + j) * 2 + 1];
      const float acc = bias2 + w8[0] * ca.x + w8[1] * ca.y + w8[2] * ca.z + w8[3] * ca.w + w8[4] * cb.x + w8[5] * cb.y + w8[6] * cb.z + w8[7] * cb.w;
      if (b0 + j < B) atomicAdd(gF1sum + (b0 + j) * 256 + t, acc);
    }
  }
  tstamp(a.stamps, 3);
  tail_arrive_wait(cnt + 0, gaveup, &s_timed_out, a.debug_skip == (int)blockIdx.x + 1);
  tstamp(a.stamps, 4);
  {
    const float4 v4 = tid < B * 64 ? reinterpret_cast<const float4*>(gF1sum)[tid] : z4;
    if (tid < B4 * 64) {
      float v[4] = {fmaxf(v4.x, 0.f), fmaxf(v4.y, 0.f), fmaxf(v4.z, 0.f), fmaxf(v4.w, 0.f)};
      if (dodrop) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= drop_mult(a.drop, SITE_FUSE, (uint32_t)(sb * 256 + 4 * tid + e));        // (element index b * 256 + n)
      }
      reinterpret_cast<float4*>(F1)[tid] = make_float4(v[0], v[1], v[2], v[3]);
      if (wg && g == 0 && tid < B * 64) reinterpret_cast<float4*>(gF1_out)[tid] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();
  tstamp(a.stamps, 5);
  // ---- L3 (by output column): fused[b][4g + wv], wave wv
  if (active) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(WfL[wv * 256 + lane + 64 * i], F1[(b0 + j) * 256 + lane + 64 * i], acc[j]);
    wave_sum4(acc);
    if (lane == 63) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fusedS[(b0 + j) * 4 + wv] = acc[j] + bias3;
        if (wg && b0 + j < B) gfused_out[(size_t)(b0 + j) * 256 + 4 * g + wv] = acc[j] + bias3;
      }
    }
  }
  __syncthreads();
  tstamp(a.stamps, 6);
  // ---- L4 (by input column): partial sums of the 4 x 128 hidden units of the heads; thread t owns units t and t + 256
  if (active) {
    const float4 wa0 = reinterpret_cast<const float4*>(wA)[t], wa1 = reinterpret_cast<const float4*>(wA)[t + 256];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 f = reinterpret_cast<const float4*>(fusedS)[b0 + j];
      if (b0 + j < B) {
        atomicAdd(ghidsum + (b0 + j) * 512 + t, bias4a + wa0.x * f.x + wa0.y * f.y + wa0.z * f.z + wa0.w * f.w);
        atomicAdd(ghidsum + (b0 + j) * 512 + t + 256, bias4b + wa1.x * f.x + wa1.y * f.y + wa1.z * f.z + wa1.w * f.w);
      }
    }
  }
  tstamp(a.stamps, 7);
  tail_arrive_wait(cnt + 1, gaveup, &s_timed_out);
  tstamp(a.stamps, 8);
  {
    float4 v4[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) { const int i = tid + TF_THREADS * it; v4[it] = i < B * 128 ? reinterpret_cast<const float4*>(ghidsum)[i] : z4; }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + TF_THREADS * it;
      if (i < B4 * 128) {
        float v[4] = {fmaxf(v4[it].x, 0.f), fmaxf(v4[it].y, 0.f), fmaxf(v4[it].z, 0.f), fmaxf(v4[it].w, 0.f)};
        if (dodrop) {
          const int b = sb + (i >> 7), m = (4 * i) & 511;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= drop_mult(a.drop, SITE_HEAD0 + (uint32_t)(m >> 7), (uint32_t)(b * 128 + (m & 127) + e));
        }
        reinterpret_cast<float4*>(R2)[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (tid < TF_MAXB * TF_MAXW) dpre[tid] = 0.f;            // (rows past B stay zero)
  __syncthreads();
  tstamp(a.stamps, 9);
  // ---- head output layers (every block; operands in LDS), loss.  Four lanes share an output: lane kq of the quad takes
  // hidden units 16 it + 4 kq .. + 3
#pragma unroll 1
  for (int p = tid >> 2; p < B * Wd; p += TF_THREADS / 4) {
    const int kq = tid & 3;
    const int b = p / Wd, o = p - b * Wd;
    const int x = o < C ? 0 : (o < 2 * C ? 1 : (o == 2 * C ? 2 : 3));
    const float* wr = w3s + o * 129 + 4 * kq;
    const float* hr = R2 + b * 512 + x * 128 + 4 * kq;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const float4 h4 = *reinterpret_cast<const float4*>(hr + 16 * it);
      acc0 = fmaf(wr[16 * it], h4.x, acc0); acc1 = fmaf(wr[16 * it + 1], h4.y, acc1);
      acc2 = fmaf(wr[16 * it + 2], h4.z, acc2); acc3 = fmaf(wr[16 * it + 3], h4.w, acc3);
    }
    float acc = (acc0 + acc1) + (acc2 + acc3);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += w3s[o * 129 + 128];
    if (x == 3) acc = 1.0f / (1.0f + __expf(-acc));
    if (kq == 0) {
      outsS[b * TF_MAXW + o] = acc;
      if (g == 0) gouts[p] = s_timed_out ? __uint_as_float(0x7FC00000u) : acc;
    }
  }
  __syncthreads();
  tstamp(a.stamps, 10);
  if (!a.mode) { tstamp(a.stamps, 20); return; }
  if (tid < B) {
    float t4[4]; int pr = 0;
    loss_sample<true>(outsS + tid * TF_MAXW, (int)gy[tid], ge[tid], gs[tid], C, t4, nullptr, dpre + tid * TF_MAXW, &pr);
    if (g == 0) {
      if (s_timed_out) t4[0] = t4[1] = t4[2] = t4[3] = __uint_as_float(0x7FC00000u);
      gterms[4 * tid] = t4[0]; gterms[4 * tid + 1] = t4[1]; gterms[4 * tid + 2] = t4[2]; gterms[4 * tid + 3] = t4[3];
      if (gpred) gpred[tid] = pr;
    }
  }
  __syncthreads();
  tstamp(a.stamps, 11);
  // ---- backward.  Output-layer gradients: block g writes units [8g, 8g+8), block 0 the biases (they need the hidden activations,
  // which the next loop overwrites with their gradient)
  if (tid < hnout * 8) {
    const int o = tid >> 3, mi = tid & 7;
    float acc = oW3h;
#pragma unroll 1
    for (int bb = 0; bb < B; bb += 4)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = fmaf(dpre[(bb + j) * TF_MAXW + hcoff + o], R2[(bb + j) * 512 + 8 * g + mi], acc);
    put(dW3h, acc);
  }
  if (g == 0 && tid >= 128 && tid < 128 + Wd) {
    const int o = tid - 128;
    int xx, oo; head_of(o, xx, oo);
    float acc = ob3h;
#pragma unroll 1
    for (int b = 0; b < B; ++b) acc += dpre[b * TF_MAXW + o];
    put(gheadB3(xx) + oo, acc);
  }
  __syncthreads();
  tstamp(a.stamps, 12);
  // d hidden (every block, all 512 units), in place: thread t owns units t, t + 256 of its quarter's samples
  if (active) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      const int m = t + 256 * mm, x = m >> 7, ml = m & 127;
      const int nout = x < 2 ? C : 1, coff = x == 0 ? 0 : (x == 1 ? C : (x == 2 ? 2 * C : 2 * C + 1));
      float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int o = 0; o < nout; ++o) {
        const float wv3 = w3s[(coff + o) * 129 + ml];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = fmaf(dpre[(b0 + j) * TF_MAXW + coff + o], wv3, d[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float hcur = R2[(b0 + j) * 512 + m];
        const float dv = hcur > 0.f ? d[j] * dscale : 0.f;
        R2[(b0 + j) * 512 + m] = dv;
        if (wg && g == 0 && b0 + j < B) gdhid_out[(size_t)(b0 + j) * 512 + m] = dv;
      }
    }
  }
  __syncthreads();
  tstamp(a.stamps, 13);
  // ---- d fused for this block's 4 columns (wave wv: column wv): sum over the 512 hidden units; the hidden-layer weight
  // gradients of those 4 columns (2 per quarter pair); hidden-layer bias gradients of units [8g, 8g+8)
  if (active) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float wa = wA[(lane + 64 * i) * 4 + wv];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(R2[(b0 + j) * 512 + lane + 64 * i], wa, acc[j]);
    }
    wave_sum4(acc);
    if (lane == 63) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dfusedS[(b0 + j) * 4 + wv] = acc[j];
        if (wg && b0 + j < B) gdfused_out[(size_t)(b0 + j) * 256 + 4 * g + wv] = acc[j];
      }
    }
  }
  if (!wg) {
    float2 gacc = oW0;
#pragma unroll 1
    for (int bb = 0; bb < B; bb += 4)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = R2[(bb + j) * 512 + qm];
        const float2 f = reinterpret_cast<const float2*>(fusedS)[(bb + j) * 2 + qh];
        gacc.x = fmaf(d, f.x, gacc.x); gacc.y = fmaf(d, f.y, gacc.y);
      }
    put2(dstW0, gacc);
    if (tid < 8) {
      const int m = 8 * g + tid;
      float s = ob0h;
#pragma unroll 1
      for (int b = 0; b < B; ++b) s += R2[b * 512 + m];
      put(gheadB0(m >> 7) + (m & 127), s);
    }
  }
  __syncthreads();
  tstamp(a.stamps, 14);
  // ---- partial sums of d F1 (fusion layer 3 by its output rows 4g..4g+3); its weight and bias gradients
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 df = reinterpret_cast<const float4*>(dfusedS)[b0 + j];
      if (b0 + j < B) atomicAdd(gdF1sum + (b0 + j) * 256 + t, df.x * WfL[t] + df.y * WfL[256 + t] + df.z * WfL[512 + t] + df.w * WfL[768 + t]);
    }
  }
  if (!wg) {
    float gw = ofu3;
#pragma unroll 1
    for (int bb = 0; bb < B; bb += 4)
#pragma unroll
      for (int j = 0; j < 4; ++j) gw = fmaf(dfusedS[(bb + j) * 4 + q], F1[(bb + j) * 256 + t], gw);
    put(dstfu3, gw);
    if (tid < 4) {
      float s = obfu3;
#pragma unroll 1
      for (int b = 0; b < B; ++b) s += dfusedS[b * 4 + tid];
      put(a.gbfu3 + 4 * g + tid, s);
    }
  }
  tstamp(a.stamps, 15);
  tail_arrive_wait(cnt + 2, gaveup, &s_timed_out);
  tstamp(a.stamps, 16);
  {
    const float4 v4 = tid < B * 64 ? reinterpret_cast<const float4*>(gdF1sum)[tid] : z4;
    if (tid < B4 * 64) {
      const float4 f = reinterpret_cast<const float4*>(F1)[tid];
      const float4 dfv = make_float4(f.x > 0.f ? v4.x * dscale : 0.f, f.y > 0.f ? v4.y * dscale : 0.f,
                                     f.z > 0.f ? v4.z * dscale : 0.f, f.w > 0.f ? v4.w * dscale : 0.f);
      reinterpret_cast<float4*>(DF)[tid] = dfv;
      if (wg && g == 0 && tid < B * 64) reinterpret_cast<float4*>(gdF1_out)[tid] = dfv;
    }
  }
  __syncthreads();
  tstamp(a.stamps, 17);
  // ---- d comb for this block's 8 columns (wave wv: columns 2wv, 2wv+1); fusion layer 0's weight gradients of those columns,
  // its bias gradients of units [4g, 4g+4)
  if (active) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = 2 * wv + cc;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float wb = wB[(lane + 64 * i) * 8 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(DF[(b0 + j) * 256 + lane + 64 * i], wb, acc[j]);
      }
      wave_sum4(acc);
      if (lane == 63) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dcombS[(b0 + j) * 8 + c] = acc[j];
          if (b0 + j < B) gdcomb[(b0 + j) * 512 + ccol + c] = acc[j];
        }
      }
    }
  }
  tstamp(a.stamps, 18);
  if (!wg) {
    float2 gw = ofu0;
#pragma unroll 1
    for (int bb = 0; bb < B; bb += 4)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = DF[(bb + j) * 256 + t];
        const float2 cv = reinterpret_cast<const float2*>(combS)[(bb + j) * 4 + q];
        gw.x = fmaf(d, cv.x, gw.x); gw.y = fmaf(d, cv.y, gw.y);
      }
    put2(dstfu0, gw);
    if (tid < 4) {
      float s = obfu0;
#pragma unroll 1
      for (int b = 0; b < B; ++b) s += DF[b * 256 + 4 * g + tid];
      put(a.gbfu0 + 4 * g + tid, s);
    }
  }
  __syncthreads();
  tstamp(a.stamps, 19);
  // ---- partial sums of d(mean H) of this block's stream (complete when the kernel ends); the pooled FFN layer's weight gradients
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 da = reinterpret_cast<const float4*>(dcombS)[(b0 + j) * 2], db = reinterpret_cast<const float4*>(dcombS)[(b0 + j) * 2 + 1];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const float* wc = W3L + t + 256 * kk;
        const float acc = da.x * wc[0] + da.y * wc[512] + da.z * wc[1024] + da.w * wc[1536] +
                          db.x * wc[2048] + db.y * wc[2560] + db.z * wc[3072] + db.w * wc[3584];
        if (b0 + j < B) atomicAdd(dHm + (b0 + j) * 512 + t + 256 * kk, acc);
      }
    }
  }
  if (!wg) {
    float gw[4] = {ogw3[0], ogw3[1], ogw3[2], ogw3[3]};
#pragma unroll 1
    for (int bb = 0; bb < B; bb += 4)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 dc = reinterpret_cast<const float4*>(dcombS)[(bb + j) * 2 + qh];
        const float hm = R1[(bb + j) * 512 + qm];
        gw[0] = fmaf(dc.x, hm, gw[0]); gw[1] = fmaf(dc.y, hm, gw[1]); gw[2] = fmaf(dc.z, hm, gw[2]); gw[3] = fmaf(dc.w, hm, gw[3]);
      }
#pragma unroll
    for (int c = 0; c < 4; ++c) put(gW3s + (size_t)(c0 + 4 * qh + c) * 512 + qm, gw[c]);
  }
  if (!wg && tid < 8) {
    float s = ob3s;
#pragma unroll 1
    for (int b = 0; b < B; ++b) s += dcombS[b * 8 + tid];
    put(gb3s + c0 + tid, s);
  }
  tstamp(a.stamps, 20);
}

// ---------------------------------------------------------------- optimizer
constexpr int SUMSQ_BLOCKS = 256;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = g4[i];
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc = fmaf(g[i], g[i], acc);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = (red[0] + red[1]) + (red[2] + red[3]);
    // a one-launch tail whose wait gave up earlier on this stream (tail_arrive_wait): the step's norm becomes NaN, the optimizer skips it
    if (blockIdx.x == 0 && atomicExch(&g_tail_poison, 0u) != 0u) tot = __uint_as_float(0x7FC00000u);
    out[1 + blockIdx.x] = tot;
  }
}

// One element of clip + AdamW (torch semantics: decoupled decay first, denom = sqrt(v)/sqrt(bc2) + eps, p -= lr/bc1 * m/denom).
// Shared by both optimizer kernels and compiled WITHOUT floating-point contraction: the two kernels must produce bit-identical
// parameters (a test holds them to it), and whether hipcc fuses a multiply-add depends on the code around it.
__device__ __forceinline__ void adamw_update(float& pi, float& gi, float& mi, float& vi, bool skip_step, float coef, float decay, float b1, float omb1,
                                             float b2, float omb2, float step, float inv_sqrt_bc2, float eps, int zero_grads) {
#pragma clang fp contract(off)
  if (skip_step) { if (zero_grads) gi = 0.f; return; }
  gi *= coef;
  pi *= decay;
  mi = mi * b1 + gi * omb1;
  vi = vi * b2 + gi * gi * omb2;
  pi -= step * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  if (zero_grads) gi = 0.f;
}

// clip_grad_norm_(max_norm) then AdamW (torch semantics: decoupled decay first,
// denom = sqrt(v)/sqrt(bc2) + eps, p -= lr/bc1 * m/denom)
__global__ __launch_bounds__(256) void clip_adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, size_t n,
                                                         float* __restrict__ sumsq, float max_norm, float lr,
                                                         float b1, float b2, float eps, float wd,
                                                         float inv_bc1, float inv_sqrt_bc2, int zero_grads) {
  __shared__ float red[4];
  {
    const float part = wave_sum(sumsq[1 + threadIdx.x]);      // SUMSQ_BLOCKS == blockDim.x partials
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
  }
  const float total = (red[0] + red[1]) + (red[2] + red[3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) sumsq[0] = total;
  const float coef = fminf(1.0f, max_norm / (sqrtf(total) + 1e-6f));
  // A gradient norm that is not finite (a tail-kernel wait that gave up poisons it: tail_arrive_wait) makes this a no-op step:
  // parameters and moments stay, the gradients are cleared as usual.  (torch would write NaN into every parameter.)
  const bool skip_step = !(total < __builtin_huge_valf());
  const float decay = 1.0f - lr * wd, step = lr * inv_bc1, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const size_t n4 = al ? n / 4 : 0;
  float4* p4 = reinterpret_cast<float4*>(p); float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float& pi, float& gi, float& mi, float& vi) {
    adamw_update(pi, gi, mi, vi, skip_step, coef, decay, b1, omb1, b2, omb2, step, inv_sqrt_bc2, eps, zero_grads);
  };
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
    p4[i] = pp; g4[i] = gg; m4[i] = mm; v4[i] = vv;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) upd(p[i], g[i], m[i], v[i]);
}

// ---------------------------------------------------------------- prep jobs of the bf16 schedule
__global__ __launch_bounds__(256) void prep_kernel(const PrepBatch pb) {
  __shared__ float tile[64][65];
  int ji = 0;
  for (int i = 1; i < pb.n; ++i)
    if ((int)blockIdx.x >= pb.j[i].blk_begin) ji = i;
  const PrepJob& J = pb.j[ji];
  const int blk = blockIdx.x - J.blk_begin, tid = threadIdx.x;
  if (J.type == PREP_ZERO) {                                  // 16 KB per block
    uint4* d = reinterpret_cast<uint4*>(J.dst);
    const size_t n16 = J.n >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t k = (size_t)blk * 1024 + i * 256 + tid;
      if (k < n16) d[k] = make_uint4(0u, 0u, 0u, 0u);
    }
  } else if (J.type == PREP_CAST) {                           // 4096 elements per block
    const float4* s = reinterpret_cast<const float4*>(J.src);
    uint2* d = reinterpret_cast<uint2*>(J.dst);
    const size_t n4 = J.n >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t k = (size_t)blk * 1024 + i * 256 + tid;
      if (k < n4) { const float4 v = s[k]; d[k] = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w)); }
    }
  } else {                                                    // 64 x 64 tile through LDS
    const int tc = (J.cols + 63) >> 6;
    const int r0 = (blk / tc) * 64, c0 = (blk % tc) * 64;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {                            // all 16 loads in flight before the first LDS store
      const int i = tid + 256 * k, r = i >> 6, c = i & 63;
      v[k] = J.src[(size_t)min(r0 + r, J.rows - 1) * J.cols + min(c0 + c, J.cols - 1)];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int i = tid + 256 * k; tile[i >> 6][i & 63] = v[k]; }
    __syncthreads();
    unsigned short* d = reinterpret_cast<unsigned short*>(J.dst);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = tid + 256 * k, c = i >> 6, r = i & 63;
      if (r0 + r < J.rows && c0 + c < J.cols) d[(size_t)(c0 + c) * J.ld_dst + J.col_off + r0 + r] = f2bf(tile[r][c]);
    }
  }
}

// ---------------------------------------------------------------- parameter-space weight gradients (misc.h, UnfoldStream)
// Plain fp32 FMA on 64 x 64 output tiles (256 threads, 4 x 4 outputs each, 16-deep LDS tiles): 200 MFLOP in all -- the point is a
// short launch (every block walks 8 tiles of 16), not throughput.  Blocks: [0, 96) dW_in tiles (2 streams x 12 x 4),
// [96, 192) dW_p tiles (2 streams x 4 x 2 tiles x 6 slices of 128 rows, merged by atomics), [192, 204) db_p (2 x 6 slices).
constexpr int UF_H = 256, UF_D = 128;
__global__ __launch_bounds__(256) void unfold_kernel(const UnfoldStream s0, const UnfoldStream s1) {
  __shared__ __attribute__((aligned(16))) float As[16][68];
  __shared__ __attribute__((aligned(16))) float Bs[16][68];
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
  int blk = blockIdx.x;
  if (blk >= 192) {                                          // db_p[c] += sum_i W_in[i][c] db[i]: 6 slices of 128 rows per stream, 4 row phases x 64 float4 columns per block
    blk -= 192;
    const UnfoldStream& S = blk < 6 ? s0 : s1;
    const int r0 = 128 * (blk < 6 ? blk : blk - 6), ph = t >> 6, c4 = 4 * (t & 63);
    const float* W = r0 < UF_H ? S.Wq + (size_t)r0 * UF_H : S.Wkv + (size_t)(r0 - UF_H) * UF_H;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const int i = ph + 4 * j;
      const float4 wv = *reinterpret_cast<const float4*>(W + (size_t)i * UF_H + c4);
      const float d = S.db[r0 + i];
      acc.x = fmaf(wv.x, d, acc.x); acc.y = fmaf(wv.y, d, acc.y); acc.z = fmaf(wv.z, d, acc.z); acc.w = fmaf(wv.w, d, acc.w);
    }
    atomicAdd(S.Gbp + c4, acc.x); atomicAdd(S.Gbp + c4 + 1, acc.y); atomicAdd(S.Gbp + c4 + 2, acc.z); atomicAdd(S.Gbp + c4 + 3, acc.w);
    return;
  }
  const bool win = blk < 96;
  if (!win) blk -= 96;
  const UnfoldStream& S = blk < 48 ? s0 : s1;
  if (blk >= 48) blk -= 48;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  auto mac = [&]() {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&As[k][4 * ty]), b = *reinterpret_cast<const float4*>(&Bs[k][4 * tx]);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
  };
  if (win) {
    // C[i][c] over i in [64 ti, +64), c in [64 tc, +64): A = M rows (k contiguous), B = W_p rows (k contiguous): transposed into LDS
    const int ti = blk >> 2, tc = blk & 3, i0 = 64 * ti, c0 = 64 * tc;
    const int r = t >> 2, k4 = 4 * (t & 3);
    float4 av8[8], bv8[8];                                     // every operand of the block in flight at once: ONE memory round trip
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      av8[q] = *reinterpret_cast<const float4*>(S.M + (size_t)(i0 + r) * UF_D + 16 * q + k4);
      bv8[q] = *reinterpret_cast<const float4*>(S.Wp + (size_t)(c0 + r) * UF_D + 16 * q + k4);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a = av8[q], b = bv8[q];
      __syncthreads();
      As[k4][r] = a.x; As[k4 + 1][r] = a.y; As[k4 + 2][r] = a.z; As[k4 + 3][r] = a.w;
      Bs[k4][r] = b.x; Bs[k4 + 1][r] = b.y; Bs[k4 + 2][r] = b.z; Bs[k4 + 3][r] = b.w;
      __syncthreads();
      mac();
    }
    float* G = i0 < UF_H ? S.Gq + (size_t)i0 * UF_H : S.Gkv + (size_t)(i0 - UF_H) * UF_H;
    const float4 bp = *reinterpret_cast<const float4*>(S.bp + c0 + 4 * tx);
    const float bpv[4] = {bp.x, bp.y, bp.z, bp.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {                            // (one writer per element, and nothing else of the step adds to these gradients: plain +=)
      const float dbi = S.db[i0 + 4 * ty + i];
      float4* g4 = reinterpret_cast<float4*>(G + (size_t)(4 * ty + i) * UF_H + c0 + 4 * tx);
      float4 o = *g4;
      o.x += fmaf(dbi, bpv[0], acc[i][0]); o.y += fmaf(dbi, bpv[1], acc[i][1]); o.z += fmaf(dbi, bpv[2], acc[i][2]); o.w += fmaf(dbi, bpv[3], acc[i][3]);
      *g4 = o;
    }
  } else {
    // C[c][k] over c in [64 tc, +64), k in [64 tk, +64), rows [128 sl, +128) of W_in / M (both row-contiguous in the tile's direction)
    const int sl = blk / 8, tc = (blk & 7) >> 1, tk = blk & 1, c0 = 64 * tc, kk0 = 64 * tk, r0 = 128 * sl;
    const float* W = r0 < UF_H ? S.Wq + (size_t)r0 * UF_H : S.Wkv + (size_t)(r0 - UF_H) * UF_H;
    const float* Mr = S.M + (size_t)r0 * UF_D;
    const int r = t >> 4, c4 = 4 * (t & 15);
    float4 av8[8], bv8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      av8[q] = *reinterpret_cast<const float4*>(W + (size_t)(16 * q + r) * UF_H + c0 + c4);
      bv8[q] = *reinterpret_cast<const float4*>(Mr + (size_t)(16 * q + r) * UF_D + kk0 + c4);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a = av8[q], b = bv8[q];
      __syncthreads();
      *reinterpret_cast<float4*>(&As[r][c4]) = a;
      *reinterpret_cast<float4*>(&Bs[r][c4]) = b;
      __syncthreads();
      mac();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(S.Gp + (size_t)(c0 + 4 * ty + i) * UF_D + kk0 + 4 * tx + j, acc[i][j]);
  }
}

}  // namespace

// ------------------------------------------------------------------ launchers
int launch_unfold(const UnfoldStream& rg, const UnfoldStream& kg, hipStream_t stream) {
  const int prof = gemm_prof_open(stream, 2.0 * 2.0 * 2.0 * 768.0 * 256.0 * 128.0, PROF_GEMM);
  hipLaunchKernelGGL(unfold_kernel, dim3(204), dim3(256), 0, stream, rg, kg);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int launch_gather_batch(const float* rg_all, const long long* sample_off, const float* kg_all, const long long* y_all, const float* e_all, const float* s_all,
                        const long long* idx, int B, int T, int D, int KG, float* rg_out, float* kg_out, int* off_out, long long* y_out, float* e_out,
                        float* s_out, float noise_std, unsigned long long seed, hipStream_t stream) {
  if (B < 1 || B > GATHER_MAXB || T < B || D < 4 || (D & 3) || D > 1024 || KG < 2 || (KG & 1)) return (int)hipErrorInvalidValue;
  const int row_blocks = (T + GATHER_ROWS - 1) / GATHER_ROWS;
  hipLaunchKernelGGL(gather_batch_kernel, dim3(row_blocks + B), dim3(256), 0, stream, rg_all, sample_off, kg_all, y_all, e_all, s_all, idx, B, T, D, KG, row_blocks,
                     rg_out, kg_out, off_out, y_out, e_out, s_out, noise_std, (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32));
  return (int)hipGetLastError();
}
int launch_prep(PrepBatch& pb, hipStream_t stream) {
  int total = 0;
  for (int i = 0; i < pb.n; ++i) {
    PrepJob& J = pb.j[i];
    J.blk_begin = total;
    if (J.type == PREP_ZERO) total += (int)(((J.n >> 4) + 1023) / 1024);
    else if (J.type == PREP_CAST) total += (int)(((J.n >> 2) + 1023) / 1024);
    else total += ((J.rows + 63) / 64) * ((J.cols + 63) / 64);
  }
  if (total == 0) return 0;
  hipLaunchKernelGGL(prep_kernel, dim3(total), dim3(256), 0, stream, pb);
  return (int)hipGetLastError();
}

int launch_rowmap(const int* offs, int* row_sample, float* inv_nr, int* tile_off, int4* tile_desc, int B, int ntile_max, int max_nr, hipStream_t stream) {
  (void)max_nr;
  // ONE launch (a fresh descriptor per training step when every minibatch has its own Nr tuple): every block scans the B tile
  // counts itself, then writes its 256 rows' sample ids and its share of the tile table
  const int T_upper = 32 * ntile_max;                                   // >= T (ntile_max = T / 32 + B)
  const int blocks = (T_upper + 255) / 256;
  hipLaunchKernelGGL(batchdesc_kernel, dim3(blocks), dim3(256), 0, stream, offs, row_sample, inv_nr, tile_off, tile_desc, B, ntile_max);
  return (int)hipGetLastError();
}

int ln_supported(int H) { return H >= 1 && H <= 1024; }

int launch_ln_fwd(const LnSeg& s0, const LnSeg& s1, int H, hipStream_t stream) {
  const int rpb = LNF_W * LNF_R;
  const int nb0 = (s0.rows + rpb - 1) / rpb, nb1 = (s1.rows + rpb - 1) / rpb;
  if (nb0 + nb1 == 0) return 0;
  if (H <= 256) hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(nb0 + nb1), dim3(64 * LNF_W), 0, stream, s0, s1, H, nb0);
  else          hipLaunchKernelGGL(ln_fwd_kernel<16>, dim3(nb0 + nb1), dim3(64 * LNF_W), 0, stream, s0, s1, H, nb0);
  return (int)hipGetLastError();
}

int launch_ln_bwd(const LnBwdSeg& s0_, const LnBwdSeg& s1_, int H, hipStream_t stream) {
  constexpr int rpb = 32;
  const LnBwdSeg& s0 = s0_; const LnBwdSeg& s1 = s1_;
  auto nblk = [](int rows) { int b = (rows + rpb - 1) / rpb; return rows == 0 ? 0 : (b > 2048 ? 2048 : (b < 1 ? 1 : b)); };
  const int nb0 = nblk(s0.rows), nb1 = nblk(s1.rows);
  if (nb0 + nb1 == 0) return 0;
  const bool vec = (H & 3) == 0;
  if (H <= 256) {
    if (vec) hipLaunchKernelGGL((ln_bwd_kernel<4, true>), dim3(nb0 + nb1), dim3(64 * LNB_W), 0, stream, s0, s1, H, nb0);
    else     hipLaunchKernelGGL((ln_bwd_kernel<4, false>), dim3(nb0 + nb1), dim3(64 * LNB_W), 0, stream, s0, s1, H, nb0);
  } else {
    if (vec) hipLaunchKernelGGL((ln_bwd_kernel<16, true>), dim3(nb0 + nb1), dim3(64 * LNB_W), 0, stream, s0, s1, H, nb0);
    else     hipLaunchKernelGGL((ln_bwd_kernel<16, false>), dim3(nb0 + nb1), dim3(64 * LNB_W), 0, stream, s0, s1, H, nb0);
  }
  return (int)hipGetLastError();
}

int launch_seg_mean(const SegMean* segs, int nseg, int B, int max_rows, hipStream_t stream) {
  SegMean z{}; SegMean s[4] = {z, z, z, z};
  for (int i = 0; i < nseg && i < 4; ++i) s[i] = segs[i];
  const int rpb = 32;
  bool vec = true;
  for (int i = 0; i < nseg && i < 4; ++i)
    vec = vec && (s[i].C & 3) == 0 && (s[i].ld & 3) == 0 && (reinterpret_cast<uintptr_t>(s[i].X) & 15) == 0;
  const dim3 grid((max_rows + rpb - 1) / rpb, B, nseg);
  if (vec) hipLaunchKernelGGL(seg_mean_kernel<true>, grid, dim3(256), 0, stream, s[0], s[1], s[2], s[3], rpb);
  else     hipLaunchKernelGGL(seg_mean_kernel<false>, grid, dim3(256), 0, stream, s[0], s[1], s[2], s[3], rpb);
  return (int)hipGetLastError();
}

int launch_relu_bcast_bwd(const BcastSeg& s0, const BcastSeg& s1, int C, float scale, hipStream_t stream) {
  const long n0 = (long)s0.rows * C, n1 = (long)s1.rows * C;
  if (n0 + n1 == 0) return 0;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = (C & 3) == 0 && (s0.ldv & 3) == 0 && (s1.ldv & 3) == 0 && al16(s0.act) && al16(s0.v) && al16(s0.dst) &&
                   al16(s1.act) && al16(s1.v) && al16(s1.dst) && al16(s0.dst16) && al16(s1.dst16) && al16(s0.act16) && al16(s1.act16);
  if (vec) hipLaunchKernelGGL(relu_bcast_bwd_kernel<4>, dim3((unsigned)(((n0 + n1) / 4 + 255) / 256)), dim3(256), 0, stream, s0, s1, C, n0, scale);
  else     hipLaunchKernelGGL(relu_bcast_bwd_kernel<1>, dim3((unsigned)((n0 + n1 + 255) / 256)), dim3(256), 0, stream, s0, s1, C, n0, scale);
  return (int)hipGetLastError();
}

int launch_head_out_grad(const float* outs, const float* d_outs, float* d_logits, int B, int W, hipStream_t stream) {
  hipLaunchKernelGGL(head_out_grad_kernel, dim3((B * W + 255) / 256), dim3(256), 0, stream, outs, d_outs, d_logits, B, W);
  return (int)hipGetLastError();
}

int launch_loss(const float* outs, const long long* y, const float* e, const float* s, int B, int C,
                float* terms, float* d_outs, float* d_pre, int* pred, hipStream_t stream) {
  hipLaunchKernelGGL(loss_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, outs, y, e, s, B, C, terms, d_outs, d_pre, pred);
  return (int)hipGetLastError();
}

int heads_loss_ok(int B, int C) { return B >= 1 && C >= 1 && 2 * C + 2 <= HEADS_MAXW; }
int launch_heads_loss(const float* hid, const HeadsOut& hp, const long long* y, const float* e, const float* s, int B, int C,
                      int Fh, float scale, float* outs, float* terms, int* pred, float* dhid, hipStream_t stream, float* dlog) {
  if (!heads_loss_ok(B, C)) return (int)hipErrorInvalidValue;
  const int prof = gemm_prof_open(stream, 0.0, PROF_TAIL);
  // samples per block: 1 up to 64 samples (every block's latency counts), then 2 / 4 / 8 so that the grid stays at 64-128 blocks
  const int S = B <= 64 ? 1 : (B <= 256 ? 2 : (B <= 512 ? 4 : 8));
  const dim3 grid((B + S - 1) / S);
  if (C <= 8) hipLaunchKernelGGL(heads_loss_kernel<true>, grid, dim3(256), 0, stream, hid, hp, y, e, s, B, S, C, Fh, scale, outs, terms, pred, dhid, dlog);
  else        hipLaunchKernelGGL(heads_loss_kernel<false>, grid, dim3(256), 0, stream, hid, hp, y, e, s, B, S, C, Fh, scale, outs, terms, pred, dhid, dlog);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

// developer hook (camo_debug_set_option "tail_skip_arrival" = block id + 1): that block of the NEXT one-launch tail skips its first
// arrival, so the others' wait times out deterministically -- the only way to test the give-up path without sharing the GPU
thread_local int g_tail_debug_skip = 0;

int tail_fused_ok(int B, int C) {
  // every block of the launch waits for the other 63: all of them must be resident at once, one per CU (1024 threads, 148 KB of
  // LDS), so the device (or the partition this process sees) must have at least that many CUs
  static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return B >= 1 && B <= TF_MAXGROUPS * TF_MAXB && C >= 1 && 2 * C + 2 <= TF_MAXW && cus >= TG * ((B + TF_MAXB - 1) / TF_MAXB);
}
int launch_tail_fused(const TailFusedArgs& a, hipStream_t stream) {
  if (!tail_fused_ok(a.B, a.C)) return (int)hipErrorInvalidValue;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS_FLOATS * 4);
    return true;
  }();
  (void)attr;
  TailFusedArgs aa = a;
  aa.debug_skip = g_tail_debug_skip; g_tail_debug_skip = 0;
  const int prof = gemm_prof_open(stream, 0.0, PROF_TAIL);
  hipLaunchKernelGGL(tail_fused_kernel, dim3(TG * ((a.B + TF_MAXB - 1) / TF_MAXB)), dim3(TF_THREADS), TF_LDS_FLOATS * 4, stream, aa);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int tail_timeouts(unsigned int* out) {          // synchronous (hipMemcpyFromSymbol): health check, not part of a step
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_timeouts), sizeof(unsigned int));
}

// Data parallel: a rank whose tail gave up must take every OTHER rank's step down with it -- its garbage gradients are about to be
// summed into all of them.  One thread: if the poison flag is up, the first element of the flat gradient buffer becomes NaN; the
// SUM all-reduce carries it to every rank, every rank's norm launch then sees a NaN norm and every optimizer skips the same step.
// (The flag stays up: this rank's own norm launch consumes it as before.)  Enqueued by the trainer between the training call and the
// all-reduce of the bucket that holds element 0.
__global__ void tail_poison_to_grads_kernel(float* __restrict__ g) {
  if (__hip_atomic_load(&g_tail_poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) g[0] = __uint_as_float(0x7FC00000u);
}
int launch_tail_poison_to_grads(float* g, hipStream_t stream) {
  hipLaunchKernelGGL(tail_poison_to_grads_kernel, dim3(1), dim3(1), 0, stream, g);
  return (int)hipGetLastError();
}

int launch_sumsq(const float* g, size_t n, float* out, hipStream_t stream) {
  const int prof = gemm_prof_open(stream, 0.0, PROF_OPT);
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, g, n, out);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

// AdamW + weight shadows.  Blocks [0, tiles): one 32-row x 64-column tile of a shadowed matrix each -- thread (r = tid >> 3,
// c = tid & 7) owns the 8 consecutive elements (row r, columns 8c..8c+7): the same arithmetic as clip_adamw_kernel on two
// float4s per buffer, then (1) those 8 new values, rounded to bf16, ARE one 16-byte chunk of the plain shadow (fragment order
// [wave][k step][tile][lane][8], lane = row + 32 * (k half): fused_rows.h) -- stored straight from registers; (2) the tile goes
// to LDS as bf16 and comes back by columns: a chunk of the transposed shadow is 8 consecutive ROWS of one column.
// Blocks [tiles, ...): the rest of the flat buffer (biases, LayerNorm, per-sample tail), elementwise.
__device__ __forceinline__ size_t shadow_chunk(int N, int K, int n, int k0) {       // chunk index of (row n, columns k0..k0+7) in fragment order
  const int KS = K >> 4, NTw = N >> 7, tile32 = n >> 5;
  const int w = tile32 / NTw, t = tile32 - w * NTw;
  return (((size_t)w * KS + (k0 >> 4)) * NTw + t) * 64 + (n & 31) + 32 * ((k0 >> 3) & 1);
}

__global__ __launch_bounds__(256) void adamw_shadow_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ sumsq, float max_norm, float lr,
                                                           float b1, float b2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2,
                                                           int zero_grads, const AdamShadowArgs a) {
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) unsigned short tile[32 * 72];     // [32 rows][64 columns] bf16, pitch 72
  const int tid = threadIdx.x;
  // a tile block issues its eight 16-byte loads BEFORE the norm's partial sums are fetched and reduced: one memory round trip
  // for both instead of two in a row
  const bool is_tile = (int)blockIdx.x < a.tiles;
  int bi = 0, tr = 0, tc = 0;
  const int r = tid >> 3, c = tid & 7;
  float4 *p4 = nullptr, *g4 = nullptr, *m4 = nullptr, *v4 = nullptr;
  float4 pa, pb, ga, gb, ma, mb, va, vb;
  if (is_tile) {
#pragma unroll
    for (int i = 1; i < ADAM_SHADOW_MAXB; ++i)
      if (i < a.nblk && (int)blockIdx.x >= a.blk[i].tile_begin) bi = i;
    const AdamShadowBlock& B = a.blk[bi];
    const int tl = (int)blockIdx.x - B.tile_begin, ct = B.cols >> 6;
    tr = tl / ct; tc = tl - tr * ct;                        // tile (rows 32 tr.., columns 64 tc..)
    const size_t e = B.off + (size_t)(32 * tr + r) * B.cols + 64 * tc + 8 * c;
    p4 = reinterpret_cast<float4*>(p + e); g4 = reinterpret_cast<float4*>(g + e);
    m4 = reinterpret_cast<float4*>(m + e); v4 = reinterpret_cast<float4*>(v + e);
    pa = p4[0]; pb = p4[1]; ga = g4[0]; gb = g4[1]; ma = m4[0]; mb = m4[1]; va = v4[0]; vb = v4[1];
  }
  {
    const float part = wave_sum(sumsq[1 + tid]);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
  }
  const float total = (red[0] + red[1]) + (red[2] + red[3]);
  if (blockIdx.x == 0 && tid == 0) sumsq[0] = total;
  const float coef = fminf(1.0f, max_norm / (sqrtf(total) + 1e-6f));
  // A gradient norm that is not finite (a tail-kernel wait that gave up poisons it: tail_arrive_wait) makes this a no-op step:
  // parameters and moments stay, the gradients are cleared as usual.  (torch would write NaN into every parameter.)
  const bool skip_step = !(total < __builtin_huge_valf());
  const float decay = 1.0f - lr * wd, step = lr * inv_bc1, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  auto upd = [&](float& pi, float& gi, float& mi, float& vi) {
    adamw_update(pi, gi, mi, vi, skip_step, coef, decay, b1, omb1, b2, omb2, step, inv_sqrt_bc2, eps, zero_grads);
  };
  if (is_tile) {
    const AdamShadowBlock& B = a.blk[bi];
    const int row = 32 * tr + r, col = 64 * tc + 8 * c;
    upd(pa.x, ga.x, ma.x, va.x); upd(pa.y, ga.y, ma.y, va.y); upd(pa.z, ga.z, ma.z, va.z); upd(pa.w, ga.w, ma.w, va.w);
    upd(pb.x, gb.x, mb.x, vb.x); upd(pb.y, gb.y, mb.y, vb.y); upd(pb.z, gb.z, mb.z, vb.z); upd(pb.w, gb.w, mb.w, vb.w);
    p4[0] = pa; p4[1] = pb; g4[0] = ga; g4[1] = gb; m4[0] = ma; m4[1] = mb; v4[0] = va; v4[1] = vb;
    const uint4 ch = make_uint4(pack2(pa.x, pa.y), pack2(pa.z, pa.w), pack2(pb.x, pb.y), pack2(pb.z, pb.w));
    reinterpret_cast<uint4*>(B.plain)[shadow_chunk(B.pN, B.cols, B.pn0 + row, col)] = ch;
    if (B.trans) {                                          // (block-uniform)
      *reinterpret_cast<uint4*>(tile + r * 72 + 8 * c) = ch;
      __syncthreads();
      const int cl = tid & 63, rc = tid >> 6;               // column 64 tc + cl, rows 8 rc .. 8 rc + 7 of the tile
      unsigned short t8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t8[i] = tile[(8 * rc + i) * 72 + cl];
      const uint4 tch = make_uint4((unsigned)t8[0] | ((unsigned)t8[1] << 16), (unsigned)t8[2] | ((unsigned)t8[3] << 16),
                                   (unsigned)t8[4] | ((unsigned)t8[5] << 16), (unsigned)t8[6] | ((unsigned)t8[7] << 16));
      reinterpret_cast<uint4*>(B.trans)[shadow_chunk(B.cols, B.tK, 64 * tc + cl, B.tk0 + 32 * tr + 8 * rc)] = tch;
    }
    return;
  }
  // the rest.  The launcher sorted the ranges by length, longest first: ranges [0, nbig) are shared grid-stride by the
  // elementwise blocks, the short ones (biases, LayerNorm: a few hundred elements each) get a block each at the end --
  // eleven ranges looked at by every thread was eleven rounds of scalar loads of their bounds for mostly nothing
  const int eb = (int)blockIdx.x - a.tiles, neb = (int)gridDim.x - a.tiles;
  auto run = [&](int ri, size_t first, size_t stride) {
    const size_t n4 = a.range_len[ri] >> 2;
    float4* p4 = reinterpret_cast<float4*>(p + a.range_begin[ri]); float4* g4 = reinterpret_cast<float4*>(g + a.range_begin[ri]);
    float4* m4 = reinterpret_cast<float4*>(m + a.range_begin[ri]); float4* v4 = reinterpret_cast<float4*>(v + a.range_begin[ri]);
    for (size_t i = first; i < n4; i += stride) {
      float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      p4[i] = pp; g4[i] = gg; m4[i] = mm; v4[i] = vv;
    }
  };
  const int nsmall = a.nrange - a.nbig;                      // the last nsmall blocks take one short range each
  if (eb >= neb - nsmall) run(a.nbig + (eb - (neb - nsmall)), tid, 256);
  else for (int ri = 0; ri < a.nbig; ++ri) run(ri, (size_t)eb * 256 + tid, (size_t)(neb - nsmall) * 256);
}

int launch_clip_adamw_shadows(float* p, float* g, float* m, float* v, float* sumsq, float max_norm, float lr, float b1, float b2,
                              float eps, float wd, int step, int zero_grads, AdamShadowArgs& a, hipStream_t stream) {
  if (a.nblk < 1 || a.nblk > ADAM_SHADOW_MAXB || a.nrange < 0 || a.nrange > ADAM_SHADOW_MAXR) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
    return (int)hipErrorInvalidValue;
  int tiles = 0;
  for (int i = 0; i < a.nblk; ++i) {
    AdamShadowBlock& B = a.blk[i];
    if ((B.rows & 31) || (B.cols & 63) || (B.off & 3) || !B.plain || (B.pN & 127) || (B.pn0 & 31) || (B.trans && ((B.cols & 127) || (B.tK & 15) || (B.tk0 & 31))))
      return (int)hipErrorInvalidValue;
    B.tile_begin = tiles;
    tiles += (B.rows >> 5) * (B.cols >> 6);
  }
  for (int i = 0; i < a.nrange; ++i)
    if ((a.range_begin[i] & 3) || (a.range_len[i] & 3)) return (int)hipErrorInvalidValue;
  a.tiles = tiles;
  for (int i = 1; i < a.nrange; ++i)                         // longest first
    for (int j = i; j > 0 && a.range_len[j] > a.range_len[j - 1]; --j) {
      const size_t tb = a.range_begin[j], tl = a.range_len[j];
      a.range_begin[j] = a.range_begin[j - 1]; a.range_len[j] = a.range_len[j - 1]; a.range_begin[j - 1] = tb; a.range_len[j - 1] = tl;
    }
  a.nbig = 0;
  size_t big = 0;
  while (a.nbig < a.nrange && a.range_len[a.nbig] >= 8192) big += a.range_len[a.nbig++];
  size_t nb = (big / 4 + 255) / 256;                         // blocks for the long ranges ...
  if (nb < 1) nb = 1;
  if (nb > 1000) nb = 1000;
  nb += (size_t)(a.nrange - a.nbig);                         // ... + one per short range
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  const int prof = gemm_prof_open(stream, 0.0, PROF_OPT);
  hipLaunchKernelGGL(adamw_shadow_kernel, dim3((unsigned)(tiles + nb)), dim3(256), 0, stream, p, g, m, v, sumsq, max_norm, lr, b1, b2, eps, wd,
                     (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), zero_grads, a);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int launch_clip_adamw(float* p, float* g, float* m, float* v, size_t n, float* sumsq, float max_norm,
                      float lr, float b1, float b2, float eps, float wd, int step, int zero_grads, hipStream_t stream) {
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  size_t nb = (n / 4 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  const int prof = gemm_prof_open(stream, 0.0, PROF_OPT);
  hipLaunchKernelGGL(clip_adamw_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p, g, m, v, n, sumsq, max_norm, lr,
                     b1, b2, eps, wd, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), zero_grads);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}
