// Bandwidth-bound pieces of the fusion path (gfx950): row maps, LayerNorm fwd/bwd,
// per-sample column means, ReLU-mask broadcast backward, the four-term loss, and the
// clip + AdamW optimizer step.  One wave (64 lanes) per row wherever a row reduction is
// needed; everything streams coalesced fp32.
#include "misc.h"
#include "gemm.h"      // launch timing hooks


namespace {

// ---------------------------------------------------------------- row -> sample map
__global__ void rowmap_kernel(const int* __restrict__ offs, int* __restrict__ row_sample,
                              float* __restrict__ inv_nr) {
  const int b = blockIdx.y;
  const int r0 = offs[b], nr = offs[b + 1] - r0;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nr) row_sample[r0 + r] = b;
  if (r == 0) inv_nr[b] = 1.0f / (float)nr;
}

// first 32-row tile of every sample (fused_rows.hip tiles a sample's rows in 32s): exclusive scan of ceil(Nr / 32)
__global__ __launch_bounds__(256) void tileoff_kernel(const int* __restrict__ offs, int* __restrict__ tile_off, int B) {
  __shared__ int part[256];
  const int t = threadIdx.x, per = (B + 255) / 256, b0 = min(B, t * per), b1 = min(B, b0 + per);
  int s = 0;
  for (int b = b0; b < b1; ++b) s += (offs[b + 1] - offs[b] + 31) >> 5;
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = acc; acc += v; }
    tile_off[B] = acc;
  }
  __syncthreads();
  int acc = part[t];
  for (int b = b0; b < b1; ++b) { tile_off[b] = acc; acc += (offs[b + 1] - offs[b] + 31) >> 5; }
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
__device__ __forceinline__ void loss_sample(const float* __restrict__ o, int yb, float eb, float sb, int C,
                                            float* __restrict__ terms4, float* __restrict__ d_outs, float* __restrict__ d_pre,
                                            int* __restrict__ pred) {
  const int W = 2 * C + 2;
  const float sc = o[W - 1];
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
    for (int k = 0; k < C; ++k) z += expf(o[k] - mx);
    // ce from the log-sum-exp (finite when pt underflows, like torch's log_softmax); the gradient is written
    // without the division by pt:  d l / d logit_k = at * (-3 om^2 ce pt - om^3) * (delta_ky - p_k)
    const float ce = -(o[yb] - mx - logf(z));
    const float pt = expf(o[yb] - mx) / z;
    const float at = yb == 1 ? 0.75f : 0.25f;
    const float om = 1.0f - pt;
    terms4[0] = 3.0f * at * om * om * om * ce + poison;
    const float dl = at * (-3.0f * om * om * ce * pt - om * om * om);
    for (int k = 0; k < C; ++k) {
      const float pk = expf(o[k] - mx) / z;
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
    for (int k = 0; k < C; ++k) z += expf(oi[k] - mx);
    terms4[1] = -(oi[yb] - mx - logf(z));
    for (int k = 0; k < C; ++k) put(C + k, expf(oi[k] - mx) / z - (k == yb ? 1.0f : 0.0f));
  }
  // BCE with logits on edge
  {
    const float x = o[2 * C], t = eb;
    terms4[2] = 0.5f * (fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
    put(2 * C, 0.5f * (1.0f / (1.0f + expf(-x)) - t));
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
__global__ __launch_bounds__(256) void heads_loss_kernel(const float* __restrict__ hid, HeadsOut hp,
                                                         const long long* __restrict__ y, const float* __restrict__ e,
                                                         const float* __restrict__ s, int C, int Fh, float scale,
                                                         float* __restrict__ outs, float* __restrict__ terms,
                                                         int* __restrict__ pred, float* __restrict__ dhid) {
  __shared__ float so[HEADS_MAXW], sd[HEADS_MAXW];
  const int b = blockIdx.x, x = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int W = 2 * C + 2, nout = x < 2 ? C : 1, coff = x == 0 ? 0 : (x == 1 ? C : (x == 2 ? 2 * C : 2 * C + 1));
  // (static indices + selects: indexing the by-value argument struct with the runtime wave id would move it to scratch)
  const float* Wx = x == 0 ? hp.W[0] : (x == 1 ? hp.W[1] : (x == 2 ? hp.W[2] : hp.W[3]));
  const float* bx = x == 0 ? hp.b[0] : (x == 1 ? hp.b[1] : (x == 2 ? hp.b[2] : hp.b[3]));
  float* gWx = x == 0 ? hp.gW[0] : (x == 1 ? hp.gW[1] : (x == 2 ? hp.gW[2] : hp.gW[3]));
  float* gbx = x == 0 ? hp.gb[0] : (x == 1 ? hp.gb[1] : (x == 2 ? hp.gb[2] : hp.gb[3]));
  const float* h = hid + (size_t)b * 4 * Fh + x * Fh;
  const float bias0 = bx[0];                                  // (in flight with the hidden row)
  // labels first: their load latency hides under the dot products instead of following the barrier
  const int yb = (int)y[b]; const float eb = e[b], sb = s[b];
  constexpr int HC = 4;                                       // hidden values cached per lane (Fh <= 256), else re-read
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
  auto back = [&](int c, float hval) {
    float dh = 0.f;
    for (int o = 0; o < nout; ++o) {
      const float d = sd[coff + o];
      dh = fmaf(d, Wx[(size_t)o * Fh + c], dh);
      atomicAdd(gWx + (size_t)o * Fh + c, d * hval);
    }
    dhid[(size_t)b * 4 * Fh + x * Fh + c] = hval > 0.f ? dh * scale : 0.f;
  };
#pragma unroll
  for (int i = 0; i < HC; ++i) { const int c = lane + 64 * i; if (c < Fh) back(c, hv[i]); }     // (static register indices)
  for (int c = lane + 64 * HC; c < Fh; c += 64) back(c, h[c]);
  if (lane < nout) atomicAdd(gbx + lane, sd[coff + lane]);
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
  if (threadIdx.x == 0) out[1 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
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
  const float decay = 1.0f - lr * wd, step = lr * inv_bc1, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const size_t n4 = al ? n / 4 : 0;
  float4* p4 = reinterpret_cast<float4*>(p); float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float& pi, float& gi, float& mi, float& vi) {
    gi *= coef;
    pi *= decay;
    mi = mi * b1 + gi * omb1;
    vi = vi * b2 + gi * gi * omb2;
    pi -= step * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    if (zero_grads) gi = 0.f;
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

}  // namespace

// ------------------------------------------------------------------ launchers
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

int launch_rowmap(const int* offs, int* row_sample, float* inv_nr, int* tile_off, int B, int max_nr, hipStream_t stream) {
  hipLaunchKernelGGL(rowmap_kernel, dim3((max_nr + 255) / 256, B), dim3(256), 0, stream, offs, row_sample, inv_nr);
  hipLaunchKernelGGL(tileoff_kernel, dim3(1), dim3(256), 0, stream, offs, tile_off, B);
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
                      int Fh, float scale, float* outs, float* terms, int* pred, float* dhid, hipStream_t stream) {
  if (!heads_loss_ok(B, C)) return (int)hipErrorInvalidValue;
  const int prof = gemm_prof_open(stream, 0.0, PROF_TAIL);
  hipLaunchKernelGGL(heads_loss_kernel, dim3(B), dim3(256), 0, stream, hid, hp, y, e, s, C, Fh, scale, outs, terms, pred, dhid);
  gemm_prof_close(prof, stream);
  return (int)hipGetLastError();
}

int launch_sumsq(const float* g, size_t n, float* out, hipStream_t stream) {
  const int prof = gemm_prof_open(stream, 0.0, PROF_OPT);
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, g, n, out);
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
