// Bandwidth-bound pieces of the fusion path (gfx950): row maps, LayerNorm fwd/bwd,
// per-sample column means, ReLU-mask broadcast backward, the four-term loss, and the
// clip + AdamW optimizer step.  One wave (64 lanes) per row wherever a row reduction is
// needed; everything streams coalesced fp32.
#include "misc.h"

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

// ---------------------------------------------------------------- LayerNorm forward
// y = (u - mean) * rstd * gamma + beta, eps 1e-5 (nn.LayerNorm default, fusion_model.py:49-50)
template <int HPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnSeg s0, LnSeg s1, int H, int nb0) {
  const bool first = (int)blockIdx.x < nb0;
  const LnSeg& S = first ? s0 : s1;
  const int row = ((first ? blockIdx.x : blockIdx.x - nb0) * 4) + (threadIdx.x >> 6);
  if (row >= S.rows) return;
  const int lane = threadIdx.x & 63;
  const float* u = S.U + (size_t)row * H;
  float x[HPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < HPL; ++i) {
    const int c = lane + 64 * i;
    x[i] = c < H ? u[c] : 0.f;
    sum += x[i];
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < HPL; ++i) {
    const int c = lane + 64 * i;
    const float d = c < H ? x[i] - mean : 0.f;
    sq = fmaf(d, d, sq);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + 1e-5f);
  float* y = S.Y + (size_t)row * H;
#pragma unroll
  for (int i = 0; i < HPL; ++i) {
    const int c = lane + 64 * i;
    if (c < H) y[c] = (x[i] - mean) * rstd * S.gamma[c] + S.beta[c];
  }
  if (lane == 0) { S.stats[2 * row] = mean; S.stats[2 * row + 1] = rstd; }
}

// ---------------------------------------------------------------- LayerNorm backward
// dU = (g - mean(g) - xh*mean(g*xh)) * rstd, g = dY*gamma; dgamma += sum dY*xh; dbeta += sum dY
template <int HPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdSeg s0, LnBwdSeg s1, int H, int nb0) {
  __shared__ float red[2][4][64 * HPL];
  const bool first = (int)blockIdx.x < nb0;
  const LnBwdSeg& S = first ? s0 : s1;
  const int blk = first ? blockIdx.x : blockIdx.x - nb0;
  const int nblk = first ? nb0 : (int)gridDim.x - nb0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[HPL], db[HPL], gam[HPL];
#pragma unroll
  for (int i = 0; i < HPL; ++i) {
    const int c = lane + 64 * i;
    dg[i] = 0.f; db[i] = 0.f; gam[i] = c < H ? S.gamma[c] : 0.f;
  }
  for (int row = blk * 4 + wave; row < S.rows; row += nblk * 4) {
    const float mean = S.stats[2 * row], rstd = S.stats[2 * row + 1];
    const float* u = S.U + (size_t)row * H;
    const float* dy = S.dY + (size_t)row * H;
    float xh[HPL], g[HPL];
    float s1_ = 0.f, s2_ = 0.f;
#pragma unroll
    for (int i = 0; i < HPL; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < H;
      const float d = ok ? dy[c] : 0.f;
      xh[i] = ok ? (u[c] - mean) * rstd : 0.f;
      g[i] = d * gam[i];
      s1_ += g[i];
      s2_ = fmaf(g[i], xh[i], s2_);
      dg[i] = fmaf(d, xh[i], dg[i]);
      db[i] += d;
    }
    const float m1 = wave_sum(s1_) / (float)H, m2 = wave_sum(s2_) / (float)H;
    float* du = S.dU + (size_t)row * H;
#pragma unroll
    for (int i = 0; i < HPL; ++i) {
      const int c = lane + 64 * i;
      if (c < H) du[c] = (g[i] - m1 - xh[i] * m2) * rstd;
    }
  }
#pragma unroll
  for (int i = 0; i < HPL; ++i) { red[0][wave][lane + 64 * i] = dg[i]; red[1][wave][lane + 64 * i] = db[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    const float a = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    const float b = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    atomicAdd(S.dgamma + c, a);
    atomicAdd(S.dbeta + c, b);
  }
}

// ---------------------------------------------------------------- per-sample column means
// out[b][c] += (1/n_b) * sum_{rows of b in this block's chunk} X[row][c]   (out zeroed by the caller)
// grid (chunks, B, nseg)
__global__ __launch_bounds__(256) void seg_mean_kernel(SegMean a, SegMean b2, SegMean c3, SegMean d4, int rows_per_block) {
  const SegMean& S = blockIdx.z == 0 ? a : (blockIdx.z == 1 ? b2 : (blockIdx.z == 2 ? c3 : d4));
  if (!S.X) return;
  const int b = blockIdx.y;
  int r0, nr;
  if (S.offs) { r0 = S.offs[b]; nr = S.offs[b + 1] - r0; } else { r0 = b * S.uniform_n; nr = S.uniform_n; }
  const int c0 = blockIdx.x * rows_per_block;
  if (c0 >= nr) return;
  const int c1 = min(nr, c0 + rows_per_block);
  const float inv = 1.0f / (float)nr;
  for (int c = threadIdx.x; c < S.C; c += 256) {
    float acc = 0.f;
    for (int r = c0; r < c1; ++r) acc += S.X[(size_t)(r0 + r) * S.ld + c];
    atomicAdd(S.out + (size_t)b * S.ldo + c, acc * inv);
  }
}

// ---------------------------------------------------------------- ReLU-mask broadcast backward
// dst[t][c] = v[sample(t)][c] * inv_n[sample(t)] * (act[t][c] > 0 ? scale : 0)
__global__ __launch_bounds__(256) void relu_bcast_bwd_kernel(BcastSeg s0, BcastSeg s1, int C, long n0, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const bool first = i < n0;
  const BcastSeg& S = first ? s0 : s1;
  const long k = first ? i : i - n0;
  if (k >= (long)S.rows * C) return;
  const int t = (int)(k / C), c = (int)(k - (long)t * C);
  int sb; float inv;
  if (S.row_sample) { sb = S.row_sample[t]; inv = S.inv_n[sb]; } else { sb = t / S.uniform_n; inv = 1.0f / (float)S.uniform_n; }
  const float a = S.act[k];
  S.dst[k] = a > 0.f ? S.v[(size_t)sb * S.ldv + c] * inv * scale : 0.f;
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
__global__ void loss_kernel(const float* __restrict__ outs, const long long* __restrict__ y,
                            const float* __restrict__ e, const float* __restrict__ s, int B, int C,
                            float* __restrict__ terms, float* __restrict__ d_outs, float* __restrict__ d_pre,
                            int* __restrict__ pred) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int W = 2 * C + 2;
  const float* o = outs + (size_t)b * W;
  const float sc = o[W - 1];
  auto put = [&](int k, float val) {        // gradient w.r.t. output k; d_pre: score column w.r.t. the pre-sigmoid value
    if (d_outs) d_outs[(size_t)b * W + k] = val;
    if (d_pre) d_pre[(size_t)b * W + k] = (k == W - 1) ? val * sc * (1.0f - sc) : val;
  };
  const int yb = (int)y[b];
  // focal on mask logits
  {
    float mx = -INFINITY; int am = 0;
    for (int k = 0; k < C; ++k) if (o[k] > mx) { mx = o[k]; am = k; }
    float z = 0.f;
    for (int k = 0; k < C; ++k) z += expf(o[k] - mx);
    const float pt = expf(o[yb] - mx) / z;
    const float ce = -logf(pt);
    const float at = yb == 1 ? 0.75f : 0.25f;
    const float om = 1.0f - pt;
    terms[4 * b + 0] = 3.0f * at * om * om * om * ce;
    const float dl_dpt = at * (-3.0f * om * om * ce - om * om * om / pt);
    for (int k = 0; k < C; ++k) {
      const float pk = expf(o[k] - mx) / z;
      put(k, 3.0f * dl_dpt * pt * ((k == yb ? 1.0f : 0.0f) - pk));
    }
    if (pred) pred[b] = am;
  }
  // cross entropy on instance logits
  {
    const float* oi = o + C;
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) mx = fmaxf(mx, oi[k]);
    float z = 0.f;
    for (int k = 0; k < C; ++k) z += expf(oi[k] - mx);
    terms[4 * b + 1] = -(oi[yb] - mx - logf(z));
    for (int k = 0; k < C; ++k) put(C + k, expf(oi[k] - mx) / z - (k == yb ? 1.0f : 0.0f));
  }
  // BCE with logits on edge
  {
    const float x = o[2 * C], t = e[b];
    terms[4 * b + 2] = 0.5f * (fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
    put(2 * C, 0.5f * (1.0f / (1.0f + expf(-x)) - t));
  }
  // MSE on the (post-sigmoid) score
  {
    const float x = o[2 * C + 1], t = s[b];
    terms[4 * b + 3] = 0.3f * (x - t) * (x - t);
    put(2 * C + 1, 0.6f * (x - t));
  }
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

}  // namespace

// ------------------------------------------------------------------ launchers
int launch_rowmap(const int* offs, int* row_sample, float* inv_nr, int B, int max_nr, hipStream_t stream) {
  hipLaunchKernelGGL(rowmap_kernel, dim3((max_nr + 255) / 256, B), dim3(256), 0, stream, offs, row_sample, inv_nr);
  return (int)hipGetLastError();
}

int ln_supported(int H) { return H >= 1 && H <= 1024; }

int launch_ln_fwd(const LnSeg& s0, const LnSeg& s1, int H, hipStream_t stream) {
  const int nb0 = (s0.rows + 3) / 4, nb1 = (s1.rows + 3) / 4;
  if (nb0 + nb1 == 0) return 0;
  if (H <= 256) hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(nb0 + nb1), dim3(256), 0, stream, s0, s1, H, nb0);
  else          hipLaunchKernelGGL(ln_fwd_kernel<16>, dim3(nb0 + nb1), dim3(256), 0, stream, s0, s1, H, nb0);
  return (int)hipGetLastError();
}

int launch_ln_bwd(const LnBwdSeg& s0, const LnBwdSeg& s1, int H, hipStream_t stream) {
  auto nblk = [](int rows) { int b = (rows + 15) / 16; return rows == 0 ? 0 : (b > 512 ? 512 : (b < 1 ? 1 : b)); };
  const int nb0 = nblk(s0.rows), nb1 = nblk(s1.rows);
  if (nb0 + nb1 == 0) return 0;
  if (H <= 256) hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nb0 + nb1), dim3(256), 0, stream, s0, s1, H, nb0);
  else          hipLaunchKernelGGL(ln_bwd_kernel<16>, dim3(nb0 + nb1), dim3(256), 0, stream, s0, s1, H, nb0);
  return (int)hipGetLastError();
}

int launch_seg_mean(const SegMean* segs, int nseg, int B, int max_rows, hipStream_t stream) {
  SegMean z{}; SegMean s[4] = {z, z, z, z};
  for (int i = 0; i < nseg && i < 4; ++i) s[i] = segs[i];
  const int rpb = 32;
  hipLaunchKernelGGL(seg_mean_kernel, dim3((max_rows + rpb - 1) / rpb, B, nseg), dim3(256), 0, stream,
                     s[0], s[1], s[2], s[3], rpb);
  return (int)hipGetLastError();
}

int launch_relu_bcast_bwd(const BcastSeg& s0, const BcastSeg& s1, int C, float scale, hipStream_t stream) {
  const long n0 = (long)s0.rows * C, n1 = (long)s1.rows * C;
  if (n0 + n1 == 0) return 0;
  hipLaunchKernelGGL(relu_bcast_bwd_kernel, dim3((unsigned)((n0 + n1 + 255) / 256)), dim3(256), 0, stream, s0, s1, C, n0, scale);
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

int launch_sumsq(const float* g, size_t n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, g, n, out);
  return (int)hipGetLastError();
}

int launch_clip_adamw(float* p, float* g, float* m, float* v, size_t n, float* sumsq, float max_norm,
                      float lr, float b1, float b2, float eps, float wd, int step, int zero_grads, hipStream_t stream) {
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  size_t nb = (n / 4 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(clip_adamw_kernel, dim3((unsigned)nb), dim3(256), 0, stream, p, g, m, v, n, sumsq, max_norm, lr,
                     b1, b2, eps, wd, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), zero_grads);
  return (int)hipGetLastError();
}
