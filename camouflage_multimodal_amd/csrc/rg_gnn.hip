// Sparse gather kernels of the Region-Graph GNN embedding path (GATConv + 3 x GCNConv with
// eval-mode BatchNorm + ReLU fused in).  A region-adjacency graph has ~500 nodes of in-degree 5-8 per image, so the
// work is latency- and HBM-bound gathers: one wave (GCN) or one block of `heads` waves (GAT) per target node, the
// node's incoming edges walked from a CSR-by-target, source rows read as coalesced 16-byte pieces.  Dense
// projections go through the exact-fp32 grouped GEMM (gemm.hip).  Algorithm source: oracle/rg_gnn_oracle.py header.
#include "rg_gnn.h"

namespace {

__device__ __forceinline__ float bn_relu(float v, const BnEval& bn, int c) {
  v = (v - bn.mean[c]) / sqrtf(bn.var[c] + 1e-5f) * bn.weight[c] + bn.bias[c];
  return fmaxf(v, 0.f);
}

__global__ void gcn_dinv_kernel(const int* __restrict__ rowptr, const float* __restrict__ w, float* __restrict__ dinv, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float d = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) d += w[e];
  dinv[i] = d > 0.f ? 1.0f / sqrtf(d) : 0.f;
}

// grid N, block 64 * heads: wave k of block n
__global__ void gat_alpha_kernel(const float* __restrict__ Hh, const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                 float* __restrict__ a_src, float* __restrict__ a_dst, int heads, int C) {
  const int n = blockIdx.x, k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* h = Hh + ((size_t)n * heads + k) * C;
  float s = 0.f, d = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = h[c]; s = fmaf(v, att_src[k * C + c], s); d = fmaf(v, att_dst[k * C + c], d); }
  s = wave_sum(s); d = wave_sum(d);
  if (lane == 0) { a_src[n * heads + k] = s; a_dst[n * heads + k] = d; }
}

// grid N, block 64 * heads (heads <= 8): wave k runs head k's softmax-weighted sum with an online softmax, the heads
// are averaged through LDS.  CPL = channels per lane (C <= 64 * CPL).
template <int CPL>
__global__ void gat_aggregate_kernel(const float* __restrict__ Hh, const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                     const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ bias,
                                     BnEval bn, float* __restrict__ out, int heads, int C) {
  __shared__ float red[8][64 * CPL];
  const int i = blockIdx.x, k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float ad = a_dst[i * heads + k];
  float m = -INFINITY, s = 0.f, acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e];
    float v = a_src[j * heads + k] + ad;
    v = v > 0.f ? v : 0.2f * v;
    const float mn = fmaxf(m, v);
    const float corr = __expf(m - mn), p = __expf(v - mn);       // (m = -inf on the first edge: corr = 0)
    s = s * corr + p;
    const float* h = Hh + ((size_t)j * heads + k) * C;
#pragma unroll
    for (int q = 0; q < CPL; ++q) { const int c = lane + 64 * q; acc[q] = acc[q] * corr + (c < C ? p * h[c] : 0.f); }
    m = mn;
  }
  const float inv = s > 0.f ? 1.0f / s : 0.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) red[k][lane + 64 * q] = acc[q] * inv;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = 0.f;
    for (int kk = 0; kk < heads; ++kk) v += red[kk][c];
    out[(size_t)i * C + c] = bn_relu(v / (float)heads + bias[c], bn, c);
  }
}

// one wave per target node, 4 nodes per block
template <int CPL>
__global__ void gcn_aggregate_kernel(const float* __restrict__ XW, const int* __restrict__ rowptr, const int* __restrict__ col,
                                     const float* __restrict__ w, const float* __restrict__ dinv, const float* __restrict__ bias,
                                     BnEval bn, float* __restrict__ out, int N, int C) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  const float di = dinv[i];
  float acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e];
    const float nrm = dinv[j] * w[e] * di;
    const float* h = XW + (size_t)j * C;
#pragma unroll
    for (int q = 0; q < CPL; ++q) { const int c = lane + 64 * q; if (c < C) acc[q] = fmaf(nrm, h[c], acc[q]); }
  }
#pragma unroll
  for (int q = 0; q < CPL; ++q) { const int c = lane + 64 * q; if (c < C) out[(size_t)i * C + c] = bn_relu(acc[q] + bias[c], bn, c); }
}

// ---- CSR-by-target construction from a COO edge list (counting sort by target; one self-loop per node) ----
// counts[i] = number of non-loop edges into i; loopw[i] = weight of an explicit self-loop of i (else stays 1)
__global__ void csr_count_kernel(const long long* __restrict__ src, const long long* __restrict__ dst, const float* __restrict__ w,
                                 int E, int* __restrict__ counts, float* __restrict__ loopw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = (int)src[e], d = (int)dst[e];
  if (s == d) loopw[s] = w ? w[e] : 1.0f;
  else atomicAdd(counts + d, 1);
}
// one block: rowptr = exclusive scan of (counts + 1); cursor = rowptr (the first slot of a row takes the self-loop)
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int* __restrict__ counts, const float* __restrict__ loopw, int N,
                                                        int* __restrict__ rowptr, int* __restrict__ cursor, int* __restrict__ col,
                                                        float* __restrict__ wout) {
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (N + 1023) / 1024, b0 = t * per, b1 = min(N, b0 + per);
  int s = 0;
  for (int i = b0; i < b1; ++i) s += counts[i] + 1;
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = t ? part[t - 1] : 0;
  for (int i = b0; i < b1; ++i) {
    rowptr[i] = run;
    col[run] = i; wout[run] = loopw[i];                     // the self-loop
    cursor[i] = run + 1;
    run += counts[i] + 1;
  }
  if (t == 1023) rowptr[N] = part[1023];
}
__global__ void csr_fill_kernel(const long long* __restrict__ src, const long long* __restrict__ dst, const float* __restrict__ w,
                                int E, int* __restrict__ cursor, int* __restrict__ col, float* __restrict__ wout) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = (int)src[e], d = (int)dst[e];
  if (s == d) return;
  const int pos = atomicAdd(cursor + d, 1);
  col[pos] = s; wout[pos] = w ? w[e] : 1.0f;
}
__global__ void csr_init_kernel(int* __restrict__ counts, float* __restrict__ loopw, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { counts[i] = 0; loopw[i] = 1.0f; }
}

}  // namespace

int launch_build_csr(const long long* src, const long long* dst, const float* w, int N, int E, int* counts, float* loopw, int* cursor,
                     int* rowptr, int* col, float* wout, hipStream_t stream) {
  hipLaunchKernelGGL(csr_init_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, counts, loopw, N);
  if (E > 0) hipLaunchKernelGGL(csr_count_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, src, dst, w, E, counts, loopw);
  hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, loopw, N, rowptr, cursor, col, wout);
  if (E > 0) hipLaunchKernelGGL(csr_fill_kernel, dim3((E + 255) / 256), dim3(256), 0, stream, src, dst, w, E, cursor, col, wout);
  return (int)hipGetLastError();
}

int launch_gcn_dinv(const int* rowptr, const float* w, float* dinv, int N, hipStream_t stream) {
  hipLaunchKernelGGL(gcn_dinv_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, rowptr, w, dinv, N);
  return (int)hipGetLastError();
}

int launch_gat_alpha(const float* Hh, const float* att_src, const float* att_dst, float* a_src, float* a_dst, int N, int heads, int C,
                     hipStream_t stream) {
  if (heads < 1 || heads > 8) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gat_alpha_kernel, dim3(N), dim3(64 * heads), 0, stream, Hh, att_src, att_dst, a_src, a_dst, heads, C);
  return (int)hipGetLastError();
}

int launch_gat_aggregate(const float* Hh, const float* a_src, const float* a_dst, const int* rowptr, const int* col, const float* bias,
                         BnEval bn, float* out, int N, int heads, int C, hipStream_t stream) {
  if (heads < 1 || heads > 8 || C > 512) return (int)hipErrorInvalidValue;
  if (C <= 128) hipLaunchKernelGGL(gat_aggregate_kernel<2>, dim3(N), dim3(64 * heads), 0, stream, Hh, a_src, a_dst, rowptr, col, bias, bn, out, heads, C);
  else          hipLaunchKernelGGL(gat_aggregate_kernel<8>, dim3(N), dim3(64 * heads), 0, stream, Hh, a_src, a_dst, rowptr, col, bias, bn, out, heads, C);
  return (int)hipGetLastError();
}

int launch_gcn_aggregate(const float* XW, const int* rowptr, const int* col, const float* w, const float* dinv, const float* bias,
                         BnEval bn, float* out, int N, int C, hipStream_t stream) {
  if (C > 512) return (int)hipErrorInvalidValue;
  if (C <= 128) hipLaunchKernelGGL(gcn_aggregate_kernel<2>, dim3((N + 3) / 4), dim3(256), 0, stream, XW, rowptr, col, w, dinv, bias, bn, out, N, C);
  else          hipLaunchKernelGGL(gcn_aggregate_kernel<8>, dim3((N + 3) / 4), dim3(256), 0, stream, XW, rowptr, col, w, dinv, bias, bn, out, N, C);
  return (int)hipGetLastError();
}
