// Sparse kernels of the Region-Graph GNN embedding path (rg_gnn.hip).  All return hipError_t as int.
// Graph = CSR by TARGET node with exactly one self-loop per node (the host mirror builds it from edge_index):
// rowptr [N+1], col [E] = source node of every incoming edge, w [E] = edge weight.
#pragma once
#include "common.h"

// COO (src, dst int64 [E], w fp32 [E] or null) -> CSR by target with one self-loop per node (an explicit self-loop keeps
// its weight, the others get 1).  counts, cursor: int [N] scratch; loopw: float [N] scratch; col / wout sized E + N.
// The order of a row's edges after its leading self-loop is not deterministic (atomic slot allocation).
int launch_build_csr(const long long* src, const long long* dst, const float* w, int N, int E, int* counts, float* loopw, int* cursor,
                     int* rowptr, int* col, float* wout, hipStream_t stream);

struct BnEval { const float* weight; const float* bias; const float* mean; const float* var; };   // BatchNorm1d, eval mode

// dinv[i] = (sum of row i's weights)^-1/2, 0 for an empty / zero-weight row (gcn_norm)
int launch_gcn_dinv(const int* rowptr, const float* w, float* dinv, int N, hipStream_t stream);
// a_src[n,k] = <Hh[n,k,:], att_src[k,:]>, a_dst likewise; Hh [N, heads, C]
int launch_gat_alpha(const float* Hh, const float* att_src, const float* att_dst, float* a_src, float* a_dst, int N, int heads, int C,
                     hipStream_t stream);
// out[i,:] = relu(bn(mean_k sum_{j->i} softmax_j(leaky_relu(a_src[j,k] + a_dst[i,k], 0.2)) Hh[j,k,:] + bias))
int launch_gat_aggregate(const float* Hh, const float* a_src, const float* a_dst, const int* rowptr, const int* col, const float* bias,
                         BnEval bn, float* out, int N, int heads, int C, hipStream_t stream);
// out[i,:] = relu(bn(sum_{j->i} dinv[j] w dinv[i] XW[j,:] + bias))
int launch_gcn_aggregate(const float* XW, const int* rowptr, const int* col, const float* w, const float* dinv, const float* bias,
                         BnEval bn, float* out, int N, int C, hipStream_t stream);
