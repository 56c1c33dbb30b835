/* C ABI of the Region-Graph GNN embedding path on MI355X ("next" row 3), exported by the same
 * libcamo_fusion.so as include/camo_fusion.h (error text: camo_last_error()).
 *
 * Stands behind RegionGraphGNN.extract_node_embeddings (models/region_graph/extract_rg_embeddings.py:94-122), the
 * step that feeds the fusion model at inference (models/multimodal/test_multimodal.py:93):
 *   GATConv(in -> hidden, heads, concat=False) -> BatchNorm1d(eval) -> ReLU
 *   3 x [ GCNConv(hidden -> hidden, edge_weight) -> BatchNorm1d(eval) -> ReLU ]
 *   Linear(hidden -> hidden) -> ReLU                                    -> node embeddings [N, hidden]
 * PARITY UNPINNED: the graph layers are torch_geometric's (absent here, no version pinned by the reference, no RG
 * checkpoint or fixture shipped); their published algorithms are restated in oracle/rg_gnn_oracle.py, which the HIP
 * kernels are tested against.
 *
 * Device pointers only, fp32, enqueue-only on `stream`, 0 = ok / negative CAMO_E_* as in camo_fusion.h. */
#ifndef CAMO_RG_GNN_H
#define CAMO_RG_GNN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct camo_rg_dims {
  int32_t in_channels; /* 15 region features (extract_rg_embeddings.py:213-226) */
  int32_t hidden;      /* 128 */
  int32_t heads;       /* 4 (GATConv heads, concat=False) */
} camo_rg_dims_t;

/* parameter table: device pointers in this order (state_dict names of the reference module) */
enum {
  CAMO_RG_C1_ATT_SRC = 0, /* conv1.att_src [1, heads, hidden] */
  CAMO_RG_C1_ATT_DST,     /* conv1.att_dst */
  CAMO_RG_C1_BIAS,        /* conv1.bias [hidden] */
  CAMO_RG_C1_W,           /* conv1.lin.weight [heads*hidden, in] (lin_src.weight in older torch_geometric) */
  CAMO_RG_BN1,            /* bn1.weight, .bias, .running_mean, .running_var : 4 consecutive slots */
  CAMO_RG_C2_BIAS = CAMO_RG_BN1 + 4, /* conv2.bias, conv2.lin.weight [hidden, hidden], bn2 x 4 : 6 slots; conv3, conv4 follow */
  CAMO_RG_FC_W = CAMO_RG_C2_BIAS + 18, /* fc_shared.weight [hidden, hidden] */
  CAMO_RG_FC_B,                        /* fc_shared.bias */
  CAMO_RG_NPARAMS
};

size_t camo_rg_workspace_bytes(const camo_rg_dims_t* dims, int32_t N);

/* The graph as the reference builds it (extract_rg_embeddings.py:215-246: edge_index [2, E] int64 with row 0 = source,
 * row 1 = target, contiguous; edge_weight [E] fp32 or NULL) -> the CSR-by-target the kernels walk: one self-loop per
 * node first in its row (an explicit self-loop keeps its weight, the others get weight 1 -- PyG's
 * add_remaining_self_loops; GATConv's remove-then-add has the same structure), then the incoming edges in no
 * particular order.  scratch: 3 N int32; rowptr [N+1]; col, w: [E + N]. */
int camo_rg_build_csr(const int64_t* edge_index, const float* edge_weight, int32_t N, int32_t E, int32_t* scratch,
                      int32_t* rowptr, int32_t* col, float* w, void* stream);

/* Graph: CSR by TARGET node with exactly one self-loop per node already inserted (existing self-loops keep their
 * weight, added ones have weight 1 -- PyG's add_remaining_self_loops; GATConv ignores the weights):
 * rowptr [N+1], col [E] = source of each incoming edge, w [E].  Several graphs batch as one block-diagonal graph.
 * x [N, in_channels] -> out [N, hidden]. */
int camo_rg_node_embeddings(const camo_rg_dims_t* dims, const float* const* params, const float* x,
                            const int32_t* rowptr, const int32_t* col, const float* w, int32_t N, int32_t E,
                            void* workspace, size_t workspace_bytes, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
