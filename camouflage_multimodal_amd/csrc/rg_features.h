// Region-graph construction downstream of the superpixel segmentation (rg_features.hip): per-region features,
// label adjacency, edge list with weights.  All launchers return hipError_t as int.
#pragma once
#include "common.h"

constexpr int RGF_NACC = 18;      // doubles per label: see rg_features.hip
constexpr int RGF_NFEAT = 15;

struct RgGraphWs {
  double* acc;             // [n_labels][RGF_NACC]           zeroed by the first launch
  unsigned char* adj;      // [n_labels][n_labels] bytes     "     (adj[a][b], a < b: labels a and b touch under 8-connectivity)
  int* rowcount;           // [n_labels] kept neighbours b > a of label a
  int* rowoff;             // [n_labels + 1]
  size_t bytes;
};
RgGraphWs rg_graph_carve(int n_labels, void* base);

int launch_region_graph(const float* image, const int* segments, const unsigned char* canny, int H, int W, int n_labels,
                        const RgGraphWs& ws, float* x, int* region_map, long long* edge_index, float* edge_attr, int edge_capacity,
                        int* counts, hipStream_t stream);
