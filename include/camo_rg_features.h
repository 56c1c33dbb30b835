/* C ABI of the Region-Graph construction on MI355X ("next" row 4), exported by the same libcamo_fusion.so
 * as include/camo_fusion.h (error text: camo_last_error()).
 *
 * Stands behind the body of create_region_graph (models/region_graph/extract_rg_embeddings.py:138-246) BETWEEN its three
 * skimage calls: given the image, the superpixel label map `segments` (slic, :144) and the boolean edge map (canny,
 * :152), it produces what the function returns -- node features x [n, 15] (:154-228), edge_index, edge weights
 * (:219-236; the RAG's edge set = pairs of labels that touch under 8-connectivity) -- on the device, ready for
 * the embedding call of camo_rg_gnn.h.  The reference does this in an O(regions x pixels) numpy loop, ~2 s per image.
 * PARITY UNPINNED: skimage is absent here and the reference ships no region-graph fixture; the arithmetic between the
 * skimage calls is restated in oracle/rg_features_oracle.py (with the reference's own scipy.ndimage calls), which these
 * kernels are tested against.  slic and canny themselves stay the caller's.
 *
 * Regions are renumbered in increasing label order, empty labels dropped (the reference's region_id_map); edges come
 * sorted by (i, j), i < j, each followed by its reverse (the reference's order is networkx's: a permutation).
 * Device pointers only, enqueue-only on `stream`, 0 = ok / negative CAMO_E_* as in camo_fusion.h. */
#ifndef CAMO_RG_FEATURES_H
#define CAMO_RG_FEATURES_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CAMO_RG_NFEAT 15
#define CAMO_RG_MAX_LABELS 4096

size_t camo_rg_graph_workspace_bytes(int32_t n_labels);

/* image [H, W, 3] fp32 in [0, 1]; segments [H, W] int32 labels in [0, n_labels); canny [H, W] uint8 (0 / 1).
 * Out: x [n_labels, 15] (first counts[0] rows valid), region_map [n_labels] (new index or -1),
 * edge_index [2, edge_capacity] int64 (row 0 sources, row 1 targets; first counts[1] columns valid), edge_attr
 * [edge_capacity], counts [2] = {regions kept, directed edges}.  When the graph has more than edge_capacity directed
 * edges, counts[1] still reports the number needed and nothing is written beyond the capacity.  The position features
 * divide by 256 and the size feature by 256^2 whatever H and W are, as the reference does (its images are 256 x 256). */
int camo_rg_region_graph(const float* image, const int32_t* segments, const uint8_t* canny, int32_t H, int32_t W,
                         int32_t n_labels, void* workspace, size_t workspace_bytes, float* x, int32_t* region_map,
                         int64_t* edge_index, float* edge_attr, int32_t edge_capacity, int32_t* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif
