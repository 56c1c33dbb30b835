"""CPU restatement of the Region-Graph construction DOWNSTREAM of the superpixel segmentation.  TEST INFRASTRUCTURE
(see oracle/__init__.py): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

PARITY UNPINNED.  The reference's ``create_region_graph`` (models/region_graph/extract_rg_embeddings.py:138-246) starts with
``skimage.segmentation.slic`` (:144), ``skimage.feature.canny`` (:152) and ends with ``skimage.graph.rag_mean_color`` (:215);
skimage is not importable here, the reference pins no version of it and ships no region-graph fixture (its
``rg_embeddings/*.pt`` are LFS pointers).  What is restated here is everything BETWEEN those calls, taking their results as
inputs -- the label map ``segments`` and the boolean edge map ``edges_canny`` -- plus the one property of the RAG the
reference uses, its edge set: two labels are joined iff some pixel of one has a pixel of the other in its 3x3
neighbourhood (``skimage.graph.RAG(label_image, connectivity=2)``, the default of rag_mean_color).  The per-region
arithmetic is the reference's, with the same scipy.ndimage calls (scipy IS available):

  15 features per non-empty region [:154-228]: mean RGB (3), std RGB (3, population), mean / std of the luma
  0.2989 R + 0.5870 G + 0.1140 B, centroid x / 256 and y / 256, pixel count / 256^2, compactness = perimeter^2 / (4 pi area
  + 1e-10) with perimeter = |binary_dilation(mask) xor mask|, boundary contrast = || mean RGB - mean RGB of the ring
  binary_dilation(mask, iterations=2) minus mask ||, mean of the edge map over the region, variance of the luma;
  float64 arithmetic, rounded to float32 at the end; regions renumbered in increasing label order, empty labels dropped.
  Edges [:219-236]: both directions of every RAG edge between kept regions, weight
  exp(-|d mean RGB| / 0.15) * exp(-|d luma mean| / 0.08) * exp(-|d ring contrast| / 0.1) computed from the float32 features.

The reference emits edges in networkx's iteration order; here they come sorted by (i, j), i < j, each followed by its
reverse -- a permutation the GNN is invariant to.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

LUMA = np.array([0.2989, 0.5870, 0.1140])


def region_features(image, segments, edges_canny):
    """image [H, W, 3] float in [0, 1]; segments [H, W] int labels >= 0; edges_canny [H, W] bool.
    Returns (x float32 [n, 15], region_map int32 [max label + 1] with -1 for empty labels)."""
    image = np.asarray(image, np.float64)
    segments = np.asarray(segments)
    n_labels = int(segments.max()) + 1
    luma = image[..., :3] @ LUMA
    rows, region_map = [], np.full(n_labels, -1, np.int32)
    for rid in range(n_labels):
        mask = segments == rid
        area = int(mask.sum())
        if area == 0:
            continue
        px, gp = image[mask], luma[mask]
        ys, xs = np.nonzero(mask)
        perimeter = int((ndimage.binary_dilation(mask) ^ mask).sum())
        ring = ndimage.binary_dilation(mask, iterations=2) & ~mask
        mean_rgb = px.mean(axis=0)
        contrast = float(np.linalg.norm(mean_rgb - image[ring].mean(axis=0))) if ring.any() else 0.0
        feat = np.concatenate([mean_rgb, px.std(axis=0), [gp.mean(), gp.std(), xs.mean() / 256.0, ys.mean() / 256.0,
                                                          area / (256 * 256), perimeter ** 2 / (4 * np.pi * area + 1e-10),
                                                          contrast, edges_canny[mask].mean(), gp.var()]])
        region_map[rid] = len(rows)
        rows.append(np.nan_to_num(feat, nan=0.0))
    return np.asarray(rows, np.float64).astype(np.float32).reshape(-1, 15), region_map


def adjacent_label_pairs(segments):
    """Sorted unique (a, b), a < b, of labels that touch under 8-connectivity."""
    s = np.asarray(segments)
    H, W = s.shape
    pairs = set()
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):       # the other four directions are these four seen from the other pixel
        a = s[:H - dy, max(0, -dx):W - max(0, dx)]
        b = s[dy:, max(0, dx):W + min(0, dx)]
        m = a != b
        lo, hi = np.minimum(a[m], b[m]), np.maximum(a[m], b[m])
        pairs.update(zip(lo.tolist(), hi.tolist()))
    return sorted(pairs)


def region_graph(image, segments, edges_canny):
    """-> (x [n, 15] f32, edge_index [2, E] int64, edge_attr [E] f32, region_map)."""
    x, region_map = region_features(image, segments, edges_canny)
    src, dst, w = [], [], []
    for a, b in adjacent_label_pairs(segments):
        i, j = int(region_map[a]), int(region_map[b])
        if i < 0 or j < 0:
            continue
        color = np.linalg.norm(x[i, :3] - x[j, :3])                       # float32 operands, as the reference's tensors
        wt = np.exp(-color / np.float32(0.15)) * np.exp(-np.abs(x[i, 6] - x[j, 6]) / np.float32(0.08)) * \
            np.exp(-np.abs(x[i, 12] - x[j, 12]) / np.float32(0.1))
        src += [i, j]; dst += [j, i]; w += [wt, wt]
    return x, np.asarray([src, dst], np.int64).reshape(2, -1), np.asarray(w, np.float32), region_map


def voronoi_segments(H, W, n, seed):
    """Synthetic stand-in for a SLIC label map: nearest-seed regions of n jittered grid points (labels start at 1 like
    skimage >= 0.19's slic, so label 0 is empty -- the reference's loop skips it [:155-157])."""
    rs = np.random.RandomState(seed)
    g = int(np.ceil(np.sqrt(n)))
    pts = np.stack(np.meshgrid(np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 2)[:n].astype(np.float64)
    pts = (pts + 0.5 + 0.35 * rs.uniform(-1, 1, pts.shape)) * np.array([H / g, W / g])
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d = (yy[..., None] - pts[:, 0]) ** 2 + (xx[..., None] - pts[:, 1]) ** 2
    return d.argmin(-1).astype(np.int32) + 1
