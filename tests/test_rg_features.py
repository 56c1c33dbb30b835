"""Region-graph construction downstream of the superpixel segmentation (include/camo_rg_features.h).

PARITY UNPINNED (skimage absent, no fixture shipped by the reference): the checker is oracle/rg_features_oracle.py, which
restates extract_rg_embeddings.py:146-236 with the reference's own scipy.ndimage calls.  CPU tests hold the oracle to
first-principles definitions on small inputs; GPU tests hold the HIP kernels to the oracle."""
import numpy as np
import pytest
import torch

from oracle import rg_features_oracle as RO


def _inputs(H, W, n, seed):
    rs = np.random.RandomState(seed)
    seg = RO.voronoi_segments(H, W, n, seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    img = np.clip(np.stack([0.5 + 0.4 * np.sin(7 * xx + seed), 0.5 + 0.4 * np.cos(5 * yy), xx * yy], -1) + 0.05 * rs.standard_normal((H, W, 3)), 0, 1)
    canny = rs.uniform(0, 1, (H, W)) > 0.85
    return img, seg, canny


def test_oracle_features_follow_their_definitions():
    img, seg, canny = _inputs(48, 40, 23, 3)
    x, rmap = RO.region_features(img, seg, canny)
    assert rmap[0] == -1 and (rmap[1:] == np.arange(23)).all() and x.shape == (23, 15)     # label 0 is empty (slic starts at 1)
    luma = img @ RO.LUMA
    H, W = seg.shape
    for lab in (1, 7, 23):
        m = seg == lab
        f = x[rmap[lab]]
        assert np.allclose(f[:3], img[m].mean(0), atol=1e-6) and np.allclose(f[3:6], img[m].std(0), atol=1e-6)
        assert abs(f[6] - luma[m].mean()) < 1e-6 and abs(f[14] - luma[m].var()) < 1e-6 and abs(f[10] - m.sum() / 65536) < 1e-9
        # perimeter: pixels outside the region with a 4-neighbour inside; ring: outside pixels within L1 distance 2
        per, ring = 0, np.zeros_like(m)
        for y in range(H):
            for xq in range(W):
                if m[y, xq]:
                    continue
                near4 = any(0 <= y + dy < H and 0 <= xq + dx < W and m[y + dy, xq + dx] for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)))
                per += near4
                ring[y, xq] = any(0 <= y + dy < H and 0 <= xq + dx < W and m[y + dy, xq + dx]
                                  for dy in range(-2, 3) for dx in range(-2, 3) if 0 < abs(dy) + abs(dx) <= 2)
        assert abs(f[11] - per ** 2 / (4 * np.pi * m.sum() + 1e-10)) < 1e-4 * f[11]
        assert abs(f[12] - np.linalg.norm(img[m].mean(0) - img[ring].mean(0))) < 1e-6
        assert abs(f[13] - canny[m].mean()) < 1e-6
        ys, xs = np.nonzero(m)
        assert abs(f[8] - xs.mean() / 256) < 1e-7 and abs(f[9] - ys.mean() / 256) < 1e-7


def test_oracle_graph_structure():
    img, seg, canny = _inputs(64, 64, 40, 5)
    seg[seg == 17] = 18                                     # an empty label in the middle: indices after it shift down
    x, ei, ea, rmap = RO.region_graph(img, seg, canny)
    assert rmap[17] == -1 and rmap[18] == 16 and x.shape[0] == 39
    assert ei.shape[1] == ea.shape[0] and ei.shape[1] % 2 == 0
    assert (ei[0, 0::2] == ei[1, 1::2]).all() and (ei[1, 0::2] == ei[0, 1::2]).all() and (ea[0::2] == ea[1::2]).all()   # both directions
    assert (ei[0, 0::2] < ei[1, 0::2]).all() and ((0 < ea) & (ea <= 1)).all()
    k = ei[0, 0::2] * 1000 + ei[1, 0::2]
    assert (np.diff(k) > 0).all()                           # sorted, no duplicates
    i, j = ei[0, 0], ei[1, 0]
    w = np.exp(-np.linalg.norm(x[i, :3] - x[j, :3]) / 0.15) * np.exp(-abs(x[i, 6] - x[j, 6]) / 0.08) * np.exp(-abs(x[i, 12] - x[j, 12]) / 0.1)
    assert abs(ea[0] - w) < 1e-6


def test_host_wrapper_needs_a_device_and_valid_labels():
    from camouflage_multimodal_amd import _lib, create_region_graph, create_region_graph_from_segments
    img, seg, canny = _inputs(16, 16, 4, 0)
    with pytest.raises(_lib.CamoError):
        create_region_graph_from_segments(img, seg, canny, device="cpu")
    with pytest.raises(_lib.CamoError, match="scikit-image"):
        create_region_graph(img)                            # slic / canny are skimage's: absent here, and said so


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,n,seed", [(256, 256, 500, 0), (256, 256, 303, 1), (64, 48, 7, 2), (33, 70, 60, 3), (256, 256, 1, 4)])
def test_region_graph_kernels_match_oracle(H, W, n, seed):
    from camouflage_multimodal_amd import create_region_graph_from_segments
    img, seg, canny = _inputs(H, W, n, seed)
    if n > 10:
        seg[seg == 5] = 6                                   # an empty label besides 0
    x, ei, ea, rmap = RO.region_graph(img, seg, canny)
    data, rm = create_region_graph_from_segments(img.astype(np.float32), seg, canny)
    assert (rm.cpu().numpy() == rmap).all()
    gx = data.x.cpu().numpy()
    assert gx.shape == x.shape
    scale = np.maximum(np.abs(x).max(0), 1e-3)
    assert (np.abs(gx - x) <= 2e-6 * scale + 2e-6 * np.abs(x)).all(), np.abs(gx - x).max(0) / scale
    assert (data.edge_index.cpu().numpy() == ei).all()      # same order: sorted (i, j), each followed by its reverse
    ga = data.edge_attr.cpu().numpy()
    assert ga.shape == (ea.shape[0], 1) and (np.abs(ga[:, 0] - ea) <= 2e-5 * ea + 1e-9).all(), np.abs(ga[:, 0] - ea).max()


@pytest.mark.gpu
def test_region_graph_small_capacity_is_regrown_and_feeds_the_gnn():
    from camouflage_multimodal_amd import RegionGraphGNN, create_region_graph_from_segments
    img, seg, canny = _inputs(256, 256, 500, 9)
    x, ei, ea, rmap = RO.region_graph(img, seg, canny)
    data, _ = create_region_graph_from_segments(img.astype(np.float32), seg, canny, edge_capacity=64)     # far too small: reported, regrown
    assert data.edge_index.shape[1] == ei.shape[1] and (data.edge_index.cpu().numpy() == ei).all()
    torch.manual_seed(0)
    gnn = RegionGraphGNN(15, 128, 2).cuda().eval()
    emb = gnn.extract_node_embeddings(data)
    assert emb.shape == (x.shape[0], 128) and torch.isfinite(emb).all()
    g = gnn.extract_graph_embedding(data)
    assert g.shape == (1, 128) and torch.allclose(g, emb.mean(0, keepdim=True), atol=1e-6)
