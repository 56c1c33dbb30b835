"""Region-Graph GNN embedding path.  PARITY UNPINNED: torch_geometric is absent and the
reference ships neither RG weights nor RG fixtures, so (1) the numpy restatement of the published GATConv /
GCNConv algorithms is checked for algebraic self-consistency on CPU, (2) the HIP kernels are checked against that
restatement on the GPU, (3) the host mirror keeps the reference's state_dict names."""
import numpy as np
import pytest
import torch

from oracle import rg_gnn_oracle as RO


def test_oracle_gat_attention_rows_sum_to_one_and_gcn_matches_dense_form():
    p = RO.make_params(1)
    x, ei, ew = RO.make_graph(37, seed=2)
    src, dst, w = RO.with_self_loops(37, ei, ew)
    assert (np.bincount(dst[src == dst], minlength=37) == 1).all()          # exactly one self-loop per node
    _, alpha = RO.gat_conv(x, src, dst, p, return_alpha=True)
    sums = np.zeros((37, 4), dtype=np.float64)
    np.add.at(sums, dst, alpha)
    assert np.abs(sums - 1.0).max() < 1e-5
    # GCN propagation == D^-1/2 (A + I) D^-1/2 X W^T + b with A[i, j] = w(j -> i)
    A = np.zeros((37, 37)); A[dst, src] = w
    d = A.sum(1); Dm = np.diag(1.0 / np.sqrt(d))
    h = np.random.RandomState(0).standard_normal((37, 128)).astype(np.float32)
    want = Dm @ A @ Dm @ (h.astype(np.float64) @ p["conv2.lin.weight"].T.astype(np.float64)) + p["conv2.bias"]
    assert np.abs(RO.gcn_conv(h, src, dst, w, p, 2) - want).max() < 2e-5


def test_oracle_is_permutation_equivariant_and_ignores_isolated_neighbours():
    p = RO.make_params(3)
    x, ei, ew = RO.make_graph(50, seed=4)
    y = RO.node_embeddings(p, x, ei, ew)
    perm = np.random.RandomState(5).permutation(50)
    inv = np.argsort(perm)
    y2 = RO.node_embeddings(p, x[perm], inv[ei], ew)                         # node k of the new graph = old node perm[k]
    assert np.abs(y2 - y[perm]).max() < 2e-5
    # an extra isolated node changes nothing for the others and gets the self-loop-only embedding
    x3 = np.concatenate([x, x[:1]])
    y3 = RO.node_embeddings(p, x3, ei, ew)
    assert np.abs(y3[:50] - y).max() < 1e-6 and np.isfinite(y3[50]).all()


def test_host_mirror_keeps_reference_state_dict_names_and_accepts_old_gat_names():
    from camouflage_multimodal_amd import RegionGraphGNN
    m = RegionGraphGNN()
    keys = set(m.state_dict().keys())
    for name, shape in RO.param_specs():
        assert name in keys and tuple(m.state_dict()[name].shape) == shape, name
    for head in ("fc_mask_1.weight", "fc_mask_2.bias", "fc_instance_1.weight", "fc_edge_2.weight", "bn4.num_batches_tracked"):
        assert head in keys
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["conv1.lin_src.weight"] = sd.pop("conv1.lin.weight"); sd["conv1.lin_dst.weight"] = sd["conv1.lin_src.weight"].clone()
    RegionGraphGNN().load_state_dict(sd, strict=True)                        # torch_geometric < 2.3 checkpoint layout
    with pytest.raises(Exception):
        m(None)
    with pytest.raises(Exception):                                           # CPU tensors: no fallback
        m.extract_node_embeddings(x=torch.zeros(3, 15), edge_index=torch.zeros(2, 0, dtype=torch.long))


def test_csr_builder_matches_oracle_edge_order():
    from camouflage_multimodal_amd import build_target_csr
    x, ei, ew = RO.make_graph(23, seed=7)
    ei = np.concatenate([ei, np.array([[3, 3], [3, 3]])[:, :1]], axis=1); ew = np.concatenate([ew, [0.25]]).astype(np.float32)   # one explicit self-loop
    src, dst, w = RO.with_self_loops(23, ei, ew)
    rowptr, col, wt = build_target_csr(23, torch.from_numpy(ei), torch.from_numpy(ew))
    assert np.array_equal(col.numpy(), src) and np.allclose(wt.numpy(), w)
    assert np.array_equal(rowptr.numpy(), np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=23))]))
    assert w[(src == 3) & (dst == 3)][0] == np.float32(0.25)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(500, 0), (303, 1), (530, 2), (1, 3), (65, 4), (2000, 5)])
def test_hip_node_embeddings_match_oracle(n, seed):
    from camouflage_multimodal_amd import RegionGraphGNN
    p = RO.make_params(seed)
    x, ei, ew = RO.make_graph(n, seed=seed + 10)
    m = RegionGraphGNN()
    sd = m.state_dict()
    for k, v in p.items():
        sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()

    class Data:                                                              # what the reference hands over (:103-107)
        pass
    d = Data(); d.x = torch.from_numpy(x).cuda(); d.edge_index = torch.from_numpy(ei).cuda(); d.edge_attr = torch.from_numpy(ew).cuda().unsqueeze(1)
    got = m.extract_node_embeddings(d).cpu().numpy()
    want = RO.node_embeddings(p, x, ei, ew)
    scale = max(float(np.abs(want).max()), 1e-6)
    assert got.shape == (n, 128) and np.isfinite(got).all()
    assert float(np.abs(got - want).max()) <= 2e-5 * scale + 2e-6, float(np.abs(got - want).max())


@pytest.mark.gpu
def test_hip_batched_graphs_equal_separate_graphs_and_unweighted_edges():
    from camouflage_multimodal_amd import RegionGraphGNN
    torch.manual_seed(0)
    m = RegionGraphGNN().cuda().eval()
    for bn in (m.bn1, m.bn2, m.bn3, m.bn4):
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    gs = [RO.make_graph(n, seed=20 + i) for i, n in enumerate((40, 77, 5))]
    outs = [m.extract_node_embeddings(x=torch.from_numpy(x).cuda(), edge_index=torch.from_numpy(ei).cuda(), edge_attr=torch.from_numpy(ew).cuda())
            for x, ei, ew in gs]
    off = np.cumsum([0] + [g[0].shape[0] for g in gs])
    xb = np.concatenate([g[0] for g in gs]); eib = np.concatenate([g[1] + off[i] for i, g in enumerate(gs)], axis=1); ewb = np.concatenate([g[2] for g in gs])
    ob = m.extract_node_embeddings(x=torch.from_numpy(xb).cuda(), edge_index=torch.from_numpy(eib).cuda(), edge_attr=torch.from_numpy(ewb).cuda())
    # block-diagonal batch (PyG Batch) == per graph, up to the summation order of a row's edges (the device CSR builder
    # allocates their slots with atomics)
    assert float((ob - torch.cat(outs)).abs().max()) < 1e-5 * max(float(ob.abs().max()), 1.0)
    # edge_attr with no elements -> unweighted GCN (extract_rg_embeddings.py:98)
    x, ei, _ = gs[0]
    a = m.extract_node_embeddings(x=torch.from_numpy(x).cuda(), edge_index=torch.from_numpy(ei).cuda(), edge_attr=torch.zeros(0, 1).cuda())
    b = m.extract_node_embeddings(x=torch.from_numpy(x).cuda(), edge_index=torch.from_numpy(ei).cuda(), edge_attr=torch.ones(ei.shape[1]).cuda())
    assert float((a - b).abs().max()) < 1e-5 * max(float(a.abs().max()), 1.0)


@pytest.mark.gpu
def test_graph_to_logits_on_device_equals_two_stage_inference(kg_real):
    """predict_from_region_graph (region graph -> RG GNN -> fusion model, nothing leaves the device) against running the
    two stages by hand through the host, the way the reference script does (test_multimodal.py:93-106)."""
    from camouflage_multimodal_amd import RegionGraphGNN, build_multimodal_model, predict_from_embeddings, predict_from_region_graph
    torch.manual_seed(1)
    rgm = RegionGraphGNN().cuda().eval()
    fm = build_multimodal_model({}).cuda().eval().set_precision("f32")
    x, ei, ew = RO.make_graph(481, seed=3)

    class Data:
        pass
    d = Data(); d.x = torch.from_numpy(x); d.edge_index = torch.from_numpy(ei); d.edge_attr = torch.from_numpy(ew).unsqueeze(1)
    kg = {f"cat{i:02d}": torch.from_numpy(kg_real[i:i + 1]) for i in range(13)}
    pred, attn, _ = predict_from_region_graph(fm, rgm, d, kg, "cuda")
    emb = rgm.extract_node_embeddings(x=d.x.cuda(), edge_index=d.edge_index.cuda(), edge_attr=d.edge_attr.cuda()).cpu()
    pred2, _, _ = predict_from_embeddings(fm, emb, kg, "cuda")
    # (equal up to the summation order of the fusion model's atomically accumulated mean pools)
    assert torch.allclose(pred["mask_logits"], pred2["mask_logits"], rtol=0, atol=1e-6) and pred["mask_pred"] == pred2["mask_pred"]
    assert abs(pred["score"] - pred2["score"]) < 1e-6 and attn is not None


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 23, 500, 5000])
def test_device_csr_builder_matches_host_builder_row_by_row(n):
    from camouflage_multimodal_amd import build_target_csr
    from camouflage_multimodal_amd.region_graph import build_target_csr_device
    x, ei, ew = RO.make_graph(n, seed=n)
    if n > 3:                                                                 # an explicit self-loop keeps its weight
        ei = np.concatenate([ei, np.array([[3], [3]])], axis=1); ew = np.concatenate([ew, [0.25]]).astype(np.float32)
    for weights in (ew, None):
        eit = torch.from_numpy(ei).cuda(); ewt = None if weights is None else torch.from_numpy(weights).cuda()
        r0, c0, w0 = [t.cpu().numpy() for t in build_target_csr(n, eit, ewt)]
        r1, c1, w1 = [t.cpu().numpy() for t in build_target_csr_device(n, eit, ewt)]
        assert np.array_equal(r0, r1)
        for i in range(n):
            a = sorted(zip(c0[r0[i]:r0[i + 1]].tolist(), w0[r0[i]:r0[i + 1]].tolist()))
            b = sorted(zip(c1[r1[i]:r1[i + 1]].tolist(), w1[r1[i]:r1[i + 1]].tolist()))
            assert a == b, i
            assert c1[r1[i]] == i                                             # the self-loop leads its row
