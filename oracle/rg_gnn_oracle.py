"""CPU oracle of the Region-Graph GNN embedding path -- TEST INFRASTRUCTURE ONLY.

Restates ``RegionGraphGNN.extract_node_embeddings`` (models/region_graph/extract_rg_embeddings.py:94-122):
GATConv(15 -> 128, heads = 4, concat = False) -> BatchNorm1d(eval) -> ReLU, three times
[GCNConv(128 -> 128, edge_weight) -> BatchNorm1d(eval) -> ReLU], Linear(128 -> 128) -> ReLU.

PARITY UNPINNED.  The two graph layers live in ``torch_geometric`` (imported at :17, no version pinned anywhere in
the reference), which is not installable here, and the reference ships no RG checkpoint or RG fixture (its
``rg_embeddings/*.pt`` are LFS pointers).  The layer algorithms are therefore restated from their published
definitions and PyG's documented defaults:

* GATConv (Velickovic et al., ICLR 2018; PyG ``GATConv`` defaults ``negative_slope=0.2, add_self_loops=True,
  bias=True``): h = x W^T reshaped [N, heads, C]; a_src[n,k] = <h[n,k], att_src[k]>, a_dst likewise; self-loops of
  the input are removed and one self-loop per node is added; for every edge j -> i (edge_index[0] = source j,
  edge_index[1] = target i) e = leaky_relu(a_src[j] + a_dst[i]); alpha = softmax of e over the incoming edges of i;
  out[i,k] = sum_j alpha[j->i,k] h[j,k]; ``concat=False`` averages the heads; + bias.
* GCNConv (Kipf & Welling, ICLR 2017; PyG ``gcn_norm`` with ``add_self_loops=True, improved=False``): nodes without a
  self-loop get one of weight 1 (existing self-loop weights are kept); deg[i] = sum of incoming weights;
  out[i] = sum_{j->i} deg[j]^-1/2 w deg[i]^-1/2 (x W^T)[j] + bias  (deg^-1/2 := 0 where deg = 0).
* BatchNorm1d in eval mode: (x - running_mean) / sqrt(running_var + 1e-5) * weight + bias.

What the tests can and do check: algebraic self-consistency of this restatement (attention rows sum to one,
permutation equivariance, the dense-matrix form of the GCN propagation) and bit-level agreement of the HIP kernels
with it.  Only tests/ may import this module.
"""
import numpy as np

f32 = np.float32


def param_specs(in_channels=15, hidden=128, heads=4):
    """(name, shape) in state_dict order of the layers the embedding path uses (PyG >= 2.3 names)."""
    specs = [("conv1.att_src", (1, heads, hidden)), ("conv1.att_dst", (1, heads, hidden)), ("conv1.bias", (hidden,)),
             ("conv1.lin.weight", (heads * hidden, in_channels))]
    specs += [("bn1.weight", (hidden,)), ("bn1.bias", (hidden,)), ("bn1.running_mean", (hidden,)), ("bn1.running_var", (hidden,))]
    for k in (2, 3, 4):
        specs += [(f"conv{k}.bias", (hidden,)), (f"conv{k}.lin.weight", (hidden, hidden)),
                  (f"bn{k}.weight", (hidden,)), (f"bn{k}.bias", (hidden,)), (f"bn{k}.running_mean", (hidden,)), (f"bn{k}.running_var", (hidden,))]
    specs += [("fc_shared.weight", (hidden, hidden)), ("fc_shared.bias", (hidden,))]
    return specs


def make_params(seed=0, in_channels=15, hidden=128, heads=4):
    rs = np.random.RandomState(seed)
    p = {}
    for name, shape in param_specs(in_channels, hidden, heads):
        if name.endswith("running_var"):
            p[name] = (0.5 + rs.uniform(size=shape)).astype(f32)
        elif name.endswith("running_mean"):
            p[name] = (0.1 * rs.standard_normal(shape)).astype(f32)
        elif name.startswith("bn") and name.endswith("weight"):
            p[name] = (1.0 + 0.1 * rs.standard_normal(shape)).astype(f32)
        elif name.endswith("bias"):
            p[name] = (0.05 * rs.standard_normal(shape)).astype(f32)
        else:
            fan_in = shape[-1]
            p[name] = (rs.standard_normal(shape) / np.sqrt(fan_in)).astype(f32)
    return p


def make_graph(n, seed=0, in_channels=15):
    """A region-adjacency-like graph: nodes on a jittered grid, 4-neighbour edges plus a few diagonals, both
    directions, weights in (0, 1] (the reference's exp(-d/.15) exp(-d/.08) exp(-d/.1) products, :228-238)."""
    rs = np.random.RandomState(seed)
    w = max(1, int(np.sqrt(n)))
    und = set()
    for i in range(n):
        r, c = divmod(i, w)
        for j in (i + 1 if c + 1 < w else -1, i + w, i + w + 1 if (c + 1 < w and rs.uniform() < 0.2) else -1):
            if 0 <= j < n and j != i:
                und.add((min(i, j), max(i, j)))
    src, dst, wt = [], [], []
    for (i, j) in sorted(und):
        v = float(np.exp(-rs.uniform(0, 3)))
        src += [i, j]; dst += [j, i]; wt += [v, v]
    x = rs.uniform(0, 1, size=(n, in_channels)).astype(f32)
    return x, np.array([src, dst], dtype=np.int64).reshape(2, -1), np.array(wt, dtype=f32)


def with_self_loops(n, edge_index, edge_weight):
    """One self-loop per node: existing self-loops keep their weight (GCN semantics; GAT ignores weights), the
    others get weight 1.  Returns (src, dst, w) sorted by (dst, src)."""
    src, dst = edge_index[0].astype(np.int64), edge_index[1].astype(np.int64)
    w = np.ones(src.shape[0], dtype=f32) if edge_weight is None else edge_weight.astype(f32).reshape(-1)
    loop = src == dst
    lw = np.ones(n, dtype=f32)
    lw[src[loop]] = w[loop]                       # (a repeated self-loop keeps the last weight)
    src = np.concatenate([src[~loop], np.arange(n)]); dst = np.concatenate([dst[~loop], np.arange(n)])
    w = np.concatenate([w[~loop], lw])
    order = np.lexsort((src, dst))
    return src[order], dst[order], w[order]


def bn_eval(x, p, k):
    return (x - p[f"bn{k}.running_mean"]) / np.sqrt(p[f"bn{k}.running_var"] + f32(1e-5)) * p[f"bn{k}.weight"] + p[f"bn{k}.bias"]


def gat_conv(x, src, dst, p, heads=4, return_alpha=False):
    n = x.shape[0]
    hidden = p["conv1.bias"].shape[0]
    h = (x @ p["conv1.lin.weight"].T).reshape(n, heads, hidden).astype(f32)
    a_src = (h * p["conv1.att_src"].reshape(1, heads, hidden)).sum(-1)
    a_dst = (h * p["conv1.att_dst"].reshape(1, heads, hidden)).sum(-1)
    e = a_src[src] + a_dst[dst]
    e = np.where(e > 0, e, f32(0.2) * e).astype(f32)
    out = np.zeros((n, heads, hidden), dtype=f32)
    alpha = np.zeros_like(e)
    for i in range(n):
        m = dst == i
        if not m.any():
            continue
        ei = e[m]
        a = np.exp(ei - ei.max(axis=0, keepdims=True))
        a = (a / a.sum(axis=0, keepdims=True)).astype(f32)
        alpha[m] = a
        out[i] = (a[:, :, None] * h[src[m]]).sum(axis=0)
    y = out.mean(axis=1) + p["conv1.bias"]
    return (y.astype(f32), alpha) if return_alpha else y.astype(f32)


def gcn_conv(x, src, dst, w, p, k):
    n = x.shape[0]
    deg = np.zeros(n, dtype=f32)
    np.add.at(deg, dst, w)
    dinv = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, f32(1e-30))), 0.0).astype(f32)
    norm = dinv[src] * w * dinv[dst]
    xw = (x @ p[f"conv{k}.lin.weight"].T).astype(f32)
    out = np.zeros_like(xw)
    np.add.at(out, dst, norm[:, None] * xw[src])
    return (out + p[f"conv{k}.bias"]).astype(f32)


def node_embeddings(p, x, edge_index, edge_weight, heads=4):
    """extract_node_embeddings (extract_rg_embeddings.py:94-122) -> [N, hidden]."""
    n = x.shape[0]
    src, dst, w = with_self_loops(n, edge_index, edge_weight)
    h = np.maximum(bn_eval(gat_conv(x.astype(f32), src, dst, p, heads), p, 1), 0).astype(f32)
    for k in (2, 3, 4):
        h = np.maximum(bn_eval(gcn_conv(h, src, dst, w, p, k), p, k), 0).astype(f32)
    return np.maximum(h @ p["fc_shared.weight"].T + p["fc_shared.bias"], 0).astype(f32)
