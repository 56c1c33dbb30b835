"""Region-Graph construction and GNN embedding path on MI355X.

``RegionGraphGNN`` keeps the reference module's name, constructor and ``state_dict`` keys
(models/region_graph/extract_rg_embeddings.py:27-52), so ``best_model.pth`` loads with ``strict=True``
(test_multimodal.py:420-423); ``extract_node_embeddings(data)`` (:94-122) -- the call that feeds the fusion model at
inference -- runs as HIP kernels behind ``camo_rg_node_embeddings`` (include/camo_rg_gnn.h).  The graph layers are
torch_geometric's in the reference; the published algorithms they are restated from and the CPU checker the kernels are
tested against are named in include/camo_rg_gnn.h (PARITY UNPINNED: no PyG here, no RG weights or fixtures shipped).  The node-classification ``forward`` (training of the RG model,
models/region_graph/train.py) is outside the path and raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .engine import _ptr, _stream_ptr


def build_target_csr_device(num_nodes, edge_index, edge_weight=None):
    """The same CSR as ``build_target_csr`` built by the library (``camo_rg_build_csr``: counting sort by target, four
    small launches instead of a chain of torch index kernels); a row's edges after its leading self-loop come in no
    particular order.  Returns (rowptr int32 [N+1], col int32 [E+N], w fp32 [E+N])."""
    _lib.require_device(edge_index, "edge_index")
    dev = edge_index.device
    ei = edge_index.to(torch.int64).contiguous()
    E = ei.shape[1]
    ew = None if edge_weight is None else edge_weight.reshape(-1).to(torch.float32).contiguous()
    scratch = torch.empty(3 * num_nodes, dtype=torch.int32, device=dev)
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(E + num_nodes, dtype=torch.int32, device=dev)
    w = torch.empty(E + num_nodes, dtype=torch.float32, device=dev)
    rc = _lib.lib().camo_rg_build_csr(_ptr(ei), _ptr(ew), num_nodes, E, _ptr(scratch), _ptr(rowptr), _ptr(col), _ptr(w), _stream_ptr())
    _lib.check(rc, "camo_rg_build_csr")
    return rowptr, col, w


def build_target_csr(num_nodes, edge_index, edge_weight=None):
    """edge_index [2, E] (row 0 = source j, row 1 = target i, PyG convention), edge_weight [E] or None ->
    (rowptr int32 [N+1], col int32 [E'], w fp32 [E']) sorted by target with exactly one self-loop per node: existing
    self-loops keep their weight, missing ones get weight 1 (PyG ``add_remaining_self_loops``; GATConv's
    remove-then-add gives the same structure and ignores weights).  Index plumbing on the tensors' device."""
    dev = edge_index.device
    src, dst = edge_index[0].long(), edge_index[1].long()
    w = torch.ones(src.shape[0], dtype=torch.float32, device=dev) if edge_weight is None else edge_weight.reshape(-1).to(torch.float32)
    loop = src == dst
    lw = torch.ones(num_nodes, dtype=torch.float32, device=dev)
    lw[src[loop]] = w[loop]
    ar = torch.arange(num_nodes, device=dev)
    src = torch.cat([src[~loop], ar]); dst = torch.cat([dst[~loop], ar]); w = torch.cat([w[~loop], lw])
    order = torch.argsort(dst * num_nodes + src, stable=True)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=num_nodes), 0)
    return rowptr.to(torch.int32), src[order].to(torch.int32).contiguous(), w[order].contiguous()


class RegionGraphData:
    """What the reference's ``create_region_graph`` returns as a torch_geometric ``Data`` (extract_rg_embeddings.py:239-243):
    ``x`` [n, 15], ``edge_index`` [2, E] int64, ``edge_attr`` [E, 1]; duck-typed for ``extract_node_embeddings``."""

    def __init__(self, x, edge_index, edge_attr):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr

    def to(self, device):
        return RegionGraphData(self.x.to(device), self.edge_index.to(device), self.edge_attr.to(device))

    cpu = lambda self: self.to("cpu")  # noqa: E731


def create_region_graph_from_segments(image, segments, edges_canny, device="cuda", edge_capacity=None):
    """The body of ``create_region_graph`` (extract_rg_embeddings.py:146-246) between its skimage calls, on the device
    (``camo_rg_region_graph``, include/camo_rg_features.h): ``image`` [H, W, 3] float in [0, 1], ``segments`` [H, W] integer
    superpixel labels (what ``slic`` returned, :144), ``edges_canny`` [H, W] bool (what ``canny`` returned, :152) ->
    (RegionGraphData on the device, region_map int32 [labels] = new index of each label or -1).  Regions are renumbered in
    increasing label order with empty labels dropped; edges come sorted by (i, j) with each followed by its reverse (the
    reference's order is networkx's iteration order: a permutation).  PARITY UNPINNED, see the header."""
    dev = torch.device(device)
    img = torch.as_tensor(image).to(device=dev, dtype=torch.float32).contiguous()
    seg = torch.as_tensor(segments).to(device=dev, dtype=torch.int32).contiguous()
    can = torch.as_tensor(edges_canny).to(device=dev).to(torch.uint8).contiguous()
    _lib.require_device(img, "image")
    if img.dim() != 3 or img.shape[2] != 3 or seg.shape != img.shape[:2] or can.shape != img.shape[:2]:
        raise ValueError(f"need image [H, W, 3], segments [H, W], edges_canny [H, W]; got {tuple(img.shape)}, {tuple(seg.shape)}, {tuple(can.shape)}")
    H, W = seg.shape
    lo, hi = int(seg.min()), int(seg.max())
    if lo < 0 or hi >= _lib.RG_MAX_LABELS:
        raise ValueError(f"segment labels must lie in [0, {_lib.RG_MAX_LABELS}), got [{lo}, {hi}]")
    n_labels = hi + 1
    cap = int(edge_capacity) if edge_capacity else 16 * n_labels        # (a planar adjacency has < 3 n pairs; 8-connectivity adds corner contacts)
    L = _lib.lib()
    while True:
        ws = torch.empty(L.camo_rg_graph_workspace_bytes(n_labels), dtype=torch.uint8, device=dev)
        x = torch.empty(n_labels, 15, dtype=torch.float32, device=dev)
        rmap = torch.empty(n_labels, dtype=torch.int32, device=dev)
        ei = torch.empty(2, cap, dtype=torch.int64, device=dev)
        ea = torch.empty(cap, dtype=torch.float32, device=dev)
        counts = torch.zeros(2, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = L.camo_rg_region_graph(_ptr(img), _ptr(seg), _ptr(can), H, W, n_labels, _ptr(ws), ws.numel(), _ptr(x), _ptr(rmap),
                                        _ptr(ei), _ptr(ea), cap, _ptr(counts), _stream_ptr(dev))
        _lib.check(rc, "camo_rg_region_graph")
        n, e = (int(v) for v in counts.tolist())                         # (the one synchronisation: the sizes of what was built)
        if e <= cap:
            break
        cap = e                                                          # the library reported the capacity needed: once more
    return RegionGraphData(x[:n], ei[:, :e], ea[:e].unsqueeze(1)), rmap


def create_region_graph(image, n_segments=500, device="cuda"):
    """``create_region_graph(image, n_segments)`` of the reference (extract_rg_embeddings.py:138): slic and canny are
    skimage's and stay on the host when skimage is installed; everything after them runs on the device.  Returns
    (RegionGraphData, segments)."""
    try:
        from skimage import feature
        from skimage.segmentation import slic
    except ImportError as err:
        raise _lib.CamoError("create_region_graph needs scikit-image for slic / canny (models/region_graph/extract_rg_embeddings.py:144,152); "
                             "pass their results to create_region_graph_from_segments instead") from err
    import numpy as np
    image = np.asarray(image)
    segments = slic((image * 255).astype(np.uint8), n_segments=n_segments, compactness=10, sigma=1)       # :143-144
    gray = np.dot(image[..., :3], [0.2989, 0.5870, 0.1140])                                               # :151
    data, _ = create_region_graph_from_segments(image, segments, feature.canny(gray, sigma=2), device)  # :152
    return data, segments


class _GATParams(nn.Module):
    """Parameter container with torch_geometric.nn.GATConv's state_dict names (``lin.weight``; ``lin_src.weight`` of
    older releases is accepted on load)."""

    def __init__(self, in_channels, out_channels, heads):
        super().__init__()
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.att_src); nn.init.xavier_uniform_(self.att_dst); nn.init.xavier_uniform_(self.lin.weight)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for old in ("lin_src.weight", "lin_l.weight"):
            if prefix + old in state_dict and prefix + "lin.weight" not in state_dict:
                state_dict[prefix + "lin.weight"] = state_dict.pop(prefix + old)
        for dup in ("lin_dst.weight", "lin_r.weight"):
            state_dict.pop(prefix + dup, None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class _GCNParams(nn.Module):
    """Parameter container with torch_geometric.nn.GCNConv's state_dict names."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.lin.weight)


class RegionGraphGNN(nn.Module):
    def __init__(self, in_channels=15, hidden_channels=128, num_classes=2, heads=4):
        super().__init__()
        h = hidden_channels
        self.conv1 = _GATParams(in_channels, h, heads)
        self.bn1 = nn.BatchNorm1d(h)
        self.conv2 = _GCNParams(h, h); self.bn2 = nn.BatchNorm1d(h)
        self.conv3 = _GCNParams(h, h); self.bn3 = nn.BatchNorm1d(h)
        self.conv4 = _GCNParams(h, h); self.bn4 = nn.BatchNorm1d(h)
        self.fc_shared = nn.Linear(h, h)
        # node-classification heads: parameters kept so that the reference checkpoint loads strictly; not on the path
        self.fc_mask_1 = nn.Linear(h, h // 2); self.fc_mask_2 = nn.Linear(h // 2, num_classes)
        self.fc_instance_1 = nn.Linear(h, h // 2); self.fc_instance_2 = nn.Linear(h // 2, num_classes)
        self.fc_edge_1 = nn.Linear(h, h // 2); self.fc_edge_2 = nn.Linear(h // 2, 1)
        self._dims = _lib.CamoRgDims(in_channels, h, heads)

    def _param_table(self):
        t = [self.conv1.att_src, self.conv1.att_dst, self.conv1.bias, self.conv1.lin.weight,
             self.bn1.weight, self.bn1.bias, self.bn1.running_mean, self.bn1.running_var]
        for conv, bn in ((self.conv2, self.bn2), (self.conv3, self.bn3), (self.conv4, self.bn4)):
            t += [conv.bias, conv.lin.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        t += [self.fc_shared.weight, self.fc_shared.bias]
        assert len(t) == _lib.RG_NPARAMS
        for p in t:
            _lib.require_device(p, "RegionGraphGNN parameters")
        keep = [p.detach().to(torch.float32).contiguous() for p in t]
        tab = (C.c_void_p * len(keep))(*[p.data_ptr() for p in keep])
        return tab, keep

    @torch.no_grad()
    def extract_node_embeddings(self, data=None, x=None, edge_index=None, edge_attr=None):
        """[num_nodes, hidden] node embeddings (eval-mode BatchNorm, no dropout), extract_rg_embeddings.py:94-122.
        ``data``: any object with ``x``, ``edge_index``, ``edge_attr`` (a torch_geometric ``Data`` / ``Batch``)."""
        if data is not None:
            x, edge_index = data.x, data.edge_index
            edge_attr = getattr(data, "edge_attr", None)
        _lib.require_device(x, "x")
        _lib.require_device(edge_index, "edge_index")
        if x.dim() != 2 or x.shape[1] != self._dims.in_channels:
            raise RuntimeError(f"x of shape {tuple(x.shape)} does not match in_channels {self._dims.in_channels}")
        n = x.shape[0]
        ew = None if edge_attr is None or edge_attr.numel() == 0 else edge_attr.reshape(-1)     # :98
        rowptr, col, w = build_target_csr_device(n, edge_index, ew)
        x = x.detach().to(torch.float32).contiguous()
        L = _lib.lib()
        need = L.camo_rg_workspace_bytes(C.byref(self._dims), n)
        if need == 0:
            _lib.check(-1, "camo_rg_workspace_bytes")
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        out = torch.empty(n, self._dims.hidden, dtype=torch.float32, device=x.device)
        tab, keep = self._param_table()
        rc = L.camo_rg_node_embeddings(C.byref(self._dims), tab, _ptr(x), _ptr(rowptr), _ptr(col), _ptr(w), n, col.shape[0],
                                       _ptr(ws), ws.numel(), _ptr(out), _stream_ptr())
        _lib.check(rc, "camo_rg_node_embeddings")
        return out

    def extract_graph_embedding(self, data):
        """[num_graphs, hidden]: mean of the node embeddings per graph (global_mean_pool, extract_rg_embeddings.py:124-135).
        Not an input of the fusion model (the reference only stores it next to the node embeddings); the pooling is index
        plumbing on the device."""
        emb = self.extract_node_embeddings(data)
        batch = getattr(data, "batch", None)
        if batch is None:
            return emb.mean(dim=0, keepdim=True)
        g = int(batch.max().item()) + 1
        out = torch.zeros(g, emb.shape[1], dtype=emb.dtype, device=emb.device).index_add_(0, batch.long(), emb)
        return out / torch.bincount(batch.long(), minlength=g).clamp(min=1).unsqueeze(1).to(emb.dtype)

    def forward(self, data):
        raise _lib.CamoError("RegionGraphGNN.forward (node-classification heads, used only to train the RG model) is outside "
                             "the MI355X path; use extract_node_embeddings()")
