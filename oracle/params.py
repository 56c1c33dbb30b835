"""Parameter table and seeded synthetic data shared by the oracle, the golden
generator and the tests.  TEST INFRASTRUCTURE (see oracle/__init__.py).

The table restates the reference module's ``state_dict`` (names, shapes, order)
as it results from /root/reference/models/multimodal/fusion_model.py:21-73
(CrossAttentionFusion), :151-162 (LateFusion) and :208-235 (the four heads).
``tests/golden/make_golden.py`` loads a dict built from this table into the
reference module with ``strict=True``, which is what pins the table.
"""
from __future__ import annotations

import numpy as np

DEFAULT_CFG = dict(rg_dim=128, kg_dim=128, hidden_dim=256, num_heads=8,
                   fusion_type="cross_attention", num_classes=2, dropout=0.3)


def full_cfg(cfg=None):
    out = dict(DEFAULT_CFG)
    if cfg:
        out.update(cfg)
    return out


def param_specs(cfg=None):
    """Ordered [(name, shape)] of the model's state_dict for ``cfg``."""
    c = full_cfg(cfg)
    D, Dk, H, C = c["rg_dim"], c["kg_dim"], c["hidden_dim"], c["num_classes"]
    s = []
    if c["fusion_type"] == "cross_attention":
        if D != H:   # nn.Identity otherwise (fusion_model.py:29)
            s += [("fusion.rg_proj.weight", (H, D)), ("fusion.rg_proj.bias", (H,))]
        if Dk != H:  # fusion_model.py:30
            s += [("fusion.kg_proj.weight", (H, Dk)), ("fusion.kg_proj.bias", (H,))]
        for a in ("cross_attn_rg2kg", "cross_attn_kg2rg"):
            s += [(f"fusion.{a}.in_proj_weight", (3 * H, H)),
                  (f"fusion.{a}.in_proj_bias", (3 * H,)),
                  (f"fusion.{a}.out_proj.weight", (H, H)),
                  (f"fusion.{a}.out_proj.bias", (H,))]
        for n in ("ln_rg", "ln_kg"):
            s += [(f"fusion.{n}.weight", (H,)), (f"fusion.{n}.bias", (H,))]
        for n in ("ffn_rg", "ffn_kg"):
            s += [(f"fusion.{n}.0.weight", (2 * H, H)), (f"fusion.{n}.0.bias", (2 * H,)),
                  (f"fusion.{n}.3.weight", (H, 2 * H)), (f"fusion.{n}.3.bias", (H,))]
        s += [("fusion.fusion_layer.0.weight", (H, 2 * H)), ("fusion.fusion_layer.0.bias", (H,)),
              ("fusion.fusion_layer.3.weight", (H, H)), ("fusion.fusion_layer.3.bias", (H,))]
        F = H
    elif c["fusion_type"] == "late":
        s += [("fusion.fusion.0.weight", (H, D + Dk)), ("fusion.fusion.0.bias", (H,)),
              ("fusion.fusion.3.weight", (H // 2, H)), ("fusion.fusion.3.bias", (H // 2,)),
              ("fusion.fusion.6.weight", (H // 2, H // 2)), ("fusion.fusion.6.bias", (H // 2,))]
        F = H // 2
    else:
        raise ValueError(f"Unknown fusion_type: {c['fusion_type']}")
    for head, n_out in (("mask_head", C), ("instance_head", C), ("edge_head", 1), ("score_head", 1)):
        s += [(f"{head}.0.weight", (F // 2, F)), (f"{head}.0.bias", (F // 2,)),
              (f"{head}.3.weight", (n_out, F // 2)), (f"{head}.3.bias", (n_out,))]
    return s


def make_params(cfg=None, seed=0):
    """Seeded parameters.  Every bias and LayerNorm affine term is non-trivial
    on purpose (torch's default zeros/ones would hide bias and affine bugs)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in param_specs(cfg):
        if name.endswith("weight") and len(shape) == 2:
            a = 1.0 / np.sqrt(shape[1])
            v = rs.uniform(-a, a, size=shape)
        elif ".ln_" in name and name.endswith("weight"):
            v = 1.0 + rs.uniform(-0.2, 0.2, size=shape)
        else:
            v = rs.uniform(-0.1, 0.1, size=shape)
        out[name] = v.astype(np.float32)
    return out


def make_rg(nr, dim=128, seed=0, kind="relu"):
    """Synthetic region-graph node embeddings [nr, dim].  ``relu``: |N(0,1)|*0.3
    (RG embeddings are post-ReLU); ``randn``: the reference self-test's
    plain N(0,1) (fusion_model.py:273)."""
    rs = np.random.RandomState(1000 + seed)
    x = rs.standard_normal((nr, dim))
    if kind == "relu":
        x = np.abs(x) * 0.3
    return x.astype(np.float32)


def make_kg(nk, dim=128, seed=0):
    rs = np.random.RandomState(2000 + seed)
    return (np.abs(rs.standard_normal((nk, dim))) * 0.3).astype(np.float32)


def make_labels(b, seed=0):
    """(mask_label int64 [b], edge_label f32 [b], score_label f32 [b]) as the
    reference dataset would supply them (train_multimodal.py:183-186)."""
    rs = np.random.RandomState(3000 + seed)
    y = (rs.uniform(size=b) < 0.5).astype(np.int64)
    e = (rs.uniform(size=b) < 0.5).astype(np.float32)
    s = rs.uniform(size=b).astype(np.float32)
    return y, e, s
