"""MI355X-native drop-in for the reference's ``models/multimodal/fusion_model.py``.

Same public surface (reference file:line in brackets):

* ``build_multimodal_model(config)``                                   [:249-259]
* ``MultimodalCamouflageDetector(rg_dim, kg_dim, hidden_dim, num_heads,
  fusion_type, num_classes, dropout)``                                  [:176-183]
* ``.forward(rg_embeddings, kg_embeddings, return_attention=False)`` ->
  ``(mask [B,C], instance [B,C], edge [B,1], score [B,1])`` plus, on request,
  ``{'rg2kg': [B,Nr,Nk], 'kg2rg': [B,Nk,Nr]}`` (``None`` for late fusion)  [:237-246]
* ``state_dict()`` names and shapes of the reference module (44 tensors for
  cross-attention), so reference-trained checkpoints load with ``strict=True``
* ``ValueError`` for an unknown ``fusion_type`` [:206] and for inputs that are
  not 2-/3-/4-D [:102].

The sub-modules below exist to own parameters under the reference's names and
to initialise them like torch does for the reference; none of them computes
anything.  All arithmetic happens in ``libcamo_fusion.so`` (hand-written HIP for
gfx950) through :mod:`camouflage_multimodal_amd.engine`.  There is no CPU path.

Extension over the reference: :meth:`MultimodalCamouflageDetector.forward_packed`
takes a variable-Nr minibatch as one packed ``[sum(Nr), rg_dim]`` matrix, which is
what the native trainer uses instead of the reference's per-sample Python loop.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import engine as _engine


class _AttentionParams(nn.Module):
    """Parameter holder named like nn.MultiheadAttention (in_proj_weight packs
    Wq|Wk|Wv as rows 0..E-1 | E..2E-1 | 2E..3E-1) with torch's initialisation:
    xavier-uniform in-projection, zero biases."""

    def __init__(self, embed_dim):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


def _mlp(d_in, d_hidden, d_out, dropout):
    # indices 0 and 3 carry the parameters, as in the reference's nn.Sequential stacks
    return nn.Sequential(nn.Linear(d_in, d_hidden), nn.ReLU(), nn.Dropout(dropout), nn.Linear(d_hidden, d_out))


class CrossAttentionFusion(nn.Module):
    """Parameters of the reference class of the same name [fusion_model.py:16-73]."""

    def __init__(self, rg_dim=128, kg_dim=128, hidden_dim=256, num_heads=8, dropout=0.3):
        super().__init__()
        if hidden_dim % num_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.rg_dim, self.kg_dim, self.hidden_dim = rg_dim, kg_dim, hidden_dim
        self.rg_proj = nn.Linear(rg_dim, hidden_dim) if rg_dim != hidden_dim else nn.Identity()
        self.kg_proj = nn.Linear(kg_dim, hidden_dim) if kg_dim != hidden_dim else nn.Identity()
        self.cross_attn_rg2kg = _AttentionParams(hidden_dim)
        self.cross_attn_kg2rg = _AttentionParams(hidden_dim)
        self.ln_rg = nn.LayerNorm(hidden_dim)
        self.ln_kg = nn.LayerNorm(hidden_dim)
        self.ffn_rg = _mlp(hidden_dim, hidden_dim * 2, hidden_dim, dropout)
        self.ffn_kg = _mlp(hidden_dim, hidden_dim * 2, hidden_dim, dropout)
        self.fusion_layer = _mlp(hidden_dim * 2, hidden_dim, hidden_dim, dropout)


class LateFusion(nn.Module):
    """Parameters of the reference class of the same name [fusion_model.py:149-162]."""

    def __init__(self, rg_dim=128, kg_dim=128, hidden_dim=256, dropout=0.3):
        super().__init__()
        self.fusion = nn.Sequential(
            nn.Linear(rg_dim + kg_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(hidden_dim // 2, hidden_dim // 2))


def _collapse_to_3d(t, name):
    """Input normalisation of the reference forward [fusion_model.py:86-105]."""
    if t.dim() == 2:
        t = t.unsqueeze(1)
    if t.dim() == 3:
        return t
    if t.dim() == 4:
        B, a, b, d = t.shape
        if a == 1:
            return t.squeeze(1)
        if b == 1:
            return t.squeeze(2)
        return t.reshape(B, a * b, d)
    raise ValueError(f"{name} must be 2D/3D/4D tensor, got shape {t.shape}")


class MultimodalCamouflageDetector(nn.Module):
    def __init__(self, rg_dim=128, kg_dim=128, hidden_dim=256, num_heads=8, fusion_type="cross_attention",
                 num_classes=2, dropout=0.3):
        super().__init__()
        self.fusion_type = fusion_type
        if fusion_type == "cross_attention":
            self.fusion = CrossAttentionFusion(rg_dim, kg_dim, hidden_dim, num_heads, dropout)
            final_dim = hidden_dim
        elif fusion_type == "late":
            self.fusion = LateFusion(rg_dim, kg_dim, hidden_dim, dropout)
            final_dim = hidden_dim // 2
        else:
            raise ValueError(f"Unknown fusion_type: {fusion_type}")
        self.mask_head = _mlp(final_dim, final_dim // 2, num_classes, dropout)
        self.instance_head = _mlp(final_dim, final_dim // 2, num_classes, dropout)
        self.edge_head = _mlp(final_dim, final_dim // 2, 1, dropout)
        self.score_head = _mlp(final_dim, final_dim // 2, 1, dropout)   # + Sigmoid, applied by the kernel
        self.config = dict(rg_dim=rg_dim, kg_dim=kg_dim, hidden_dim=hidden_dim, num_heads=num_heads,
                           fusion_type=fusion_type, num_classes=num_classes, dropout=float(dropout))
        #: "f32" (default: exact fp32 MFMA -- the reference is fp32 throughout, so an import swap keeps its numerics to
        #: ~2e-5 on the logits) or "bf16" (bf16 MFMA operands, fp32 accumulate: logits within 1e-3; what bench.py times).
        #: Opt in with ``model.set_precision("bf16")`` or the ``precision`` key of the trainer config.
        self.precision = "f32"
        self._engine = _engine.FusionEngine(self)

    # copy.deepcopy / pickle (EMA or best-model snapshots, torch.save(model)): the engine holds a weak reference to ITS
    # module, a ctypes pointer table and the flat buffers -- none of which may be shared with a copy.  The copy gets
    # its parameters by value and an engine of its own.
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._engine = _engine.FusionEngine(self)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_engine" else copy.deepcopy(v, memo)
        new._engine = _engine.FusionEngine(new)
        return new

    # nn.Module plumbing: keep the flat parameter buffer coherent across .to()/.cuda()/.float()
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        if getattr(self, "_engine", None) is not None:
            self._engine.reflatten()
        return self

    def set_precision(self, precision: str):
        if precision not in ("f32", "bf16"):
            raise ValueError("precision must be 'f32' or 'bf16'")
        self.precision = precision
        return self

    def forward(self, rg_embeddings, kg_embeddings, return_attention=False):
        rg = _collapse_to_3d(rg_embeddings, "rg_embeddings")
        kg = _collapse_to_3d(kg_embeddings, "kg_embeddings")
        if rg.shape[0] != kg.shape[0]:
            raise RuntimeError(f"batch sizes differ: rg {tuple(rg.shape)} vs kg {tuple(kg.shape)}")
        B, Nr, _ = rg.shape
        outs, attn = self._engine.forward_autograd(rg.reshape(B * Nr, rg.shape[2]), [Nr] * B, kg,
                                                   want_attention=return_attention)
        res = self._split(outs)
        if not return_attention:
            return res
        if attn is None:
            return res + (None,)
        Nk = kg.shape[1]
        return res + ({"rg2kg": attn[0].view(B, Nr, Nk), "kg2rg": attn[1].view(B, Nr, Nk).transpose(1, 2)},)

    def forward_packed(self, rg_packed, nr_per_sample, kg_embeddings, return_attention=False):
        """Variable-Nr minibatch: ``rg_packed`` [sum(Nr), rg_dim] holds the samples' node rows back
        to back, ``nr_per_sample`` their lengths (host ints), ``kg_embeddings`` [B, Nk, kg_dim].
        Attention maps come back as per-sample lists."""
        kg = _collapse_to_3d(kg_embeddings, "kg_embeddings")
        outs, attn = self._engine.forward_autograd(rg_packed, list(nr_per_sample), kg, want_attention=return_attention)
        res = self._split(outs)
        if not return_attention:
            return res
        if attn is None:
            return res + (None,)
        Nk = kg.shape[1]
        a1 = list(torch.split(attn[0], list(nr_per_sample), dim=0))
        a2 = [a.t() for a in torch.split(attn[1], list(nr_per_sample), dim=0)]
        return res + ({"rg2kg": a1, "kg2rg": a2},)

    def _split(self, outs):
        C = self.config["num_classes"]
        return outs[:, :C], outs[:, C:2 * C], outs[:, 2 * C:2 * C + 1], outs[:, 2 * C + 1:]


def build_multimodal_model(config):
    return MultimodalCamouflageDetector(
        rg_dim=config.get("rg_dim", 128),
        kg_dim=config.get("kg_dim", 128),
        hidden_dim=config.get("hidden_dim", 256),
        num_heads=config.get("num_heads", 8),
        fusion_type=config.get("fusion_type", "cross_attention"),
        num_classes=config.get("num_classes", 2),
        dropout=config.get("dropout", 0.3),
    )
