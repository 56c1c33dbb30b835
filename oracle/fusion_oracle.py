"""Numpy restatement of the reference fusion hot path.  TEST INFRASTRUCTURE
(see oracle/__init__.py): forward, the four-term loss, backward, global-norm
clip and AdamW, all in float32, one sample at a time like the reference's
training loop.

Each function cites the reference lines it follows (paths relative to
/root/reference).  torch's nn.MultiheadAttention / LayerNorm / AdamW are third
party to the reference (torch 2.x); their documented arithmetic is restated
here and pinned through the reference's own call sites by the golden vectors.

Dropout: the reference draws masks from torch's global RNG, which cannot be
reproduced off-box.  The oracle and the HIP kernels instead share one
counter-based hash (``dropout_keep``) keyed by (seed, site, element index in
the packed batch layout), so train-mode runs with dropout>0 are comparable
element for element between the two.  Eval mode and dropout=0 train mode are
the cases pinned against the reference.
"""
from __future__ import annotations

import numpy as np

from .params import full_cfg

f32 = np.float32

# dropout sites (same ids in csrc/common.h)
SITE_ATTN_RG2KG, SITE_ATTN_KG2RG, SITE_FFN_RG, SITE_FFN_KG, SITE_FUSE = 1, 2, 3, 4, 5
SITE_HEAD0 = 6  # +0 mask, +1 instance, +2 edge, +3 score
SITE_LATE0 = 10  # +0, +1: the two hidden layers of LateFusion


def bf16_round(x):
    """float32 -> bfloat16 -> float32, round to nearest even (what v_cvt_pk_bf16_f32 / a cast to __bf16 does)."""
    x = np.ascontiguousarray(x, dtype=f32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)
    return (u.astype(np.uint32) << np.uint32(16)).view(f32).reshape(x.shape)


def _fmix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def dropout_keep(seed, site, idx, p):
    """Boolean keep-mask for element indices ``idx`` (any shape, < 2**32).
    u = top 24 bits of a murmur3 finaliser over (idx + site constant + seed_lo) ^ seed_hi; keep iff u >= p."""
    idx = np.asarray(idx, dtype=np.uint64)
    lo = np.uint64(seed & 0xFFFFFFFF)
    hi = np.uint64((seed >> 32) & 0xFFFFFFFF)
    x = (idx + ((np.uint64(site) * np.uint64(0x85EBCA77) + lo) & np.uint64(0xFFFFFFFF))) & np.uint64(0xFFFFFFFF)
    x = _fmix32(x ^ hi)
    u = (x >> np.uint64(8)).astype(np.float32) * f32(1.0 / 16777216.0)
    return u >= f32(p)


def _drop(x, seed, site, base_idx, p, training):
    """Apply inverted dropout to x whose element (i...) has packed linear index
    base_idx + ravel position.  Returns (y, mult) with y = x*mult."""
    if not training or p <= 0.0:
        return x, None
    idx = base_idx + np.arange(x.size, dtype=np.uint64).reshape(x.shape)
    mult = dropout_keep(seed, site, idx, p).astype(f32) * f32(1.0 / (1.0 - p))
    return x * mult, mult


def _linear(x, w, b):
    return x @ w.T + b


def _layernorm(u, g, b, eps=1e-5):
    mu = u.mean(axis=-1, keepdims=True, dtype=f32)
    xc = u - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=f32)
    rstd = f32(1.0) / np.sqrt(var + f32(eps))
    xh = xc * rstd
    return xh * g + b, xh, rstd


def _layernorm_bwd(dy, xh, rstd, g):
    dxh = dy * g
    n = f32(dy.shape[-1])
    dx = (dxh - dxh.mean(axis=-1, keepdims=True, dtype=f32)
          - xh * (dxh * xh).mean(axis=-1, keepdims=True, dtype=f32)) * rstd
    return dx, (dy * xh).sum(axis=0, dtype=f32), dy.sum(axis=0, dtype=f32)


def _softmax(s, axis):
    m = s.max(axis=axis, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(axis=axis, keepdims=True, dtype=f32)


def _sigmoid(x):
    return f32(1.0) / (f32(1.0) + np.exp(-x))


def collapse_inputs(rg, kg):
    """Input normalisation of fusion_model.py:86-105 on numpy arrays."""
    def to3(t, name):
        if t.ndim == 2:
            t = t[:, None, :]
        if t.ndim == 3:
            return t
        if t.ndim == 4:
            B, a, b, d = t.shape
            return t.reshape(B, a * b, d)   # squeeze(1)/squeeze(2)/view all equal this reshape
        raise ValueError(f"{name} must be 2D/3D/4D tensor, got shape {t.shape}")
    return to3(rg, "rg_embeddings"), to3(kg, "kg_embeddings")


class FusionOracle:
    """MultimodalCamouflageDetector (fusion_model.py:174-246) in numpy."""

    def __init__(self, cfg, params, bf16_operands=False):
        """``bf16_operands``: restate the bf16 mode of the HIP path instead of the reference's fp32 arithmetic -- every tensor the
        fused kernels hand to an MFMA as a bf16 operand (inputs, node-level weights, R / G, the pre-scaled
        queries, keys, values, attention probabilities, attention outputs, LayerNorm outputs; in backward the FFN gradient rows,
        dU, dO, dS, the dropped probabilities, dQ / dK / dV, dR / dG and the saved normalised LayerNorm input) is rounded to
        bf16 where the kernels round it; sums stay fp32, the per-sample tail stays fp32.  The f32 mode is the one pinned to the
        reference's golden vectors; this mode differs from it only through ``self.q`` (identity when off)."""
        self.cfg = full_cfg(cfg)
        self.p = {k: np.asarray(v, dtype=f32) for k, v in params.items()}
        self.bf16 = bool(bf16_operands)
        self.q = bf16_round if self.bf16 else (lambda x: x)
        # Per-sample tail units (fusion layer 0, the heads' hidden layers) whose pre-activation sits within ``near_eps`` of the ReLU
        # threshold are recorded in ``near`` as (site, sample, unit, value) when it is a list: an implementation that sums in
        # another order may land on the other side of zero there, and ONE such unit moves a head's gradient by percent.
        # ``relu_flip``: set of (site, sample, unit) whose ReLU decision is inverted -- tests enumerate the admissible sign
        # patterns of the recorded units instead of picking inputs that happen to have none (tests/helpers.py).
        self.kg_block = 64                  # keys per flash block of the KG->RG attention in the fused kernels (bf16 mode only)
        # sample index -> keys in its FIRST block where that differs from kg_block: the 64-row forward (csrc/fused_wide2.hip) cuts blocks
        # from the batch's global table of 32-row tiles, two per block, so a sample that starts on an odd tile has a 32-key first block
        self.kg_first_block = {}
        self.near = None
        self.near_eps = 2e-4
        self.relu_flip = frozenset()

    def _relu_tail(self, x, site, b_index):
        pos = x > 0
        if self.near is not None:
            for u in np.nonzero(np.abs(x[0]) < self.near_eps)[0]:
                self.near.append((site, b_index, int(u), float(x[0, u])))
        for (s_, b_, u) in self.relu_flip:
            if s_ == site and b_ == b_index:
                pos[0, u] = not pos[0, u]
        return np.where(pos, x, f32(0)), pos
        if self.cfg["fusion_type"] not in ("cross_attention", "late"):
            raise ValueError(f"Unknown fusion_type: {self.cfg['fusion_type']}")

    # ------------------------------------------------------------------ fwd
    def forward_sample(self, rg, kg, training=False, seed=0, row_base=0, b_index=0, kg_row_base=None):
        """One sample: rg [Nr, rg_dim], kg [Nk, kg_dim] -> (outs, cache).

        ``row_base`` / ``b_index`` / ``kg_row_base`` give this sample's position in
        the packed batch (first RG row, sample number, first KG row); they only
        enter the dropout element indices."""
        if self.cfg["fusion_type"] == "late":
            return self._forward_late(rg, kg, training, seed, b_index)
        P, c = self.p, self.cfg
        H, nh = c["hidden_dim"], c["num_heads"]
        dh = H // nh
        pd = float(c["dropout"])
        Nr, Nk = rg.shape[0], kg.shape[0]
        if kg_row_base is None:
            kg_row_base = b_index * Nk
        scale = f32(1.0 / np.sqrt(dh))
        ca = {}
        q = self.q                          # (bf16 mode: operand rounding; identity in the pinned f32 mode)
        W = (lambda name: q(P[name]))      # node-level weights are MFMA operands
        rg, kg = q(rg), q(kg)
        # fusion_model.py:108-109  (bf16 mode: R / G exist only as bf16 -- operand of the in-projections AND the residual)
        R = q(_linear(rg, W("fusion.rg_proj.weight"), P["fusion.rg_proj.bias"])) if "fusion.rg_proj.weight" in P else rg
        G = q(_linear(kg, W("fusion.kg_proj.weight"), P["fusion.kg_proj.bias"])) if "fusion.kg_proj.weight" in P else kg
        # fusion_model.py:112-118 -- MHA, query=rg, key=value=kg.  (bf16 mode: the queries are scaled BEFORE they are rounded,
        # as the kernels store them; Q here stays the unscaled fp32 value of the reference, Qs is what the scores see)
        Wi, bi = W("fusion.cross_attn_rg2kg.in_proj_weight"), P["fusion.cross_attn_rg2kg.in_proj_bias"]
        Q = _linear(R, Wi[:H], bi[:H]); Kk = q(_linear(G, Wi[H:2 * H], bi[H:2 * H])); Vk = q(_linear(G, Wi[2 * H:], bi[2 * H:]))
        Qs = q(Q * scale)
        Qh = Qs.reshape(Nr, nh, dh); Kh = Kk.reshape(Nk, nh, dh); Vh = Vk.reshape(Nk, nh, dh)
        S = np.einsum("thd,jhd->thj", Qh, Kh).astype(f32)                     # [Nr, nh, Nk]
        if not self.bf16:                                                     # (the pinned mode keeps the reference's order: scale after the product)
            S = np.einsum("thd,jhd->thj", Q.reshape(Nr, nh, dh), Kh).astype(f32) * scale
        Pm = _softmax(S, axis=2)
        Pd, m_a1 = _drop(Pm, seed, SITE_ATTN_RG2KG, row_base * nh * Nk, pd, training)
        O = q(np.einsum("thj,jhd->thd", q(Pd), Vh).astype(f32).reshape(Nr, H))
        A = _linear(O, W("fusion.cross_attn_rg2kg.out_proj.weight"), P["fusion.cross_attn_rg2kg.out_proj.bias"])
        # fusion_model.py:119-120  (the mean pool of Z uses the fp32 LayerNorm output; the first FFN layer its bf16 copy)
        Y, xh1, rstd1 = _layernorm(R + A, P["fusion.ln_rg.weight"], P["fusion.ln_rg.bias"])
        H1 = np.maximum(_linear(q(Y), W("fusion.ffn_rg.0.weight"), P["fusion.ffn_rg.0.bias"]), 0)
        H1d, m_f1 = _drop(H1, seed, SITE_FFN_RG, row_base * 2 * H, pd, training)
        Z = Y + _linear(H1d, P["fusion.ffn_rg.3.weight"], P["fusion.ffn_rg.3.bias"])
        # fusion_model.py:123-129 -- MHA, query=kg, key=value=rg_proj
        Wi2, bi2 = W("fusion.cross_attn_kg2rg.in_proj_weight"), P["fusion.cross_attn_kg2rg.in_proj_bias"]
        Q2 = _linear(G, Wi2[:H], bi2[:H]); K2 = q(_linear(R, Wi2[H:2 * H], bi2[H:2 * H])); V2 = q(_linear(R, Wi2[2 * H:], bi2[2 * H:]))
        Q2s = q(Q2 * scale)
        Q2h = Q2s.reshape(Nk, nh, dh); K2h = K2.reshape(Nr, nh, dh); V2h = V2.reshape(Nr, nh, dh)
        S2 = np.einsum("jhd,thd->thj", Q2h, K2h).astype(f32)                  # stored [Nr, nh, Nk]
        if not self.bf16:
            S2 = np.einsum("jhd,thd->thj", Q2.reshape(Nk, nh, dh), K2h).astype(f32) * scale
        P2 = _softmax(S2, axis=0)                                             # softmax over keys t
        P2d, m_a2 = _drop(P2, seed, SITE_ATTN_KG2RG, row_base * nh * Nk, pd, training)
        if self.bf16:
            # the kernels run this direction flash-style over blocks of ``kg_block`` keys: exponentials relative to the BLOCK's
            # maximum, dropped, rounded to bf16 for the value product; the blocks' fp32 partial sums {max, sum, Z} are combined
            # afterwards.  (Rounding the normalised probabilities instead has the same relative error but not the same bits, and
            # the 13 KG rows of a sample are few enough for that to show in their FFN's ReLU decisions.)
            kb = int(self.kg_block)
            M = S2.max(axis=0)                                                # [nh, Nk]
            num = np.zeros((Nk, nh, dh), f32); den = np.zeros((nh, Nk), f32)
            first = int(self.kg_first_block.get(b_index, kb))
            for t0 in [0] + list(range(first, Nr, kb)):
                sl = slice(t0, min(Nr, first if t0 == 0 else t0 + kb))
                mb = S2[sl].max(axis=0)
                eb = np.exp(S2[sl] - mb).astype(f32)
                ed = eb if m_a2 is None else eb * m_a2[sl]
                zb = np.einsum("thj,thd->jhd", q(ed), V2h[sl]).astype(f32)
                w = np.exp(mb - M).astype(f32)                                # [nh, Nk]
                num += zb * w.T[:, :, None]; den += eb.sum(axis=0, dtype=f32) * w
            O2 = q((num / den.T[:, :, None]).astype(f32).reshape(Nk, H))
        else:
            O2 = np.einsum("thj,thd->jhd", P2d, V2h).astype(f32).reshape(Nk, H)
        A2 = _linear(O2, W("fusion.cross_attn_kg2rg.out_proj.weight"), P["fusion.cross_attn_kg2rg.out_proj.bias"])
        # fusion_model.py:130-131
        Y2, xh2, rstd2 = _layernorm(G + A2, P["fusion.ln_kg.weight"], P["fusion.ln_kg.bias"])
        H2 = np.maximum(_linear(q(Y2), W("fusion.ffn_kg.0.weight"), P["fusion.ffn_kg.0.bias"]), 0)
        H2d, m_f2 = _drop(H2, seed, SITE_FFN_KG, kg_row_base * 2 * H, pd, training)
        Zk = Y2 + _linear(H2d, P["fusion.ffn_kg.3.weight"], P["fusion.ffn_kg.3.bias"])
        # fusion_model.py:134-139
        comb = np.concatenate([Z.mean(axis=0, dtype=f32), Zk.mean(axis=0, dtype=f32)])[None, :]
        F1, F1pos = self._relu_tail(_linear(comb, P["fusion.fusion_layer.0.weight"], P["fusion.fusion_layer.0.bias"]), SITE_FUSE, b_index)
        F1d, m_fu = _drop(F1, seed, SITE_FUSE, b_index * H, pd, training)
        fused = _linear(F1d, P["fusion.fusion_layer.3.weight"], P["fusion.fusion_layer.3.bias"])
        outs, hc = self._heads(fused, training, seed, b_index)
        # returned maps: head-average of the (post-dropout) probabilities, as
        # torch's MHA does with need_weights=True, average_attn_weights=True
        outs["attn_rg2kg"] = Pd.mean(axis=1, dtype=f32)            # [Nr, Nk]
        outs["attn_kg2rg"] = P2d.mean(axis=1, dtype=f32).T.copy()  # [Nk, Nr]
        ca.update(rg=rg, kg=kg, R=R, G=G, Q=Q, Qs=Qs, Q2s=Q2s, Kk=Kk, Vk=Vk, Pm=Pm, Pd=Pd, m_a1=m_a1, O=O, xh1=xh1, rstd1=rstd1,
                  Y=Y, H1=H1, H1d=H1d, m_f1=m_f1, Q2=Q2, K2=K2, V2=V2, P2=P2, P2d=P2d, m_a2=m_a2, O2=O2,
                  xh2=xh2, rstd2=rstd2, Y2=Y2, H2=H2, H2d=H2d, m_f2=m_f2, comb=comb, F1=F1, F1pos=F1pos, F1d=F1d, m_fu=m_fu,
                  fused=fused, heads=hc, Nr=Nr, Nk=Nk)
        return outs, ca

    def _heads(self, fused, training, seed, b_index):
        """fusion_model.py:208-235, 239-242."""
        P, pd = self.p, float(self.cfg["dropout"])
        F = fused.shape[1]
        outs, hc = {}, {}
        for i, (hn, on) in enumerate((("mask_head", "mask"), ("instance_head", "instance"),
                                      ("edge_head", "edge"), ("score_head", "score"))):
            h, hpos = self._relu_tail(_linear(fused, P[f"{hn}.0.weight"], P[f"{hn}.0.bias"]), SITE_HEAD0 + i, b_index)
            hd, m = _drop(h, seed, SITE_HEAD0 + i, b_index * (F // 2), pd, training)
            o = _linear(hd, P[f"{hn}.3.weight"], P[f"{hn}.3.bias"])
            if on == "score":
                o = _sigmoid(o)
            outs[on] = o[0]
            hc[hn] = (h, hd, m, hpos)
        return outs, hc

    def _forward_late(self, rg, kg, training, seed, b_index):
        """LateFusion, fusion_model.py:149-171."""
        P, c = self.p, self.cfg
        H, pd = c["hidden_dim"], float(c["dropout"])
        comb = np.concatenate([rg.mean(axis=0, dtype=f32), kg.mean(axis=0, dtype=f32)])[None, :]
        a1 = np.maximum(_linear(comb, P["fusion.fusion.0.weight"], P["fusion.fusion.0.bias"]), 0)
        a1d, m1 = _drop(a1, seed, SITE_LATE0, b_index * H, pd, training)
        a2 = np.maximum(_linear(a1d, P["fusion.fusion.3.weight"], P["fusion.fusion.3.bias"]), 0)
        a2d, m2 = _drop(a2, seed, SITE_LATE0 + 1, b_index * (H // 2), pd, training)
        fused = _linear(a2d, P["fusion.fusion.6.weight"], P["fusion.fusion.6.bias"])
        outs, hc = self._heads(fused, training, seed, b_index)
        ca = dict(late=True, comb=comb, a1=a1, a1d=a1d, m1=m1, a2=a2, a2d=a2d, m2=m2, fused=fused, heads=hc)
        return outs, ca

    def forward(self, rg, kg, training=False, seed=0):
        """Dense batch like the reference forward(): rg [B,Nr,D] (2-/4-D accepted),
        kg [B,Nk,Dk].  Returns dict of stacked outputs and the per-sample caches."""
        rg, kg = collapse_inputs(np.asarray(rg, f32), np.asarray(kg, f32))
        return self.forward_list([rg[b] for b in range(rg.shape[0])], kg, training, seed)

    def forward_list(self, rg_list, kg, training=False, seed=0):
        """Packed variable-Nr batch: rg_list[b] is [Nr_b, D]; kg [B, Nk, Dk]."""
        outs, caches, base = [], [], 0
        for b, rg in enumerate(rg_list):
            o, ca = self.forward_sample(np.asarray(rg, f32), np.asarray(kg[b], f32), training, seed, base, b)
            outs.append(o); caches.append(ca); base += rg.shape[0]
        st = {k: np.stack([o[k] for o in outs]) for k in ("mask", "instance", "edge", "score")}
        if "attn_rg2kg" in outs[0]:
            st["attn_rg2kg"] = [o["attn_rg2kg"] for o in outs]
            st["attn_kg2rg"] = [o["attn_kg2rg"] for o in outs]
        return st, caches

    # ------------------------------------------------------------------ bwd
    def _heads_bwd(self, ca, d, g):
        """d: dict of grads w.r.t. the four outputs (score: post-sigmoid)."""
        P = self.p
        fused = ca["fused"]
        dfused = np.zeros_like(fused)
        for hn, on in (("mask_head", "mask"), ("instance_head", "instance"), ("edge_head", "edge"), ("score_head", "score")):
            h, hd, m, hpos = ca["heads"][hn]
            do = np.asarray(d[on], f32).reshape(1, -1)
            if on == "score":
                o = _sigmoid(_linear(hd, P[f"{hn}.3.weight"], P[f"{hn}.3.bias"]))
                do = do * o * (f32(1.0) - o)
            _acc(g, f"{hn}.3.weight", do.T @ hd); _acc(g, f"{hn}.3.bias", do.sum(0))
            dh = do @ P[f"{hn}.3.weight"]
            if m is not None:
                dh = dh * m
            dh = dh * hpos
            _acc(g, f"{hn}.0.weight", dh.T @ fused); _acc(g, f"{hn}.0.bias", dh.sum(0))
            dfused = dfused + dh @ P[f"{hn}.0.weight"]
        return dfused

    def backward_sample(self, ca, d, g, dbg=None):
        """Accumulate this sample's parameter gradients into dict ``g``.  ``dbg``: optional dict that receives the
        intermediate activation gradients (tests compare the fused backward kernels with them stage by stage)."""
        P, c = self.p, self.cfg
        dfused = self._heads_bwd(ca, d, g)
        if ca.get("late"):
            _acc(g, "fusion.fusion.6.weight", dfused.T @ ca["a2d"]); _acc(g, "fusion.fusion.6.bias", dfused.sum(0))
            da2 = dfused @ P["fusion.fusion.6.weight"]
            if ca["m2"] is not None:
                da2 = da2 * ca["m2"]
            da2 = da2 * (ca["a2"] > 0)
            _acc(g, "fusion.fusion.3.weight", da2.T @ ca["a1d"]); _acc(g, "fusion.fusion.3.bias", da2.sum(0))
            da1 = da2 @ P["fusion.fusion.3.weight"]
            if ca["m1"] is not None:
                da1 = da1 * ca["m1"]
            da1 = da1 * (ca["a1"] > 0)
            _acc(g, "fusion.fusion.0.weight", da1.T @ ca["comb"]); _acc(g, "fusion.fusion.0.bias", da1.sum(0))
            return
        H, nh = c["hidden_dim"], c["num_heads"]
        dh_ = H // nh
        Nr, Nk = ca["Nr"], ca["Nk"]
        scale = f32(1.0 / np.sqrt(dh_))
        # fusion layer
        _acc(g, "fusion.fusion_layer.3.weight", dfused.T @ ca["F1d"]); _acc(g, "fusion.fusion_layer.3.bias", dfused.sum(0))
        dF1 = dfused @ P["fusion.fusion_layer.3.weight"]
        if ca["m_fu"] is not None:
            dF1 = dF1 * ca["m_fu"]
        dF1 = dF1 * ca["F1pos"]
        _acc(g, "fusion.fusion_layer.0.weight", dF1.T @ ca["comb"]); _acc(g, "fusion.fusion_layer.0.bias", dF1.sum(0))
        dcomb = dF1 @ P["fusion.fusion_layer.0.weight"]
        if dbg is not None:
            dbg["dcomb"] = dcomb
        dZ = np.repeat(dcomb[:, :H] / f32(Nr), Nr, axis=0)
        dZk = np.repeat(dcomb[:, H:] / f32(Nk), Nk, axis=0)

        q = self.q
        W = (lambda name: q(P[name]))

        def ffn_ln_bwd(dZ_, pre, Y_, H_, Hd_, m_, xh_, rstd_, ln):
            _acc(g, f"fusion.{pre}.3.weight", dZ_.T @ Hd_); _acc(g, f"fusion.{pre}.3.bias", dZ_.sum(0))
            # (bf16 mode: the per-sample gradient row is rounded, then masked: the kernels' dH operand)
            dH = q(dZ_ @ P[f"fusion.{pre}.3.weight"] * (f32(1.0) if m_ is None else f32(m_.max() if m_.size else 1.0)))
            dH = dH * (H_ > 0) if m_ is None else dH * (m_ > 0) * (H_ > 0)
            if dbg is not None:
                dbg["dH_" + pre] = dH
            _acc(g, f"fusion.{pre}.0.weight", dH.T @ q(Y_)); _acc(g, f"fusion.{pre}.0.bias", dH.sum(0))
            dY = dZ_ + dH @ W(f"fusion.{pre}.0.weight")
            dU, dgam, dbet = _layernorm_bwd(dY, q(xh_), rstd_, P[f"fusion.{ln}.weight"])
            _acc(g, f"fusion.{ln}.weight", dgam); _acc(g, f"fusion.{ln}.bias", dbet)
            return q(dU)

        dU = ffn_ln_bwd(dZ, "ffn_rg", ca["Y"], ca["H1"], ca["H1d"], ca["m_f1"], ca["xh1"], ca["rstd1"], "ln_rg")
        dU2 = ffn_ln_bwd(dZk, "ffn_kg", ca["Y2"], ca["H2"], ca["H2d"], ca["m_f2"], ca["xh2"], ca["rstd2"], "ln_kg")
        dR = dU.copy(); dG = dU2.copy()
        # --- rg2kg attention  (scores were taken with the pre-scaled queries Qs = q(Q * scale): dK = dS^T Qs, dQ = scale * dS K)
        a = "fusion.cross_attn_rg2kg"
        _acc(g, f"{a}.out_proj.weight", dU.T @ ca["O"]); _acc(g, f"{a}.out_proj.bias", dU.sum(0))
        dO = q(dU @ W(f"{a}.out_proj.weight")).reshape(Nr, nh, dh_)
        Vh = ca["Vk"].reshape(Nk, nh, dh_); Kh = ca["Kk"].reshape(Nk, nh, dh_); Qsh = ca["Qs"].reshape(Nr, nh, dh_)
        dPd = np.einsum("thd,jhd->thj", dO, Vh).astype(f32)
        dVk = np.einsum("thj,thd->jhd", q(ca["Pd"]), dO).astype(f32).reshape(Nk, H)
        dP = dPd if ca["m_a1"] is None else dPd * ca["m_a1"]
        dS = q(ca["Pm"] * (dP - (ca["Pm"] * dP).sum(axis=2, keepdims=True, dtype=f32)))
        dQ = q((np.einsum("thj,jhd->thd", dS, Kh).astype(f32) * scale).reshape(Nr, H))
        dKk = np.einsum("thj,thd->jhd", dS, Qsh).astype(f32).reshape(Nk, H)
        if not self.bf16:
            dKk = (np.einsum("thj,thd->jhd", dS, ca["Q"].reshape(Nr, nh, dh_)).astype(f32) * scale).reshape(Nk, H)
        dKk, dVk = q(dKk), q(dVk)                   # (fp32 sums over the rows, rounded when they become operands)
        Wi = W(f"{a}.in_proj_weight")
        gW = np.concatenate([dQ.T @ ca["R"], dKk.T @ ca["G"], dVk.T @ ca["G"]], axis=0)
        _acc(g, f"{a}.in_proj_weight", gW); _acc(g, f"{a}.in_proj_bias", np.concatenate([dQ.sum(0), dKk.sum(0), dVk.sum(0)]))
        dR += dQ @ Wi[:H]; dG += dKk @ Wi[H:2 * H] + dVk @ Wi[2 * H:]
        # --- kg2rg attention
        a = "fusion.cross_attn_kg2rg"
        _acc(g, f"{a}.out_proj.weight", dU2.T @ ca["O2"]); _acc(g, f"{a}.out_proj.bias", dU2.sum(0))
        dO2 = q(dU2 @ W(f"{a}.out_proj.weight")).reshape(Nk, nh, dh_)
        V2h = ca["V2"].reshape(Nr, nh, dh_); K2h = ca["K2"].reshape(Nr, nh, dh_); Q2sh = ca["Q2s"].reshape(Nk, nh, dh_)
        dP2d = np.einsum("jhd,thd->thj", dO2, V2h).astype(f32)
        dV2 = q(np.einsum("thj,jhd->thd", q(ca["P2d"]), dO2).astype(f32).reshape(Nr, H))
        dP2 = dP2d if ca["m_a2"] is None else dP2d * ca["m_a2"]
        dS2 = q(ca["P2"] * (dP2 - (ca["P2"] * dP2).sum(axis=0, keepdims=True, dtype=f32)))
        dQ2 = (np.einsum("thj,thd->jhd", dS2, K2h).astype(f32) * scale).reshape(Nk, H)
        dK2 = np.einsum("thj,jhd->thd", dS2, Q2sh).astype(f32).reshape(Nr, H)
        if not self.bf16:
            dK2 = (np.einsum("thj,jhd->thd", dS2, ca["Q2"].reshape(Nk, nh, dh_)).astype(f32) * scale).reshape(Nr, H)
        dQ2, dK2 = q(dQ2), q(dK2)
        Wi2 = W(f"{a}.in_proj_weight")
        gW2 = np.concatenate([dQ2.T @ ca["G"], dK2.T @ ca["R"], dV2.T @ ca["R"]], axis=0)
        _acc(g, f"{a}.in_proj_weight", gW2); _acc(g, f"{a}.in_proj_bias", np.concatenate([dQ2.sum(0), dK2.sum(0), dV2.sum(0)]))
        dG += dQ2 @ Wi2[:H]; dR += dK2 @ Wi2[H:2 * H] + dV2 @ Wi2[2 * H:]
        dR, dG = q(dR), q(dG)
        if dbg is not None:
            dbg.update(dU=dU, dU2=dU2, dO=dO.reshape(Nr, H), dO2=dO2.reshape(Nk, H), dQ=dQ, dKk=dKk, dVk=dVk, dQ2=dQ2, dK2=dK2, dV2=dV2,
                       dR=dR, dG=dG)
        # --- input projections
        if "fusion.rg_proj.weight" in P:
            _acc(g, "fusion.rg_proj.weight", dR.T @ ca["rg"]); _acc(g, "fusion.rg_proj.bias", dR.sum(0))
        if "fusion.kg_proj.weight" in P:
            _acc(g, "fusion.kg_proj.weight", dG.T @ ca["kg"]); _acc(g, "fusion.kg_proj.bias", dG.sum(0))

    def zero_grads(self):
        return {k: np.zeros_like(v) for k, v in self.p.items()}


def _acc(g, name, val):
    g[name] += np.asarray(val, dtype=f32).reshape(g[name].shape)


# ---------------------------------------------------------------------- loss
def focal_loss(logits, targets, alpha=0.75, gamma=3.0):
    """AggressiveFocalLoss.forward, train_multimodal.py:40-57 (mean over batch).
    Returns (loss, dloss/dlogits)."""
    logits = np.asarray(logits, f32); B = logits.shape[0]
    p = _softmax(logits, axis=1)
    idx = np.arange(B)
    pt = p[idx, targets]
    ce = -np.log(pt)
    at = np.where(targets == 1, f32(alpha), f32(1.0 - alpha)).astype(f32)
    li = at * (f32(1.0) - pt) ** f32(gamma) * ce
    # d li / d pt, then d pt / d logits = pt*(onehot - p)
    dli_dpt = at * (-(f32(gamma)) * (f32(1.0) - pt) ** f32(gamma - 1.0) * ce - (f32(1.0) - pt) ** f32(gamma) / pt)
    onehot = np.zeros_like(p); onehot[idx, targets] = 1
    dlog = (dli_dpt * pt)[:, None] * (onehot - p) / f32(B)
    return f32(li.mean(dtype=f32)), dlog.astype(f32)


def cross_entropy(logits, targets):
    """F.cross_entropy, mean reduction (train_multimodal.py:260, :312)."""
    logits = np.asarray(logits, f32); B = logits.shape[0]
    p = _softmax(logits, axis=1)
    idx = np.arange(B)
    onehot = np.zeros_like(p); onehot[idx, targets] = 1
    return f32((-np.log(p[idx, targets])).mean(dtype=f32)), ((p - onehot) / f32(B)).astype(f32)


def bce_with_logits(x, y):
    """nn.BCEWithLogitsLoss, mean (train_multimodal.py:233, :263)."""
    x = np.asarray(x, f32); y = np.asarray(y, f32)
    l = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    return f32(l.mean(dtype=f32)), ((_sigmoid(x) - y) / f32(x.size)).astype(f32)


def mse(x, y):
    """nn.MSELoss, mean (train_multimodal.py:234, :266)."""
    x = np.asarray(x, f32); y = np.asarray(y, f32)
    return f32(((x - y) ** 2).mean(dtype=f32)), (f32(2.0) * (x - y) / f32(x.size)).astype(f32)


LOSS_WEIGHTS = (3.0, 1.0, 0.5, 0.3)   # train_multimodal.py:257,260,263,266


def sample_loss(outs_b, y, e, s):
    """The per-sample (B=1) loss of train_multimodal.py:256-268 and its gradient
    w.r.t. the four outputs.  outs_b: dict with mask[2], instance[2], edge[1],
    score[1] (score post-sigmoid)."""
    yv = np.array([y], dtype=np.int64)
    lf, dm = focal_loss(outs_b["mask"][None, :], yv)
    lc, di = cross_entropy(outs_b["instance"][None, :], yv)
    lb, de = bce_with_logits(outs_b["edge"], np.array([e], f32))
    lm, ds = mse(outs_b["score"], np.array([s], f32))
    w = [f32(x) for x in LOSS_WEIGHTS]
    terms = np.array([lf * w[0], lc * w[1], lb * w[2], lm * w[3]], f32)
    d = dict(mask=dm[0] * w[0], instance=di[0] * w[1], edge=de * w[2], score=ds * w[3])
    return f32(terms.sum(dtype=f32)), terms, d


def f1_scores(pred, lab):
    """calculate_f1_score, train_multimodal.py:197-220."""
    pred = np.asarray(pred); lab = np.asarray(lab)
    tp = float(((pred == 1) & (lab == 1)).sum()); fp = float(((pred == 1) & (lab == 0)).sum())
    fn = float(((pred == 0) & (lab == 1)).sum()); tn = float(((pred == 0) & (lab == 0)).sum())
    p1 = tp / (tp + fp + 1e-8); r1 = tp / (tp + fn + 1e-8); f1 = 2 * p1 * r1 / (p1 + r1 + 1e-8)
    p0 = tn / (tn + fn + 1e-8); r0 = tn / (tn + fp + 1e-8); f0 = 2 * p0 * r0 / (p0 + r0 + 1e-8)
    return dict(f1_class_0=f0, f1_class_1=f1, f1_avg=(f0 + f1) / 2, precision_1=p1, recall_1=r1)


# ----------------------------------------------------------------- optimiser
def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (train_multimodal.py:278): global L2 norm,
    coef = min(1, max_norm/(norm+1e-6)), grads scaled in place."""
    tot = f32(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    coef = min(f32(1.0), f32(max_norm) / (tot + f32(1e-6)))
    for k in grads:
        grads[k] = (grads[k] * f32(coef)).astype(f32)
    return tot


class AdamW:
    """torch.optim.AdamW with default betas/eps (train_multimodal.py:403-407)."""

    def __init__(self, params, lr=5e-4, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.wd, self.b1, self.b2, self.eps = lr, weight_decay, betas[0], betas[1], eps
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, params, grads, lr=None):
        lr = self.lr if lr is None else lr
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2s = np.sqrt(1.0 - self.b2 ** self.t)
        for k, p in params.items():
            g = grads[k]
            p *= f32(1.0 - lr * self.wd)
            self.m[k] = (self.m[k] * f32(self.b1) + g * f32(1.0 - self.b1)).astype(f32)
            self.v[k] = (self.v[k] * f32(self.b2) + g * g * f32(1.0 - self.b2)).astype(f32)
            denom = np.sqrt(self.v[k]) / f32(bc2s) + f32(self.eps)
            p -= (f32(lr / bc1) * self.m[k] / denom).astype(f32)


def cosine_warm_restarts_lr(base_lr, epoch, T_0=10, T_mult=2, eta_min=0.0):
    """Learning rate in effect during ``epoch`` (0-based) under
    CosineAnnealingWarmRestarts stepped once per epoch (train_multimodal.py:409-411,439)."""
    t, Ti = epoch, T_0
    while t >= Ti:
        t -= Ti
        Ti *= T_mult
    return eta_min + (base_lr - eta_min) * (1 + np.cos(np.pi * t / Ti)) / 2


def train_step(oracle, opt, rg_list, kg, y, e, s, training=True, seed=0, lr=None, max_norm=1.0, debug=False):
    """One optimizer step over a minibatch with the reference's semantics
    (train_multimodal.py:238-279): per-sample forward/backward, gradients
    SUMMED over the minibatch, one clip, one AdamW step.
    Returns dict(loss_terms [B,4], losses [B], grad_norm, grads (clipped), outs)."""
    outs, caches = oracle.forward_list(rg_list, kg, training=training, seed=seed)
    g = oracle.zero_grads()
    losses, terms, dbgs = [], [], []
    for b, ca in enumerate(caches):
        ob = {k: outs[k][b] for k in ("mask", "instance", "edge", "score")}
        l, t, d = sample_loss(ob, int(y[b]), float(e[b]), float(s[b]))
        dbg = {} if debug else None
        oracle.backward_sample(ca, d, g, dbg)
        losses.append(l); terms.append(t); dbgs.append(dbg)
    raw = {k: v.copy() for k, v in g.items()}
    norm = clip_grad_norm(g, max_norm)
    opt.step(oracle.p, g, lr=lr)
    return dict(losses=np.array(losses, f32), loss_terms=np.stack(terms), grad_norm=norm, grads=g, raw_grads=raw, outs=outs,
                caches=caches, dbg=dbgs)
