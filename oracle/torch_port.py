"""torch-CPU restatement of the fusion training step.  TEST / MEASUREMENT INFRASTRUCTURE (see oracle/__init__.py).

bench.py's ``cpu_baseline`` leg times this next to the numpy oracle: the reference itself runs on torch's CPU kernels
(oneDNN/BLAS GEMMs, autograd), which is faster than per-sample numpy on a many-core host, so this is the fairer
"reference CPU path" figure.  It restates the same arithmetic as oracle/fusion_oracle.py with torch ops and autograd --
per-sample forward at batch size 1, the 4-term loss, gradients summed over the minibatch, one clip, one AdamW step
(/root/reference/models/multimodal/train_multimodal.py:238-279; model: fusion_model.py:75-146, 208-246) -- and is held
to the numpy oracle by tests/test_oracle_golden.py::test_torch_port_matches_numpy_oracle (eval mode / dropout 0; in
train mode with dropout > 0 it draws torch's own masks, which is what the reference does and costs the same).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _mha(q_in, kv_in, w_in, b_in, w_out, b_out, nh, p, training):
    """nn.MultiheadAttention forward for one sample: q_in [Nq,H], kv_in [Nk,H] (fusion_model.py:112-118, 123-129)."""
    H = q_in.shape[1]
    dh = H // nh
    q = F.linear(q_in, w_in[:H], b_in[:H]).view(-1, nh, dh).transpose(0, 1)              # [nh, Nq, dh]
    k = F.linear(kv_in, w_in[H:2 * H], b_in[H:2 * H]).view(-1, nh, dh).transpose(0, 1)
    v = F.linear(kv_in, w_in[2 * H:], b_in[2 * H:]).view(-1, nh, dh).transpose(0, 1)
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(dh), dim=-1)
    a = F.dropout(a, p, training)
    o = (a @ v).transpose(0, 1).reshape(-1, H)
    return F.linear(o, w_out, b_out)


def forward_sample(P, cfg, rg, kg, training):
    """One sample -> (mask [C], instance [C], edge [1], score [1]); P: dict of torch parameters under the reference's names."""
    H, nh, p = cfg["hidden_dim"], cfg["num_heads"], float(cfg["dropout"])
    drop = lambda x: F.dropout(x, p, training)
    pre = "fusion."
    R = F.linear(rg, P[pre + "rg_proj.weight"], P[pre + "rg_proj.bias"]) if pre + "rg_proj.weight" in P else rg
    G = F.linear(kg, P[pre + "kg_proj.weight"], P[pre + "kg_proj.bias"]) if pre + "kg_proj.weight" in P else kg
    a1, a2 = pre + "cross_attn_rg2kg.", pre + "cross_attn_kg2rg."
    A = _mha(R, G, P[a1 + "in_proj_weight"], P[a1 + "in_proj_bias"], P[a1 + "out_proj.weight"], P[a1 + "out_proj.bias"], nh, p, training)
    Y = F.layer_norm(R + A, (H,), P[pre + "ln_rg.weight"], P[pre + "ln_rg.bias"])
    Z = Y + F.linear(drop(F.relu(F.linear(Y, P[pre + "ffn_rg.0.weight"], P[pre + "ffn_rg.0.bias"]))), P[pre + "ffn_rg.3.weight"], P[pre + "ffn_rg.3.bias"])
    A2 = _mha(G, R, P[a2 + "in_proj_weight"], P[a2 + "in_proj_bias"], P[a2 + "out_proj.weight"], P[a2 + "out_proj.bias"], nh, p, training)
    Y2 = F.layer_norm(G + A2, (H,), P[pre + "ln_kg.weight"], P[pre + "ln_kg.bias"])
    Zk = Y2 + F.linear(drop(F.relu(F.linear(Y2, P[pre + "ffn_kg.0.weight"], P[pre + "ffn_kg.0.bias"]))), P[pre + "ffn_kg.3.weight"], P[pre + "ffn_kg.3.bias"])
    comb = torch.cat([Z.mean(0), Zk.mean(0)])[None]
    fused = F.linear(drop(F.relu(F.linear(comb, P[pre + "fusion_layer.0.weight"], P[pre + "fusion_layer.0.bias"]))),
                     P[pre + "fusion_layer.3.weight"], P[pre + "fusion_layer.3.bias"])
    outs = []
    for hn in ("mask_head", "instance_head", "edge_head", "score_head"):
        o = F.linear(drop(F.relu(F.linear(fused, P[hn + ".0.weight"], P[hn + ".0.bias"]))), P[hn + ".3.weight"], P[hn + ".3.bias"])[0]
        outs.append(torch.sigmoid(o) if hn == "score_head" else o)
    return outs


def sample_loss(outs, y, e, s):
    """3*focal(alpha .75, gamma 3) + CE + 0.5*BCEWithLogits + 0.3*MSE at batch size 1 (train_multimodal.py:29-57, 256-268)."""
    mask, inst, edge, score = outs
    logp = F.log_softmax(mask, dim=0)[y]
    pt = logp.exp()
    focal = (0.75 if y == 1 else 0.25) * (1 - pt) ** 3 * (-logp)
    ce = -F.log_softmax(inst, dim=0)[y]
    bce = F.binary_cross_entropy_with_logits(edge, torch.tensor([e], dtype=edge.dtype))
    mse = (score - s).pow(2).mean()
    return 3.0 * focal + ce + 0.5 * bce + 0.3 * mse


class TorchPort:
    def __init__(self, cfg, params, lr=5e-4, weight_decay=1e-4):
        self.cfg = cfg
        self.P = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in params.items()}
        self.opt = torch.optim.AdamW(list(self.P.values()), lr=lr, weight_decay=weight_decay)

    def train_step(self, rg_list, kg, y, e, s, training=True):
        """One optimizer step with the reference's schedule; returns the per-sample losses."""
        self.opt.zero_grad()
        losses = []
        for b, rg in enumerate(rg_list):
            outs = forward_sample(self.P, self.cfg, torch.from_numpy(rg), torch.from_numpy(kg[b]), training)
            l = sample_loss(outs, int(y[b]), float(e[b]), float(s[b]))
            l.backward()
            losses.append(float(l.detach()))
        norm = torch.nn.utils.clip_grad_norm_(list(self.P.values()), max_norm=1.0)
        self.opt.step()
        return losses, float(norm)
