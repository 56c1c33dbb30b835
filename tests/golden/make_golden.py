#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Run in the build container only (needs /root/reference; the GPU box never runs
this):   python tests/golden/make_golden.py

It imports the reference's own modules
  /root/reference/models/multimodal/fusion_model.py      (model)
  /root/reference/models/multimodal/train_multimodal.py  (AggressiveFocalLoss,
                                                          calculate_f1_score)
with two accommodations recorded in SURVEY.md 8(c): ``torch_geometric.nn`` and
``cv2`` are absent from the image and are imported at module top level but never
used on this path, so empty stand-in modules are registered for the import to
succeed.  Nothing else of the reference is modified and none of its source is
written anywhere: the outputs are inputs/expected-output arrays only.

Parameters and inputs are NOT stored: they are regenerated from seeds by
oracle/params.py (numpy RandomState streams), and each fixture stores checksums
of them so a drifted generator is caught.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import params as OP  # noqa: E402


def import_reference():
    tg = types.ModuleType("torch_geometric"); tgn = types.ModuleType("torch_geometric.nn")
    tgn.global_mean_pool = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    tg.nn = tgn
    sys.modules.setdefault("torch_geometric", tg); sys.modules.setdefault("torch_geometric.nn", tgn)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, os.path.join(REF, "models", "multimodal"))
    import fusion_model as FM
    import train_multimodal as TM
    return FM, TM


def ref_model(FM, cfg, seed):
    m = FM.build_multimodal_model(dict(cfg))
    sd = {k: torch.from_numpy(v) for k, v in OP.make_params(cfg, seed).items()}
    m.load_state_dict(sd, strict=True)          # pins oracle/params.py's name/shape table
    return m


def t2n(t):
    return t.detach().cpu().numpy().astype(np.float32)


def checksum(a):
    a = np.asarray(a, np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * np.arange(1, a.size + 1).reshape(a.shape) % 7).sum()])


def sub(a, stride=37):
    return np.ascontiguousarray(np.asarray(a).reshape(-1)[::stride])


def real_kg():
    d = torch.load(os.path.join(REF, "models/knowledge_graph/kg_embeddings/all_embeddings.pt"), weights_only=True)
    names = list(d.keys())
    return names, np.concatenate([t2n(v) for v in d.values()], axis=0)   # [13,128], insertion order


def eval_cases(FM, out):
    names, kg = real_kg()
    np.savez(os.path.join(HERE, "kg_embeddings.npz"), kg=kg, names=np.array(names))
    cfg = OP.full_cfg()
    m = ref_model(FM, cfg, 0).eval()
    for nr in (303, 481, 500, 530):
        rg = OP.make_rg(nr, 128, seed=nr)
        with torch.no_grad():
            # the layout real data arrives in: rg [1,Nr,128], kg [1,13,1,128] (embedding_matcher.py:95-96)
            o = m(torch.from_numpy(rg)[None], torch.from_numpy(kg)[None, :, None, :], return_attention=True)
        out[f"eval_nr{nr}"] = dict(mask=t2n(o[0]), instance=t2n(o[1]), edge=t2n(o[2]), score=t2n(o[3]),
                                   attn_rg2kg=t2n(o[4]["rg2kg"]), attn_kg2rg=t2n(o[4]["kg2rg"]),
                                   rg_sum=checksum(rg))
    # the reference self-test shape (fusion_model.py:267-286): B=4, Nr=500, Nk=10, randn
    rg = np.stack([OP.make_rg(500, 128, seed=40 + b, kind="randn") for b in range(4)])
    kg10 = np.stack([OP.make_rg(10, 128, seed=50 + b, kind="randn") for b in range(4)])
    with torch.no_grad():
        o = m(torch.from_numpy(rg), torch.from_numpy(kg10), return_attention=True)
    out["eval_selftest_b4"] = dict(mask=t2n(o[0]), instance=t2n(o[1]), edge=t2n(o[2]), score=t2n(o[3]),
                                   attn_rg2kg_sub=sub(t2n(o[4]["rg2kg"])), attn_kg2rg_sub=sub(t2n(o[4]["kg2rg"])),
                                   rg_sum=checksum(rg), kg_sum=checksum(kg10))
    # input-shape handling, fusion_model.py:86-105
    rg2 = OP.make_rg(6, 128, seed=60)           # 2-D [B=6,128] -> Nr=1
    kg2 = OP.make_kg(6, 128, seed=61)           # 2-D [B=6,128] -> Nk=1
    rg4 = np.stack([OP.make_rg(12, 128, seed=62 + b) for b in range(2)]).reshape(2, 3, 4, 128)   # general view
    kg4 = np.stack([OP.make_kg(5, 128, seed=64 + b) for b in range(2)]).reshape(2, 1, 5, 128)    # a == 1
    with torch.no_grad():
        o2 = m(torch.from_numpy(rg2), torch.from_numpy(kg2))
        o4 = m(torch.from_numpy(rg4), torch.from_numpy(kg4))
    out["eval_2d"] = dict(mask=t2n(o2[0]), instance=t2n(o2[1]), edge=t2n(o2[2]), score=t2n(o2[3]))
    out["eval_4d"] = dict(mask=t2n(o4[0]), instance=t2n(o4[1]), edge=t2n(o4[2]), score=t2n(o4[3]))
    try:
        m(torch.zeros(1, 1, 1, 2, 128), torch.from_numpy(kg2))
        raise SystemExit("reference accepted a 5-D input?")
    except ValueError as ex:
        out["eval_5d_error"] = dict(message=np.array(str(ex)))


def ref_train_steps(FM, TM, cfg, seed, nrs, nk, steps, kg_fixed=None, lr=5e-4, wd=1e-4, kg4d=True):
    """The reference's schedule (train_multimodal.py:238-279) with dropout=0:
    per-sample B=1 forward/loss/backward, grads summed, clip 1.0, AdamW."""
    import torch.nn as nn
    import torch.nn.functional as F
    m = ref_model(FM, cfg, seed).train()
    opt = torch.optim.AdamW(m.parameters(), lr=lr, weight_decay=wd)
    focal = TM.AggressiveFocalLoss(alpha=0.75, gamma=3.0)
    bce, mse = nn.BCEWithLogitsLoss(), nn.MSELoss()
    rec = []
    for st in range(steps):
        B = len(nrs)
        y, e, s = OP.make_labels(B, seed=100 * seed + st)
        opt.zero_grad()
        terms, outs = [], []
        for b, nr in enumerate(nrs):
            rg = torch.from_numpy(OP.make_rg(nr, cfg["rg_dim"], seed=1000 * st + b))[None]
            kgb = kg_fixed if kg_fixed is not None else OP.make_kg(nk, cfg["kg_dim"], seed=1000 * st + 500 + b)
            # LateFusion only averages 3-D inputs (fusion_model.py:165-168); the 4-D layout is a cross-attention case
            kg = torch.from_numpy(kgb)[None, :, None, :] if kg4d else torch.from_numpy(kgb)[None]
            yl = torch.tensor([int(y[b])]); el = torch.tensor([float(e[b])]); sl = torch.tensor([float(s[b])])
            mo, io, eo, so = m(rg, kg)
            l1 = focal(mo, yl) * 3.0; l2 = F.cross_entropy(io, yl) * 1.0
            l3 = bce(eo.squeeze(1), el) * 0.5; l4 = mse(so.squeeze(1), sl) * 0.3
            (l1 + l2 + l3 + l4).backward()
            terms.append([float(l1.detach()), float(l2.detach()), float(l3.detach()), float(l4.detach())])
            outs.append(np.concatenate([t2n(mo)[0], t2n(io)[0], t2n(eo)[0], t2n(so)[0]]))
        raw = {k: t2n(p.grad) for k, p in m.named_parameters()}
        norm = float(torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0))
        opt.step()
        rec.append(dict(terms=np.array(terms, np.float32), outs=np.stack(outs), norm=norm, raw=raw,
                        params={k: t2n(p) for k, p in m.named_parameters()}))
    return rec


def pack_step(rec, full):
    d = dict(loss_terms=rec["terms"], outs=rec["outs"], grad_norm=np.array(rec["norm"]))
    for k, v in rec["raw"].items():
        d[f"gnorm/{k}"] = np.array(np.sqrt((v.astype(np.float64) ** 2).sum()))
        d[f"gsum/{k}"] = np.array(v.astype(np.float64).sum())
        d[f"g/{k}"] = v if full else sub(v)
        d[f"p/{k}"] = rec["params"][k] if full else sub(rec["params"][k])
    return d


def train_cases(FM, TM, out):
    _, kg = real_kg()
    cfg = OP.full_cfg(dict(dropout=0.0))
    # default config, B=4 with the real-data Nr spread, real KG rows, two optimizer steps
    recs = ref_train_steps(FM, TM, cfg, seed=0, nrs=(303, 481, 500, 530), nk=13, steps=2, kg_fixed=kg)
    for i, r in enumerate(recs):
        out[f"train_default_step{i}"] = pack_step(r, full=False)
    # small configs with FULL gradients (different rg/kg dims, odd Nr/Nk, Identity projections)
    small = dict(
        small_a=dict(cfg=dict(rg_dim=32, kg_dim=48, hidden_dim=64, num_heads=4, dropout=0.0), nrs=(7, 20, 1), nk=5),
        small_ident=dict(cfg=dict(rg_dim=64, kg_dim=64, hidden_dim=64, num_heads=2, dropout=0.0), nrs=(9, 33), nk=3),
        small_cls3=dict(cfg=dict(rg_dim=16, kg_dim=16, hidden_dim=32, num_heads=8, num_classes=3, dropout=0.0), nrs=(65,), nk=1),
        late=dict(cfg=dict(fusion_type="late", hidden_dim=64, dropout=0.0), nrs=(11, 40), nk=13),
    )
    for name, sc in small.items():
        recs = ref_train_steps(FM, TM, OP.full_cfg(sc["cfg"]), seed=3, nrs=sc["nrs"], nk=sc["nk"], steps=2,
                               kg4d=(name != "late"))
        for i, r in enumerate(recs):
            out[f"train_{name}_step{i}"] = pack_step(r, full=True)
        out[f"train_{name}_meta"] = dict(cfg=np.array(json.dumps(OP.full_cfg(sc["cfg"]))), nrs=np.array(sc["nrs"]), nk=np.array(sc["nk"]))


def loss_cases(TM, out):
    import torch.nn as nn
    import torch.nn.functional as F
    rs = np.random.RandomState(7)
    logits = (rs.standard_normal((8, 2)) * 2).astype(np.float32)
    tg = (rs.uniform(size=8) < 0.5).astype(np.int64)
    x = torch.from_numpy(logits).requires_grad_(True)
    l = TM.AggressiveFocalLoss(alpha=0.75, gamma=3.0)(x, torch.from_numpy(tg)); l.backward()
    d = dict(logits=logits, targets=tg, focal=np.array(float(l.detach())), focal_grad=t2n(x.grad))
    x = torch.from_numpy(logits).requires_grad_(True)
    l = F.cross_entropy(x, torch.from_numpy(tg)); l.backward()
    d.update(ce=np.array(float(l)), ce_grad=t2n(x.grad))
    ex = (rs.standard_normal(8) * 3).astype(np.float32); ey = (rs.uniform(size=8) < 0.5).astype(np.float32)
    x = torch.from_numpy(ex).requires_grad_(True)
    l = nn.BCEWithLogitsLoss()(x, torch.from_numpy(ey)); l.backward()
    d.update(edge=ex, edge_t=ey, bce=np.array(float(l)), bce_grad=t2n(x.grad))
    sx = rs.uniform(size=8).astype(np.float32); sy = rs.uniform(size=8).astype(np.float32)
    x = torch.from_numpy(sx).requires_grad_(True)
    l = nn.MSELoss()(x, torch.from_numpy(sy)); l.backward()
    d.update(score=sx, score_t=sy, mse=np.array(float(l)), mse_grad=t2n(x.grad))
    pred = (rs.uniform(size=50) < 0.4).astype(np.int64); lab = (rs.uniform(size=50) < 0.5).astype(np.int64)
    f = TM.calculate_f1_score(torch.from_numpy(pred), torch.from_numpy(lab))
    d.update(f1_pred=pred, f1_lab=lab, **{f"f1/{k}": np.array(float(v)) for k, v in f.items()})
    # CosineAnnealingWarmRestarts(T_0=10, T_mult=2) stepped per epoch (train_multimodal.py:409-411,439)
    p = torch.nn.Parameter(torch.zeros(1)); o = torch.optim.AdamW([p], lr=5e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(o, T_0=10, T_mult=2)
    lrs = []
    for _ in range(35):
        lrs.append(o.param_groups[0]["lr"]); o.step(); sch.step()
    d["lr_schedule"] = np.array(lrs)
    out["loss"] = d


def nr_histogram():
    """num_nodes of the 6000 shipped RG embeddings (data, not code):
    models/region_graph/rg_embeddings/embedding_summary.json."""
    with open(os.path.join(REF, "models/region_graph/rg_embeddings/embedding_summary.json")) as f:
        s = json.load(f)
    n = np.array([v["num_nodes"] for v in s["images"].values()], np.int32)
    vals, cnt = np.unique(n, return_counts=True)
    np.savez(os.path.join(HERE, "nr_histogram.npz"), values=vals.astype(np.int32), counts=cnt.astype(np.int32))
    print("Nr histogram: min %d mean %.1f max %d" % (n.min(), n.mean(), n.max()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    FM, TM = import_reference()
    out = {}
    eval_cases(FM, out)
    train_cases(FM, TM, out)
    loss_cases(TM, out)
    nr_histogram()
    for name, d in out.items():
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **d)
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz"))
    print(f"wrote {len(out) + 2} fixtures, {tot / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
