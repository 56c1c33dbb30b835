"""Developer aid: one training call (dropout on) at B samples through the 64-row forward (wide2 = 1) and through the 32-row kernels (wide2 = 0,
fused_rt = 0) on the same inputs and seed; prints, per saved tensor, the rows where the two disagree.   python tools/dev/dev_w2_compare.py [B]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from oracle import params as OP
from test_hip_parity import make_model, t2n
from test_hip_fused import ws_bf16, ws_f32, _ws_raw, _opt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 70
kg_real = load_golden("kg_embeddings")["kg"]
cfg = OP.full_cfg()
m = make_model(cfg, 4, "bf16").train()
eng = m._engine
nrs = [380 + 3 * (i % 50) for i in range(B)]
T, Nk, H = sum(nrs), 13, 256
rg = np.concatenate([OP.make_rg(n, 128, seed=900 + i) for i, n in enumerate(nrs)])
kg = np.stack([kg_real] * B)
y, e, s = OP.make_labels(B, seed=21)
batch = eng.make_batch(torch.from_numpy(rg).cuda(), nrs, torch.from_numpy(kg).cuda())
res = []
for w2 in (1, 0):
    _opt("wide2", w2); _opt("fused_rt", -1 if w2 else 0)
    ws = eng.workspace(batch, private=True); ws.zero_()
    g = eng.ensure_flat_grads(attach=True); g.zero_()
    outs, terms, pred = eng.train_raw(batch, ws, torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), True, 41, eng._gtab)
    torch.cuda.synchronize()
    d = {"outs": t2n(outs)}
    for name, rows, cols in (("X16", T, 128), ("R16", T, H), ("Q16", T, H), ("KV2_16", T, 2 * H), ("O16", T, H), ("XH16", T, H), ("Y16", T, H),
                             ("O2_16", B * Nk, H), ("Y2_16", B * Nk, H), ("XH2_16", B * Nk, H)):
        d[name] = ws_bf16(eng, batch, ws, name, rows, cols)
    d["rstd1"] = ws_f32(eng, batch, ws, "rstd1", T)[:, None]
    d["lse2"] = ws_f32(eng, batch, ws, "lse2", B * 8 * 16 * 2).reshape(B, 8, 16, 2)[:, :, :Nk, :].reshape(B, -1)
    d["mask1"] = _ws_raw(eng, batch, ws, "mask1", 4 * T * 16).view(np.uint32).reshape(T, 16).astype(np.float64)
    d["mask2"] = _ws_raw(eng, batch, ws, "mask2", 4 * B * Nk * 16).view(np.uint32).reshape(B * Nk, 16).astype(np.float64)
    for name, n in (("Ymean", B * H), ("H1mean", B * 2 * H), ("Y2mean", B * H), ("H2mean", B * 2 * H)):
        d[name] = ws_f32(eng, batch, ws, name, n).reshape(B, -1)
    d["grads"] = {k: t2n(p.grad).copy() for k, p in m.named_parameters()}
    res.append(d)
a, b = res
off = np.cumsum([0] + nrs)
for k in a:
    if k == "grads":
        continue
    x, yv = a[k].astype(np.float64), b[k].astype(np.float64)
    scale = max(np.abs(yv).max(), 1e-9)
    if k.startswith("mask"):
        bad = (x != yv).any(axis=1)
        frac = ((x != yv).sum()) / x.size
        print(f"{k:8s} rows with a differing word: {int(bad.sum())} of {len(bad)} (words differing: {frac:.4f})", np.nonzero(bad)[0][:12])
        continue
    err = np.abs(x - yv).max(axis=1) / scale
    bad = np.nonzero(err > 0.03)[0]
    print(f"{k:8s} max rel err {err.max():.4f}  rows > 3 %: {len(bad)} of {len(err)}", bad[:12], ("samples " + str(sorted(set(np.searchsorted(off, bad, side='right') - 1))[:12])) if len(err) == T and len(bad) else "")
num = sum(((a["grads"][k].astype(np.float64) - b["grads"][k]) ** 2).sum() for k in a["grads"]); den = sum((b["grads"][k].astype(np.float64) ** 2).sum() for k in a["grads"])
print("global gradient difference between the two forwards:", np.sqrt(num / den))
