"""Developer probe: gradient error of the fused training step vs the bf16-operand oracle for a few shapes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fusion_oracle as FO, params as OP
from test_hip_parity import make_model, outs6, t2n
from test_hip_fused import _opt
cfg = OP.full_cfg()
for nrs, tail17 in (([64] * 16, -1), ([64] * 17, -1), ([64] * 16, 0), ([65] * 17, -1), ([64] * 17 , -1)):
    m = make_model(cfg, 6, "bf16").train(); eng = m._engine
    B = len(nrs)
    rgl = [OP.make_rg(n, 128, seed=300 + i) for i, n in enumerate(nrs)]
    kg = np.stack([OP.make_kg(13, 128, seed=400 + i) for i in range(B)])
    y, e, s = OP.make_labels(B, seed=21)
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, 6), bf16_operands=True)
    ref = FO.train_step(orc, FO.AdamW(orc.p), rgl, kg, y, e, s, training=True, seed=1234)
    batch = eng.make_batch(torch.from_numpy(np.concatenate(rgl)).cuda(), list(nrs), torch.from_numpy(kg).cuda())
    _opt("tail17", tail17)
    ws = eng.workspace(batch, private=True); g = eng.ensure_flat_grads(attach=True); g.zero_()
    outs, terms, pred = eng.train_raw(batch, ws, torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), True, 1234, eng._gtab)
    torch.cuda.synchronize()
    grads = {k: t2n(p.grad).copy() for k, p in m.named_parameters()}
    den = sum(float((ref["raw_grads"][k].astype(np.float64) ** 2).sum()) for k in grads)
    num = sum(float(((grads[k].astype(np.float64) - ref["raw_grads"][k]) ** 2).sum()) for k in grads)
    per = sorted(((float(np.sqrt(((grads[k].astype(np.float64) - ref["raw_grads"][k]) ** 2).sum() / max((ref["raw_grads"][k].astype(np.float64) ** 2).sum(), 1e-30))), k) for k in grads), reverse=True)
    dl = np.abs(t2n(outs) - outs6(ref["outs"]))
    dt = np.abs(t2n(terms) - ref["loss_terms"])
    print(f"B={B} Nr={nrs[0]} tail17={tail17}: global {np.sqrt(num/den):.5f}; logits max diff {dl.max():.2e} (sample {dl.max(1).argmax()}); loss terms max diff {dt.max():.2e}; worst {per[:3]}")
_opt("tail17", -1)
