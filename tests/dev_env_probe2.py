import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fusion_oracle as FO, params as OP
from test_hip_parity import make_model, outs6, t2n
from test_hip_fused import _opt
cfg = OP.full_cfg()
for nrs, prec, drop in (([64] * 4, "bf16", True), ([64] * 4, "f32", True), ([64] * 4, "bf16", False), ([32] * 4, "bf16", True), ([96] * 4, "bf16", True), ([63] * 4, "bf16", True)):
    m = make_model(cfg, 6, prec).train(); eng = m._engine
    B = len(nrs)
    rgl = [OP.make_rg(n, 128, seed=300 + i) for i, n in enumerate(nrs)]
    kg = np.stack([OP.make_kg(13, 128, seed=400 + i) for i in range(B)])
    y, e, s = OP.make_labels(B, seed=21)
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, 6), bf16_operands=(prec == "bf16"))
    ref = FO.train_step(orc, FO.AdamW(orc.p), rgl, kg, y, e, s, training=drop, seed=1234)
    batch = eng.make_batch(torch.from_numpy(np.concatenate(rgl)).cuda(), list(nrs), torch.from_numpy(kg).cuda())
    ws = eng.workspace(batch, private=True); g = eng.ensure_flat_grads(attach=True); g.zero_()
    outs, terms, pred = eng.train_raw(batch, ws, torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), drop, 1234, eng._gtab)
    torch.cuda.synchronize()
    grads = {k: t2n(p.grad).copy() for k, p in m.named_parameters()}
    out = []
    for k in ("fusion.ffn_rg.3.bias", "fusion.ffn_kg.3.bias", "fusion.ln_rg.weight", "fusion.fusion_layer.0.weight", "fusion.ffn_rg.0.bias"):
        a, b = grads[k].astype(np.float64).ravel(), ref["raw_grads"][k].astype(np.float64).ravel()
        out.append(f"{k.split('fusion.')[-1]}: rel {np.linalg.norm(a - b) / np.linalg.norm(b):.4f} slope {np.dot(a, b) / np.dot(b, b):.4f}")
    print(f"Nr={nrs[0]} {prec} dropout={drop}: " + " | ".join(out))
