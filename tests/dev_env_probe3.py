import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fusion_oracle as FO, params as OP
from test_hip_parity import make_model, outs6, t2n
from test_hip_fused import _opt, ws_f32, ws_bf16
cfg = OP.full_cfg()
nrs = [64] * 16
m = make_model(cfg, 6, "bf16").train(); eng = m._engine
B = len(nrs); T = sum(nrs)
rgl = [OP.make_rg(n, 128, seed=300 + i) for i, n in enumerate(nrs)]
kg = np.stack([OP.make_kg(13, 128, seed=400 + i) for i in range(B)])
y, e, s = OP.make_labels(B, seed=21)
orc = FO.FusionOracle(cfg, OP.make_params(cfg, 6), bf16_operands=True)
ref = FO.train_step(orc, FO.AdamW(orc.p), rgl, kg, y, e, s, training=True, seed=1234, debug=True)
batch = eng.make_batch(torch.from_numpy(np.concatenate(rgl)).cuda(), list(nrs), torch.from_numpy(kg).cuda())
ws = eng.workspace(batch, private=True); ws.zero_(); g = eng.ensure_flat_grads(attach=True); g.zero_()
outs, terms, pred = eng.train_raw(batch, ws, torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), True, 1234, eng._gtab)
torch.cuda.synchronize()
dc = ws_f32(eng, batch, ws, "dcomb", B * 512).reshape(B, 512)
want = np.concatenate([d["dcomb"] for d in ref["dbg"]])
for b in range(B):
    a_, w_ = dc[b].astype(np.float64), want[b].astype(np.float64)
    print(b, "dcomb rel err RG half %.4f KG half %.4f | loss terms kernel %s oracle %s" % (np.linalg.norm(a_[:256] - w_[:256]) / np.linalg.norm(w_[:256]), np.linalg.norm(a_[256:] - w_[256:]) / np.linalg.norm(w_[256:]), np.round(t2n(terms)[b], 5), np.round(ref["loss_terms"][b], 5)))
dU = ws_bf16(eng, batch, ws, "dU16", T, 256); wdU = np.concatenate([d["dU"] for d in ref["dbg"]])
for b in range(B):
    sl = slice(64 * b, 64 * b + 64)
    print(b, "dU rel err %.4f" % (np.linalg.norm(dU[sl] - wdU[sl]) / np.linalg.norm(wdU[sl])))
