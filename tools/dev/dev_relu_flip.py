"""Developer aid: why one six-sample batch shows a 12 % gradient difference between the fused and the GEMM-per-layer bf16
schedules (and the oracle): forward tensors agree to 1e-3, but in three samples ONE head unit with a large gradient has its
pre-activation within 2e-4 of zero and lands on different sides of the ReLU.  Prints the pooled / tail tensors of both
schedules, the per-tensor gradient differences and the flipped units.  python tools/dev/dev_relu_flip.py"""
import sys, os, ctypes as C
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np, torch
from oracle import fusion_oracle as FO, params as OP
from test_hip_parity import make_model, outs6, t2n
from camouflage_multimodal_amd import _lib
def opt(n, v): _lib.check(_lib.lib().camo_debug_set_option(n.encode(), v), "opt")
cfg = OP.full_cfg()
nk = 13
nrs = [33, 31, 1, 2, 530, 96]
m = make_model(cfg, 6, "bf16").train(True); eng = m._engine
B = len(nrs)
rgl = [OP.make_rg(n, 128, seed=300 + i) for i, n in enumerate(nrs)]
kg = np.stack([OP.make_kg(nk, 128, seed=400 + i) for i in range(B)])
y, e, s = OP.make_labels(B, seed=21)
orc = FO.FusionOracle(cfg, OP.make_params(cfg, 6))
ref = FO.train_step(orc, FO.AdamW(orc.p), rgl, kg, y, e, s, training=True, seed=1234, debug=True)
batch = eng.make_batch(torch.from_numpy(np.concatenate(rgl)).cuda(), list(nrs), torch.from_numpy(kg).cuda())
def wsf(ws, name, n):
    off = _lib.lib().camo_debug_ws_offset(C.byref(eng.dims), batch.B, batch.T, batch.Nk, name.encode())
    return ws[off:off + 4 * n].cpu().numpy().view(np.float32).copy()
res = []
for fused in (1, 0):
    opt("fused", fused); opt("tail17", 0)
    ws = eng.workspace(batch, private=True); ws.zero_()
    g = eng.ensure_flat_grads(attach=True); g.zero_()
    o, t, p = eng.train_raw(batch, ws, torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), True, 1234, eng._gtab)
    torch.cuda.synchronize()
    gr = {k: t2n(pp.grad).copy() for k, pp in m.named_parameters()}
    pooled = {n: wsf(ws, n, B * w).reshape(B, w) for n, w in (("Ymean", 256), ("H1mean", 512), ("Y2mean", 256), ("H2mean", 512), ("comb", 512), ("fused", 256), ("F1", 256), ("hid", 512), ("dhid", 512), ("dfused", 256), ("dF1", 256), ("dcomb", 512), ("dHm1", 512), ("dHm2", 512))}
    res.append((t2n(o), t2n(t), gr, pooled))
opt("fused", -1); opt("tail17", -1)
(oa, ta, ga, pa), (ob, tb, gb, pb) = res
print("outs max diff", np.abs(oa - ob).max(), "terms", np.abs(ta - tb).max())
for n in pa:
    d = np.abs(pa[n] - pb[n]); print(n, "max diff per sample", np.round(d.max(1), 5), "scale", np.round(np.abs(pb[n]).max(1), 3))
rows = sorted(((float(np.linalg.norm(ga[k] - gb[k]) / max(np.linalg.norm(gb[k]), 1e-30)), k) for k in ga), reverse=True)
for r, k in rows[:12]: print(f"{k:45s} rel diff fused vs gemm-per-layer {r:.4f}")
# per-sample contribution check for a head bias: oracle per-sample grads if available
for k in ("instance_head.0.bias", "fusion.rg_proj.weight", "mask_head.3.weight", "fusion.fusion_layer.0.weight"):
    a, b = ga[k].ravel().astype(np.float64), gb[k].ravel().astype(np.float64)
    print(k, "projection coefficient <fused, ref>/<ref, ref> =", round(float(a @ b / (b @ b)), 4))
for b in (0, 2, 3, 1):
    ha, hb = pa["hid"][b], pb["hid"][b]
    flip = (ha > 0) != (hb > 0)
    da, db = pa["dhid"][b], pb["dhid"][b]
    big = np.abs(da - db) > 1e-3
    print("sample", b, "relu pattern flips", int(flip.sum()), "units with |d dhid| > 1e-3:", int(big.sum()),
          "of which hid>0 in both:", int((big & (ha > 0) & (hb > 0)).sum()))
    idx = np.nonzero(big)[0][:6]
    for i in idx: print("    unit", i, "hid", ha[i], hb[i], "dhid", da[i], db[i])
    print("    outs", oa[b], ob[b])
