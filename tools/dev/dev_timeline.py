"""Developer aid: per-block phase timeline of the fused forward kernels (100 MHz wall clock stamps).
   python tools/dev/dev_timeline.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_batches  # noqa: E402
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model = build_multimodal_model({}).cuda().set_precision("bf16").eval()
tr = NativeTrainer(model)
rg, nrs, kg, *_ = make_batches(1, B, 0)[0]
rg, kg = torch.from_numpy(rg).cuda(), torch.from_numpy(kg).cuda()
NB = 8192
_lib.lib().camo_debug_set_option(b"fused_variant", variant)
nkg = B * ((max(nrs) + 63) // 64)
buf = torch.zeros(2 * NB * 8, dtype=torch.int64, device="cuda")
train = len(sys.argv) > 3 and sys.argv[3] == "train"
buf = torch.zeros(5 * NB * 8, dtype=torch.int64, device="cuda")
if train:
    model.train()
    y, e, s_ = (torch.from_numpy(x).cuda() for x in make_batches(1, B, 0)[0][3:])
    step = lambda: tr.step(rg, nrs, kg, y, e, s_)
else:
    step = lambda: tr.evaluate(rg, nrs, kg)
for i in range(5):
    step()
_lib.lib().camo_debug_set_stamps(buf.data_ptr(), NB)
step()
torch.cuda.synchronize()
_lib.lib().camo_debug_set_stamps(None, 0)
st = buf.cpu().numpy().reshape(5, NB, 8)
if train:
    t = st[4].reshape(-1)[:64 * 32].reshape(64, 32).astype(np.float64)
    if t[:, 0].min() > 0:
        t0 = t[:, 0].min()
        names = {0: "start", 1: "staged", 2: "L1 done", 3: "L2 issued", 4: "AR1 done", 5: "F1 staged", 6: "L3 done", 7: "L4 issued", 8: "AR2 done",
                 9: "hid staged", 10: "head outputs", 11: "loss", 12: "output-layer grads", 13: "d hidden", 14: "d fused + hidden-layer grads",
                 15: "d F1 issued", 16: "AR3 done", 17: "dF staged", 18: "d comb", 19: "fusion layer 0 grads", 20: "end"}
        print("--- tail (one launch): stamp (median / max over the 64 blocks, us from first block start; delta to previous)")
        prev = 0.0
        for k in range(21):
            med = np.median(t[:, k] - t0) / 100
            print(f"   {names[k]:30s} {med:7.2f} / {(t[:, k] - t0).max() / 100:7.2f}   +{med - prev:5.2f}")
            prev = med
for k, name in enumerate(("front", "back", "bwd1", "bwd2") if train else ("front", "back")):
    s = st[k]
    if name == "back":          # KG->RG attention splits: stamps 4 (split done), 5 (ticket drawn), 6 (combine done, last arriver only)
        sp = s[(s[:, 0] > 0) & (s[:, 5] > 0)].astype(np.float64)
        if len(sp):
            la = sp[sp[:, 6] > 0]
            nl = sp[sp[:, 6] == 0]
            if len(nl) and (nl[:, 1] > 0).all():
                print(f"    KG split blocks that are not last arrivers: scores ready {np.median(nl[:, 1] - nl[:, 0]) / 100:.2f} us, exp/PV done "
                      f"{np.median(nl[:, 2] - nl[:, 0]) / 100:.2f}, partial stored {np.median(nl[:, 4] - nl[:, 0]) / 100:.2f}, ticket {np.median(nl[:, 5] - nl[:, 0]) / 100:.2f}")
            print(f"    KG->RG attention: {len(sp)} splits: split {np.median(sp[:, 4] - sp[:, 0]) / 100:.2f} us (max {(sp[:, 4] - sp[:, 0]).max() / 100:.2f}), "
                  f"drain+ticket {np.median(sp[:, 5] - sp[:, 4]) / 100:.2f} (max {(sp[:, 5] - sp[:, 4]).max() / 100:.2f}); {len(la)} last arrivers: "
                  f"start->ticket {np.median(la[:, 5] - la[:, 0]) / 100:.2f}, combine {np.median(la[:, 6] - la[:, 5]) / 100:.2f} us; "
                  f"last ticket drawn at {(sp[:, 5].max() - sp[:, 0].min()) / 100:.2f} us of the kernel")
    act = (s[:, 0] > 0) & (s[:, 3] > 0)          # (blocks that return early leave later stamps empty)
    s = s[act].astype(np.float64)
    t0 = s[:, 0].min()
    if name in ("bwd1", "bwd2", "back", "front"):
        st0 = st[k][:, 0].astype(np.float64); ok = st0 > 0
        ids = np.nonzero(ok)[0]; late = ids[np.argsort(-st0[ok])[:14]]
        print(f"    {name} latest starters (block id: start us):", ", ".join(f"{i}: {(st0[i] - st0[ok].min()) / 100:.2f}" for i in sorted(late)))
    print(f"--- {name} (variant {variant}): {act.sum()} blocks, kernel span {(s[:, 3].max() - t0) / 100:.2f} us; first block starts at 0, last starts at {(s[:, 0].max() - t0) / 100:.2f} us")
    ph = np.diff(s[:, :4], axis=1) / 100.0
    idx = np.nonzero(act)[0]
    if name == "back":
        groups = [("KG splits", idx < nkg), ("RG tiles", idx >= nkg)]
    elif name == "bwd1":                      # grid order: RG tiles, KG blocks, (writer blocks: no stamps)
        rgmax = sum(nrs) // 32 + B
        groups = [("KG blocks", (idx >= rgmax) & (idx < rgmax + B)), ("RG tiles", idx < rgmax)]
    elif name == "bwd2":
        groups = [("KG blocks", idx < B), ("RG tiles", idx >= B)]
    else:
        groups = [("all tiles", idx >= 0)]
    for gname, sel in groups:
        if sel.any():
            print(f"  {gname:10s} n={sel.sum():4d}  phase durations us (median / max): " +
                  "  ".join(f"{np.median(ph[sel, j]):.2f}/{ph[sel, j].max():.2f}" for j in range(3)) +
                  f"   block total median {np.median(ph[sel].sum(1)):.2f} max {ph[sel].sum(1).max():.2f}")
