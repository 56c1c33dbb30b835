"""Developer aid: per-wave phase timeline of the wide-tile forward kernels.   python tools/dev/dev_wide_timeline.py B rt [train]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_batches  # noqa: E402
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model  # noqa: E402

B = int(sys.argv[1]); rt = int(sys.argv[2]); train = len(sys.argv) > 3 and sys.argv[3] == "train"
model = build_multimodal_model({}).cuda().set_precision("bf16").eval()
tr = NativeTrainer(model)
b0 = make_batches(1, B, 0)[0]
rg, nrs, kg = torch.from_numpy(b0[0]).cuda(), b0[1], torch.from_numpy(b0[2]).cuda()
NB = 32768
_lib.lib().camo_debug_set_option(b"fused_rt", rt)
for kv in sys.argv[3:]:
    if "=" in kv:
        _lib.lib().camo_debug_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1]))
if "--two" in sys.argv:
    _lib.lib().camo_debug_set_option(b"fused_one", 0)
buf = torch.zeros(5 * NB * 8, dtype=torch.int64, device="cuda")
if train:
    model.train()
    y, e, s_ = (torch.from_numpy(x).cuda() for x in b0[3:])
    step = lambda: tr.step(rg, nrs, kg, y, e, s_)
else:
    step = lambda: tr.evaluate(rg, nrs, kg)
for i in range(5):
    step()
_lib.lib().camo_debug_set_stamps(buf.data_ptr(), NB)
step()
torch.cuda.synchronize()
_lib.lib().camo_debug_set_stamps(None, 0)
st = buf.cpu().numpy().reshape(5, NB // 16, 8, 16).astype(np.float64)      # [kernel][block][wave][slot]
names = {"front": {1: "X in LDS (barrier)", 2: "proj + R tile (barrier)", 3: "pass 0 + stores", 4: "pass 1 + stores", 5: "pass 2 + stores", 7: "R copy-out, end"},
         "back": {1: "loads + KG partial", 2: "barrier 1", 3: "RG attention + drain", 4: "barrier 2 + tickets", 5: "out-proj MFMAs", 6: "epilogue + row total 1",
                  7: "row total 2", 8: "LN outputs + pool", 9: "barrier 3", 10: "FFN MFMAs", 11: "FFN epilogue", 12: "saves", 13: "KG combine", 14: "KG chain"}}
one = "--two" not in sys.argv
if one:
    names["back"] = {9: "X tile written", 10: "consts written", 1: "X in LDS (barrier)", 2: "proj + R tile (barrier)", 3: "pass k2 MFMAs", 4: "scores/exp + pass v2", 5: "KG partial stores", 6: "pass q MFMAs", 7: "RG attention",
                     8: "barriers + tickets", 12: "chain (out-proj .. FFN)", 13: "KG combine", 14: "KG chain"}
for k, name in enumerate(("front", "back")):
    s = st[k]
    last = 7 if name == "front" else 12
    act = (s[:, 0, 0] > 0) & (s[:, 0, last] > 0)
    s = s[act]
    t0 = s[:, :, 0].min()
    end = s[:, :, last].max(axis=1)
    kgm = s[:, 0, 14] > 0 if name == "back" else np.zeros(len(s), bool)
    end[kgm] = s[kgm][:, :, 14].max(axis=1)
    start = s[:, :, 0].min(axis=1)
    print(f"--- {name} rt={rt} B={B}: {act.sum()} blocks, span {(end.max() - t0) / 100:.2f} us; starts: median {np.median(start - t0) / 100:.2f} last {(start.max() - t0) / 100:.2f}; block time median {np.median(end - start) / 100:.2f} max {(end - start).max() / 100:.2f}")
    prev = 0
    for slot in ([9, 10] + [x for x in sorted(names[name]) if x not in (9, 10)] if (name == 'back' and one) else sorted(names[name])):
        if slot > last and not kgm.any():
            continue
        sel = s[kgm] if slot > last else s
        d = (sel[:, :, slot] - sel[:, :, prev]) / 100                      # per wave
        rel = (sel[:, :, slot] - sel[:, :, 0].min(axis=1, keepdims=True)) / 100
        print(f"   {slot:2d} {names[name][slot]:26s} wave-median {np.median(d):6.2f}  max {d.max():6.2f} | reached at (from block start) median {np.median(rel):6.2f}, slowest wave median {np.median(rel.max(axis=1)):6.2f}; wave 0 vs wave 4: {np.median(rel[:, 0]):.2f} / {np.median(rel[:, 4]):.2f}")
        prev = slot if slot not in (9, 10) else prev
    if name == "back" and one:
        ok = (s[:, :, 15] > 0) & (s[:, :, 11] > 0) & (s[:, :, 12] > s[:, :, 0])
        mhz = (s[:, :, 15] - s[:, :, 11])[ok] / ((s[:, :, 12] - s[:, :, 0])[ok] / 100.0)
        print(f"   shader clock over a block (s_memtime ticks per us of s_memrealtime): median {np.median(mhz):.0f}, min {mhz.min():.0f}, max {mhz.max():.0f}")
    pts = np.linspace(t0, end.max(), 12)[1:-1]
    print("   running blocks over the span:", [int(((start <= p) & (end > p)).sum()) for p in pts])
