"""Developer aid: per-wave phase timeline of rgfwd2_kernel (csrc/fused_wide2.hip).   python tools/dev/dev_wide2_timeline.py B [train] [option=value ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_batches  # noqa: E402
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model  # noqa: E402

B = int(sys.argv[1])
train = "train" in sys.argv[2:]
model = build_multimodal_model({}).cuda().set_precision("bf16").train(train)
tr = NativeTrainer(model)
b0 = make_batches(1, B, 0)[0]
rg, nrs, kg = torch.from_numpy(b0[0]).cuda(), b0[1], torch.from_numpy(b0[2]).cuda()
for kv in sys.argv[2:]:
    if "=" in kv:
        _lib.lib().camo_debug_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1]))
grid = (sum(nrs) // 32 + B + 1) // 2                         # rgfwd2's grid; the KG launch's blocks stamp behind it
blocks = grid + (B + 1) // 2 + 8
NB = 16 * (1 << int(np.ceil(np.log2(blocks))))                 # stamp slots per kernel (16 per wave x 8 wave rows per block)
buf = torch.zeros(5 * NB * 8, dtype=torch.int64, device="cuda")
if train:
    yl, el, sl = (torch.from_numpy(x).cuda() for x in b0[3:])
    step = lambda: tr.step(rg, nrs, kg, yl, el, sl)
else:
    step = lambda: tr.evaluate(rg, nrs, kg)
for i in range(5):
    step()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(10):
    step()
t1.record(); torch.cuda.synchronize()
print(f"B = {B}, T = {sum(nrs)}: {'training step' if train else 'eval forward'} {t0.elapsed_time(t1) / 10 * 1e3:.1f} us per call (unstamped)")
_lib.lib().camo_debug_set_stamps(buf.data_ptr(), NB)
step()
torch.cuda.synchronize()
_lib.lib().camo_debug_set_stamps(None, 0)
raw = buf.cpu().numpy().reshape(5, NB // 16, 8, 16).astype(np.float64)[1]
st = raw[:, :4, :]                                            # the back kernel's region: [block][wave 0..3][slot]
xcc = raw[:, 4, 0] - 1                                        # the block's XCD (wave row 4, slot 0)
kgst = raw[grid:grid + (B + 1) // 2, :4, :]
st = st[:grid]; xcc = xcc[:grid]
names = {1: "input tile in LDS (barrier)", 2: "projection + R tile (barrier)", 3: "pass k2 MFMAs", 4: "scores / exp + pass v2 MFMAs", 5: "KG partial stores",
         6: "pass q MFMAs", 7: "RG->KG attention + strip", 8: "barrier (strips complete) + tickets", 9: "out-projection MFMAs", 10: "LayerNorm (one exchange) + Y barrier",
         12: "pool MFMAs + FFN (2 passes) + pooled atomics"}
act = (st[:, 0, 0] > 0) & (st[:, 0, 12] > 0)
s = st[act]
tz = s[:, :, 0].min()
end = s[:, :, 12].max(axis=1)
kgm = np.zeros(len(s), bool)
start = s[:, :, 0].min(axis=1)
print(f"--- rgfwd2: {act.sum()} blocks, span {(end.max() - tz) / 100:.2f} us; block time median {np.median(end - start) / 100:.2f} max {(end - start).max() / 100:.2f}; "
      f"blocks that ran a KG chain: {int(kgm.sum())}")
prev = 0
for slot in sorted(names):
    if slot > 12 and not kgm.any():
        continue
    sel = s[kgm] if slot > 12 else s
    d = (sel[:, :, slot] - sel[:, :, prev]) / 100
    rel = (sel[:, :, slot] - sel[:, :, 0].min(axis=1, keepdims=True)) / 100
    print(f"   {slot:2d} {names[slot]:46s} wave-median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} | reached at median {np.median(rel):6.2f}")
    prev = slot
ok = (s[:, :, 15] > 0) & (s[:, :, 11] > 0) & (s[:, :, 12] > s[:, :, 0])
mhz = (s[:, :, 15] - s[:, :, 11])[ok] / ((s[:, :, 12] - s[:, :, 0])[ok] / 100.0)
print(f"   shader clock over a block (s_memtime ticks per us of s_memrealtime): median {np.median(mhz):.0f}, min {mhz.min():.0f}, max {mhz.max():.0f}")
pts = np.linspace(tz, end.max(), 12)[1:-1]
print("   running blocks over the span:", [int(((start <= p) & (end > p)).sum()) for p in pts])
k = kgst[kgst[:, 0, 12] > 0]
if len(k):
    ks, ke = k[:, :, 0].min(axis=1), k[:, :, 12].max(axis=1)
    print(f"--- kgchain: {len(k)} blocks, first start {(ks.min() - end.max()) / 100:.2f} us after the last RG block's end, span {(ke.max() - ks.min()) / 100:.2f} us; "
          f"block time median {np.median(ke - ks) / 100:.2f} max {(ke - ks).max() / 100:.2f}; combine (2 samples) median {np.median(k[:, :, 13] - k[:, :, 0]) / 100:.2f}, chain median {np.median(k[:, :, 12] - k[:, :, 13]) / 100:.2f}")

x = xcc[act]
dur = (end - start) / 100
print("   per XCD: blocks, median / p90 block time, KG-chain blocks, first start, last end (us from the first stamp)")
for i in sorted(set(x.astype(int))):
    sel = x == i
    print(f"     XCD {i}: {int(sel.sum()):5d}  {np.median(dur[sel]):6.2f} / {np.percentile(dur[sel], 90):6.2f}   {int((kgm & sel).sum()):4d}   {(start[sel].min() - tz) / 100:8.2f}  {(end[sel].max() - tz) / 100:8.2f}")
bi = np.nonzero(act)[0]
print("   block index -> XCD of the first 24 blocks:", [int(v) for v in x[:24]])
late = np.argsort(start)[-8:]
print("   the 8 blocks that started last: index", bi[late].tolist(), "XCD", x[late].astype(int).tolist(), "start", np.round((start[late] - tz) / 100, 1).tolist())
order = np.argsort(bi)
ds = np.diff(start[order]) / 100
print(f"   start-time gaps between consecutive block indices: median {np.median(ds):.3f} us, p99 {np.percentile(ds, 99):.2f}, max {ds.max():.2f}")

# ---- the backward's first half on 64-row half-blocks (bwd_wide2.hip): region 2 of the stamp buffer, the same [block][wave][slot] layout
if train:
    rawb = buf.cpu().numpy().reshape(5, NB // 16, 8, 16).astype(np.float64)[2]
    sb = rawb[:grid, :4, :]
    actb = (sb[:, 0, 0] > 0) & (sb[:, 0, 12] > 0)
    if actb.any():
        sb = sb[actb]
        bnames = {1: "tables + dH tile in LDS (2 barriers)", 2: "dH16 rows out + dY MFMAs (K = 512)", 3: "LayerNorm backward + column-sum MFMAs (2 barriers)",
                  4: "dO MFMAs (K = 256) (barrier)", 5: "dU16 rows out (barrier)", 12: "RG->KG attention backward, 2 sub-tiles"}
        tzb = sb[:, :, 0].min(); endb = sb[:, :, 12].max(axis=1); startb = sb[:, :, 0].min(axis=1)
        print(f"--- bwd1w: {actb.sum()} blocks, span {(endb.max() - tzb) / 100:.2f} us; block time median {np.median(endb - startb) / 100:.2f} max {(endb - startb).max() / 100:.2f}")
        prev = 0
        for slot in sorted(bnames):
            d = (sb[:, :, slot] - sb[:, :, prev]) / 100
            rel = (sb[:, :, slot] - sb[:, :, 0].min(axis=1, keepdims=True)) / 100
            print(f"   {slot:2d} {bnames[slot]:52s} wave-median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} | reached at median {np.median(rel):6.2f}")
            prev = slot
        pts = np.linspace(tzb, endb.max(), 12)[1:-1]
        print("   running blocks over the span:", [int(((startb <= p) & (endb > p)).sum()) for p in pts])
