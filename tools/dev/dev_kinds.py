"""Developer aid: HIP-event time per kernel family of an eval forward / training step.   python tools/dev/dev_kinds.py B rt [train] [option=value ...]"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_batches
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model
B, rt = int(sys.argv[1]), int(sys.argv[2]); train = len(sys.argv) > 3 and sys.argv[3] == "train"
cfg = {"dropout": float([a.split("=")[1] for a in sys.argv if a.startswith("dropout=")][0])} if any(a.startswith("dropout=") for a in sys.argv) else {}
model = build_multimodal_model(cfg).cuda().set_precision("bf16"); model.train(train)
tr = NativeTrainer(model)
L = _lib.lib()
_lib.check(L.camo_debug_set_option(b"fused_rt", rt), "opt")
for kv in sys.argv[3:]:
    if "=" in kv and not kv.startswith("dropout="):
        _lib.check(L.camo_debug_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])), kv)
hb = make_batches(2, B, 0, seed=100 + B)
db = [tuple(torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in b) for b in hb]
fn = (lambda i: tr.step(*db[i % 2])) if train else (lambda i: tr.evaluate(*db[i % 2][:3]))
for i in range(5): fn(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): fn(i)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e6
k = 20
_lib.check(L.camo_prof_begin(64 * k), "begin")
for i in range(k): fn(i)
torch.cuda.synchronize()
ms, n, fl = C.c_double(), C.c_int32(), C.c_double()
_lib.check(L.camo_prof_end(C.byref(ms), C.byref(n), C.byref(fl)), "end")
names = ["gemm", "front", "back/rgfwd", "bwd1", "bwd2", "tail", "opt", "shadow", "attn", "other"]
print(f"B={B} rt={rt} {'train' if train else 'eval'}: wall {wall:.1f} us per call; event-timed launches {n.value / k:.1f} per call, sum {ms.value * 1e3 / k:.1f} us")
for kind, nm in enumerate(names):
    kms, kn, kfl = C.c_double(), C.c_int32(), C.c_double()
    _lib.check(L.camo_prof_kind(kind, C.byref(kms), C.byref(kn), C.byref(kfl)), "kind")
    if kn.value: print(f"   {nm:12s} {kn.value / k:5.1f} launches  {kms.value * 1e3 / k:8.1f} us per call  ({kms.value * 1e3 / kn.value:.1f} us each)")
