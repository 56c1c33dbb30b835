"""Developer aid: forward time with experiment bits set.  python tools/dev/dev_wide_exp.py B rt exp [exp ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_batches
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model
B, rt = int(sys.argv[1]), int(sys.argv[2]); exps = [int(x) for x in sys.argv[3:]] or [0]
model = build_multimodal_model({}).cuda().set_precision("bf16").eval()
tr = NativeTrainer(model)
opt = lambda k, v: _lib.check(_lib.lib().camo_debug_set_option(k.encode(), v), k)
hb = make_batches(2, B, 0, seed=100 + B)
db = [tuple(torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in b) for b in hb]
opt("fused_rt", rt)
res = {}
for rnd in range(3):
    for e in exps:
        opt("exp", e)
        for i in range(3): tr.evaluate(*db[i % 2][:3])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): tr.evaluate(*db[i % 2][:3])
        torch.cuda.synchronize(); res.setdefault(e, []).append((time.perf_counter() - t0) / 20 * 1e3)
opt("exp", 0)
print(f"B={B} rt={rt}: " + "  ".join(f"exp{e}: {min(v):.4f} ms" for e, v in res.items()))
