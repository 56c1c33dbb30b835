"""Developer aid: forward-only (eval) and training-step time of the 32-row and the wide-tile kernels, interleaved in one process.
   python tools/dev/dev_wide_bench.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_batches  # noqa: E402
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model  # noqa: E402

Bs = [int(x) for x in sys.argv[1:]] or [16, 64, 256]
model = build_multimodal_model({}).cuda().set_precision("bf16")
tr = NativeTrainer(model)
opt = lambda k, v: _lib.check(_lib.lib().camo_debug_set_option(k.encode(), v), k)
for B in Bs:
    hb = make_batches(2, B, 0, seed=100 + B)
    db = [tuple(torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in b) for b in hb]
    res = {}
    for rnd in range(3):
        for rt in (0, 2, 4):
            opt("fused_rt", rt)
            for mode in ("fwd", "step"):
                model.train(mode == "step")
                fn = (lambda i: tr.evaluate(*db[i % 2][:3])) if mode == "fwd" else (lambda i: tr.step(*db[i % 2]))
                for i in range(3):
                    fn(i)
                torch.cuda.synchronize()
                n = 20
                t0 = time.perf_counter()
                for i in range(n):
                    fn(i)
                torch.cuda.synchronize()
                res.setdefault((rt, mode), []).append((time.perf_counter() - t0) / n * 1e3)
    print(f"B = {B}: " + "   ".join(f"rt{rt} {mode} {min(v):.4f} ms" for (rt, mode), v in sorted(res.items())), flush=True)
opt("fused_rt", -1)
