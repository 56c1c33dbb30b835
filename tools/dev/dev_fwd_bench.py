"""Developer aid: time the inference forward (eval mode, no attention maps) at several batch sizes.
   python tools/dev/dev_fwd_bench.py [B ...]      (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import FWDBWD_OVER_FWD, algorithmic_flops, make_batches  # noqa: E402
from camouflage_multimodal_amd import NativeTrainer, build_multimodal_model  # noqa: E402

Bs = [int(x) for x in sys.argv[1:]] or [1, 16, 64, 256]
from camouflage_multimodal_amd import _lib  # noqa: E402
torch.manual_seed(0)
model = build_multimodal_model({}).cuda().set_precision("bf16").eval()
tr = NativeTrainer(model)
for variant, B in [(v, B) for B in Bs for v in (0, 1, 2)]:
    _lib.lib().camo_debug_set_option(b"fused_variant", variant)
    hb = make_batches(2, B, 0, seed=B)
    db = [(torch.from_numpy(rg).cuda(), nrs, torch.from_numpy(kg).cuda()) for rg, nrs, kg, *_ in hb]
    for i in range(10):
        tr.evaluate(*db[i % 2])
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        tr.evaluate(*db[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = np.mean([algorithmic_flops(b[1]) for b in hb]) / FWDBWD_OVER_FWD
    print(f"variant {variant} B={B:4d}  {dt * 1e6:8.1f} us/call  {B / dt:10.0f} img/s  {fl / dt / 1e12:7.1f} TFLOP/s fwd ({fl / dt / 2.5e15 * 100:.1f} % of bf16 peak)")
