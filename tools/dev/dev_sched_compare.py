"""Developer aid: fused row-tile schedule vs the GEMM-per-layer bf16 schedule across batch sizes.
  python tools/dev/dev_sched_compare.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model
dev = torch.device("cuda", 0)
L = _lib.lib()
for B in (16, 32, 64, 128, 256):
    host = bench.make_batches(4, B, 0)
    bt = [(torch.from_numpy(rg).to(dev), nrs, torch.from_numpy(kg).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(e).to(dev), torch.from_numpy(s).to(dev)) for rg, nrs, kg, y, e, s in host]
    out = []
    for fused in (1, 0):
        L.camo_debug_set_option(b"fused", fused)
        torch.manual_seed(0)
        model = build_multimodal_model({}).to(dev).set_precision("bf16").train()
        tr = NativeTrainer(model)
        for i in range(10): tr.step(*bt[i % 4])
        torch.cuda.synchronize()
        n = max(20, 2000 // B)
        t0 = time.perf_counter()
        for i in range(n): tr.step(*bt[i % 4])
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    print(f"B={B:4d}: fused {out[0]:.4f} ms/step ({B / out[0] * 1e3:.0f} img/s)   gemm-per-layer {out[1]:.4f} ms/step ({B / out[1] * 1e3:.0f} img/s)")
L.camo_debug_set_option(b"fused", 1)
