"""Developer aid: host-side enqueue time per training step vs GPU time.  python tools/dev/dev_host_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from camouflage_multimodal_amd import NativeTrainer, build_multimodal_model
dev = torch.device("cuda", 0)
model = build_multimodal_model({}).to(dev).set_precision("bf16").train()
tr = NativeTrainer(model)
host = bench.make_batches(8, 16, 0)
bt = [(torch.from_numpy(rg).to(dev), nrs, torch.from_numpy(kg).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(e).to(dev), torch.from_numpy(s).to(dev)) for rg, nrs, kg, y, e, s in host]
for i in range(20): tr.step(*bt[i % 8])
torch.cuda.synchronize()
for n in (10, 50, 200):
    t0 = time.perf_counter()
    for i in range(n): tr.step(*bt[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
# pieces
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(200): tr.step(*bt[i % 8])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
