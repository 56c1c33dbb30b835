"""Developer aid: eval-forward time per call at several batch sizes under the values of one schedule option.
  python tools/dev/dev_ab_eval.py wide2 0,1 24 32 48 64"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from camouflage_multimodal_amd import NativeTrainer, build_multimodal_model, _lib
name, values, sizes = sys.argv[1], [int(v) for v in sys.argv[2].split(",")], [int(x) for x in sys.argv[3:]]
dev = torch.device("cuda", 0)
m = build_multimodal_model({}).to(dev).set_precision("bf16").eval(); tr = NativeTrainer(m)
for B in sizes:
    host = bench.make_batches(2, B, 0)
    bt = [(torch.from_numpy(rg).to(dev), nrs, torch.from_numpy(kg).to(dev)) for rg, nrs, kg, y, e, s in host]
    for v in values:
        m._engine.set_option(name, v)
        for i in range(10): tr.evaluate(*bt[i % 2])
        torch.cuda.synchronize()
        n = 300 if B <= 128 else 60
        t0 = time.perf_counter()
        for i in range(n): tr.evaluate(*bt[i % 2])
        torch.cuda.synchronize()
        print("B", B, "T", sum(bt[0][1]), name, v, "us/call", round((time.perf_counter() - t0) / n * 1e6, 1), flush=True)
m._engine.set_option(name, -1)
