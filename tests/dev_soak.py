"""Developer aid: a few hundred optimizer steps on repeated synthetic batches (loss must fall, parameters stay finite),\nat B = 128 and B = 16.  python tests/dev_soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from camouflage_multimodal_amd import NativeTrainer, build_multimodal_model
dev = torch.device("cuda", 0)
for B in (128, 16):
    torch.manual_seed(0)
    model = build_multimodal_model({}).to(dev).set_precision("bf16").train()
    tr = NativeTrainer(model)
    host = bench.make_batches(4, B, 0)
    bt = [(torch.from_numpy(rg).to(dev), nrs, torch.from_numpy(kg).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(e).to(dev), torch.from_numpy(s).to(dev)) for rg, nrs, kg, y, e, s in host]
    losses = []
    steps = 400 if B == 16 else 40
    for i in range(steps):
        terms, pred = tr.step(*bt[i % 4])
        if i % (steps // 8) == 0 or i == steps - 1:
            losses.append(float(terms.sum().item()) / B)
    torch.cuda.synchronize()
    p = model._engine.flat_params
    print(f"B={B}: loss/sample over time {['%.3f' % l for l in losses]}; params finite: {bool(torch.isfinite(p).all())}; grad_norm {float(tr.opt.grad_norm()[0]):.3f}")
    t0 = time.perf_counter()
    for i in range(50): tr.step(*bt[i % 4])
    torch.cuda.synchronize()
    print(f"   {B * 50 / (time.perf_counter() - t0):.0f} images/s at B={B}")
