"""Developer aid: optimizer steps on repeated synthetic batches (loss must fall, parameters stay finite): 1 000 at B = 128 (two-plane tail launches, parameter-space backward), 2 000 at B = 40 (grouped one-launch tail) and\n20 000 at B = 16 (the one-launch tail with its in-kernel all-reduces); reports the tail kernel's timeout counter.\n  python tools/dev/dev_soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model
dev = torch.device("cuda", 0)
for B in (128, 40, 16):          # 128: separate-launch tail, parameter-space backward; 40: grouped one-launch tail; 16: the headline step
    torch.manual_seed(0)
    model = build_multimodal_model({}).to(dev).set_precision("bf16").train()
    tr = NativeTrainer(model)
    host = bench.make_batches(4, B, 0)
    bt = [(torch.from_numpy(rg).to(dev), nrs, torch.from_numpy(kg).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(e).to(dev), torch.from_numpy(s).to(dev)) for rg, nrs, kg, y, e, s in host]
    losses = []
    steps = 20000 if B == 16 else (2000 if B == 40 else 1000)
    for i in range(steps):
        terms, pred = tr.step(*bt[i % 4])
        if i % 50 == 49:                       # a validation pass now and then: inference calls share the optimizer's weight shadows
            model.eval(); tr.evaluate(*bt[(i // 50) % 4][:3]); tr.evaluate(*bt[(i // 50 + 1) % 4][:3]); model.train()
        if i % (steps // 8) == 0 or i == steps - 1:
            losses.append(float(terms.sum().item()) / B)
    torch.cuda.synchronize()
    p = model._engine.flat_params
    print(f"B={B}: loss/sample over time {['%.3f' % l for l in losses]}; params finite: {bool(torch.isfinite(p).all())}; grad_norm {float(tr.opt.grad_norm()[0]):.3f}")
    t0 = time.perf_counter()
    for i in range(50): tr.step(*bt[i % 4])
    torch.cuda.synchronize()
    print(f"   {B * 50 / (time.perf_counter() - t0):.0f} images/s at B={B}")
print(f"tail-kernel waits that gave up in this process: {_lib.tail_timeouts()} (must be 0)")
