"""Developer aid: A/B of a camo_debug_set_option switch on the training step (bf16, dropout 0.3): ms/step at several batch sizes
and the loss trajectory under both settings (they must agree to the run-to-run noise of the fp32 atomics).
  python tools/dev/dev_ab_option.py tn_pipe4 [batch sizes ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, time
from camouflage_multimodal_amd import NativeTrainer, build_multimodal_model, _lib
name = sys.argv[1].encode()
sizes = [int(x) for x in sys.argv[2:]] or [16, 64, 4]
dev = torch.device("cuda", 0)
L = _lib.lib()
for B in sizes:
    host = bench.make_batches(4, B, 0)
    bt = [(torch.from_numpy(rg).to(dev), nrs, torch.from_numpy(kg).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(e).to(dev), torch.from_numpy(s).to(dev)) for rg, nrs, kg, y, e, s in host]
    for opt in [int(x) for x in os.environ.get('AB_VALUES', '0,1,0,1').split(',')]:
        _lib.check(L.camo_debug_set_option(name, opt), "camo_debug_set_option")
        torch.manual_seed(0)
        m = build_multimodal_model({}).to(dev).set_precision("bf16").train(); tr = NativeTrainer(m)
        losses = [float(tr.step(*bt[i % 4])[0].sum()) for i in range(12)]
        for i in range(20): tr.step(*bt[i % 4])
        torch.cuda.synchronize()
        n = 1500 if B <= 16 else (400 if B <= 64 else 100)
        t0 = time.perf_counter()
        for i in range(n): tr.step(*bt[i % 4])
        torch.cuda.synchronize()
        print("B", B, name.decode(), opt, "ms/step", round((time.perf_counter() - t0) / n * 1e3, 4), "losses", [round(x, 4) for x in losses[:3] + losses[-2:]], flush=True)
_lib.check(L.camo_debug_set_option(name, 1), "camo_debug_set_option")
print("tail timeouts", _lib.tail_timeouts())
