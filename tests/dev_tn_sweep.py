"""Developer aid: times the weight-gradient (TN, split-K + atomics) shapes of the backward under the
CAMO_DEV_TN_KCAP knob.  for c in 4 6 12 24; do CAMO_DEV_TN_KCAP=$c python tests/dev_tn_sweep.py; done"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camouflage_multimodal_amd import _lib
L = _lib.lib()
AKM, BKM, ATOMIC = 64, 128, 4
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def bench(M, N, K, flags, prec, iters=30):
    akm, bkm = bool(flags & AKM), bool(flags & BKM)
    A = [torch.randn((K, M) if akm else (M, K), device="cuda") for _ in range(4)]
    B = [torch.randn((K, N) if bkm else (N, K), device="cuda") for _ in range(4)]
    Cm = torch.zeros(M, N, device="cuda")
    run = lambda i: _lib.check(L.camo_debug_gemm(p(A[i % 4]), A[0].shape[1], p(B[i % 4]), B[0].shape[1], p(Cm), N, None, None, 0, None, M, N, K, flags, prec, st()))
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

cap = os.environ.get("CAMO_DEV_TN_KCAP", "12")
res = []
for (M, N) in ((512, 256), (256, 256), (256, 128)):
    res.append("%dx%d %.1f us" % (M, N, bench(M, N, 7200, AKM | BKM | ATOMIC, 1)))
print("kcap", cap, " | ".join(res))
