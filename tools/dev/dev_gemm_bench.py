"""Developer aid: times camo_debug_gemm on a few shapes.  python tools/dev/dev_gemm_bench.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from camouflage_multimodal_amd import _lib
L = _lib.lib()
AKM, BKM, ATOMIC = 64, 128, 4
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def bench(M, N, K, flags, prec, iters=20):
    akm, bkm = bool(flags & AKM), bool(flags & BKM)
    A = torch.randn((K, M) if akm else (M, K), device="cuda")
    B = torch.randn((K, N) if bkm else (N, K), device="cuda")
    Cm = torch.zeros(M, N, device="cuda")
    run = lambda: _lib.check(L.camo_debug_gemm(p(A), A.shape[1], p(B), B.shape[1], p(Cm), N, None, None, 0, None, M, N, K, flags, prec, st()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"M={M:6d} N={N:5d} K={K:5d} flags={flags:3d} prec={'bf16' if prec else 'f32 '}: {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")

import sys
for prec in (1,):
    bench(7700, 256, 128, 0, prec, 20)
    bench(7700, 768, 256, 0, prec, 20)
    bench(7700, 256, 256, 0, prec, 20)
    bench(7700, 512, 256, 0, prec, 20)
    bench(7700, 256, 512, BKM, prec, 20)
    bench(7700, 256, 256, BKM, prec, 20)
    bench(512, 256, 7700, AKM | BKM | ATOMIC, prec, 20)
    bench(256, 256, 7700, AKM | BKM | ATOMIC, prec, 20)
    bench(256, 128, 7700, AKM | BKM | ATOMIC, prec, 20)
    bench(64 * 254, 128, 32, 0, prec, 20)
    bench(64 * 254, 128, 64, 0, prec, 20)
