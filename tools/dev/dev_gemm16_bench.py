"""Developer aid: times camo_debug_gemm16 on the step's shapes.  python tools/dev/dev_gemm16_bench.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from camouflage_multimodal_amd import _lib
L = _lib.lib()
AKM, BKM, ATOMIC = 64, 128, 4
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def bench(M, N, K, tn, iters=30, out16=False):
    Kp = (K + 127) // 128 * 128
    if tn:
        A = [torch.randn(Kp, M, device="cuda").to(torch.bfloat16) for _ in range(4)]
        B = [torch.randn(Kp, N, device="cuda").to(torch.bfloat16) for _ in range(4)]
        lda, ldb = M, N
    else:
        A = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(4)]
        B = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(4)]
        lda, ldb = K, K
    Cm = torch.zeros(M, N, device="cuda")
    C16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if out16 else None
    fl = (AKM | BKM | ATOMIC) if tn else 0
    run = lambda i: _lib.check(L.camo_debug_gemm16(p(A[i % 4]), lda, p(B[i % 4]), ldb, p(Cm), N, p(C16), N, None, None, 0, None, M, N, K, fl, st()), "g16")
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{'TN' if tn else 'NT'} M={M:6d} N={N:5d} K={K:5d}{' +bf16 out' if out16 else ''}: {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")

bench(7700, 256, 128, False)
bench(7700, 768, 256, False)
bench(7700, 256, 256, False)
bench(7700, 256, 256, False, out16=True)
bench(7700, 512, 256, False)
bench(7700, 256, 512, False)
bench(7700, 256, 768, False)
bench(512, 256, 7700, True)
bench(256, 256, 7700, True)
bench(256, 128, 7700, True)
bench(768, 256, 7700, True)
