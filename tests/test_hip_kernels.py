"""Kernel-level checks of the grouped MFMA GEMM through the C ABI's testing hook."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RELU, ATOMIC, AKM, BKM = 1, 4, 64, 128


def run_gemm(A, B, M, N, K, flags, prec, bias=None, res=None, c0=None, want_bias_grad=False):
    from camouflage_multimodal_amd import _lib
    L = _lib.lib()
    dev = "cuda"
    a = torch.from_numpy(A).to(dev); b = torch.from_numpy(B).to(dev)
    c = torch.from_numpy(c0).to(dev) if c0 is not None else torch.full((M, N), float("nan"), device=dev)
    bt = torch.from_numpy(bias).to(dev) if bias is not None else None
    rt = torch.from_numpy(res).to(dev) if res is not None else None
    bg = torch.zeros(M, device=dev) if want_bias_grad else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    rc = L.camo_debug_gemm(p(a), A.shape[1], p(b), B.shape[1], p(c), N, p(bt), p(rt), N, p(bg), M, N, K, flags, prec,
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "camo_debug_gemm")
    torch.cuda.synchronize()
    return c.cpu().numpy(), (bg.cpu().numpy() if bg is not None else None)


# error bound per output element: tol * sum_k |a_k||b_k|  (fp32: rounding of the running sum;
# bf16: 2^-9 relative rounding of each operand -> ~2^-8 per product)
@pytest.mark.parametrize("prec,tol", [(0, 2e-6), (1, 8e-3)])
@pytest.mark.parametrize("M,N,K", [(64, 128, 32), (1, 1, 1), (70, 130, 37), (200, 256, 128), (16, 2, 128), (500, 512, 256), (33, 6, 2)])
def test_gemm_layouts(M, N, K, prec, tol):
    rs = np.random.RandomState(M * 7 + N * 3 + K)
    for flags in (0, BKM, AKM | BKM | ATOMIC):
        akm, bkm = bool(flags & AKM), bool(flags & BKM)
        Am = rs.standard_normal((M, K)).astype(np.float32)          # logical A [M,K]
        Bm = rs.standard_normal((K, N)).astype(np.float32)          # logical B [K,N]
        # asymmetric integer-ish structure catches transposed fragment maps
        Am += (np.arange(M)[:, None] % 5) * 0.25; Bm += (np.arange(N)[None, :] % 3) * 0.5
        A = np.ascontiguousarray(Am.T if akm else Am)
        B = np.ascontiguousarray(Bm if bkm else Bm.T)
        bias = rs.standard_normal(N).astype(np.float32) if not akm else None
        res = rs.standard_normal((M, N)).astype(np.float32) if not akm else None
        c0 = rs.standard_normal((M, N)).astype(np.float32) if flags & ATOMIC else None
        want = Am.astype(np.float64) @ Bm.astype(np.float64)
        if bias is not None: want = want + bias
        if res is not None: want = want + res
        if c0 is not None: want = want + c0
        got, bg = run_gemm(A, B, M, N, K, flags, prec, bias, res, c0, want_bias_grad=akm)
        bound = tol * (np.abs(Am).astype(np.float64) @ np.abs(Bm).astype(np.float64)) + 1e-6
        if bias is not None: bound = bound + 1e-6 * (np.abs(bias) + np.abs(res))
        if c0 is not None: bound = bound + 1e-6 * np.abs(c0)
        viol = np.abs(got - want) - bound
        assert viol.max() <= 0, f"flags={flags} M={M} N={N} K={K} prec={prec}: err {np.abs(got - want).max()} bound {bound.max()}"
        if akm:
            assert (np.abs(bg - Am.sum(1)) <= max(tol, 1e-5) * np.abs(Am).sum(1) + 1e-5).all()


def test_gemm_relu_and_split_k():
    rs = np.random.RandomState(3)
    M, N, K = 256, 256, 4000                      # weight-gradient shape: split-K + atomics
    Am = rs.standard_normal((M, K)).astype(np.float32); Bm = rs.standard_normal((K, N)).astype(np.float32)
    got, bg = run_gemm(np.ascontiguousarray(Am.T), Bm, M, N, K, AKM | BKM | ATOMIC, 0, c0=np.zeros((M, N), np.float32), want_bias_grad=True)
    want = Am.astype(np.float64) @ Bm.astype(np.float64)
    assert np.abs(got - want).max() < 2e-3
    assert np.abs(bg - Am.sum(1)).max() < 1e-3
    x = rs.standard_normal((100, 64)).astype(np.float32); w = rs.standard_normal((96, 64)).astype(np.float32)
    got, _ = run_gemm(x, w, 100, 96, 64, RELU, 0)
    assert np.abs(got - np.maximum(x.astype(np.float64) @ w.T, 0)).max() < 1e-4
