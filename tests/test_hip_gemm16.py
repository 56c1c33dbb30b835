"""The bf16-resident grouped GEMM (csrc/gemm16.hip) through camo_debug_gemm16 against fp64 numpy on the
same bf16-rounded operands.  Tolerance: fp32 accumulation of exact bf16 products -> 2e-6 * sum|a||b| per
element (order-of-summation noise only); the bf16 copy of the output within one bf16 ulp (2^-8 relative)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

AKM, BKM, RELU = 64, 128, 1


def _lib():
    from camouflage_multimodal_amd import _lib
    return _lib, _lib.lib()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _bf(x):
    return torch.from_numpy(x).cuda().to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(7219, 256, 128), (500, 512, 256), (64, 128, 64), (1, 256, 768), (333, 64, 128), (208, 768, 256)])
def test_nt_matches_numpy_with_epilogue_and_bf16_copy(M, N, K):
    L_, L = _lib()
    rs = np.random.RandomState(M + N + K)
    a = _bf(rs.standard_normal((M, K)).astype(np.float32)); b = _bf(rs.standard_normal((N, K)).astype(np.float32))
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).cuda()
    res = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).cuda()
    c = torch.full((M, N), 7.0, device="cuda"); c16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_.check(L.camo_debug_gemm16(_p(a), K, _p(b), K, _p(c), N, _p(c16), N, _p(bias), _p(res), N, None, M, N, K, RELU, st), "gemm16")
    torch.cuda.synchronize()
    an, bn = a.float().cpu().numpy().astype(np.float64), b.float().cpu().numpy().astype(np.float64)
    want = np.maximum(an @ bn.T + bias.cpu().numpy(), 0.0) + res.cpu().numpy()
    bound = 2e-6 * (np.abs(an) @ np.abs(bn).T + 1.0) + 1e-6 * np.abs(want)
    got = c.cpu().numpy()
    assert np.all(np.abs(got - want) <= bound), float(np.abs(got - want).max())
    got16 = c16.float().cpu().numpy()
    assert np.all(np.abs(got16 - got) <= np.abs(got) * 2.0 ** -8 + 1e-30)


@pytest.mark.parametrize("M,N,K", [(256, 128, 7219), (512, 256, 1000), (256, 256, 208), (64, 64, 13), (768, 256, 130)])
def test_tn_accumulates_weight_gradient_and_bias_gradient(M, N, K):
    L_, L = _lib()
    rs = np.random.RandomState(M + N + K)
    Kp = (K + 127) // 128 * 128
    a = torch.zeros(Kp, M, device="cuda", dtype=torch.bfloat16); b = torch.zeros(Kp, N, device="cuda", dtype=torch.bfloat16)
    a[:K] = _bf(rs.standard_normal((K, M)).astype(np.float32)); b[:K] = _bf(rs.standard_normal((K, N)).astype(np.float32))
    b[K:] = 3.0                                  # pad rows: zero in one operand, finite in the other
    c0 = rs.standard_normal((M, N)).astype(np.float32); g0 = rs.standard_normal(M).astype(np.float32)
    c = torch.from_numpy(c0).cuda(); bg = torch.from_numpy(g0).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L_.check(L.camo_debug_gemm16(_p(a), M, _p(b), N, _p(c), N, None, 0, None, None, 0, _p(bg), M, N, K, AKM | BKM | 4, st), "gemm16 tn")
    torch.cuda.synchronize()
    an, bn = a[:K].float().cpu().numpy().astype(np.float64), b[:K].float().cpu().numpy().astype(np.float64)
    want = c0 + an.T @ bn
    bound = 4e-6 * (np.abs(an).T @ np.abs(bn) + np.abs(c0) + 1.0)
    assert np.all(np.abs(c.cpu().numpy() - want) <= bound), float(np.abs(c.cpu().numpy() - want).max())
    wantg = g0 + an.sum(0)
    assert np.all(np.abs(bg.cpu().numpy() - wantg) <= 4e-6 * (np.abs(an).sum(0) + np.abs(g0) + 1.0))


def test_unsupported_problems_are_rejected_not_miscomputed():
    L_, L = _lib()
    a = torch.zeros(64, 96, device="cuda", dtype=torch.bfloat16); b = torch.zeros(64, 96, device="cuda", dtype=torch.bfloat16)
    c = torch.zeros(64, 64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.camo_debug_gemm16(_p(a), 96, _p(b), 96, _p(c), 64, None, 0, None, None, 0, None, 64, 64, 96, 0, st) != 0     # K % 64
    assert L.camo_debug_gemm16(_p(a), 96, _p(b), 96, _p(c), 64, None, 0, None, None, 0, None, 64, 64, 64, AKM, st) != 0   # mixed layout
