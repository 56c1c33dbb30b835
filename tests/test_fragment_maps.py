"""Executable statement of the data layouts the fused row-tile kernels (csrc/fused_rows.hip) rely on, checked on the CPU
against plain matrix arithmetic with a lane-level model of the gfx950 MFMA / transposing-LDS-read semantics
(cdna_hip_programming.md section 3: A[i = l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col = l&31], C/D row = (r&3)+8(r>>2)+4(l>>5),
col = l&31; T10: ds_read_b64_tr_b16).

Nothing here runs product code; the maps are mirrored in Python (``frag_order``) so that the GPU tests can check the
weight-shadow kernel against them, and so that a reader can see why each operand is laid out the way it is.
"""
import numpy as np

LANES = np.arange(64)
L31, LH = LANES & 31, LANES >> 5


def mfma_32x32x16(a, b, c=None):
    """a, b: [64 lanes][8] operand fragments; c: [64][16] accumulator -> d [64][16]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a[l]
        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b[l]
    D = A @ B
    d = np.zeros((64, 16))
    for l in range(64):
        for r in range(16):
            d[l, r] = D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return d if c is None else c + d


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def frag_order(W, nwaves=4):
    """Weight shadow of a stage: W [N][K] -> flat array in the order wave w streams it: [w][ks][t][lane][8]."""
    N, K = W.shape
    ntw, KS = N // 32 // nwaves, K // 16
    out = np.zeros((nwaves, KS, ntw, 64, 8), W.dtype)
    for w in range(nwaves):
        for ks in range(KS):
            for t in range(ntw):
                for l in range(64):
                    out[w, ks, t, l] = W[32 * (w * ntw + t) + (l & 31), 16 * ks + 8 * (l >> 5):16 * ks + 8 * (l >> 5) + 8]
    return out.reshape(-1)


def act_frag(X, ks):
    """ds_read_b128 of an activation tile X [32 rows][K]: lane (row l&31, half h) takes k = 16 ks + 8 h .. +7."""
    return np.stack([X[l & 31, 16 * ks + 8 * (l >> 5):16 * ks + 8 * (l >> 5) + 8] for l in range(64)])


def test_transposed_and_plain_stage_orientations():
    rs = np.random.RandomState(0)
    N, K = 256, 128
    W = rs.standard_normal((N, K)); X = rs.standard_normal((32, K))
    Wf = frag_order(W).reshape(4, K // 16, N // 128, 64, 8)
    ref = X @ W.T                                                   # [32 rows][N]
    for w in range(4):
        for t in range(N // 128):
            accT = np.zeros((64, 16)); acc = np.zeros((64, 16))
            for ks in range(K // 16):
                accT = mfma_32x32x16(Wf[w, ks, t], act_frag(X, ks), accT)      # weights as A: out^T, lane = row
                acc = mfma_32x32x16(act_frag(X, ks), Wf[w, ks, t], acc)        # activations as A: out, lane = feature
            f0 = 32 * (w * (N // 128) + t)
            for l in range(64):
                for r in range(16):
                    assert abs(accT[l, r] - ref[l & 31, f0 + acc_row(r, l >> 5)]) < 1e-9
                    assert abs(acc[l, r] - ref[acc_row(r, l >> 5), f0 + (l & 31)]) < 1e-9


def tr_read(img, row0, col0, lane):
    """ds_read_b64_tr_b16 of a [rows][cols] 16-bit image as the kernels address it: the lane's 16-lane group reads the 4x16
    block at (row0, col0 + 16*((lane>>4)&1)); lane i of the group receives column i of the block, rows 0..3."""
    c = col0 + 16 * ((lane >> 4) & 1) + (lane & 15)
    return img[row0:row0 + 4, c]


def test_rg2kg_attention_chain_in_one_wave():
    """S^T = K_h . Q_h^T (lane = RG row), softmax over the <= 16 keys held in registers 0..7 of the two lane halves, P^T as
    the B operand of O^T = V_h^T . P^T with V_h^T fragments from transposing reads of the row-major [key][feature] image."""
    rs = np.random.RandomState(1)
    Nk = 13
    Q = rs.standard_normal((32, 32)); Kh = np.zeros((16, 32)); Vh = np.zeros((16, 32))
    Kh[:Nk] = rs.standard_normal((Nk, 32)); Vh[:Nk] = rs.standard_normal((Nk, 32))
    S = np.zeros((64, 16))
    for s in range(2):
        a = np.stack([Kh[(l & 31) & 15, 16 * s + 8 * (l >> 5):16 * s + 8 * (l >> 5) + 8] for l in range(64)])   # lanes j >= 16 read duplicates
        S = mfma_32x32x16(a, act_frag(Q, s), S)
    P = np.zeros((64, 8))
    for l in range(64):
        j = np.array([acc_row(i, l >> 5) for i in range(8)])
        valid = j < Nk
        mine = np.where(valid, S[l, :8], -np.inf)
        other = np.where(np.array([acc_row(i, 1 - (l >> 5)) for i in range(8)]) < Nk, S[l ^ 32, :8], -np.inf)
        m = max(mine.max(), other.max())
        e = np.where(valid, np.exp(mine - m), 0.0); eo = np.exp(other - m)
        P[l] = e / (e.sum() + eo.sum())
    Pref = np.exp(Q @ Kh[:Nk].T); Pref /= Pref.sum(1, keepdims=True)
    for l in range(64):
        for i in range(8):
            j = acc_row(i, l >> 5)
            assert abs(P[l, i] - (Pref[l & 31, j] if j < Nk else 0.0)) < 1e-12
    # O^T = V^T . P^T: P's registers 0..7 are the B fragment of ONE k step; the A fragment (lane = feature) takes its
    # element jj from key 8 (jj>>2) + 4 h + (jj&3): two transposing reads at key rows 4h and 8 + 4h
    a = np.zeros((64, 8))
    for l in range(64):
        h = l >> 5
        a[l, :4] = tr_read(Vh, 4 * h, 0, l)
        a[l, 4:] = tr_read(Vh, 8 + 4 * h, 0, l)
        assert (l & 31) == 16 * ((l >> 4) & 1) + (l & 15)          # the column a lane receives is its feature
    O = mfma_32x32x16(a, P)
    Oref = Pref @ Vh[:Nk]                                           # [row][feature]
    for l in range(64):
        for r in range(16):
            assert abs(O[l, r] - Oref[l & 31, acc_row(r, l >> 5)]) < 1e-9


def test_kg2rg_attention_chunk_lane_is_query():
    """KG->RG direction: S[row][j] with the key ROWS in the registers (lane = query j), so max / sum over keys are in-lane;
    the exponentials go back in as the A operand (X^T . B) against V2 fragments from transposing reads."""
    rs = np.random.RandomState(2)
    K2 = 0.3 * rs.standard_normal((32, 32)); V2 = rs.standard_normal((32, 32)); Q2 = 0.3 * rs.standard_normal((32, 32))   # 32 keys, 32 queries (13 used)
    S = np.zeros((64, 16))
    for s in range(2):
        S = mfma_32x32x16(act_frag(K2, s), act_frag(Q2, s), S)      # A = key rows, B = queries: D[row][j]
    Sref = K2 @ Q2.T
    for l in range(64):
        for r in range(16):
            assert abs(S[l, r] - Sref[acc_row(r, l >> 5), l & 31]) < 1e-9
    E = np.exp(S - 1.0)
    Z = np.zeros((64, 16))
    for s in range(2):
        b = np.zeros((64, 8))
        for l in range(64):
            h = l >> 5
            b[l, :4] = tr_read(V2, 16 * s + 4 * h, 0, l)
            b[l, 4:] = tr_read(V2, 16 * s + 8 + 4 * h, 0, l)
        Z = mfma_32x32x16(E[:, 8 * s:8 * s + 8], b, Z)              # registers 8s..8s+7 of the accumulator as the A fragment
    Zref = np.exp(Sref - 1.0).T @ V2                                # [j][feature]
    for l in range(64):
        for r in range(16):
            assert abs(Z[l, r] - Zref[acc_row(r, l >> 5), l & 31]) < 1e-9


def test_relu_mask_words_from_ballots():
    """FFN activation tile in the plain orientation (lane = feature, registers = rows): one 64-lane ballot per register is
    two 32-bit words of the [row][512-bit] mask image -- rows acc_row(r, 0) and acc_row(r, 1), word index = feature / 32."""
    rs = np.random.RandomState(3)
    Hact = rs.standard_normal((32, 32)) > 0                        # [row][feature] of one 32-feature tile
    words = np.zeros(32, np.uint32)
    for r in range(16):
        bal = 0
        for l in range(64):
            if Hact[acc_row(r, l >> 5), l & 31]:
                bal |= 1 << l
        words[acc_row(r, 0)] = bal & 0xFFFFFFFF
        words[acc_row(r, 1)] = bal >> 32
    for row in range(32):
        for f in range(32):
            assert bool((int(words[row]) >> f) & 1) == bool(Hact[row, f])
