"""Pins the CPU oracle (oracle/) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import TRAIN_CASES, assert_close, assert_params_close, sub, train_batch, train_case
from oracle import fusion_oracle as FO
from oracle import params as OP


def _oracle(cfg=None, seed=0):
    cfg = OP.full_cfg(cfg)
    return FO.FusionOracle(cfg, OP.make_params(cfg, seed))


@pytest.mark.parametrize("nr", [303, 481, 500, 530])
def test_eval_forward_real_kg(nr, kg_real):
    g = load_golden(f"eval_nr{nr}")
    o = _oracle()
    outs, _ = o.forward(OP.make_rg(nr, 128, seed=nr)[None], kg_real[None, :, None, :])   # 4-D KG layout
    for k in ("mask", "instance", "edge", "score"):
        assert_close(outs[k], g[k], 2e-6, 1e-5, k)
    assert_close(outs["attn_rg2kg"][0], g["attn_rg2kg"][0], 1e-7, 1e-5, "attn_rg2kg")
    assert_close(outs["attn_kg2rg"][0], g["attn_kg2rg"][0], 1e-8, 1e-5, "attn_kg2rg")


def test_eval_selftest_shape_batch4():
    g = load_golden("eval_selftest_b4")
    rg = np.stack([OP.make_rg(500, 128, seed=40 + b, kind="randn") for b in range(4)])
    kg = np.stack([OP.make_rg(10, 128, seed=50 + b, kind="randn") for b in range(4)])
    outs, _ = _oracle().forward(rg, kg)
    for k in ("mask", "instance", "edge", "score"):
        assert outs[k].shape == g[k].shape
        assert_close(outs[k], g[k], 5e-6, 1e-5, k)
    assert_close(sub(np.stack(outs["attn_rg2kg"])), g["attn_rg2kg_sub"], 1e-7, 1e-4, "attn_rg2kg")
    assert_close(sub(np.stack(outs["attn_kg2rg"])), g["attn_kg2rg_sub"], 1e-8, 1e-4, "attn_kg2rg")


def test_eval_input_shapes():
    o = _oracle()
    g2, g4 = load_golden("eval_2d"), load_golden("eval_4d")
    o2, _ = o.forward(OP.make_rg(6, 128, seed=60), OP.make_kg(6, 128, seed=61))
    rg4 = np.stack([OP.make_rg(12, 128, seed=62 + b) for b in range(2)]).reshape(2, 3, 4, 128)
    kg4 = np.stack([OP.make_kg(5, 128, seed=64 + b) for b in range(2)]).reshape(2, 1, 5, 128)
    o4, _ = o.forward(rg4, kg4)
    for k in ("mask", "instance", "edge", "score"):
        assert_close(o2[k], g2[k], 2e-6, 1e-5, "2d " + k)
        assert_close(o4[k], g4[k], 2e-6, 1e-5, "4d " + k)
    with pytest.raises(ValueError, match="must be 2D/3D/4D tensor"):
        o.forward(np.zeros((1, 1, 1, 2, 128), np.float32), OP.make_kg(1, 128))
    assert "must be 2D/3D/4D tensor" in str(load_golden("eval_5d_error")["message"])


def _replay(name, steps=2):
    cfg, seed, nrs, nk, kg_fixed, full = train_case(name)
    o = FO.FusionOracle(cfg, OP.make_params(cfg, seed))
    opt = FO.AdamW(o.p, lr=5e-4, weight_decay=1e-4)
    real = {}
    for st in range(steps):
        g = load_golden(f"train_{name}_step{st}")
        rg, kg, y, e, s = train_batch(cfg, seed, nrs, nk, kg_fixed, st)
        r = FO.train_step(o, opt, rg, kg, y, e, s, training=True, seed=0)
        outs6 = np.concatenate([r["outs"]["mask"], r["outs"]["instance"], r["outs"]["edge"], r["outs"]["score"]], axis=1)
        assert_close(outs6, g["outs"], 5e-6, 2e-5, f"{name} step{st} outs")
        assert_close(r["loss_terms"], g["loss_terms"], 2e-6, 2e-5, f"{name} step{st} loss terms")
        assert_close(r["grad_norm"], g["grad_norm"], 0, 2e-5, f"{name} step{st} grad norm")
        for k, _ in OP.param_specs(cfg):
            raw = r["raw_grads"][k]
            gn = float(g[f"gnorm/{k}"])
            atol = 2e-6 * max(gn / np.sqrt(raw.size), 1e-6) + 1e-7
            assert_close(np.sqrt((raw.astype(np.float64) ** 2).sum()), g[f"gnorm/{k}"], 1e-7, 5e-5, f"{name} step{st} |g| {k}")
            assert_close(raw if full else sub(raw), g[f"g/{k}"], 50 * atol, 2e-4, f"{name} step{st} grad {k}")
            real[k] = (np.abs(g[f"g/{k}"]) >= 1e-6) & real.get(k, True)
            assert_params_close(o.p[k] if full else sub(o.p[k]), g[f"p/{k}"], 5e-4 * (st + 1), real[k], f"{name} step{st} param {k}")


def test_train_steps_default_config():
    _replay("default")


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_steps_small_configs(name):
    _replay(name)


def test_losses_and_metrics():
    g = load_golden("loss")
    l, d = FO.focal_loss(g["logits"], g["targets"])
    assert_close(l, g["focal"], 1e-7, 1e-5, "focal"); assert_close(d, g["focal_grad"], 1e-7, 1e-4, "focal grad")
    l, d = FO.cross_entropy(g["logits"], g["targets"])
    assert_close(l, g["ce"], 1e-7, 1e-5, "ce"); assert_close(d, g["ce_grad"], 1e-7, 1e-5, "ce grad")
    l, d = FO.bce_with_logits(g["edge"], g["edge_t"])
    assert_close(l, g["bce"], 1e-7, 1e-5, "bce"); assert_close(d, g["bce_grad"], 1e-7, 1e-5, "bce grad")
    l, d = FO.mse(g["score"], g["score_t"])
    assert_close(l, g["mse"], 1e-7, 1e-5, "mse"); assert_close(d, g["mse_grad"], 1e-7, 1e-5, "mse grad")
    f = FO.f1_scores(g["f1_pred"], g["f1_lab"])
    for k, v in f.items():
        assert_close(v, g[f"f1/{k}"], 1e-6, 1e-5, k)
    lrs = [FO.cosine_warm_restarts_lr(5e-4, ep) for ep in range(35)]
    assert_close(lrs, g["lr_schedule"], 1e-10, 1e-6, "lr schedule")


def test_param_table_matches_reference_counts():
    # 1 448 710 parameters cross-attention, 148 614 late fusion
    n = sum(int(np.prod(s)) for _, s in OP.param_specs())
    assert n == 1448710 and len(OP.param_specs()) == 44
    assert sum(int(np.prod(s)) for _, s in OP.param_specs(dict(fusion_type="late"))) == 148614
    with pytest.raises(ValueError, match="Unknown fusion_type"):
        OP.param_specs(dict(fusion_type="nope"))


def test_invariances(kg_real):
    """Known-answer properties of the model: no positional encoding and
    mean pooling => logits invariant to KG-row and RG-row order; batch == B x (B=1)."""
    o = _oracle()
    rg = OP.make_rg(481, 128, seed=5)
    base, _ = o.forward(rg[None], kg_real[None])
    rs = np.random.RandomState(0)
    p1, _ = o.forward(rg[None], kg_real[rs.permutation(13)][None])
    p2, _ = o.forward(rg[rs.permutation(481)][None], kg_real[None])
    for k in ("mask", "instance", "edge", "score"):
        assert_close(p1[k], base[k], 2e-6, 0, "kg perm " + k)
        assert_close(p2[k], base[k], 2e-6, 0, "rg perm " + k)
    assert_close(base["attn_rg2kg"][0].sum(axis=1), np.ones(481), 1e-5, 0, "rows sum to 1")
    assert_close(base["attn_kg2rg"][0].sum(axis=1), np.ones(13), 1e-5, 0, "rows sum to 1")


def test_dropout_hash_statistics():
    idx = np.arange(1 << 18)
    for p in (0.1, 0.3, 0.5):
        keep = FO.dropout_keep(0x1234567890AB, FO.SITE_FFN_RG, idx, p)
        assert abs(keep.mean() - (1 - p)) < 4e-3
    a = FO.dropout_keep(1, FO.SITE_FFN_RG, idx, 0.3); b = FO.dropout_keep(2, FO.SITE_FFN_RG, idx, 0.3)
    c = FO.dropout_keep(1, FO.SITE_FFN_KG, idx, 0.3)
    assert 0.55 < (a == b).mean() < 0.61 and 0.55 < (a == c).mean() < 0.61   # independent streams: 0.7^2+0.3^2
    # neighbouring elements uncorrelated, also at the strides the kernels index with (row pitch 512, head pitch 13, 104)
    for lag in (1, 2, 13, 104, 512, 4096):
        assert abs(np.corrcoef(a[:-lag], a[lag:])[0, 1]) < 0.01, lag
    # the streams of two sites are not shifted copies of each other at small shifts
    for lag in range(1, 64):
        assert abs(np.corrcoef(a[:-lag], c[lag:])[0, 1]) < 0.02 and abs(np.corrcoef(c[:-lag], a[lag:])[0, 1]) < 0.02
    # seeds that differ only in the high word give independent masks
    d = FO.dropout_keep(1 | (7 << 32), FO.SITE_FFN_RG, idx, 0.3)
    assert 0.55 < (a == d).mean() < 0.61


def test_train_mode_dropout_gradcheck():
    """With dropout>0 the oracle's backward must be the derivative of its own
    forward under the same hash masks (finite differences on a tiny config)."""
    cfg = OP.full_cfg(dict(rg_dim=8, kg_dim=8, hidden_dim=16, num_heads=2, dropout=0.3))
    prm = {k: v.astype(np.float64) for k, v in OP.make_params(cfg, 5).items()}
    rg = [OP.make_rg(5, 8, seed=1).astype(np.float64), OP.make_rg(3, 8, seed=2).astype(np.float64)]
    kg = np.stack([OP.make_kg(4, 8, seed=3), OP.make_kg(4, 8, seed=4)]).astype(np.float64)
    y, e, s = OP.make_labels(2, seed=9)

    import oracle.fusion_oracle as M
    old = M.f32
    M.f32 = np.float64          # run the same code in float64 for a clean finite difference
    try:
        def total(p):
            o = FO.FusionOracle(cfg, p); o.p = {k: np.asarray(v, np.float64) for k, v in p.items()}
            outs, caches = o.forward_list(rg, kg, training=True, seed=77)
            tot, ds = 0.0, []
            for b in range(2):
                l, _, d = FO.sample_loss({k: outs[k][b] for k in ("mask", "instance", "edge", "score")}, int(y[b]), float(e[b]), float(s[b]))
                tot += float(l); ds.append(d)
            return tot, o, caches, ds
        _, o, caches, ds = total(prm)
        g = {k: np.zeros_like(v) for k, v in o.p.items()}
        for ca, d in zip(caches, ds):
            o.backward_sample(ca, d, g)
        rs = np.random.RandomState(0)
        for k in prm:
            for _ in range(3):
                i = tuple(rs.randint(0, n) for n in prm[k].shape)
                h = 1e-6
                pp = {a: b.copy() for a, b in prm.items()}; pp[k][i] += h
                pm = {a: b.copy() for a, b in prm.items()}; pm[k][i] -= h
                fd = (total(pp)[0] - total(pm)[0]) / (2 * h)
                assert abs(fd - g[k][i]) <= 1e-5 + 1e-4 * abs(fd), (k, i, fd, g[k][i])
    finally:
        M.f32 = old


def test_torch_port_matches_numpy_oracle_and_golden_step():
    """oracle/torch_port.py (the torch-CPU restatement bench.py times as ``cpu_baseline``) against the golden optimizer
    step of the default configuration and against the numpy oracle: same losses, gradient norm and updated parameters."""
    import torch
    from helpers import train_batch, train_case
    from oracle.torch_port import TorchPort
    cfg, seed, nrs, nk, kg_fixed, _ = train_case("default")
    g = load_golden("train_default_step0")
    rg, kg, y, e, s = train_batch(cfg, seed, nrs, nk, kg_fixed, 0)
    tp = TorchPort(cfg, OP.make_params(cfg, seed))
    torch.set_num_threads(4)
    losses, norm = tp.train_step(rg, kg, y, e, s, training=True)          # dropout 0 in this case: deterministic
    assert np.abs(np.array(losses) - g["loss_terms"].sum(1)).max() < 2e-5
    assert abs(norm - float(g["grad_norm"])) < 2e-4 * float(g["grad_norm"])
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, seed))
    FO.train_step(orc, FO.AdamW(orc.p), rg, kg, y, e, s, training=True)
    for k, v in orc.p.items():
        err = np.abs(tp.P[k].detach().numpy() - v)
        assert err.max() <= 2.2 * 5e-4 and (err <= 3e-6 + 1e-5 * np.abs(v)).mean() > 0.99, k


def test_per_sample_batch_step_equals_train_step_and_flips_by_replacement():
    """tests/helpers.py::oracle_batch_step (the large-batch yardstick of tests/test_hip_large_batch.py: one sample at a time, a ReLU
    decision tried flipped by replacing that sample's gradient contribution) against FO.train_step on a small case, with and
    without a forced flip: samples are independent, so both routes must give the same sums."""
    from helpers import oracle_batch_step
    cfg = OP.full_cfg()
    nrs = [40, 7, 65]
    rg = [OP.make_rg(n, 128, seed=30 + i) for i, n in enumerate(nrs)]
    kg = np.stack([OP.make_kg(13, 128, seed=40 + i) for i in range(3)])
    y, e, s = OP.make_labels(3, seed=2)
    mk = lambda: FO.FusionOracle(cfg, OP.make_params(cfg, 3), bf16_operands=True)
    o = mk()
    ref = FO.train_step(o, FO.AdamW(o.p), rg, kg, y, e, s, training=True, seed=77)
    got = oracle_batch_step(mk, rg, kg, y, e, s, 77)
    for k in ref["raw_grads"]:
        assert np.array_equal(got["raw_grads"][k], ref["raw_grads"][k]), k
    assert np.array_equal(got["loss_terms"], ref["loss_terms"])
    for k in ("mask", "instance", "edge", "score"):
        assert np.array_equal(got["outs"][k], ref["outs"][k])
    # a flipped unit: the whole-batch step with relu_flip set == the per-sample replacement route, when the flipped pattern is
    # what the "kernel" computed.  A wide near_eps makes candidates; the target gradients are those of one chosen flip.
    probe = mk(); probe.near = []; probe.near_eps = 5e-3
    FO.train_step(probe, FO.AdamW(probe.p), rg, kg, y, e, s, training=True, seed=77)
    assert probe.near, "no tail unit within 5e-3 of the threshold in this case: pick another seed"
    site, b, u, _ = probe.near[0]
    of = mk(); of.relu_flip = frozenset([(site, b, u)])
    target = FO.train_step(of, FO.AdamW(of.p), rg, kg, y, e, s, training=True, seed=77)["raw_grads"]
    fit = oracle_batch_step(mk, rg, kg, y, e, s, 77, got_grads=target, near_eps=5e-3, max_near=len(probe.near), max_flips=len(probe.near))
    assert (site, b, u) in fit["flips"]
    err = max(float(np.abs(fit["raw_grads"][k] - target[k]).max()) / max(float(np.abs(target[k]).max()), 1e-12) for k in target)
    assert err < 1e-5, err
