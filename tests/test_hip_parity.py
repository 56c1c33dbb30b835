"""HIP path vs the CPU oracle and the reference's golden vectors.  Needs an MI355X.

Everything goes through the product's public surface (nn.Module / NativeTrainer), i.e. through
the C ABI of libcamo_fusion.so.  Tolerances:
  * precision 'f32' (v_mfma_f32_32x32x2_f32): logits within 2e-5 of the oracle -- well inside
    north_star's 1e-3 -- gradients within 2e-4 relative to each tensor's RMS;
  * precision 'bf16' (bf16 MFMA operands, fp32 accumulate): logits within 1e-3 (north_star's
    bound), prediction agreement IoU >= 0.999 on the synthetic set, gradients within a few
    percent (reported, loosely asserted).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import TRAIN_CASES, assert_close, assert_params_close, sub, train_batch, train_case
from oracle import fusion_oracle as FO
from oracle import params as OP

pytestmark = pytest.mark.gpu

OUT_KEYS = ("mask", "instance", "edge", "score")


def make_model(cfg, seed, precision="f32"):
    from camouflage_multimodal_amd import build_multimodal_model
    from camouflage_multimodal_amd import _lib
    assert _lib.lib() is not None
    m = build_multimodal_model(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in OP.make_params(cfg, seed).items()}, strict=True)
    return m.to("cuda").set_precision(precision)


def outs6(o):
    return np.concatenate([np.asarray(o[k]).reshape(len(o["mask"]), -1) for k in OUT_KEYS], axis=1)


def t2n(t):
    return t.detach().float().cpu().numpy()


@pytest.mark.parametrize("nr", [303, 481, 500, 530])
def test_eval_forward_golden_real_kg(nr, kg_real):
    g = load_golden(f"eval_nr{nr}")
    m = make_model(OP.full_cfg(), 0).eval()
    rg = torch.from_numpy(OP.make_rg(nr, 128, seed=nr))[None].cuda()
    kg = torch.from_numpy(kg_real)[None, :, None, :].cuda()          # the 4-D layout real data arrives in
    with torch.no_grad():
        mo, io, eo, so, attn = m(rg, kg, return_attention=True)
    for k, v in zip(OUT_KEYS, (mo, io, eo, so)):
        assert v.shape == g[k].shape
        assert_close(t2n(v), g[k], 2e-5, 1e-5, k)
    assert attn["rg2kg"].shape == (1, nr, 13) and attn["kg2rg"].shape == (1, 13, nr)
    assert_close(t2n(attn["rg2kg"]), g["attn_rg2kg"], 2e-6, 1e-4, "attn rg2kg")
    assert_close(t2n(attn["kg2rg"]), g["attn_kg2rg"], 2e-7, 1e-4, "attn kg2rg")


def test_eval_selftest_shape_and_input_layouts():
    m = make_model(OP.full_cfg(), 0).eval()
    g = load_golden("eval_selftest_b4")
    rg = np.stack([OP.make_rg(500, 128, seed=40 + b, kind="randn") for b in range(4)])
    kg = np.stack([OP.make_rg(10, 128, seed=50 + b, kind="randn") for b in range(4)])
    with torch.no_grad():
        o = m(torch.from_numpy(rg).cuda(), torch.from_numpy(kg).cuda(), return_attention=True)
    for k, v in zip(OUT_KEYS, o[:4]):
        assert_close(t2n(v), g[k], 5e-5, 1e-5, k)
    assert_close(sub(t2n(o[4]["rg2kg"])), g["attn_rg2kg_sub"], 2e-6, 2e-4, "attn rg2kg")
    assert_close(sub(t2n(o[4]["kg2rg"].contiguous())), g["attn_kg2rg_sub"], 2e-7, 2e-4, "attn kg2rg")
    # 2-D and 4-D inputs (fusion_model.py:86-105), 5-D raises the reference's ValueError
    g2, g4 = load_golden("eval_2d"), load_golden("eval_4d")
    rg4 = np.stack([OP.make_rg(12, 128, seed=62 + b) for b in range(2)]).reshape(2, 3, 4, 128)
    kg4 = np.stack([OP.make_kg(5, 128, seed=64 + b) for b in range(2)]).reshape(2, 1, 5, 128)
    with torch.no_grad():
        o2 = m(torch.from_numpy(OP.make_rg(6, 128, seed=60)).cuda(), torch.from_numpy(OP.make_kg(6, 128, seed=61)).cuda())
        o4 = m(torch.from_numpy(rg4).cuda(), torch.from_numpy(kg4).cuda())
    for k, a, b in zip(OUT_KEYS, o2, o4):
        assert_close(t2n(a), g2[k], 2e-5, 1e-5, "2d " + k)
        assert_close(t2n(b), g4[k], 2e-5, 1e-5, "4d " + k)
    with pytest.raises(ValueError, match="must be 2D/3D/4D tensor"):
        m(torch.zeros(1, 1, 1, 2, 128).cuda(), torch.zeros(1, 128).cuda())


def _native_replay(name, precision="f32", steps=2):
    """Replays the golden optimizer steps with the native trainer; checks outs, loss terms,
    raw gradients (vs the oracle run in lock-step) and post-step parameters (vs golden)."""
    from camouflage_multimodal_amd import NativeTrainer
    cfg, seed, nrs, nk, kg_fixed, full = train_case(name)
    m = make_model(cfg, seed, precision).train()
    tr = NativeTrainer(m, lr=5e-4, weight_decay=1e-4, keep_grads=True)
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, seed))
    oopt = FO.AdamW(orc.p, lr=5e-4, weight_decay=1e-4)
    real = {}
    for st in range(steps):
        g = load_golden(f"train_{name}_step{st}")
        rg, kg, y, e, s = train_batch(cfg, seed, nrs, nk, kg_fixed, st)
        ref = FO.train_step(orc, oopt, rg, kg, y, e, s, training=True, seed=0)
        terms, pred = tr.step(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda(),
                              torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s))
        assert_close(t2n(terms), g["loss_terms"], 2e-5, 1e-4, f"{name} step{st} loss terms")
        assert_close(t2n(tr.opt.grad_norm())[0], g["grad_norm"], 0, 2e-4, f"{name} step{st} grad norm")
        assert (t2n(pred) == outs6(ref["outs"])[:, :cfg["num_classes"]].argmax(1)).all()
        # clipped gradients left behind in .grad, like clip_grad_norm_ does
        coef = min(1.0, 1.0 / (float(g["grad_norm"]) + 1e-6))
        named = dict(m.named_parameters())
        tr.engine.ensure_flat_grads(attach=True)
        for k, _ in OP.param_specs(cfg):
            got = t2n(named[k].grad) / coef
            want = ref["raw_grads"][k]
            rms = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
            if st == 0:
                assert_close(got, want, 3e-4 * rms + 2e-7, 3e-4, f"{name} step{st} grad {k}")
            else:
                # after one AdamW step the two runs' parameters differ by up to ~2*lr on elements whose
                # step-0 gradient was rounding noise (see helpers.assert_params_close); a few ReLU units
                # near their threshold then flip for some rows, so later gradients agree in bulk only
                err = np.abs(got.astype(np.float64) - want)
                ok = err <= 3e-4 * rms + 2e-7 + 3e-4 * np.abs(want)
                assert ok.mean() >= 0.995 and err.max() <= 0.1 * rms + 1e-6, \
                    f"{name} step{st} grad {k}: {ok.mean():.4f} within tolerance, max err {err.max():.3e} (rms {rms:.3e})"
            real[k] = (np.abs(g[f"g/{k}"]) >= 1e-6) & real.get(k, True)
            p = t2n(named[k])
            assert_params_close(p if full else sub(p), g[f"p/{k}"], 5e-4 * (st + 1), real[k], f"{name} step{st} param {k}")


def test_native_train_steps_default_config_golden():
    _native_replay("default")


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_native_train_steps_small_configs_golden(name):
    _native_replay(name)


def test_dropin_autograd_loop_matches_golden():
    """The reference's own training loop shape (train_multimodal.py:238-279): per-sample B=1
    forward, torch loss functions, loss.backward(), clip_grad_norm_, torch AdamW -- with only
    the model swapped for the HIP one."""
    import torch.nn as nn
    import torch.nn.functional as F
    from camouflage_multimodal_amd import AggressiveFocalLoss
    cfg, seed, nrs, nk, kg_fixed, full = train_case("default")
    m = make_model(cfg, seed).train()
    opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=1e-4)
    focal, bce, mse = AggressiveFocalLoss(0.75, 3.0), nn.BCEWithLogitsLoss(), nn.MSELoss()
    real = {}
    for st in range(2):
        g = load_golden(f"train_default_step{st}")
        rg, kg, y, e, s = train_batch(cfg, seed, nrs, nk, kg_fixed, st)
        opt.zero_grad()
        for b in range(len(nrs)):
            yl = torch.tensor([int(y[b])]).cuda(); el = torch.tensor([float(e[b])]).cuda(); sl = torch.tensor([float(s[b])]).cuda()
            mo, io, eo, so = m(torch.from_numpy(rg[b])[None].cuda(), torch.from_numpy(kg[b])[None, :, None, :].cuda())
            loss = focal(mo, yl) * 3.0 + F.cross_entropy(io, yl) + bce(eo.squeeze(1), el) * 0.5 + mse(so.squeeze(1), sl) * 0.3
            loss.backward()
            assert_close(np.concatenate([t2n(mo)[0], t2n(io)[0], t2n(eo)[0], t2n(so)[0]]), g["outs"][b], 2e-5, 1e-5, "outs")
        norm = float(torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0))
        assert_close(norm, g["grad_norm"], 0, 2e-4, "grad norm")
        opt.step()
        for k, p in m.named_parameters():
            real[k] = (np.abs(g[f"g/{k}"]) >= 1e-6) & real.get(k, True)
            assert_params_close(sub(t2n(p)), g[f"p/{k}"], 5e-4 * (st + 1), real[k], f"dropin step{st} param {k}")


@pytest.mark.parametrize("name", ["default", "small_a", "late"])
def test_train_mode_dropout_matches_oracle_masks(name):
    """dropout>0, train mode: the kernels regenerate the oracle's counter-hash masks, so outputs
    and gradients are comparable element for element."""
    from camouflage_multimodal_amd import NativeTrainer
    cfg, seed, nrs, nk, kg_fixed, _ = train_case(name)
    cfg = dict(cfg, dropout=0.3)
    m = make_model(cfg, seed).train()
    tr = NativeTrainer(m, keep_grads=True)
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, seed))
    rg, kg, y, e, s = train_batch(cfg, seed, nrs, nk, kg_fixed, 0)
    dseed = 0x0123456789ABCDEF
    ref = FO.train_step(orc, FO.AdamW(orc.p), rg, kg, y, e, s, training=True, seed=dseed)
    terms, _ = tr.step(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda(),
                       torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), seed=dseed)
    assert_close(t2n(terms), ref["loss_terms"], 3e-5, 2e-4, "loss terms under dropout")
    coef = min(1.0, 1.0 / (float(ref["grad_norm"]) + 1e-6))
    tr.engine.ensure_flat_grads(attach=True)
    for k, p in m.named_parameters():
        want = ref["raw_grads"][k]
        rms = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
        assert_close(t2n(p.grad) / coef, want, 5e-4 * rms + 2e-7, 5e-4, f"dropout grad {k}")
    # a different seed must give a different loss (the masks really are applied)
    m2 = make_model(cfg, seed).train()
    t2, _ = NativeTrainer(m2).step(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda(),
                                   torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), seed=dseed + 1)
    assert np.abs(t2n(t2) - t2n(terms)).max() > 1e-4


def test_edge_shapes_vs_oracle():
    """Nr=1, Nk=1, Nk=16/17 (register-array dispatch boundary), odd dims, many heads."""
    small = [((1, 2, 70), 1), ((5, 64, 65), 16), ((3, 130), 17)]
    cases = [(dict(rg_dim=20, kg_dim=12, hidden_dim=32, num_heads=4), small + [((257,), 64)]),
             (dict(rg_dim=128, kg_dim=128, hidden_dim=256, num_heads=8), small + [((40, 9), 64)]),
             (dict(rg_dim=64, kg_dim=64, hidden_dim=64, num_heads=64), small),
             (dict(rg_dim=8, kg_dim=8, hidden_dim=512, num_heads=8), small)]
    for ci, (c, shapes) in enumerate(cases):
        cfg = OP.full_cfg(dict(c, dropout=0.0))
        m = make_model(cfg, 11 + ci).eval()
        orc = FO.FusionOracle(cfg, OP.make_params(cfg, 11 + ci))
        for nrs, nk in shapes:
            rg = [OP.make_rg(n, cfg["rg_dim"], seed=i + 7, kind="randn") for i, n in enumerate(nrs)]
            kg = np.stack([OP.make_kg(nk, cfg["kg_dim"], seed=90 + i) for i in range(len(nrs))])
            ref, _ = orc.forward_list(rg, kg)
            with torch.no_grad():
                o = m.forward_packed(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda(),
                                     return_attention=True)
            got = np.concatenate([t2n(v) for v in o[:4]], axis=1)
            assert_close(got, outs6(ref), 5e-5, 1e-4, f"cfg{ci} nrs={nrs} nk={nk}")
            for b in range(len(nrs)):
                assert_close(t2n(o[4]["rg2kg"][b]), ref["attn_rg2kg"][b], 2e-6, 2e-4, "attn rg2kg")
                assert_close(t2n(o[4]["kg2rg"][b]), ref["attn_kg2rg"][b], 2e-6, 2e-4, "attn kg2rg")


def test_full_size_properties_b16(kg_real):
    """BASELINE config size (B=16, real Nr spread): packed batch == 16 x (B=1); logits invariant
    to KG-row and RG-row permutations; attention rows sum to 1."""
    h = load_golden("nr_histogram")
    rs = np.random.RandomState(0)
    nrs = [int(x) for x in rs.choice(h["values"], size=16, p=h["counts"] / h["counts"].sum())]
    m = make_model(OP.full_cfg(), 0).eval()
    rg = [OP.make_rg(n, 128, seed=200 + i) for i, n in enumerate(nrs)]
    kg = np.stack([kg_real] * 16)
    with torch.no_grad():
        o = m.forward_packed(torch.from_numpy(np.concatenate(rg)).cuda(), nrs, torch.from_numpy(kg).cuda(), return_attention=True)
        packed = np.concatenate([t2n(v) for v in o[:4]], axis=1)
        singles = []
        for b in range(16):
            ob = m(torch.from_numpy(rg[b])[None].cuda(), torch.from_numpy(kg_real)[None].cuda())
            singles.append(np.concatenate([t2n(v) for v in ob], axis=1)[0])
        assert_close(packed, np.stack(singles), 2e-6, 0, "packed vs singles")
        perm_k = rs.permutation(13)
        rgp = [r[rs.permutation(len(r))] for r in rg]
        o2 = m.forward_packed(torch.from_numpy(np.concatenate(rgp)).cuda(), nrs, torch.from_numpy(kg[:, perm_k]).cuda())
        assert_close(np.concatenate([t2n(v) for v in o2], axis=1), packed, 5e-6, 0, "permutation invariance")
    for b in range(16):
        assert_close(t2n(o[4]["rg2kg"][b]).sum(1), np.ones(nrs[b]), 1e-5, 0, "rg2kg rows sum to 1")
        assert_close(t2n(o[4]["kg2rg"][b]).sum(1), np.ones(13), 1e-5, 0, "kg2rg rows sum to 1")
    orc = FO.FusionOracle(OP.full_cfg(), OP.make_params(OP.full_cfg(), 0))
    ref, _ = orc.forward_list(rg[:4], kg[:4])
    assert_close(packed[:4], outs6(ref), 2e-5, 1e-5, "first four samples vs oracle")


def test_bf16_mode_within_north_star_tolerance(kg_real):
    """bf16 MFMA operands: logits within 1e-3 of the fp32 oracle and prediction-agreement IoU
    >= 0.999 (utils/metrics.py:9-18's IoU formula on arg-max vectors) over a 64-sample synthetic set."""
    cfg = OP.full_cfg()
    m = make_model(cfg, 0, "bf16").eval()
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, 0))
    h = load_golden("nr_histogram")
    rs = np.random.RandomState(1)
    nrs = [int(x) for x in rs.choice(h["values"], size=64, p=h["counts"] / h["counts"].sum())]
    rg = [OP.make_rg(n, 128, seed=300 + i) for i, n in enumerate(nrs)]
    kg = np.stack([kg_real] * 64)
    ref, _ = orc.forward_list(rg, kg)
    with torch.no_grad():
        o = m.forward_packed(torch.from_numpy(np.concatenate(rg)).cuda(), nrs, torch.from_numpy(kg).cuda())
    got = np.concatenate([t2n(v) for v in o], axis=1)
    err = np.abs(got - outs6(ref)).max()
    print("bf16 max |logit err| =", err)
    assert err < 1e-3
    pa, pb = got[:, :2].argmax(1), outs6(ref)[:, :2].argmax(1)
    inter = float(((pa == 1) & (pb == 1)).sum()); union = float(((pa == 1) | (pb == 1)).sum())
    assert (inter + 1e-8) / (union + 1e-8) >= 0.999


def test_bf16_training_step_close_to_oracle():
    """The golden 'default' minibatch (dropout 0, Nr = 303 / 481 / 500 / 530) through NativeTrainer in bf16 mode against the train
    step of the oracle in its bf16-operand mode, at the absolute bounds of every other bf16 training test: global relative gradient
    error < 2e-3, every tensor that carries weight < 1e-2 (round 1 asserted 5 % / 15 % here against the f32 oracle)."""
    from camouflage_multimodal_amd import NativeTrainer
    from helpers import oracle_step_at_relu_thresholds
    cfg, seed, nrs, nk, kg_fixed, _ = train_case("default")
    m = make_model(cfg, seed, "bf16").train()
    tr = NativeTrainer(m, keep_grads=True)
    rg, kg, y, e, s = train_batch(cfg, seed, nrs, nk, kg_fixed, 0)
    terms, _ = tr.step(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda(),
                       torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s))
    gnorm = float(t2n(tr.opt.grad_norm())[0])
    coef = min(1.0, 1.0 / (gnorm + 1e-6))
    tr.engine.ensure_flat_grads(attach=True)
    grads = {k: t2n(p.grad).astype(np.float32) / np.float32(coef) for k, p in m.named_parameters()}
    ref, near, flipped = oracle_step_at_relu_thresholds(
        lambda: FO.FusionOracle(cfg, OP.make_params(cfg, seed), bf16_operands=True),
        lambda o: FO.train_step(o, FO.AdamW(o.p), rg, kg, y, e, s, training=True), grads)
    assert_close(t2n(terms), ref["loss_terms"], 5e-4, 5e-4, "bf16 loss terms")
    assert_close(gnorm, ref["grad_norm"], 0, 5e-3, "bf16 grad norm")
    num = den = 0.0
    rels = []
    for k in grads:
        want = ref["raw_grads"][k].astype(np.float64); got = grads[k].astype(np.float64)
        num += ((got - want) ** 2).sum(); den += (want ** 2).sum()
        rels.append((np.sqrt(((got - want) ** 2).sum()) / max(np.sqrt((want ** 2).sum()), 1e-30), np.sqrt((want ** 2).sum()), k))
    total = np.sqrt(num / den)
    rels.sort(reverse=True)
    print("bf16 global relative gradient error vs the bf16-operand oracle =", total, "; worst tensors:", [(f"{r:.4f}", f"{n:.2e}", k) for r, n, k in rels[:4]],
          "; near the ReLU threshold:", near, "flipped:", flipped)
    assert total < 2e-3
    gn = np.sqrt(den)
    assert all(r < 1e-2 for r, n, _ in rels if n > 1e-3 * gn), rels[:4]


def test_product_path_loaded_native_library():
    import ctypes
    from camouflage_multimodal_amd import _lib
    L = _lib.lib()
    assert isinstance(L, ctypes.CDLL) and L.camo_abi_version() == _lib.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libcamo_fusion.so" in f.read()


def _set_sched16(value):
    """-1: the bf16-resident schedule of round 1 where the configuration allows it; 0: general schedule forced.  Both legs
    switch the fused row-tile schedule (the default at the reference configuration) off: these tests are about the two
    grouped-GEMM schedules it falls back to."""
    from camouflage_multimodal_amd import _lib
    _lib.check(_lib.lib().camo_debug_set_option(b"fused", 0), "camo_debug_set_option")
    _lib.check(_lib.lib().camo_debug_set_option(b"sched16", value), "camo_debug_set_option")


@pytest.fixture
def sched_switch():
    yield _set_sched16
    from camouflage_multimodal_amd import _lib
    _lib.check(_lib.lib().camo_debug_set_option(b"fused", -1), "camo_debug_set_option")
    _lib.check(_lib.lib().camo_debug_set_option(b"sched16", -1), "camo_debug_set_option")


def _ws_get(eng, batch, ws, name, shape):
    import ctypes as C
    from camouflage_multimodal_amd import _lib
    off = _lib.lib().camo_debug_ws_offset(C.byref(eng.dims), batch.B, batch.T, batch.Nk, name.encode())
    assert off >= 0, name
    n = int(np.prod(shape))
    return ws[off:off + 4 * n].view(torch.float32).view(*shape).float().cpu().numpy().copy()


@pytest.mark.parametrize("training", [False, True])
def test_bf16_resident_schedule_matches_fp32_operand_schedule(training, kg_real, sched_switch):
    """The default bf16 schedule (bf16-resident operands, gemm16.hip, concatenated in-projection backward) against the
    general schedule in bf16 mode (fp32 operands rounded while staging, gemm.hip).  Both round the same values to
    bf16 at the same points and accumulate in fp32, so forward activations, outputs and every parameter gradient
    agree up to summation order (and rare 1-ulp bf16 flips downstream of it)."""
    cfg = OP.full_cfg()
    m = make_model(cfg, 0, "bf16")
    m.train(training)
    eng = m._engine
    nrs = [303, 64, 1, 530, 65, 127, 128, 500]
    rg = [OP.make_rg(n, 128, seed=90 + i) for i, n in enumerate(nrs)]
    kg = np.stack([kg_real] * len(nrs))
    batch = eng.make_batch(torch.from_numpy(np.concatenate(rg)).cuda(), nrs, torch.from_numpy(kg).cuda())
    T, B = sum(nrs), len(nrs)
    names = [("R", (T, 256)), ("Q", (T, 256)), ("KV2", (T, 512)), ("KV", (B * 13, 512)), ("P", (T, 8, 13)), ("P2", (T, 8, 13)),
             ("U", (T, 256)), ("U2", (B * 13, 256)),
             ("comb", (B, 512)), ("fused", (B, 256))]     # (the bf16 schedule keeps Y, H1, H2 only as bf16)
    d_outs = torch.from_numpy(np.random.RandomState(5).standard_normal((B, 6)).astype(np.float32)).cuda()
    res = {}
    for mode in ("sched16", "general"):
        sched_switch(-1 if mode == "sched16" else 0)
        ws = eng.workspace(batch, private=True)
        ws.zero_()
        outs, attn = eng.forward_raw(batch, ws, training, 0xABCDEF0123, want_attention=True)
        r = dict(outs=t2n(outs), a1=t2n(attn[0]), a2=t2n(attn[1]), **{n: _ws_get(eng, batch, ws, n, sh) for n, sh in names})
        g = eng.ensure_flat_grads(attach=True)
        g.zero_()
        eng.backward_raw(batch, ws, outs, d_outs, training, 0xABCDEF0123, eng._gtab)
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            r["grad:" + k] = t2n(p.grad).copy()
        res[mode] = r
    worst = []
    for k in res["sched16"]:
        a, b = res["sched16"][k].astype(np.float64), res["general"][k].astype(np.float64)
        scale = max(float(np.abs(b).max()), 1e-6)
        err = float(np.abs(a - b).max())
        worst.append((err / scale, k))
        tol = 2e-6 if k in ("R", "Q", "KV2", "KV") else 4e-3
        assert err <= tol * scale, f"{k}: max diff {err:.3e} (scale {scale:.3e})"
        assert float(np.abs(a - b).mean()) <= 5e-5 * scale, f"{k}: mean diff {np.abs(a - b).mean():.3e} (scale {scale:.3e})"
    worst.sort(reverse=True)
    print("sched16 vs general, worst relative max-diffs:", [(f"{e:.2e}", k) for e, k in worst[:5]])


@pytest.mark.parametrize("name,precision", [("default", "bf16"), ("default", "f32"), ("small_cls3", "f32"), ("late", "f32")])
def test_fused_training_call_matches_forward_loss_backward(name, precision):
    """camo_forward_loss_backward (head output layer + loss + its backward in one kernel) against the three separate
    calls: same loss terms, predictions and gradients up to fp32 summation order."""
    from camouflage_multimodal_amd import NativeTrainer
    cfg, seed, nrs, nk, kg_fixed, _ = train_case(name)
    rg, kg, y, e, s = train_batch(cfg, seed, nrs, nk, kg_fixed, 0)
    res = []
    for fused in (True, False):
        m = make_model(cfg, seed, precision).train()
        tr = NativeTrainer(m, keep_grads=True, fused_call=fused)
        terms, pred = tr.step(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda(),
                              torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), seed=1234)
        tr.engine.ensure_flat_grads(attach=True)
        res.append((t2n(terms), t2n(pred), {k: t2n(p.grad).copy() for k, p in m.named_parameters()}, t2n(tr.opt.grad_norm())))
    (ta, pa, ga, na), (tb, pb, gb, nb) = res
    assert_close(ta, tb, 1e-6, 1e-5, "loss terms")
    assert np.array_equal(pa, pb)
    assert_close(na, nb, 0, 1e-4, "grad norm")
    for k in ga:
        scale = max(float(np.abs(gb[k]).max()), 1e-8)
        tol = (4e-3 if precision == "bf16" else 2e-5) * scale      # bf16: a 1-ulp fp32 change upstream can flip a bf16 rounding
        assert float(np.abs(ga[k] - gb[k]).max()) <= tol, k


@pytest.mark.parametrize("nrs", [[768, 5], [769, 5], [128] * 4, [1], [300] * 33])
def test_bf16_schedules_agree_on_boundary_batches(nrs, kg_real, sched_switch):
    """Boundary batches of the bf16-resident schedule: the largest sample it takes (768 nodes), one node more (the
    call must fall back to the general schedule by itself), row counts that are exact multiples of the 128-row
    padding, a single one-node sample, an odd batch above 32.  One training call on the default configuration in
    bf16 mode against the same call forced onto the general schedule."""
    from camouflage_multimodal_amd import NativeTrainer
    cfg = OP.full_cfg()
    B = len(nrs)
    rg = np.concatenate([OP.make_rg(n, 128, seed=500 + i) for i, n in enumerate(nrs)])
    kg = np.stack([kg_real] * B)
    y, e, s = OP.make_labels(B, seed=9)
    res = []
    for mode in (-1, 0):
        sched_switch(mode)
        m = make_model(cfg, 3, "bf16").train()
        tr = NativeTrainer(m, keep_grads=True)
        terms, pred = tr.step(torch.from_numpy(rg).cuda(), list(nrs), torch.from_numpy(kg).cuda(), torch.from_numpy(y),
                              torch.from_numpy(e), torch.from_numpy(s), seed=77)
        tr.engine.ensure_flat_grads(attach=True)
        res.append((t2n(terms), {k: t2n(p.grad).copy() for k, p in m.named_parameters()}))
    (ta, ga), (tb, gb) = res
    assert np.isfinite(ta).all() and all(np.isfinite(v).all() for v in ga.values())
    assert_close(ta, tb, 2e-4, 2e-3, "loss terms")
    for k in ga:
        scale = max(float(np.abs(gb[k]).max()), 1e-8)
        assert float(np.abs(ga[k] - gb[k]).max()) <= 6e-3 * scale, (k, float(np.abs(ga[k] - gb[k]).max()), scale)


def test_bf16_dropin_autograd_loop_equals_native_packed_step():
    """Drop-in mode in bf16 precision (per-sample B = 1 forward through autograd, torch losses, loss.backward()): the
    bf16-resident schedule at T = Nr of a single sample, its fused LayerNorm launches and virtual operand included.
    The summed per-sample gradients must equal the gradients of one native packed step on the same samples (dropout
    off: the mask index of an element depends on its position in the packed batch)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from camouflage_multimodal_amd import AggressiveFocalLoss, NativeTrainer
    cfg = OP.full_cfg(dict(dropout=0.0))
    nrs = [303, 64, 500, 1]
    rg = [OP.make_rg(n, 128, seed=800 + i) for i, n in enumerate(nrs)]
    kg = np.stack([OP.make_kg(13, 128, seed=60 + i) for i in range(len(nrs))])
    y, e, s = OP.make_labels(len(nrs), seed=12)
    m = make_model(cfg, 5, "bf16").train()
    focal, bce, mse = AggressiveFocalLoss(0.75, 3.0), nn.BCEWithLogitsLoss(), nn.MSELoss()
    m.zero_grad()
    for b in range(len(nrs)):
        yl = torch.tensor([int(y[b])]).cuda(); el = torch.tensor([float(e[b])]).cuda(); sl = torch.tensor([float(s[b])]).cuda()
        mo, io, eo, so = m(torch.from_numpy(rg[b])[None].cuda(), torch.from_numpy(kg[b])[None, :, None, :].cuda())
        loss = focal(mo, yl) * 3.0 + F.cross_entropy(io, yl) + bce(eo.squeeze(1), el) * 0.5 + mse(so.squeeze(1), sl) * 0.3
        loss.backward()
    drop_in = {k: t2n(p.grad).copy() for k, p in m.named_parameters()}
    m2 = make_model(cfg, 5, "bf16").train()
    tr = NativeTrainer(m2, keep_grads=True, max_norm=1e9)          # (no clipping: compare raw gradients)
    tr.step(torch.from_numpy(np.concatenate(rg)).cuda(), nrs, torch.from_numpy(kg).cuda(), torch.from_numpy(y), torch.from_numpy(e),
            torch.from_numpy(s))
    tr.engine.ensure_flat_grads(attach=True)
    for k, p in m2.named_parameters():
        want = t2n(p.grad)
        scale = max(float(np.abs(want).max()), 1e-8)
        assert float(np.abs(drop_in[k] - want).max()) <= 6e-3 * scale, (k, float(np.abs(drop_in[k] - want).max()), scale)
