"""Case definitions shared by the oracle-vs-golden and HIP-vs-oracle tests.
The seeds here mirror tests/golden/make_golden.py exactly."""
import json

import numpy as np

from conftest import load_golden
from oracle import params as OP

TRAIN_CASES = ("small_a", "small_ident", "small_cls3", "late")


def train_case(name):
    """-> (cfg, seed, nrs, nk, kg_fixed or None, full_grads)"""
    if name == "default":
        return OP.full_cfg(dict(dropout=0.0)), 0, (303, 481, 500, 530), 13, load_golden("kg_embeddings")["kg"], False
    meta = load_golden(f"train_{name}_meta")
    cfg = json.loads(str(meta["cfg"]))
    return cfg, 3, tuple(int(x) for x in meta["nrs"]), int(meta["nk"]), None, True


def train_batch(cfg, seed, nrs, nk, kg_fixed, step):
    """The minibatch make_golden.ref_train_steps feeds at optimizer step ``step``."""
    B = len(nrs)
    y, e, s = OP.make_labels(B, seed=100 * seed + step)
    rg = [OP.make_rg(nr, cfg["rg_dim"], seed=1000 * step + b) for b, nr in enumerate(nrs)]
    kg = np.stack([kg_fixed if kg_fixed is not None else OP.make_kg(nk, cfg["kg_dim"], seed=1000 * step + 500 + b)
                   for b in range(B)])
    return rg, kg, y, e, s


def sub(a, stride=37):
    return np.ascontiguousarray(np.asarray(a).reshape(-1)[::stride])


def assert_close(a, b, atol, rtol, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: max violation at {i}: got {a[i]!r} want {b[i]!r} (|err|={err[i]:.3e}, tol={tol[i]:.3e})")


def assert_params_close(a, b, lr, real, what=""):
    """Post-AdamW parameters.  Early Adam steps move every element by ~lr*sign(g)
    whatever |g| is, so an element whose gradient is rounding noise around zero
    (e.g. an attention block's K-bias gradient, which is exactly zero in exact
    arithmetic) may legitimately land anywhere within +-lr of where it started.
    ``real`` marks the elements whose reference gradient was >= 1e-6 in magnitude
    at every step so far: those must agree tightly; the others are bounded by
    2.2*lr per step taken."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape == real.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = np.where(real, 3e-6 + 1e-5 * np.abs(b), 2.2 * lr)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: at {i}: got {a[i]!r} want {b[i]!r} (|err|={err[i]:.3e}, tol={tol[i]:.3e}, real={real[i]})")


def _grad_err(got_grads, ref_grads):
    num = sum(float(((got_grads[k].astype(np.float64) - ref_grads[k]) ** 2).sum()) for k in got_grads)
    den = sum(float((ref_grads[k].astype(np.float64) ** 2).sum()) for k in got_grads)
    return np.sqrt(num / den)


def oracle_step_at_relu_thresholds(make_oracle, step, got_grads, max_near=6, max_flips=3, near_eps=5e-5):
    """The oracle's training step whose admissible ReLU sign pattern fits ``got_grads`` best.  A per-sample tail unit (fusion
    layer 0, a head's hidden layer) whose pre-activation is within ``near_eps`` of zero may come out on the other side of the
    threshold in an implementation that sums in another order (the kernels' pre-activations differ from the oracle's by ~2e-5,
    hence 5e-5), and one such unit moves a head's gradient by percent -- an admissible difference, not an error.
    ``make_oracle()`` -> a fresh oracle, ``step(oracle)`` -> its train_step result.  The units near the threshold are recorded
    in a first pass; each is then tried flipped, greedily, and kept flipped when that fits better (their effects are separate
    paths).  The door is narrow by construction: at most ``max_near`` candidate units and ``max_flips`` flips taken per case --
    more than that is a failure of the case, not something to fit."""
    orc = make_oracle()
    orc.near = []
    orc.near_eps = near_eps
    ref = step(orc)
    units = [(s, b, u) for s, b, u, _ in orc.near]
    assert len(units) <= max_near, f"{len(units)} tail units within {orc.near_eps} of the ReLU threshold (allowed {max_near}): {orc.near}"
    best, best_err, flips = ref, _grad_err(got_grads, ref["raw_grads"]), []
    for u in units:
        o = make_oracle()
        o.relu_flip = frozenset(flips + [u])
        r = step(o)
        e = _grad_err(got_grads, r["raw_grads"])
        if e < best_err:
            best, best_err, flips = r, e, flips + [u]
    assert len(flips) <= max_flips, f"{len(flips)} ReLU decisions taken flipped (allowed {max_flips}): {flips} of {orc.near}"
    return best, list(orc.near), flips


def oracle_batch_step(make_oracle, rg_list, kg, y, e, s, seed, got_grads=None, training=True, near_eps=5e-5, max_near=None, max_flips=None):
    """The reference-semantics gradient of a LARGE minibatch (train_multimodal.py:238-279: per-sample forward / backward, gradients
    summed) from the oracle, one sample at a time and without keeping the samples' caches: -> dict(outs [B, 6 ...] as the
    oracle's forward_list gives them, loss_terms [B, 4], raw_grads, near, flips).  Samples are independent, so the ReLU-threshold
    treatment of ``oracle_step_at_relu_thresholds`` costs one sample's step per candidate unit here instead of a whole batch's:
    a unit of sample b is tried flipped by replacing that sample's gradient contribution.  Bounds on the candidates and on the
    flips taken scale with the batch (about one unit per 15 samples sits within 5e-5 of the threshold)."""
    from oracle import fusion_oracle as FO
    B = len(rg_list)
    max_near = max(6, B // 6) if max_near is None else max_near
    max_flips = max(3, B // 12) if max_flips is None else max_flips
    orc = make_oracle()
    orc.near = []
    orc.near_eps = near_eps
    g = orc.zero_grads()
    outs, terms, bases = [], [], []
    base = 0

    def sample_grad(o, b, into):
        out, ca = o.forward_sample(np.asarray(rg_list[b], np.float32), np.asarray(kg[b], np.float32), training, seed, bases[b], b)
        ob = {k: out[k] for k in ("mask", "instance", "edge", "score")}
        _, t, d = FO.sample_loss(ob, int(y[b]), float(e[b]), float(s[b]))
        o.backward_sample(ca, d, into)
        return ob, t
    for b in range(B):
        bases.append(base)
        ob, t = sample_grad(orc, b, g)
        outs.append(ob); terms.append(t)
        base += len(rg_list[b])
    res = dict(outs={k: np.stack([o[k] for o in outs]) for k in ("mask", "instance", "edge", "score")}, loss_terms=np.stack(terms),
               raw_grads=g, near=list(orc.near), flips=[])
    if got_grads is None or not orc.near:
        return res
    assert len(orc.near) <= max_near, f"{len(orc.near)} tail units within {near_eps} of the ReLU threshold (allowed {max_near}): {orc.near}"
    best_err = _grad_err(got_grads, g)
    taken = {}                                   # sample -> flips kept so far
    for (site, b, u, _) in orc.near:
        plain, flipped = make_oracle(), make_oracle()
        plain.relu_flip = frozenset(taken.get(b, []))
        flipped.relu_flip = frozenset(taken.get(b, []) + [(site, b, u)])
        g0, g1 = plain.zero_grads(), flipped.zero_grads()
        sample_grad(plain, b, g0); sample_grad(flipped, b, g1)
        trial = {k: g[k] + (g1[k] - g0[k]) for k in g}
        err = _grad_err(got_grads, trial)
        if err < best_err:
            g, best_err = trial, err
            taken.setdefault(b, []).append((site, b, u))
            res["flips"].append((site, b, u))
    assert len(res["flips"]) <= max_flips, f"{len(res['flips'])} ReLU decisions taken flipped (allowed {max_flips}): {res['flips']}"
    res["raw_grads"] = g
    return res


def first_blocks_64row_forward(nrs):
    """FusionOracle.kg_first_block for a batch whose forward runs on the 64-row half-blocks of csrc/fused_wide2.hip: blocks are cut
    from the batch's global table of 32-row tiles, two per block, so a sample that starts on an odd tile has a 32-key first flash
    block in its KG->RG attention (the bf16-operand oracle rounds the exponentials per block: the partition is part of the model)."""
    tiles = np.cumsum([0] + [(int(n) + 31) // 32 for n in nrs])
    return {b: 32 for b in range(len(nrs)) if tiles[b] % 2 == 1}


def bf16_oracle(cfg, params, nrs=None, wide2=False):
    """The oracle in its bf16-operand mode; ``wide2``: with the flash-block partition of the 64-row forward for this batch."""
    from oracle import fusion_oracle as FO
    o = FO.FusionOracle(cfg, params, bf16_operands=True)
    if wide2:
        o.kg_first_block = first_blocks_64row_forward(nrs)
    return o
