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


def oracle_step_at_relu_thresholds(make_oracle, step, got_grads, max_units=24):
    """The oracle's training step whose admissible ReLU sign pattern fits ``got_grads`` best.  A per-sample tail unit (fusion
    layer 0, a head's hidden layer) whose pre-activation is within 1e-4 of zero may come out on the other side of the threshold
    in an implementation that sums in another order (the kernels' pre-activations differ from the oracle's by ~2e-5), and one
    such unit moves a head's gradient by percent -- an admissible difference, not an error.  ``make_oracle()`` -> a fresh oracle,
    ``step(oracle)`` -> its train_step result.  The units near the threshold are recorded in a first pass (about one per seven
    samples); each is then tried flipped, greedily, and kept flipped when that fits better (their effects are separate paths)."""
    orc = make_oracle()
    orc.near = []
    orc.near_eps = 1e-4
    ref = step(orc)
    units = [(s, b, u) for s, b, u, _ in orc.near]
    assert len(units) <= max_units, f"{len(units)} tail units within {orc.near_eps} of the ReLU threshold: {orc.near}"

    def err(r):
        num = sum(float(((got_grads[k].astype(np.float64) - r["raw_grads"][k]) ** 2).sum()) for k in got_grads)
        den = sum(float((r["raw_grads"][k].astype(np.float64) ** 2).sum()) for k in got_grads)
        return np.sqrt(num / den)
    best, best_err, flips = ref, err(ref), []
    for u in units:
        o = make_oracle()
        o.relu_flip = frozenset(flips + [u])
        r = step(o)
        e = err(r)
        if e < best_err:
            best, best_err, flips = r, e, flips + [u]
    return best, list(orc.near), flips
