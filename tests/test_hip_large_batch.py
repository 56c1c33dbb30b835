"""The benchmarked LARGE-batch configurations against the oracle.  Needs an MI355X.

bench.py's sweep times eval forwards at B = 256 / 1024 / 4096 and training steps at B = 64 ... 4096; these sizes take kernel
configurations by SIZE that the small cases of the other test files reach only through developer options: several rounds of
wide row blocks per CU, the one-launch two-plane tail over many blocks, the 128 x 256 weight-gradient tiles (K >= 40 000 packed
rows), the two-plane tail forward + backward of training calls past 64 samples, several samples per block in the loss launch.
Here they run as the product runs them -- no option is touched -- and are held to the oracle:

  * eval forward, B = 256 of the real Nr histogram: logits of 16 samples spread over the batch (first, last, neighbours of the
    128-row block boundaries) against the reference-exact f32 oracle at north_star's 1e-3, and packed == one-by-one calls;
  * training step, B = 70 (wide front half by size) and B = 100 (T > 40 000: every by-size path of a large training call) against
    the train step of the oracle in its bf16-operand mode at the absolute bounds of the small cases: global relative gradient
    error < 2e-3, every tensor that carries weight < 1e-2;
  * eval forward, B = 1024: finite, packed == one-by-one for 8 sampled samples, and those 8 against the f32 oracle.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import assert_close, bf16_oracle, oracle_batch_step
from oracle import fusion_oracle as FO
from oracle import params as OP
from test_hip_parity import make_model, outs6, t2n

pytestmark = pytest.mark.gpu


def _histogram_batch(B, seed, kg_real):
    h = load_golden("nr_histogram")
    rs = np.random.RandomState(seed)
    nrs = [int(x) for x in rs.choice(h["values"], size=B, p=h["counts"] / h["counts"].sum())]
    # (inputs differ per sample through the seed; every 7th sample gets its own scaling of the real KG rows)
    rg = [OP.make_rg(n, 128, seed=5000 + 13 * seed + i) for i, n in enumerate(nrs)]
    kg = np.stack([kg_real * np.float32(1.0 + 0.02 * (i % 7)) for i in range(B)]).astype(np.float32)
    return nrs, rg, kg


def _packed_forward(m, rg, nrs, kg):
    with torch.no_grad():
        o = m.forward_packed(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda())
    return np.concatenate([t2n(v) for v in o], axis=1)


def _block_boundary_samples(nrs, rows_per_block, want):
    """The first two and last two samples, then samples that START inside a row block of the wide kernels (they share that block
    with their predecessor: two samples' keys, pooled sums and KG partials in one block) and samples that start exactly on a block
    boundary, alternating.  Blocks are cut from the per-sample 32-row tile table."""
    tiles = np.cumsum([0] + [(n + 31) // 32 for n in nrs])
    per = rows_per_block // 32
    inside = [b for b in range(2, len(nrs) - 2) if tiles[b] % per != 0]
    aligned = [b for b in range(2, len(nrs) - 2) if tiles[b] % per == 0]
    rs = np.random.RandomState(len(nrs))
    rs.shuffle(inside); rs.shuffle(aligned)
    out = [0, 1, len(nrs) - 2, len(nrs) - 1]
    while len(out) < want and (inside or aligned):
        for pool in (inside, aligned):
            if pool and len(out) < want:
                out.append(pool.pop())
    return sorted(out)


def test_eval_forward_b256_histogram_against_oracle(kg_real):
    cfg = OP.full_cfg()
    m = make_model(cfg, 0, "bf16").eval()
    B = 256
    nrs, rg, kg = _histogram_batch(B, 3, kg_real)
    assert sum(nrs) >= 4 * 22528, "several rounds of 128-row blocks per CU"
    got = _packed_forward(m, rg, nrs, kg)
    assert got.shape == (B, 6) and np.isfinite(got).all()
    picks = _block_boundary_samples(nrs, 128, 16)
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, 0))
    ref, _ = orc.forward_list([rg[b] for b in picks], kg[picks])
    err = np.abs(got[picks] - outs6(ref)).max()
    print(f"B = 256 eval forward (T = {sum(nrs)}): samples {picks}: max |logit err| vs the f32 oracle {err:.2e}")
    assert_close(got[picks], outs6(ref), 1e-3, 0, "B = 256 packed forward vs the f32 oracle")
    singles = np.stack([_packed_forward(m, [rg[b]], [nrs[b]], kg[b:b + 1])[0] for b in picks])
    # (a one-sample call runs the 32-row kernels: the same bf16 operands, another partition of the KG->RG keys into flash
    # blocks and another summation order)
    assert_close(got[picks], singles, 4e-4, 0, "packed == one-by-one")
    # prediction agreement over the whole batch with the one-by-one path on 16 samples is implied; over the oracle's 16:
    assert (got[picks][:, :2].argmax(1) == outs6(ref)[:, :2].argmax(1)).all()


def test_eval_forward_b1024_packed_equals_singles(kg_real):
    cfg = OP.full_cfg()
    m = make_model(cfg, 1, "bf16").eval()
    B = 1024
    nrs, rg, kg = _histogram_batch(B, 4, kg_real)
    got = _packed_forward(m, rg, nrs, kg)
    assert got.shape == (B, 6) and np.isfinite(got).all()
    picks = _block_boundary_samples(nrs, 128, 8)
    singles = np.stack([_packed_forward(m, [rg[b]], [nrs[b]], kg[b:b + 1])[0] for b in picks])
    assert_close(got[picks], singles, 4e-4, 0, "packed == one-by-one")
    ref, _ = FO.FusionOracle(cfg, OP.make_params(cfg, 1)).forward_list([rg[b] for b in picks], kg[picks])
    assert_close(got[picks], outs6(ref), 1e-3, 0, "B = 1024 packed forward vs the f32 oracle")
    # a second, identical call must return the same bits where nothing is summed by atomics in another order, and the same
    # logits to fp32 rounding everywhere (stale arrival counters / partials from the first call would show here)
    again = _packed_forward(m, rg, nrs, kg)
    assert_close(again, got, 2e-5, 0, "second call")


def _train_step_vs_bf16_oracle(B, pseed, kg_real, nrs=None):
    cfg = OP.full_cfg()
    m = make_model(cfg, pseed, "bf16").train()
    eng = m._engine
    if nrs is None:
        nrs, rg, kg = _histogram_batch(B, 10 + pseed, kg_real)
    else:
        rg = [OP.make_rg(n, 128, seed=900 + i) for i, n in enumerate(nrs)]
        kg = np.stack([kg_real] * B)
    y, e, s = OP.make_labels(B, seed=21)
    dseed = 41
    batch = eng.make_batch(torch.from_numpy(np.concatenate(rg)).cuda(), list(nrs), torch.from_numpy(kg).cuda())
    ws = eng.workspace(batch, private=True)
    ws.zero_()
    g = eng.ensure_flat_grads(attach=True)
    g.zero_()
    outs, terms, pred = eng.train_raw(batch, ws, torch.from_numpy(y), torch.from_numpy(e), torch.from_numpy(s), True, dseed, eng._gtab)
    torch.cuda.synchronize()
    grads = {k: t2n(p.grad).copy() for k, p in m.named_parameters()}
    assert np.isfinite(t2n(outs)).all() and all(np.isfinite(v).all() for v in grads.values())
    # (training calls from 57 344 packed rows run their forward on the 64-row half-blocks: the oracle then takes that kernel's flash-block partition)
    ref = oracle_batch_step(lambda: bf16_oracle(cfg, OP.make_params(cfg, pseed), nrs, wide2=sum(nrs) >= 57344), rg, kg, y, e, s, dseed, grads)
    if ref["near"]:
        print(f"B = {B}: tail units at the ReLU threshold (site, sample, unit, pre-activation): {ref['near']}; taken flipped: {ref['flips']}")
    assert_close(t2n(outs), outs6(ref["outs"]), 5e-4, 0, "outputs vs the bf16-operand oracle")
    assert_close(t2n(terms), ref["loss_terms"], 2e-3, 1e-3, "loss terms")
    o6 = outs6(ref["outs"])
    clear = np.abs(o6[:, 0] - o6[:, 1]) > 2e-3                    # (a prediction is only defined where the two mask logits differ by more than the tolerance)
    assert (t2n(pred)[clear] == o6[clear, :2].argmax(1)).all()
    want = ref["raw_grads"]
    den = sum(float((want[k].astype(np.float64) ** 2).sum()) for k in grads)
    num = sum(float(((grads[k].astype(np.float64) - want[k]) ** 2).sum()) for k in grads)
    per = sorted(((float(np.sqrt(((grads[k].astype(np.float64) - want[k]) ** 2).sum() / max((want[k].astype(np.float64) ** 2).sum(), 1e-30))), k)
                  for k in grads if (want[k].astype(np.float64) ** 2).sum() > 1e-6 * den), reverse=True)
    total = float(np.sqrt(num / den))
    print(f"B = {B} (T = {sum(nrs)}) training step: global relative gradient error vs the bf16-operand oracle {total:.5f}; worst {per[:3]}")
    assert total < 2e-3, (total, per[:4])
    assert per[0][0] < 1e-2, per[:4]
    return grads, per


def test_training_step_b70_wide_front_against_bf16_oracle(kg_real):
    """The case of test_large_batch_training_takes_the_wide_front_half (B = 70, 31 k rows: 128-row front blocks, two-plane tail
    forward and backward, parameter-space weight gradients) -- there compared with the HIP path's own 32-row schedule, here
    with the oracle."""
    B = 70
    _train_step_vs_bf16_oracle(B, 4, kg_real, nrs=[380 + 3 * (i % 50) for i in range(B)])


def test_training_step_b100_every_by_size_path_against_bf16_oracle(kg_real):
    """B = 100 of the real histogram (T ~ 48 k >= 40 000: gemm16_tnbig_kernel by size; > 64 samples: the two-plane tail forward
    and backward and the loss launch's several-samples-per-block form by size).  Named tensors of every family are inside the
    per-tensor bound: in-projections, out-projection, FFN, LayerNorm, fusion layer, heads."""
    grads, per = _train_step_vs_bf16_oracle(100, 5, kg_real)
    named = ("fusion.rg_proj.weight", "fusion.cross_attn_rg2kg.in_proj_weight", "fusion.cross_attn_kg2rg.in_proj_weight",
             "fusion.cross_attn_rg2kg.out_proj.weight", "fusion.ffn_rg.0.weight", "fusion.ffn_rg.3.weight", "fusion.ln_rg.weight",
             "fusion.fusion_layer.0.weight", "mask_head.0.weight", "instance_head.3.weight")
    got = {k: r for r, k in per}
    missing = [k for k in named if k not in got]
    assert not missing, f"tensors without weight in the gradient (cannot be judged): {missing}"
    assert all(got[k] < 1e-2 for k in named), {k: got[k] for k in named}


def test_training_step_b124_takes_the_64row_forward_by_size(kg_real):
    """B = 124 of the real histogram (T ~ 59.6 k >= 57 344): by size the training call's forward is rgfwd2_kernel + kgchain_kernel in their saving +
    dropout variants (csrc/fused_wide2.hip) with two blocks per CU, in front of the same backward kernels -- against the oracle's
    train step with the kernel's flash-block partition, same bounds."""
    B = 124
    nrs, _, _ = _histogram_batch(B, 16, kg_real)
    assert sum(nrs) >= 57344, sum(nrs)
    _train_step_vs_bf16_oracle(B, 6, kg_real)
