"""Data-parallel semantics (ddp.py) with world_size 2 over gloo on CPU.

The HIP kernels cannot run here, so the per-rank compute is the CPU oracle; what is under test is
the distributed host logic the product uses unchanged on RCCL: row-balanced sharding of a global
minibatch, ONE flat gradient all-reduce with SUM (not mean) BEFORE the clip, identical parameters on
every rank afterwards, the common-seed sharded sampler and the metric reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fusion_oracle as FO
from oracle import params as OP

CFG = OP.full_cfg(dict(rg_dim=16, kg_dim=16, hidden_dim=32, num_heads=4, dropout=0.0))
NRS = (9, 30, 4, 17, 22, 11)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _global_batch():
    rg = [OP.make_rg(n, 16, seed=i) for i, n in enumerate(NRS)]
    kg = np.stack([OP.make_kg(5, 16, seed=50 + i) for i in range(len(NRS))])
    y, e, s = OP.make_labels(len(NRS), seed=4)
    return rg, kg, y, e, s


def _flat(d, specs):
    return torch.from_numpy(np.concatenate([d[k].reshape(-1) for k, _ in specs]).astype(np.float32))


def _unflat(t, specs):
    out, o = {}, 0
    for k, shape in specs:
        n = int(np.prod(shape)); out[k] = t[o:o + n].numpy().reshape(shape).copy(); o += n
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camouflage_multimodal_amd.ddp import (GradAllReducer, broadcast_parameters, reduce_metrics, shard_by_rows,
                                                   sharded_weighted_sampler)
        specs = OP.param_specs(CFG)
        rg, kg, y, e, s = _global_batch()
        # ranks start from different parameters; broadcast makes them rank 0's
        prm = OP.make_params(CFG, seed=10 + rank)
        flatp = _flat(prm, specs)
        broadcast_parameters(flatp, src=0)
        prm = _unflat(flatp, specs)
        orc = FO.FusionOracle(CFG, prm)
        opt = FO.AdamW(orc.p, lr=5e-4, weight_decay=1e-4)
        mine = shard_by_rows(NRS, world, rank)
        for step in range(2):
            outs, caches = orc.forward_list([rg[i] for i in mine], kg[mine], training=True, seed=0)
            g = orc.zero_grads()
            loss = 0.0
            for j, i in enumerate(mine):
                l, _, d = FO.sample_loss({k: outs[k][j] for k in ("mask", "instance", "edge", "score")}, int(y[i]), float(e[i]), float(s[i]))
                orc.backward_sample(caches[j], d, g); loss += float(l)
            flatg = _flat(g, specs)
            GradAllReducer()(flatg)                                     # SUM over ranks
            g = _unflat(flatg, specs)
            norm = FO.clip_grad_norm(g, 1.0)                            # clip AFTER the reduce, on every rank
            opt.step(orc.p, g)
            m = reduce_metrics(torch.tensor([loss, float(len(mine))], dtype=torch.float64))
        flat_after = _flat(orc.p, specs)
        gathered = [torch.zeros_like(flat_after) for _ in range(world)]
        dist.all_gather(gathered, flat_after)
        idx = sharded_weighted_sampler([1.0, 5.0, 1.0, 2.0, 1.0, 3.0, 1.0, 1.0], 12, epoch=3, world=world, rank=rank, seed=7)
        # an epoch whose length is NOT a multiple of the world size (65 draws, 2 ranks, minibatches of 16): every rank must cut the same
        # number of minibatches -- each optimizer step is a collective, a rank with one step more would wait forever
        odd = sharded_weighted_sampler([1.0] * 9, 65, epoch=1, world=world, rank=rank, seed=7)
        steps = torch.tensor([len(range(0, len(odd), 16)), len(odd)], dtype=torch.int64)
        all_steps = [torch.zeros_like(steps) for _ in range(world)]
        dist.all_gather(all_steps, steps)
        assert all(torch.equal(a, all_steps[0]) for a in all_steps), all_steps
        q.put((rank, mine, flat_after.numpy(), float(norm), m.tolist(), bool(torch.equal(gathered[0], gathered[1])), idx))
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process_reference_semantics():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0

    # single process: the reference schedule on the whole global minibatch
    specs = OP.param_specs(CFG)
    rg, kg, y, e, s = _global_batch()
    orc = FO.FusionOracle(CFG, OP.make_params(CFG, seed=10))
    opt = FO.AdamW(orc.p, lr=5e-4, weight_decay=1e-4)
    losses = 0.0
    for step in range(2):
        r = FO.train_step(orc, opt, rg, kg, y, e, s, training=True, seed=0)
        losses = float(r["losses"].sum())
    want = _flat(orc.p, specs).numpy()

    (r0, mine0, p0, norm0, m0, same0, idx0), (r1, mine1, p1, norm1, m1, same1, idx1) = res
    assert sorted(mine0 + mine1) == list(range(len(NRS))) and not set(mine0) & set(mine1)
    rows0, rows1 = sum(NRS[i] for i in mine0), sum(NRS[i] for i in mine1)
    assert abs(rows0 - rows1) <= max(NRS)                       # balanced by rows, not by count
    assert same0 and same1 and np.array_equal(p0, p1)           # replicas stay bit-identical without a broadcast
    assert abs(norm0 - norm1) == 0.0 and abs(norm0 - float(r["grad_norm"])) < 1e-5 * norm0
    assert np.abs(p0 - want).max() < 2e-6                       # SUM-then-clip == one process with batch_size = 6
    assert m0 == m1 and abs(m0[0] - losses) < 1e-4 and m0[1] == len(NRS)
    # sampler: same sequence on every rank, strided shares, union = the single-process draw
    g = torch.Generator(); g.manual_seed(7 * 1000003 + 3)
    full = torch.multinomial(torch.tensor([1.0, 5.0, 1.0, 2.0, 1.0, 3.0, 1.0, 1.0], dtype=torch.double), 12, replacement=True, generator=g).tolist()
    assert idx0 == full[0::2] and idx1 == full[1::2]


def test_mean_instead_of_sum_would_differ():
    """Guards the semantic: averaging gradients (torch DDP's default) is NOT the reference."""
    rg, kg, y, e, s = _global_batch()
    orc = FO.FusionOracle(CFG, OP.make_params(CFG, seed=10))
    outs, caches = orc.forward_list(rg, kg, training=True, seed=0)
    g = orc.zero_grads()
    for j in range(len(NRS)):
        _, _, d = FO.sample_loss({k: outs[k][j] for k in ("mask", "instance", "edge", "score")}, int(y[j]), float(e[j]), float(s[j]))
        orc.backward_sample(caches[j], d, g)
    n_sum = FO.clip_grad_norm({k: v.copy() for k, v in g.items()}, 1.0)
    n_mean = FO.clip_grad_norm({k: v / 2 for k, v in g.items()}, 1.0)
    assert n_sum > 1.0 and abs(n_mean - n_sum / 2) < 1e-6 * n_sum   # the clip sees a different norm => different update


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camouflage_multimodal_amd.ddp import BucketedGradAllReducer, GradAllReducer, OneShotGradAllReducer
        g = torch.Generator().manual_seed(7 + rank)
        flat = torch.randn(1000, generator=g)
        a, b, c = flat.clone(), flat.clone(), flat.clone()
        d = flat[:997].clone()                             # (a length the world size does not divide)
        one = OneShotGradAllReducer()
        one(d); one(d_again := flat[:997].clone())
        GradAllReducer()(a)
        red = BucketedGradAllReducer()
        assert red.tail_event("cpu") is None              # no GPU: no event, the two buckets are reduced one after the other
        red(b, split=640)
        red(c, split=None)                                 # no split point (late fusion): one piece
        assert torch.equal(d, d_again)
        q.put((rank, a.numpy(), b.numpy(), c.numpy(), d.numpy()))
    finally:
        dist.destroy_process_group()


def test_bucketed_all_reduce_is_the_same_sum():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60); assert p.exitcode == 0
    for rank, a, b, c, d in res:
        assert np.array_equal(a, b) and np.array_equal(a, c)
        assert np.allclose(d, a[:997], rtol=0, atol=1e-6)  # all_to_all + local sum + all_gather: the same sum
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][4], res[1][4])
