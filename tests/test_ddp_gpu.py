"""The PRODUCT data-parallel path with two ranks: NativeTrainer + GradAllReducer /
BucketedGradAllReducer + FusedClipAdamW, one process per rank, both ranks sharing the single GPU of the test box over the
gloo backend (RCCL needs one GPU per rank; the host logic -- SUM all-reduce of the flat gradient buffer before the clip,
in one piece or in two buckets with the first one issued behind the training call's tail event on a side stream,
identical optimizer step on every rank -- is the same).  Needs an MI355X.

The children are spawned BEFORE this process touches the GPU (a GPU-initialised process must not fork+exec on this pool):
tests/conftest.py moves this module to the front of the run, and the test skips itself if HIP is already initialised."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import params as OP

pytestmark = pytest.mark.gpu

CFG = OP.full_cfg(dict(dropout=0.0))
NRS = (303, 40, 481, 64, 530, 129, 7, 350)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _global_batch(step):
    rg = [OP.make_rg(n, 128, seed=1000 * step + i) for i, n in enumerate(NRS)]
    kg = np.stack([OP.make_kg(13, 128, seed=1000 * step + 500 + i) for i in range(len(NRS))])
    y, e, s = OP.make_labels(len(NRS), seed=40 + step)
    return rg, kg, y, e, s


def _build(precision):
    from camouflage_multimodal_amd import build_multimodal_model
    m = build_multimodal_model(CFG)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in OP.make_params(CFG, 5).items()}, strict=True)
    return m.to("cuda:0").set_precision(precision).train()


def _step(tr, idx, step):
    rg, kg, y, e, s = _global_batch(step)
    nrs = [NRS[i] for i in idx]
    tr.step(torch.from_numpy(np.concatenate([rg[i] for i in idx])).cuda(), nrs, torch.from_numpy(kg[idx]).cuda(),
            torch.from_numpy(y[idx]), torch.from_numpy(e[idx]), torch.from_numpy(s[idx]))


CONFIGS = (("f32", "single"), ("f32", "bucketed"), ("bf16", "bucketed"))


def _worker(rank, world, port, q, precision, reducer="single"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camouflage_multimodal_amd import NativeTrainer
        from camouflage_multimodal_amd.ddp import BucketedGradAllReducer, GradAllReducer, broadcast_parameters, shard_by_rows
        torch.cuda.set_device(0)
        m = _build(precision)
        broadcast_parameters(m._engine)
        tr = NativeTrainer(m, grad_allreduce=BucketedGradAllReducer() if reducer == "bucketed" else GradAllReducer())
        seed_folded = tr.engine._seed_base
        mine = shard_by_rows(NRS, world, rank)
        for step in range(2):
            _step(tr, mine, step)
        torch.cuda.synchronize()
        q.put((rank, mine, m._engine.flat_params.cpu().numpy(), float(tr.opt.grad_norm().item()), seed_folded))
    finally:
        dist.destroy_process_group()


def _poison_worker(rank, world, port, q, reducer):
    """Three steps in bf16 mode (one-launch tail); on rank 1 the tail of step 1 times out (developer hook).  Reports the
    parameters after step 0, after step 1 and after step 2, and step 1's gradient norm."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camouflage_multimodal_amd import NativeTrainer, _lib
        from camouflage_multimodal_amd.ddp import BucketedGradAllReducer, GradAllReducer, broadcast_parameters, shard_by_rows
        torch.cuda.set_device(0)
        m = _build("bf16")
        broadcast_parameters(m._engine)
        tr = NativeTrainer(m, grad_allreduce=BucketedGradAllReducer() if reducer == "bucketed" else GradAllReducer())
        mine = shard_by_rows(NRS, world, rank)
        snaps, norms = [], []
        for step in range(3):
            if step == 1 and rank == 1:
                m._engine.set_option("tail_skip_arrival", 6)     # (one shot, this engine's next call)
            _step(tr, mine, step)
            torch.cuda.synchronize()
            snaps.append(m._engine.flat_params.cpu().numpy().copy()); norms.append(float(tr.opt.grad_norm().item()))
        q.put((rank, snaps, norms, _lib.tail_timeouts()))
    finally:
        dist.destroy_process_group()


def _collect(q, procs, limit=300.0):
    """One result per child; a child that dies (or the time limit) fails the test at once instead of blocking on the queue."""
    import queue, time
    res, t0 = [], time.time()
    while len(res) < len(procs):
        try:
            res.append(q.get(timeout=2.0))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > limit:
                for p in procs:
                    if p.is_alive(): p.terminate()
                pytest.fail(f"data-parallel worker failed (exit codes {[p.exitcode for p in procs]}, {time.time() - t0:.0f} s)")
    for p in procs:
        p.join(120); assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def test_two_rank_product_path_equals_single_process():
    import torch.multiprocessing as mp
    if torch.cuda.is_initialized():
        pytest.skip("HIP is already initialised in this process: this test must run before any other GPU test "
                    "(tests/conftest.py orders it first)")
    ctx = mp.get_context("spawn")
    results = {}
    for precision, reducer in CONFIGS:                   # every pair of children runs before this process touches the GPU
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q, precision, reducer)) for r in range(2)]
        for p in procs: p.start()
        results[(precision, reducer)] = _collect(q, procs)
    poison = {}
    for reducer in ("single", "bucketed"):               # (a tail timeout on ONE rank under data parallelism)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_poison_worker, args=(r, 2, port, q, reducer)) for r in range(2)]
        for p in procs: p.start()
        poison[reducer] = _collect(q, procs)
    for reducer, ((_, s0, n0, t0), (_, s1, n1, t1)) in poison.items():
        # rank 1's tail gave up in step 1: BOTH ranks must skip that step (its garbage gradients were summed into both buffers)
        assert t1 > 0 and t0 == 0, (reducer, t0, t1)
        assert not np.isfinite(n0[1]) and not np.isfinite(n1[1]), (reducer, n0, n1)
        for k in range(3):
            assert np.array_equal(s0[k], s1[k]), f"{reducer}: replicas differ after step {k}"
        assert np.array_equal(s0[1], s0[0]), f"{reducer}: the poisoned step was applied"
        assert not np.array_equal(s0[2], s0[1]) and np.isfinite(s0[2]).all() and np.isfinite(n0[2]), f"{reducer}: the step after it must be a normal one"
    from camouflage_multimodal_amd import NativeTrainer
    want, norm = {}, {}
    for precision in ("f32", "bf16"):                    # single process, global batch = the reference with batch_size = 8
        m = _build(precision)
        tr = NativeTrainer(m)
        for step in range(2):
            _step(tr, list(range(len(NRS))), step)
        torch.cuda.synchronize()
        want[precision] = m._engine.flat_params.cpu().numpy(); norm[precision] = float(tr.opt.grad_norm().item())
    for (precision, reducer), res in results.items():
        (r0, mine0, p0, n0, sd0), (r1, mine1, p1, n1, sd1) = res
        assert sorted(mine0 + mine1) == list(range(len(NRS))) and not set(mine0) & set(mine1)
        assert np.array_equal(p0, p1) and n0 == n1       # replicas bit-identical after two steps, no parameter broadcast
        assert sd0 != sd1                                # but each rank draws its own dropout masks
        err = np.abs(p0 - want[precision])
        if precision == "f32":
            assert abs(norm[precision] - n0) < 2e-4 * n0
            # Adam moves an element whose gradient is rounding noise by up to ~lr per step whatever its sign (helpers.assert_params_close)
            assert err.max() <= 2.2 * 5e-4 * 2 and (err <= 5e-6 + 1e-5 * np.abs(want[precision])).mean() > 0.99, (reducer, err.max(), (err <= 5e-6).mean())
        else:                                            # bf16 operands: the two packings round the same products, sums differ in order
            assert abs(norm[precision] - n0) < 5e-3 * n0
            assert err.max() <= 2.2 * 5e-4 * 2 and (err <= 1e-4 + 1e-3 * np.abs(want[precision])).mean() > 0.97, (reducer, err.max(), err.mean())
    # the two reducers are the same sum (two runs agree to the run-to-run noise of the fp32 atomics, not bit for bit)
    a, b = results[("f32", "single")][0][2], results[("f32", "bucketed")][0][2]
    assert (np.abs(a - b) <= 5e-6 + 1e-5 * np.abs(a)).mean() > 0.99 and np.abs(a - b).max() <= 2.2 * 5e-4 * 2
