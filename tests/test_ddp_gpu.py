"""The PRODUCT data-parallel path with two ranks: NativeTrainer + GradAllReducer + FusedClipAdamW,
one process per rank, both ranks sharing the single GPU of the test box over the gloo backend (RCCL needs one GPU per
rank; the host logic -- SUM all-reduce of the flat gradient buffer before the clip, identical optimizer step on every
rank -- is the same).  Needs an MI355X.

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


def _worker(rank, world, port, q, precision):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camouflage_multimodal_amd import NativeTrainer
        from camouflage_multimodal_amd.ddp import GradAllReducer, broadcast_parameters, shard_by_rows
        torch.cuda.set_device(0)
        m = _build(precision)
        broadcast_parameters(m._engine.flat_params)
        tr = NativeTrainer(m, grad_allreduce=GradAllReducer())
        seed_folded = tr.engine._seed_base
        mine = shard_by_rows(NRS, world, rank)
        for step in range(2):
            _step(tr, mine, step)
        torch.cuda.synchronize()
        q.put((rank, mine, m._engine.flat_params.cpu().numpy(), float(tr.opt.grad_norm().item()), seed_folded))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["f32"])
def test_two_rank_product_path_equals_single_process(precision):
    import torch.multiprocessing as mp
    if torch.cuda.is_initialized():
        pytest.skip("HIP is already initialised in this process: this test must run before any other GPU test "
                    "(tests/conftest.py orders it first)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, precision)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(120); assert p.exitcode == 0
    (r0, mine0, p0, n0, sd0), (r1, mine1, p1, n1, sd1) = res
    assert sorted(mine0 + mine1) == list(range(len(NRS))) and not set(mine0) & set(mine1)
    assert np.array_equal(p0, p1) and n0 == n1           # replicas bit-identical after two steps, no parameter broadcast
    assert sd0 != sd1                                    # but each rank draws its own dropout masks
    # single process, global batch = the reference with batch_size = 8
    from camouflage_multimodal_amd import NativeTrainer
    m = _build(precision)
    tr = NativeTrainer(m)
    for step in range(2):
        _step(tr, list(range(len(NRS))), step)
    torch.cuda.synchronize()
    want = m._engine.flat_params.cpu().numpy()
    assert abs(float(tr.opt.grad_norm().item()) - n0) < 2e-4 * n0
    err = np.abs(p0 - want)
    # Adam moves an element whose gradient is rounding noise by up to ~lr per step whatever its sign (helpers.assert_params_close)
    assert err.max() <= 2.2 * 5e-4 * 2 and (err <= 5e-6 + 1e-5 * np.abs(want)).mean() > 0.99, (err.max(), (err <= 5e-6).mean())
