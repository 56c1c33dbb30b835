#!/usr/bin/env python3
"""Headline benchmark of the fusion hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 16] [--precision bf16|f32]

N>1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
(one rank per GPU, RCCL).  A "step" is one optimizer step of the training path on one packed
minibatch of synthetic inputs resident in HBM:
    forward -> 4-term loss -> backward -> [grad all-reduce SUM] -> clip_grad_norm(1.0) -> AdamW
in train mode with the reference's dropout 0.3 (models/multimodal/train_multimodal.py:238-279).

Workload = BASELINE.json configs[1] expressed on the path that exists: the
reference has no RGB+D image encoder; one "image" is one sample = (RG node embeddings [Nr,128],
Nr drawn from the real 303..530 histogram, 13 KG category embeddings [13,128]); batch 16 per GPU,
bf16 MFMA operands with fp32 accumulation/activations.  Weak scaling: 16 samples per GPU.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the grouped MFMA GEMM --
gemm16_kernel on bf16-resident operands in bf16 mode, gemm_grouped_kernel in f32 mode; >= 98 % of the
FLOPs) with HIP events recorded on its launch stream during the timed steps;
`cpu_baseline` times the CPU oracle (a numpy port of the reference path) on the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_MFLOP_PER_ROW, FWD_MFLOP_CONST = 1.1407, 15.14      # fwd = 1.1407*Nr + 15.14 MFLOP (Nk = 13)
FWDBWD_OVER_FWD = 1722.9 / 585.5                        # 585.5 MFLOP fwd, 1722.9 fwd+bwd at Nr = 500
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}            # MI355X_MICROARCH.md: dense MFMA peaks


def algorithmic_flops(nrs):
    return sum((FWD_MFLOP_PER_ROW * n + FWD_MFLOP_CONST) * 1e6 * FWDBWD_OVER_FWD for n in nrs)


def load_fixture(name):
    p = os.path.join(ROOT, "tests", "golden", name)
    return np.load(p, allow_pickle=False) if os.path.exists(p) else None


def make_batches(n_batches, B, rank, seed=0):
    """Synthetic minibatches: rg ~ |N(0,1)|*0.3, Nr from the real histogram, kg = the 13
    shipped KG vectors (fixture copy), labels y~Bern(.5), e in {0,1}, s~U(0,1)."""
    rs = np.random.RandomState(1234 + 7919 * rank + seed)
    h = load_fixture("nr_histogram.npz")
    kgf = load_fixture("kg_embeddings.npz")
    kg1 = kgf["kg"].astype(np.float32) if kgf is not None else (np.abs(rs.standard_normal((13, 128))) * 0.3).astype(np.float32)
    out = []
    for _ in range(n_batches):
        if h is not None:
            nrs = [int(x) for x in rs.choice(h["values"], size=B, p=h["counts"] / h["counts"].sum())]
        else:
            nrs = [int(x) for x in rs.randint(303, 531, size=B)]
        rg = (np.abs(rs.standard_normal((sum(nrs), 128))) * 0.3).astype(np.float32)
        kg = np.broadcast_to(kg1, (B, 13, 128)).copy()
        y = (rs.uniform(size=B) < 0.5).astype(np.int64)
        e = (rs.uniform(size=B) < 0.5).astype(np.float32)
        s = rs.uniform(size=B).astype(np.float32)
        out.append((rg, nrs, kg, y, e, s))
    return out


def cpu_baseline(batches, budget_s=15.0):
    """The CPU oracle (oracle/: numpy port of the reference path, pinned to the reference's golden
    vectors) running the same training step -- per-sample forward/backward, gradient sum, clip,
    AdamW -- on the host cores.  A reported baseline, not the optimisation target."""
    from oracle import fusion_oracle as FO
    from oracle import params as OP
    cfg = OP.full_cfg()
    orc = FO.FusionOracle(cfg, OP.make_params(cfg, 0))
    opt = FO.AdamW(orc.p)
    threads = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        blas = [i["num_threads"] for i in threadpool_info() if i.get("user_api") == "blas"]
        if blas:
            threads = max(blas)
    except Exception:
        pass
    rg, nrs, kg, y, e, s = batches[0]
    split = lambda: np.split(rg, np.cumsum(nrs)[:-1])
    FO.train_step(orc, opt, split()[:2], kg[:2], y[:2], e[:2], s[:2], training=True, seed=1)   # warm-up
    n, t0 = 0, time.perf_counter()
    steps = 0
    while True:
        rg, nrs, kg, y, e, s = batches[steps % len(batches)]
        FO.train_step(orc, opt, np.split(rg, np.cumsum(nrs)[:-1]), kg, y, e, s, training=True, seed=steps)
        n += len(nrs); steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 50:
            break
    return {"value": round(n / el, 2), "unit": "images/s", "cores": int(threads), "kind": "port",
            "sample": f"{steps} optimizer steps x {len(nrs)} samples of the same synthetic workload "
                      f"({el:.1f} s of numpy/BLAS work, fp32, dropout 0.3 via the shared counter hash)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="samples per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event roofline leg")
    ap.add_argument("--backend", default="nccl", help="process-group backend for --gpus N > 1 (nccl = RCCL; gloo lets a "
                    "one-GPU box rehearse the multi-rank path with every rank on the same device)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from camouflage_multimodal_amd import NativeTrainer, build_multimodal_model, _lib
    from camouflage_multimodal_amd.ddp import GradAllReducer, broadcast_parameters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    torch.manual_seed(0)
    model = build_multimodal_model({}).to(dev).set_precision(args.precision).train()   # reference defaults, dropout 0.3
    if world > 1:
        broadcast_parameters(model._engine.flat_params)
    trainer = NativeTrainer(model, lr=5e-4, weight_decay=1e-4, grad_allreduce=GradAllReducer() if world > 1 else None)

    host = make_batches(8, args.batch, rank)
    batches = [(torch.from_numpy(rg).to(dev), nrs, torch.from_numpy(kg).to(dev), torch.from_numpy(y).to(dev),
                torch.from_numpy(e).to(dev), torch.from_numpy(s).to(dev)) for rg, nrs, kg, y, e, s in host]

    def run(k, start=0):
        for i in range(start, start + k):
            trainer.step(*batches[i % len(batches)])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg: same K steps again with HIP events around every launch of the dominant kernel
    # (every rank runs these steps -- they contain the gradient all-reduce -- only rank 0 records events)
    roof = None
    if not args.no_kernel_timing:
        L = _lib.lib()
        k = min(args.steps, 50)
        if rank == 0:
            _lib.check(L.camo_prof_begin(64 * k), "camo_prof_begin")
        t1 = time.perf_counter()
        run(k, args.warmup)
        torch.cuda.synchronize()
        t_prof = time.perf_counter() - t1
    if rank == 0 and not args.no_kernel_timing:
        ms, n, fl = C.c_double(), C.c_int32(), C.c_double()
        _lib.check(L.camo_prof_end(C.byref(ms), C.byref(n), C.byref(fl)), "camo_prof_end")
        alg = sum(algorithmic_flops(batches[i % len(batches)][1]) for i in range(args.warmup, args.warmup + k))
        # An event pair also times its own marker packets: an EMPTY pair on the same stream is reported next to the
        # figure (not subtracted: it over-corrects -- rocprofv3's kernel-only average, profiles/, sits in between).
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for a, b in pairs:
            a.record(); b.record()
        torch.cuda.synchronize()
        ev_us = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)[len(pairs) // 2]
        gemm_s = ms.value * 1e-3
        achieved = alg / gemm_s / 1e12
        peak = PEAK_TFLOPS[args.precision]
        # HBM bytes of the same kernel from the PMC passes (profiles/summarize_pmc.py), per launch
        traffic, tnote = None, None
        kname = "gemm16_kernel" if args.precision == "bf16" else "gemm_grouped_kernel"
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pm = None
        if os.path.exists(pj):
            with open(pj) as f:
                pm = json.load(f)
        if pm is not None and pm.get("kernel") == kname:
            traffic = round(pm["hbm_bytes_per_launch"])
            tnote = {"hbm_bytes_per_step": round(pm["hbm_bytes_per_step"]), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                     "separate passes of this bench (profiles/pmc_traffic.json): " + pm["correction"]}
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 5), "traffic": traffic, "traffic_note": tnote,
                "kernel": kname + ("<bf16-resident operands>" if args.precision == "bf16" else "<f32>"),
                "launches_per_step": round(n.value / k, 1),
                "avg_launch_us": round(ms.value * 1e3 / max(n.value, 1), 2),
                "empty_event_pair_us": round(ev_us, 2),
                "kernel_ms_per_step": round(ms.value / k, 4),
                "algorithmic_gflop_per_step": round(alg / k / 1e9, 3),
                "executed_gflop_per_step": round(fl.value / k / 1e9, 3),
                "share_of_step_time": round(gemm_s / t_prof, 3)}
        if traffic:      # the same launches seen from the other roof: measured HBM bytes / launch time vs 8 TB/s
            gbps = traffic / (ms.value * 1e-3 / max(n.value, 1)) / 1e9
            roof["hbm_view"] = {"achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        total = args.batch * world * args.steps
        nr_mean = float(np.mean([n for b in host for n in b[1]]))
        line = {
            "metric": "images/sec fwd+bwd (fusion model train step: fwd + loss + bwd + clip + AdamW)",
            "value": round(total / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] on the path that exists (SURVEY 8d): cross-attention fusion "
                                   "model, packed variable-Nr minibatch (Nr ~ real 303..530 histogram, mean %.0f), "
                                   "Nk=13 real KG rows, train mode dropout 0.3, random-init weights" % nr_mean,
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "rg_dim": 128, "hidden_dim": 256,
                       "num_heads": 8, "parallelism": f"dp{world}",
                       "precision": "bf16 MFMA operands, fp32 accumulate/activations/optimizer" if args.precision == "bf16"
                                    else "fp32 (f32-input MFMA)"},
        }
        if roof is not None:
            line["roofline"] = roof
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(host)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
