"""Developer aid: host-side profile of the epoch loop (where the Python time of a training step goes)."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import load_fixture
from camouflage_multimodal_amd import DeviceResidentDataset, NativeTrainer, build_multimodal_model
from camouflage_multimodal_amd.ddp import sharded_weighted_sampler
from camouflage_multimodal_amd.train_multimodal import train_epoch_fixed
dev = torch.device("cuda", 0)
model = build_multimodal_model({}).to(dev).set_precision("bf16").train()
trainer = NativeTrainer(model)
rs = np.random.RandomState(5); hist = load_fixture("nr_histogram.npz")
nr_all = rs.choice(hist["values"], size=1024, p=hist["counts"] / hist["counts"].sum())
kg1 = load_fixture("kg_embeddings.npz")["kg"].astype(np.float32)
samples = [dict(rg_node_emb=torch.from_numpy((np.abs(rs.standard_normal((int(n), 128))) * 0.3).astype(np.float32)), kg_emb=torch.from_numpy(kg1)[:, None, :],
                mask_label=int(rs.uniform() < 0.5), edge_label=float(rs.uniform() < 0.5), score_label=float(rs.uniform())) for n in nr_all]
ds = DeviceResidentDataset(samples, dev, augment=True, seed=0)
def loader(ep):
    draw = sharded_weighted_sampler([1.0] * len(ds), len(ds), ep, 1, 0, seed=0)
    dd = torch.tensor(draw, device=dev)
    return (ds.batch(draw[i:i + 16], idx_dev=dd[i:i + 16]) for i in range(0, len(draw), 16))
train_epoch_fixed(model, loader(0), trainer, dev, 1); torch.cuda.synchronize()
t0 = time.perf_counter(); train_epoch_fixed(model, loader(1), trainer, dev, 2); torch.cuda.synchronize()
print("epoch of 64 steps: %.1f us per step" % ((time.perf_counter() - t0) / 64 * 1e6))
pr = cProfile.Profile(); pr.enable(); train_epoch_fixed(model, loader(2), trainer, dev, 3); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
