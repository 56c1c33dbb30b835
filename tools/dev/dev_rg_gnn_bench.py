"""Developer aid / measurement of the Region-Graph GNN embedding path: graphs per second for
batches of region-adjacency-like graphs of ~500 nodes, and the achieved HBM rate against the algorithmic bytes.
  python tools/dev/dev_rg_gnn_bench.py [graphs_per_batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from camouflage_multimodal_amd import RegionGraphGNN, build_target_csr
from oracle import rg_gnn_oracle as RO

G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = RegionGraphGNN().cuda().eval()
gs = [RO.make_graph(int(n), seed=i) for i, n in enumerate(np.random.RandomState(0).randint(303, 531, size=G))]
off = np.cumsum([0] + [g[0].shape[0] for g in gs])
x = torch.from_numpy(np.concatenate([g[0] for g in gs])).cuda()
ei = torch.from_numpy(np.concatenate([g[1] + off[i] for i, g in enumerate(gs)], axis=1)).cuda()
ew = torch.from_numpy(np.concatenate([g[2] for g in gs])).cuda()
N, E = x.shape[0], ei.shape[1] + x.shape[0]
for _ in range(5): m.extract_node_embeddings(x=x, edge_index=ei, edge_attr=ew)
torch.cuda.synchronize()
it = 50
t0 = time.perf_counter()
for _ in range(it): m.extract_node_embeddings(x=x, edge_index=ei, edge_attr=ew)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / it
# algorithmic bytes per call: x, the projected features written once and gathered once per edge, per-layer outputs
C, K = 128, 4
alg = 4 * (N * 15 + N * K * C + E * K * C + 3 * (N * C + E * C + N * C) + 2 * N * C) + E * 12
print(f"{G} graphs / {N} nodes / {E} edges per call: {dt * 1e6:.1f} us per call = {G / dt:.0f} graphs/s; "
      f"algorithmic {alg / 1e6:.1f} MB -> {alg / dt / 1e9:.0f} GB/s ({alg / dt / 8e12 * 100:.2f} % of 8 TB/s; includes the host-side CSR build)")
