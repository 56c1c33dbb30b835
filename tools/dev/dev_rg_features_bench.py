"""Developer aid / measurement of the region-graph construction: images per second for
create_region_graph_from_segments on a 256 x 256 image with 500 superpixels, the oracle (= the reference's per-region
numpy/scipy loop given slic and canny) beside it.
  python tools/dev/dev_rg_features_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from camouflage_multimodal_amd import RegionGraphGNN, create_region_graph_from_segments
from oracle import rg_features_oracle as RO

rs = np.random.RandomState(0)
seg = RO.voronoi_segments(256, 256, 500, 0)
img = rs.uniform(0, 1, (256, 256, 3)).astype(np.float32)
can = rs.uniform(0, 1, (256, 256)) > 0.85
t0 = time.perf_counter(); RO.region_graph(img, seg, can); cpu = time.perf_counter() - t0
dimg, dseg, dcan = torch.from_numpy(img).cuda(), torch.from_numpy(seg).cuda(), torch.from_numpy(can).cuda()
gnn = RegionGraphGNN().cuda().eval()
for _ in range(5):
    data, _ = create_region_graph_from_segments(dimg, dseg, dcan); gnn.extract_node_embeddings(data)
torch.cuda.synchronize()
it = 50
t0 = time.perf_counter()
for _ in range(it): data, _ = create_region_graph_from_segments(dimg, dseg, dcan)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / it
t0 = time.perf_counter()
for _ in range(it):
    data, _ = create_region_graph_from_segments(dimg, dseg, dcan); emb = gnn.extract_node_embeddings(data)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / it
alg = 256 * 256 * (12 + 4 + 1) + 500 * 15 * 4 + data.edge_index.shape[1] * 20
print(f"region graph (500 regions, {data.edge_index.shape[1]} directed edges): {dt * 1e6:.0f} us per image on the device "
      f"(inputs resident; includes the host read-back of the two counts), {1 / dt:.0f} images/s; with the GNN embedding {dt2 * 1e6:.0f} us; "
      f"algorithmic {alg / 1e6:.2f} MB -> {alg / dt / 1e9:.1f} GB/s; oracle (numpy/scipy loop of the reference) {cpu:.2f} s per image on one core")
