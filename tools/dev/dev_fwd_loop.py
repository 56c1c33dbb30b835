"""Developer aid for rocprofv3: N eval forwards (or training steps) at batch B.   python tools/dev/dev_fwd_loop.py B rt [train] [N]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_batches
from camouflage_multimodal_amd import NativeTrainer, _lib, build_multimodal_model
B, rt = int(sys.argv[1]), int(sys.argv[2]); train = len(sys.argv) > 3 and sys.argv[3] == "train"; N = int(sys.argv[4]) if len(sys.argv) > 4 else 50
model = build_multimodal_model({}).cuda().set_precision("bf16")
model.train(train)
tr = NativeTrainer(model)
_lib.check(_lib.lib().camo_debug_set_option(b"fused_rt", rt), "opt")
hb = make_batches(2, B, 0, seed=100 + B)
db = [tuple(torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in b) for b in hb]
for i in range(N):
    tr.step(*db[i % 2]) if train else tr.evaluate(*db[i % 2][:3])
torch.cuda.synchronize()
