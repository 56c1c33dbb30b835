"""Developer aid (not collected by pytest): compares saved activations in the workspace with
the oracle's cache for one small batch.  python tools/dev/debug_intermediates.py"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from camouflage_multimodal_amd import build_multimodal_model, _lib
from oracle import fusion_oracle as FO, params as OP

cfg = OP.full_cfg(dict(dropout=0.0))
prm = OP.make_params(cfg, 0)
m = build_multimodal_model(cfg); m.load_state_dict({k: torch.from_numpy(v) for k, v in prm.items()})
m = m.cuda().set_precision("f32").eval()
nrs = [40, 70]
rg = [OP.make_rg(n, 128, seed=i) for i, n in enumerate(nrs)]
kg = np.stack([OP.make_kg(13, 128, seed=5 + i) for i in range(2)])
eng = m._engine
b = eng.make_batch(torch.from_numpy(np.concatenate(rg)).cuda(), nrs, torch.from_numpy(kg).cuda())
ws = eng.workspace(b)
outs, attn = eng.forward_raw(b, ws, False, 0, want_attention=True)
torch.cuda.synchronize()
orc = FO.FusionOracle(cfg, prm)
ref, caches = orc.forward_list(rg, kg)
H = 256; T = sum(nrs); TK = 26

def get(name, shape):
    off = _lib.lib().camo_debug_ws_offset(C.byref(eng.dims), b.B, b.T, b.Nk, name.encode())
    assert off >= 0, name
    n = int(np.prod(shape))
    return ws[off:off + 4 * n].view(torch.float32).view(*shape).cpu().numpy()

def cat(key): return np.concatenate([c[key] for c in caches])
checks = [("R", (T, H), cat("R")), ("G", (TK, H), cat("G")), ("Q", (T, H), cat("Q")),
          ("KV2", (T, 2 * H), np.concatenate([cat("K2"), cat("V2")], 1)), ("KV", (TK, 2 * H), np.concatenate([cat("Kk"), cat("Vk")], 1)),
          ("Q2", (TK, H), cat("Q2")), ("P", (T, 8, 13), cat("Pm")), ("O", (T, H), cat("O")), ("P2", (T, 8, 13), cat("P2")),
          ("O2", (TK, H), cat("O2")), ("Y", (T, H), cat("Y")), ("Y2", (TK, H), cat("Y2")), ("H1", (T, 2 * H), cat("H1d")),
          ("H2", (TK, 2 * H), cat("H2d")), ("comb", (2, 2 * H), cat("comb")), ("fused", (2, H), cat("fused"))]
for name, shape, want in checks:
    got = get(name, shape)
    print(f"{name:6s} max|err| {np.abs(got - want.reshape(shape)).max():.3e}   (|want| max {np.abs(want).max():.3f})")
print("outs", outs.cpu().numpy(), "\nref ", np.concatenate([ref[k] for k in ("mask", "instance", "edge", "score")], 1))
