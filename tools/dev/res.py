"""Developer aid: per-kernel VGPRs / scratch bytes / SGPR spills of one csrc file under the build's flags.
  python tools/dev/res.py fused_rows.hip"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from camouflage_multimodal_amd import build as B
src = sys.argv[1]
r = subprocess.run([B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), "-c", os.path.join(B.CSRC, src), "-o", "/tmp/res_tmp.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
name = None; row = {}
for l in r.stderr.splitlines():
    if "error" in l: print(l)
    m = re.search(r"Function Name: (\S+)", l)
    if m: name = m.group(1).replace("_ZN12_GLOBAL__N_1", ""); row = {}
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, l)
        if m: row[key] = int(m.group(1))
    if "LDS Size" in l and name:
        print(f"{name[:60]:60s} vgpr {row.get('vgpr'):4d} scratch {row.get('scratch'):4d} sgpr-spill {row.get('sspill'):3d} vgpr-spill {row.get('vspill'):3d}"); name = None
