#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes of bench.py (one --pmc FETCH_SIZE, one --pmc WRITE_SIZE, each with
--kernel-trace --output-format csv, as MI355X_MICROARCH.md prescribes: separate passes) into
profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.

  python profiles/summarize_pmc.py <fetch_dir> <write_dir> <steps+warmup> [kernel substring]

gfx950 correction (guide, 'HBM'): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced
reads, i.e. reports half the bytes -> doubled here; WRITE_SIZE is exact for 16-B/lane stores and
float atomics.  Both counters are in KB.
"""
import csv, glob, json, sys

def per_kernel(d, name):
    out = {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"]
                out.setdefault(k, [0.0, 0])
                out[k][0] += float(r["Counter_Value"]); out[k][1] += 1
    return out

def main():
    fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    sub = sys.argv[4] if len(sys.argv) > 4 else "gemm_grouped_kernel"
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    f_kb = sum(v[0] for k, v in fe.items() if sub in k); launches = sum(v[1] for k, v in fe.items() if sub in k)
    w_kb = sum(v[0] for k, v in wr.items() if sub in k)
    all_f = sum(v[0] for v in fe.values()); all_w = sum(v[0] for v in wr.values())
    res = {
        "kernel": sub, "steps_profiled": steps, "launches_per_step": launches / steps,
        "fetch_size_kb_per_step_raw": f_kb / steps, "write_size_kb_per_step": w_kb / steps,
        "hbm_bytes_per_step": (2.0 * f_kb + w_kb) * 1024 / steps,
        "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024 / max(launches, 1),
        "all_kernels_hbm_bytes_per_step": (2.0 * all_f + all_w) * 1024 / steps,
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests at 64 B); WRITE_SIZE as reported",
    }
    json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
