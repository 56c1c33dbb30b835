#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes of bench.py (one --pmc FETCH_SIZE, one --pmc WRITE_SIZE, each with
--kernel-trace --output-format csv, as MI355X_MICROARCH.md prescribes: separate passes) into
profiles/pmc_traffic.json, which bench.py reports as roofline.traffic for its dominant kernel.

  python profiles/summarize_pmc.py <fetch_dir> <write_dir>

gfx950 correction (guide, 'HBM'): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced
reads, i.e. reports half the bytes -> doubled here; WRITE_SIZE is exact for 16-B/lane stores and
float atomics.  Both counters are in KB.  Per kernel (short name: namespace, template arguments and
argument list stripped): launches, bytes per launch; plus the all-kernel bytes per step, a step
being one launch of tail_fused_kernel (once per training step of the fused schedule at B <= 16).
"""
import csv, glob, json, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    return re.split(r"[<(]", name, 1)[0].strip()


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                out.setdefault(k, [0.0, 0])
                out[k][0] += float(r["Counter_Value"]); out[k][1] += 1
    return out


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    steps_f = max(fe.get("tail_fused_kernel", [0, 1])[1], 1); steps_w = max(wr.get("tail_fused_kernel", [0, 1])[1], 1)
    kernels, per_step = {}, 0.0
    for k in sorted(set(fe) | set(wr)):
        f_kb, nf = fe.get(k, [0.0, 0]); w_kb, nw = wr.get(k, [0.0, 0])
        per_launch = 2.0 * f_kb * 1024 / max(nf, 1) + w_kb * 1024 / max(nw, 1)      # (the two passes ran different numbers of steps)
        lps = nf / steps_f if nf else nw / steps_w
        kernels[k] = {"launches_per_step": round(lps, 3), "fetch_size_kb_per_launch_raw": round(f_kb / max(nf, 1), 2),
                      "write_size_kb_per_launch": round(w_kb / max(nw, 1), 2), "hbm_bytes_per_launch": round(per_launch)}
        per_step += per_launch * lps
    res = {"steps_profiled": {"fetch_pass": steps_f, "write_pass": steps_w}, "kernels": kernels,
           "all_kernels_hbm_bytes_per_step": round(per_step),
           "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests at 64 B); WRITE_SIZE as reported",
           "source": {"per_kernel_records": "profiles/%s_pmc_bench/FETCH_SIZE_per_kernel.csv, WRITE_SIZE_per_kernel.csv" % (re.search(r"(r\d+)_pmc_bench", fetch_dir).group(1) if re.search(r"(r\d+)_pmc_bench", fetch_dir) else "rNN"),
                      "fetch_pass": fetch_dir, "write_pass": write_dir, "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline (tools/dev/dev_final_run.sh)"}}
    json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
