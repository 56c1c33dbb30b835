#!/usr/bin/env python3
"""Reduces a rocprofv3 ``*counter_collection.csv`` (one row per dispatch and counter: megabytes for a few hundred steps) to one
row per (kernel, counter): dispatches, sum, mean, min, max -- what the records under profiles/ quote.  The raw dump is scratch.

  python profiles/compact_pmc.py <counter_collection.csv> [...]     ->  <name>_per_kernel.csv next to each input
"""
import csv, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void\s+", "", name).strip()


def compact(path):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            v = float(r["Counter_Value"])
            a = agg.setdefault(k, [0, 0.0, v, v, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")])
            a[0] += 1; a[1] += v; a[2] = min(a[2], v); a[3] = max(a[3], v)
    out = re.sub(r"_?counter_collection\.csv$", "", path) + "_per_kernel.csv"
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "Mean", "Min", "Max", "Grid_Size", "Workgroup_Size", "VGPR_Count", "LDS_Block_Size"])
        for (k, c), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, a[0], f"{a[1]:.6g}", f"{a[1] / a[0]:.6g}", f"{a[2]:.6g}", f"{a[3]:.6g}", *a[4:]])
    return out


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(compact(p))
