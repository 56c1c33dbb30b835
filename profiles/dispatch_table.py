#!/usr/bin/env python3
"""One training step's dispatch table from rocprofv3 --kernel-trace CSVs of bench.py:

  python profiles/dispatch_table.py out.csv LABEL=trace.csv [LABEL=trace.csv ...]

A step = the launches from one shadow_kernel (or, when the optimizer left the weight shadows ready, one front_kernel) up to the next.  Durations are
averaged by position over every complete step of the trace that has the most common launch sequence (warm-up, the batch
sweep and the forward-only leg of bench.py have other sequences and drop out); gap = start minus the previous launch's end.
"""
import csv, re, sys
from collections import Counter, defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void\s+", "", name)
    return re.split(r"\(", name, 1)[0].strip()


def table(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    steps, cur, prev = [], None, ""
    for r in rows:
        name = short(r["Kernel_Name"])
        # a step starts with the shadow launch, or -- when the optimizer left the shadows ready -- with the front kernel
        if name == "shadow_kernel" or (name.startswith(("front_kernel", "front8_kernel")) and prev != "shadow_kernel"):
            if cur: steps.append(cur)
            cur = []
        if cur is not None: cur.append(r)
        prev = name
    seqs = Counter(tuple(short(r["Kernel_Name"]) for r in s) for s in steps)
    seq = max(seqs, key=lambda k: (seqs[k] * ("sumsq_kernel" in k), len(k)))        # the training step's sequence
    sel = [s for s in steps if tuple(short(r["Kernel_Name"]) for r in s) == seq]
    out = []
    for i, name in enumerate(seq):
        d = [int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"]) for s in sel]
        g = [int(s[i]["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"]) for s in sel] if i else [0]
        r = sel[0][i]
        out.append(dict(position=i, kernel=name, grid=r.get("Grid_Size_X", r.get("Grid_Size", "")), workgroup=r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                        lds_bytes=r.get("LDS_Block_Size", ""), vgprs=r.get("VGPR_Count", ""), avg_us=round(sum(d) / len(d) / 1e3, 2),
                        min_us=round(min(d) / 1e3, 2), gap_before_us=round(sum(g) / len(g) / 1e3, 2), steps_averaged=len(sel)))
    return out


def main():
    dst, rows = sys.argv[1], []
    for spec in sys.argv[2:]:
        label, path = spec.split("=", 1)
        t = table(path)
        tot = sum(r["avg_us"] for r in t); gaps = sum(r["gap_before_us"] for r in t)
        for r in t: rows.append(dict(config=label, **r))
        rows.append(dict(config=label, position="", kernel=f"TOTAL ({len(t)} launches)", grid="", workgroup="", lds_bytes="", vgprs="",
                         avg_us=round(tot, 2), min_us="", gap_before_us=round(gaps, 2), steps_averaged=t[0]["steps_averaged"]))
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    for r in rows: print(r)


if __name__ == "__main__":
    main()
