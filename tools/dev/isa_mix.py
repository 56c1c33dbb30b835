#!/usr/bin/env python3
"""Developer aid: static instruction mix of a kernel's hot path from hipcc's -save-temps assembly.
  python tools/dev/isa_mix.py <file.s> <kernel name fragment> [<last line of the hot path>]
Counts MFMA / VALU / SALU / LDS / VMEM instructions from the kernel's label to the given line (default: its s_endpgm) and
lists the most frequent VALU opcodes -- the figure VERDICT r3 item 1 judges (VALU : MFMA) in its static form."""
import sys


def main():
    path, frag = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if frag in l and not l.startswith(("\t", ".")) and ":" in l.split(";")[0])
    end = int(sys.argv[3]) if len(sys.argv) > 3 else next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    c, ops = {}, {}
    for l in lines[start:end]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";", "//")) or t[0].endswith(":"):
            continue
        op = t[0]
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
             "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
        c[k] = c.get(k, 0) + 1
        if k == "valu":
            ops[op] = ops.get(op, 0) + 1
    print(f"lines {start}..{end}: {c}  VALU : MFMA = {c.get('valu', 0) / max(c.get('mfma', 1), 1):.2f}")
    # SGPR spills go to VGPRs the compiler names in a comment ("; implicit-def: $vgpr246 : SGPR spill to VGPR lane"); the other
    # v_writelane instructions are the source's own (SET_LANE: a ballot's halves into lane i of the saved ReLU mask words)
    import re
    spill_regs = {m.group(1) for l in lines[start:end] for m in [re.search(r"implicit-def: \$vgpr(\d+) : SGPR spill", l)] if m}
    lane_ops = [l.split(";")[0] for l in lines[start:end] if l.strip().startswith(("v_readlane", "v_writelane"))]
    spills = sum(1 for l in lane_ops if any(re.search(rf"\bv{r}\b", l) for r in spill_regs))
    print(f"lane spills (v_readlane / v_writelane on the SGPR-spill VGPRs {sorted(spill_regs)}): {spills}   other lane ops (SET_LANE etc.): {len(lane_ops) - spills}",
          "  scratch ops:", sum(1 for l in lines[start:end] if "scratch_" in l))
    print(sorted(ops.items(), key=lambda kv: -kv[1])[:30])


if __name__ == "__main__":
    main()
