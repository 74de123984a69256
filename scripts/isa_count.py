#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing: per basic block, how many VALU / SALU / VMEM / LDS
instructions, and which VALU mnemonics dominate.  Used to see what the solver's iteration loop is made of without a GPU.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o solver.s csrc/psfm_solver.hip
    python scripts/isa_count.py solver.s _Z20psfm_pc_fused_kernelILi3EEv8PcParams [min_block_size]
"""
import collections
import re
import sys


def main():
    path, fn = sys.argv[1], sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(fn + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        cur.append(t.split()[0])
    blocks.append((name, cur))
    tot = collections.Counter()
    for name, ins in blocks:
        c = collections.Counter()
        for op in ins:
            k = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else
                 "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
            c[k] += 1
            tot[k] += 1
        if len(ins) >= minsz:
            top = collections.Counter(op for op in ins if op.startswith("v_")).most_common(14)
            print("%-12s n=%4d  %s" % (name, len(ins), dict(c)))
            print("             " + ", ".join("%s:%d" % kv for kv in top))
    print("TOTAL", dict(tot))


if __name__ == "__main__":
    main()
