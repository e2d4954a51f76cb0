#!/usr/bin/env python3
"""Static instruction mix of one kernel of a `hipcc -S` listing: opcode histogram, by class, optionally per `; LHW_PHASE n` region
(analysis builds with -DLHW_ASM_MARKS).  Usage: isa_mix.py listing.s <kernel-name-substring> [--phases] [--top N]"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64")): return "f64_fma"
    if op.startswith("v_mul_f64"): return "f64_mul"
    if op.startswith("v_add_f64"): return "f64_add"
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return "f64_div_sqrt"
    if op.startswith(("v_cmp", "v_cmpx")): return "cmp"
    if op.startswith("v_cndmask"): return "select"
    if "_dpp" in op: return "dpp"
    if op.startswith(("v_mov_b", "v_accvgpr")): return "mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane_xfer"
    if op.startswith("v_permlane"): return "permlane"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    phases = "--phases" in sys.argv
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and l.startswith("_Z") and name in l and l.split(":")[0].endswith(l.split(":")[0]) and ":" in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    body = lines[start + 1:end]
    ops = collections.Counter()
    cls = collections.Counter()
    per_phase = collections.defaultdict(collections.Counter)
    ph = "pre"
    for l in body:
        s = l.strip()
        m = re.match(r"; LHW_PHASE (\S+)", s)
        if m:
            ph = m.group(1)
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        op = s.split()[0]
        dpp = " row_" in s or "quad_perm" in s or "row_newbcast" in s
        c = "dpp" if dpp and op.startswith("v_") else classify(op)
        ops[op] += 1
        cls[c] += 1
        per_phase[ph][c] += 1
    n = sum(cls.values())
    print(f"{n} instructions in {lines[start].split(':')[0]}")
    for k, v in cls.most_common():
        print(f"  {k:14s} {v:7d} {100.0 * v / n:5.1f} %")
    print("top opcodes:")
    for k, v in ops.most_common(top):
        print(f"  {v:6d} {k}")
    if phases:
        keys = [k for k, _ in cls.most_common()]
        print("phase".ljust(8) + "".join(k[:9].rjust(10) for k in keys) + "     total")
        for p, c in per_phase.items():
            print(str(p).ljust(8) + "".join(str(c[k]).rjust(10) for k in keys) + str(sum(c.values())).rjust(10))


if __name__ == "__main__":
    main()
