#!/usr/bin/env python3
"""Static opcode histogram of a hand-traced hot path through a kernel's ISA listing.

    hipcc ... -S --cuda-device-only rtm_kernels.hip -o k.s
    python profiles/isa_hist.py k.s <kernel-symbol-substring> seg:a-b,c-d seg2:e-f ...

Line ranges are 1-based and relative to the kernel's label line.  Prints, per segment, the number of
instructions by class (the classes of the SQ_INSTS_VALU_* counters plus what they lump into "other")."""
import collections
import re
import sys

CLASSES = [
    ("valu_f64_arith", r"v_(add|mul|fma|fmac|min|max)_f64"),
    ("valu_f64_trans", r"v_(rcp|rsq|sqrt)_f64"),
    ("valu_f64_fixup/scale/ldexp/frexp/rndne", r"v_(div_fixup|div_scale|div_fmas|ldexp|frexp_exp_i32|frexp_mant|rndne|trig_preop)_f64|v_frexp"),
    ("valu_f64_cmp", r"v_cmp\w*_f64|v_cmp_class_f64"),
    ("valu_cvt", r"v_cvt_"),
    ("valu_cndmask", r"v_cndmask"),
    ("valu_mov", r"v_mov_b(32|64)|v_accvgpr"),
    ("valu_f32", r"v_\w+_f32"),
    ("valu_int_mul", r"v_mul_(lo|hi)_[ui]32|v_mad_u64_u32|v_mad_[ui]32"),
    ("valu_cmp_int", r"v_cmp\w*_[ui](16|32|64)"),
    ("valu_lane", r"v_readlane|v_writelane|v_readfirstlane|v_mbcnt|v_permlane|v_bpermute"),
    ("valu_int_other", r"v_"),
    ("salu", r"s_(?!waitcnt|nop|cbranch|branch|load|barrier|endpgm|sleep|setprio)"),
    ("branch", r"s_cbranch|s_branch"),
    ("waitcnt/nop", r"s_waitcnt|s_nop"),
    ("smem", r"s_load"),
    ("lds", r"ds_"),
    ("vmem", r"global_|buffer_|flat_|scratch_"),
]


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return "misc"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and re.match(r"^_Z\w+:", l))
    total = collections.Counter()
    for spec in sys.argv[3:]:
        seg, ranges = spec.split(":")
        weight = 1.0
        if "@" in seg:
            seg, w = seg.split("@")
            weight = float(w)
        c = collections.Counter()
        ops = collections.Counter()
        adj_e32_cnd = 0
        prev = ""
        for r in ranges.split(","):
            a, b = (int(v) for v in r.split("-"))
            for l in lines[start + a - 1:start + b]:
                t = l.strip()
                if not t or t.startswith((";", ".")) or t.endswith(":"):
                    continue
                op = t.split()[0]
                c[classify(op)] += 1
                ops[op] += 1
                if op == "v_cndmask_b32_e32" and prev == "v_cndmask_b32_e32":
                    adj_e32_cnd += 1
                prev = op
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print(f"== {seg} (weight {weight}): {sum(c.values())} instructions, {valu} VALU, "
              f"{adj_e32_cnd} back-to-back v_cndmask_b32_e32")
        for k, _ in CLASSES:
            if c[k]:
                print(f"   {k:45s} {c[k]:5d}")
        print("   top opcodes: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))
        for k, v in c.items():
            total[k] += v * weight
    valu = sum(v for k, v in total.items() if k.startswith("valu"))
    print(f"== weighted total per wave-iteration: {sum(total.values()):.0f} instructions, {valu:.0f} VALU")
    for k, _ in CLASSES:
        if total[k]:
            print(f"   {k:45s} {total[k]:7.1f}")


if __name__ == "__main__":
    main()
