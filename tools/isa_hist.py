#!/usr/bin/env python3
"""Static instruction-class histogram of one kernel's gfx950 listing, per basic block and in total (the classes the issue-cycle
table of profiles/r03_valu_issue_cycles.txt prices differently).

    hipcc --offload-arch=gfx950 <flags of the Makefile> -S --cuda-device-only k_lattice_filter3.hip -o /tmp/klm.s        (or k_lattice_prologue / _refine / _select .hip)
    python tools/isa_hist.py /tmp/klm.s k_lattice_filter3ILi2ELb0ELb0 [--min 30]

classes: fast (v_add/sub/mul/fmac/fma_f32, and/or/xor, lshrrev, add_u32, mov -- 2.5 cycles), slow (min/max, cvt, cmp, cndmask, bfe, med3,
lshlrev, add3 / lshl_add / mad, DPP forms, readlane / writelane, fp64, packed f32 -- 4.3), trans (sin / cos / rcp / rsq / sqrt / exp / log --
8.3), lds (ds_*), vmem (global_ / buffer_ / flat_ / scratch_), salu, branch, wait (s_waitcnt / s_nop / s_barrier)."""
import re
import sys

FAST = re.compile(r"^v_(add|sub|subrev|mul|fmac|fma|fmaak|fmamk)_f32|^v_(and|or|xor)_b32|^v_lshrrev_b32|^v_(add|sub|subrev)_u32|^v_mov_b32|^v_(mul|fma|add)_legacy")
TRANS = re.compile(r"^v_(sin|cos|rcp|rsq|sqrt|exp|log)_f(32|16)|^v_rcp_iflag")


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if "_dpp" in ins or " row_" in ins or "quad_perm" in ins or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "slow:dpp/lane"
        if TRANS.match(op):
            return "trans"
        if "_f64" in op:
            return "slow:f64"
        if op.startswith("v_pk_"):
            return "slow:packed"
        if FAST.match(op):
            # an SGPR operand moves a fast instruction to the slow class
            ops = ins.split(None, 1)[1] if " " in ins else ""
            if re.search(r"(^|[ ,\[])s\d+|s\[\d+:\d+\]|vcc|exec", ops):
                return "slow:sgpr-operand"
            return "fast"
        if op.startswith(("v_cmp", "v_cmpx")):
            return "slow:cmp"
        if op.startswith("v_cndmask"):
            return "slow:cndmask"
        if op.startswith(("v_min", "v_max", "v_med3")):
            return "slow:minmax"
        if op.startswith("v_cvt"):
            return "slow:cvt"
        return "slow:other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 30
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(rf"^_Z\w*{kern}\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ("entry", [])
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = (m.group(1), [])
            continue
        t = l.strip()
        if t and not t.startswith((";", ".")):
            cur[1].append(t)
    blocks.append(cur)
    order = {b[0]: i for i, b in enumerate(blocks)}
    classes = ["fast", "slow:sgpr-operand", "slow:cmp", "slow:cndmask", "slow:minmax", "slow:cvt", "slow:dpp/lane", "slow:f64", "slow:packed", "slow:other", "trans",
               "lds", "vmem", "salu", "branch", "wait"]
    tot = {c: 0 for c in classes}
    print(f"{'block':<14}{'n':>5} " + " ".join(f"{c.replace('slow:', 's:')[:9]:>9}" for c in classes) + "  loop")
    for i, (label, ins) in enumerate(blocks):
        h = {c: 0 for c in classes}
        for x in ins:
            c = classify(x)
            h[c if c in h else "slow:other"] += 1
        for c in classes:
            tot[c] += h[c]
        back = [m.group(1) for x in ins for m in [re.search(r"(\.LBB\d+_\d+)", x)] if m and x.startswith(("s_cbranch", "s_branch")) and order.get(m.group(1), 1 << 30) <= i]
        if len(ins) >= mn or back:
            print(f"{label:<14}{len(ins):>5} " + " ".join(f"{h[c]:>9}" for c in classes) + ("  <- " + " ".join(sorted(set(back))) if back else ""))
    n = sum(tot.values())
    print(f"{'total':<14}{n:>5} " + " ".join(f"{tot[c]:>9}" for c in classes))
    valu = sum(v for c, v in tot.items() if c == "fast" or c.startswith("slow") or c == "trans")
    cyc = tot["fast"] * 2.5 + tot["trans"] * 8.3 + sum(v for c, v in tot.items() if c.startswith("slow")) * 4.3
    print(f"static VALU {valu}: fast {tot['fast']} ({tot['fast'] / valu:.2f}), slow {sum(v for c, v in tot.items() if c.startswith('slow'))}, trans {tot['trans']}; "
          f"issue cycles at the measured table: {cyc:.0f} = {cyc / valu:.2f} per instruction (static: every block once)")


if __name__ == "__main__":
    main()
