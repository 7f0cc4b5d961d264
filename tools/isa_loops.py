#!/usr/bin/env python3
"""Static look at one kernel's gfx950 listing (hipcc -S --cuda-device-only): per basic block the VALU / SALU / readlane+writelane
(SGPR spill traffic) / scratch counts, with the loop back-edges marked -- to see whether spills sit inside hot loops.

    python tools/isa_loops.py /tmp/k_lattice.s k_lattice_filter3 [--min 20]
"""
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 15
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(rf"^_Z\w*{kern}\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # extend to the real end (several s_endpgm possible): stop at .Lfunc_end
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "ins": []}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = {"label": m.group(1), "ins": []}
            continue
        t = l.strip()
        if t and not t.startswith((";", ".")):
            cur["ins"].append(t)
    blocks.append(cur)
    order = {b["label"]: i for i, b in enumerate(blocks)}
    tot = dict(valu=0, salu=0, lane=0, scratch=0)
    print(f"{'block':<14}{'n':>5}{'valu':>6}{'salu':>6}{'lane':>6}{'scr':>5}  back-edge")
    for i, b in enumerate(blocks):
        v = sum(1 for x in b["ins"] if x.startswith("v_"))
        s = sum(1 for x in b["ins"] if x.startswith("s_") and not x.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier")))
        ln = sum(1 for x in b["ins"] if x.startswith(("v_readlane", "v_writelane")))
        sc = sum(1 for x in b["ins"] if x.startswith("scratch_"))
        tot["valu"] += v; tot["salu"] += s; tot["lane"] += ln; tot["scratch"] += sc
        back = [m.group(1) for x in b["ins"] for m in [re.search(r"(\.LBB\d+_\d+)", x)] if m and x.startswith(("s_cbranch", "s_branch")) and order.get(m.group(1), 1 << 30) <= i]
        if len(b["ins"]) >= mn or back or ln:
            print(f"{b['label']:<14}{len(b['ins']):>5}{v:>6}{s:>6}{ln:>6}{sc:>5}  {' '.join(back)}")
    print("total", tot, "blocks", len(blocks))


if __name__ == "__main__":
    main()
