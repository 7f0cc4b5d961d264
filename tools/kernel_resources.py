#!/usr/bin/env python3
"""Per-kernel register / spill / scratch table of every .hip translation unit in csrc/ ("check with
-Rpass-analysis=kernel-resource-usage ... and fail on VGPR spills per instantiation").

    python tools/kernel_resources.py            # table
    python tools/kernel_resources.py --check    # exit 1 when a kernel carries more scratch than its budget (BUDGET; default 0)
    python tools/kernel_resources.py --objdir build_x    # another object directory of csrc/ (variant builds)

Reads the compiler's own remarks of the REAL build (csrc/Makefile keeps them next to every object, <obj>.res: same flags as the shipped
object, per-file flags such as k_stmpc.o's -fno-slp-vectorize included); `make libf1p.so` is run first so they are current.  Nothing is
compiled here, nothing is written outside the build directory.  CPU only (hipcc cross-compiles gfx950).
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "f1tenth_planning_amd", "csrc")
# scratch bytes per lane a kernel may carry (mangled-name substring -> budget).  Everything else must have none.
#   ILb1E...            the materialising variants (all_traj requested): HBM-write-bound, not VALU-bound.  Round 6: 160 -> 56 -- the clothoid + footprint
#                       instantiation is spill-free at two waves per SIMD (was 104 B); what is left is cubic + footprint's 52 B (12 VGPRs; two waves measured 4 % slower)
#   k_kmpc_*            spills sit in the once-per-workgroup setup blocks and the rarely taken serial fp64 fallback, not in the
#                       filter loop (tools/isa_loops.py); a lower register cap was measured slower (LABNOTES.md 5b)
#   k_lattice / g1      the out-of-line fp64 fit's 8-byte frame
#   k_lattice_filter3<CR, true, ..>   the instantiations with the test hooks compiled in.  They ARE what launch_lattice_mixed launches for host-supplied
#                       goals (<CR, true, true>), cubic + footprint and footprint + host goals -- production plan shapes of the add_sample_function
#                       path -- so those shapes carry this budget (8-24 B of scratch); the spill-free claim holds for the device-goal shapes
#                       (<CR, false, false, GEN, FOOT = false>: 0 scratch, asserted below by the default budget of 0)
BUDGET = {"ILb1E": 56, "k_kmpc_plan_gen": 32, "k_kmpc_shoot_mixed": 0, "k_clothoid_g1": 8, "9k_latticeILb0E": 8,
          "k_lattice_filter3ILi1ELb1E": 24, "k_lattice_filter3ILi2ELb1E": 24,
          "k_lattice_filter3ILi1ELb0ELb0ELi0ELb1E": 8, "k_lattice_filter3ILi2ELb0ELb0ELi0ELb1E": 8}   # (the instantiations WITH test hooks -- incl. the host-goal shapes; device goals, point footprint: 0)
# The headline kernels (k_lattice_prologue, k_lattice_filter3<CR, false, false, GEN, false>) carry NO scratch and no VGPR spills;
# `make resources` is part of __graft_entry__.build().


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"\(.*$", "", o).replace("f1p::", "") for o in out[:len(names)]]
    except OSError:
        return names


def resources(res_file):
    rows, cur = [], None
    for ln in open(res_file, errors="replace").read().splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]):\s*(\S+)", ln)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
    return rows


def main():
    check = "--check" in sys.argv
    quiet = "--quiet" in sys.argv
    bad = []
    out = sys.stdout
    if quiet:
        sys.stdout = open(os.devnull, "w")
    print(f"{'kernel':<58} {'SGPR':>5} {'VGPR':>5} {'AGPR':>5} {'sgpr-spill':>10} {'vgpr-spill':>10} {'scratch':>8} {'occ':>4} {'LDS':>7}")
    objdir = sys.argv[sys.argv.index("--objdir") + 1] if "--objdir" in sys.argv else "build"
    if objdir == "build":                                   # the default build: bring it (and its remarks) up to date first
        r = subprocess.run(["make", "-C", CSRC, "-j4", "libf1p.so"], capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout + r.stderr)
            raise SystemExit(r.returncode)
    res = sorted(glob.glob(os.path.join(CSRC, objdir, "k_*.o.res"))) + [os.path.join(CSRC, objdir, "f1p_api.o.res")]
    srcs = sorted(glob.glob(os.path.join(CSRC, "k_*.hip")))
    if len(res) != len(srcs) + 1 or not all(os.path.exists(p) for p in res):
        raise SystemExit(f"kernel_resources: {objdir}/ holds remarks for {len(res) - 1} of {len(srcs)} kernel files -- rebuild (make clean; make)")
    for rf in res:
        rows = resources(rf)
        names = demangle([r["name"] for r in rows])
        for r, nm in zip(rows, names):
            print(f"{nm[:58]:<58} {r.get('TotalSGPRs', '?'):>5} {r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('SGPRs Spill', '?'):>10} "
                  f"{r.get('VGPRs Spill', '?'):>10} {r.get('ScratchSize', '?'):>8} {r.get('Occupancy', '?'):>4} {r.get('LDS Size', '?'):>7}")
            budget = max([v for k, v in BUDGET.items() if k in r["name"]] + [0])
            if int(r.get("ScratchSize", 0)) > budget:
                bad.append(f"{nm} ({r.get('ScratchSize')} B > {budget})")
    sys.stdout = out
    if check and bad:
        print("scratch over budget:", ", ".join(bad))
        raise SystemExit(1)


if __name__ == "__main__":
    main()
