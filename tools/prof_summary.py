#!/usr/bin/env python3
"""Condense a rocprofv3 output directory (csv format) into the small markdown summary that is committed under profiles/.

    rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_X -o run -- python3 bench.py ...
    rocprofv3 --pmc FETCH_SIZE -f csv -d gpurun_out/pmc_fetch_X -o run -- python3 bench.py ...
    python tools/prof_summary.py gpurun_out/prof_X [gpurun_out/pmc_fetch_X ...] > profiles/rNN_name.md
    python tools/prof_summary.py --json profiles/rNN_name_pmc.json --headline 'k_lattice<false, 0, false>' \
           --config '{"egos": 4096, ...}' gpurun_out/prof_X gpurun_out/pmc_* > profiles/rNN_name.md

--json writes the machine-readable file bench.py parses at run time (roofline.traffic, roofline.valu_fp64): per kernel the
mean counter values per dispatch, the average duration, and `headline: true` on the kernel the timed step launches.

Kernel-trace directories give per-kernel call count / average / min / max duration; PMC directories give per-kernel
mean counter values per dispatch.  FETCH_SIZE / WRITE_SIZE are reported in the counter's own unit (KiB per the
rocprofv3 definition) next to the corrected byte figure that MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE x 2 for wide coalesced reads).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def _find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True))


def kernel_trace(d):
    rows = []
    for f in _find(d, "kernel_trace.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    if not rows:
        return None
    agg = defaultdict(list)
    meta = {}
    for r in rows:
        name = r.get("Kernel_Name", "?")
        agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta[name] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?"),
                      r.get("LDS_Block_Size", "?"), r.get("Scratch_Size", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                      r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    return agg, meta


def pmc(d):
    rows = []
    for f in _find(d, "counter_collection.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    agg = defaultdict(lambda: defaultdict(list))
    for r in rows:
        agg[r.get("Kernel_Name", "?")][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", "nan")))
    return agg


def write_json(path, dirs, headline, config):
    import json
    kernels = {}
    for d in dirs:
        kt = kernel_trace(d)
        if kt and not pmc(d):                      # durations from the plain trace only (PMC passes perturb them)
            for name, v in kt[0].items():
                kernels.setdefault(name, {})["avg_us"] = sum(v) / len(v)
                kernels[name]["calls"] = len(v)
        for name, cs in pmc(d).items():
            for cname, v in cs.items():
                key = {"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB", "SQ_WAVES": "waves"}.get(cname, cname)
                kernels.setdefault(name, {})[key] = sum(v) / len(v)
    out = {"config": config, "dirs": dirs,
           "kernels": [dict(kernel=n, headline=bool(headline and headline in n), **v) for n, v in sorted(kernels.items())]}
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)


def main():
    import json
    dirs = sys.argv[1:]
    jpath = headline = None
    config = None
    while dirs and dirs[0].startswith("--"):
        opt = dirs.pop(0)
        if opt == "--json":
            jpath = dirs.pop(0)
        elif opt == "--headline":
            headline = dirs.pop(0)
        elif opt == "--config":
            config = json.loads(dirs.pop(0))
        else:
            raise SystemExit(f"unknown option {opt}")
    if not dirs:
        raise SystemExit(__doc__)
    if jpath:
        write_json(jpath, dirs, headline, config)
    print("# rocprofv3 summary\n")
    for d in dirs:
        kt = kernel_trace(d)
        if kt:
            agg, meta = kt
            total = sum(sum(v) for v in agg.values())
            print(f"## kernel trace: `{d}`\n")
            print("| kernel | calls | avg us | min us | max us | total us | % | VGPR | AGPR | SGPR | LDS B | scratch | grid | wg |")
            print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
            for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                m = meta[name]
                print(f"| `{name[:90]}` | {len(v)} | {sum(v)/len(v):.2f} | {min(v):.2f} | {max(v):.2f} | {sum(v):.1f} | "
                      f"{100*sum(v)/total:.2f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} | {m[5]} | {m[6]} |")
            print()
        pc = pmc(d)
        if pc:
            print(f"## PMC counters: `{d}` (mean per dispatch)\n")
            print("| kernel | counter | dispatches | mean | note |")
            print("|---|---|---|---|---|")
            for name, cs in sorted(pc.items()):
                for cname, v in sorted(cs.items()):
                    mean = sum(v) / len(v)
                    note = ""
                    if cname == "FETCH_SIZE":
                        note = f"KiB; x1024 = {mean*1024:.0f} B; gfx950 wide-read correction x2 = {mean*2048:.0f} B"
                    elif cname == "WRITE_SIZE":
                        note = f"KiB; x1024 = {mean*1024:.0f} B (uncalibrated on gfx950)"
                    print(f"| `{name[:70]}` | {cname} | {len(v)} | {mean:.4g} | {note} |")
            print()


if __name__ == "__main__":
    main()
