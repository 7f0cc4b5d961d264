"""The record bench.py prints (round 5's 27 KB line was not parsed by the driver).

No GPU: the formatter is fed stub records -- the committed full record of an earlier run, and a synthetic worst case with
non-finite floats, numpy scalars and oversize strings -- and the LAST stdout line must be strict JSON, below the size cap,
with the contract's keys, `roofline` and `cpu_baseline`; the full record goes to the side file.
"""
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _strict_loads(line):
    def no_constants(tok):
        raise ValueError("non-strict JSON token " + tok)
    return json.loads(line, parse_constant=no_constants)


def _check(line):
    assert "\n" not in line and len(line) <= bench.COMPACT_MAX_BYTES < 6000
    rec = _strict_loads(line)
    for k in CONTRACT:
        assert k in rec, k
    assert isinstance(rec["config"]["workload"], str) and "model" not in rec["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    return rec


@pytest.mark.parametrize("name", ["r05_bench_full.json", "r04_bench_full.json"])
def test_compact_line_of_a_committed_full_record(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip(name + " not in profiles/")
    full = json.load(open(path))
    assert len(json.dumps(full)) > 8188                  # the record that did not fit the driver's tail
    rec, line = bench.compact_record(full)
    rec2 = _check(line)
    assert rec == rec2
    assert rec["value"] == pytest.approx(full["value"], rel=1e-5) and rec["steps"] == full["steps"] and rec["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert rec["roofline"]["kernel"] == "k_lattice_filter3"
    assert rec["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"] and rec["parity"]["best_idx_mismatches"] == full["parity"]["best_idx_mismatches"]


def _worst_case():
    big = "x" * 5000
    full = {"metric": "candidate-trajectory-steps/sec per GPU; p50 plan() latency @4096 egos", "value": np.float64(7.1e11), "unit": "candidate-trajectory-steps/s",
            "n_gpus": 8, "steps": np.int64(20), "warmup": 5, "ms_per_step": 0.07, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 filter + f64 decision", "data": "synthetic",
            "config": {"workload": big, "egos_per_gpu": 4096, "candidates": 256, "stations": 50, "generator": "clothoid", "state": big, "grid": [2000, 2000], "nested": {"a": big}},
            "roofline": {"bound": "valu", "achieved": float("nan"), "peak": 78.6, "unit": "T lane-instr/s", "frac": float("inf"), "traffic": None, "kernel": big,
                         "kernel_ms": 0.07, "dominant_kernel_ms": 0.03, "hbm": {"note": big}, "valu": {"note": big}},
            "cpu_baseline": {"value": 3e7, "unit": "candidate-trajectory-steps/s", "cores": 256, "kind": "port", "sample": big, "note": big},
            "cpu_baseline_numpy": {"value": 3e6, "cores": 1, "note": big},
            "parity": {"egos_checked": 4096, "best_idx_mismatches": 0, "max_abs_dsteer": float("-inf"), "checked_outputs": big},
            "per_rank_ms_per_step": {"min": 0.07, "max": 0.08, "ranks": 8, "all": [0.07] * 8, "note": big},
            "scene_sweep": {k: {"note": big} for k in "abcde"}, "variants": {"note": big}, "kmpc_c4": {"note": big}}
    for k in bench.COMPACT_SCALARS:
        full[k] = 0.123456789012345
    full["other_schedules_bit_identical"] = np.bool_(True)
    return full


def test_compact_line_of_a_worst_case_record():
    rec = _check(bench.compact_record(_worst_case())[1])
    assert rec["roofline"]["achieved"] is None and rec["roofline"]["frac"] is None and rec["parity"]["max_abs_dsteer"] is None
    assert rec["value"] == 7.1e11 and rec["steps"] == 20 and rec["other_schedules_bit_identical"] is True
    assert rec["per_rank_ms_per_step"] == {"min": 0.07, "max": 0.08, "ranks": 8}
    assert len(rec["config"]["workload"]) <= 200 and len(rec["cpu_baseline"]["sample"]) <= 200


def test_scalars_are_dropped_before_the_line_outgrows_the_cap():
    full = _worst_case()
    rec, line = bench.compact_record(full, max_bytes=2048)
    assert len(line) <= 2048
    _strict_loads(line)
    for k in CONTRACT:
        assert k in rec
    kept = [k for k in bench.COMPACT_SCALARS if k in rec]
    assert kept == list(bench.COMPACT_SCALARS[:len(kept)])         # the tail of the list goes first


def test_emit_writes_the_full_record_and_prints_the_compact_line_last(tmp_path, monkeypatch, capsys):
    path = os.path.join(tmp_path, "full.json")
    monkeypatch.setattr(bench, "FULL_RECORD_PATH", path)
    full = _worst_case()
    bench.emit(full)
    cap = capsys.readouterr()
    out_lines = cap.out.splitlines()
    assert len(out_lines) == 1                                   # ONE stdout line
    rec = _check(out_lines[-1])
    assert rec["full_record"] == "full.json" and "full.json" in cap.err
    side = _strict_loads(open(path).read())                      # strict JSON too: nan / inf became null
    assert side["roofline"]["achieved"] is None and side["scene_sweep"]["a"]["note"] == "x" * 5000 and side["steps"] == 20
