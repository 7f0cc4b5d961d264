"""The C-ABI library loads without a GPU and exports every symbol include/f1p.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from f1tenth_planning_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "f1p.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(f1p_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_prototypes_agree():
    syms = _declared_symbols()
    assert len(syms) >= 30
    assert sorted(_abi.PROTOTYPES) == syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _abi.load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.f1p_version().decode() == "0.1.0"


def test_struct_layout_matches_defaults():
    lib = _abi.load_library()
    cfg = _abi.LatticeCfg()
    lib.f1p_lattice_cfg_default(C.byref(cfg))
    ref = _abi.lattice_cfg()
    assert (cfg.n_stations, cfg.n_lookahead, cfg.n_width, cfg.n_shift, cfg.n_cull) == (100, 4, 7, 1, 1)
    assert list(cfg.lookahead[:4]) == [0.4, 0.6, 0.8, 1.0] == list(ref.lookahead[:4])
    assert abs(cfg.width[0] + 1.0) < 1e-15 and cfg.width[6] == 1.0 and abs(cfg.width[3]) < 1e-15
    assert (cfg.track_lookahead, cfg.wheelbase, cfg.max_reacquire, cfg.w_length) == (0.8, 0.33, 20.0, 1.0)
    assert C.sizeof(cfg) == 10 * 4 + 8 * 128 + 8 * 7 and cfg.generator == 0
    k = _abi.KmpcCfg()
    lib.f1p_kmpc_cfg_default(C.byref(k))
    d = _abi.kmpc_cfg()
    for f, _ in _abi.KmpcCfg._fields_:
        a, b = getattr(k, f), getattr(d, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), f


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_abi.F1PLibraryError):
        _abi.load_library(str(tmp_path / "nope.so"))


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_no_gpu_means_no_context():
    from f1tenth_planning_amd.runtime import Context, F1PError
    with pytest.raises(F1PError):
        Context(0)


def test_struct_mirrors_match_the_compiler(tmp_path):
    """the product's ctypes structs (_abi.py) and the oracle's own mirrors (oracle/structs.py, generated from
    include/f1p.h) both against what gcc lays out for the header: size and every field's offset.  The oracle copies incoming
    configurations into ITS mirrors field by field, so a mistake in _abi.py cannot be shared by the checker."""
    import subprocess
    from oracle import structs
    pairs = (("f1p_lattice_cfg", _abi.LatticeCfg, structs.LatticeCfg), ("f1p_kmpc_cfg", _abi.KmpcCfg, structs.KmpcCfg),
             ("f1p_stmpc_cfg", _abi.StmpcCfg, structs.StmpcCfg), ("f1p_kmpc_sampler", _abi.KmpcSampler, structs.KmpcSampler))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "f1p.h"', 'int main(void) {']
    for name, _, mine in pairs:
        lines.append(f'printf("{name} %zu", sizeof({name}));')
        for f, _ct in mine._fields_:
            lines.append(f'printf(" {f}:%zu", offsetof({name}, {f}));')
        lines.append('printf("\\n");')
    lines += ['return 0; }']
    src = tmp_path / "probe.c"; src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().split("\n")
    for (name, prod, mine), line in zip(pairs, out):
        toks = line.split()
        assert toks[0] == name
        size = int(toks[1]); offs = dict((t.split(":")[0], int(t.split(":")[1])) for t in toks[2:])
        for cls in (prod, mine):
            assert C.sizeof(cls) == size, (name, cls.__name__)
            assert [f[0] for f in cls._fields_] == list(offs), (name, cls.__name__)
            for f, _ct in cls._fields_:
                assert getattr(cls, f).offset == offs[f], (name, cls.__name__, f)
    # the by-name copy: a product-side struct reaches the oracle's layout unchanged
    cfg = _abi.lattice_cfg(lookaheads=[0.7, 1.9], widths=[-0.5, 0.0, 0.5], n_stations=33, weights=(0.1, 0.2, 0.3, 0.4), n_shift=2, n_cull=1)
    own = structs.mirror(cfg, structs.LatticeCfg)
    assert bytes(own) == bytes(cfg) and own.n_cand == 6
