"""Loaders for the on-disk formats of the reference's assets, on files written in those formats."""
import numpy as np
import pytest

from f1tenth_planning_amd import io as fio


def test_raceline_formats(tmp_path, tracks):
    spl, lev = tracks["spielberg"], tracks["levine"]
    p1 = tmp_path / "raceline.csv"      # examples/control/Spielberg_raceline.csv:1 layout
    with open(p1, "w") as fh:
        fh.write("#x_m        ; y_m         ; vx_mps      ; psi_rad     ; kappa_radpm\n")
        for r in spl[:50]:
            fh.write(" ; ".join(f"{v:.7f}" for v in r) + "\n")
    a = fio.load_raceline(str(p1))
    np.testing.assert_allclose(a, spl[:50], atol=5e-8)
    assert fio.raceline_columns(a) == (0, 1, 2, 3, 4)
    np.testing.assert_array_equal(a, np.loadtxt(p1, delimiter=";", skiprows=0))       # what the example script does
    p2 = tmp_path / "centerline.csv"    # examples/control/levine_centerline.csv:1-3 layout
    with open(p2, "w") as fh:
        fh.write("# f5f6776c;;;;;;\n# 7834a7db;;;;;;\n# s_m; x_m; y_m; psi_rad; kappa_radpm; vx_mps; ax_mps2\n")
        for r in lev[:40]:
            fh.write(";".join(repr(float(v)) for v in r) + "\n")
    b = fio.load_raceline(str(p2))
    np.testing.assert_array_equal(b, lev[:40])
    np.testing.assert_array_equal(b, np.loadtxt(p2, delimiter=";", skiprows=3))
    assert fio.raceline_columns(b) == (1, 2, 5, 3, 4)
    p3 = tmp_path / "bad.csv"
    p3.write_text("1;2\n3;4\n")
    with pytest.raises(ValueError):
        fio.load_raceline(str(p3))


def test_map_yaml_pgm_png(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.choice(np.array([0, 205, 254], np.uint8), size=(41, 61))                 # trinary like levine_slam.pgm
    with open(tmp_path / "m.pgm", "wb") as fh:
        fh.write(b"P5\n# CREATOR: map_saver\n61 41\n255\n" + img.tobytes())
    (tmp_path / "m.yaml").write_text("image: m.pgm\nmode: trinary\nresolution: 0.05\norigin: [-25, -6.19, 0]\nnegate: 0\n"
                                     "occupied_thresh: 0.65\nfree_thresh: 0.25\n")
    m = fio.load_map(str(tmp_path / "m.yaml"))
    np.testing.assert_array_equal(m["image"], img)
    assert m["resolution"] == 0.05 and m["origin"] == (-25.0, -6.19, 0.0)
    assert m["occupied_below"] == 90                                                  # v < 255 * 0.35 = 89.25
    assert ((m["image"] < m["occupied_below"]) == (img == 0)).all()
    from PIL import Image
    Image.fromarray(img).save(tmp_path / "n.png")
    (tmp_path / "n.yaml").write_text("image: n.png\nresolution: 0.05796\norigin: [-84.85359914210505,-36.30299725862132, 0.000000]\n"
                                     "negate: 1\noccupied_thresh: 0.45\nfree_thresh: 0.196\n")
    n = fio.load_map(str(tmp_path / "n.yaml"))
    np.testing.assert_array_equal(n["image"], 255 - img)
    assert n["occupied_below"] == 141 and n["negate"] == 1                            # 255 * 0.55 = 140.25
    with open(tmp_path / "a.pgm", "w") as fh:                                         # ASCII PGM
        fh.write("P2\n3 2\n255\n0 205 254\n254 0 205\n")
    (tmp_path / "a.yaml").write_text("image: a.pgm\nresolution: 1.0\norigin: [0, 0, 0]\n")
    a = fio.load_map(str(tmp_path / "a.yaml"))
    np.testing.assert_array_equal(a["image"], np.array([[0, 205, 254], [254, 0, 205]], np.uint8))
