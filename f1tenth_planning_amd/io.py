"""On-disk formats either side of the path: raceline CSVs and ROS map_server maps.

* racelines: ';'-delimited text with '#' comment lines -- `examples/control/Spielberg_raceline.csv:1` (5 columns
  x; y; vx; psi; kappa, one '#' header line) and `examples/control/levine_centerline.csv:1-3` (7 columns
  s; x; y; psi; kappa; vx; ax after three '#' lines).  The examples read them with np.loadtxt(delimiter=';', skiprows=k).
* maps: a YAML next to a PNG / PGM (`examples/control/Spielberg_map.yaml:1-6`, `levine_slam.yaml:1-7`) with the
  map_server keys image, resolution, origin, negate, occupied_thresh, free_thresh.
"""
import os

import numpy as np


def load_raceline(path, delimiter=";"):
    """Rows of floats; '#' lines are skipped wherever they are (np.loadtxt's comment handling), so both reference files load
    without a skiprows count."""
    arr = np.loadtxt(path, delimiter=delimiter, comments="#", ndmin=2)
    if arr.shape[1] < 3:
        raise ValueError('Waypoints needs to be a (Nxm), m >= 3, numpy array!')
    return np.ascontiguousarray(arr, dtype=np.float64)


def raceline_columns(arr):
    """(x, y, v, psi, kappa) column indices for the two layouts the reference ships: 5 columns [x, y, v, psi, kappa] or
    7 columns [s, x, y, psi, kappa, vx, ax]."""
    if arr.shape[1] >= 7:
        return (1, 2, 5, 3, 4)
    if arr.shape[1] >= 5:
        return (0, 1, 2, 3, 4)
    return (0, 1, 2, 3 if arr.shape[1] >= 4 else -1, -1)


def _read_pgm(path):
    """Binary (P5) or ASCII (P2) PGM, 8-bit."""
    with open(path, "rb") as fh:
        data = fh.read()
    tokens, pos = [], 0
    while len(tokens) < 4:
        while pos < len(data) and data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while pos < len(data) and data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        start = pos
        while pos < len(data) and not data[pos:pos + 1].isspace():
            pos += 1
        tokens.append(data[start:pos])
    magic, w, h, maxval = tokens[0], int(tokens[1]), int(tokens[2]), int(tokens[3])
    if maxval > 255:
        raise ValueError("only 8-bit PGM maps are supported")
    if magic == b"P5":
        img = np.frombuffer(data, dtype=np.uint8, count=w * h, offset=pos + 1).reshape(h, w)
    elif magic == b"P2":
        img = np.array(data[pos:].split(), dtype=np.int64).astype(np.uint8).reshape(h, w)
    else:
        raise ValueError(f"not a PGM file: {path}")
    return np.ascontiguousarray(img)


def load_map(yaml_path):
    """ROS map_server map -> dict(image u8 [h, w] (row 0 = top), resolution, origin (x, y, yaw), occupied_below, negate,
    occupied_thresh, free_thresh).  `occupied_below` is the u8 threshold libf1p uses: a cell is occupied iff its (negate-
    corrected) value v satisfies (255 - v)/255 > occupied_thresh, i.e. v < 255 (1 - occupied_thresh)."""
    import yaml
    with open(yaml_path) as fh:
        meta = yaml.safe_load(fh)
    img_path = meta["image"]
    if not os.path.isabs(img_path):
        img_path = os.path.join(os.path.dirname(os.path.abspath(yaml_path)), img_path)
    if img_path.lower().endswith(".pgm"):
        img = _read_pgm(img_path)
    else:
        from PIL import Image
        with Image.open(img_path) as im:
            img = np.ascontiguousarray(np.asarray(im.convert("L"), dtype=np.uint8))
    negate = int(meta.get("negate", 0))
    if negate:
        img = np.ascontiguousarray(255 - img)
    occ = float(meta.get("occupied_thresh", 0.65))
    origin = [float(v) for v in meta.get("origin", [0.0, 0.0, 0.0])]
    while len(origin) < 3:
        origin.append(0.0)
    return dict(image=img, resolution=float(meta["resolution"]), origin=tuple(origin),
                occupied_below=int(np.ceil(255.0 * (1.0 - occ))), negate=negate, occupied_thresh=occ,
                free_thresh=float(meta.get("free_thresh", 0.196)))
