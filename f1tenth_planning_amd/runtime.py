"""Host-side runtime above the C-ABI (include/f1p.h): one `Context` per GPU.

Plumbing only -- numpy arrays in and out, ctypes calls into libf1p.so; no planning arithmetic happens in
Python.  There is no CPU fallback: constructing a Context without the HIP library or without a GPU raises.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import F1PLibraryError, KmpcCfg, LatticeCfg  # noqa: F401


import contextlib
import os
import sys


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner to the C stdout when a communicator is created (flushed at process exit, i.e. AFTER a
    caller's own output -- bench.py must print exactly one JSON line): send fd 1 to stderr for the duration of the call."""
    libc = C.CDLL(None)
    sys.stdout.flush(); libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush(); libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class F1PError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libf1p error {code}: {msg}")
        self.code = code


def _raise(code, msg):
    if code == _abi.F1P_EINVAL:
        raise ValueError(msg)
    raise F1PError(code, msg)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    # (`a.ctypes.data` builds a ctypes helper object per call: 2.3 us; the array interface's address is 1.5 us -- nine of them per plan() call)
    return None if a is None else C.c_void_p(a.__array_interface__["data"][0])


try:                      # xxh3 over the bytes: 6 us for a 1692 x 5 raceline
    import xxhash as _xxhash
except ImportError:       # numpy-only twin below (24 us)
    _xxhash = None
_SIG_WEIGHTS = {}


def _sig_weights(n):
    w = _SIG_WEIGHTS.get(n)
    if w is None:
        i = np.arange(1, n + 1, dtype=np.uint64)
        w = i * np.uint64(0x9E3779B97F4A7C15)
        w ^= w >> np.uint64(29)
        w |= np.uint64(1)
        _SIG_WEIGHTS[n] = w
    return w


def _content_signature(a):
    """Cheap ORDER-SENSITIVE fingerprint of an array's CONTENT for change detection on every plan() call (the reference keeps a
    live reference to the caller's waypoints, so in-place edits -- a reversed raceline, swapped columns, a mirrored track -- must
    be seen).  xxh3-64 of the bytes when xxhash is importable; otherwise a dot product of the 64-bit words, each folded with its
    own high half (so sign-bit flips do not cancel pairwise), with position-dependent odd weights, plus the XOR of the words.
    (Round 2's XOR + plain sum was invariant under permutations and under an even number of sign flips: ADVICE r2.)"""
    a = np.ascontiguousarray(a)
    if _xxhash is not None:
        return _xxhash.xxh3_64_intdigest(a.reshape(-1).view(np.uint8))
    if a.dtype.itemsize == 8 and a.size:
        w = a.reshape(-1).view(np.uint64)
        v = w ^ (w >> np.uint64(31))
        return (int(np.bitwise_xor.reduce(w)), int(np.dot(v, _sig_weights(w.size))))
    import zlib
    return zlib.crc32(a.view(np.uint8).reshape(-1))


class DeviceBuffer:
    """A caller-visible HBM buffer (f1p_dev_alloc) for the *_dev entry points."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ctx._check(ctx.lib.f1p_dev_alloc(ctx.h, C.byref(p), C.c_size_t(self.nbytes)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.f1p_h2d(self.ctx.h, self.ptr, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)))
        self.ctx.sync()   # the host array may die right after the call
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.f1p_d2h(self.ctx.h, C.c_void_p(out.ctypes.data), self.ptr, C.c_size_t(out.nbytes)))
        self.ctx.sync()
        return out

    def free(self):
        if self.ptr is not None and self.ctx.h is not None:
            self.ctx.lib.f1p_dev_free(self.ctx.h, self.ptr)
        self.ptr = None


class Context:
    """One f1p_ctx: one device, one HIP stream.  Not thread-safe; use one Context per thread / rank."""

    def __init__(self, device=0):
        self.lib = _abi.load_library()
        h = C.c_void_p()
        rc = self.lib.f1p_create(C.byref(h), int(device))
        if rc != _abi.F1P_OK:
            msg = self.lib.f1p_last_error(None).decode()
            raise F1PError(rc, msg + " -- the HIP path is mandatory, there is no CPU fallback")
        self.h = h
        self.device = int(device)
        self.n_waypoints = 0
        self._wp_key = None
        self.has_grid = False
        self._pinned = {}        # (tag, shape, dtype) -> numpy view of page-locked memory
        self._pinned_ptrs = []
        self._bundles = {}       # per batch shape: the page-locked arrays of lattice_plan(reuse_outputs=True) / lattice_step and their addresses

    # ---- housekeeping ------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != _abi.F1P_OK:
            _raise(rc, self.lib.f1p_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) is not None:
            for ptr in self._pinned_ptrs:
                self.lib.f1p_host_free(self.h, ptr)
            self._pinned_ptrs = []
            self._pinned = {}
            self._bundles = {}
            self.lib.f1p_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._check(self.lib.f1p_sync(self.h))

    def device_info(self):
        name = C.create_string_buffer(256); arch = C.create_string_buffer(256); cu = C.c_int32()
        self._check(self.lib.f1p_device_info(self.h, name, 256, C.byref(cu), arch, 256))
        return dict(name=name.value.decode(), compute_units=cu.value, arch=arch.value.decode())

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def pinned(self, tag, shape, dtype):
        """numpy array on page-locked host memory (f1p_host_alloc), cached per (tag, shape, dtype) and owned by the
        context: valid until close().  The *_batch calls DMA directly from / into such arrays."""
        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        key = (tag, shape, dtype.str)
        arr = self._pinned.get(key)
        if arr is None:
            nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
            ptr = C.c_void_p()
            self._check(self.lib.f1p_host_alloc(self.h, C.byref(ptr), C.c_size_t(nbytes)))
            self._pinned_ptrs.append(ptr)
            buf = (C.c_char * nbytes).from_address(ptr.value)
            arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
            self._pinned[key] = arr
        return arr

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 1)).upload(arr)

    def timer_begin(self):
        self._check(self.lib.f1p_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float()
        self._check(self.lib.f1p_timer_end(self.h, C.byref(ms)))
        return ms.value

    # ---- scene ---------------------------------------------------------------------------------------
    def set_waypoints(self, waypoints, cols=None):
        """waypoints [N, m>=3]; cols = (x, y, v, psi) column indices, psi = -1 for none.  Default: the
        pure-pursuit layout [x, y, v, psi, ...] (pure_pursuit.py:49)."""
        wp = np.asarray(waypoints)
        if wp.ndim != 2 or wp.shape[1] < 3:
            raise ValueError('Waypoints needs to be a (Nxm), m >= 3, numpy array!')   # pure_pursuit.py:101-102
        wp = _f64(wp)
        if cols is None:
            cols = (0, 1, 2, 3 if wp.shape[1] >= 4 else -1, 4 if wp.shape[1] >= 5 else -1)
        cols = tuple(int(c) for c in cols) + (-1,) * (5 - len(cols))      # (x, y, v, psi, kappa)
        self._check(self.lib.f1p_set_waypoints_ex(self.h, _ptr(wp), wp.shape[0], wp.shape[1], *cols))
        self.n_waypoints = wp.shape[0]

    def set_waypoints_cached(self, waypoints, cols=None):
        """Upload only when the caller's array changed (the reference keeps a live reference to the caller's
        array, pure_pursuit.py:103, so in-place edits must be seen)."""
        wp = np.asarray(waypoints)
        if wp.ndim != 2 or wp.shape[1] < 3:
            raise ValueError('Waypoints needs to be a (Nxm), m >= 3, numpy array!')
        key = (wp.shape, wp.dtype.str, cols, _content_signature(wp))
        if key != self._wp_key:
            self.set_waypoints(wp, cols)
            self._wp_key = key

    def set_grid(self, img, resolution, origin, occupied_below):
        """img [h, w] u8, row 0 = top (ROS map_server); a cell is occupied iff value < occupied_below."""
        if img is None:
            self._check(self.lib.f1p_set_grid(self.h, None, 0, 0, 0.0, 0.0, 0.0, 0))
            self.has_grid = False
            return
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.ndim != 2:
            raise ValueError("occupancy image must be 2-D u8")
        self._check(self.lib.f1p_set_grid(self.h, _ptr(img), img.shape[1], img.shape[0], float(resolution),
                                          float(origin[0]), float(origin[1]), int(occupied_below)))
        self.has_grid = True
        self._grid_shape = (int(img.shape[0]), int(img.shape[1]))

    def grid_distance(self, cap_cells=64):
        """Euclidean distance transform of the installed grid -> f32 [h, w] metres in the image's row order, saturated at
        cap_cells * resolution (f1p_grid_distance_batch)."""
        if not self.has_grid:
            raise F1PError(_abi.F1P_ESTATE, "occupancy grid not set")
        dist = np.empty(self._grid_shape, dtype=np.float32)
        self._check(self.lib.f1p_grid_distance_batch(self.h, _ptr(dist), int(cap_cells)))
        return dist

    def inflate_grid(self, radius):
        """Dilate the collision bitmap by a disc of `radius` metres (0 restores the uploaded grid)."""
        self._check(self.lib.f1p_inflate_grid(self.h, float(radius)))

    def set_footprint(self, offsets, radius):
        """Oriented footprint: discs of `radius` at longitudinal `offsets` [m] along the heading (f1p_set_footprint); offsets = ()
        restores the point test."""
        off = _f64(list(offsets)).reshape(-1)
        self._check(self.lib.f1p_set_footprint(self.h, int(off.shape[0]), _ptr(off) if off.shape[0] else None, float(radius)))

    # ---- leaf kernels ----------------------------------------------------------------------------------
    def nearest_point(self, pts):
        pts = _f64(pts, (-1, 2)); E = pts.shape[0]
        proj = np.empty((E, 2)); dist = np.empty(E); t = np.empty(E); idx = np.empty(E, np.int32)
        self._check(self.lib.f1p_nearest_point_batch(self.h, _ptr(pts), E, _ptr(proj), _ptr(dist), _ptr(t), _ptr(idx)))
        return proj, dist, t, idx

    def intersect_point(self, pts, radius, start_t, wrap=False):
        pts = _f64(pts, (-1, 2)); E = pts.shape[0]
        st = _f64(np.broadcast_to(start_t, (E,)))
        p = np.empty((E, 2)); i = np.empty(E, np.int32); t = np.empty(E); found = np.empty(E, np.int32)
        self._check(self.lib.f1p_intersect_point_batch(self.h, _ptr(pts), _ptr(st), E, float(radius), 1 if wrap else 0,
                                                       _ptr(p), _ptr(i), _ptr(t), _ptr(found)))
        return p, i, t, found.astype(bool)

    def clothoid_g1(self, goals):
        g = _f64(goals, (-1, 3)); n = g.shape[0]
        k0 = np.empty(n); dk = np.empty(n); L = np.empty(n); ok = np.empty(n, np.int32)
        self._check(self.lib.f1p_clothoid_g1_batch(self.h, _ptr(g), n, _ptr(k0), _ptr(dk), _ptr(L), _ptr(ok)))
        return k0, dk, L, ok.astype(bool)

    def clothoid_sample(self, params, npts):
        """params [n, 3] = (kappa0, dkappa, length) -> rows [n, npts, 4] (x, y, theta, |kappa|) in each clothoid's start frame"""
        p = _f64(params, (-1, 3)); n = p.shape[0]
        rows = np.empty((n, int(npts), 4))
        self._check(self.lib.f1p_clothoid_sample_batch(self.h, _ptr(p), n, int(npts), _ptr(rows)))
        return rows

    # ---- pure pursuit ------------------------------------------------------------------------------------
    def pure_pursuit(self, poses, lookahead, wheelbase=0.33, max_reacquire=20.0):
        poses = _f64(poses, (-1, 3)); E = poses.shape[0]
        cols = np.empty(28 * E + 8, np.uint8)                     # the five result columns as views of one buffer: one address look-up (lattice_plan does the same)
        base = cols.__array_interface__["data"][0]
        o8, o4 = 8 * E, 4 * E
        out = dict(steer=cols[0:o8].view(np.float64), speed=cols[o8:2 * o8].view(np.float64), near_idx=cols[2 * o8:2 * o8 + o4].view(np.int32),
                   la_idx=cols[2 * o8 + o4:2 * o8 + 2 * o4].view(np.int32), status=cols[2 * o8 + 2 * o4:2 * o8 + 3 * o4].view(np.int32))
        V = C.c_void_p
        self._check(self.lib.f1p_pure_pursuit_batch(self.h, _ptr(poses), E, float(lookahead), float(wheelbase),
                                                    float(max_reacquire), V(base), V(base + o8), V(base + 2 * o8), V(base + 2 * o8 + o4), V(base + 2 * o8 + 2 * o4)))
        return out

    def pure_pursuit_dev(self, d_poses, E, lookahead, d_steer, d_speed, d_near_idx=None, d_la_idx=None, d_status=None,
                         wheelbase=0.33, max_reacquire=20.0):
        """Asynchronous launch on HBM-resident buffers; poses [E][3]."""
        p = lambda b: None if b is None else b.ptr   # noqa: E731
        self._check(self.lib.f1p_pure_pursuit_dev(self.h, p(d_poses), int(E), float(lookahead), float(wheelbase),
                                                  float(max_reacquire), p(d_steer), p(d_speed), p(d_near_idx), p(d_la_idx),
                                                  p(d_status)))

    def pure_pursuit_set_form(self, egos_per_wave=0):
        """Egos per wave of the batched pure pursuit: 0 (default) by batch size, 1 = k_pure_pursuit, 4 | 8 | 16 = k_pure_pursuit16<G>.  Identical outputs (A/B, tests)."""
        self._check(self.lib.f1p_pure_pursuit_set_form(self.h, int(egos_per_wave)))

    # ---- Stanley / LQR --------------------------------------------------------------------
    def stanley(self, states, wheelbase=0.33, k_path=5.0):
        st = _f64(states, (-1, 4)); E = st.shape[0]
        out = dict(steer=np.empty(E), speed=np.empty(E), near_idx=np.empty(E, np.int32))
        self._check(self.lib.f1p_stanley_batch(self.h, _ptr(st), E, float(wheelbase), float(k_path), _ptr(out["steer"]),
                                               _ptr(out["speed"]), _ptr(out["near_idx"])))
        return out

    def lqr(self, states, err, wheelbase=0.33, timestep=0.01, q=(0.999, 0.0, 0.0066, 0.0), r=0.75, max_iter=50, eps=0.001):
        """err [E, 2] = (e_cog, theta_e) of the previous call; the updated errors come back in out['err']"""
        st = _f64(states, (-1, 4)); E = st.shape[0]
        err = _f64(err, (E, 2)).copy(); qa = _f64(q, (4,))
        out = dict(steer=np.empty(E), speed=np.empty(E), near_idx=np.empty(E, np.int32), err=err)
        self._check(self.lib.f1p_lqr_batch(self.h, _ptr(st), _ptr(err), E, float(wheelbase), float(timestep), _ptr(qa), float(r),
                                           int(max_iter), float(eps), _ptr(out["steer"]), _ptr(out["speed"]), _ptr(out["near_idx"])))
        return out

    # ---- lattice -------------------------------------------------------------------------------------------
    def lattice_plan(self, poses, cfg: LatticeCfg, goals=None, prev_theta=None, want_traj=True, want_all=False,
                     reuse_outputs=False, traj_dtype=np.float64):
        """reuse_outputs: results land in page-locked arrays owned by the context (no bounce buffers, no fresh pages per
        call); they are overwritten by the next call with the same batch shape.
        traj_dtype=np.float32: best_traj comes back as f32 rows (f1p_lattice_plan_batch_f32: the fp64 rows rounded once on the
        device, half the PCIe bytes); everything else is unchanged."""
        f32 = np.dtype(traj_dtype) == np.float32
        if f32 and want_all:
            raise ValueError("traj_dtype=float32 is a winner-only mode (no all_cost / all_traj)")
        poses = _f64(poses, (-1, 4)); E = poses.shape[0]; Cn = cfg.n_cand; S = cfg.n_stations
        g = None if goals is None else _f64(goals, (E, Cn, 3))
        pt = None if prev_theta is None else _f64(prev_theta, (E, S))
        ptrs = None
        if reuse_outputs:
            # the page-locked arrays of this batch shape and their addresses are looked up ONCE (eight pinned() look-ups and nine ctypes pointer objects were
            # ~15 us of a 0.22 ms call)
            key = ("plan", E, S, f32, bool(want_traj))
            b = self._bundles.get(key)
            if b is None:
                pin = self.pinned
                hp = pin("lat_poses", (E, 4), np.float64)
                o = dict(steer=pin("lat_steer", E, np.float64), speed=pin("lat_speed", E, np.float64),
                         best_idx=pin("lat_bidx", E, np.int32), best_cost=pin("lat_bcost", E, np.float64),
                         status=pin("lat_status", E, np.int32), near_idx=pin("lat_near", E, np.int32))
                if want_traj:
                    o["best_traj"] = pin("lat_traj32" if f32 else "lat_traj", (E, S, 4), np.float32 if f32 else np.float64)
                b = self._bundles[key] = (hp, o, {k: _ptr(v) for k, v in o.items()}, _ptr(hp))
            hp, o, ptrs, php = b
            hp[...] = poses; poses = hp
            out = dict(o)
        else:
            # fresh arrays for the caller: the six result columns are views of ONE buffer, so that one address look-up (1.5 us each) serves all of them
            cols = np.empty(36 * E + 8, np.uint8)
            base = cols.__array_interface__["data"][0]
            o8, o4 = 8 * E, 4 * E
            out = dict(steer=cols[0:o8].view(np.float64), speed=cols[o8:2 * o8].view(np.float64), best_cost=cols[2 * o8:3 * o8].view(np.float64),
                       best_idx=cols[3 * o8:3 * o8 + o4].view(np.int32), status=cols[3 * o8 + o4:3 * o8 + 2 * o4].view(np.int32),
                       near_idx=cols[3 * o8 + 2 * o4:3 * o8 + 3 * o4].view(np.int32))
            ptrs = dict(steer=C.c_void_p(base), speed=C.c_void_p(base + o8), best_cost=C.c_void_p(base + 2 * o8), best_idx=C.c_void_p(base + 3 * o8),
                        status=C.c_void_p(base + 3 * o8 + o4), near_idx=C.c_void_p(base + 3 * o8 + 2 * o4))
            php = None
            if want_traj:
                out["best_traj"] = np.empty((E, S, 4), np.float32 if f32 else np.float64)
                ptrs["best_traj"] = _ptr(out["best_traj"])
        if want_all:
            out["all_cost"] = np.empty((E, Cn)); out["all_traj"] = np.empty((E, Cn, S, 4))
        if cfg.cand_count > 0:            # a candidate shard only evaluates: (best_idx, best_cost, near_idx)
            for k in ("steer", "speed", "status", "best_traj"):
                out.pop(k, None)
        if ptrs is not None and not want_all:
            P = lambda k: ptrs[k] if k in out else None   # noqa: E731
            pp = php if php is not None else _ptr(poses)
        else:
            P = lambda k: _ptr(out.get(k))                # noqa: E731
            pp = _ptr(poses)
        if f32:
            self._check(self.lib.f1p_lattice_plan_batch_f32(self.h, pp, _ptr(g), _ptr(pt), E, C.byref(cfg),
                                                            P("steer"), P("speed"), P("best_idx"), P("best_cost"), P("status"), P("near_idx"), P("best_traj")))
            return out
        self._check(self.lib.f1p_lattice_plan_batch(self.h, pp, _ptr(g), _ptr(pt), E, C.byref(cfg),
                                                    P("steer"), P("speed"), P("best_idx"), P("best_cost"), P("status"), P("near_idx"), P("best_traj"),
                                                    _ptr(out.get("all_cost")), _ptr(out.get("all_traj"))))
        return out

    def lattice_plan_dev(self, d_poses, E, cfg: LatticeCfg, d_steer, d_speed, d_best_idx, d_best_cost=None, d_status=None,
                         d_near_idx=None, d_best_traj=None, d_goals=None, d_prev_theta=None, d_all_cost=None,
                         d_all_traj=None):
        """Asynchronous launch on HBM-resident buffers (DeviceBuffer or None)."""
        p = lambda b: None if b is None else b.ptr   # noqa: E731
        self._check(self.lib.f1p_lattice_plan_dev(self.h, p(d_poses), p(d_goals), p(d_prev_theta), int(E), C.byref(cfg),
                                                  p(d_steer), p(d_speed), p(d_best_idx), p(d_best_cost), p(d_status),
                                                  p(d_near_idx), p(d_best_traj), p(d_all_cost), p(d_all_traj)))

    def lattice_step(self, poses, cfg: LatticeCfg, keep_traj=False):
        """One closed-loop control step (f1p_lattice_step_batch): poses [E, 4] -> dict(steer, speed, status), page-locked arrays owned by
        the context (overwritten by the next step of the same batch size).  The previous plan's headings (similarity term) stay on the
        device; keep_traj=True keeps the winners' rows there too (lattice_fetch_traj)."""
        E = int(np.shape(poses)[0])
        b = self._bundles.get(("step", E))
        if b is None:
            hp = self.pinned("step_poses", (E, 4), np.float64)
            o = dict(steer=self.pinned("step_steer", E, np.float64), speed=self.pinned("step_speed", E, np.float64),
                     status=self.pinned("step_status", E, np.int32))
            b = self._bundles[("step", E)] = (hp, o, (_ptr(hp), _ptr(o["steer"]), _ptr(o["speed"]), _ptr(o["status"])))
        hp, o, (php, ps, pv, pt) = b
        hp[...] = poses
        self._check(self.lib.f1p_lattice_step_batch(self.h, php, E, C.byref(cfg), ps, pv, pt, 1 if keep_traj else 0))
        return dict(o)

    def lattice_fetch_traj(self, E, S):
        """the winners' rows [E, S, 4] of the last lattice_step(keep_traj=True)"""
        out = np.empty((int(E), int(S), 4))
        self._check(self.lib.f1p_lattice_fetch_traj(self.h, _ptr(out), int(E), int(S)))
        return out

    def lattice_set_closed_loop(self, on=True):
        """closed-loop mode: every plan's winning headings stay on the device and are the next plan's prev_theta (similarity cost,
        lattice_planner.py:287-296) whenever prev_theta is None; (re)arming forgets the previous path"""
        self._check(self.lib.f1p_lattice_set_closed_loop(self.h, 1 if on else 0))

    def lattice_closed_loop_prev(self):
        """the headings the NEXT closed-loop plan would use as prev_theta: numpy [E, S] (a copy), or None"""
        ptr = C.c_void_p(); E = C.c_int32(); S = C.c_int32()
        self._check(self.lib.f1p_lattice_closed_loop_state(self.h, C.byref(ptr), C.byref(E), C.byref(S)))
        if not ptr.value:
            return None
        out = np.empty((E.value, S.value))
        self._check(self.lib.f1p_d2h(self.h, C.c_void_p(out.ctypes.data), ptr, C.c_size_t(out.nbytes)))
        self.sync()
        return out

    def lattice_set_mode(self, mixed=1, d_cost32=None, d_state=None):
        """0: all fp64; 1 (default): f32 filter + fp64 decision, every plan shape from one ego (two-egos-per-wave prologue from 3072 egos); 2: always,
        two-ego prologue at any size; 3: as 2 with the one-ego-per-wave prologue (A/B, tests).  Optional device buffers [E][C] receive
        the filter's costs (f32) and states (i32)."""
        self._check(self.lib.f1p_lattice_set_mode(self.h, int(mixed), None if d_cost32 is None else d_cost32.ptr,
                                                  None if d_state is None else d_state.ptr))

    def lattice_set_split(self, groups=0):
        """workgroups per ego of the single-kernel lattice schedules (0 = automatic)"""
        self._check(self.lib.f1p_lattice_set_split(self.h, int(groups)))

    def lattice_set_clearance(self, stations_each_side=2):
        """f32 filter's occupancy test: one station in 2 r + 1 against the clearance map (r > 0) or every station against the bitmap (0)"""
        self._check(self.lib.f1p_lattice_set_clearance(self.h, int(stations_each_side)))

    def lattice_debug_queue(self, E):
        """entries per ego the last mixed-schedule plan of E egos handed to the fp64 refinement (numpy int32 [E])"""
        out = np.empty(int(E), np.int32)
        self._check(self.lib.f1p_lattice_debug_queue(self.h, _ptr(out), int(E)))
        return out

    def lattice_debug_bound(self, d_bound=None):
        """test hook: [E][C] f32 device buffer for the f32 filter's per-candidate a-priori cost error bounds (None = off)"""
        self._check(self.lib.f1p_lattice_debug_bound(self.h, None if d_bound is None else d_bound.ptr))

    def lattice_set_order(self, heavy_first=True):
        """dispatch order of the candidate kernel: egos whose previous plan took the long station pass first (default) or ego order; outputs identical"""
        self._check(self.lib.f1p_lattice_set_order(self.h, 1 if heavy_first else 0))

    def lattice_debug_pass(self, d_pass=None):
        """measurement hook: [E][4] i32 device buffer (zeroed by the caller) for the lazy station pass's per-ego statistics -- candidates
        looked at, of them lane-per-candidate, rounds, queue entries (None = off)"""
        self._check(self.lib.f1p_lattice_debug_pass(self.h, None if d_pass is None else d_pass.ptr))

    def lattice_set_audit(self, every_n=0, n_egos=64):
        """every every_n-th mixed plan is re-planned on a moving window of n_egos egos by the all-fp64 kernel and compared bit for bit"""
        self._check(self.lib.f1p_lattice_set_audit(self.h, int(every_n), int(n_egos)))

    def lattice_audit_read(self, reset=False):
        """dict(plans, egos, mismatching_egos) of the runtime audit since the last reset"""
        out = (C.c_uint64 * 3)()
        self._check(self.lib.f1p_lattice_audit_read(self.h, out, 1 if reset else 0))
        return dict(plans=int(out[0]), egos=int(out[1]), mismatching_egos=int(out[2]))

    def lattice_set_pipeline(self, chunks=0):
        """chunks of egos a mixed plan is pipelined in over two internal streams (0 = automatic, 1 = off)"""
        self._check(self.lib.f1p_lattice_set_pipeline(self.h, int(chunks)))

    def lattice_profile(self, enable=True, read=False):
        """HIP-event timing around the kernels of the mixed schedule; read=True returns (prologue, filter, refine, select) ms of the
        last profiled plan (prologue = 0 when the one-kernel filter ran)"""
        ms = (C.c_float * 4)()
        self._check(self.lib.f1p_lattice_profile(self.h, 1 if enable else 0, ms if read else None))
        return tuple(ms) if read else None

    def lattice_emit_dev(self, d_poses, E, cfg: LatticeCfg, d_cand_idx, d_cand_cost, d_steer, d_speed, d_status=None,
                         d_near_idx=None, d_best_traj=None, d_goals=None):
        p = lambda b: None if b is None else b.ptr   # noqa: E731
        self._check(self.lib.f1p_lattice_emit_dev(self.h, p(d_poses), p(d_goals), int(E), C.byref(cfg), p(d_cand_idx),
                                                  p(d_cand_cost), p(d_steer), p(d_speed), p(d_status), p(d_near_idx),
                                                  p(d_best_traj)))

    # ---- kinematic MPC -------------------------------------------------------------------------------------
    def kmpc_ref(self, states, horizon, dt=0.1, dl=0.03):
        st = _f64(states, (-1, 4)); E = st.shape[0]
        ref = np.empty((E, 4, horizon + 1))
        self._check(self.lib.f1p_kmpc_ref_batch(self.h, _ptr(st), E, int(horizon), float(dt), float(dl), _ptr(ref)))
        return ref

    def kmpc_set_mode(self, mixed=True, d_cost32=None, d_n_refined=None):
        """mixed: f32 filter + fp64 refinement (default) or plain fp64; optional device buffers receive the filter diagnostics"""
        self._check(self.lib.f1p_kmpc_set_mode(self.h, 1 if mixed else 0, None if d_cost32 is None else d_cost32.ptr,
                                               None if d_n_refined is None else d_n_refined.ptr))

    def kmpc_predict(self, x0, oa, od, cfg: KmpcCfg):
        """predict_motion_kinematic (kinematic_mpc.py:208-221) for E egos -> path [E, 4, T+1]"""
        x0 = _f64(x0, (-1, 4)); E = x0.shape[0]; T = cfg.horizon
        oa = _f64(oa, (E, T)); od = _f64(od, (E, T))
        path = np.empty((E, 4, T + 1))
        self._check(self.lib.f1p_kmpc_predict_batch(self.h, _ptr(x0), _ptr(oa), _ptr(od), E, C.byref(cfg), _ptr(path)))
        return path

    def kmpc_shoot(self, x0, ref, controls, cfg: KmpcCfg, want_seq=True):
        x0 = _f64(x0, (-1, 4)); E = x0.shape[0]; T = cfg.horizon; R = cfg.n_rollouts
        ref = _f64(ref, (E, 4, T + 1))
        controls = np.ascontiguousarray(controls, dtype=np.float32)
        if controls.shape != (E, T, 2, R):
            raise ValueError(f"controls must be f32 [E={E}, T={T}, 2, R={R}]")
        out = dict(steer=np.empty(E), speed=np.empty(E), best_idx=np.empty(E, np.int32), best_cost=np.empty(E))
        if want_seq:
            out["best_seq"] = np.empty((E, T, 2))
        self._check(self.lib.f1p_kmpc_shoot_batch(self.h, _ptr(x0), _ptr(ref), _ptr(controls), E, C.byref(cfg),
                                                  _ptr(out["steer"]), _ptr(out["speed"]), _ptr(out["best_idx"]),
                                                  _ptr(out["best_cost"]), _ptr(out.get("best_seq"))))
        return out

    def kmpc_shoot_dev(self, d_x0, d_ref, d_controls, E, cfg: KmpcCfg, d_steer, d_speed, d_best_idx, d_best_cost=None,
                       d_best_seq=None):
        p = lambda b: None if b is None else b.ptr   # noqa: E731
        self._check(self.lib.f1p_kmpc_shoot_dev(self.h, p(d_x0), p(d_ref), p(d_controls), int(E), C.byref(cfg), p(d_steer),
                                                p(d_speed), p(d_best_idx), p(d_best_cost), p(d_best_seq)))

    # in-kernel control generation + device-resident warm start (f1p_kmpc_plan_*)
    def kmpc_plan(self, x0, cfg: KmpcCfg, sampler, dl=0.03, want_seq=True, want_cost=True):
        """KMPCPlanner.plan for E egos in ONE call: reference extraction, sampling around the ctx's warm start, rollouts,
        argmin, new warm start -- nothing but x0 goes up and the winners come down."""
        x0 = _f64(x0, (-1, 4)); E = x0.shape[0]; T = cfg.horizon
        out = dict(steer=np.empty(E), speed=np.empty(E), best_idx=np.empty(E, np.int32))
        if want_cost:
            out["best_cost"] = np.empty(E)
        if want_seq:
            out["best_seq"] = np.empty((E, T, 2))
        self._check(self.lib.f1p_kmpc_plan_batch(self.h, _ptr(x0), E, C.byref(cfg), float(dl), C.byref(sampler), _ptr(out["steer"]),
                                                 _ptr(out["speed"]), _ptr(out["best_idx"]), _ptr(out.get("best_cost")), _ptr(out.get("best_seq"))))
        return out

    def kmpc_plan_dev(self, d_x0, d_ref, E, cfg: KmpcCfg, sampler, d_steer, d_speed, d_best_idx, d_best_cost=None, d_best_seq=None):
        p = lambda b: None if b is None else b.ptr   # noqa: E731
        self._check(self.lib.f1p_kmpc_plan_dev(self.h, p(d_x0), p(d_ref), int(E), C.byref(cfg), C.byref(sampler), p(d_steer), p(d_speed),
                                               p(d_best_idx), p(d_best_cost), p(d_best_seq)))

    def kmpc_gen_controls_dev(self, d_controls, E, cfg: KmpcCfg, sampler):
        self._check(self.lib.f1p_kmpc_gen_controls_dev(self.h, d_controls.ptr, int(E), C.byref(cfg), C.byref(sampler)))

    def kmpc_warm_reset(self):
        self._check(self.lib.f1p_kmpc_warm_reset(self.h))

    def kmpc_warm_get(self, E, T):
        w = np.empty((int(E), int(T), 2), np.float32)
        self._check(self.lib.f1p_kmpc_warm_get(self.h, _ptr(w), int(E), int(T)))
        return w

    def kmpc_warm_set(self, warm):
        w = np.ascontiguousarray(warm, np.float32)
        self._check(self.lib.f1p_kmpc_warm_set(self.h, _ptr(w), w.shape[0], w.shape[1]))

    def kmpc_set_yaw_fixup(self, on=True):
        """k_kmpc_ref's per-ego heading fold (kinematic_mpc.py:198-203) on the gathered values; off = the caller maintains the array"""
        self._check(self.lib.f1p_kmpc_set_yaw_fixup(self.h, 1 if on else 0))

    def kmpc_set_groups(self, groups=0):
        self._check(self.lib.f1p_kmpc_set_groups(self.h, int(groups)))

    def kmpc_sample_controls_dev(self, d_controls, E, cfg: KmpcCfg, seed, sigma_accel=1.5, sigma_steer=0.15):
        self._check(self.lib.f1p_kmpc_sample_controls_dev(self.h, d_controls.ptr, int(E), C.byref(cfg),
                                                          C.c_uint64(int(seed)), float(sigma_accel), float(sigma_steer)))

    # ---- dynamic single-track shooting ------------------------------------------------------
    def stmpc_predict(self, x0, oa, od_v, cfg):
        """predict_motion (dynamic_mpc.py:280-300) for E egos -> path [E, 7, T+1]"""
        x0 = _f64(x0, (-1, 7)); E = x0.shape[0]; T = cfg.horizon
        oa = _f64(oa, (E, T)); od = _f64(od_v, (E, T))
        path = np.empty((E, 7, T + 1))
        self._check(self.lib.f1p_stmpc_predict_batch(self.h, _ptr(x0), _ptr(oa), _ptr(od), E, C.byref(cfg), _ptr(path)))
        return path

    def stmpc_ref(self, states, horizon, dt=0.025, dl=0.03):
        st = _f64(states, (-1, 4)); E = st.shape[0]
        ref = np.empty((E, 7, horizon + 1))
        self._check(self.lib.f1p_stmpc_ref_batch(self.h, _ptr(st), E, int(horizon), float(dt), float(dl), _ptr(ref)))
        return ref

    def stmpc_set_mode(self, mixed=True, d_cost32=None, d_n_refined=None):
        """f32 filter + fp64 decision (default) or plain fp64 for the dynamic single-track shooting; the buffers are test hooks"""
        self._check(self.lib.f1p_stmpc_set_mode(self.h, 1 if mixed else 0, None if d_cost32 is None else d_cost32.ptr,
                                                None if d_n_refined is None else d_n_refined.ptr))

    def stmpc_shoot_dev(self, d_x0, d_ref, d_controls, E, cfg, d_steer, d_speed, d_best_idx, d_best_cost=None, d_best_seq=None):
        """Asynchronous launch on HBM-resident buffers: x0 [E][7], ref [E][7][T+1], controls f32 [E][T][2][R]."""
        p = lambda b: None if b is None else b.ptr   # noqa: E731
        self._check(self.lib.f1p_stmpc_shoot_dev(self.h, p(d_x0), p(d_ref), p(d_controls), int(E), C.byref(cfg), p(d_steer), p(d_speed),
                                                 p(d_best_idx), p(d_best_cost), p(d_best_seq)))

    def stmpc_shoot(self, x0, ref, controls, cfg, want_seq=True):
        x0 = _f64(x0, (-1, 7)); E = x0.shape[0]; T = cfg.horizon; R = cfg.n_rollouts
        ref = _f64(ref, (E, 7, T + 1))
        controls = np.ascontiguousarray(controls, dtype=np.float32)
        if controls.shape != (E, T, 2, R):
            raise ValueError(f"controls must be f32 [E={E}, T={T}, 2, R={R}]")
        out = dict(steer=np.empty(E), speed=np.empty(E), best_idx=np.empty(E, np.int32), best_cost=np.empty(E))
        if want_seq:
            out["best_seq"] = np.empty((E, T, 2))
        self._check(self.lib.f1p_stmpc_shoot_batch(self.h, _ptr(x0), _ptr(ref), _ptr(controls), E, C.byref(cfg), _ptr(out["steer"]),
                                                   _ptr(out["speed"]), _ptr(out["best_idx"]), _ptr(out["best_cost"]),
                                                   _ptr(out.get("best_seq"))))
        return out

    # ---- multi-GPU exchange step -----------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_uint8 * _abi.COMM_ID_BYTES)()
        with _stdout_to_stderr():
            rc = self.lib.f1p_comm_unique_id(self.h, buf)
        self._check(rc)
        return bytes(buf)

    def comm_init(self, uid, nranks, rank):
        buf = (C.c_uint8 * _abi.COMM_ID_BYTES).from_buffer_copy(uid)
        with _stdout_to_stderr():
            rc = self.lib.f1p_comm_init(self.h, buf, int(nranks), int(rank))
        self._check(rc)

    def comm_info(self):
        """(nranks, rank) as the RCCL communicator reports them."""
        n = C.c_int32(); r = C.c_int32()
        self._check(self.lib.f1p_comm_info(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_set_exchange(self, mode=0):
        """form of the cross-rank argmin: 0 = two RCCL all-reduces (default), 1 = one all-gather of (key, index) + a local minimum"""
        self._check(self.lib.f1p_comm_set_exchange(self.h, int(mode)))

    def argmin_gather_reduce(self, cost, idx):
        """the local kernels of exchange form 1 on N emulated ranks: cost / idx [N, E] -> (idx [E], cost [E])"""
        cost = _f64(cost); idx = np.ascontiguousarray(idx, np.int32)
        N, E = cost.shape
        io = np.empty(E, np.int32); co = np.empty(E)
        self._check(self.lib.f1p_argmin_gather_reduce_batch(self.h, _ptr(cost), _ptr(idx), N, E, _ptr(io), _ptr(co)))
        return io, co

    def comm_argmin_dev(self, d_cost, d_idx, E):
        self._check(self.lib.f1p_comm_argmin_dev(self.h, d_cost.ptr, d_idx.ptr, int(E)))

    def argmin_key(self, cost):
        """the cost -> u64 key map of the cross-rank argmin (np.argmin order, NaN first)"""
        cost = _f64(cost).reshape(-1); E = cost.shape[0]
        keys = np.empty(E, np.uint64)
        self._check(self.lib.f1p_argmin_key_batch(self.h, _ptr(cost), E, _ptr(keys)))
        return keys

    def argmin_mask(self, own_keys, min_keys, idx):
        own = np.ascontiguousarray(own_keys, np.uint64); mn = np.ascontiguousarray(min_keys, np.uint64)
        idx = np.ascontiguousarray(idx, np.int32); E = own.shape[0]
        masked = np.empty(E, np.int32); cost = np.empty(E)
        self._check(self.lib.f1p_argmin_mask_batch(self.h, _ptr(own), _ptr(mn), _ptr(idx), E, _ptr(masked), _ptr(cost)))
        return masked, cost


class MultiContext:
    """One process, several GPUs (SURVEY.md 8b "Threading"): one Context per device, each driven by its own host thread
    (ctypes releases the GIL for the duration of a C call, and distinct f1p_ctx are thread-safe).  Egos are independent, so a
    batch is cut into contiguous ego ranges -- GPU g gets egos [g E/G, (g+1) E/G) -- with NO collective; the scene (waypoints,
    grid) is replicated.  Results are concatenated in ego order, so the output is identical to one Context planning the whole
    batch.  `devices` may repeat an index (two contexts, two streams on one GPU): that is how a single-GPU box tests this path.
    """

    def __init__(self, devices=None):
        from concurrent.futures import ThreadPoolExecutor
        if devices is None:
            n = _abi.load_library().f1p_device_count()
            if n <= 0:
                raise F1PError(_abi.F1P_ENODEV, "no HIP device visible -- the HIP path is mandatory, there is no CPU fallback")
            devices = range(n)
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        self.ctxs = [Context(d) for d in self.devices]
        self._pool = ThreadPoolExecutor(max_workers=len(self.ctxs), thread_name_prefix="f1p-gpu")

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for c in getattr(self, "ctxs", []):
            c.close()
        self.ctxs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _each(self, fn):
        """run fn(ctx, g) on every context's thread; re-raises the first failure"""
        return [f.result() for f in [self._pool.submit(fn, c, g) for g, c in enumerate(self.ctxs)]]

    # ---- scene: replicated ---------------------------------------------------------------------------------------
    def set_waypoints(self, waypoints, cols=None):
        self._each(lambda c, g: c.set_waypoints(waypoints, cols))

    def set_waypoints_cached(self, waypoints, cols=None):
        self._each(lambda c, g: c.set_waypoints_cached(waypoints, cols))

    def set_grid(self, img, resolution, origin, occupied_below):
        self._each(lambda c, g: c.set_grid(img, resolution, origin, occupied_below))

    def inflate_grid(self, radius):
        self._each(lambda c, g: c.inflate_grid(radius))

    def set_footprint(self, offsets, radius):
        self._each(lambda c, g: c.set_footprint(offsets, radius))

    def sync(self):
        self._each(lambda c, g: c.sync())

    def lattice_set_closed_loop(self, on=True):
        """closed-loop mode on every replica: the ego ranges of a batch are stable, so each context keeps the headings of ITS
        egos' winners and the chain equals the single-context chain"""
        self._each(lambda c, g: c.lattice_set_closed_loop(on))

    def lattice_closed_loop_prev(self):
        """the headings the next closed-loop plan would use, concatenated in ego order (None when no replica holds any)"""
        parts = [p for p in self._each(lambda c, g: c.lattice_closed_loop_prev()) if p is not None]
        return np.concatenate(parts, axis=0) if parts else None

    # ---- batched planners: ego-sharded -----------------------------------------------------------------------------
    def _sharded(self, n_items, call):
        """call(ctx, lo, hi) -> dict of arrays with leading dimension hi - lo; concatenated over the ego ranges"""
        from .dist import shard_range
        G = len(self.ctxs)
        ranges = [shard_range(n_items, g, G) for g in range(G)]

        def run(c, g):
            lo, hi = ranges[g]
            return call(c, lo, hi) if hi > lo else None
        parts = [p for p in self._each(run) if p is not None]
        if not parts:
            return call(self.ctxs[0], 0, 0)
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}

    def lattice_plan(self, poses, cfg, goals=None, prev_theta=None, want_traj=True, traj_dtype=np.float64):
        poses = _f64(poses, (-1, 4)); E = poses.shape[0]
        g = None if goals is None else _f64(goals, (E, cfg.n_cand, 3))
        pt = None if prev_theta is None else _f64(prev_theta, (E, cfg.n_stations))
        return self._sharded(E, lambda c, lo, hi: c.lattice_plan(poses[lo:hi], cfg, None if g is None else g[lo:hi],
                                                                   None if pt is None else pt[lo:hi], want_traj=want_traj, traj_dtype=traj_dtype))

    def pure_pursuit(self, poses, lookahead, wheelbase=0.33, max_reacquire=20.0):
        poses = _f64(poses, (-1, 3))
        return self._sharded(poses.shape[0], lambda c, lo, hi: c.pure_pursuit(poses[lo:hi], lookahead, wheelbase, max_reacquire))

    def kmpc_ref(self, states, horizon, dt=0.1, dl=0.03):
        st = _f64(states, (-1, 4))
        return self._sharded(st.shape[0], lambda c, lo, hi: dict(ref=c.kmpc_ref(st[lo:hi], horizon, dt, dl)))["ref"]

    def kmpc_shoot(self, x0, ref, controls, cfg, want_seq=True):
        x0 = _f64(x0, (-1, 4))
        return self._sharded(x0.shape[0], lambda c, lo, hi: c.kmpc_shoot(x0[lo:hi], ref[lo:hi], controls[lo:hi], cfg, want_seq))


_default_ctx = None


def default_context():
    """Process-wide Context on LOCAL_RANK (or device 0) used by the drop-in planner classes."""
    global _default_ctx
    if _default_ctx is None:
        import os
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx
