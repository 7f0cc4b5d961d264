"""ctypes mirror of include/f1p.h and the loader of libf1p.so.

There is no CPU fallback: if the HIP library is missing or no MI355X is visible the product path raises.
The library is opened with RTLD_DEEPBIND so its HIP symbols bind to the ROCm runtime it was linked
against (/opt/rocm) even when a process also imports torch, which bundles its own copy of libamdhip64.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libf1p.so")

F1P_OK = 0
F1P_EINVAL, F1P_ENODEV, F1P_EHIP, F1P_ESTATE, F1P_ENOMEM, F1P_ECOMM = -1, -2, -3, -4, -5, -6
ST_INTERSECT, ST_REACQUIRE, ST_NO_LOOKAHEAD, ST_ALL_BLOCKED = 0, 1, 2, 3
MAX_LOOKAHEADS = 64
MAX_WIDTHS = 64
COMM_ID_BYTES = 128
LA_IDX_NONE = -(2 ** 31)
GEN_CLOTHOID, GEN_CUBIC = 0, 1


class LatticeCfg(C.Structure):
    """struct f1p_lattice_cfg (include/f1p.h)"""
    _fields_ = [
        ("n_stations", C.c_int32), ("n_lookahead", C.c_int32), ("n_width", C.c_int32),
        ("n_shift", C.c_int32), ("n_cull", C.c_int32), ("check_collision", C.c_int32),
        ("cand_begin", C.c_int32), ("cand_count", C.c_int32), ("generator", C.c_int32), ("prune", C.c_int32),
        ("lookahead", C.c_double * MAX_LOOKAHEADS), ("width", C.c_double * MAX_WIDTHS),
        ("w_length", C.c_double), ("w_max_kappa", C.c_double), ("w_mean_kappa", C.c_double),
        ("w_similarity", C.c_double), ("track_lookahead", C.c_double), ("wheelbase", C.c_double),
        ("max_reacquire", C.c_double),
    ]

    @property
    def n_cand(self):
        return self.n_lookahead * self.n_width


class KmpcCfg(C.Structure):
    """struct f1p_kmpc_cfg (include/f1p.h)"""
    _fields_ = [
        ("horizon", C.c_int32), ("n_rollouts", C.c_int32),
        ("dt", C.c_double), ("wheelbase", C.c_double), ("max_steer", C.c_double), ("max_dsteer", C.c_double),
        ("max_speed", C.c_double), ("min_speed", C.c_double), ("max_accel", C.c_double),
        ("q", C.c_double * 4), ("qf", C.c_double * 4), ("r", C.c_double * 2), ("rd", C.c_double * 2),
    ]


class KmpcSampler(C.Structure):
    """struct f1p_kmpc_sampler (include/f1p.h)"""
    _fields_ = [("seed", C.c_uint64), ("call", C.c_uint32), ("use_warm", C.c_int32), ("sigma_accel", C.c_double), ("sigma_steer", C.c_double)]


def kmpc_sampler(seed=0, call=0, use_warm=True, sigma_accel=1.5, sigma_steer=0.15):
    s = KmpcSampler()
    s.seed, s.call, s.use_warm = int(seed) & (2 ** 64 - 1), int(call) & 0xffffffff, 1 if use_warm else 0
    s.sigma_accel, s.sigma_steer = float(sigma_accel), float(sigma_steer)
    return s


class StmpcCfg(C.Structure):
    """struct f1p_stmpc_cfg (include/f1p.h)"""
    _fields_ = [
        ("horizon", C.c_int32), ("n_rollouts", C.c_int32),
        ("dt", C.c_double), ("wheelbase", C.c_double), ("max_steer", C.c_double), ("max_steer_v", C.c_double),
        ("max_speed", C.c_double), ("min_speed", C.c_double), ("max_accel", C.c_double),
        ("q", C.c_double * 7), ("qf", C.c_double * 7), ("r", C.c_double * 2), ("rd", C.c_double * 2), ("params", C.c_double * 8),
    ]


def stmpc_cfg(horizon=40, n_rollouts=512, dt=0.025, wheelbase=0.33, max_steer=0.4189, max_steer_v=3.2, max_speed=6.0,
              min_speed=0.0, max_accel=3.0, q=(32.0, 32.0, 0.0, 1.0, 0.5, 0.0, 0.0), qf=(32.0, 32.0, 0.0, 1.0, 0.5, 0.0, 0.0),
              r=(0.5, 0.01), rd=(0.3, 0.01), params=(3.74, 0.15875, 0.17145, 0.074, 4.718, 5.4562, 0.04712, 1.0489)):
    """Build a StmpcCfg with the defaults of dynamic_mpc.py's mpc_config (:40-86) and STMPCPlanner's vehicle parameters."""
    cfg = StmpcCfg()
    cfg.horizon, cfg.n_rollouts = int(horizon), int(n_rollouts)
    cfg.dt, cfg.wheelbase, cfg.max_steer, cfg.max_steer_v = float(dt), float(wheelbase), float(max_steer), float(max_steer_v)
    cfg.max_speed, cfg.min_speed, cfg.max_accel = float(max_speed), float(min_speed), float(max_accel)
    for i in range(7):
        cfg.q[i] = float(q[i]); cfg.qf[i] = float(qf[i])
    for i in range(2):
        cfg.r[i] = float(r[i]); cfg.rd[i] = float(rd[i])
    for i in range(8):
        cfg.params[i] = float(params[i])
    return cfg


def lattice_cfg(lookaheads=(0.4, 0.6, 0.8, 1.0), widths=None, n_stations=100, weights=(1.0, 0.0, 0.0, 0.0),
                n_shift=1, n_cull=1, check_collision=True, track_lookahead=0.8, wheelbase=0.33,
                max_reacquire=20.0, cand_begin=0, cand_count=0, generator="clothoid", prune=False):
    """Build a LatticeCfg.  prune: branch and bound over the candidates (bit-identical outputs, fewer station
    loops).  Defaults are the reference's: look-aheads [0.4, 0.6, 0.8, 1.0] and
    widths linspace(-1, 1, 7) (lattice_planner.py:228-229), 100 stations (:197), tracker look-ahead 0.8
    (:211), tracker wheelbase 0.33 (:55), only the length cost runnable (:268-271)."""
    import numpy as np
    if widths is None:
        widths = np.linspace(-1.0, 1.0, num=7)
    lookaheads = [float(v) for v in lookaheads]
    widths = [float(v) for v in widths]
    if not (1 <= len(lookaheads) <= MAX_LOOKAHEADS) or not (1 <= len(widths) <= MAX_WIDTHS):
        raise ValueError("between 1 and 64 look-ahead distances and widths are supported")
    if len(weights) != 4:
        raise ValueError("weights = (length, max_kappa, mean_kappa, similarity)")
    cfg = LatticeCfg()
    cfg.n_stations = int(n_stations)
    cfg.n_lookahead = len(lookaheads)
    cfg.n_width = len(widths)
    cfg.n_shift = int(n_shift)
    cfg.n_cull = int(n_cull)
    cfg.check_collision = 1 if check_collision else 0
    cfg.cand_begin = int(cand_begin)
    cfg.cand_count = int(cand_count)
    cfg.prune = 1 if prune else 0
    gens = {"clothoid": GEN_CLOTHOID, "cubic": GEN_CUBIC, GEN_CLOTHOID: GEN_CLOTHOID, GEN_CUBIC: GEN_CUBIC}
    if generator not in gens:
        raise ValueError("generator must be 'clothoid' or 'cubic'")
    cfg.generator = gens[generator]
    for i, v in enumerate(lookaheads):
        cfg.lookahead[i] = v
    for i, v in enumerate(widths):
        cfg.width[i] = v
    cfg.w_length, cfg.w_max_kappa, cfg.w_mean_kappa, cfg.w_similarity = [float(w) for w in weights]
    cfg.track_lookahead = float(track_lookahead)
    cfg.wheelbase = float(wheelbase)
    cfg.max_reacquire = float(max_reacquire)
    return cfg


def kmpc_cfg(horizon=8, n_rollouts=512, dt=0.1, wheelbase=0.33, max_steer=0.4189, max_dsteer=3.141592653589793,
             max_speed=6.0, min_speed=0.0, max_accel=3.0, q=(13.5, 13.5, 5.5, 13.0), qf=(13.5, 13.5, 5.5, 13.0),
             r=(0.01, 100.0), rd=(0.01, 100.0)):
    """Build a KmpcCfg with the defaults of mpc_config (kinematic_mpc.py:40-68)."""
    cfg = KmpcCfg()
    cfg.horizon = int(horizon)
    cfg.n_rollouts = int(n_rollouts)
    cfg.dt, cfg.wheelbase, cfg.max_steer, cfg.max_dsteer = float(dt), float(wheelbase), float(max_steer), float(max_dsteer)
    cfg.max_speed, cfg.min_speed, cfg.max_accel = float(max_speed), float(min_speed), float(max_accel)
    for i in range(4):
        cfg.q[i] = float(q[i])
        cfg.qf[i] = float(qf[i])
    for i in range(2):
        cfg.r[i] = float(r[i])
        cfg.rd[i] = float(rd[i])
    return cfg


_P = C.c_void_p
_I = C.c_int32
_D = C.c_double

# name -> (restype, argtypes); every symbol include/f1p.h declares
PROTOTYPES = {
    "f1p_lattice_cfg_default": (None, [C.POINTER(LatticeCfg)]),
    "f1p_kmpc_cfg_default": (None, [C.POINTER(KmpcCfg)]),
    "f1p_version": (C.c_char_p, []),
    "f1p_device_count": (C.c_int, []),
    "f1p_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "f1p_destroy": (None, [_P]),
    "f1p_last_error": (C.c_char_p, [_P]),
    "f1p_device_info": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(_I), C.c_char_p, C.c_size_t]),
    "f1p_dev_alloc": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "f1p_dev_free": (C.c_int, [_P, _P]),
    "f1p_host_alloc": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_size_t]),
    "f1p_host_free": (C.c_int, [_P, _P]),
    "f1p_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "f1p_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "f1p_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "f1p_sync": (C.c_int, [_P]),
    "f1p_timer_begin": (C.c_int, [_P]),
    "f1p_timer_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "f1p_set_waypoints": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I]),
    "f1p_set_waypoints_ex": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "f1p_set_grid": (C.c_int, [_P, _P, _I, _I, _D, _D, _D, _I]),
    "f1p_grid_distance_batch": (C.c_int, [_P, _P, _I]),
    "f1p_inflate_grid": (C.c_int, [_P, _D]),
    "f1p_set_footprint": (C.c_int, [_P, _I, _P, _D]),
    "f1p_nearest_point_batch": (C.c_int, [_P, _P, _I, _P, _P, _P, _P]),
    "f1p_intersect_point_batch": (C.c_int, [_P, _P, _P, _I, _D, _I, _P, _P, _P, _P]),
    "f1p_pure_pursuit_batch": (C.c_int, [_P, _P, _I, _D, _D, _D, _P, _P, _P, _P, _P]),
    "f1p_pure_pursuit_dev": (C.c_int, [_P, _P, _I, _D, _D, _D, _P, _P, _P, _P, _P]),
    "f1p_pure_pursuit_set_form": (C.c_int, [_P, _I]),
    "f1p_stanley_batch": (C.c_int, [_P, _P, _I, _D, _D, _P, _P, _P]),
    "f1p_lqr_batch": (C.c_int, [_P, _P, _P, _I, _D, _D, _P, _D, _I, _D, _P, _P, _P]),
    "f1p_lattice_plan_batch": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(LatticeCfg)] + [_P] * 9),
    "f1p_lattice_plan_dev": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(LatticeCfg)] + [_P] * 9),
    "f1p_lattice_plan_batch_f32": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(LatticeCfg)] + [_P] * 7),
    "f1p_lattice_plan_dev_f32": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(LatticeCfg)] + [_P] * 7),
    "f1p_lattice_set_closed_loop": (C.c_int, [_P, _I]),
    "f1p_lattice_closed_loop_state": (C.c_int, [_P, _P, _P, _P]),
    "f1p_lattice_step_batch": (C.c_int, [_P, _P, _I, C.POINTER(LatticeCfg), _P, _P, _P, _I]),
    "f1p_lattice_fetch_traj": (C.c_int, [_P, _P, _I, _I]),
    "f1p_lattice_set_mode": (C.c_int, [_P, _I, _P, _P]),
    "f1p_lattice_set_split": (C.c_int, [_P, _I]),
    "f1p_lattice_set_clearance": (C.c_int, [_P, _I]),
    "f1p_lattice_set_pipeline": (C.c_int, [_P, _I]),
    "f1p_lattice_set_audit": (C.c_int, [_P, _I, _I]),
    "f1p_lattice_audit_read": (C.c_int, [_P, _P, _I]),
    "f1p_lattice_debug_margins": (C.c_int, [_P, _I, C.c_float, C.c_float]),
    "f1p_lattice_debug_bound": (C.c_int, [_P, _P]),
    "f1p_lattice_debug_queue": (C.c_int, [_P, _P, _I]),
    "f1p_lattice_debug_pass": (C.c_int, [_P, _P]),
    "f1p_lattice_set_order": (C.c_int, [_P, _I]),
    "f1p_lattice_profile": (C.c_int, [_P, _I, _P]),
    "f1p_lattice_emit_dev": (C.c_int, [_P, _P, _P, _I, C.POINTER(LatticeCfg), _P, _P, _P, _P, _P, _P, _P]),
    "f1p_clothoid_g1_batch": (C.c_int, [_P, _P, _I, _P, _P, _P, _P]),
    "f1p_clothoid_sample_batch": (C.c_int, [_P, _P, _I, _I, _P]),
    "f1p_kmpc_shoot_batch": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(KmpcCfg), _P, _P, _P, _P, _P]),
    "f1p_kmpc_shoot_dev": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(KmpcCfg), _P, _P, _P, _P, _P]),
    "f1p_kmpc_set_mode": (C.c_int, [_P, _I, _P, _P]),
    "f1p_kmpc_predict_batch": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(KmpcCfg), _P]),
    "f1p_kmpc_ref_batch": (C.c_int, [_P, _P, _I, _I, _D, _D, _P]),
    "f1p_kmpc_sample_controls_dev": (C.c_int, [_P, _P, _I, C.POINTER(KmpcCfg), C.c_uint64, _D, _D]),
    "f1p_kmpc_plan_batch": (C.c_int, [_P, _P, _I, C.POINTER(KmpcCfg), _D, C.POINTER(KmpcSampler), _P, _P, _P, _P, _P]),
    "f1p_kmpc_plan_dev": (C.c_int, [_P, _P, _P, _I, C.POINTER(KmpcCfg), C.POINTER(KmpcSampler), _P, _P, _P, _P, _P]),
    "f1p_kmpc_gen_controls_dev": (C.c_int, [_P, _P, _I, C.POINTER(KmpcCfg), C.POINTER(KmpcSampler)]),
    "f1p_kmpc_warm_reset": (C.c_int, [_P]),
    "f1p_kmpc_warm_get": (C.c_int, [_P, _P, _I, _I]),
    "f1p_kmpc_warm_set": (C.c_int, [_P, _P, _I, _I]),
    "f1p_kmpc_set_groups": (C.c_int, [_P, _I]),
    "f1p_stmpc_set_mode": (C.c_int, [_P, _I, _P, _P]),
    "f1p_kmpc_set_yaw_fixup": (C.c_int, [_P, _I]),
    "f1p_stmpc_cfg_default": (None, [C.POINTER(StmpcCfg)]),
    "f1p_stmpc_predict_batch": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(StmpcCfg), _P]),
    "f1p_stmpc_ref_batch": (C.c_int, [_P, _P, _I, _I, _D, _D, _P]),
    "f1p_stmpc_shoot_batch": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(StmpcCfg), _P, _P, _P, _P, _P]),
    "f1p_stmpc_shoot_dev": (C.c_int, [_P, _P, _P, _P, _I, C.POINTER(StmpcCfg), _P, _P, _P, _P, _P]),
    "f1p_comm_unique_id": (C.c_int, [_P, _P]),
    "f1p_comm_init": (C.c_int, [_P, _P, _I, _I]),
    "f1p_comm_destroy": (C.c_int, [_P]),
    "f1p_comm_info": (C.c_int, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "f1p_comm_argmin_dev": (C.c_int, [_P, _P, _P, _I]),
    "f1p_comm_set_exchange": (C.c_int, [_P, _I]),
    "f1p_argmin_gather_reduce_batch": (C.c_int, [_P, _P, _P, _I, _I, _P, _P]),
    "f1p_argmin_key_batch": (C.c_int, [_P, _P, _I, _P]),
    "f1p_argmin_mask_batch": (C.c_int, [_P, _P, _P, _P, _I, _P, _P]),
}

_lib = None


class F1PLibraryError(RuntimeError):
    pass


def load_library(path=None):
    """Load libf1p.so (once) and attach prototypes.  Raises F1PLibraryError when the file is missing:
    build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C f1tenth_planning_amd/csrc`."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("F1P_LIBRARY", LIB_PATH)
    if not os.path.exists(p):
        raise F1PLibraryError(
            f"libf1p.so not found at {p}: the HIP extension is required (there is no CPU fallback). "
            "Run `make -C f1tenth_planning_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
    mode = os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0)
    # dmabuf IPC for multi-process GPU work (RCCL ranks, shared device memory): the host driver of this pool supports nothing
    # else, and the HSA runtime reads the variable when it is loaded -- so it is defaulted HERE, before the library pulls it in
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        lib = C.CDLL(p, mode=mode)
    except OSError as e:  # pragma: no cover
        raise F1PLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise F1PLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
