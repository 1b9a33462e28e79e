"""ctypes binding of libclothhip.so (include/clothhip.h).

The HIP library is the ONLY compute path of this package: there is no CPU fallback. If the shared object
is missing, or no HIP device is visible, the calls below raise -- loudly -- instead of degrading.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLOTHHIP_LIB: another build of the same library (A/B kernel comparisons); it must export the same C ABI
LIB_PATH = os.environ.get("CLOTHHIP_LIB") or os.path.join(_HERE, "libclothhip.so")

F64, F32 = 0, 1
REST_SHARED, KEEP_TEAR = 1, 2           # clothhip_set_state flags
ABI_VERSION = 7

OK, EINVAL, ENODEV, EHIP, ENOMEM, ESTATE = 0, -1, -2, -3, -4, -5


class ClothHipError(RuntimeError):
    """A HIP runtime failure or a missing device/library (no reference counterpart)."""


class ClothParams(C.Structure):
    _fields_ = [("n_side", C.c_int32), ("frames_per_sec", C.c_int32), ("simulation_steps", C.c_int32),
                ("_pad", C.c_int32), ("width", C.c_double), ("height", C.c_double),
                ("density", C.c_double), ("ks", C.c_double), ("damping", C.c_double),
                ("thickness", C.c_double), ("plane_friction", C.c_double), ("tear_thresh", C.c_double),
                ("gravity", C.c_double), ("minimum_z", C.c_double), ("grip_radius", C.c_double)]


class ClothSchedule(C.Structure):
    _fields_ = [("n_up_end", C.c_int32), ("n_uprest_end", C.c_int32), ("n_pull_end", C.c_int32),
                ("n_griprest_end", C.c_int32), ("n_total", C.c_int32), ("break_on_tear", C.c_int32),
                ("active", C.c_int32), ("_pad", C.c_int32), ("dz_up", C.c_double),
                ("dx_pull", C.c_double), ("dy_pull", C.c_double), ("dz_pull", C.c_double)]


SCHED_DTYPE = np.dtype([("n_up_end", "<i4"), ("n_uprest_end", "<i4"), ("n_pull_end", "<i4"),
                        ("n_griprest_end", "<i4"), ("n_total", "<i4"), ("break_on_tear", "<i4"),
                        ("active", "<i4"), ("_pad", "<i4"), ("dz_up", "<f8"), ("dx_pull", "<f8"),
                        ("dy_pull", "<f8"), ("dz_pull", "<f8")])
assert SCHED_DTYPE.itemsize == C.sizeof(ClothSchedule) == 64



class ClothEpisodeParams(C.Structure):
    _fields_ = [("max_actions", C.c_int32), ("iters_up_rest", C.c_int32), ("iters_grip_rest", C.c_int32),
                ("iters_rest", C.c_int32), ("clip_act_space", C.c_int32), ("force_grab", C.c_int32),
                ("_pad", C.c_int32 * 2), ("iters_up", C.c_double), ("reduce_factor", C.c_double),
                ("grip_radius", C.c_double), ("radius_inc", C.c_double), ("dz_up", C.c_double),
                ("act_low", C.c_double * 4), ("act_high", C.c_double * 4), ("coverage_done", C.c_double)]


class ClothRenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("cam_pos", C.c_float * 3), ("world_to_cam", C.c_float * 9),
                ("lens_mm", C.c_float), ("sensor_mm", C.c_float), ("front", C.c_float * 3), ("back", C.c_float * 3),
                ("background", C.c_float * 3), ("light_dir", C.c_float * 3), ("ambient", C.c_float), ("energy", C.c_float)]


POLICY_TABLE, POLICY_ORACLE_CORNER, POLICY_HIGHEST_POINT = 0, 1, 2
MT_WORDS = 626                      # per-env RandomState image: key[624], pos, pad (csrc/cloth_rng.hpp)

RESET_PULL_DTYPE = np.dtype([("point", "<i4"), ("need_coverage", "<i4"), ("x", "<f8"), ("y", "<f8"), ("dx", "<f8"),
                             ("dy", "<f8"), ("iters_up", "<f8"), ("coverage_min", "<f8")])
RESET_SCRIPT_DTYPE = np.dtype([("valid", "<i4"), ("n_pulls", "<i4"), ("settle_after", "<i4"), ("_pad", "<i4"),
                               ("pull", RESET_PULL_DTYPE, (3,))])
STEP_RECORD_DTYPE = np.dtype([("action", "<f8", (4,)), ("coverage", "<f8"), ("variance_inv", "<f8"),
                              ("executed", "<i4"), ("n_grabbed", "<i4"), ("iters_pull", "<i4"),
                              ("n_below_half_thickness", "<i4"), ("ran", "u1"), ("oob", "u1"), ("tear", "u1"),
                              ("done", "u1"), ("reset_before", "u1"), ("_pad", "u1", (3,))])
RESET_RECORD_DTYPE = np.dtype([("consumed", "<i4"), ("pulls_run", "<i4"), ("executed", "<i4", (3,)),
                               ("settle_executed", "<i4"), ("tear", "<i4"), ("init_side", "<i4"),
                               ("start_coverage", "<f8"), ("start_variance_inv", "<f8"), ("action", "<f8", (3, 4))])
assert RESET_PULL_DTYPE.itemsize == 56 and RESET_SCRIPT_DTYPE.itemsize == 184
assert STEP_RECORD_DTYPE.itemsize == 72 and RESET_RECORD_DTYPE.itemsize == 144
assert C.sizeof(ClothEpisodeParams) == 144

# every symbol include/clothhip.h declares: (name, restype, argtypes)
_vp, _dp, _u8p, _i32p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
_PP = C.POINTER(ClothParams)
SYMBOLS = [
    ("clothhip_last_error", C.c_char_p, []),
    ("clothhip_abi_version", C.c_int, []),
    ("clothhip_device_count", C.c_int, []),
    ("clothhip_create", C.c_int, [_PP, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    ("clothhip_destroy", C.c_int, [_vp]),
    ("clothhip_num_points", C.c_int, [_vp]),
    ("clothhip_num_springs", C.c_int, [_vp]),
    ("clothhip_num_envs", C.c_int, [_vp]),
    ("clothhip_precision", C.c_int, [_vp]),
    ("clothhip_init_grid", C.c_int, [_PP, C.c_int32, C.c_int32, _dp, _dp, _dp]),
    ("clothhip_spring_topology", C.c_int, [_PP, _i32p, _i32p, _u8p]),
    ("clothhip_set_state", C.c_int, [_vp, C.c_int32, C.c_int32, _dp, _dp, _u8p, _dp, C.c_int32]),
    ("clothhip_get_state", C.c_int, [_vp, C.c_int32, C.c_int32, _dp, _dp, _u8p]),
    ("clothhip_get_rest", C.c_int, [_vp, C.c_int32, C.c_int32, _dp]),
    ("clothhip_reset_flat", C.c_int, [_vp, _u8p]),
    ("clothhip_get_tear", C.c_int, [_vp, _u8p]),
    ("clothhip_set_tear", C.c_int, [_vp, _u8p]),
    ("clothhip_grab_top", C.c_int, [_vp, _dp, _dp, _u8p, _i32p]),
    ("clothhip_grab", C.c_int, [_vp, _dp, _dp, _u8p, _i32p]),
    ("clothhip_release", C.c_int, [_vp, _u8p]),
    ("clothhip_pin_points", C.c_int, [_vp, C.c_int32, _i32p, C.c_int32]),
    ("clothhip_run", C.c_int, [_vp, _vp, _i32p]),
    ("clothhip_run_async", C.c_int, [_vp, _vp]),
    ("clothhip_sync", C.c_int, [_vp, _i32p]),
    ("clothhip_fused_supported", C.c_int, [_vp]),
    ("clothhip_run_actions_begin", C.c_int, [_vp, C.POINTER(ClothEpisodeParams), C.c_int32, C.c_int32, _vp, C.c_int32, _i32p,
                                             _vp, C.c_int32, _i32p, _u8p, _vp, C.c_int32, C.c_uint64, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_double]),
    ("clothhip_run_actions_end", C.c_int, [_vp, _i32p, _u8p, _vp, _vp, _vp, _vp, _vp]),
    ("clothhip_run_actions_op_ticks", C.c_int, [_vp, _vp]),
    ("clothhip_run_actions_summary", C.c_int, [_vp, _dp, C.POINTER(_vp)]),
    ("clothhip_run_actions", C.c_int, [_vp, C.POINTER(ClothEpisodeParams), C.c_int32, C.c_int32, _vp, C.c_int32, _i32p, _vp,
                                       C.c_int32, _i32p, _u8p, _vp, _vp, _vp, _vp, C.c_double]),
    ("clothhip_update", C.c_int, [_vp, C.c_int32, _dp]),
    ("clothhip_metrics", C.c_int, [_vp, _dp, _dp, _u8p, _u8p]),
    ("clothhip_metrics_ex", C.c_int, [_vp, _dp, _dp, _u8p, _u8p, _i32p]),
    ("clothhip_hull_area", C.c_double, [_dp, C.c_int32]),
    ("clothhip_write_obs_f32_device", C.c_int, [_vp, _vp]),
    ("clothhip_run_device_sched_async", C.c_int, [_vp, _vp]),
    ("clothhip_render", C.c_int, [_vp, C.POINTER(ClothRenderParams), _u8p, _u8p, C.POINTER(C.c_float)]),
    ("clothhip_device_alloc", C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    ("clothhip_device_free", C.c_int, [_vp, _vp]),
    ("clothhip_device_upload", C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    ("clothhip_device_download", C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    ("clothhip_stream", _vp, [_vp]),
    ("clothhip_last_kernel_ms", C.c_double, [_vp]),
    ("clothhip_debug_stats", C.c_int, [_vp, _i32p]),
    ("clothhip_last_variant", C.c_int, [_vp, _i32p]),
    ("clothhip_last_dispatches", C.c_int, [_vp, _i32p]),
    ("clothhip_last_specialised", C.c_int, [_vp, _i32p]),
    ("clothhip_set_relaxed_order", C.c_int, [_vp, C.c_int32]),
    ("clothhip_selftest_windows", C.c_int, [_PP, _i32p, _i32p, _i32p, _i32p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_int32]),
    ("clothhip_selftest_layout", C.c_int, [_PP, C.c_int32, C.c_int32, C.c_int32, _i32p, C.c_int32]),
    ("clothhip_selftest_rng", C.c_int, [_vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, _dp]),
    ("clothhip_selftest_arith", C.c_int, [C.c_int32, C.c_int32, _dp, _dp, _dp, C.c_int64]),
]

_lib = None


def load():
    """dlopen libclothhip.so and bind every declared symbol. Raises ClothHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClothHipError(
            "libclothhip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gym_cloth_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # ROCr keeps a per-queue scratch pool and serves a kernel whose scratch at full occupancy exceeds HSA_SCRATCH_SINGLE_LIMIT
    # (default 140 MB) with a throttled wave count -- and, measured on MI355X / ROCm 7.2, a process that has once run such a kernel
    # may then run a LATER stepper variant below its residency too (two 50x50 cloths per CU became 1.33: 5.0 -> 3.3 M substeps/s).
    # The stepper variants with a VGPR cap spill a few registers in their cold episode code, so they all own some scratch. Unless
    # the caller has set the limit, raise it (512 MB) before the HIP runtime comes up; a runtime that is already up ignores this.
    os.environ.setdefault("HSA_SCRATCH_SINGLE_LIMIT", str(512 << 20))
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ClothHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if L.clothhip_abi_version() != ABI_VERSION:
        raise ClothHipError("libclothhip ABI version %d, expected %d" % (L.clothhip_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc):
    """Map a negative status to the Python exception the reference would raise for that condition."""
    if rc >= 0:
        return rc
    msg = load().clothhip_last_error().decode("utf8", "replace")
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise ClothHipError("[%d] %s" % (rc, msg))


def dp(a):
    return None if a is None else a.ctypes.data_as(_dp)


def u8p(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def i32p(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def params_from_cfg(cfg, gravity=-9.8, minimum_z=0.0):
    """cfg: the dict ClothEnv loads from cfg/*.yaml (cloth_env.py:87-118, cloth.pyx:53-56,175-186)."""
    c = cfg["cloth"]
    if c["num_width_points"] != c["num_height_points"]:
        raise AssertionError("height == width (cloth.pyx:91)")
    p = ClothParams()
    p.n_side = int(c["num_width_points"])
    p.frames_per_sec = int(cfg["frames_per_sec"])
    p.simulation_steps = int(cfg["simulation_steps"])
    p.width = float(c["width"]); p.height = float(c["height"])
    p.density = float(c["density"]); p.ks = float(c["ks"]); p.damping = float(c["damping"])
    p.thickness = float(c["thickness"]); p.plane_friction = float(c["plane_friction"])
    p.tear_thresh = float(c["tear_thresh"])
    p.gravity = float(gravity); p.minimum_z = float(minimum_z)
    p.grip_radius = float(cfg.get("env", {}).get("grip_radius", 0.003))
    return p
