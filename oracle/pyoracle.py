"""ctypes binding of the CPU oracle (oracle/cloth_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module. The product path (gym_cloth_amd) never does.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_cloth.so")


class OracleParams(C.Structure):
    _fields_ = [("n_side", C.c_int32), ("frames_per_sec", C.c_int32),
                ("simulation_steps", C.c_int32), ("_pad", C.c_int32),
                ("width", C.c_double), ("height", C.c_double),
                ("density", C.c_double), ("ks", C.c_double), ("damping", C.c_double),
                ("thickness", C.c_double), ("plane_friction", C.c_double),
                ("tear_thresh", C.c_double), ("gravity", C.c_double), ("minimum_z", C.c_double)]


def build(force=False):
    src = os.path.join(_HERE, "cloth_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src),
                                              os.path.getmtime(os.path.join(_HERE, "cloth_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_cloth.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, dp, u8p, i32p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [C.POINTER(OracleParams)]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_num_points.argtypes = [vp]
        L.oracle_num_springs.argtypes = [vp]
        L.oracle_init_grid.argtypes = [vp, C.c_int, C.c_int, dp]
        L.oracle_set_state.argtypes = [vp, dp, dp, u8p, dp]
        L.oracle_get_state.argtypes = [vp, dp, dp, u8p]
        L.oracle_get_rest.argtypes = [vp, dp]
        L.oracle_get_springs.argtypes = [vp, i32p, i32p, u8p]
        L.oracle_pin.argtypes = [vp, C.c_int]
        L.oracle_update.argtypes = [vp, C.c_int]
        L.oracle_grab_top.argtypes = [vp, C.c_double, C.c_double, C.c_double]
        L.oracle_grab.argtypes = [vp, C.c_double, C.c_double, C.c_double]
        L.oracle_adjust.argtypes = [vp, C.c_double, C.c_double, C.c_double]
        L.oracle_release.argtypes = [vp]
        L.oracle_num_grabbed.argtypes = [vp]
        L.oracle_get_grabbed.argtypes = [vp, i32p]
        L.oracle_have_tear.argtypes = [vp]
        L.oracle_set_tear.argtypes = [vp, C.c_int]
        L.oracle_cell_census.argtypes = [vp, i32p, i32p]
        L.oracle_last_stats.argtypes = [vp, i32p, i32p]
        L.oracle_run_schedule.argtypes = [vp] + [C.c_int] * 5 + [C.c_double] * 3 + [C.c_int]
        L.oracle_batch_run_schedule.argtypes = [C.POINTER(vp), C.c_int, i32p, dp, C.c_int, i32p, C.c_int]
        L.oracle_batch_update.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int]
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def params_from_dict(d):
    """d: the JSON 'cfg' record stored in every golden fixture (or an equivalent dict)."""
    p = OracleParams()
    p.n_side = int(d["n_side"]); p.frames_per_sec = int(d["frames_per_sec"])
    p.simulation_steps = int(d["simulation_steps"])
    for k in ("width", "height", "density", "ks", "damping", "thickness", "plane_friction",
              "tear_thresh", "gravity", "minimum_z"):
        setattr(p, k, float(d[k]))
    return p


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


class OracleCloth(object):
    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self._p = params_from_dict(cfg)
        self._L = lib()
        self._h = self._L.oracle_create(C.byref(self._p))
        if not self._h:
            raise ValueError("oracle_create failed")
        self.P = self._L.oracle_num_points(self._h)
        self.S = self._L.oracle_num_springs(self._h)
        self.grip_radius = float(cfg.get("grip_radius", 0.003))

    def __del__(self):
        try:
            if self._h:
                self._L.oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def init_grid(self, tier=1, init_side=False, rand_draws=None):
        rd = None if rand_draws is None else np.ascontiguousarray(rand_draws, dtype=np.float64)
        rc = self._L.oracle_init_grid(self._h, int(tier), int(bool(init_side)), _dp(rd))
        if rc != 0:
            raise ValueError("oracle_init_grid rc=%d" % rc)

    def set_state(self, pos=None, prev=None, pinned=None, rest=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        pos, prev, rest = f(pos), f(prev), f(rest)
        pin = None if pinned is None else np.ascontiguousarray(pinned, dtype=np.uint8)
        self._L.oracle_set_state(self._h, _dp(pos), _dp(prev), _u8p(pin), _dp(rest))

    def get_state(self):
        pos = np.empty((self.P, 3)); prev = np.empty((self.P, 3)); pin = np.empty(self.P, dtype=np.uint8)
        self._L.oracle_get_state(self._h, _dp(pos), _dp(prev), _u8p(pin))
        return pos, prev, pin

    @property
    def rest(self):
        r = np.empty(self.S)
        self._L.oracle_get_rest(self._h, _dp(r))
        return r

    def springs(self):
        a = np.empty(self.S, dtype=np.int32); b = np.empty(self.S, dtype=np.int32)
        t = np.empty(self.S, dtype=np.uint8)
        self._L.oracle_get_springs(self._h, _i32p(a), _i32p(b), _u8p(t))
        return a, b, t

    def pin(self, i):
        self._L.oracle_pin(self._h, int(i))

    def update(self, n=1):
        self._L.oracle_update(self._h, int(n))

    def grab_top(self, x, y, radius=None):
        return self._L.oracle_grab_top(self._h, x, y, self.grip_radius if radius is None else radius)

    def grab(self, x, y, radius=None):
        return self._L.oracle_grab(self._h, x, y, self.grip_radius if radius is None else radius)

    def adjust(self, dx, dy, dz):
        self._L.oracle_adjust(self._h, dx, dy, dz)

    def release(self):
        self._L.oracle_release(self._h)

    @property
    def grabbed(self):
        n = self._L.oracle_num_grabbed(self._h)
        a = np.empty(n, dtype=np.int32)
        if n:
            self._L.oracle_get_grabbed(self._h, _i32p(a))
        return a

    @property
    def have_tear(self):
        return bool(self._L.oracle_have_tear(self._h))

    @have_tear.setter
    def have_tear(self, v):
        self._L.oracle_set_tear(self._h, int(bool(v)))

    def cell_census(self):
        a = C.c_int32(); b = C.c_int32()
        self._L.oracle_cell_census(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def last_stats(self):
        a = C.c_int32(); b = C.c_int32()
        self._L.oracle_last_stats(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def run_schedule(self, sched, dz_up, dx, dy, break_on_tear=True):
        s = [int(v) for v in sched]
        return self._L.oracle_run_schedule(self._h, s[0], s[1], s[2], s[3], s[4], dz_up, dx, dy,
                                           int(break_on_tear))


def batch_run_schedule(cloths, sched, delta, break_on_tear=True, n_threads=0):
    n = len(cloths)
    hs = (C.c_void_p * n)(*[c.handle for c in cloths])
    sched = np.ascontiguousarray(sched, dtype=np.int32).reshape(n, 5)
    delta = np.ascontiguousarray(delta, dtype=np.float64).reshape(n, 3)
    out = np.zeros(n, dtype=np.int32)
    lib().oracle_batch_run_schedule(hs, n, _i32p(sched), _dp(delta), int(break_on_tear), _i32p(out),
                                    int(n_threads))
    return out


def batch_update(cloths, n_sub, n_threads=0):
    n = len(cloths)
    hs = (C.c_void_p * n)(*[c.handle for c in cloths])
    lib().oracle_batch_update(hs, n, int(n_sub), int(n_threads))


def replay_ops(cloth, ops, on_checkpoint=None, start=0, stop=None):
    """Replay a golden 'ops' list (tests/golden/make_golden.py::Trace) on any object exposing
    grab_top/grab/release/pin/update/adjust (OracleCloth here; the HIP façade in the gpu tests)."""
    k = 0
    for op in ops[start:stop]:
        name = op[0]
        if name == "checkpoint":
            if on_checkpoint is not None:
                on_checkpoint(k)
            k += 1
        elif name == "grab_top":
            cloth.grab_top(op[1], op[2])
        elif name == "grab":
            cloth.grab(op[1], op[2])
        elif name == "release":
            cloth.release()
        elif name == "pin":
            cloth.pin(op[1])
        elif name == "update":
            cloth.update(op[1])
        elif name == "adjust_update":
            n = op[4]
            if hasattr(cloth, "adjust_update"):
                cloth.adjust_update(op[1], op[2], op[3], n)
            else:
                for _ in range(n):
                    cloth.adjust(op[1], op[2], op[3])
                    cloth.update(1)
        else:
            raise ValueError(name)


def load_golden(name):
    here = os.path.join(os.path.dirname(_HERE), "tests", "golden")
    d = np.load(os.path.join(here, name), allow_pickle=False)
    out = {k: d[k] for k in d.files}
    for k in ("cfg", "ops", "grabbed", "info", "grab_top", "grab"):
        if k in out:
            out[k] = json.loads(str(out[k]))
    return out
