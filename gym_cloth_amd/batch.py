"""ClothBatch: E independent cloths resident on one MI355X, stepped by libclothhip.

This is the batched counterpart of the reference's `Cloth` + `Gripper` pair (gym_cloth/physics/cloth.pyx:21,
gripper.pyx:8); the single-cloth façade with the reference's attribute names lives in physics.py.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import F32, F64, SCHED_DTYPE, check


def make_schedules(n, **fields):
    """A zeroed ClothSchedule[n] numpy record array with `fields` broadcast in."""
    s = np.zeros(n, dtype=SCHED_DTYPE)
    for k, v in fields.items():
        s[k] = v
    return s


def schedule_bounds(iters_up, iters_up_rest, iters_pull, iters_grip_rest, iters_rest):
    """Integer phase boundaries of ClothEnv.step (cloth_env.py:472-475) / _pull (:352-367).

    The reference compares an int `i` with cumulative sums that may be floats (tier 3 draws a float
    iters_up, cloth_env.py:960); `i < b` for integer i is `i < ceil(b)`, so the ceilings are exact.
    The sums are formed left to right exactly as the reference writes them.
    """
    b1 = iters_up
    b2 = iters_up + iters_up_rest
    b3 = iters_up + iters_up_rest + iters_pull
    b4 = iters_up + iters_up_rest + iters_pull + iters_grip_rest
    b5 = iters_up + iters_up_rest + iters_pull + iters_grip_rest + iters_rest
    return tuple(int(np.ceil(b)) for b in (b1, b2, b3, b4, b5))


class ClothBatch(object):
    def __init__(self, cfg, n_envs=1, device=0, precision="f32", gravity=-9.8, minimum_z=0.0):
        self._L = _lib.load()
        self.cfg = cfg
        self.params = _lib.params_from_cfg(cfg, gravity=gravity, minimum_z=minimum_z)
        self.precision = {"f64": F64, "f32": F32, F64: F64, F32: F32}[precision]
        h = C.c_void_p()
        check(self._L.clothhip_create(C.byref(self.params), int(n_envs), int(device), self.precision, C.byref(h)))
        self._h = h
        self.E = int(n_envs)
        self.P = self._L.clothhip_num_points(h)
        self.S = self._L.clothhip_num_springs(h)
        self.N = self.params.n_side
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.clothhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- construction helpers (host, double) ---------------------------------------------------------
    def init_grid(self, tier=1, init_side=False, rand_draws=None):
        """(pos[P,3], rest[S]) of a freshly constructed Cloth (cloth.pyx:92-146, :411-417)."""
        pos = np.empty((self.P, 3)); rest = np.empty(self.S)
        rd = None if rand_draws is None else np.ascontiguousarray(rand_draws, dtype=np.float64)
        check(self._L.clothhip_init_grid(C.byref(self.params), int(tier), int(bool(init_side)),
                                         _lib.dp(rd), _lib.dp(pos), _lib.dp(rest)))
        return pos, rest

    def topology(self):
        a = np.empty(self.S, dtype=np.int32); b = np.empty(self.S, dtype=np.int32)
        t = np.empty(self.S, dtype=np.uint8)
        check(self._L.clothhip_spring_topology(C.byref(self.params), _lib.i32p(a), _lib.i32p(b), _lib.u8p(t)))
        return a, b, t

    # ---- state -----------------------------------------------------------------------------------------
    def set_state(self, pos=None, prev=None, pinned=None, rest=None, env0=0, n=None, rest_shared=None,
                  keep_tear=False):
        """Upload state of envs [env0, env0+n). keep_tear: a write into a live cloth (the reference's tear flag is
        sticky, cloth.pyx:272-273); without it the call is the Cloth(...) rebuild of a reset and clears the flag."""
        n = self.E - env0 if n is None else n
        f = lambda a, shp: None if a is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(a, dtype=np.float64), shp))
        pos = f(pos, (n, self.P, 3)); prev = f(prev, (n, self.P, 3))
        pin = None if pinned is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(pinned, dtype=np.uint8), (n, self.P)))
        if rest is not None:
            rest = np.asarray(rest, dtype=np.float64)
            if rest_shared is None:
                rest_shared = rest.ndim == 1
            rest = np.ascontiguousarray(rest if rest_shared else np.broadcast_to(rest, (n, self.S)))
        flags = (_lib.REST_SHARED if rest_shared else 0) | (_lib.KEEP_TEAR if keep_tear else 0)
        check(self._L.clothhip_set_state(self._h, env0, n, _lib.dp(pos), _lib.dp(prev), _lib.u8p(pin),
                                         _lib.dp(rest), flags))

    def get_rest(self, env0=0, n=None):
        """Spring.rest_length per env in reference list order, [n, S]."""
        n = self.E - env0 if n is None else n
        rest = np.empty((n, self.S))
        check(self._L.clothhip_get_rest(self._h, env0, n, _lib.dp(rest)))
        return rest

    def ensure_per_env_rest(self):
        """Give every env its own rest-length table (what tier 2 needs: cloth.pyx:417 measures them on the noisy sheet);
        a no-op once done. The tables start as copies of the current ones."""
        if not getattr(self, "_per_env_rest", False):
            self.set_state(rest=self.get_rest(), rest_shared=False)
            self._per_env_rest = True

    def reset_flat(self, mask=None):
        """Cloth(...) rebuild of the flat tiers (1, 3) on the device for the masked envs (None = all)."""
        m = None if mask is None else np.ascontiguousarray(np.broadcast_to(np.asarray(mask, dtype=np.uint8), (self.E,)))
        check(self._L.clothhip_reset_flat(self._h, _lib.u8p(m)))

    def get_state(self, env0=0, n=None, want_prev=True, want_pinned=True):
        n = self.E - env0 if n is None else n
        pos = np.empty((n, self.P, 3))
        prev = np.empty((n, self.P, 3)) if want_prev else None
        pin = np.empty((n, self.P), dtype=np.uint8) if want_pinned else None
        check(self._L.clothhip_get_state(self._h, env0, n, _lib.dp(pos), _lib.dp(prev), _lib.u8p(pin)))
        return pos, prev, pin

    def positions(self, env0=0, n=None):
        return self.get_state(env0, n, want_prev=False, want_pinned=False)[0]

    @property
    def tear(self):
        t = np.empty(self.E, dtype=np.uint8)
        check(self._L.clothhip_get_tear(self._h, _lib.u8p(t)))
        return t.astype(bool)

    @tear.setter
    def tear(self, v):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.uint8), (self.E,)))
        check(self._L.clothhip_set_tear(self._h, _lib.u8p(t)))

    def metrics(self, want_height=False):
        """(coverage[E], variance_inv[E], out_of_bounds[E], tear[E]) as ClothEnv computes them
        (cloth_env.py:1020-1098); with want_height also the 'height' reward's fraction of points with
        z < thickness/2 (cloth_env.py:603-609)."""
        cov = np.empty(self.E); vinv = np.empty(self.E)
        oob = np.empty(self.E, dtype=np.uint8); tear = np.empty(self.E, dtype=np.uint8)
        nlow = np.empty(self.E, dtype=np.int32) if want_height else None
        check(self._L.clothhip_metrics_ex(self._h, _lib.dp(cov), _lib.dp(vinv), _lib.u8p(oob), _lib.u8p(tear),
                                          _lib.i32p(nlow)))
        if want_height:
            return cov, vinv, oob.astype(bool), tear.astype(bool), nlow / float(self.P)
        return cov, vinv, oob.astype(bool), tear.astype(bool)

    # ---- headless rendering ---------------------------------------------------------------
    @staticmethod
    def camera_matrix(cam_deg=(0.0, 0.0, 0.0)):
        """world -> camera rotation for Blender's `rotation_euler` (XYZ order, degrees) of the reference's camera
        (get_image_rep_279.py:119-122): camera-to-world = Rz Ry Rx, returned transposed as float32[9]."""
        ax, ay, az = np.deg2rad(np.asarray(cam_deg, dtype=np.float64))
        cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        m = np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                      [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                      [-sy, cy * sx, cy * cx]])
        return np.ascontiguousarray(m.T.reshape(9), dtype=np.float32)

    RENDER_DEFAULTS = dict(width=224, height=224, cam_pos=(0.5, 0.5, 1.45), cam_deg=(0.0, 0.0, 0.0), lens_mm=40.0,
                           sensor_mm=36.0, front=(0.070, 0.050, 0.600), back=(0.070, 0.300, 0.900),
                           background=(1.0, 1.0, 1.0), light_dir=(0.5169, 0.0730, 0.8530), ambient=0.05, energy=1.5)

    def render_params(self, **kw):
        """_lib.ClothRenderParams of the reference's scene (get_image_rep_279.py: camera :114-122, lens :273-276, side colours
        :249-257, bed :172, lamp energy :450; light_dir = the direction from the cloth centre to Blender's default lamp);
        any field can be overridden, cam_deg instead of world_to_cam."""
        d = dict(self.RENDER_DEFAULTS); d.update(kw)
        p = _lib.ClothRenderParams()
        p.width, p.height = int(d["width"]), int(d["height"])
        w2c = d["world_to_cam"] if "world_to_cam" in d else self.camera_matrix(d["cam_deg"])
        for k in range(9):
            p.world_to_cam[k] = float(w2c[k])
        for name in ("cam_pos", "front", "back", "background", "light_dir"):
            for k in range(3):
                getattr(p, name)[k] = float(d[name][k])
        p.lens_mm, p.sensor_mm, p.ambient, p.energy = float(d["lens_mm"]), float(d["sensor_mm"]), float(d["ambient"]), float(d["energy"])
        return p

    def render(self, want_rgb=True, want_depth=True, swap_sides=None, params=None, **kw):
        """Rasterise every env's cloth mesh (cloth_env.py:212-330 without Blender): (rgb uint8 [E, H, W, 3] or None,
        depth float32 [E, H, W] camera-space distance or None)."""
        p = params if params is not None else self.render_params(**kw)
        rgb = np.empty((self.E, p.height, p.width, 3), dtype=np.uint8) if want_rgb else None
        dep = np.empty((self.E, p.height, p.width), dtype=np.float32) if want_depth else None
        sw = None if swap_sides is None else np.ascontiguousarray(np.broadcast_to(np.asarray(swap_sides, dtype=np.uint8), (self.E,)))
        check(self._L.clothhip_render(self._h, C.byref(p), _lib.u8p(sw), _lib.u8p(rgb),
                                      None if dep is None else dep.ctypes.data_as(C.POINTER(C.c_float))))
        return rgb, dep

    # ---- gripper ---------------------------------------------------------------------------------------
    def _grab(self, fn, xy, radius, active):
        xy = np.ascontiguousarray(np.broadcast_to(np.asarray(xy, dtype=np.float64), (self.E, 2)))
        rad = None if radius is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(radius, dtype=np.float64), (self.E,)))
        act = None if active is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(active, dtype=np.uint8), (self.E,)))
        n = np.zeros(self.E, dtype=np.int32)
        check(fn(self._h, _lib.dp(xy), _lib.dp(rad), _lib.u8p(act), _lib.i32p(n)))
        return n

    def grab_top(self, xy, radius=None, active=None):
        return self._grab(self._L.clothhip_grab_top, xy, radius, active)

    def grab(self, xy, radius=None, active=None):
        return self._grab(self._L.clothhip_grab, xy, radius, active)

    def release(self, active=None):
        act = None if active is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(active, dtype=np.uint8), (self.E,)))
        check(self._L.clothhip_release(self._h, _lib.u8p(act)))

    def pin_points(self, env, idx):
        idx = np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int32)
        check(self._L.clothhip_pin_points(self._h, int(env), _lib.i32p(idx), len(idx)))

    # ---- stepping --------------------------------------------------------------------------------------
    def run(self, sched):
        """Run ClothSchedule[E] (see include/clothhip.h); returns executed update() counts [E]."""
        sched = np.ascontiguousarray(sched, dtype=SCHED_DTYPE)
        if sched.shape != (self.E,):
            raise ValueError("sched must have shape (%d,)" % self.E)
        ex = np.zeros(self.E, dtype=np.int32)
        check(self._L.clothhip_run(self._h, sched.ctypes.data_as(C.c_void_p), _lib.i32p(ex)))
        return ex

    def run_async(self, sched):
        sched = np.ascontiguousarray(sched, dtype=SCHED_DTYPE)
        if sched.shape != (self.E,):
            raise ValueError("sched must have shape (%d,)" % self.E)
        check(self._L.clothhip_run_async(self._h, sched.ctypes.data_as(C.c_void_p)))

    def sync(self, want_executed=True):
        ex = np.zeros(self.E, dtype=np.int32) if want_executed else None
        check(self._L.clothhip_sync(self._h, _lib.i32p(ex)))
        return ex

    @property
    def fused_supported(self):
        return bool(check(self._L.clothhip_fused_supported(self._h)))

    def run_actions_begin(self, ep, n_actions, num_steps, done, actions=None, policy=None, policy_arg=None, scripts=None,
                          want_resets=True, want_obs=False, actions_device_ptr=None, time_budget_ms=0.0,
                          rng_states=None, rng_tier=0, domrand_words=0, reset_capacity=0):
        """First half of clothhip_run_actions (see include/clothhip.h): upload + launch, returns while the kernel runs.
        ep: _lib.ClothEpisodeParams; scripts: RESET_SCRIPT_DTYPE[E, R], each env's next R resets in order."""
        T = int(n_actions)
        pol = _lib.POLICY_TABLE if policy is None else int(policy)
        on_dev = 0
        ap = None
        if pol == _lib.POLICY_TABLE:
            if actions_device_ptr is not None:
                ap, on_dev = C.c_void_p(int(actions_device_ptr)), 1
            else:
                actions = np.ascontiguousarray(actions, dtype=np.float64)
                if actions.shape != (T, self.E, 4):
                    raise ValueError("actions must have shape (%d, %d, 4)" % (T, self.E))
                ap = actions.ctypes.data_as(C.c_void_p)
        assert num_steps.dtype == np.int32 and num_steps.shape == (self.E,) and num_steps.flags['C_CONTIGUOUS']
        assert done.dtype == np.uint8 and done.shape == (self.E,) and done.flags['C_CONTIGUOUS']
        parg = None if policy_arg is None else np.ascontiguousarray(policy_arg, dtype=np.int32)
        if pol == _lib.POLICY_HIGHEST_POINT and (parg is None or parg.shape != (1 + T, self.E)):
            raise ValueError("the highest-point policy needs policy_arg int32[1 + %d, %d]: construction codes, then which of "
                             "the highest points every env pulls in every slot" % (T, self.E))
        if pol != _lib.POLICY_HIGHEST_POINT and parg is not None and parg.shape != (self.E,):
            raise ValueError("policy_arg must have shape (%d,)" % self.E)
        if scripts is not None:
            scripts = np.ascontiguousarray(scripts, dtype=_lib.RESET_SCRIPT_DTYPE)
            if scripts.ndim != 2 or scripts.shape[0] != self.E:
                raise ValueError("scripts must have shape (%d, R)" % self.E)
        if rng_states is not None:                       # resets drawn on the device from the envs' numpy streams
            assert scripts is None and rng_states.dtype == np.uint32 and rng_states.shape == (self.E, _lib.MT_WORDS)
            assert rng_states.flags['C_CONTIGUOUS']
        have_src = scripts is not None or rng_states is not None
        R = scripts.shape[1] if scripts is not None else (int(reset_capacity) if rng_states is not None else 0)
        have_rst = bool(want_resets and have_src)
        have_robs = bool(want_obs and have_src)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        check(self._L.clothhip_run_actions_begin(self._h, C.byref(ep), T, pol, ap, on_dev, _lib.i32p(parg), vp(scripts), R,
                                                 _lib.i32p(num_steps), _lib.u8p(done), vp(rng_states), int(rng_tier),
                                                 int(domrand_words), int(have_rst), int(bool(want_obs)), int(have_robs),
                                                 float(time_budget_ms)))
        self._fused = (T, R, num_steps, done, have_rst, bool(want_obs), have_robs, rng_states)

    def run_actions_end(self):
        """Second half: wait for the launch and fetch its outputs. num_steps / done given to _begin are updated in place.
        rng_states given to _begin (uint32[E, 626]) now hold the advanced streams.
        Returns (records[T, E], resets[E, R] or None, obs float32[T, E, 3P] or None, reset_obs float32[E, R, 3P] or None:
        the first observation of every episode started inside the launch)."""
        T, R, num_steps, done, have_rst, have_obs, have_robs, rng_states = self._fused
        self._fused = None
        rec = np.zeros((T, self.E), dtype=_lib.STEP_RECORD_DTYPE)
        rst = np.zeros((self.E, R), dtype=_lib.RESET_RECORD_DTYPE) if have_rst else None
        obs = np.empty((T, self.E, 3 * self.P), dtype=np.float32) if have_obs else None
        robs = np.empty((self.E, R, 3 * self.P), dtype=np.float32) if have_robs else None
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        check(self._L.clothhip_run_actions_end(self._h, _lib.i32p(num_steps), _lib.u8p(done), vp(rec), vp(rst), vp(obs),
                                               vp(robs), vp(rng_states)))
        return rec, rst, obs, robs

    def run_summary(self):
        """float64[E, 4] written by the last episode launch: actions executed, episode over, coverage after the env's last
        action / reset of the launch (NaN: none), update() calls of its actions (clothhip_run_actions_summary)."""
        out = np.zeros((self.E, 4))
        check(self._L.clothhip_run_actions_summary(self._h, _lib.dp(out), None))
        return out

    @property
    def run_summary_device_ptr(self):
        """Device address of that table (for an in-place all-gather on the handle's stream)."""
        p = C.c_void_p()
        check(self._L.clothhip_run_actions_summary(self._h, None, C.byref(p)))
        return p.value

    def op_ticks(self):
        """Where the time of the last episode launch went, per env (clothhip_run_actions_op_ticks): (ticks uint64[E, 4] in 100 MHz
        ticks, substeps uint64[E, 4]) for the classes {actions, reset pulls, reset settling, the rest}."""
        raw = np.zeros((self.E, 8), dtype=np.uint64)
        check(self._L.clothhip_run_actions_op_ticks(self._h, raw.ctypes.data_as(C.c_void_p)))
        return raw[:, :4].copy(), raw[:, 4:].copy()

    def run_actions(self, *a, **k):
        """clothhip_run_actions: `n_actions` whole ClothEnv.step calls per env in one launch (begin + end)."""
        self.run_actions_begin(*a, **k)
        return self.run_actions_end()

    def update(self, n=1, delta=None):
        """n x Cloth.update() (cloth.pyx:169), each preceded by Gripper.adjust(*delta) if delta is given."""
        d = None if delta is None else np.ascontiguousarray(delta, dtype=np.float64)
        check(self._L.clothhip_update(self._h, int(n), _lib.dp(d)))

    def debug_stats(self):
        """[E,16]: [0..3] strain sweeps run, windows walked, passes, passes that corrected (last run);
        [4..15] per-phase shader cycles/64 when CLOTHHIP_DEBUG_PHASES has bit 32 set."""
        st = np.zeros((self.E, 16), dtype=np.int32)
        check(self._L.clothhip_debug_stats(self._h, _lib.i32p(st)))
        return st

    def last_variant(self):
        """Which compiled stepper variant the last launch of this handle ran (clothhip_last_variant): a dict with the template
        parameters, whether the LEAN arithmetic ran, LDS bytes per cloth, resident cloths per CU and the device's CU count, plus a
        one-line `name`."""
        v = np.zeros(10, dtype=np.int32)
        check(self._L.clothhip_last_variant(self._h, _lib.i32p(v)))
        d = dict(threads=int(v[0]), particles_per_thread=int(v[1]), table_mode=int(v[2]), rest_reg=int(v[3]), lean=bool(v[4]),
                 fused=int(v[5]), lds_bytes=int(v[6]), cloths_per_cu=int(v[7]), n_cus=int(v[8]), precision="f32" if v[9] else "f64")
        d["name"] = "k_run_schedule<%s,%d,%d,%d,%s,%d>%s: %d B LDS, %d cloths per CU" % (
            "float" if v[9] else "double", v[0], v[1], v[2], "true" if v[3] else "false", v[5], " (LEAN)" if v[4] else "", v[6], v[7])
        n = np.zeros(1, dtype=np.int32)
        check(self._L.clothhip_last_dispatches(self._h, _lib.i32p(n)))
        d["dispatches"] = int(n[0])            # kernel dispatches the launch went out as (one per generation of a time-sliced launch)
        check(self._L.clothhip_last_specialised(self._h, _lib.i32p(n)))
        d["spec_n_side"] = int(n[0])           # 25: the grid-specialised build of that variant ran (25x25 at compile time); 0: the generic build
        if n[0]:
            d["name"] = d["name"].replace(">", ",N%d>" % n[0], 1)
        return d

    def set_relaxed_order(self, on=True):
        """MEASUREMENT ONLY (bench.py's labelled companion): this handle's episode launches run the relaxed-order kernel -- Jacobi
        self-collision, coloured strain limit; NOT the reference's trajectories (clothhip_set_relaxed_order). Per handle."""
        check(self._L.clothhip_set_relaxed_order(self._h, 1 if on else 0))

    @property
    def last_kernel_ms(self):
        return float(self._L.clothhip_last_kernel_ms(self._h))

    # ---- device-resident paths (multi-GPU driver) --------------------------------------------------------
    def run_device_sched_async(self, d_sched_ptr):
        check(self._L.clothhip_run_device_sched_async(self._h, C.c_void_p(int(d_sched_ptr))))

    def write_obs_f32_device(self, d_out_ptr):
        check(self._L.clothhip_write_obs_f32_device(self._h, C.c_void_p(int(d_out_ptr))))

    @property
    def stream(self):
        return self._L.clothhip_stream(self._h)

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        check(self._L.clothhip_device_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def device_free(self, ptr):
        check(self._L.clothhip_device_free(self._h, C.c_void_p(ptr)))

    def device_upload(self, ptr, arr):
        a = np.ascontiguousarray(arr)
        check(self._L.clothhip_device_upload(self._h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def device_download(self, out, ptr):
        assert out.flags['C_CONTIGUOUS']
        check(self._L.clothhip_device_download(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out
