"""Single-cloth façade with the reference's object API (gym_cloth/physics/{cloth,point,gripper}.pyx):
`Cloth.update()`, `.pts[i].x/.y/.z/.px/.py/.pz/.pinned/.orig_*`, `.allpts_arr`, `.have_tear`, `.init_side`,
`Gripper.grab_top/grab/adjust/release/grabbed_pts/grip_radius`.

The particle state lives on the GPU (one env of a ClothBatch).  Attribute reads go through a host mirror
that is refreshed lazily after device work; attribute writes mark the mirror dirty and are flushed before
the next device call.  This path exists for drop-in compatibility with code that pokes at `env.cloth.pts`
(policies, cloth_env.py:854-937, examples/analytic.py:105-125); the throughput path is ClothBatch.run.
"""
import numpy as np

from .batch import ClothBatch


class Point(object):
    """View of one particle (point.pyx:17-51)."""
    __slots__ = ("_c", "_i")

    def __init__(self, cloth, i):
        self._c, self._i = cloth, i

    def _get(self, arr, ax):
        return float(getattr(self._c._host(), arr)[self._i, ax])

    def _set(self, arr, ax, v):
        getattr(self._c._host(), arr)[self._i, ax] = v
        self._c._dirty = True

    x = property(lambda s: s._get("pos", 0), lambda s, v: s._set("pos", 0, v))
    y = property(lambda s: s._get("pos", 1), lambda s, v: s._set("pos", 1, v))
    z = property(lambda s: s._get("pos", 2), lambda s, v: s._set("pos", 2, v))
    px = property(lambda s: s._get("prev", 0), lambda s, v: s._set("prev", 0, v))
    py = property(lambda s: s._get("prev", 1), lambda s, v: s._set("prev", 1, v))
    pz = property(lambda s: s._get("prev", 2), lambda s, v: s._set("prev", 2, v))
    orig_x = property(lambda s: float(s._c._orig[s._i, 0]))
    orig_y = property(lambda s: float(s._c._orig[s._i, 1]))
    orig_z = property(lambda s: float(s._c._orig[s._i, 2]))
    identity_0 = property(lambda s: float(s._i // s._c.width))
    identity_1 = property(lambda s: float(s._i % s._c.width))

    @property
    def pinned(self):
        return bool(self._c._host().pinned[self._i])

    @pinned.setter
    def pinned(self, v):
        h = self._c._host()
        if v and not h.pinned[self._i]:
            self._c._flush()
            self._c.batch.pin_points(self._c.env, [self._i])      # pinned, not a member of grabbed_pts
            self._c._invalidate()
        elif not v and h.pinned[self._i]:
            h.pinned[self._i] = 0
            self._c._dirty = self._c._pin_dirty = True

    def __repr__(self):
        return "({:.3f}, {:.3f}, {:.3f})".format(self.x, self.y, self.z)     # point.pyx:53-55


class Spring(object):
    """cloth.pyx:411-417 (read-only view: the device holds the rest lengths)."""
    __slots__ = ("ptA", "ptB", "type", "rest_length")

    def __init__(self, ptA, ptB, springtype, rest_length):
        self.ptA, self.ptB, self.type, self.rest_length = ptA, ptB, springtype, rest_length


class _Mirror(object):
    __slots__ = ("pos", "prev", "pinned")


class Cloth(object):
    """cloth.pyx:21. Construct either like the reference, Cloth(params=cfg, random_state=rng), which owns a
    one-env ClothBatch, or as a view of env `env` of an existing batch."""

    def __init__(self, gravity=-9.8, bounds=(1, 1, 1), minimum_z=0, params=None, render=False,
                 render_port=5556, random_state=None, state=None, batch=None, env=0, owner=None,
                 device=0, precision="f64"):
        if render:
            raise ValueError("render=True (ZeroMQ viewer feed, cloth.pyx:377-388) is out of scope")
        self._owner = owner
        if batch is None:
            if params is None:
                raise ValueError("params (cfg dict) is required")
            pin_cond = params['cloth'].get('pin_cond', 'default')
            if pin_cond not in ("y=0", "x=0,y=0", "y=0,x=0", "default"):
                raise ValueError(pin_cond)                                  # cloth.pyx:85
            if params['cloth'].get('color_pts', 'None') not in ('None', 'circle0', 'diag0', 'diag1'):
                raise ValueError(params['cloth']['color_pts'])              # cloth.pyx:161
            batch = ClothBatch(params, n_envs=1, device=device, precision=precision, gravity=gravity,
                               minimum_z=float(minimum_z))
            self.params = params
            if random_state is None:
                from . import seeding
                random_state, _ = seeding.np_random(params['seed'])         # cloth.pyx:67-73
            self.np_random = random_state
            self.init_type = params['init']['type']
            self.init_side = bool(random_state.rand() > 0.5)                # cloth.pyx:75
            tier = {'tier1': 1, 'tier2': 2, 'tier3': 3}.get(self.init_type)
            if tier is None:
                raise ValueError(self.init_type)                            # cloth.pyx:131-132
            if state is not None:
                batch.set_state(state['pos'][None], state['prev'][None], state['pinned'][None],
                                state.get('rest'))
            else:
                draws = random_state.rand(batch.P) if tier == 2 else None
                pos, rest = batch.init_grid(tier, self.init_side, draws)
                batch.set_state(pos[None], pos[None], np.zeros((1, batch.P), dtype=np.uint8), rest)
        else:
            self.params = batch.cfg
            self.init_side = False
        self.batch, self.env = batch, int(env)
        self.width = self.height = batch.N
        self.dx = batch.params.width * 1.0 / (batch.N - 1)
        self.dy = batch.params.height * 1.0 / (batch.N - 1)
        self.bounds, self.minimum_z, self.gravity = bounds, minimum_z, gravity
        self.render = False
        self.iter = 0
        self._mirror, self._dirty, self._pin_dirty = None, False, False
        self._orig = batch.positions(self.env, 1)[0].copy()
        self._pts = [Point(self, i) for i in range(batch.P)]

    # ---- host mirror ---------------------------------------------------------------------------------
    def _host(self):
        if self._mirror is None:
            m = _Mirror()
            pos, prev, pin = self.batch.get_state(self.env, 1)
            m.pos, m.prev, m.pinned = pos[0], prev[0], pin[0]
            self._mirror = m
        return self._mirror

    def _flush(self):
        if self._dirty and self._mirror is not None:
            m = self._mirror
            # a write into a live cloth: the sticky tear flag survives (cloth.pyx:272-273). The device-side pin
            # bookkeeping (grab multiplicity) is only overwritten when a point was un-pinned through `pt.pinned = False`.
            self.batch.set_state(m.pos[None], m.prev[None], m.pinned[None] if self._pin_dirty else None, None,
                                 env0=self.env, n=1, keep_tear=True)
        self._dirty = self._pin_dirty = False

    def _invalidate(self):
        self._mirror, self._dirty, self._pin_dirty = None, False, False

    def _rebuilt(self, orig=None):
        """The env rebuilt the cloth (ClothEnv.reset constructs a new Cloth, cloth_env.py:737-746): everything derived from the
        construction is dropped -- Spring.rest_length (tier 2 measures new ones on every reset, cloth.pyx:417), the colour set
        and the points' orig_x/y/z (cloth.pyx:147-164, point.pyx:30-32), which come from the positions the cloth is BUILT with
        (`orig`; default: the device's current ones)."""
        self._invalidate()
        self._springs = None
        self._cmask = None
        self._orig = (self.batch.positions(self.env, 1)[0] if orig is None else np.asarray(orig, dtype=np.float64)).copy()

    # ---- reference API -----------------------------------------------------------------------------------
    @property
    def pts(self):
        return self._pts

    def update(self):
        """Cloth.update (cloth.pyx:169-214) on this cloth only."""
        self._flush()
        if self.batch.E == 1:
            self.batch.update(1)
        else:
            from .batch import make_schedules
            s = make_schedules(self.batch.E, n_griprest_end=1, n_total=1)
            s['active'][self.env] = 1
            self.batch.run(s)
        self._invalidate()
        self.iter += 1

    @property
    def have_tear(self):
        return bool(self.batch.tear[self.env])

    @property
    def allpts_arr(self):
        return self._host().pos.copy()

    @property
    def pinnedpts_arr(self):
        h = self._host()
        return h.pos[h.pinned.astype(bool)].copy()

    # the debugging colour set (cloth.pyx:147-164: chosen from the INITIAL positions by the cfg's color_pts rule)
    def _color_mask(self):
        if getattr(self, "_cmask", None) is None:
            rule = self.params['cloth'].get('color_pts', 'None')
            x, y = self._orig[:, 0], self._orig[:, 1]
            if rule == 'None':
                m = np.zeros(len(x), dtype=bool)
            elif rule == 'circle0':
                m = np.abs((x - 1.0) ** 2 + (y - 1.0) ** 2 - 0.05 ** 2) < 0.10
            elif rule == 'diag0':
                m = np.abs(x - y) < 0.05
            elif rule == 'diag1':
                m = np.abs((1.0 - x) - y) < 0.05
            else:
                raise ValueError(rule)                                                   # cloth.pyx:161
            self._cmask = m
        return self._cmask

    @property
    def color_pts(self):
        """cloth.pyx:164: a set of Points."""
        return set(p for p, c in zip(self._pts, self._color_mask()) if c)

    @property
    def noncolorpts_arr(self):
        return self._host().pos[~self._color_mask()].copy()                              # cloth.pyx:398-400

    @property
    def colorpts_arr(self):
        return self._host().pos[self._color_mask()].copy()                               # cloth.pyx:402-404 (index order here)

    @property
    def springs(self):
        """cloth.pyx:134-146 / :411-417: the spring list in reference order, each with ptA, ptB, type, rest_length."""
        if getattr(self, "_springs", None) is None:
            a, b, t = self.batch.topology()
            rest = self.batch.get_rest(self.env, 1)[0]
            names = ("STRUCTURAL", "SHEARING", "BENDING")
            self._springs = [Spring(self._pts[int(a[s])], self._pts[int(b[s])], names[int(t[s])], float(rest[s]))
                             for s in range(len(a))]
        return self._springs

    def stop_render(self):
        pass


class Gripper(object):
    """gripper.pyx:8. Acts on one env of the batch through the device kernels."""

    def __init__(self, cloth, grip_radius, height, thickness):
        self.cloth, self.grip_radius = cloth, grip_radius
        self.height, self.thickness = height, thickness
        self._grabbed = []

    def _mask(self):
        m = np.zeros(self.cloth.batch.E, dtype=np.uint8)
        m[self.cloth.env] = 1
        return m

    def _after_grab(self, x, y, top):
        """gripper.pyx:39-41 / :52-53 append every hit to grabbed_pts, also points that are already pinned (a later
        adjust then moves them once per entry): rebuild the hit set of THIS call on the host from the state the
        device kernel saw, with the same tests."""
        c = self.cloth
        c._invalidate()
        h = c._host()
        d2 = (h.pos[:, 0] - x) * (h.pos[:, 0] - x) + (h.pos[:, 1] - y) * (h.pos[:, 1] - y)
        inside = d2 < self.grip_radius                                                   # gripper.pyx:35 (radius not squared)
        hits = np.zeros(0, dtype=np.int64)
        if top:
            curz = self.height
            while curz > 0:                                                              # gripper.pyx:31-41
                sel = inside & (np.abs(h.pos[:, 2] - curz) < self.thickness * 2)
                if sel.any():
                    hits = np.nonzero(sel)[0]
                    break
                curz -= self.thickness
        else:
            hits = np.nonzero(inside)[0]
        self._grabbed.extend(c.pts[i] for i in hits)

    @property
    def grabbed_pts(self):
        return self._grabbed

    def grab_top(self, x, y):
        c = self.cloth
        c._flush()
        b = c.batch
        b.grab_top(np.broadcast_to([x, y], (b.E, 2)), radius=np.full(b.E, float(self.grip_radius)),
                   active=self._mask())
        self._after_grab(x, y, True)

    def grab(self, x, y):
        c = self.cloth
        c._flush()
        b = c.batch
        b.grab(np.broadcast_to([x, y], (b.E, 2)), radius=np.full(b.E, float(self.grip_radius)),
               active=self._mask())
        self._after_grab(x, y, False)

    def adjust(self, x, y, z):
        """gripper.pyx:55-66 on the host mirror (compatibility path; ClothBatch.run fuses it on the device)."""
        h = self.cloth._host()
        for pt in self._grabbed:
            i = pt._i
            h.prev[i] = h.pos[i]
            h.pos[i] = np.array([x, y, z]) + h.pos[i]
        if self._grabbed:
            self.cloth._dirty = True

    def release(self):
        self.cloth._flush()
        self.cloth.batch.release(active=self._mask())
        self.cloth._invalidate()
        self._grabbed = []
