#!/usr/bin/env python3
"""Golden-vector generator: runs the REAL reference (DanielTakeshi/gym-cloth) and
dumps small input/output fixtures into tests/golden/*.npz.

Runs only in the development container (needs /root/reference, Cython, gcc).
Nothing from the reference is copied into this repository: the three .pyx files
are compiled *where they lie* into a scratch directory under /tmp, imported,
driven, and only numeric arrays (positions, flags, counters) are written here.

Usage:  python tests/golden/make_golden.py [--only NAME ...]

Fixture inventory:
  g_traj_lift_pull_25.npz   G1/G2  flat 25x25, grab (0.5,0.5), lift+pull+rest+release, checkpoints
  g_traj_fold_25.npz        G1     corner dragged over the cloth -> self-collision + plane contact
  g_traj_tear_25.npz        G1     cloth stretched between a fixed pinned corner and a dragged one -> tear
  g_traj_fold_50.npz        G7     50x50, thickness 0.0095, fold
  g_env_tier1_1337.npz      G3/G4  ClothEnv seed 1337 tier1: reset + oracle-corner episode, per action
  g_env_tier2_*.npz         G5     tier2 reset (2 seeds)
  g_env_tier3_*.npz         G5     tier3 reset (2 seeds)
  g_traj_friction_25.npz    G1     plane_friction 0.5, damping 1.2, ks 7000: the (1 - friction) != 0 plane response
  g_decode_modes.npz        G3     ClothEnv.step with clip_act_space / delta_actions off (4 action modes), per action
  g_gripper_25.npz          G6     grab_top index sets on several states + the curZ level table
  g_metrics.npz             G8     positions -> coverage / variance_inv / out_of_bounds
"""
import argparse
import contextlib
import importlib
import importlib.util
import io
import json
import logging
import os
import shutil
import subprocess
import sys
import tempfile
import textwrap
import time

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SCRATCH = os.environ.get("GOLDEN_SCRATCH", "/tmp/gymcloth_golden_build")


# --------------------------------------------------------------------------------------
# Build + import the reference
# --------------------------------------------------------------------------------------

_STUBS = {
    "zmq/__init__.py": """
        PUB = 1
        LAST_ENDPOINT = 32
        class Context(object):
            def socket(self, *a, **k):
                raise RuntimeError('zmq stub: rendering is disabled for golden generation')
        """,
    "gym/__init__.py": """
        from gym import error, spaces, utils
        class Env(object):
            metadata = {}
        """,
    "gym/error.py": """
        class Error(Exception):
            pass
        """,
    "gym/spaces/__init__.py": """
        import numpy as np
        class Box(object):
            def __init__(self, low, high, dtype=None):
                self.low = np.asarray(low)
                self.high = np.asarray(high)
                self.shape = self.low.shape
                self.dtype = dtype
            def sample(self):
                return np.random.uniform(low=self.low, high=self.high)
        """,
    "gym/utils/__init__.py": """
        from gym.utils import seeding
        """,
    # gym 0.12.1 is not installed and cannot be fetched; this restates its published
    # seeding algorithm (sha512 of str(seed), first 8 bytes, little-endian uint32 list).
    # RNG-derived goldens are self-consistent with this restatement ("parity with real gym unpinned").
    "gym/utils/seeding.py": """
        import hashlib, os, struct
        import numpy as np
        def _bigint_from_bytes(b):
            pad = 4 - len(b) % 4
            b += b'\\0' * pad
            n = len(b) // 4
            acc = 0
            for i, v in enumerate(struct.unpack('{}I'.format(n), b)):
                acc += 2 ** (32 * i) * v
            return acc
        def create_seed(a=None, max_bytes=8):
            if a is None:
                a = _bigint_from_bytes(os.urandom(max_bytes))
            elif isinstance(a, int):
                a = a % 2 ** (8 * max_bytes)
            else:
                raise ValueError(a)
            return a
        def hash_seed(seed=None, max_bytes=8):
            if seed is None:
                seed = create_seed(max_bytes=max_bytes)
            h = hashlib.sha512(str(seed).encode('utf8')).digest()
            return _bigint_from_bytes(h[:max_bytes])
        def _int_list_from_bigint(bigint):
            if bigint == 0:
                return [0]
            out = []
            while bigint > 0:
                bigint, mod = divmod(bigint, 2 ** 32)
                out.append(mod)
            return out
        def np_random(seed=None):
            seed = create_seed(seed)
            rng = np.random.RandomState()
            rng.seed(_int_list_from_bigint(hash_seed(seed)))
            return rng, seed
        """,
    "trimesh/__init__.py": "",
    "cv2/__init__.py": "",
}


def build_reference():
    phys = os.path.join(SCRATCH, "gym_cloth", "physics")
    envs = os.path.join(SCRATCH, "gym_cloth", "envs")
    stubs = os.path.join(SCRATCH, "stubs")
    marker = os.path.join(SCRATCH, ".built")
    if not os.path.exists(marker):
        shutil.rmtree(SCRATCH, ignore_errors=True)
        os.makedirs(phys)
        os.makedirs(envs)
        for n in ("cloth", "point", "gripper"):
            shutil.copy(os.path.join(REF, "gym_cloth", "physics", n + ".pyx"), phys)
        shutil.copy(os.path.join(REF, "gym_cloth", "envs", "cloth_env.py"), envs)
        open(os.path.join(SCRATCH, "gym_cloth", "__init__.py"), "w").close()
        open(os.path.join(phys, "__init__.py"), "w").close()
        with open(os.path.join(envs, "__init__.py"), "w") as fh:
            fh.write("from gym_cloth.envs.cloth_env import ClothEnv\n")
        with open(os.path.join(SCRATCH, "setup_probe.py"), "w") as fh:
            fh.write(textwrap.dedent("""
                from setuptools import setup, Extension
                from Cython.Build import cythonize
                exts = [Extension('gym_cloth.physics.' + n, ['gym_cloth/physics/%s.pyx' % n])
                        for n in ('point', 'gripper', 'cloth')]
                setup(name='probe', ext_modules=cythonize(exts, language_level=3))
                """))
        subprocess.check_call([sys.executable, "setup_probe.py", "build_ext", "--inplace"],
                              cwd=SCRATCH, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for rel, body in _STUBS.items():
            p = os.path.join(stubs, rel)
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "w") as fh:
                fh.write(textwrap.dedent(body))
        os.makedirs(os.path.join(SCRATCH, "logs"), exist_ok=True)
        open(marker, "w").close()
    sys.path.insert(0, SCRATCH)
    sys.path.insert(0, stubs)
    with contextlib.redirect_stdout(io.StringIO()):
        from gym_cloth.physics.cloth import Cloth
        from gym_cloth.physics.gripper import Gripper
    return Cloth, Gripper


def load_cfg(n_side=25, thickness=None, tier="tier1"):
    import yaml
    with open(os.path.join(REF, "cfg", "t1_rgbd.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg["env"]["obs_type"] = "1d"
    cfg["init"]["render_opengl"] = False
    cfg["init"]["type"] = tier
    cfg["log"]["file"] = os.path.join(SCRATCH, "logs", "golden.log")
    cfg["log"]["level"] = "info"
    cfg["cloth"]["num_width_points"] = n_side
    cfg["cloth"]["num_height_points"] = n_side
    if thickness is not None:
        cfg["cloth"]["thickness"] = thickness
    return cfg


def physics_cfg_json(cfg):
    """The numeric constants a replay needs (recorded beside every fixture)."""
    c = cfg["cloth"]
    return json.dumps({
        "n_side": c["num_width_points"], "width": c["width"], "height": c["height"],
        "density": c["density"], "ks": c["ks"], "damping": c["damping"],
        "thickness": c["thickness"], "plane_friction": c["plane_friction"],
        "tear_thresh": c["tear_thresh"], "frames_per_sec": cfg["frames_per_sec"],
        "simulation_steps": cfg["simulation_steps"], "gravity": -9.8, "minimum_z": 0.0,
        "grip_radius": cfg["env"]["grip_radius"],
        "env": {k: cfg["env"][k] for k in ("iters_up", "iters_up_rest", "iters_pull_max",
                                           "iters_grip_rest", "iters_rest", "reduce_factor",
                                           "max_actions")},
    })


# --------------------------------------------------------------------------------------
# State capture helpers
# --------------------------------------------------------------------------------------

def snap(cloth):
    pts = cloth.pts
    pos = np.array([[p.x, p.y, p.z] for p in pts], dtype=np.float64)
    prev = np.array([[p.px, p.py, p.pz] for p in pts], dtype=np.float64)
    pinned = np.array([bool(p.pinned) for p in pts], dtype=np.uint8)
    return pos, prev, pinned


def rest_lengths(cloth):
    return np.array([s.rest_length for s in cloth.springs], dtype=np.float64)


def spring_topology(cloth):
    idx = {id(p): i for i, p in enumerate(cloth.pts)}
    a = np.array([idx[id(s.ptA)] for s in cloth.springs], dtype=np.int32)
    b = np.array([idx[id(s.ptB)] for s in cloth.springs], dtype=np.int32)
    t = np.array([{"STRUCTURAL": 0, "SHEARING": 1, "BENDING": 2}[s.type] for s in cloth.springs],
                 dtype=np.uint8)
    return a, b, t


def cell_census(cloth):
    """(#occupied cells, max occupancy) of the map left over from the last update()."""
    if not cloth.map:
        return 0, 0
    return len(cloth.map), max(len(v) for v in cloth.map.values())


class Trace(object):
    """Drives Cloth/Gripper through a list of ops and records checkpoints.

    ops (JSON-serialisable) are replayed verbatim by the tests against the oracle / the HIP path:
      ["grab_top", x, y] ["grab", x, y] ["release"] ["pin", idx] ["update", n]
      ["adjust_update", dx, dy, dz, n]   (n x { gripper.adjust(dx,dy,dz); cloth.update() })
      ["checkpoint"]
    """

    def __init__(self, Cloth, Gripper, cfg, rng_seed=0):
        self.cfg = cfg
        self.cloth = Cloth(params=cfg, render=False, random_state=np.random.RandomState(rng_seed))
        self.grip = Gripper(self.cloth, cfg["env"]["grip_radius"], cfg["cloth"]["height"],
                            cfg["cloth"]["thickness"])
        self.ops = []
        self.cp = []          # list of (pos, prev, pinned, tear, n_updates)
        self.grabbed = []     # index list after every grab op
        self.census = []
        self.n_updates = 0
        self.rest = rest_lengths(self.cloth)
        self.topo = spring_topology(self.cloth)
        self.init_side = bool(self.cloth.init_side)
        self.checkpoint()

    def checkpoint(self):
        pos, prev, pinned = snap(self.cloth)
        self.cp.append((pos, prev, pinned, bool(self.cloth.have_tear), self.n_updates))
        self.census.append(cell_census(self.cloth))
        self.ops.append(["checkpoint"])

    def _grabbed_idx(self):
        idx = {id(p): i for i, p in enumerate(self.cloth.pts)}
        return [idx[id(p)] for p in self.grip.grabbed_pts]

    def grab_top(self, x, y):
        self.grip.grab_top(x, y)
        self.ops.append(["grab_top", float(x), float(y)])
        self.grabbed.append(self._grabbed_idx())

    def grab(self, x, y):
        self.grip.grab(x, y)
        self.ops.append(["grab", float(x), float(y)])
        self.grabbed.append(self._grabbed_idx())

    def pin(self, i):
        self.cloth.pts[i].pinned = True
        self.ops.append(["pin", int(i)])

    def release(self):
        self.grip.release()
        self.ops.append(["release"])

    def update(self, n):
        for _ in range(n):
            self.cloth.update()
        self.n_updates += n
        self.ops.append(["update", int(n)])

    def adjust_update(self, dx, dy, dz, n):
        for _ in range(n):
            self.grip.adjust(dx, dy, dz)
            self.cloth.update()
        self.n_updates += n
        self.ops.append(["adjust_update", float(dx), float(dy), float(dz), int(n)])

    def save(self, name):
        a, b, t = self.topo
        out = dict(
            cfg=physics_cfg_json(self.cfg), ops=json.dumps(self.ops),
            grabbed=json.dumps(self.grabbed), init_side=np.uint8(self.init_side),
            rest=self.rest, spring_a=a, spring_b=b, spring_type=t,
            cp_pos=np.stack([c[0] for c in self.cp]), cp_prev=np.stack([c[1] for c in self.cp]),
            cp_pinned=np.stack([c[2] for c in self.cp]),
            cp_tear=np.array([c[3] for c in self.cp], dtype=np.uint8),
            cp_n_updates=np.array([c[4] for c in self.cp], dtype=np.int64),
            cp_cells=np.array(self.census, dtype=np.int32),
        )
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **out)
        print("wrote %s (%.1f KB, %d checkpoints, %d updates)" %
              (name, os.path.getsize(path) / 1024.0, len(self.cp), self.n_updates))


# --------------------------------------------------------------------------------------
# Physics-only trajectories
# --------------------------------------------------------------------------------------

def traj_lift_pull_25(Cloth, Gripper):
    cfg = load_cfg(25)
    tr = Trace(Cloth, Gripper, cfg)
    tr.update(3); tr.checkpoint()                      # flat / inert fixed point
    tr.grab_top(0.5, 0.5); tr.checkpoint()
    d = np.array([0.6, 0.8]) / (1.0 + 1e-5) * 0.002
    tr.adjust_update(0.0, 0.0, 0.0025, 1); tr.checkpoint()     # substep 1 of the lift
    tr.adjust_update(0.0, 0.0, 0.0025, 9); tr.checkpoint()     # 10
    tr.adjust_update(0.0, 0.0, 0.0025, 40); tr.checkpoint()    # 50
    tr.update(50); tr.checkpoint()                             # 100
    tr.update(30); tr.checkpoint()                             # 130
    tr.adjust_update(d[0], d[1], 0.0, 1); tr.checkpoint()      # first lateral substep
    tr.adjust_update(d[0], d[1], 0.0, 69); tr.checkpoint()     # 200, strain limiting engaged
    tr.adjust_update(d[0], d[1], 0.0, 1); tr.checkpoint()      # single substep mid-pull
    tr.adjust_update(d[0], d[1], 0.0, 79); tr.checkpoint()
    tr.update(60); tr.checkpoint()
    tr.release(); tr.update(1); tr.checkpoint()                # release kick
    tr.update(199); tr.checkpoint()                            # landing on the plane
    tr.update(1); tr.checkpoint()
    tr.save("g_traj_lift_pull_25.npz")
    return tr


def traj_fold(Cloth, Gripper, n_side, thickness, name, lift=50, pull=330, rest=150):
    cfg = load_cfg(n_side, thickness=thickness)
    tr = Trace(Cloth, Gripper, cfg)
    eps = 0.5 / (n_side - 1)
    tr.grab_top(eps, eps); tr.checkpoint()
    tr.adjust_update(0.0, 0.0, 0.0025, lift); tr.checkpoint()
    d = np.array([1.0, 1.0]) / (np.sqrt(2.0) + 1e-5) * 0.002
    half = pull // 2
    tr.adjust_update(d[0], d[1], 0.0, half); tr.checkpoint()
    tr.adjust_update(d[0], d[1], 0.0, pull - half); tr.checkpoint()
    tr.update(20); tr.checkpoint()
    tr.release(); tr.update(rest); tr.checkpoint()
    tr.update(1); tr.checkpoint()                               # single substep, layers stacked
    tr.save(name)
    return tr


def traj_tear_25(Cloth, Gripper):
    cfg = load_cfg(25)
    tr = Trace(Cloth, Gripper, cfg)
    tr.pin(0); tr.pin(1); tr.pin(25)           # fixed corner (pinned, not grabbed): both-pinned springs
    tr.grab_top(1.0, 1.0); tr.checkpoint()
    d = np.array([1.0, 1.0]) / (np.sqrt(2.0) + 1e-5) * 0.002
    tr.adjust_update(0.0, 0.0, 0.0025, 20); tr.checkpoint()
    n = 0
    while not tr.cloth.have_tear and n < 3000:
        tr.adjust_update(d[0], d[1], 0.0, 1)
        n += 1
        if n % 100 == 0:
            tr.checkpoint()
    # collapse the single-step ops just recorded into one op list that the tests can replay
    tr.checkpoint()                                             # first state with tear == True
    tr.adjust_update(d[0], d[1], 0.0, 1); tr.checkpoint()       # sticky
    print("  tear after %d lateral substeps (have_tear=%s)" % (n, tr.cloth.have_tear))
    tr.save("g_traj_tear_25.npz")
    return tr


def traj_friction_25(Cloth, Gripper):
    """Non-default material and plane constants: plane_friction 0.5 makes (1. - friction) != 0 in
    _handle_plane_collision (cloth.pyx:345-370), so a particle below the plane is NOT simply restored."""
    cfg = load_cfg(25)
    cfg["cloth"]["plane_friction"] = 0.5
    cfg["cloth"]["damping"] = 1.2
    cfg["cloth"]["ks"] = 7000.0
    tr = Trace(Cloth, Gripper, cfg)
    eps = 0.5 / 24
    tr.grab_top(eps, eps); tr.checkpoint()
    tr.adjust_update(0.0, 0.0, 0.0025, 30); tr.checkpoint()
    d = np.array([1.0, 0.6]) / (np.sqrt(1.36) + 1e-5) * 0.002
    tr.adjust_update(d[0], d[1], 0.0, 1); tr.checkpoint()
    tr.adjust_update(d[0], d[1], 0.0, 119); tr.checkpoint()
    tr.release(); tr.update(1); tr.checkpoint()
    tr.update(60); tr.checkpoint()                              # falling back onto the plane: z < 0 corrections
    tr.update(1); tr.checkpoint()
    tr.update(140); tr.checkpoint()
    below = int(np.sum(tr.cp[-1][0][:, 2] < 1e-3))
    print("  friction trajectory: %d particles within 1e-3 of the plane at the end" % below)
    tr.save("g_traj_friction_25.npz")
    return tr


def decode_modes_fixture():
    """ClothEnv.step in the action modes the shipped cfgs do not use (cloth_env.py:402-470): clip_act_space off and/or
    delta_actions off. The reference cannot reset() in non-delta mode (cloth_env.py:862 raises), so the scripted reset
    pulls are skipped and every action starts from the flat cloth; what is captured per action is the start and end
    state, the number of update() calls and the (obs, reward, done, info) the reference returned."""
    import yaml
    recs = []
    for clip, delta in ((True, True), (False, True), (True, False), (False, False)):
        cfg = load_cfg(25)
        cfg["env"]["clip_act_space"], cfg["env"]["delta_actions"] = clip, delta
        cfg["seed"] = 7
        path = os.path.join(SCRATCH, "cfg_modes_%d%d.yaml" % (clip, delta))
        with open(path, "w") as fh:
            yaml.safe_dump(cfg, fh)
        with contextlib.redirect_stdout(io.StringIO()):
            from gym_cloth.envs import ClothEnv
        if delta:
            acts = [(0.52, 0.48, 0.3, -0.2), (-0.4, 0.7, 0.9, 0.8)] if clip else \
                   [(0.75, 0.25, -0.25, 0.15), (1.2, 0.5, 0.1, -1.4)]          # the 2nd is out of the action bounds
        else:
            acts = [(0.1, -0.3, 0.2, 0.35), (0.6, 0.6, -0.5, -0.8)] if clip else \
                   [(0.3, 0.7, 0.35, 2.2), (0.5, 0.5, 1.3, -4.0)]              # length > 1 and |angle| > pi get truncated
        for a in acts:
            env = ClothEnv(path)
            logging.getLogger().setLevel(logging.WARNING)
            env.logger.setLevel(logging.WARNING)
            env.seed(7)
            env._wd = env._hd = 224
            env._reset_actions = lambda: None
            with contextlib.redirect_stdout(io.StringIO()):
                env.reset()
            pos0, prev0, pin0 = snap(env.cloth)
            it0 = env.cloth.iter
            with contextlib.redirect_stdout(io.StringIO()):
                obs, rew, done, info = env.step(a)
            pos1, prev1, pin1 = snap(env.cloth)
            recs.append(dict(clip=clip, delta=delta, action=[float(x) for x in a], n_updates=env.cloth.iter - it0,
                             pos0=pos0, pos1=pos1, prev1=prev1, rew=float(rew), done=bool(done),
                             info={k: (float(v) if isinstance(v, (float, np.floating)) else
                                       (bool(v) if isinstance(v, (bool, np.bool_)) else int(v))) for k, v in info.items()},
                             low=[float(x) for x in env.action_space.low], high=[float(x) for x in env.action_space.high]))
            print("  mode clip=%s delta=%s action %s -> %d updates, rew %.4f" % (clip, delta, a, recs[-1]["n_updates"], rew))
    out = dict(cfg=physics_cfg_json(load_cfg(25)),
               meta=json.dumps([{k: r[k] for k in ("clip", "delta", "action", "n_updates", "rew", "done", "info", "low", "high")}
                                for r in recs]),
               pos0=np.stack([r["pos0"] for r in recs]), pos1=np.stack([r["pos1"] for r in recs]),
               prev1=np.stack([r["prev1"] for r in recs]))
    path = os.path.join(HERE, "g_decode_modes.npz")
    np.savez_compressed(path, **out)
    print("wrote g_decode_modes.npz (%.1f KB)" % (os.path.getsize(path) / 1024.0))


def gripper_fixture(Cloth, Gripper, traces):
    """grab_top / grab index sets on a few harvested states + the curZ table (G6)."""
    cfg = load_cfg(25)
    th = cfg["cloth"]["thickness"]
    levels = []
    curz = float(cfg["cloth"]["height"])
    while curz > 0:
        levels.append(curz)
        curz -= th
    states, queries, res_top, res_grab = [], [], [], []
    qs = [(0.5, 0.5), (0.0, 0.0), (1.0, 1.0), (0.3, 0.7), (0.52, 0.48), (1.2, 0.5), (0.21, 0.23),
          (0.75, 0.75), (0.4, 0.4)]
    for tr, cps in traces:
        for ci in cps:
            pos, prev, pinned = tr.cp[ci][0], tr.cp[ci][1], tr.cp[ci][2]
            for (x, y) in qs:
                c = Cloth(params=cfg, render=False, random_state=np.random.RandomState(0))
                for p, xyz in zip(c.pts, pos):
                    p.x, p.y, p.z = xyz
                g = Gripper(c, cfg["env"]["grip_radius"], cfg["cloth"]["height"], th)
                g.grab_top(x, y)
                idx = {id(p): i for i, p in enumerate(c.pts)}
                top = sorted(idx[id(p)] for p in g.grabbed_pts)
                g2 = Gripper(c, cfg["env"]["grip_radius"], cfg["cloth"]["height"], th)
                for p in c.pts:
                    p.pinned = False
                g2.grab(x, y)
                allc = sorted(idx[id(p)] for p in g2.grabbed_pts)
                states.append(pos); queries.append((x, y)); res_top.append(top); res_grab.append(allc)
    np.savez_compressed(os.path.join(HERE, "g_gripper_25.npz"),
                        cfg=physics_cfg_json(cfg), levels=np.array(levels),
                        pos=np.stack(states), xy=np.array(queries),
                        grab_top=json.dumps(res_top), grab=json.dumps(res_grab))
    print("wrote g_gripper_25.npz (%d queries, %d levels, last %.17g)" %
          (len(queries), len(levels), levels[-1]))


# --------------------------------------------------------------------------------------
# Env-level fixtures (ClothEnv + the oracle-corner policy from examples/analytic.py)
# --------------------------------------------------------------------------------------

def make_env(tier, seed):
    import yaml
    cfg = load_cfg(25, tier=tier)
    cfg["seed"] = seed
    path = os.path.join(SCRATCH, "cfg_%s_%d.yaml" % (tier, seed))
    with open(path, "w") as fh:
        yaml.safe_dump(cfg, fh)
    with contextlib.redirect_stdout(io.StringIO()):
        from gym_cloth.envs import ClothEnv
    env = ClothEnv(path)
    logging.getLogger().setLevel(logging.WARNING)
    env.logger.setLevel(logging.WARNING)
    env.seed(seed)
    env._wd = env._hd = 224     # reset() reads them unconditionally (cloth_env.py:789)
    return env, cfg


class StepSpy(object):
    """Records every env.step() call (incl. the initialize=True ones issued by reset())."""

    def __init__(self, env):
        self.env = env
        self.calls = []
        self._orig = env.step
        env.step = self._step

    def _step(self, action, initialize=False):
        cloth = self.env.cloth
        pos0, prev0, pin0 = snap(cloth)
        ns0 = self.env.num_sim_steps
        it0 = cloth.iter
        out = self._orig(action, initialize=initialize)
        pos1, prev1, pin1 = snap(cloth)
        self.calls.append(dict(action=[float(a) for a in action], initialize=bool(initialize),
                               pos0=pos0, prev0=prev0, pin0=pin0, pos1=pos1, prev1=prev1, pin1=pin1,
                               n_updates=cloth.iter - it0, tear=bool(cloth.have_tear),
                               iters_up=float(self.env.iters_up)))
        return out


def env_fixture(tier, seed, episode, name):
    t0 = time.time()
    env, cfg = make_env(tier, seed)
    np.random.seed(seed)                      # analytic.py:853
    spy = StepSpy(env)
    # capture the cloth as constructed, before _reset_actions mutate it
    cap = {}
    orig_reset_actions = env._reset_actions

    def _ra():
        cap["init_pos"] = snap(env.cloth)[0]
        cap["rest"] = rest_lengths(env.cloth)
        cap["init_side"] = bool(env.cloth.init_side)
        it0 = env.cloth.iter
        orig_reset_actions()
        cap["reset_updates"] = env.cloth.iter - it0
    env._reset_actions = _ra
    obs = env.reset()
    post_pos, post_prev, post_pin = snap(env.cloth)
    out = dict(cfg=physics_cfg_json(cfg), tier=tier, seed=np.int64(seed),
               init_pos=cap["init_pos"], rest=cap["rest"], init_side=np.uint8(cap["init_side"]),
               reset_updates=np.int64(cap["reset_updates"]),
               post_pos=post_pos, post_prev=post_prev, post_pinned=post_pin,
               start_coverage=np.float64(env._start_coverage),
               start_variance_inv=np.float64(env._start_variance_inv),
               reset_obs=np.asarray(obs, dtype=np.float64))
    n_reset_calls = len(spy.calls)
    rews, dones, infos = [], [], []
    if episode:
        spec = importlib.util.spec_from_file_location("ref_analytic",
                                                      os.path.join(REF, "examples", "analytic.py"))
        mod = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
        pol = mod.OracleCornerPolicy()
        pol.set_env_cfg(env, cfg)
        done, t = False, 0
        while not done:
            with contextlib.redirect_stdout(io.StringIO()):
                a = pol.get_action(obs, t=t)
            obs, rew, done, info = env.step(a)
            rews.append(float(rew)); dones.append(bool(done))
            infos.append({k: (float(v) if not isinstance(v, (bool, np.bool_)) else bool(v))
                          for k, v in info.items()})
            t += 1
    calls = spy.calls
    out.update(
        n_reset_calls=np.int64(n_reset_calls),
        act=np.array([c["action"] for c in calls]),
        act_initialize=np.array([c["initialize"] for c in calls], dtype=np.uint8),
        act_n_updates=np.array([c["n_updates"] for c in calls], dtype=np.int64),
        act_tear=np.array([c["tear"] for c in calls], dtype=np.uint8),
        act_iters_up=np.array([c["iters_up"] for c in calls]),
        act_pos0=np.stack([c["pos0"] for c in calls]), act_prev0=np.stack([c["prev0"] for c in calls]),
        act_pin0=np.stack([c["pin0"] for c in calls]),
        act_pos1=np.stack([c["pos1"] for c in calls]), act_prev1=np.stack([c["prev1"] for c in calls]),
        act_pin1=np.stack([c["pin1"] for c in calls]),
        rew=np.array(rews), done=np.array(dones, dtype=np.uint8), info=json.dumps(infos))
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB; %d reset actions, %d episode actions, start cov %.16f) in %.0fs" %
          (name, os.path.getsize(path) / 1024.0, n_reset_calls, len(rews),
           float(env._start_coverage), time.time() - t0))
    return env



def _load_analytic():
    spec = importlib.util.spec_from_file_location("ref_analytic", os.path.join(REF, "examples", "analytic.py"))
    mod = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    return mod


def episodes_fixture(tier, seed, n_episodes, max_actions_total, name):
    """The reference's data-collection loop (examples/analytic.py:866-882: reset, step until done, reset again ...) with RANDOM
    clip-space actions from RandomState(seed + 5000) -- whole episodes incl. the reset that follows an episode's end, so that a
    replay through the fused episode launch (which draws that reset from the env's RandomState on the device) is pinned to the
    reference directly, not to this repository's own sequential path."""
    t0 = time.time()
    env, cfg = make_env(tier, seed)
    np.random.seed(seed)
    rs = np.random.RandomState(seed + 5000)
    acts, rews, dones, infos, n_updates, reset_before = [], [], [], [], [], []
    reset_cov, reset_obs = [], []
    n_ep = 0
    obs = env.reset()
    reset_cov.append(float(env._start_coverage)); reset_obs.append(np.asarray(obs, dtype=np.float64))
    pending_reset = 1
    while n_ep < n_episodes and len(acts) < max_actions_total:
        a = rs.uniform(-1, 1, size=4)
        it0 = env.cloth.iter
        obs, rew, done, info = env.step(a)
        acts.append(a); rews.append(float(rew)); dones.append(bool(done)); n_updates.append(env.cloth.iter - it0)
        infos.append({k: (float(v) if not isinstance(v, (bool, np.bool_)) else bool(v)) for k, v in info.items()})
        reset_before.append(pending_reset); pending_reset = 0
        if done:
            n_ep += 1
            if n_ep < n_episodes and len(acts) < max_actions_total:
                obs = env.reset()
                reset_cov.append(float(env._start_coverage)); reset_obs.append(np.asarray(obs, dtype=np.float64))
                pending_reset = 1
    pos, prev, pin = snap(env.cloth)
    out = dict(cfg=physics_cfg_json(cfg), tier=tier, seed=np.int64(seed), act=np.array(acts), rew=np.array(rews),
               done=np.array(dones, dtype=np.uint8), n_updates=np.array(n_updates, dtype=np.int64), info=json.dumps(infos),
               reset_before=np.array(reset_before, dtype=np.uint8), reset_start_coverage=np.array(reset_cov),
               reset_obs=np.stack(reset_obs), final_pos=pos, final_prev=prev, final_pinned=pin)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB; %d actions in %d episodes, %d resets) in %.0fs" %
          (name, os.path.getsize(path) / 1024.0, len(acts), n_ep, len(reset_cov), time.time() - t0))


def highest_point_fixture(name):
    """examples/analytic.py's HighestPointPolicy.get_action (:792-808) under np.random.seed(k), k = 0..7 (so that every rank of its
    randint(5) pick occurs), on a post-reset and on a post-action (folded, points in the air settled) state, tiers 1 and 2 with
    both tier-2 sides. Stored: the state, the seeds and the actions."""
    mod = _load_analytic()
    out, t0 = {}, time.time()
    cases = [("tier1", 1337), ("tier2", 1337), ("tier2", 1338)]
    sides = []
    for ci, (tier, seed) in enumerate(cases):
        env, cfg = make_env(tier, seed)
        np.random.seed(seed)
        obs = env.reset()
        pol = mod.HighestPointPolicy()
        pol.set_env_cfg(env, cfg)
        sides.append(int(bool(env.cloth.init_side)))
        for si in range(2):
            if si == 1:                                   # one real action first: a folded / crumpled state
                with contextlib.redirect_stdout(io.StringIO()):
                    np.random.seed(99)
                    a = pol.get_action(obs, t=0)
                obs, _, _, _ = env.step(a)
            pos, prev, pin = snap(env.cloth)
            acts = []
            for k in range(8):
                np.random.seed(k)
                with contextlib.redirect_stdout(io.StringIO()):
                    acts.append([float(v) for v in pol.get_action(obs, t=si)])
            np.random.seed(0)
            picks = [int(np.random.RandomState(k).randint(5)) for k in range(8)]     # what np.random.seed(k); randint(5) draws
            out["c%d_s%d_pos" % (ci, si)] = pos
            out["c%d_s%d_act" % (ci, si)] = np.array(acts)
            out["c%d_s%d_pick" % (ci, si)] = np.array(picks, dtype=np.int64)
    out["tiers"] = np.array([c[0] for c in cases]); out["seeds"] = np.array([c[1] for c in cases], dtype=np.int64)
    out["init_side"] = np.array(sides, dtype=np.uint8)
    out["cfg"] = physics_cfg_json(cfg)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB) in %.0fs" % (name, os.path.getsize(path) / 1024.0, time.time() - t0))


def coverage_reward_fixture(name):
    """reward_type 'coverage' (cloth_env.py:656-662: the non-delta reward) over a short tier-1 episode of random actions."""
    import yaml
    t0 = time.time()
    cfg = load_cfg(25, tier="tier1")
    cfg["seed"] = 1337
    cfg["env"]["reward_type"] = "coverage"
    path_cfg = os.path.join(SCRATCH, "cfg_covrew.yaml")
    with open(path_cfg, "w") as fh:
        yaml.safe_dump(cfg, fh)
    with contextlib.redirect_stdout(io.StringIO()):
        from gym_cloth.envs import ClothEnv
    env = ClothEnv(path_cfg)
    logging.getLogger().setLevel(logging.WARNING); env.logger.setLevel(logging.WARNING)
    env.seed(1337); env._wd = env._hd = 224
    np.random.seed(1337)
    env.reset()
    rs = np.random.RandomState(77)
    acts, rews, dones, covs = [], [], [], []
    done = False
    while not done and len(acts) < 4:
        a = rs.uniform(-0.8, 0.8, size=4)
        _, rew, done, info = env.step(a)
        acts.append(a); rews.append(float(rew)); dones.append(bool(done)); covs.append(float(info["actual_coverage"]))
    np.savez_compressed(os.path.join(HERE, name), cfg=physics_cfg_json(cfg), act=np.array(acts), rew=np.array(rews),
                        done=np.array(dones, dtype=np.uint8), coverage=np.array(covs))
    print("wrote %s (%d actions) in %.0fs" % (name, len(acts), time.time() - t0))


def mesh_fixture(name):
    """What the reference hands to Blender for an image observation (cloth_env.py:212-276): the triangle mesh built from the
    particles (vertex order, faces, winding) and the command line (init_side, tier, image size), captured with trimesh.Trimesh and
    subprocess.call replaced by recorders; plus the numeric scene constants of the Blender script (camera pose and lens, the two
    cloth colours) read out of gym_cloth/blender/get_image_rep_279.py at generation time. Two states: post-reset, post-action."""
    import re
    import subprocess as sp
    import trimesh
    t0 = time.time()
    env, cfg = make_env("tier1", 1337)
    np.random.seed(1337)
    env.reset()
    rec = {}

    class _TM(object):
        def __init__(self, vertices, faces):
            rec["v"], rec["f"] = np.array(vertices, dtype=np.float64), np.array(faces, dtype=np.int64)

        def export(self, path):
            pass
    trimesh.Trimesh = _TM
    import gym_cloth.envs.cloth_env as ce
    real_call = sp.call
    ce.subprocess.call = lambda argv, *a, **k: rec.__setitem__("argv", list(argv)) or 0
    ce.time.sleep = lambda *_: None
    out = {}
    try:
        for si in range(2):
            if si == 1:
                env.step(np.array([-0.2, 0.1, 0.6, 0.5]))
            try:
                env.get_blender_rep("False")
            except Exception:                              # no image comes back: everything of interest happened before
                pass
            out["s%d_vertices" % si], out["s%d_faces" % si] = rec["v"], rec["f"]
            out["s%d_pos" % si] = snap(env.cloth)[0]
            argv = rec["argv"]
            k = argv.index("--")
            out["s%d_argv" % si] = np.array([str(a) for a in argv[k + 2:k + 6]])        # height, width, init_side, init type
    finally:
        ce.subprocess.call = real_call
    src = open(os.path.join(REF, "gym_cloth", "blender", "get_image_rep_279.py")).read()
    num = lambda pat: float(re.search(pat, src).group(1))
    out["camera_location"] = np.array([num(r"location\[0\] = ([0-9.]+)\s*\+ cp\[0\]"), num(r"location\[1\] = ([0-9.]+)\s*\+ cp\[1\]"),
                                       num(r"location\[2\] = ([0-9.]+)\s*\+ cp\[2\]")])
    out["camera_lens_mm"] = np.float64(num(r"data\.lens = ([0-9.]+)"))
    out["camera_sensor_mm"] = np.float64(num(r"data\.sensor_width = ([0-9.]+)"))
    m = re.search(r"else:\s*\n\s*b = np\.array\(\[([0-9., ]+)\]\)\s*\n\s*f = np\.array\(\[([0-9., ]+)\]\)", src)
    out["color_back"] = np.array([float(v) for v in m.group(1).split(",")])
    out["color_front"] = np.array([float(v) for v in m.group(2).split(",")])
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote %s (%d faces) in %.0fs" % (name, len(out["s0_faces"]), time.time() - t0))


def metrics_fixture(states):
    """positions -> (coverage via scipy ConvexHull.volume, variance_inv, out_of_bounds) as the env computes them
    (cloth_env.py:1020-1098)."""
    from scipy.spatial import ConvexHull
    rng = np.random.RandomState(7)
    P = states[0].shape[0]
    extra = []
    extra.append(states[0] * np.array([1.0, 1.0, 0.0]))                      # exactly flat
    s = states[-1].copy(); s[:, 0] += 0.3; extra.append(s)                   # clipped at x=1, OOB on x
    s = states[-1].copy(); s[:, 1] -= 0.26; extra.append(s)                  # OOB low y
    s = states[1].copy(); s[5, 2] = -1e-9; extra.append(s)                   # z<0 -> OOB
    s = states[1].copy(); s[7, 2] = 1.0; extra.append(s)                     # z>=1 -> OOB
    s = rng.uniform(-0.2, 1.2, size=(P, 3)); s[:, 2] = np.abs(s[:, 2]) * 0.1; extra.append(s)
    s = rng.uniform(0.4, 0.6, size=(P, 3)); extra.append(s)                  # small blob
    allst = list(states) + extra
    cov, vinv, oob = [], [], []
    for st in allst:
        pts = np.array([[min(max(x, 0), 1), min(max(y, 0), 1)] for x, y, _ in st])
        cov.append(ConvexHull(pts).volume)
        var = np.var(st[:, 2])
        vinv.append(1000.0 if var < 0.000001 else 0.001 / var)
        o = (np.max(st[:, 0]) >= 1.25 or np.min(st[:, 0]) < -0.25 or np.max(st[:, 1]) >= 1.25 or
             np.min(st[:, 1]) < -0.25 or np.max(st[:, 2]) >= 1 or np.min(st[:, 2]) < 0)
        oob.append(bool(o))
    np.savez_compressed(os.path.join(HERE, "g_metrics.npz"), pos=np.stack(allst),
                        coverage=np.array(cov), variance_inv=np.array(vinv),
                        oob=np.array(oob, dtype=np.uint8))
    print("wrote g_metrics.npz (%d states)" % len(allst))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    want = lambda k: args.only is None or k in args.only
    Cloth, Gripper = build_reference()
    tr_lp = tr_fold = None
    if want("lift_pull") or want("gripper") or want("metrics"):
        tr_lp = traj_lift_pull_25(Cloth, Gripper)
    if want("fold") or want("gripper") or want("metrics"):
        tr_fold = traj_fold(Cloth, Gripper, 25, None, "g_traj_fold_25.npz")
    if want("tear"):
        traj_tear_25(Cloth, Gripper)
    if want("fold50"):
        traj_fold(Cloth, Gripper, 50, 0.0095, "g_traj_fold_50.npz", lift=40, pull=260, rest=60)
    if want("friction"):
        traj_friction_25(Cloth, Gripper)
    if want("decode"):
        decode_modes_fixture()
    if want("gripper"):
        gripper_fixture(Cloth, Gripper, [(tr_lp, [0, 6, 10]), (tr_fold, [3, 5])])
    if want("metrics"):
        sts = [tr_lp.cp[i][0] for i in (0, 4, 6, 10, 13)] + [tr_fold.cp[i][0] for i in (1, 2, 3, 5)]
        metrics_fixture(sts)
    if want("env1"):
        env_fixture("tier1", 1337, True, "g_env_tier1_1337.npz")
    if want("env2"):
        env_fixture("tier2", 1337, False, "g_env_tier2_1337.npz")
        env_fixture("tier2", 1338, False, "g_env_tier2_1338.npz")
    if want("env3"):
        env_fixture("tier3", 1337, False, "g_env_tier3_1337.npz")
        env_fixture("tier3", 1339, False, "g_env_tier3_1339.npz")
    if want("episodes1"):
        episodes_fixture("tier1", 1337, 3, 12, "g_episodes_tier1_1337.npz")
    if want("episodes3"):
        episodes_fixture("tier3", 1339, 2, 8, "g_episodes_tier3_1339.npz")
    if want("highest"):
        highest_point_fixture("g_highest_point.npz")
    if want("covrew"):
        coverage_reward_fixture("g_coverage_reward.npz")
    if want("mesh"):
        mesh_fixture("g_mesh_export.npz")


if __name__ == "__main__":
    main()
