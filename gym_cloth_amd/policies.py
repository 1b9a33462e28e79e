"""Workload drivers: vectorised counterparts of the scripted policies of the reference's examples/analytic.py
that BASELINE configs name (oracle-corner, random), plus highest-point. They read the '1d' observation
(positions) only, so they work on a ClothVecEnv batch without touching per-particle Python objects.

OracleCornerPolicy   examples/analytic.py:70-155  (distance method, delta actions, inset corners 26/48/576/598)
HighestPointPolicy   examples/analytic.py:723-808
RandomPolicy         examples/analytic.py:811-823 (the reference samples from an UNSEEDED space RNG,
                     cloth_env.py:1004; here each env gets its own RandomState so runs are reproducible)
"""
import numpy as np


def _data_delta(x, y, targx, targy, shrink=True):
    """examples/analytic.py:45-67: clip-space grasp point, delta towards the target (x0.9), distance."""
    cx = (x - 0.5) * 2.0
    cy = (y - 0.5) * 2.0
    dx = targx - x
    dy = targy - y
    dist = np.sqrt((x - targx) ** 2 + (y - targy) ** 2)
    if shrink:
        dx = dx * 0.90
        dy = dy * 0.90
    return cx, cy, dx, dy, dist, x, y


class OracleCornerPolicy(object):
    """Pull the (inset) cloth corner that is farthest from its target plane corner."""

    def __init__(self, env):
        self.env = env
        assert env.cfg['env']['delta_actions']
        assert env.num_points == 625, env.num_points                     # analytic.py:106

    def get_action(self, obs, t=0):
        E = self.env.E
        pos = np.asarray(obs, dtype=np.float64).reshape(E, -1, 3)
        tier2 = self.env.cfg['init']['type'] == 'tier2'
        acts = np.zeros((E, 4))
        for e in range(E):
            if tier2 and not self.env.init_side[e]:                      # analytic.py:108-114
                ll, ul, lr, ur = 576, 598, 26, 48
            else:
                ll, ul, lr, ur = 26, 48, 576, 598
            cands = [_data_delta(pos[e, ur, 0], pos[e, ur, 1], 1, 1), _data_delta(pos[e, lr, 0], pos[e, lr, 1], 1, 0),
                     _data_delta(pos[e, ll, 0], pos[e, ll, 1], 0, 0), _data_delta(pos[e, ul, 0], pos[e, ul, 1], 0, 1)]
            maxdist = max(c[4] for c in cands)
            for c in cands:                                              # first match wins (analytic.py:143-150)
                if c[4] == maxdist:
                    cx, cy, dx, dy, _, x, y = c
                    break
            if self.env.cfg['env']['clip_act_space']:                   # analytic.py:151-154
                acts[e] = (cx, cy, dx, dy)
            else:
                acts[e] = (x, y, dx, dy)
        return acts


class HighestPointPolicy(object):
    """Pick one of the top-k highest points at random and pull it to where it sits on the flat cloth
    (examples/analytic.py:723-808; the reference draws the pick from the global numpy stream, here every env has its own
    RandomState(seed + e) so that runs are reproducible and the device evaluation, ClothVecEnv.step_many(policy=
    'highest_point'), can be fed the same picks)."""

    def __init__(self, env, top_k=5, seed=0):
        self.env, self.top_k = env, top_k
        self.rngs = [np.random.RandomState(seed + e) for e in range(env.E)]
        self.orig = env.batch.init_grid(1)[0]                            # pt.orig_x/y of tiers 1 and 3 (cloth.pyx:122-124)
        n = int(round(np.sqrt(env.P)))
        r, c = np.divmod(np.arange(env.P), n)
        self.orig_y2 = c * (1.0 / (n - 1))                               # tier 2 (cloth.pyx:109-110): orig_y, orig_z
        self.orig_z2 = r * (1.0 / (n - 1))

    def draw(self, e):
        """The next pick of env e: which of the highest points (0 = the highest)."""
        return int(self.rngs[e].randint(self.top_k))

    def target(self, e, i):
        """analytic.py:742-789: (orig_x, orig_y) on the flat tiers; tier 2: (orig_z, orig_y) or (1 - orig_z, orig_y)."""
        if self.env._init_type == 'tier2':
            z = self.orig_z2[i]
            return (z if self.env.init_side[e] else 1.0 - z), self.orig_y2[i]
        return self.orig[i, 0], self.orig[i, 1]

    def get_action(self, obs, t=0):
        E = self.env.E
        pos = np.asarray(obs, dtype=np.float64).reshape(E, -1, 3)
        acts = np.zeros((E, 4))
        for e in range(E):
            order = np.argsort(-pos[e, :, 2], kind="stable")             # sorted(..., key=z, reverse=True)
            i = int(order[self.draw(e)])
            tx, ty = self.target(e, i)
            cx, cy, dx, dy, _, x, y = _data_delta(pos[e, i, 0], pos[e, i, 1], tx, ty)
            acts[e] = (cx, cy, dx, dy) if self.env.cfg['env']['clip_act_space'] else (x, y, dx, dy)
        return acts


class RandomPolicy(object):
    """Uniform actions over the action space ('over_xy_plane', cloth_env.py:1003-1004), one RNG per env."""

    def __init__(self, env, seed=2000):
        self.env = env
        self.rngs = [np.random.RandomState(seed + e) for e in range(env.E)]

    def get_action(self, obs=None, t=0):
        sp = self.env.action_space
        return np.stack([r.uniform(low=sp.low, high=sp.high) for r in self.rngs])
