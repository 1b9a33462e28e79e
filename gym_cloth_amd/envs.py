"""Host side of the drop-in: the gym `ClothEnv.step()/reset()` surface of the reference
(gym_cloth/envs/cloth_env.py) driving the HIP stepper, vectorised over E independent cloths.

ClothVecEnv  E envs on one GPU (new: the reference is one env per process, analysis/README.md:8-13)
ClothEnv     the reference's single-env API (cloth_env.py:55) = a ClothVecEnv with E = 1

What is mirrored here is exactly what sits between the gym API and the physics hot path:
  action clipping / un-clipping / direction / iters_pull        cloth_env.py:396-470
  phase schedule + per-iteration adjust -> update -> tear?      cloth_env.py:352-367, :472-515
  reset procedures of the three tiers and their RNG sequence    cloth_env.py:717-987, cloth.pyx:75,101
  reward / terminal / out-of-bounds / coverage / variance       cloth_env.py:536-715, :1020-1098
Blender observations, the OpenGL viewer, logging-to-file and pickle I/O are out of scope;
obs_type is forced to the '1d' observation (cloth_env.py:196-200).
"""
import copy

import numpy as np
import ctypes as _ctypes
import yaml

from . import seeding
from .batch import ClothBatch, make_schedules

_REWARD_THRESHOLDS = {           # cloth_env.py:42-50
    'coverage': 0.92, 'coverage-delta': 0.92, 'height': 0.85, 'height-delta': 0.85,
    'variance': 2, 'variance-delta': 2, 'folding-number': 0,
}
_EPS = 1e-5                      # cloth_env.py:52



def _mt_state_address(rng):
    """Address of numpy's `mt19937_state { uint32 key[624]; int pos; }` behind a RandomState (its MT19937 bit generator exports
    it for exactly this kind of access), or None: then get_state() / set_state() are used. Copying 2 500 bytes per env this way
    instead of building state tuples takes the per-launch RNG hand-over of 512 envs from ~30 ms to ~1 ms."""
    try:
        bg = rng._bit_generator
        if type(bg).__name__ != 'MT19937':
            return None
        return int(bg.ctypes.state_address)
    except Exception:
        return None

class Box(object):
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency of the hot path)."""

    def __init__(self, low, high, dtype=np.float64):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = dtype
        self.np_random = np.random.RandomState()

    def seed(self, seed=None):
        self.np_random.seed(seed)

    def sample(self):
        return self.np_random.uniform(low=self.low, high=self.high)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def load_cfg(cfg):
    if isinstance(cfg, dict):
        return copy.deepcopy(cfg)
    with open(cfg, 'r') as fh:                       # cloth_env.py:87-88
        return yaml.safe_load(fh)


def decode_actions(actions, low, high, clip_act_space, delta_actions, reduce_factor, iters_up, iters_up_rest,
                   iters_pull_max, iters_grip_rest, iters_rest):
    """The action -> schedule arithmetic of ClothEnv.step (cloth_env.py:396-475), vectorised over envs.

    actions [E,4]; iters_up scalar or [E] (tier 3 draws a float per env, cloth_env.py:960).
    Returns dict(x, y, x_dir_r, y_dir_r, iters_pull, bounds[E,5]); bounds are the integer ceilings of the
    cumulative phase boundaries (`i < b` for integer i is `i < ceil(b)`), formed left to right as the reference
    writes them (cloth_env.py:472-475).
    """
    a = np.asarray(actions, dtype=np.float64)
    E = a.shape[0]
    a = np.maximum(np.minimum(a, np.asarray(high)[None, :]), np.asarray(low)[None, :])    # :402-415
    x_coord, y_coord, c2, c3 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    if clip_act_space:                                                                  # :417-426
        x_coord = (x_coord / 2.0) + 0.5
        y_coord = (y_coord / 2.0) + 0.5
        if not delta_actions:
            c2 = (c2 / 2.0) + 0.5
            c3 = c3 * np.pi
    if delta_actions:                                                                   # :448-453
        total_length = np.sqrt((c2) ** 2 + (c3) ** 2)
        x_dir = c2 / (total_length + _EPS)
        y_dir = c3 / (total_length + _EPS)
    else:
        x_dir = np.cos(c3)
        y_dir = np.sin(c3)
    x_dir_r = x_dir * reduce_factor                                                     # :455-456
    y_dir_r = y_dir * reduce_factor
    if delta_actions:                                                                   # :460-468
        step = np.sqrt((x_dir_r) ** 2 + (y_dir_r) ** 2)
        cur = np.zeros(E)
        ii = np.zeros(E, dtype=np.int64)
        alive = np.ones(E, dtype=bool)
        guard = 0
        while alive.any():
            cur = cur + step
            alive &= ~(cur >= total_length)
            ii[alive] += 1
            guard += 1
            if guard > 1000000:
                raise FloatingPointError("iters_pull does not terminate (non-finite action?)")
        iters_pull = ii
    else:
        iters_pull = (iters_pull_max * c2).astype(np.int64)                             # :470
    iu = np.broadcast_to(np.asarray(iters_up, dtype=np.float64), (E,))
    b1 = iu
    b2 = iu + iters_up_rest
    b3 = iu + iters_up_rest + iters_pull
    b4 = iu + iters_up_rest + iters_pull + iters_grip_rest
    b5 = iu + iters_up_rest + iters_pull + iters_grip_rest + iters_rest
    bounds = np.ceil(np.stack([b1, b2, b3, b4, b5], axis=1)).astype(np.int64)
    return dict(x=x_coord, y=y_coord, x_dir_r=x_dir_r, y_dir_r=y_dir_r, iters_pull=iters_pull, bounds=bounds)


class ClothVecEnv(object):
    metadata = {'render.modes': ['human']}

    def __init__(self, cfg_file, n_envs=1, device=0, precision='f32', consume_domrand_draws=True):
        cfg = load_cfg(cfg_file)
        self.cfg, self.cfg_file = cfg, cfg_file
        e = cfg['env']
        self.max_actions = e['max_actions']
        self.iters_up = e['iters_up']
        self.iters_up_rest = e['iters_up_rest']
        self.iters_pull_max = e['iters_pull_max']
        self.iters_grip_rest = e['iters_grip_rest']
        self.iters_rest = e['iters_rest']
        self.reduce_factor = e['reduce_factor']
        self.grip_radius = e['grip_radius']
        self._init_type = cfg['init']['type']
        self._clip_act_space = e['clip_act_space']
        self._delta_actions = e['delta_actions']
        self._force_grab = e['force_grab']
        self._radius_inc = 0.02
        self.bounds = (1, 1, 1)
        self.reward_type = e['reward_type']
        assert 'coverage' in self.reward_type                       # cloth_env.py:130
        self._neg_living_rew = 0.0
        self._nogrip_penalty = -0.01
        self._tear_penalty = 0.0
        self._oob_penalty = 0.0
        self._cover_success = 5.
        self._act_bound_factor = 1.0
        self._act_pen_limit = 3.0
        self._slack = 0.25
        self.num_w = cfg['cloth']['num_width_points']
        self.num_h = cfg['cloth']['num_height_points']
        self.num_points = self.num_w * self.num_h
        self.obslow = np.ones((3 * self.num_points,)) * -100
        self.obshigh = np.ones((3 * self.num_points,)) * 100
        self.observation_space = Box(self.obslow, self.obshigh)
        b0, b1 = self.bounds[0], self.bounds[1]
        if self._clip_act_space:                                     # cloth_env.py:162-181
            self.action_space = Box([-1., -1., -1., -1.], [1., 1., 1., 1.])
        elif self._delta_actions:
            self.action_space = Box([0., 0., -1., -1.], [1., 1., 1., 1.])
        else:
            self.action_space = Box([-self._slack, -self._slack, 0.0, -np.pi],
                                    [b0 + self._slack, b1 + self._slack, 1.0, np.pi])
        self._wd = self._hd = 224
        self._consume_domrand = consume_domrand_draws

        self.E = int(n_envs)
        self.batch = ClothBatch(cfg, n_envs=self.E, device=device, precision=precision)
        self.P = self.batch.P
        E = self.E
        self.np_randoms = [None] * E
        self.seed(None)
        self.init_side = np.zeros(E, dtype=bool)
        self.num_steps = np.zeros(E, dtype=np.int64)
        self.num_sim_steps = np.zeros(E, dtype=np.int64)
        self.have_tear = np.zeros(E, dtype=bool)
        self._prev_reward = np.zeros(E)
        self._start_coverage = np.zeros(E)
        self._start_variance_inv = np.zeros(E)
        self._current_coverage = np.zeros(E)
        self._iters_up_env = np.full(E, float(self.iters_up))       # tier 3 overrides it per env during reset
        self.last_executed = np.zeros(E, dtype=np.int64)
        self.last_grabbed = np.zeros(E, dtype=np.int64)
        self.last_iters_pull = np.zeros(E, dtype=np.int64)
        self._ep_done = np.zeros(E, dtype=bool)                      # episode over, reset pending (step_many's auto-reset)
        self._pending = [None] * E                                   # pre-drawn reset scripts of step_many, per env
        self.total_substeps = 0                                      # executed update() calls, all envs

    def close(self):
        self.batch.close()

    # ------------------------------------------------------------------------------------------------
    def seed(self, seed=None):
        """cloth_env.py:332-341, one RandomState per env. `seed` may be None, an int (env e gets
        seed + e) or a sequence of E ints."""
        if seed is None or isinstance(seed, (int, np.integer)):
            seeds = [None if seed is None else int(seed) + e for e in range(self.E)]
        else:
            seeds = list(seed)
            assert len(seeds) == self.E
        out = []
        for e, s in enumerate(seeds):
            self.np_randoms[e], s2 = seeding.np_random(s)
            out.append(s2)
        self._pending = [None] * self.E                                # reset scripts pre-drawn from the old generators
        return out

    @property
    def state(self):
        """'1d' observation of every env, [E, 3P] (cloth_env.py:196-200)."""
        return self.batch.positions().reshape(self.E, 3 * self.P)

    def image_obs(self, use_depth=False, rgbd=False, **render_kw):
        """Image observations without Blender (the reference's obs_type 'blender', cloth_env.py:196-209, :212-330), rendered
        by the library's own rasteriser from the device-resident particles: uint8 [E, H, W, 3] colour, or with use_depth the
        depth image replicated to 3 channels (the Z pass normalised over the image, get_image_rep_279.py:390-406, minus the 50
        grey levels of cloth_env.py:301-302), or with rgbd [E, H, W, 4] = colour + depth channel (:202-205). The camera /
        colour parameters follow the reference's Blender script; the pixels are not Blender's (no smoothing filter, no
        domain randomisation, own shading)."""
        swap = (~self.init_side).astype(np.uint8) if self._init_type == 'tier2' else None      # get_image_rep_279.py:235-239
        rgb, dep = self.batch.render(want_rgb=(rgbd or not use_depth), want_depth=(rgbd or use_depth), swap_sides=swap,
                                     **render_kw)
        d8 = None
        if dep is not None:
            lo = dep.min(axis=(1, 2), keepdims=True); hi = dep.max(axis=(1, 2), keepdims=True)
            nz = np.where(hi > lo, (dep - lo) / np.where(hi > lo, hi - lo, 1.0), 0.0)
            d8 = np.uint8(np.maximum(0.0, np.rint(nz * 255.0) - 50.0))
        if rgbd:
            return np.concatenate([rgb, d8[..., None]], axis=-1)
        return np.repeat(d8[..., None], 3, axis=-1) if use_depth else rgb

    # ---- action decoding (cloth_env.py:396-475) -------------------------------------------------------
    def decode_actions(self, actions, iters_up=None):
        """-> dict(x, y, x_dir_r, y_dir_r, iters_pull, bounds[E,5])."""
        return decode_actions(actions, self.action_space.low, self.action_space.high, self._clip_act_space,
                              self._delta_actions, self.reduce_factor,
                              self._iters_up_env if iters_up is None else iters_up, self.iters_up_rest,
                              self.iters_pull_max, self.iters_grip_rest, self.iters_rest)

    # ---- one action for every (active) env ----------------------------------------------------------------
    def step(self, actions, initialize=False, active=None, auto_reset=False):
        """ClothEnv.step (cloth_env.py:369-534) for all envs at once.
        Returns (obs[E,3P], rew[E], done[E], info dict of arrays); with initialize=True returns None.
        auto_reset: envs that finish their episode in this step are reset before the call returns, as the
        reference's episode loop does (`while not done: step` then `env.reset()`, examples/analytic.py:872-882);
        their row of `obs` is then the first observation of the new episode, info['terminal_observation'] holds
        the last one of the finished episode and info['reset_mask'] says which envs were reset."""
        E = self.E
        act_mask = np.ones(E, dtype=bool) if active is None else np.asarray(active, dtype=bool).copy()
        d = self.decode_actions(actions)
        xy = np.stack([d['x'], d['y']], axis=1)
        n_grab = self.batch.grab_top(xy, active=act_mask).astype(np.int64)              # :431
        if self._force_grab:                                                            # :434-444
            radius = np.full(E, float(self.grip_radius))
            need = act_mask & (n_grab == 0)
            while need.any():
                radius[need] += self._radius_inc
                n2 = self.batch.grab_top(xy, radius=radius, active=need)
                n_grab[need] = n2[need]
                need = need & (n_grab == 0)
        exit_early = act_mask & (n_grab == 0)                                           # :490-493
        run_mask = act_mask & ~exit_early
        s = make_schedules(E, break_on_tear=1, dz_up=0.0025)                            # :359
        b = d['bounds']
        s['n_up_end'], s['n_uprest_end'], s['n_pull_end'] = b[:, 0], b[:, 1], b[:, 2]
        s['n_griprest_end'], s['n_total'] = b[:, 3], b[:, 4]
        s['dx_pull'], s['dy_pull'] = d['x_dir_r'], d['y_dir_r']
        s['active'] = run_mask
        executed = self.batch.run(s).astype(np.int64)
        executed[~run_mask] = 0
        self.total_substeps += int(executed.sum())
        self.last_executed, self.last_grabbed, self.last_iters_pull = executed, n_grab, d['iters_pull']
        cov, vinv, oob, tear = self.batch.metrics()
        self.have_tear |= (run_mask & tear)                                             # :511-514
        if initialize:                                                                  # :517-518
            return None
        self.num_sim_steps[act_mask] += executed[act_mask]
        self.num_steps[act_mask] += 1
        rew = self._reward(actions, exit_early, cov, vinv, oob, act_mask)
        term = self._terminal(oob, act_mask)
        self._ep_done = np.where(act_mask, term, self._ep_done)
        info = {
            'num_steps': self.num_steps.copy(), 'num_sim_steps': self.num_sim_steps.copy(),
            'actual_coverage': self._current_coverage.copy(), 'start_coverage': self._start_coverage.copy(),
            'variance_inv': vinv, 'start_variance_inv': self._start_variance_inv.copy(),
            'have_tear': self.have_tear.copy(), 'out_of_bounds': oob,
            'executed': executed.copy(), 'n_grabbed': n_grab.copy(),      # (not in the reference's info) this action's update() calls
        }
        obs = self.state
        if auto_reset and term.any():
            info['terminal_observation'] = obs
            info['reset_mask'] = term.copy()
            obs = self.reset(mask=term)
        return obs, rew, term, info

    # ---- whole episodes on the device (clothhip_run_actions) ------------------------------------------------------
    def _episode_params(self):
        from ._lib import ClothEpisodeParams
        ep = ClothEpisodeParams()
        ep.max_actions = int(self.max_actions)
        ep.iters_up_rest, ep.iters_grip_rest, ep.iters_rest = int(self.iters_up_rest), int(self.iters_grip_rest), int(self.iters_rest)
        ep.clip_act_space, ep.force_grab = int(bool(self._clip_act_space)), int(bool(self._force_grab))
        ep.iters_up = float(self.iters_up)
        ep.reduce_factor, ep.grip_radius = float(self.reduce_factor), float(self.grip_radius)
        ep.radius_inc, ep.dz_up = float(self._radius_inc), 0.0025
        for k in range(4):
            ep.act_low[k], ep.act_high[k] = float(self.action_space.low[k]), float(self.action_space.high[k])
        ep.coverage_done = float(_REWARD_THRESHOLDS[self.reward_type])
        return ep

    def _domrand_draws(self, rng):
        """cloth_env.py:786-789: the draws every reset makes after the scripted actions."""
        rng.uniform(low=40, high=50)
        rng.uniform(low=0.7, high=1.3)
        lim = rng.uniform(low=-15.0, high=15.0)
        rng.uniform(low=-lim, high=lim, size=(self._wd, self._hd, 3))

    def _draw_script(self, rng, tier, out):
        """Draw ONE reset from `rng` in the reference's order (cloth.pyx:75; cloth_env.py:851-877 / :959-978) into the
        script record `out`. Returns (init_side, rng state if the reset runs 2 pulls, rng state if it runs 3 (tier 1))."""
        init_side = bool(rng.rand() > 0.5)                                               # cloth.pyx:75
        out['valid'], out['_pad'] = 1, 0
        if tier == 1:
            lim = 0.20
            out['n_pulls'], out['settle_after'] = 3, 0
            s2 = None
            for k in range(3):
                if k == 2:
                    s2 = rng.get_state()                          # the third pull's draws happen only if coverage >= 0.90
                pl = out['pull'][k]
                pl['point'] = rng.randint(self.P)
                pl['dx'] = self._randval_minabs(rng, -lim, lim, 0.08)
                pl['dy'] = self._randval_minabs(rng, -lim, lim, 0.08)
                pl['x'] = pl['y'] = 0.0
                pl['need_coverage'], pl['coverage_min'] = int(k == 2), 0.90
                pl['iters_up'] = float(self.iters_up)
            return init_side, s2, rng.get_state()
        lim = 0.25                                                                       # tier 3
        out['n_pulls'], out['settle_after'] = 1, 800
        pl = out['pull'][0]
        pl['iters_up'] = rng.uniform(low=200, high=280)
        pl['x'] = self._randval_minabs(rng, 0.30, 0.70)
        pl['y'] = self._randval_minabs(rng, 0.30, 0.70)
        pl['dx'] = self._randval_minabs(rng, -lim, lim, 0.10)
        pl['dy'] = self._randval_minabs(rng, -lim, lim, 0.10)
        pl['point'], pl['need_coverage'], pl['coverage_min'] = -1, 0, 0.0
        st = rng.get_state()
        return init_side, st, st

    def _drop_pending(self, e):
        """Give back the draws of the env's pre-drawn reset scripts (the RNG is parked at the first one's start)."""
        self._pending[e] = None

    def _extend_chain(self, e, n):
        """Make env e's chain of pre-drawn resets at least n long (see _prepare_scripts); the RandomState stays parked.
        A chain is {'nodes': [{before, after: (state after the unconditional pulls, state after all pulls)}], 'recs':
        RESET_SCRIPT_DTYPE[n], 'sides': bool[n]}."""
        from ._lib import RESET_SCRIPT_DTYPE
        chain = self._pending[e]
        if chain is None:
            chain = self._pending[e] = {'nodes': [], 'recs': np.zeros(0, dtype=RESET_SCRIPT_DTYPE), 'sides': np.zeros(0, dtype=bool)}
        nodes = chain['nodes']
        if len(nodes) >= n:
            return chain
        tier = {'tier1': 1, 'tier3': 3}[self._init_type]
        rng = self.np_randoms[e]
        s_start = rng.get_state()
        if nodes:
            rng.set_state(nodes[-1]['after'][0])
            if self._consume_domrand:
                self._domrand_draws(rng)
        k0 = len(nodes)
        recs = np.zeros(n, dtype=RESET_SCRIPT_DTYPE)
        sides = np.zeros(n, dtype=bool)
        recs[:k0], sides[:k0] = chain['recs'], chain['sides']
        for k in range(k0, n):
            before = rng.get_state()
            sides[k], s2, s3 = self._draw_script(rng, tier, recs[k])
            nodes.append({'before': before, 'after': (s2, s3)})
            rng.set_state(s2)                                     # the chain continues as if only the unconditional pulls ran
            if self._consume_domrand:
                self._domrand_draws(rng)
        chain['recs'], chain['sides'] = recs, sides
        rng.set_state(s_start)
        return chain

    def _prepare_scripts(self, n_scripts):
        """The next `n_scripts` resets of every env, [E, R]. Script k+1 is drawn from the RNG state script k leaves when
        only its unconditional pulls run (tier 1: two pulls; the third, coverage-conditional one draws further numbers and
        forks the stream, cloth_env.py:866-877). Scripts that were not consumed stay cached for the next launch; each env's
        RandomState stays parked where its first pending script starts, so a host-side reset() simply re-draws."""
        from ._lib import RESET_SCRIPT_DTYPE
        R = int(n_scripts)
        scripts = np.zeros((self.E, R), dtype=RESET_SCRIPT_DTYPE)
        self._script_sides = np.zeros((self.E, R), dtype=bool)        # Cloth.init_side of each scripted reset (cloth.pyx:75)
        for e in range(self.E):
            chain = self._extend_chain(e, R)
            scripts[e] = chain['recs'][:R]
            self._script_sides[e] = chain['sides'][:R]
        return scripts

    def _apply_reset_records(self, ie, rb, rst, n_consumed, substeps_out, device_rng):
        """ClothEnv.reset bookkeeping (cloth_env.py:717-790) for the envs `ie` whose rb[e]-th reset of this launch ran."""
        if not len(ie):
            return
        k = rb[ie] - 1
        q = rst[ie, k]
        self.init_side[ie] = (q['init_side'] != 0) if device_rng else self._script_sides[ie, k]
        n_consumed[ie] = k + 1
        sub = q['executed'].sum(axis=1).astype(np.int64) + q['settle_executed']
        substeps_out[ie] = sub
        self.total_substeps += int(sub.sum())
        self.num_steps[ie] = 0; self.num_sim_steps[ie] = 0
        self.have_tear[ie] = q['tear'] != 0
        self._prev_reward[ie] = q['start_coverage']
        self._start_coverage[ie] = q['start_coverage']
        self._start_variance_inv[ie] = q['start_variance_inv']
        self._current_coverage[ie] = 0.0
        self._ep_done[ie] = False

    def step_many(self, actions=None, n_actions=None, policy=None, auto_reset=True, want_obs=False, reset_tail=False,
                  actions_device_ptr=None, max_resets=None, time_budget_ms=0.0, device_rng=True, policy_choices=None):
        """T consecutive `step(a_t, auto_reset=auto_reset)` calls for every env in ONE device launch
        (clothhip_run_actions): decoding, grab, the substep loop, metrics, the terminal test and the episode resets all
        run in the kernel, envs never wait for each other, and the host only does the reward / info bookkeeping below.

        actions: float64[T, E, 4], or policy='oracle_corner' (examples/analytic.py's oracle, evaluated on the device)
        with n_actions=T, or policy='highest_point' (analytic.py:723-808 on the device) with n_actions=T and
        policy_choices int[T, E]: which of the highest points (0 = the highest; the reference draws randint(5)) the env
        pulls in its t-th slot. With auto_reset=False an env whose episode ends idles for the rest of the launch (`ran` False). Up to `max_resets` (default min(T, 255), the
        launch's limit) resets per env and launch; an env that has used them idles until the launch ends. The resets are drawn on the device from
        each env's numpy RandomState stream (device_rng=True: the states are uploaded before and read back after the
        launch; csrc/cloth_rng.hpp reproduces numpy's MT19937 draws bit for bit; all three tiers, tier 2 rebuilding the
        env's noisy sheet and rest lengths in the kernel), or, with device_rng=False (tiers 1 and 3), pre-drawn
        on the host as a chain of scripts -- then, when tier 1's coverage-conditional third reset pull runs, that env's
        later scripts are void (the RNG stream forked) and it idles after its next episode until the launch ends.
        An episode that ends in the last slot is reset by the NEXT launch, or here on the host with reset_tail=True (then
        the returned obs is what T sequential steps return).

        time_budget_ms > 0 turns the launch into a time slice: an env starts no further action once the launch has run that
        long, so envs advance at their own pace instead of waiting for the one with the most work (episode resets make
        the work per env very uneven). out['ran'][:, e] is then True for the first n_e slots only, and the caller passes
        actions[n_e:, e] again in the next call. Results do not depend on how an env's actions are split into launches.

        Returns a dict of arrays [T, E] (rew, done, ran, executed, n_grabbed, reset_before, and the info keys of step())
        plus 'obs' [E, 3P] (state after the launch), 'actions' [T, E, 4], with want_obs 'obs_t' [T, E, 3P], and the launch's time
        accounting 'op_ticks' / 'op_substeps' uint64[E, 4] (ClothBatch.op_ticks)."""
        from . import _lib
        import time as _time
        _tp = [_time.perf_counter()]
        prof = self.__dict__.setdefault('host_prof', {})

        def _lap(name):
            now = _time.perf_counter()
            prof[name] = prof.get(name, 0.0) + now - _tp[0]
            _tp[0] = now
        E = self.E
        if policy in (None, 'table'):
            pol = _lib.POLICY_TABLE
            if actions_device_ptr is None:
                actions = np.ascontiguousarray(actions, dtype=np.float64)
                T = actions.shape[0]
            else:
                T = int(n_actions)
        elif policy == 'oracle_corner':
            pol, T = _lib.POLICY_ORACLE_CORNER, int(n_actions)
            assert self._delta_actions and self.num_points == 625                        # analytic.py:103-106
        elif policy == 'highest_point':                                                  # analytic.py:723-808, evaluated on the device
            pol, T = _lib.POLICY_HIGHEST_POINT, int(n_actions)
            policy_choices = np.asarray(policy_choices, dtype=np.int32)                  # [T, E]: 0 = the highest point, 1 = the next ...
            if policy_choices.shape != (T, E):
                raise ValueError("policy_choices must have shape (%d, %d)" % (T, E))
        else:
            raise ValueError(policy)
        if not self._delta_actions:
            raise NotImplementedError("non-delta actions are decoded on the host only (cos/sin, cloth_env.py:452-453)")
        dev_reset = auto_reset and (self._init_type in ('tier1', 'tier3') or (self._init_type == 'tier2' and device_rng))
        R = min(T, 255) if max_resets is None else int(max_resets)      # clothhip_run_actions: n_scripts in [1, 255]
        if dev_reset and not 1 <= R <= 255:
            raise ValueError("max_resets must be in [1, 255] (got %d)" % R)
        use_rng = dev_reset and device_rng
        scripts = self._prepare_scripts(R) if (dev_reset and not use_rng) else None
        mt = gauss = None
        if use_rng:                                                   # every env's numpy stream, as RandomState.get_state() has it
            if self._init_type == 'tier2':
                self.batch.ensure_per_env_rest()                      # the kernel rebuilds each env's rest lengths at a reset
            for e in range(E):
                self._drop_pending(e)
            mt = np.zeros((E, _lib.MT_WORDS), dtype=np.uint32)
            gauss = [None] * E
            mt_addr = [_mt_state_address(self.np_randoms[e]) for e in range(E)]
            row0, row_bytes = mt.ctypes.data, mt.strides[0]
            for e in range(E):
                if mt_addr[e]:                                        # key[624] + pos, straight out of numpy's generator state
                    _ctypes.memmove(row0 + e * row_bytes, mt_addr[e], 625 * 4)
                else:
                    st = self.np_randoms[e].get_state()
                    mt[e, :624], mt[e, 624], gauss[e] = st[1], st[2], st[3:]
            mt_before = mt.copy()
        parg = None
        if pol != _lib.POLICY_TABLE and self._init_type == 'tier2':
            parg = np.where(self.init_side, 2, 1).astype(np.int32)                       # analytic.py:108-114, :781-788
        if pol == _lib.POLICY_HIGHEST_POINT:
            parg = np.concatenate([(np.zeros(E, dtype=np.int32) if parg is None else parg)[None, :], policy_choices], axis=0)
        nsteps = np.ascontiguousarray(self.num_steps, dtype=np.int32)
        done_io = np.ascontiguousarray(self._ep_done, dtype=np.uint8)
        _lap('prepare_resets')
        self.batch.run_actions_begin(self._episode_params(), T, nsteps, done_io, actions=actions, policy=pol,
                                     policy_arg=parg, scripts=scripts, want_obs=want_obs,
                                     actions_device_ptr=actions_device_ptr, time_budget_ms=time_budget_ms,
                                     rng_states=mt, rng_tier={'tier1': 1, 'tier2': 2, 'tier3': 3}.get(self._init_type, 0),
                                     domrand_words=2 * (3 + self._wd * self._hd * 3) if self._consume_domrand else 0,
                                     reset_capacity=R)
        _lap('launch')
        if dev_reset and not use_rng:                                 # while the kernel runs: draw ahead for the NEXT launch
            for e in range(E):
                self._extend_chain(e, 2 * R)
        _lap('draw_ahead(overlapped)')
        rec, rst, obs_t, robs = self.batch.run_actions_end()
        op_ticks, op_substeps = self.batch.op_ticks()
        _lap('wait+download')
        if (rec['ran'] == 2).any():
            raise FloatingPointError("iters_pull does not terminate (non-finite action?)")
        out = {k: np.zeros((T, E)) for k in ('rew', 'actual_coverage', 'start_coverage', 'variance_inv', 'start_variance_inv')}
        for k in ('done', 'ran', 'have_tear', 'out_of_bounds'):
            out[k] = np.zeros((T, E), dtype=bool)
        for k in ('executed', 'n_grabbed', 'num_steps', 'num_sim_steps', 'reset_before', 'reset_substeps'):
            out[k] = np.zeros((T, E), dtype=np.int64)
        out['actions'] = rec['action'].copy()
        # per env: 100 MHz ticks of this launch spent in {actions, reset pulls, reset settling, the rest} and the update() calls of each
        out['op_ticks'], out['op_substeps'] = op_ticks, op_substeps
        n_consumed = np.zeros(E, dtype=np.int64)
        for t in range(T):
            r = rec[t]
            rb = r['reset_before'].astype(np.int64)
            self._apply_reset_records(np.nonzero(rb)[0], rb, rst, n_consumed, out['reset_substeps'][t], use_rng)
            ran = r['ran'] == 1
            executed = np.where(ran, r['executed'], 0).astype(np.int64)
            self.total_substeps += int(executed.sum())
            self.have_tear |= ran & (r['tear'] != 0)
            self.num_sim_steps[ran] += executed[ran]
            self.num_steps[ran] += 1
            oob = r['oob'] != 0
            rew = self._reward(r['action'], ran & (r['n_grabbed'] == 0), r['coverage'], r['variance_inv'], oob, ran,
                               height=r['n_below_half_thickness'] / float(self.P))
            term = self._terminal(oob, ran)
            if not np.array_equal(term[ran], (r['done'] != 0)[ran]):
                raise RuntimeError("device and host disagree on the terminal test")
            self._ep_done = np.where(ran, term, self._ep_done)
            self.last_executed = np.where(ran, executed, self.last_executed)
            self.last_grabbed = np.where(ran, r['n_grabbed'], self.last_grabbed)
            self.last_iters_pull = np.where(ran, r['iters_pull'], self.last_iters_pull)
            out['rew'][t], out['done'][t], out['ran'][t] = rew, term, ran
            out['executed'][t], out['n_grabbed'][t], out['reset_before'][t] = executed, r['n_grabbed'], rb
            out['num_steps'][t], out['num_sim_steps'][t] = self.num_steps, self.num_sim_steps
            out['actual_coverage'][t], out['start_coverage'][t] = self._current_coverage, self._start_coverage
            out['variance_inv'][t], out['start_variance_inv'][t] = r['variance_inv'], self._start_variance_inv
            out['have_tear'][t], out['out_of_bounds'][t] = self.have_tear, oob
        if dev_reset:                                                 # a time slice may end right after a reset
            tail = np.nonzero((rst['consumed'] == 1).sum(axis=1) > n_consumed)[0]
            if len(tail):
                rb = np.zeros(E, dtype=np.int64)
                rb[tail] = n_consumed[tail] + 1
                out['tail_reset_substeps'] = np.zeros(E, dtype=np.int64)
                out['tail_reset_index'] = rb.copy()                   # 1-based index into reset_obs[e] of that reset, 0 = none
                self._apply_reset_records(tail, rb, rst, n_consumed, out['tail_reset_substeps'], use_rng)
        # the device's view of the episode state must be the host's, except where a time slice cut a reset in the middle (the
        # device has already zeroed that env's counters; the host learns of the reset from the launch that completes it)
        settled = np.ones(E, dtype=bool) if rst is None else ~(rst['consumed'] == 2).any(axis=1)
        assert np.array_equal(self.num_steps.astype(np.int32)[settled], nsteps[settled])
        assert np.array_equal(self._ep_done[settled], (done_io != 0)[settled])
        _lap('bookkeeping')
        if use_rng:                                                   # the streams as the device left them (a reset cut by the
            for e in np.nonzero((mt != mt_before).any(axis=1))[0]:    # time slice has drawn too, before any record of it is complete)
                if mt_addr[e]:                                        # (the cached gaussian of RandomState is not the device's business)
                    _ctypes.memmove(mt_addr[e], row0 + int(e) * row_bytes, 625 * 4)
                else:
                    self.np_randoms[e].set_state(('MT19937', mt[e, :624], int(mt[e, 624])) + tuple(gauss[e]))
        elif dev_reset:                                               # commit the RNG draws the device consumed
            for e in np.nonzero(n_consumed)[0]:
                chain, c = self._pending[e], int(n_consumed[e])
                nodes, last_rec = chain['nodes'], chain['recs'][c - 1]
                n_uncond = int((last_rec['pull']['need_coverage'][:int(last_rec['n_pulls'])] == 0).sum())
                forked = int(rst[e, c - 1]['pulls_run']) > n_uncond
                rng = self.np_randoms[e]
                if forked or c >= len(nodes):                         # forked: the conditional pull ran, later scripts are void
                    rng.set_state(nodes[c - 1]['after'][1 if forked else 0])
                    if self._consume_domrand:
                        self._domrand_draws(rng)
                    self._pending[e] = None
                else:
                    rng.set_state(nodes[c]['before'])
                    self._pending[e] = {'nodes': nodes[c:], 'recs': chain['recs'][c:], 'sides': chain['sides'][c:]}
        _lap('rng_commit')
        obs = self.state
        _lap('obs_download')
        if reset_tail and auto_reset and self._ep_done.any():
            obs = self.reset(mask=self._ep_done.copy())
        out['obs'] = obs
        if want_obs:
            out['obs_t'], out['reset_obs'] = obs_t, robs
        return out

    # ---- reward / terminal (cloth_env.py:536-715) ------------------------------------------------------------
    def _height_fraction(self):
        """compute_height (cloth_env.py:603-609): the count comes from the device metrics kernel."""
        return self.batch.metrics(want_height=True)[4]

    def _reward(self, actions, exit_early, cov, vinv, oob, mask, height=None):
        E = self.E
        hf = (lambda: height) if height is not None else self._height_fraction
        rew = np.zeros(E)
        rew += np.where(self.have_tear, self._tear_penalty, np.where(oob, self._oob_penalty, 0.0))   # :557-562
        rew += np.where(exit_early, self._nogrip_penalty, 0.0)                                       # :563-565
        if not self._clip_act_space:                                                                 # :579-593
            a = np.asarray(actions, dtype=np.float64).reshape(E, 4)
            low, high = self.action_space.low, self.action_space.high
            diff = np.where(a < low, low - a, np.where(a > high, a - high, 0.0))
            pen = -np.minimum(diff ** 2, self._act_pen_limit) * self._act_bound_factor
            rew += pen.sum(axis=1)
        self._current_coverage = np.where(mask, cov, self._current_coverage)                         # :647
        rew += np.where(cov > _REWARD_THRESHOLDS['coverage'], self._cover_success, 0.0)              # :648-650
        rew += self._neg_living_rew

        def delta(val):
            diff = val - self._prev_reward
            self._prev_reward = np.where(mask, val, self._prev_reward)
            return diff
        rt = self.reward_type                                                                        # :656-679
        if rt == 'coverage':
            rew += cov
        elif rt == 'coverage-delta':
            rew += delta(cov)
        elif rt == 'height':
            rew += hf()
        elif rt == 'height-delta':
            rew += delta(hf())
        elif rt == 'variance':
            rew += vinv
        elif rt == 'variance-delta':
            rew += delta(vinv)
        elif rt == 'folding-number':
            raise NotImplementedError()
        else:
            raise ValueError(rt)
        return np.where(mask, rew, 0.0)

    def _terminal(self, oob, mask):
        done = (self.num_steps >= self.max_actions) | self.have_tear | oob                           # :692-703
        done |= self._current_coverage > _REWARD_THRESHOLDS[self.reward_type]                        # :706-710
        return done & mask

    def _compute_coverage(self):
        return self.batch.metrics()[0]

    def _compute_variance(self):
        return self.batch.metrics()[1]

    def _out_of_bounds(self):
        return self.batch.metrics()[2]

    def _convert_action_to_clip_space(self, a):                     # cloth_env.py:1207-1215, vectorised
        a = np.asarray(a, dtype=np.float64)
        if not self._clip_act_space:
            return a
        out = a.copy()
        out[..., 0] = (a[..., 0] - 0.5) * 2
        out[..., 1] = (a[..., 1] - 0.5) * 2
        if not self._delta_actions:
            out[..., 2] = (a[..., 2] - 0.5) * 2
            out[..., 3] = a[..., 3] / np.pi
        return out

    def get_random_action(self, atype='over_xy_plane'):
        """cloth_env.py:989-1018; [E,4]. NB the reference samples from the action space's own, unseeded RNG."""
        if atype == 'over_xy_plane':
            return np.stack([self.action_space.sample() for _ in range(self.E)])
        raise ValueError(atype)

    # ---- reset (cloth_env.py:717-987; cloth.pyx:75, :94-130) -------------------------------------------------
    def _update_masked(self, n, mask):
        s = make_schedules(self.E, n_griprest_end=n, n_total=n, break_on_tear=0)
        s['active'] = mask
        ex = self.batch.run(s)
        self.total_substeps += int(ex[mask].sum())

    def reset(self, mask=None):
        """Reset the envs selected by `mask` (default all) and return the '1d' observation of ALL envs."""
        E, P = self.E, self.P
        m = np.ones(E, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        idx = np.nonzero(m)[0]
        tier = {'tier1': 1, 'tier2': 2, 'tier3': 3}.get(self._init_type)
        if tier is None:
            raise ValueError(self._init_type)                         # cloth.pyx:131-132
        if not self._delta_actions:
            raise NotImplementedError()                               # cloth_env.py:862, :917, :968
        for e in idx:                                                 # scripts pre-drawn for step_many: re-drawn below
            self._drop_pending(e)
        # ---- Cloth(...) construction: RNG draws in the reference's order (cloth.pyx:75, :101) --------------
        if tier == 2:
            pos_all = np.empty((len(idx), P, 3))
            rest_all = np.empty((len(idx), self.batch.S))
            for k, e in enumerate(idx):
                rng = self.np_randoms[e]
                self.init_side[e] = rng.rand() > 0.5
                draws = rng.rand(P)                                   # one rand() per point, r-major
                pos_all[k], rest_all[k] = self.batch.init_grid(2, self.init_side[e], draws)
                if e == 0:
                    self._built_pos0 = pos_all[k].copy()              # what env 0's cloth was built with (Point.orig_*, the façade's view)
            zeros_pin = np.zeros((len(idx), P), dtype=np.uint8)
            k = 0
            while k < len(idx):                                       # one upload per run of consecutive envs
                j = k
                while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1:
                    j += 1
                self.batch.set_state(pos_all[k:j + 1], pos_all[k:j + 1], zeros_pin[k:j + 1], rest_all[k:j + 1],
                                     env0=int(idx[k]), n=j + 1 - k, rest_shared=False)
                k = j + 1
        else:
            for e in idx:
                self.init_side[e] = self.np_randoms[e].rand() > 0.5
            self.batch.reset_flat(None if len(idx) == E else m)       # flat grid, nothing pinned, no tear: on the device
            if m[0]:
                self._built_pos0 = None                                # the flat grid (ClothEnv.reset asks init_grid for it)
        self.num_steps[m] = 0
        self.num_sim_steps[m] = 0
        self.have_tear[m] = False
        self._ep_done[m] = False
        self._iters_up_env[m] = float(self.iters_up)
        self._reset_actions(m, idx, tier)
        cov, vinv, _, _ = self.batch.metrics()                        # cloth_env.py:780-782
        self._prev_reward = np.where(m, cov, self._prev_reward)
        self._start_coverage = np.where(m, cov, self._start_coverage)
        self._start_variance_inv = np.where(m, vinv, self._start_variance_inv)
        self._current_coverage = np.where(m, 0.0, self._current_coverage)
        if self._consume_domrand:                                     # cloth_env.py:786-789 advance the env RNG
            for e in idx:
                rng = self.np_randoms[e]
                rng.uniform(low=40, high=50)
                rng.uniform(low=0.7, high=1.3)
                lim = rng.uniform(low=-15.0, high=15.0)
                rng.uniform(low=-lim, high=lim, size=(self._wd, self._hd, 3))
        return self.state

    @staticmethod
    def _randval_minabs(rng, low, high, minabs=None):                 # cloth_env.py:824-832
        val = rng.uniform(low=low, high=high)
        if minabs is not None:
            assert minabs > 0, minabs
            assert low < -minabs or high > minabs
            while np.abs(val) < minabs:
                val = rng.uniform(low=low, high=high)
        return val

    @staticmethod
    def _prevent_oob(val, dval, lower=0.0, upper=1.0):                # cloth_env.py:834-840
        if val + dval < lower:
            dval = lower - val
        elif val + dval > upper:
            dval = upper - val
        return dval

    def _reset_actions(self, m, idx, tier):
        E = self.E
        acts = np.zeros((E, 4))
        if tier == 1:                                                 # cloth_env.py:843-891
            lim = 0.20
            for pull in range(3):
                if pull < 2:
                    who = idx
                else:
                    cov = self.batch.metrics()[0]
                    who = np.array([e for e in idx if cov[e] >= 0.90], dtype=np.int64)
                    if len(who) == 0:
                        break
                pos = self.batch.positions()
                sel = np.zeros(E, dtype=bool)
                for e in who:
                    rng = self.np_randoms[e]
                    p = pos[e, rng.randint(self.P)]
                    dx0 = self._randval_minabs(rng, -lim, lim, 0.08)
                    dy0 = self._randval_minabs(rng, -lim, lim, 0.08)
                    dx0 = self._prevent_oob(p[0], dx0)
                    dy0 = self._prevent_oob(p[1], dy0)
                    acts[e] = (p[0], p[1], dx0, dy0)
                    sel[e] = True
                self.step(self._convert_action_to_clip_space(acts), initialize=True, active=sel)
        elif tier == 2:                                               # cloth_env.py:893-949
            self._update_masked(1500, m)
            pos = self.batch.positions()
            choice = {}
            sel = np.zeros(E, dtype=bool)
            for e in idx:
                rng = self.np_randoms[e]
                side = 1 if self.init_side[e] else -1
                pi = -25 if rng.rand() < 0.5 else -1
                choice[e] = pi
                p0 = pos[e, pi]
                dx0 = rng.uniform(0.30, 0.50) * side
                dy0 = rng.uniform(0.30, 0.60) if pi == -25 else rng.uniform(-0.60, -0.30)
                acts[e] = (p0[0], p0[1], dx0, dy0)
                sel[e] = True
            self.step(self._convert_action_to_clip_space(acts), initialize=True, active=sel)
            pos = self.batch.positions()
            for e in idx:
                rng = self.np_randoms[e]
                side = 1 if self.init_side[e] else -1
                if choice[e] == -25:
                    p1 = pos[e, -19]
                    dx1 = rng.uniform(0.30, 0.60) * side
                    dy1 = rng.uniform(-0.30, -0.60)
                else:
                    p1 = pos[e, -7]
                    dx1 = rng.uniform(0.30, 0.60) * side
                    dy1 = rng.uniform(0.30, 0.60)
                acts[e] = (p1[0], p1[1], dx1, dy1)
            self.step(self._convert_action_to_clip_space(acts), initialize=True, active=sel)
            self._update_masked(500, m)
        elif tier == 3:                                               # cloth_env.py:951-982
            lim = 0.25
            sel = np.zeros(E, dtype=bool)
            for e in idx:
                rng = self.np_randoms[e]
                self._iters_up_env[e] = rng.uniform(low=200, high=280)
                p0x = self._randval_minabs(rng, 0.30, 0.70)
                p0y = self._randval_minabs(rng, 0.30, 0.70)
                dx0 = self._randval_minabs(rng, -lim, lim, 0.10)
                dy0 = self._randval_minabs(rng, -lim, lim, 0.10)
                dx0 = self._prevent_oob(p0x, dx0)
                dy0 = self._prevent_oob(p0y, dy0)
                acts[e] = (p0x, p0y, dx0, dy0)
                sel[e] = True
            self.step(self._convert_action_to_clip_space(acts), initialize=True, active=sel)
            self._update_masked(800, m)
            self._iters_up_env[m] = float(self.iters_up)


class ClothEnv(object):
    """The reference's single-environment API (cloth_env.py:55): same constructor signature, seed/reset/
    step/state/get_random_action, same return types (scalars, dict of scalars)."""
    metadata = {'render.modes': ['human']}

    def __init__(self, cfg_file, subrank=None, start_state_path=None, device=0, precision='f32'):
        self._vec = ClothVecEnv(cfg_file, n_envs=1, device=device, precision=precision)
        v = self._vec
        self.cfg, self.cfg_file = v.cfg, cfg_file
        self._logger_idx = subrank
        self._start_state = None
        if start_state_path is not None:                              # npz instead of the reference's pickle
            with np.load(start_state_path) as d:
                self._start_state = {k: d[k] for k in d.files}
        for k in ('max_actions', 'iters_up', 'iters_up_rest', 'iters_pull_max', 'iters_grip_rest', 'iters_rest',
                  'reduce_factor', 'grip_radius', 'bounds', 'reward_type', 'num_w', 'num_h', 'num_points',
                  'observation_space', 'action_space', 'obslow', 'obshigh'):
            setattr(self, k, getattr(v, k))
        from .physics import Cloth, Gripper
        self.cloth = Cloth(batch=v.batch, env=0, owner=v)
        self.gripper = Gripper(self.cloth, self.grip_radius, self.cfg['cloth']['height'],
                               self.cfg['cloth']['thickness'])

    def close(self):
        self._vec.close()

    @property
    def np_random(self):
        return self._vec.np_randoms[0]

    def seed(self, seed=None):
        return self._vec.seed([seed])

    @property
    def state(self):
        return self._vec.state[0]

    @property
    def num_steps(self):
        return int(self._vec.num_steps[0])

    @property
    def num_sim_steps(self):
        return int(self._vec.num_sim_steps[0])

    @property
    def have_tear(self):
        return bool(self._vec.have_tear[0])

    def reset(self):
        self.cloth._invalidate()
        if self._start_state is not None:                             # cloth_env.py:736-741, :771-772
            s = self._start_state
            v = self._vec
            rng = v.np_randoms[0]
            v.init_side[0] = rng.rand() > 0.5                          # Cloth(state=...) still draws init_side, cloth.pyx:75
            self.cloth.init_side = bool(v.init_side[0])
            v.batch.set_state(s['pos'][None], s['prev'][None], s['pinned'][None],
                              s['rest'] if 'rest' in s else None, rest_shared=True if 'rest' in s else None)
            v.num_steps[:] = 0; v.num_sim_steps[:] = 0; v.have_tear[:] = False
            self.cloth._rebuilt(s['pos'])                              # the new Cloth's springs / colour set / orig_* 
            cov, vinv, _, _ = v.batch.metrics()
            v._prev_reward[:] = cov; v._start_coverage[:] = cov; v._start_variance_inv[:] = vinv
            v._current_coverage[:] = 0.0
            if v._consume_domrand:                                     # cloth_env.py:786-789, as in the normal reset
                rng.uniform(low=40, high=50)
                rng.uniform(low=0.7, high=1.3)
                lim = rng.uniform(low=-15.0, high=15.0)
                rng.uniform(low=-lim, high=lim, size=(v._wd, v._hd, 3))
            return self.state
        obs = self._vec.reset()[0]
        self.cloth.init_side = bool(self._vec.init_side[0])
        built = getattr(self._vec, "_built_pos0", None)
        self.cloth._rebuilt(built if built is not None else self._vec.batch.init_grid(1)[0])
        return obs

    def step(self, action, initialize=False):
        self.cloth._invalidate()
        out = self._vec.step(np.asarray(action, dtype=np.float64)[None], initialize=initialize)
        if out is None:
            return None
        obs, rew, done, info = out
        info1 = {k: (v[0].item() if hasattr(v[0], 'item') else v[0]) for k, v in info.items()}
        return obs[0], float(rew[0]), bool(done[0]), info1

    def get_random_action(self, atype='over_xy_plane'):
        if atype == 'over_xy_plane':
            return self.action_space.sample()
        raise ValueError(atype)

    def save_state(self, cloth_file):
        """cloth_env.py:343-350 (npz of SoA arrays instead of a pickle of Python objects)."""
        pos, prev, pin = self._vec.batch.get_state()
        np.savez(cloth_file, pos=pos[0], prev=prev[0], pinned=pin[0], rest=self._vec.batch.get_rest(0, 1)[0])

    def _compute_coverage(self):
        return float(self._vec._compute_coverage()[0])

    def _compute_variance(self):
        return float(self._vec._compute_variance()[0])

    def _out_of_bounds(self):
        return bool(self._vec._out_of_bounds()[0])

    def _convert_action_to_clip_space(self, a):
        return tuple(self._vec._convert_action_to_clip_space(np.asarray(a, dtype=np.float64)))

    def render(self, *a, **k):
        """The OpenGL viewer / Blender renderer are out of scope (components 7, 13)."""
        return None
