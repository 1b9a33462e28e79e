"""Demonstration writer: the data-collection loop of the reference's examples/analytic.py:825-910 (`run(args, policy)`:
reset, act until done, append obs / act / rew / done / info per step, pickle the list of episodes) over a ClothVecEnv.

With the oracle-corner policy the whole loop -- policy, steps, episode resets -- runs on the device
(ClothVecEnv.step_many(policy='oracle_corner')): the host only cuts the per-slot records into episodes. Any other
policy object (gym_cloth_amd/policies.py, or anything with get_action(obs, t)) is driven through ClothVecEnv.step.

An episode has the reference's layout (analytic.py:866-882):
    {'obs': [obs_0 (what reset() returned), obs_1, ...], 'act': [...], 'rew': [...], 'done': [...], 'info': [dict, ...]}
with the '1d' observation (cloth_env.py:196-200) as float32[3P] on the device path (float64 on the host path).
"""
import pickle

import numpy as np

_INFO_KEYS = ('num_steps', 'num_sim_steps', 'actual_coverage', 'start_coverage', 'variance_inv', 'start_variance_inv',
              'have_tear', 'out_of_bounds')


def _new_episode(obs0, env_index):
    return {'obs': [obs0], 'act': [], 'rew': [], 'done': [], 'info': [], 'env': int(env_index)}


def _info_at(src, t, e):
    out = {}
    for k in _INFO_KEYS:
        v = src[k][t, e] if t is not None else src[k][e]
        out[k] = v.item() if hasattr(v, 'item') else v
    return out


def collect_demos(env, policy='oracle_corner', max_episodes=10, slots_per_launch=12, path=None, time_budget_ms=0.0,
                  on_device=False):
    """Run `policy` until `max_episodes` episodes have finished (over all envs of `env`, in order of completion) and return
    them as a list of episode dicts; with `path` the list is also pickled there (analytic.py:900-901).
    `env` must have been seeded; it is reset here. A policies.HighestPointPolicy with on_device=True is evaluated in the
    kernel as well: its per-env pick streams are drawn here and handed to the launch slot by slot."""
    episodes = []
    E = env.E
    obs = env.reset()
    from .policies import HighestPointPolicy
    hp = policy if (on_device and isinstance(policy, HighestPointPolicy)) else None
    if isinstance(policy, str) or hp is not None:
        if hp is None and policy != 'oracle_corner':
            raise ValueError(policy)
        cur = [_new_episode(obs[e].astype(np.float32), e) for e in range(E)]
        picks = [[] for _ in range(E)]                                # highest point: picks drawn but not consumed yet
        while len(episodes) < max_episodes:
            if hp is not None:
                for e in range(E):
                    while len(picks[e]) < slots_per_launch:
                        picks[e].append(hp.draw(e))
                tbl = np.array([[picks[e][t] for e in range(E)] for t in range(slots_per_launch)], dtype=np.int32)
                out = env.step_many(policy='highest_point', n_actions=slots_per_launch, policy_choices=tbl, auto_reset=True,
                                    want_obs=True, time_budget_ms=time_budget_ms)
                for e in range(E):
                    del picks[e][:int(out['ran'][:, e].sum())]
            else:
                out = env.step_many(policy='oracle_corner', n_actions=slots_per_launch, auto_reset=True, want_obs=True,
                                    time_budget_ms=time_budget_ms)
            for t in range(slots_per_launch):
                for e in np.nonzero(out['ran'][t])[0]:
                    k = int(out['reset_before'][t, e])
                    if k:                                             # a new episode started right before this action
                        cur[e] = _new_episode(out['reset_obs'][e, k - 1].copy(), e)
                    ep = cur[e]
                    ep['obs'].append(out['obs_t'][t, e].copy())
                    ep['act'].append(tuple(out['actions'][t, e]))
                    ep['rew'].append(float(out['rew'][t, e]))
                    ep['done'].append(bool(out['done'][t, e]))
                    ep['info'].append(_info_at(out, t, e))
                    if ep['done'][-1]:
                        episodes.append(ep)
            # a time slice can end right after a completed reset: no action of this launch carries its reset_before mark, the
            # next launch's first action belongs to the NEW episode (analytic.py:866-882: every episode opens with reset()'s obs)
            for e in np.nonzero(out.get('tail_reset_index', np.zeros(E, dtype=np.int64)))[0]:
                cur[e] = _new_episode(out['reset_obs'][e, int(out['tail_reset_index'][e]) - 1].copy(), e)
    else:
        cur = [_new_episode(obs[e].copy(), e) for e in range(E)]
        steps = np.zeros(E, dtype=np.int64)
        while len(episodes) < max_episodes:
            act = np.asarray(policy.get_action(obs, t=int(steps.max())), dtype=np.float64)
            obs, rew, done, info = env.step(act, auto_reset=True)
            last = info.get('terminal_observation', obs)
            for e in range(E):
                ep = cur[e]
                ep['obs'].append(last[e].copy())
                ep['act'].append(tuple(act[e]))
                ep['rew'].append(float(rew[e]))
                ep['done'].append(bool(done[e]))
                ep['info'].append(_info_at(info, None, e))
                steps[e] += 1
                if done[e]:
                    episodes.append(ep)
                    cur[e] = _new_episode(obs[e].copy(), e)
                    steps[e] = 0
    episodes = episodes[:max_episodes]
    if path is not None:
        with open(path, 'wb') as fh:
            pickle.dump(episodes, fh)
    return episodes
