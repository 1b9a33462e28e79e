"""Multi-GPU data path: one process per GPU, env blocks sharded across ranks, `torch.distributed` collectives
(backend 'nccl' = RCCL over xGMI on the GPU box; 'gloo' in the CPU tests).

Cloth instances never interact, so the only exchange steps are
  * broadcast of the action table float64[world*E][4] from rank 0 (what a central policy would produce), and
  * all-gather of per-env results (reward, done, coverage, executed substeps) and, optionally, of the '1d'
    observations (cloth_env.py:196-200) float32[world*E][3P].
There is no all-reduce on the data path. torch is used for the collectives only (plumbing).
"""
import numpy as np


def shard_range(rank, world, envs_per_rank):
    """Contiguous env block of `rank`: [g0, g1) in global env indices."""
    g0 = rank * envs_per_rank
    return g0, g0 + envs_per_rank


class StepExchange(object):
    """Per-step collectives of the sharded vector env. `device` is a torch device (cuda:k or cpu)."""

    N_RES = 4          # reward, done, coverage, executed substeps

    def __init__(self, envs_per_rank, obs_dim=0, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.E = int(envs_per_rank)
        self.device = device if device is not None else torch.device("cpu")
        self.g0, self.g1 = shard_range(self.rank, self.world, self.E)
        self.act_buf = torch.empty((self.world * self.E, 4), dtype=torch.float64, device=self.device)
        self.res_loc = torch.empty((self.E, self.N_RES), dtype=torch.float64, device=self.device)
        self.res_all = torch.empty((self.world * self.E, self.N_RES), dtype=torch.float64, device=self.device)
        self.obs_loc = self.obs_all = None
        if obs_dim:
            self.obs_loc = torch.empty((self.E, obs_dim), dtype=torch.float32, device=self.device)
            self.obs_all = torch.empty((self.world * self.E, obs_dim), dtype=torch.float32, device=self.device)

    def broadcast_actions(self, actions_all):
        """rank 0 passes float64[world*E,4] (others None); returns this rank's block as a numpy array."""
        if self.world == 1:
            return np.asarray(actions_all, dtype=np.float64)[self.g0:self.g1]
        if self.rank == 0:
            self.act_buf.copy_(self.torch.from_numpy(np.ascontiguousarray(actions_all, dtype=np.float64)))
        self.dist.broadcast(self.act_buf, src=0, group=self.group)
        return self.act_buf[self.g0:self.g1].cpu().numpy()

    def gather_results(self, rew, done, coverage, executed):
        """All-gather the per-env step results; returns float64[world*E, 4] on every rank."""
        loc = np.stack([np.asarray(rew, dtype=np.float64), np.asarray(done, dtype=np.float64),
                        np.asarray(coverage, dtype=np.float64), np.asarray(executed, dtype=np.float64)], axis=1)
        if self.world == 1:
            return loc
        self.res_loc.copy_(self.torch.from_numpy(loc))
        self.dist.all_gather_into_tensor(self.res_all, self.res_loc, group=self.group)
        return self.res_all.cpu().numpy()

    def gather_obs(self):
        """All-gather of obs_loc (filled by the caller, e.g. ClothBatch.write_obs_f32_device) -> obs_all."""
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.obs_all, self.obs_loc, group=self.group)
        else:
            self.obs_all.copy_(self.obs_loc)
        return self.obs_all

    def max_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.group)
