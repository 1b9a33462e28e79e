"""GPU test of the RCCL transport (gym_cloth_amd/rccl.py through dist.RcclTransport): a one-rank communicator on the box's
GPU exercises the real ncclCommInitRank / ncclBroadcast / ncclAllGather / ncclAllReduce bindings, the unique-id file
rendezvous and the device staging buffers on the cloth handle's stream. (More ranks need more GPUs: the world_size-2 path
is covered on CPU by tests/test_dist_sockets.py over the same StepExchange.)"""
import os

import numpy as np
import pytest

from test_gpu_env import base_cfg

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_collectives(tmp_path):
    from gym_cloth_amd import ClothBatch
    from gym_cloth_amd.dist import RcclTransport, StepExchange
    b = ClothBatch(base_cfg("tier1", 1), n_envs=4, precision="f32")
    path = str(tmp_path / "rccl.id")
    t = RcclTransport(0, 1, b, rdzv_path=path)
    assert not os.path.exists(path), "rank 0 removes the rendezvous file after the first collective"
    ex = StepExchange(4, t)
    a = np.random.RandomState(3).uniform(-1, 1, size=(4, 4))
    assert np.array_equal(ex.broadcast_actions(a), a)
    tbl = np.random.RandomState(4).uniform(-1, 1, size=(3, 4, 4))
    assert np.array_equal(ex.broadcast_actions(tbl, n_actions=3), tbl)
    res = ex.gather_results(a[:, 0], a[:, 1], a[:, 2], a[:, 3])
    assert np.array_equal(res, a)
    obs = np.arange(4 * 6, dtype=np.float32).reshape(4, 6)
    assert np.array_equal(ex.gather_obs(obs), obs)
    assert ex.max_over_ranks(2.5) == 2.5 and ex.sum_over_ranks(1.25) == 1.25
    ex.barrier()
    # a device-resident action table broadcast in place, then consumed by the fused launch without touching the host
    from gym_cloth_amd.envs import ClothVecEnv
    v = ClothVecEnv(base_cfg("tier1", 9), n_envs=4, precision="f32")
    v.seed(9); v.reset()
    w = ClothVecEnv(base_cfg("tier1", 9), n_envs=4, precision="f32")
    w.seed(9); w.reset()
    acts = np.random.RandomState(5).uniform(-1, 1, size=(2, 4, 4))
    tv = RcclTransport(0, 1, v.batch, rdzv_path=path)
    d = v.batch.device_alloc(acts.nbytes)
    v.batch.device_upload(d, acts)
    tv.broadcast_device(d, acts.nbytes)
    o1 = v.step_many(n_actions=2, actions_device_ptr=d, auto_reset=False)
    o2 = w.step_many(acts, auto_reset=False)
    assert np.array_equal(o1["rew"], o2["rew"]) and np.array_equal(o1["obs"], o2["obs"])
    v.batch.device_free(d)
    # the bench's exchange of an episode launch: rank-major action blocks broadcast in place, the kernel's per-env summary
    # all-gathered straight from the handle's device table
    xv = StepExchange(4, tv)
    blocks = np.random.RandomState(6).uniform(-1, 1, size=(1, 2, 4, 4))               # [world, slots, E, 4]
    host_blk, d_blk = xv.broadcast_action_blocks(blocks, 2, v.batch)
    assert host_blk is None and d_blk
    o3 = v.step_many(n_actions=2, actions_device_ptr=d_blk, auto_reset=False)
    o4 = w.step_many(blocks[0], auto_reset=False)
    assert np.array_equal(o3["rew"], o4["rew"]) and np.array_equal(o3["actions"], o4["actions"])
    summ = xv.gather_summary(v.batch)
    assert summ.shape == (4, 4)
    assert np.array_equal(summ[:, 0], o3["ran"].sum(axis=0)) and np.array_equal(summ[:, 1] != 0, v._ep_done)
    assert np.array_equal(summ[:, 3], o3["executed"].sum(axis=0))
    last = o3["actual_coverage"][-1]
    ran_any = o3["ran"].any(axis=0)
    assert np.array_equal(summ[ran_any, 2], last[ran_any])
    # the same through the host transport (what world size 1 / the TCP transport use)
    from gym_cloth_amd.dist import LocalTransport
    xl = StepExchange(4, LocalTransport())
    hb, db = xl.broadcast_action_blocks(blocks, 2)
    assert db is None and np.array_equal(hb, blocks[0])
    assert np.array_equal(xl.gather_summary(v.batch), summ, equal_nan=True)
    tk, sb = v.batch.op_ticks()
    assert tk.shape == (4, 4) and sb[:, 0].sum() == o3["executed"].sum() and tk[:, 0].sum() > 0
    tv.close(); t.close(); b.close(); v.close(); w.close()


def test_rccl_unique_id_file_round_trip_and_comm_count(tmp_path, monkeypatch):
    """The REAL 128-byte ncclUniqueId (binary: it holds NUL bytes early on) written by rank 0 must come back whole for the ranks that
    poll for it -- a world size > 1 cannot be run on this box, but this is exactly what its ranks 1.. do; and ncclCommCount of the
    one-rank communicator (the bench prints it as config.rccl_nranks)."""
    import ctypes as C
    from gym_cloth_amd import ClothBatch, rccl
    from gym_cloth_amd.dist import RcclTransport
    monkeypatch.setenv("CLOTHHIP_RDZV_NONCE", "roundtrip")
    b = ClothBatch(base_cfg("tier1", 1), n_envs=2, precision="f32")         # (brings the HIP runtime up)
    path = str(tmp_path / "pair.id")
    rccl._generation.pop(path, None)
    uid0, _ = rccl.exchange_unique_id(0, 2, path)
    raw0 = C.string_at(C.byref(uid0), rccl.NCCL_UNIQUE_ID_BYTES)
    assert os.path.getsize(path) == len(rccl._MAGIC) + 16 + 4 + rccl.NCCL_UNIQUE_ID_BYTES
    rccl._generation.pop(path, None)
    uid1, _ = rccl.exchange_unique_id(1, 2, path, timeout_s=10.0)
    assert C.string_at(C.byref(uid1), rccl.NCCL_UNIQUE_ID_BYTES) == raw0
    t = RcclTransport(0, 1, b, rdzv_path=str(tmp_path / "one.id"))
    assert t.comm.nranks == 1
    t.close(); b.close()


def test_bench_two_ranks_on_one_gpu_over_tcp(tmp_path):
    """bench.py's multi-rank path end to end with REAL kernels: two ranks (two processes, both on this box's one GPU, the exchange forced
    onto the TCP transport because two ranks on one device cannot form an RCCL communicator) run the fused workload on 64 cloths each.
    Checks what a 2-GPU run's line must satisfy whatever the transport: one JSON line from rank 0, n_gpus 2, the env blocks of both ranks
    counted (env steps, substeps), `value` = the actions' substeps of BOTH ranks over the actions' share of the max-over-ranks clock, and
    the per-rank streams sharded by global env index (rank 1's envs are envs 64..127 of a single-process run)."""
    import json
    import socket
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # two free ports, both held until both are known: MASTER_PORT and the one the TCP transport binds (bench.tcp_port)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    s2 = socket.socket(); s2.bind(("127.0.0.1", 0)); tcp_port = s2.getsockname()[1]
    s.close(); s2.close()
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "64", "--steps", "4", "--warmup", "2", "--fuse", "2",
            "--no-extra", "--no-cpu-baseline"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CLOTH_BENCH_FORCE_TCP="1", CLOTH_BENCH_TCP_PORT=str(tcp_port), CLOTHHIP_RDZV_FILE=str(tmp_path / "unused.id"))
        procs.append(subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-2000:] for o in outs]
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")]
    rec = json.loads(lines[0])
    c = rec["config"]
    assert rec["n_gpus"] == 2 and c["envs_per_gpu"] == 64 and "TCP" in c["transport"] and c["rccl_nranks"] is None
    assert rec["cpu_baseline"].startswith("skipped")
    assert 2 * 64 * 2 <= c["env_steps_executed"] <= 2 * 64 * 16           # both ranks' envs are counted (about 4 steps per env)
    assert 0.2 < c["action_time_frac"] <= 1.0 and rec["value"] > 0 and c["blended_substeps_per_s"] > 0
    # value x (the actions' share of the clock) = action substeps of the whole job
    n_act = c["action_substeps_per_env_step"] * c["env_steps_executed"]
    # (the printed line carries six significant digits)
    assert abs(rec["value"] * c["timed_region_s"] * c["action_time_frac"] - n_act) <= 1e-4 * n_act
    assert abs(rec["ms_per_step"] * c["steps_equivalent"] - 1e3 * c["timed_region_s"] * c["action_time_frac"]) <= 1e-4 * 1e3 * c["timed_region_s"]
    assert rec["roofline"]["traffic"] is None or rec["roofline"]["traffic_source"]
