"""world_size-2 test of the multi-GPU data path (gym_cloth_amd/dist.py) on CPU: env blocks sharded over ranks,
action table broadcast from rank 0, per-env results and observations all-gathered, through the plain-socket
transport (the GPU path runs the same StepExchange over RCCL, tests/test_gpu_dist.py). The physics stand-in on each
rank is the CPU oracle (tests may use it); the sharded result must equal the single-process result env for env."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _step_block(actions, g):
    """one short schedule per env on the oracle: returns (rew, done, coverage, executed) per env."""
    from oracle import pyoracle
    from gym_cloth_amd.envs import decode_actions
    out = []
    for a in actions:
        c = pyoracle.OracleCloth(g["cfg"])
        d = decode_actions(a[None], [-1.] * 4, [1.] * 4, True, True, 0.002, 5, 5, 400, 5, 10)
        n = c.grab_top(float(d["x"][0]), float(d["y"][0]))
        ex = c.run_schedule(d["bounds"][0] if n else np.zeros(5, int), 0.0025, float(d["x_dir_r"][0]),
                            float(d["y_dir_r"][0]), True)
        pos = c.get_state()[0]
        out.append((float(pos[:, 2].max()), float(c.have_tear), float(pos[:, 0].mean()), float(ex)))
    return np.array(out)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from gym_cloth_amd.dist import SocketTransport, StepExchange, shard_range
    from oracle import pyoracle
    E = 3
    ex = StepExchange(E, SocketTransport(rank, world, "127.0.0.1", port))
    assert (ex.g0, ex.g1) == shard_range(rank, world, E) == (rank * E, rank * E + E)
    g = pyoracle.load_golden("g_traj_lift_pull_25.npz")
    acts_all = np.random.RandomState(5).uniform(-1, 1, size=(world * E, 4)) if rank == 0 else None
    mine = ex.broadcast_actions(acts_all)
    multi = ex.broadcast_actions(np.arange(2 * world * E * 4, dtype=np.float64).reshape(2, world * E, 4)
                                 if rank == 0 else None, n_actions=2)
    r = _step_block(mine, g)
    res = ex.gather_results(r[:, 0], r[:, 1], r[:, 2], r[:, 3])
    obs = ex.gather_obs(np.full((E, 6), float(rank), dtype=np.float32))
    tmax = ex.max_over_ranks(1.0 + rank)
    tsum = ex.sum_over_ranks(r[:, 3].sum())
    ex.barrier()
    q.put((rank, mine, multi, res, obs, tmax, tsum))
    ex.t.close()


def test_sharded_step_equals_single_process():
    from oracle import pyoracle
    pyoracle.build()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    acts_all = np.random.RandomState(5).uniform(-1, 1, size=(6, 4))
    ref = _step_block(acts_all, pyoracle.load_golden("g_traj_lift_pull_25.npz"))
    table = np.arange(2 * 6 * 4, dtype=np.float64).reshape(2, 6, 4)
    for rank, mine, multi, res, obs, tmax, tsum in got:
        assert np.array_equal(mine, acts_all[rank * 3:rank * 3 + 3])          # broadcast + slice
        assert np.array_equal(multi, table[:, rank * 3:rank * 3 + 3])         # multi-action table
        assert np.array_equal(res, ref)                                       # all-gather, env for env
        assert np.array_equal(obs[:3], np.zeros((3, 6))) and np.array_equal(obs[3:], np.ones((3, 6)))
        assert tmax == 2.0 and tsum == ref[:, 3].sum()


def test_local_transport_is_identity():
    sys.path.insert(0, ROOT)
    from gym_cloth_amd.dist import LocalTransport, StepExchange
    ex = StepExchange(4, LocalTransport())
    a = np.random.RandomState(1).uniform(-1, 1, size=(4, 4))
    assert np.array_equal(ex.broadcast_actions(a), a)
    assert ex.max_over_ranks(3.5) == 3.5 and ex.sum_over_ranks(2.0) == 2.0
    assert ex.gather_results(a[:, 0], a[:, 1], a[:, 2], a[:, 3]).shape == (4, 4)


def test_package_does_not_import_torch():
    """north_star: no PyTorch in the stepper or its multi-GPU driver."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import gym_cloth_amd, gym_cloth_amd.envs, gym_cloth_amd.dist, "
            "gym_cloth_amd.rccl, gym_cloth_amd.policies, gym_cloth_amd.physics; "
            "assert 'torch' not in sys.modules, 'torch was imported'" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])


def test_bench_self_launcher_two_ranks_dry_run():
    """bench.py --gpus 2 without a launcher environment starts one fresh process per rank; --dry-run replaces the GPU work by a
    stand-in, so this covers what a real 2-GPU run does around it: rank / world parsing from the environment the launcher
    sets, the rendezvous file (magic + nonce + world header, private directory), the per-launch exchange pattern (rank-major
    action blocks, summary all-gather, max / sum all-reduce) over the TCP transport, and the relay of rank 0's ONE JSON line."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--envs", "5"],
                         capture_output=True, timeout=120, env={k: v for k, v in os.environ.items()
                                                               if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr.decode()
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3
    # every launch gathers 100 substeps per env of rank 0 and 200 per env of rank 1; rank 0 sums the gathered table itself
    assert rec["config"]["gathered_substeps"] == 3 * (100.0 + 200.0) * 5


def test_bench_self_launcher_returns_a_failing_ranks_code():
    """A rank that dies must take the job down with a non-zero code instead of leaving the others waiting for it."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CLOTH_BENCH_DRY_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--envs", "4"],
                         capture_output=True, timeout=120, env=env)
    assert out.returncode != 0


def test_rendezvous_file_ignores_stale_and_foreign_ids(tmp_path, monkeypatch):
    """The unique id travels behind a (magic, nonce, world) header: a file left by an earlier communicator on the same path -- or
    by a crashed run -- is not this communicator's and must be ignored by the polling ranks (a fast rank could pick
    up the previous id and hang in ncclCommInitRank)."""
    sys.path.insert(0, ROOT)
    from gym_cloth_amd import rccl
    path = str(tmp_path / "rccl.id")
    monkeypatch.setenv("CLOTHHIP_RDZV_NONCE", "abc")
    rccl._generation.pop(path, None)
    n0 = rccl._nonce(path, 2)            # first communicator on the path
    n1 = rccl._nonce(path, 2)            # second one: another nonce
    assert n0 != n1 and len(n0) == 16
    rccl._generation.pop(path, None)
    assert rccl._nonce(path, 2) == n0    # what the other rank (same order of communicators) derives
    assert rccl._nonce(path, 4) != n1    # world size is part of it
    monkeypatch.setenv("CLOTHHIP_RDZV_NONCE", "xyz")
    rccl._generation.pop(path, None)
    assert rccl._nonce(path, 2) != n0    # another launch


def test_rendezvous_reader_takes_only_a_regular_file_of_this_user(tmp_path, monkeypatch):
    """the polling ranks open the id file without following symlinks and accept it only when it is a regular file owned
    by this user; the default location is a private (0700) per-user directory."""
    sys.path.insert(0, ROOT)
    from gym_cloth_amd import rccl
    try:
        rccl.load()
    except rccl.RcclError:
        pytest.skip("librccl.so not installed")
    monkeypatch.setenv("CLOTHHIP_RDZV_NONCE", "n1")
    real, link = str(tmp_path / "real.id"), str(tmp_path / "link.id")
    rccl._generation.pop(link, None)
    head = rccl._MAGIC + rccl._nonce(link, 2) + (2).to_bytes(4, "little")
    with open(real, "wb") as fh:
        fh.write(head + bytes(range(128)))
    os.symlink(real, link)
    rccl._generation.pop(link, None)
    with pytest.raises(rccl.RcclError):                      # a symlink to a perfectly good id: not followed
        rccl.exchange_unique_id(1, 2, link, timeout_s=0.3)
    os.remove(link)
    os.rename(real, link)
    rccl._generation.pop(link, None)
    uid, _ = rccl.exchange_unique_id(1, 2, link, timeout_s=5.0)
    import ctypes as C
    assert C.string_at(C.byref(uid), 128) == bytes(range(128))
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.delenv("CLOTHHIP_RDZV_FILE", raising=False)
    d = os.path.dirname(rccl.rendezvous_path())
    assert os.stat(d).st_mode & 0o777 == 0o700 and os.stat(d).st_uid == os.getuid()
    os.chmod(d, 0o755)
    with pytest.raises(rccl.RcclError):                      # somebody else could write there: refused
        rccl.rendezvous_path()


# a stand-in for ncclGetUniqueId's output with what made round 4's bug: binary, NUL bytes early on (a real id starts with the
# little-endian magic / port words of a socket address: b"\x02\x00..." )
_FAKE_ID = bytes([2, 0, 0xb0, 0x9d, 0, 0, 0, 0, 10, 0, 0, 7]) + bytes((37 * i + 11) & 0xFF for i in range(116))


def _uid_worker(rank, world, path, nonce, q, timeout_s):
    sys.path.insert(0, ROOT)
    os.environ["CLOTHHIP_RDZV_NONCE"] = nonce
    import ctypes as C
    from gym_cloth_amd import rccl
    try:
        uid, p = rccl.exchange_unique_id(rank, world, path, timeout_s=timeout_s,
                                         make_id=lambda u: C.memmove(C.byref(u), _FAKE_ID, rccl.NCCL_UNIQUE_ID_BYTES))
        q.put((rank, C.string_at(C.byref(uid), rccl.NCCL_UNIQUE_ID_BYTES), p))
    except rccl.RcclError as exc:
        q.put((rank, None, str(exc)))


def test_unique_id_rendezvous_two_processes(tmp_path):
    """Rank 0's and rank 1's REAL exchange_unique_id code paths in two processes against one rendezvous file (the id itself is a
    stand-in: ncclGetUniqueId needs a GPU): rank 1 starts first and polls; a stale file of another communicator lies in the way and is
    ignored; the id arrives whole -- 128 bytes, NULs and all (round 4 found rank 0 writing it cut at its first NUL, so that no world > 1
    could ever have come up). A rank whose rank 0 never shows up gives up with an error after its timeout instead of hanging."""
    assert len(_FAKE_ID) == 128 and _FAKE_ID.index(0) == 1
    path = str(tmp_path / "pair.id")
    with open(path, "wb") as fh:                              # left by some earlier communicator: other nonce, full length
        fh.write(b"CLTHRCCL" + bytes(16) + (2).to_bytes(4, "little") + bytes(128))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p1 = ctx.Process(target=_uid_worker, args=(1, 2, path, "nonce-a", q, 60.0))
    p1.start()
    import time
    time.sleep(0.5)                                           # rank 1 is polling the stale file by now
    p0 = ctx.Process(target=_uid_worker, args=(0, 2, path, "nonce-a", q, 60.0))
    p0.start()
    got = dict((r, (raw, info)) for r, raw, info in (q.get(timeout=120) for _ in range(2)))
    for p in (p0, p1):
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0][0] == _FAKE_ID and got[1][0] == _FAKE_ID, (got[0][0], got[1][0])
    assert os.path.getsize(path) == 8 + 16 + 4 + 128
    # nobody publishes for this nonce: the poller must time out, with an error that names the path
    p2 = ctx.Process(target=_uid_worker, args=(1, 2, path, "nonce-b", q, 1.0))
    p2.start()
    r, raw, info = q.get(timeout=60)
    p2.join(timeout=30)
    assert r == 1 and raw is None and "no RCCL unique id" in info and path in info
