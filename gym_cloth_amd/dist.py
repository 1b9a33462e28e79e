"""Multi-GPU data path: one process per GPU, env blocks sharded across ranks.

Cloth instances never interact, so the only exchange steps are
  * broadcast of the action table float64[T][world*E][4] from rank 0 (what a central policy produces), and
  * all-gather of per-env result records and, optionally, of the '1d' observations (cloth_env.py:196-200)
    float32[world*E][3P].
There is no all-reduce on the data path (the bench uses one for its max-over-ranks clock and as barrier).

Two transports behind one interface:
  RcclTransport    RCCL over xGMI, bound directly through ctypes (rccl.py); buffers are device allocations of the
                   cloth handle, collectives run on the handle's HIP stream. The GPU path.
  SocketTransport  plain TCP star through rank 0, host buffers. For the CPU tests (world_size 2) only: there is
                   no GPU in the development container.
No torch anywhere.
"""
import os
import socket
import struct
import time

import numpy as np


def shard_range(rank, world, envs_per_rank):
    """Contiguous env block of `rank`: [g0, g1) in global env indices."""
    g0 = rank * envs_per_rank
    return g0, g0 + envs_per_rank


def env_from_launcher():
    """(rank, local_rank, world) from the environment torch.distributed.run / bench.py's own launcher set up."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


# ---------------------------------------------------------------------------------------------------------------
class SocketTransport(object):
    """Collectives over TCP through rank 0 (host memory). Test transport."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29611, timeout_s=120.0):
        self.rank, self.world = int(rank), int(world)
        self._peers, self._sock = [], None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout_s)
            peers = {}
            while len(peers) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack("<i", self._recv_exact(c, 4))[0]
                peers[r] = c
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=timeout_s)
                    break
                except OSError:
                    if time.time() - t0 > timeout_s:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack("<i", self.rank))
            self._sock = s

    @staticmethod
    def _recv_exact(s, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = s.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("peer closed the connection")
            buf.extend(chunk)
        return bytes(buf)

    def _send_msg(self, s, raw):
        s.sendall(struct.pack("<q", len(raw)) + raw)

    def _recv_msg(self, s):
        n = struct.unpack("<q", self._recv_exact(s, 8))[0]
        return self._recv_exact(s, n)

    def broadcast(self, arr, shape, dtype):
        """rank 0 passes `arr`; every rank gets an array of (shape, dtype)."""
        if self.rank == 0:
            a = np.ascontiguousarray(arr, dtype=dtype).reshape(shape)
            for p in self._peers:
                self._send_msg(p, a.tobytes())
            return a
        return np.frombuffer(self._recv_msg(self._sock), dtype=dtype).reshape(shape).copy()

    def allgather(self, loc):
        """loc: this rank's block [n, ...]; returns the rank-ordered concatenation on every rank."""
        loc = np.ascontiguousarray(loc)
        if self.world == 1:
            return loc.copy()
        if self.rank == 0:
            parts = [loc.tobytes()] + [self._recv_msg(p) for p in self._peers]
            raw = b"".join(parts)
            for p in self._peers:
                self._send_msg(p, raw)
        else:
            self._send_msg(self._sock, loc.tobytes())
            raw = self._recv_msg(self._sock)
        return np.frombuffer(raw, dtype=loc.dtype).reshape((self.world * loc.shape[0],) + loc.shape[1:]).copy()

    def allreduce(self, value, op):
        vals = self.allgather(np.array([float(value)]))
        return float(vals.max() if op == "max" else vals.sum())

    def barrier(self):
        self.allreduce(0.0, "sum")

    def close(self):
        for p in self._peers:
            p.close()
        if self._sock:
            self._sock.close()
        self._peers, self._sock = [], None


# ---------------------------------------------------------------------------------------------------------------
class RcclTransport(object):
    """Collectives through RCCL on the cloth handle's stream; staging buffers are device allocations of the handle
    (clothhip_device_alloc). `batch` is the rank's ClothBatch."""

    def __init__(self, rank, world, batch, rdzv_path=None):
        from . import rccl
        self.rank, self.world, self.batch = int(rank), int(world), batch
        self._bufs = {}
        self.comm = rccl.Communicator(rank, world, batch.stream, rdzv_path)
        self._rccl = rccl
        self.barrier()                               # first collective: every rank has joined
        self.comm.rendezvous_done()

    def _buf(self, key, nbytes):
        b = self._bufs.get(key)
        if b is None or b[1] < nbytes:
            if b is not None:
                self.batch.device_free(b[0])
            b = (self.batch.device_alloc(nbytes), nbytes)
            self._bufs[key] = b
        return b[0]

    def broadcast(self, arr, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        d = self._buf("bc", nbytes)
        if self.rank == 0:
            self.batch.device_upload(d, np.ascontiguousarray(arr, dtype=dtype).reshape(shape))
        self.comm.broadcast(d, nbytes, 0)
        out = np.empty(shape, dtype=dtype)
        self.batch.device_download(out, d)           # synchronises the stream
        return out

    def broadcast_device(self, d_buf, nbytes):
        """In-place broadcast of a device table (e.g. the action table a launch reads directly)."""
        self.comm.broadcast(d_buf, nbytes, 0)

    def allgather(self, loc):
        loc = np.ascontiguousarray(loc)
        ds, dr = self._buf("ag_s", loc.nbytes), self._buf("ag_r", loc.nbytes * self.world)
        self.batch.device_upload(ds, loc)
        self.comm.allgather(ds, dr, loc.nbytes)
        out = np.empty((self.world * loc.shape[0],) + loc.shape[1:], dtype=loc.dtype)
        self.batch.device_download(out, dr)
        return out

    def allgather_device(self, d_send, d_recv, nbytes_per_rank):
        self.comm.allgather(d_send, d_recv, nbytes_per_rank)

    def allreduce(self, value, op):
        d = self._buf("ar", 8)
        self.batch.device_upload(d, np.array([float(value)]))
        self.comm.allreduce_f64(d, 1, self._rccl.MAX if op == "max" else self._rccl.SUM)
        out = np.empty(1)
        self.batch.device_download(out, d)
        return float(out[0])

    def barrier(self):
        self.allreduce(0.0, "sum")

    def close(self):
        for d, _ in self._bufs.values():
            self.batch.device_free(d)
        self._bufs = {}
        self.comm.close()


class LocalTransport(object):
    """world_size 1: nothing to exchange."""
    rank, world = 0, 1

    def broadcast(self, arr, shape, dtype):
        return np.ascontiguousarray(arr, dtype=dtype).reshape(shape)

    def allgather(self, loc):
        return np.ascontiguousarray(loc).copy()

    def allreduce(self, value, op):
        return float(value)

    def barrier(self):
        pass

    def close(self):
        pass


# ---------------------------------------------------------------------------------------------------------------
class StepExchange(object):
    """Per-step collectives of the sharded vector env over any of the transports above."""

    N_RES = 4          # reward, done, coverage, executed substeps

    def __init__(self, envs_per_rank, transport):
        self.t = transport
        self.world, self.rank = transport.world, transport.rank
        self.E = int(envs_per_rank)
        self.g0, self.g1 = shard_range(self.rank, self.world, self.E)

    def broadcast_actions(self, actions_all, n_actions=None):
        """rank 0 passes float64[world*E, 4] (or, with n_actions, float64[n_actions, world*E, 4]); the other ranks
        pass None. Returns this rank's env block: [E, 4] (or [n_actions, E, 4])."""
        if n_actions is None:
            return self.t.broadcast(actions_all, (self.world * self.E, 4), np.float64)[self.g0:self.g1]
        shape = (int(n_actions), self.world * self.E, 4)
        return np.ascontiguousarray(self.t.broadcast(actions_all, shape, np.float64)[:, self.g0:self.g1])

    def gather_results(self, rew, done, coverage, executed):
        """All-gather the per-env step results; returns float64[world*E, 4] on every rank."""
        loc = np.stack([np.asarray(rew, dtype=np.float64), np.asarray(done, dtype=np.float64),
                        np.asarray(coverage, dtype=np.float64), np.asarray(executed, dtype=np.float64)], axis=1)
        return self.t.allgather(loc)

    # ---- device-resident exchange of the episode launches (RCCL transport; the others fall back to the host calls above) ----
    def broadcast_action_blocks(self, table_all, n_actions, batch=None):
        """Action table of an episode launch, rank-major: rank 0 passes float64[world, n_actions, E, 4] (the others None).
        Returns (host_block, device_ptr): with the RCCL transport the table is broadcast IN PLACE in device memory and
        device_ptr addresses this rank's block [n_actions, E, 4] -- the launch reads it directly
        (step_many(actions_device_ptr=...)), nothing is staged through the host on the receiving side; otherwise host_block is
        this rank's block as an array."""
        shape = (self.world, int(n_actions), self.E, 4)
        if hasattr(self.t, "broadcast_device") and batch is not None:
            nbytes = int(np.prod(shape)) * 8
            d = self.t._buf("act_tbl", nbytes)
            if self.rank == 0:
                batch.device_upload(d, np.ascontiguousarray(table_all, dtype=np.float64).reshape(shape))
            self.t.broadcast_device(d, nbytes)
            return None, d + self.rank * (nbytes // self.world)
        return np.ascontiguousarray(self.t.broadcast(table_all, shape, np.float64)[self.rank]), None

    def gather_summary(self, batch):
        """All-gather of the per-env summary the launch wrote on the device (ClothBatch.run_summary): float64[world*E, 4] =
        {actions executed, episode over, coverage, action substeps}. RCCL transport: ncclAllGather straight from the handle's
        table into a device buffer, one download of the result; otherwise through the host."""
        if hasattr(self.t, "allgather_device"):
            nb = self.E * 32
            d_all = self.t._buf("sum_all", nb * self.world)
            self.t.allgather_device(batch.run_summary_device_ptr, d_all, nb)
            out = np.empty((self.world * self.E, 4))
            batch.device_download(out, d_all)                 # synchronises the stream
            return out
        return self.t.allgather(batch.run_summary())

    def gather_obs(self, obs_loc):
        """All-gather of the '1d' observations float32[E, 3P] -> float32[world*E, 3P]."""
        return self.t.allgather(np.asarray(obs_loc, dtype=np.float32))

    def max_over_ranks(self, value):
        return self.t.allreduce(value, "max")

    def sum_over_ranks(self, value):
        return self.t.allreduce(value, "sum")

    def barrier(self):
        self.t.barrier()
