"""ctypes binding of RCCL (librccl.so) for the sharded vector env: no torch, no MPI.

One process per GPU. The communicator is created with ncclCommInitRank; the 128-byte ncclUniqueId travels from
rank 0 to the other ranks of the node through a file (single node by construction: xGMI only). All collectives are
enqueued on the cloth handle's own HIP stream, so they are ordered with the stepper launches without extra events.

Only the three collectives the data path needs are bound, plus all-reduce for the bench's
max-over-ranks clock and barrier:
    ncclBroadcast   action / schedule tables, rank 0 -> all
    ncclAllGather   per-env result records and (optionally) '1d' observations
    ncclAllReduce   scalars (max / sum), barrier
"""
import ctypes as C
import os
import time

NCCL_UNIQUE_ID_BYTES = 128
# ncclDataType_t / ncclRedOp_t (nccl.h)
UINT8, INT32, INT64, FLOAT32, FLOAT64 = 1, 2, 4, 7, 8
SUM, PROD, MAX, MIN = 0, 1, 2, 3


class RcclError(RuntimeError):
    pass


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def load():
    """dlopen librccl.so (ROCm's NCCL). Raises RcclError when it is not installed: there is no fallback transport on
    the GPU path."""
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for name in (os.environ.get("CLOTHHIP_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if not name:
            continue
        try:
            L = C.CDLL(name)
            break
        except OSError as e:
            err = e
    else:
        raise RcclError("librccl.so not found: %s" % err)
    vp = C.c_void_p
    L.ncclGetErrorString.restype = C.c_char_p
    L.ncclGetErrorString.argtypes = [C.c_int]
    L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
    L.ncclCommDestroy.argtypes = [vp]
    L.ncclCommCount.argtypes = [vp, C.POINTER(C.c_int)]
    L.ncclBroadcast.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    L.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RcclError("%s failed: %s" % (what, load().ncclGetErrorString(rc).decode("utf8", "replace")))


def _private_dir():
    """A directory only this user can write to (0700, owned by us, not a symlink) under TMPDIR: the default home of the id file
    when no launcher handed one over, so that nobody else on the box can plant or pre-create it."""
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "clothhip_rdzv_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RcclError("rendezvous directory %s is not a private directory of this user (mode %o, uid %d); set "
                        "CLOTHHIP_RDZV_FILE to a path in one" % (d, st.st_mode & 0o7777, st.st_uid))
    return d


def rendezvous_path():
    """Where rank 0 leaves the unique id. CLOTHHIP_RDZV_FILE wins (bench.py's own launcher sets it, inside a fresh 0700
    directory); under torch.distributed.run the ranks share MASTER_PORT, the run id and their parent (the agent process), and the
    file lives in a per-user 0700 directory."""
    p = os.environ.get("CLOTHHIP_RDZV_FILE")
    if p:
        return p
    key = "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
    return os.path.join(_private_dir(), "clothhip_rccl_%s.id" % key)


_MAGIC = b"CLTHRCCL"
_generation = {}                   # rendezvous path -> communicators created on it by this process so far


def _nonce(path, world):
    """16 bytes every rank of ONE communicator derives identically and no other communicator shares: the launcher's random
    CLOTHHIP_RDZV_NONCE (bench.py's own launcher) or the launcher-wide key of rendezvous_path(), plus how many communicators
    this process has already built on that path (ranks build them in the same order)."""
    import hashlib
    gen = _generation.get(path, 0)
    _generation[path] = gen + 1
    base = os.environ.get("CLOTHHIP_RDZV_NONCE") or "%s|%s|%s" % (os.environ.get("MASTER_ADDR", ""), os.environ.get("MASTER_PORT", ""),
                                                                 os.environ.get("TORCHELASTIC_RUN_ID", ""))
    return hashlib.sha256(("%s|%s|%d|%d" % (base, path, int(world), gen)).encode()).digest()[:16]


def exchange_unique_id(rank, world, path=None, timeout_s=300.0, make_id=None):
    """rank 0 creates the id and publishes it atomically (exclusive temp file + rename) behind a header of
    (magic, nonce, world); the others poll for a file that carries THEIR nonce -- a file left by a crashed run or by the previous
    communicator on the same path is ignored (and replaced by rank 0).
    `make_id(uid)`: fills the ncclUniqueId structure on rank 0 (default: ncclGetUniqueId, which needs a GPU); the CPU test suite
    passes a stand-in so that BOTH ranks' real code paths -- structure, header, file protocol, polling -- run in two processes."""
    path = path or rendezvous_path()
    nonce = _nonce(path, world)
    head = _MAGIC + nonce + int(world).to_bytes(4, "little")
    uid = _UniqueId()
    if rank == 0:
        if make_id is None:
            _check(load().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        else:
            make_id(uid)
        try:
            os.remove(path)                                   # whatever an earlier run left there
        except OSError:
            pass
        import tempfile
        fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".tmp", dir=os.path.dirname(path) or ".")   # unique, 0600, O_EXCL:
        try:                                                                   # a leftover of a crashed run (reused pid) cannot collide
            with os.fdopen(fd, "wb") as fh:
                fh.write(head + C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES))    # (NOT uid.internal: ctypes cuts a c_char array field at its first NUL)
            os.replace(tmp, path)
        except BaseException:
            try:
                os.remove(tmp)
            except OSError:
                pass
            raise
    else:
        t0 = time.time()
        while True:
            try:
                # never through a symlink, and only a regular file of OUR user counts (a foreign file with the right header is ignored)
                fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                try:
                    import stat as _stat
                    st = os.fstat(fd)
                    raw = os.read(fd, len(head) + NCCL_UNIQUE_ID_BYTES + 1) if (_stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid()) else b""
                finally:
                    os.close(fd)
                if len(raw) == len(head) + NCCL_UNIQUE_ID_BYTES and raw[:len(head)] == head:
                    raw = raw[len(head):]
                    break
            except OSError:
                pass
            if time.time() - t0 > timeout_s:
                raise RcclError("rank %d: no RCCL unique id for this communicator at %s after %.0f s" % (rank, path, timeout_s))
            time.sleep(0.01)
        C.memmove(C.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
    return uid, path


class Communicator(object):
    """One RCCL communicator bound to a HIP stream (an opaque hipStream_t pointer value)."""

    def __init__(self, rank, world, stream, rdzv_path=None):
        self._L = load()
        self.rank, self.world, self.stream = int(rank), int(world), C.c_void_p(stream)
        uid, self._path = exchange_unique_id(self.rank, self.world, rdzv_path)
        comm = C.c_void_p()
        _check(self._L.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self._comm = comm

    @property
    def nranks(self):
        """ncclCommCount: how many ranks RCCL itself says this communicator has (the bench line prints it)."""
        n = C.c_int(-1)
        _check(self._L.ncclCommCount(self._comm, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def rendezvous_done(self):
        """Call after the first collective has completed on every rank: rank 0 removes the id file."""
        if self.rank == 0:
            try:
                os.remove(self._path)
            except OSError:
                pass

    def broadcast(self, d_buf, nbytes, root=0):
        _check(self._L.ncclBroadcast(C.c_void_p(d_buf), C.c_void_p(d_buf), int(nbytes), UINT8, int(root), self._comm,
                                     self.stream), "ncclBroadcast")

    def allgather(self, d_send, d_recv, nbytes_per_rank):
        _check(self._L.ncclAllGather(C.c_void_p(d_send), C.c_void_p(d_recv), int(nbytes_per_rank), UINT8, self._comm,
                                     self.stream), "ncclAllGather")

    def allreduce_f64(self, d_buf, count, op):
        _check(self._L.ncclAllReduce(C.c_void_p(d_buf), C.c_void_p(d_buf), int(count), FLOAT64, int(op), self._comm,
                                     self.stream), "ncclAllReduce")

    def close(self):
        if getattr(self, "_comm", None):
            self._L.ncclCommDestroy(self._comm)
            self._comm = None
