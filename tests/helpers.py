"""Shared test helpers: adapters that let the golden op lists drive either the oracle or the HIP path."""
import numpy as np


class BatchReplay(object):
    """Drives env `env` of a ClothBatch (all envs receive the same ops) with the op vocabulary of
    tests/golden/make_golden.py::Trace."""

    def __init__(self, batch):
        self.b = batch

    def grab_top(self, x, y):
        return self.b.grab_top([x, y])

    def grab(self, x, y):
        return self.b.grab([x, y])

    def release(self):
        self.b.release()

    def pin(self, i):
        for e in range(self.b.E):
            self.b.pin_points(e, [i])

    def update(self, n):
        self.b.update(n)

    def adjust_update(self, dx, dy, dz, n):
        self.b.update(n, delta=[dx, dy, dz])


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


class DeviceBuffer(object):
    """A raw device allocation through the HIP runtime that libclothhip already brought into the process (ctypes,
    no torch: the GPU tests must not depend on how long `import torch` takes on a fresh box)."""

    def __init__(self, nbytes):
        import ctypes as C
        from gym_cloth_amd import _lib
        _lib.load()
        hip = None
        for name in (None, "libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                h = C.CDLL(name)
                h.hipMalloc
                hip = h
                break
            except (OSError, AttributeError):
                continue
        assert hip is not None, "HIP runtime not found in the process"
        self._hip, self._C, self.nbytes = hip, C, int(nbytes)
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipFree.argtypes = [C.c_void_p]
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), self.nbytes) == 0
        self.ptr = p.value

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        assert self._hip.hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, 1) == 0          # hipMemcpyHostToDevice

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        assert self._hip.hipDeviceSynchronize() == 0
        assert self._hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0      # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            self._hip.hipFree(self.ptr)
            self.ptr = None
