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
