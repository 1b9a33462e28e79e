"""Seeding helper with the semantics of gym.utils.seeding.np_random of gym==0.12.1, the version the
reference pins (requirements.txt:5; call sites gym_cloth/envs/cloth_env.py:339, physics/cloth.pyx:73).

gym is a third-party dependency that is NOT vendored in the reference and is not installed here, so this
restates gym 0.12.1's published algorithm: the integer seed is hashed with SHA-512 of its decimal string,
the first 8 bytes are read as little-endian uint32 words, and those words seed a legacy numpy RandomState
(MT19937).  Parity with the real gym 0.12.1 is unpinned (no golden vector from gym itself is available);
the golden fixtures of this repo were generated with the same restatement, so they are self-consistent.
"""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(b):
    b += b"\0" * (4 - len(b) % 4)
    n = len(b) // 4
    acc = 0
    for i, v in enumerate(struct.unpack("{}I".format(n), b)):
        acc += 2 ** (32 * i) * v
    return acc


def create_seed(a=None, max_bytes=8):
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    if isinstance(a, (int, np.integer)):
        return int(a) % 2 ** (8 * max_bytes)
    raise ValueError("Invalid type for seed: {} ({})".format(type(a), a))


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode("utf8")).digest()[:max_bytes])


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise ValueError("Seed must be non-negative, not {}".format(bigint))
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
