"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

numpy restatement of mava_amd/csrc/permutation.hip: the epoch permutation that stands where the reference calls
jax.random.permutation(key, batch_size) (mava/systems/ppo/ff_mappo.py:272-273; rec_mappo.py:277-279).  Integer work:
the kernel must match this bit for bit.  Parity unpinned against Mava's own stream (threefry, JAX absent here): what
the reference fixes is the CONTRACT - a uniformly distributed bijection of [0, n) per key - and tests/ check that
(bijection exactly; position statistics over many keys).
"""
from __future__ import annotations

import numpy as np

ROUNDS = 16
M64 = (1 << 64) - 1


def round_keys(seed: int, counter: int):
    st = (seed + 0x9E3779B97F4A7C15 * (counter + 1)) & M64
    keys = []
    for _ in range(ROUNDS):
        st = (st + 0x9E3779B97F4A7C15) & M64
        z = st
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        z ^= z >> 31
        keys.append(z >> 32)
    return keys


def _mix(x: np.ndarray, k: int) -> np.ndarray:
    h = (x + np.uint32(k)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x7FEB352D)).astype(np.uint32)
    h ^= h >> np.uint32(15)
    h = (h * np.uint32(0x846CA68B)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def permutation(n: int, seed: int, counter: int) -> np.ndarray:
    """int32 (n,): out[i] = the cycle walk of i under the keyed Feistel bijection of [0, 2^b)."""
    assert 1 <= n < (1 << 31)
    b = 2
    while (1 << b) < n:
        b += 1
    lb = b >> 1
    rb = b - lb
    keys = round_keys(seed, counter)
    mask_r = np.uint32((1 << rb) - 1)

    def P(v):
        L, R = v >> np.uint32(rb), v & mask_r
        for r in range(0, ROUNDS, 2):
            L = L ^ (_mix(R, keys[r]) >> np.uint32(32 - lb))
            R = R ^ (_mix(L, keys[r + 1]) >> np.uint32(32 - rb))
        return (L << np.uint32(rb)) | R

    with np.errstate(over="ignore"):
        out = P(np.arange(n, dtype=np.uint32))
        while True:
            todo = np.nonzero(out >= n)[0]
            if todo.size == 0:
                break
            out[todo] = P(out[todo])
    return out.astype(np.int32)
