"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mava_amd/).

NumPy restatement of the Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw,
"Parallel random numbers: as easy as 1, 2, 3", SC'11) exactly as coded in
mava_amd/csrc/common.h, so that sampled actions and synthetic observations can be compared
bit for bit.  The reference (Mava) samples with JAX threefry2x32 (jax.random.categorical,
mava/systems/ppo/ff_mappo.py:81-84); reproducing JAX's bit stream is out of scope
 - parity tests treat the noise as an input.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over broadcastable uint32 counter arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        np.asarray(c0, np.uint32), np.asarray(c1, np.uint32), np.asarray(c2, np.uint32), np.asarray(c3, np.uint32)
    )
    c0 = c0.astype(np.uint32).copy()
    c1 = c1.astype(np.uint32).copy()
    c2 = c2.astype(np.uint32).copy()
    c3 = c3.astype(np.uint32).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n2 = hi0 ^ c3 ^ k1
            c0, c1, c2, c3 = n0, lo1, n2, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u01_open(x):
    """(top 24 bits + 0.5) * 2^-24, exact in float32 - same as csrc/common.h u01_open."""
    return ((np.asarray(x, np.uint32) >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(
        1.0 / 16777216.0
    )


POLICY_STREAM = 0x504F4C49  # "POLI"
ENV_STREAM = 0x454E5653  # "ENVS"


def policy_uniforms(seed: int, step: int, rows: int, n_actions: int, row_offset: int = 0):
    """Uniforms used by the Gumbel-max sampler of mava_policy_step_f32: shape (rows, n_actions)."""
    slo, shi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    gid = (np.arange(rows, dtype=np.uint64) + np.uint64(row_offset)).astype(np.uint32)
    out = np.empty((rows, n_actions), np.float32)
    for c in range((n_actions + 3) // 4):
        w = philox4x32_10(gid, step, c, POLICY_STREAM, slo, shi)
        for q in range(4):
            o = 4 * c + q
            if o < n_actions:
                out[:, o] = u01_open(w[q])
    return out
