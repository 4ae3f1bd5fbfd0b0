"""JAX-compatible counter PRNG pieces the envs need (threefry2x32 `split`).

The reference advances `state.key` with `key, _ = jax.random.split(key)` once per robot_step
(cloth_simulator.py:172) and in reset / auto_reset (cloth_env.py:182, whip_rope_env.py:98).  jax is a
third-party dependency that is not in /root/reference; the algorithm restated here is the published
Threefry-2x32 (20 rounds, Salmon et al. 2011) with JAX's split layout.  It is pinned by the keys recorded
in the reference's fold_cloth1 demos (tests/test_prng.py): 40 successive `split(key)[0]` map each recorded
state.key to the next one exactly.  `normal` / `uniform` are NOT reproduced bit-exactly.
"""
from __future__ import annotations

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_PARITY = np.uint32(0x1BD11BDA)


def _rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def threefry2x32(key, x0, x1):
    """key: [...,2] uint32; x0, x1: uint32 arrays broadcastable against key[...,0]."""
    with np.errstate(over="ignore"):
        k0 = key[..., 0].astype(np.uint32)
        k1 = key[..., 1].astype(np.uint32)
        ks = (k0, k1, k0 ^ k1 ^ _PARITY)
        x0 = (x0.astype(np.uint32) + ks[0]).astype(np.uint32)
        x1 = (x1.astype(np.uint32) + ks[1]).astype(np.uint32)
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = (x0 + x1).astype(np.uint32)
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = (x0 + ks[(i + 1) % 3]).astype(np.uint32)
            x1 = (x1 + ks[(i + 2) % 3] + np.uint32(i + 1)).astype(np.uint32)
    return x0, x1


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def split(key: np.ndarray, num: int = 2) -> np.ndarray:
    """jax.random.split: hash counters 0..2*num-1, first half as x0 lanes, second half as x1 lanes.
    key [...,2] -> [..., num, 2]."""
    key = np.asarray(key, dtype=np.uint32)
    cnt = np.arange(2 * num, dtype=np.uint32)
    lead = key.shape[:-1]
    x0 = np.broadcast_to(cnt[:num], lead + (num,))
    x1 = np.broadcast_to(cnt[num:], lead + (num,))
    y0, y1 = threefry2x32(key[..., None, :], x0, x1)
    return np.concatenate([y0, y1], axis=-1).reshape(lead + (num, 2))


def split_first(key: np.ndarray, times: int = 1) -> np.ndarray:
    """`key, _ = split(key)` applied `times` times."""
    key = np.asarray(key, dtype=np.uint32)
    for _ in range(times):
        key = split(key, 2)[..., 0, :]
    return key
