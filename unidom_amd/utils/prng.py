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


def random_bits(key: np.ndarray, n: int) -> np.ndarray:
    """threefry_random_bits for 32-bit words: hash iota(n) (padded to even), halves as x0/x1 lanes.
    key [..., 2] -> [..., n] (leading dims = vmap over keys, evaluated in one vectorised pass)."""
    key = np.asarray(key, dtype=np.uint32)
    m = n + (n % 2)
    cnt = np.arange(m, dtype=np.uint32)
    y0, y1 = threefry2x32(key[..., None, :], cnt[: m // 2], cnt[m // 2:])
    return np.concatenate([y0, y1], axis=-1)[..., :n]


def uniform(key: np.ndarray, n: int, minval=0.0, maxval=1.0) -> np.ndarray:
    """jax.random.uniform (f32): mantissa bits -> [1,2) - 1, scaled; recalled semantics, NOT pinned by data."""
    bits = random_bits(key, n)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


def normal(key: np.ndarray, n: int) -> np.ndarray:
    """jax.random.normal (f32) = sqrt(2) * erfinv(uniform(-1+ulp, 1)).  erfinv is evaluated in f64 here
    (XLA uses an f32 polynomial), so values agree with JAX to f32 round-off, not bit for bit."""
    from scipy.special import erfinv
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, n, lo, 1.0)
    return (np.float32(np.sqrt(2)) * erfinv(u.astype(np.float64)).astype(np.float32)).astype(np.float32)


def normal_batch(keys: np.ndarray, n: int) -> np.ndarray:
    """vmap(lambda k: normal(k, (n,)))(keys): keys [B,2] -> [B,n]."""
    return normal(np.asarray(keys, dtype=np.uint32), n)
