"""JAX-compatible counter PRNG (threefry2x32): `split`, `fold_in`, `uniform`, `normal`, and flax's parameter-key derivation.

The reference threads one threefry key through everything random: `state.key` advances by `key, _ = jax.random.split(key)`
once per robot_step (cloth_simulator.py:172) and in reset / auto_reset (cloth_env.py:182, whip_rope_env.py:98); APG splits
its seed into model / env / noise keys (apg.py:75-80), initialises the policy from `key_models` (apg.py:107, flax Dense
params via lecun_uniform) and draws the action noise with `jax.random.normal(key_sample, loc.shape)` per step (apg.py:179-184).
jax / flax are third-party and not in /root/reference; what is restated here is the published Threefry-2x32 (20 rounds,
Salmon et al. 2011) with JAX 0.3.14's layouts, recalled from its source:

  split      pinned by data: the keys recorded in the reference's cloth demos (tests/test_prng.py) -- 40 successive
             `split(key)[0]` map each recorded state.key to the next one exactly.
  fold_in    threefry_2x32(key, [0, data]); unpinned (recalled).
  uniform    mantissa bits -> [1,2) - 1, scaled, max(minval, .); unpinned (recalled).
  normal     sqrt(2) * erf_inv(uniform(nextafter(-1, 0), 1)) with XLA's f32 erf_inv (Giles' two-branch polynomial in
             w = -log1p(-x*x), Horner form, one rounding per operation) -- unpinned (recalled), but **bit-reproducible
             across hosts**: every step is an IEEE f32 operation except log1p, which is taken in f64 and rounded once.
  flax key   Scope.make_rng / LazyRng (flax >= 0.4.1): a parameter's key is fold_in(rng, first 4 bytes of
             sha1(path names + counter bytes)); unpinned (recalled; the reference does not pin flax's version).
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
    if n % 2:
        cnt[-1] = 0          # threefry_2x32 pads an odd count array with a literal 0, not with the next counter
    y0, y1 = threefry2x32(key[..., None, :], cnt[: m // 2], cnt[m // 2:])
    return np.concatenate([y0, y1], axis=-1)[..., :n]


def fold_in(key: np.ndarray, data: int) -> np.ndarray:
    """jax.random.fold_in: threefry_2x32(key, threefry_seed(data)) with threefry_seed(uint32 d) = [0, d]."""
    key = np.asarray(key, dtype=np.uint32)
    y0, y1 = threefry2x32(key, np.zeros((), np.uint32), np.asarray(int(data) & 0xFFFFFFFF, dtype=np.uint32))
    return np.stack([y0, y1], axis=-1).astype(np.uint32)


def flax_param_key(rng: np.ndarray, path, counter: int = 0) -> np.ndarray:
    """The key flax hands a parameter initialiser: Scope.make_rng('params') of the module at `path` (tuple of submodule
    names) for its `counter`-th draw = LazyRng(rng, path + (counter,)).as_jax_rng() = fold_in(rng, sha1(...)[:4])
    (flax/core/scope.py `_fold_in_static`: strings as utf-8, ints as minimal big-endian bytes -- 0 is the empty string)."""
    import hashlib
    m = hashlib.sha1()
    for x in tuple(path) + (int(counter),):
        if isinstance(x, str):
            m.update(x.encode("utf-8"))
        else:
            m.update(int(x).to_bytes((int(x).bit_length() + 7) // 8, byteorder="big"))
    return fold_in(rng, int.from_bytes(m.digest()[:4], byteorder="big"))


def uniform(key: np.ndarray, n: int, minval=0.0, maxval=1.0) -> np.ndarray:
    """jax.random.uniform (f32): mantissa bits -> [1,2) - 1, scaled; recalled semantics, NOT pinned by data."""
    bits = random_bits(key, n)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


# XLA's ErfInv32 (xla/client/lib/math.cc; M. Giles, "Approximating the erfinv function"): coefficients of the two branches,
# highest degree first, evaluated as p = c[i] + p * w.
_ERFINV_LT5 = np.float32([2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
                          -0.00125372503, -0.00417768164, 0.246640727, 1.50140941])
_ERFINV_GE5 = np.float32([-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
                          -0.0076224613, 0.00943887047, 1.00167406, 2.83297682])


def erf_inv_f32(x: np.ndarray) -> np.ndarray:
    """lax.erf_inv for float32 as XLA expands it.  All arithmetic is f32 with one rounding per operation (numpy does
    not contract); log1p is evaluated in f64 and rounded once, i.e. correctly rounded for all practical purposes, so
    the result does not depend on the host's libm."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (-np.log1p((-(x * x)).astype(np.float64))).astype(np.float32)
        lt = w < np.float32(5.0)
        w = np.where(lt, w - np.float32(2.5), np.sqrt(w) - np.float32(3.0)).astype(np.float32)
        p = np.where(lt, _ERFINV_LT5[0], _ERFINV_GE5[0]).astype(np.float32)
        for i in range(1, 9):
            p = (np.where(lt, _ERFINV_LT5[i], _ERFINV_GE5[i]).astype(np.float32) + p * w).astype(np.float32)
        r = (p * x).astype(np.float32)
    return np.where(np.abs(x) == np.float32(1.0), np.copysign(np.float32(np.inf), x), r).astype(np.float32)


def normal(key: np.ndarray, n: int) -> np.ndarray:
    """jax.random.normal (f32) = sqrt(2) * erf_inv(uniform(nextafter(-1, 0), 1)); row-major over `n` values
    (a shape (B, A) draw is normal(key, B * A).reshape(B, A))."""
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, n, lo, 1.0)
    return (np.float32(np.sqrt(2)) * erf_inv_f32(u)).astype(np.float32)


def normal_batch(keys: np.ndarray, n: int) -> np.ndarray:
    """vmap(lambda k: normal(k, (n,)))(keys): keys [B,2] -> [B,n]."""
    return normal(np.asarray(keys, dtype=np.uint32), n)
