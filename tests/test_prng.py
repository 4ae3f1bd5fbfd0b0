"""CPU: threefry2x32 `split` pinned by the keys recorded in the reference's fold_cloth1 demos
(one split per robot_step, 40 robot_steps per step_diff: cloth_simulator.py:172, cloth_env.py:134-173)."""
import os

import numpy as np

from conftest import GOLDEN
from unidom_amd.utils import prng


def test_split_known_value():
    assert prng.split(prng.PRNGKey(0))[0].tolist() == [4146024105, 967050713]


def test_split_chain_matches_recorded_keys():
    d = np.load(os.path.join(GOLDEN, "fold_cloth1_demos.npz"))
    assert len(d["demo"]) == 17
    for i in range(len(d["demo"])):
        assert np.array_equal(prng.split_first(d["s0_key"][i], 40), d["s1_key"][i])


def test_normal_is_standard():
    z = prng.normal(prng.PRNGKey(3), 20000)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03 and z.dtype == np.float32


# ---- known answers printed in JAX's own public documentation (jax.readthedocs.io), reproduced here digit for digit.
# They pin threefry_random_bits (even and odd sizes), uniform's bit trick, and XLA's f32 erf_inv as restated in
# utils/prng.py.  numpy prints the shortest decimal that identifies a float32, as jax's repr does, so equality of the
# printed strings is equality of the bits.
def _show(a):
    return np.array2string(np.asarray(a, np.float32), separator=" ").replace("\n", "")


def test_normal_matches_jax_quickstart_vector():
    # "JAX Quickstart": key = random.PRNGKey(0); x = random.normal(key, (10,)); print(x)
    doc = [-0.3721109, 0.26423115, -0.18252768, -0.7368197, -0.44030377, -0.1521442, -0.67135346, -0.5908641,
           0.73168886, 0.5673026]
    z = prng.normal(prng.PRNGKey(0), 10)
    assert np.array_equal(z, np.float32(doc)), _show(z)


def test_normal_matches_jax_sharp_bits_scalars():
    # "JAX - The Sharp Bits", random numbers: random.normal(PRNGKey(0), shape=(1,)) -> [-0.20584226];
    # key, subkey = random.split(key); random.normal(subkey, shape=(1,)) -> [-1.2515389]   (odd size: the counter pad is a 0)
    key = prng.PRNGKey(0)
    assert prng.normal(key, 1)[0] == np.float32(-0.20584226)
    _, subkey = prng.split(key)
    assert prng.normal(subkey, 1)[0] == np.float32(-1.2515389)


def test_uniform_matches_jax_documentation_scalar():
    # random.uniform(random.PRNGKey(0), (1,)) -> [0.41845703]
    assert prng.uniform(prng.PRNGKey(0), 1)[0] == np.float32(0.41845703)


def test_erf_inv_f32_tracks_the_f64_function_and_is_odd():
    from scipy.special import erfinv
    u = np.linspace(-0.9999999, 0.9999999, 100001).astype(np.float32)
    a = prng.erf_inv_f32(u)
    b = erfinv(u.astype(np.float64))
    assert np.abs(a - b).max() < 1e-4 and (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max() < 2e-5   # far tail: |erfinv| ~ 3.8, f32 ulp 2.4e-7
    assert np.array_equal(prng.erf_inv_f32(-u), -a)
    assert np.array_equal(prng.erf_inv_f32(np.float32([1, -1, 0])), np.float32([np.inf, -np.inf, 0]))


def test_fold_in_and_flax_param_key_are_deterministic_functions_of_the_path():
    k = prng.PRNGKey(7)
    a = prng.flax_param_key(k, ("hidden_0",))
    assert a.dtype == np.uint32 and a.shape == (2,)
    assert np.array_equal(a, prng.flax_param_key(k, ("hidden_0",), 0))
    assert not np.array_equal(a, prng.flax_param_key(k, ("hidden_1",)))
    assert not np.array_equal(a, prng.flax_param_key(k, ("hidden_0",), 1))
    # fold_in(key, d) is threefry over the counter pair (0, d)
    y0, y1 = prng.threefry2x32(k, np.uint32(0), np.uint32(12345))
    assert np.array_equal(prng.fold_in(k, 12345), np.array([y0, y1], np.uint32))
