import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def fold_cloth1_mask(N=80, size=16):
    """fold_cloth1_env.py:48-53"""
    m = np.zeros((N, N), dtype=np.float32)
    m[size * 2:size * 3, size * 2:size * 4] = 1
    return m


def cloth_reset_x(N=80, mask=None):
    """cloth_simulator.py:339-353"""
    mask = fold_cloth1_mask(N) if mask is None else mask
    c = 1.0 / N
    ii, jj = np.nonzero(mask)
    return np.stack([ii * c, np.zeros_like(ii, dtype=np.float64), (N - jj) * c], -1).astype(np.float32)


def make_cloth_case(rng, B, T, P_x=None, deform=0.002, v_scale=0.05, grasp=True):
    """Seeded synthetic cloth states/actions: deformed lattice, gripper 0 placed on a particle (so the
    discrete grasp event fires), macro actions that lift/move it."""
    x0 = cloth_reset_x() if P_x is None else P_x
    P = x0.shape[0]
    x = np.repeat(x0[None], B, 0).astype(np.float32)
    x[:, :, [0, 2]] += (rng.normal(size=(B, 1, 2)) * 0.05).astype(np.float32)
    x += (rng.normal(size=x.shape) * deform).astype(np.float32)
    x = np.abs(x).astype(np.float32)
    v = (rng.normal(size=x.shape) * v_scale).astype(np.float32)
    prim = np.zeros((B, 2, 4), np.float32)
    prim[:, 1] = [1, 1, 1, 0.01]
    for b in range(B):
        p = rng.integers(0, P)
        prim[b, 0, :3] = x[b, p] + (np.array([0, 0.002, 0]) if grasp else np.array([0, 0.2, 0]))
        prim[b, 0, 3] = 0.01
    actions = (rng.normal(size=(T, B, 8)) * 0.3).astype(np.float32)
    actions[..., 3] = rng.uniform(0, 1, size=(T, B)) < 0.3
    actions[..., 4:] = 0
    k = rng.uniform(600, 1500, size=B).astype(np.float32)
    mu = rng.uniform(0.3, 0.9, size=B).astype(np.float32)
    return x, v, prim, k, mu, actions
