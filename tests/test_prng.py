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
