"""CPU: the data files the envs load are the reference's: every task whose reference conf points at a
goal file that the reference ships loads that goal (no zeros((1,3)) fallback, cloth_env.py:60-65 / mpm_env.py:46-51), with
the particle count of the task; the one conf whose goal_path the reference does not ship (pour_soup -> goals/pour_soup2)
falls back exactly like the reference."""
import importlib
import os

import numpy as np
import pytest

CASES = [   # registry key, module, conf class, particles in the goal cloud (None: the reference ships no file at conf.goal_path)
    ("fold_cloth1", "fold_cloth1_env", "DefaultConf", 512),
    ("fold_cloth1_para", "fold_cloth1_para_env", "DefaultConf", 512),
    ("fold_cloth3", "fold_cloth3_env", "DefaultConf", 512),
    ("unfold_cloth1", "unfold_cloth1_env", "DefaultConf", 512),
    ("unfold_cloth3", "unfold_cloth1_env", "DefaultConf3", 512),
    ("fold_tshirt", "fold_cloth_tshirt_env", "DefaultConf", 3573),          # fold_cloth_tshirt_env.py:36-37
    ("whip_rope", "whip_rope_env", "DefaultConf", 67),
    ("shape_rope", "shape_rope_env", "DefaultConf", 582),
    ("shape_rope_hard", "shape_rope_hard_env", "DefaultConf", 582),         # shape_rope_hard_env.py:4 re-uses shape_rope's conf
    ("pour_water", "pour_water_env", "DefaultConf", 702),
    ("pour_soup", "pour_soup_env", "DefaultConf", None),                    # pour_soup_env.py:59-60: task = "pour_soup2"
]


@pytest.mark.parametrize("name,module,conf_name,n", CASES)
def test_goal_files(name, module, conf_name, n):
    mod = importlib.import_module(f"unidom_amd.envs.{module}")
    conf_cls = getattr(mod, conf_name, None) or getattr(mod, "DefaultConf")
    conf = conf_cls()
    if n is None:
        assert not os.path.exists(conf.goal_path)
        return
    assert os.path.exists(conf.goal_path), f"{name}: {conf.goal_path} missing -> the env would score against zeros((1,3))"
    goal = np.load(conf.goal_path)
    assert goal.shape == (n, 3) and goal.dtype == np.float32 and np.isfinite(goal).all()


def test_every_registry_key_is_covered():
    from unidom_amd.envs.registration import env_functions
    covered = {c[0] for c in CASES} | {"push_rope", "push_rope_hard"}      # aliases of shape_rope(_hard), registration.py:18-21
    assert set(env_functions) == covered
