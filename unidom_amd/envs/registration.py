"""Name -> env class, same keys as the reference's registry for the envs on the hot path
(/root/reference/DaXBench/daxbench/core/envs/registration.py:13-27).  fold_cloth3 / unfold_cloth1 / unfold_cloth3 are
the same 512-particle cloth and the same kernels under other confs; shape_rope / shape_rope_hard run the MPM kernels
in soft-contact mode with the plastic material.  pour_water adds the liquid material, two primitives and the container SDF.
fold_tshirt (3573 particles) runs the several-particles-per-lane cloth kernels.  pour_soup is pour_water at n_grid 128 with a mixed
cloud of 7631 particles (liquid, two tofu blocks, a vegetable point cloud).  That is every key of the reference's registry."""
from .fold_cloth1_env import FoldCloth1Env
from .fold_cloth1_para_env import FoldCloth1ParaEnv
from .fold_cloth3_env import FoldCloth3Env
from .fold_cloth_tshirt_env import FoldTshirtEnv
from .unfold_cloth1_env import UnfoldCloth1Env, UnfoldCloth3Env

env_functions = {
    "fold_cloth1": FoldCloth1Env,
    "fold_cloth1_para": FoldCloth1ParaEnv,
    "fold_cloth3": FoldCloth3Env,
    "unfold_cloth1": UnfoldCloth1Env,
    "unfold_cloth3": UnfoldCloth3Env,
    "fold_tshirt": FoldTshirtEnv,
}

from .whip_rope_env import WhipRopeEnv
from .shape_rope_env import ShapeRopeEnv
from .shape_rope_hard_env import ShapeRopeHardEnv
from .pour_water_env import PourWaterEnv
from .pour_soup_env import PourSoupEnv

env_functions["whip_rope"] = WhipRopeEnv
env_functions["shape_rope"] = env_functions["push_rope"] = ShapeRopeEnv                  # both names, registration.py:18-21
env_functions["shape_rope_hard"] = env_functions["push_rope_hard"] = ShapeRopeHardEnv
env_functions["pour_water"] = PourWaterEnv
env_functions["pour_soup"] = PourSoupEnv
