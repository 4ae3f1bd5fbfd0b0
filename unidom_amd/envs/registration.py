"""Name -> env class, same keys as the reference's registry for the envs on the hot path
(/root/reference/DaXBench/daxbench/core/envs/registration.py:13-27).  fold_cloth3 / unfold_cloth1 / unfold_cloth3 are
the same 512-particle cloth and the same kernels under other confs; shape_rope / shape_rope_hard run the MPM kernels
in soft-contact mode with the plastic material.  pour_water adds the liquid material, two primitives and the container SDF.
fold_tshirt (3573 particles) runs the several-particles-per-lane cloth kernels.  The one remaining reference env, pour_soup,
needs an open3d point-cloud asset the reference loads at reset and is absent from the registry."""
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

try:  # MPM envs (whip_rope) register themselves once the MPM kernels are built
    from .whip_rope_env import WhipRopeEnv
    from .shape_rope_env import ShapeRopeEnv
    from .shape_rope_hard_env import ShapeRopeHardEnv
    from .pour_water_env import PourWaterEnv
    env_functions["whip_rope"] = WhipRopeEnv
    env_functions["shape_rope"] = ShapeRopeEnv
    env_functions["shape_rope_hard"] = ShapeRopeHardEnv
    env_functions["pour_water"] = PourWaterEnv
except ImportError:  # pragma: no cover
    pass
