"""Name -> env class, same keys as the reference's registry for the envs on the hot path
(/root/reference/DaXBench/daxbench/core/envs/registration.py:13-27).  The other reference envs
(fold_cloth3, fold_tshirt, shape_rope, pour_water, ...) are "next" rows and raise."""
from .fold_cloth1_env import FoldCloth1Env
from .fold_cloth1_para_env import FoldCloth1ParaEnv

env_functions = {
    "fold_cloth1": FoldCloth1Env,
    "fold_cloth1_para": FoldCloth1ParaEnv,
}

try:  # MPM envs (whip_rope) register themselves once the MPM kernels are built
    from .whip_rope_env import WhipRopeEnv
    env_functions["whip_rope"] = WhipRopeEnv
except ImportError:  # pragma: no cover
    pass
