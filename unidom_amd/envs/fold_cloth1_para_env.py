"""fold_cloth1_para -- mirrors /root/reference/DaXBench/daxbench/core/envs/fold_cloth1_para_env.py:15-53
(parameter-aware observation: normalised stiffness appended, cloth_env_para.py:130)."""
from .basic.cloth_conf import patch_mask
from .basic.cloth_env import ClothEnv
from .fold_cloth1_env import DefaultConf as _Base


class DefaultConf(_Base):
    pass


FoldCloth1Conf = DefaultConf


class FoldCloth1ParaEnv(ClothEnv):

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, stiffness=900, eval_min_max_stiff=[100, 2000],
                 device="cuda"):
        conf = DefaultConf() if conf is None else conf
        max_steps = 3
        conf.stiffness = stiffness
        super().__init__(conf, batch_size, max_steps, aux_reward, list(eval_min_max_stiff), device=device)
        self.observation_size = 1545

    def create_cloth_mask(self, conf):
        return patch_mask(conf)
