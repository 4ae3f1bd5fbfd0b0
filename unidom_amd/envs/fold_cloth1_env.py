"""fold_cloth1 -- mirrors /root/reference/DaXBench/daxbench/core/envs/fold_cloth1_env.py:15-53."""
from .basic.cloth_conf import ClothConfBase, patch_mask
from .basic.cloth_env import ClothEnv


class DefaultConf(ClothConfBase):            # fold_cloth1_env.py:15-33
    stiffness = 9             # the constructor argument (default 900) replaces it
    task = "fold_cloth1"


FoldCloth1Conf = DefaultConf


class FoldCloth1Env(ClothEnv):

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, stiffness=900, device="cuda"):
        conf = DefaultConf() if conf is None else conf
        max_steps = 3
        conf.stiffness = stiffness
        super().__init__(conf, batch_size, max_steps, aux_reward, device=device)
        self.observation_size = 1544

    def create_cloth_mask(self, conf):
        return patch_mask(conf)