"""fold_cloth3 -- mirrors /root/reference/DaXBench/daxbench/core/envs/fold_cloth3_env.py:18-58: the fold_cloth1 cloth
(same 16 x 32 mask, same kernels) with stiffness fixed at 900 in the conf and four pick-and-place steps per episode."""
from .basic.cloth_conf import ClothConfBase, patch_mask
from .basic.cloth_env import ClothEnv


class DefaultConf(ClothConfBase):            # fold_cloth3_env.py:18-38
    stiffness = 900
    task = "fold_cloth3"


FoldCloth3Config = DefaultConf


class FoldCloth3Env(ClothEnv):

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, device="cuda"):
        conf = DefaultConf() if conf is None else conf
        max_steps = 4                                                      # :47
        super().__init__(conf, batch_size, max_steps, aux_reward, device=device)
        self.observation_size = 1544

    def create_cloth_mask(self, conf):
        return patch_mask(conf)