"""fold_cloth3 -- mirrors /root/reference/DaXBench/daxbench/core/envs/fold_cloth3_env.py:18-58: the fold_cloth1 cloth
(same 16 x 32 mask, same kernels) with stiffness fixed at 900 in the conf and four pick-and-place steps per episode."""
import os
from dataclasses import dataclass

import numpy as np

from .basic.cloth_env import ClothEnv

my_path = os.path.dirname(os.path.abspath(__file__))


@dataclass
class DefaultConf:            # fold_cloth3_env.py:18-38
    N = 80
    cell_size = 1.0 / N
    gravity = 0.5
    stiffness = 900
    damping = 2
    dt = 2e-3
    max_v = 2.
    small_num = 1e-8
    mu = 0.5  # friction
    seed = 1
    size = int(N / 5.0)
    mem_saving_level = 2      # interface parity only: the HIP path checkpoints per substep
    task = "fold_cloth3"
    goal_path = f"{my_path}/goals/{task}/goal.npy"
    use_substep_obs = True


FoldCloth3Config = DefaultConf


class FoldCloth3Env(ClothEnv):

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, device="cuda"):
        conf = DefaultConf() if conf is None else conf
        max_steps = 4                                                      # :47
        super().__init__(conf, batch_size, max_steps, aux_reward, device=device)
        self.observation_size = 1544

    def create_cloth_mask(self, conf):   # :51-56
        N, size = conf.N, conf.size
        cloth_mask = np.zeros((N, N), dtype=np.float32)
        cloth_mask[size * 2:size * 3, size * 2:size * 4] = 1
        return cloth_mask
