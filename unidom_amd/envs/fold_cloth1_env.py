"""fold_cloth1 -- mirrors /root/reference/DaXBench/daxbench/core/envs/fold_cloth1_env.py:15-53."""
import os
from dataclasses import dataclass

import numpy as np

from .basic.cloth_env import ClothEnv

my_path = os.path.dirname(os.path.abspath(__file__))


@dataclass
class DefaultConf:            # fold_cloth1_env.py:15-33
    N = 80
    cell_size = 1.0 / N
    gravity = 0.5
    stiffness = 9
    damping = 2
    dt = 2e-3
    max_v = 2.
    small_num = 1e-8
    mu = 0.5  # friction
    seed = 1
    size = int(N / 5.0)
    mem_saving_level = 2      # kept for interface parity; the HIP path checkpoints per substep instead
    task = "fold_cloth1"
    goal_path = f"{my_path}/goals/{task}/goal.npy"
    use_substep_obs = True


FoldCloth1Conf = DefaultConf


class FoldCloth1Env(ClothEnv):

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, stiffness=900, device="cuda"):
        conf = DefaultConf() if conf is None else conf
        max_steps = 3
        conf.stiffness = stiffness
        super().__init__(conf, batch_size, max_steps, aux_reward, device=device)
        self.observation_size = 1544

    def create_cloth_mask(self, conf):   # :48-53
        N, size = conf.N, conf.size
        cloth_mask = np.zeros((N, N), dtype=np.float32)
        cloth_mask[size * 2:size * 3, size * 2:size * 4] = 1
        return cloth_mask
