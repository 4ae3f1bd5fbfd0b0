"""fold_tshirt -- mirrors /root/reference/DaXBench/daxbench/core/envs/fold_cloth_tshirt_env.py:19-114 (DefaultConf :19-40,
create_cloth_mask :50-67, get_obs :69-111): a T-shirt-shaped cloth of 3573 particles on a 180 x 180 lattice, stiffness 5000,
dt 0.5e-3, five pick-and-place steps per episode, observation = every tenth particle + the two grippers.

The reference reads the mask from others/t-shirt.jpg with cv2 (resize, rotate, threshold); here it is the data file
others/tshirt_mask.npy, recovered from the recorded reset state of the reference's expert_demo/fold_tshirt/demo_0.pkl
(tests/golden/make_golden.py).  The goal cloud is the reference's goals/fold_tshirt/goal.npy ((3573, 3) f32, conf.goal_path
:36-37), re-packed by the same script into envs/goals/fold_tshirt/.  Bodies above 1024 particles run the
several-particles-per-lane kernels of csrc/cloth.hip."""
import numpy as np
import torch

from .basic.cloth_conf import ENVS_DIR, ClothConfBase
from .basic.cloth_env import ClothEnv


class DefaultConf(ClothConfBase):            # fold_cloth_tshirt_env.py:19-40
    N = 180
    stiffness = 5000
    dt = 0.5e-3
    mu = 0.9
    task = "fold_tshirt"


FoldTshirtConfig = DefaultConf


class FoldTshirtEnv(ClothEnv):

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, device="cuda"):
        conf = DefaultConf() if conf is None else conf
        max_steps = 5                                                      # :46
        super().__init__(conf, batch_size, max_steps, aux_reward, device=device)
        self.observation_size = 1082                                       # 358 sampled particles x 3 + 2 x 4

    def create_cloth_mask(self, conf):   # :50-67 (see the module docstring)
        mask = np.load(f"{ENVS_DIR}/others/tshirt_mask.npy")
        assert mask.shape == (conf.N, conf.N)
        return mask.astype(np.float32)

    def get_obs(self, state, eval_min_max_stiff=None, obs_type=ClothEnv.PARTICLE):   # :69-111
        if obs_type != ClothEnv.PARTICLE:
            raise NotImplementedError("only PARTICLE observations are on the hot path")
        x = state.x[..., ::10, :]                                          # sample x (N,3) every 10 points
        lead = x.shape[:-2]
        return torch.cat([x.reshape(lead + (-1,)), state.primitive0, state.primitive1], -1)
