"""unfold_cloth1 / unfold_cloth3 -- mirror /root/reference/DaXBench/daxbench/core/envs/unfold_cloth1_env.py:15-79 and
unfold_cloth3_env.py (identical but for the number of random folds applied at reset, 1 vs 3): the fold_cloth1 cloth
with friction mu = 3, 15 steps per episode, no substep observations, and a reset that jitters the particles
(normal * 1e-4) and then folds the cloth with random pick-and-place actions through step_diff itself."""
import numpy as np
import torch

from ..utils import prng
from .basic.cloth_conf import ClothConfBase, patch_mask
from .basic.cloth_env import ClothEnv


class DefaultConf(ClothConfBase):            # unfold_cloth1_env.py:15-35
    stiffness = 900
    mu = 3                    # a sticky table
    use_substep_obs = False
    task = "unfold_cloth1"


UnfoldCloth1Config = DefaultConf


class UnfoldCloth1Env(ClothEnv):
    random_fold_steps = 1                                                  # unfold_cloth1_env.py:75

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, device="cuda"):
        conf = DefaultConf() if conf is None else conf
        max_steps = 15                                                     # :44
        super().__init__(conf, batch_size, max_steps, aux_reward, device=device)
        self.observation_size = 1544
        self.reset = self.build_reset()                                    # :47

    def create_cloth_mask(self, conf):
        return patch_mask(conf)

    def random_fold(self, state, key, step=10):   # :56-66 (draws from numpy's global stream, as the reference does)
        num_particle = state.x.shape[1]
        B = state.x.shape[0]
        bi = torch.arange(B, device=state.x.device)
        for _ in range(step):
            st_point = np.random.randint(0, num_particle, size=(B,))
            ed_point = np.random.randint(0, num_particle, size=(B,))
            actions = torch.cat((state.x[bi, torch.as_tensor(st_point, device=state.x.device)],
                                 state.x[bi, torch.as_tensor(ed_point, device=state.x.device)]), -1)
            with torch.no_grad():
                _, _, _, info = self.step_diff(actions, state)
            state = info["state"]
        return state

    def build_reset(self):               # :68-79
        init_state = self.simulator.reset_jax()

        def reset(key):
            key = prng.split(np.asarray(key, dtype=np.uint32))[0]
            n = int(np.prod(init_state.x.shape))
            noise = prng.normal(key, n).reshape(tuple(init_state.x.shape)) * np.float32(0.0001)
            new_x = init_state.x + torch.as_tensor(noise, device=init_state.x.device)
            state = init_state._replace(x=new_x)
            state = self.random_fold(state, key, step=self.random_fold_steps)
            return self.get_obs(state), state

        return reset


class DefaultConf3(DefaultConf):   # unfold_cloth3_env.py:15-35
    task = "unfold_cloth3"


UnfoldCloth3Config = DefaultConf3


class UnfoldCloth3Env(UnfoldCloth1Env):
    random_fold_steps = 3                                                  # unfold_cloth3_env.py:78

    def __init__(self, batch_size, conf=None, aux_reward=False, seed=1, device="cuda"):
        super().__init__(batch_size, DefaultConf3() if conf is None else conf, aux_reward, seed, device=device)
