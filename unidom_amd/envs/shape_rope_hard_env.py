"""shape_rope_hard -- mirrors /root/reference/DaXBench/daxbench/core/envs/shape_rope_hard_env.py:8-20:
shape_rope with max_steps = 20 and eight more random pushes at reset."""
import torch

from .shape_rope_env import DefaultConf, ShapeRopeEnv

ShapeRopeHardConfig = DefaultConf


class ShapeRopeHardEnv(ShapeRopeEnv):

    def __init__(self, batch_size, seed, max_steps=20, conf=None, aux_reward=False, device="cuda", **kwargs):
        super().__init__(batch_size, seed, max_steps=max_steps, conf=conf, aux_reward=aux_reward, device=device)

    def reset(self, key):
        super().reset(key)
        with torch.no_grad():
            self.random_push(step=8)
        return self.get_obs(self.state), self.state
