"""whip_rope -- mirrors /root/reference/DaXBench/daxbench/core/envs/whip_rope_env.py:27-137
(DefaultConf :27-73, auto_reset :94-106, get_primitive_actions :108-115, reset :117-137)."""
import os
from dataclasses import dataclass

import numpy as np
import torch

from ..engine.primitives.box import _sdf_batch as box_sdf
from ..engine.primitives.primitives import set_sdf
from ..utils import prng
from .basic.mpm_env import MPMEnv

my_path = os.path.dirname(os.path.abspath(__file__))
_R50 = float(np.float32(1.0) / np.float32(50.0))   # f32 reciprocal of the literal in get_primitive_actions (:112)


@dataclass
class DefaultConf:
    seed = 1
    n_primitive = 1
    focus_computation = True
    use_position_control = True
    obs_type = MPMEnv.PARTICLE
    key = prng.PRNGKey(0)

    ground_friction: float = 0.1
    n_grid: int = 64
    dt: float = 1e-4
    primitive_action_steps = 1
    primitive_action_duration = 0.007  # seconds
    steps = int(primitive_action_duration / primitive_action_steps / dt)  # internal steps (= 70)
    E: float = 100
    nu: float = 0.1
    res: tuple = (n_grid // 2, n_grid // 2, n_grid // 2)

    dx, inv_dx = 1 / n_grid, float(n_grid)
    p_vol, p_rho = (dx * 0.5) ** 2, 1
    p_mass = p_vol * p_rho
    gravity = (0, -9.8, 0)

    task = "whip_rope"
    goal_path = f"{my_path}/goals/{task}/goal.npy"

    # Rope property
    rope_width = [0.38, 0.006, 0.006]
    rope_init_pos = [0.5, 0.01, 0.5]
    rope_z_rotation_angle = np.pi / 2
    rope_hardness = 1.0


WhipRopeConfig = DefaultConf


class WhipRopeEnv(MPMEnv):

    def __init__(self, batch_size, seed, max_steps=70, conf=None, aux_reward=False, device="cuda", **kwargs):
        conf = DefaultConf() if conf is None else conf
        self.conf = conf
        self.focus_computation = True
        super().__init__(conf, batch_size, max_steps, seed, conf.focus_computation, conf.use_position_control, device=device)
        self.observation_size = 612

    @staticmethod
    def process_pre_step_actions(actions, shift):
        return actions

    def auto_reset(self, state, state_new, key):   # :94-106 (vmapped over envs in the reference)
        key = prng.split(np.asarray(key, dtype=np.uint32))[..., 0, :]
        # pinned staging + non_blocking copy: torch.tensor(ndarray, device=...) would wait for every kernel queued so far
        shift = torch.from_numpy(prng.normal_batch(key, 2) * np.float32(0.02)).pin_memory().to(state.x.device, non_blocking=True)   # [B,2]
        p = state.primitives[0]
        position = p.position.clone()
        position[:, 0, 0] = position[:, 0, 0] + shift[:, 0]
        position[:, 0, 2] = position[:, 0, 2] + shift[:, 1]
        x = state.x.clone()
        x[:, :, 0] = x[:, :, 0] + shift[:, None, 0]
        x[:, :, 2] = x[:, :, 2] + shift[:, None, 1]
        return state._replace(x=x, key=key, primitives=[p._replace(position=position)])

    @staticmethod
    def get_primitive_actions(actions, state):   # :108-115
        actions = actions + 1e-12  # hack to avoid nan
        actions = actions * _R50   # `/ 50.0` as XLA executes a division by a literal under jit: A * (1 / Const)
        actions = torch.cat([actions[..., :3], torch.zeros_like(actions[..., 3:])], -1)
        return actions[None, ...], state

    def reset(self, key):   # :117-137
        self.clean_up_b4_reset()
        set_sdf(box_sdf)
        state = self.simulator.add_box(conf=self.conf, state=None, hardness=self.conf.rope_hardness,
                                       size=self.conf.rope_width, init_pos=self.conf.rope_init_pos,
                                       z_rotation_angle=self.conf.rope_z_rotation_angle, material=1, density=2.75)
        state = self.create_primitive(self.conf, state=state, friction=0.1, color=[0.5, 0.5, 0.5],
                                      size=[0.02, 0.02, 0.02], init_pos=[0.5, 0.01, 0.3])
        self.initialize_after_adding_particle_primitives(state)
        self.state = self.auto_reset(self.init_state, self.init_state, self.init_state.key)
        return self.get_obs(self.state), self.state
