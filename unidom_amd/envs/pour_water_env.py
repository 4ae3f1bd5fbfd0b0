"""pour_water -- mirrors /root/reference/DaXBench/daxbench/core/envs/pour_water_env.py:29-133
(DefaultConf :29-62, get_primitive_actions :77-90, auto_reset :96-105, reset :114-133).  702 liquid particles
(material 0: mu = 0, la = 1), two bowl primitives with the container SDF (container.py:8-16) in soft-contact mode; the
policy moves and tilts the first bowl, the second one stands still (its six action components are zeros)."""
import os
from dataclasses import dataclass

import numpy as np
import torch

from ..engine.primitives.container import _sdf_batch as container_sdf
from ..engine.primitives.primitives import set_sdf
from ..utils import prng
from .basic.mpm_env import MPMEnv

my_path = os.path.dirname(os.path.abspath(__file__))
_R500 = float(np.float32(1.0) / np.float32(500.0))   # f32 reciprocal of the literal in get_primitive_actions (:80)


@dataclass
class DefaultConf:
    seed = 1
    n_primitive = 2
    obs_type = MPMEnv.PARTICLE
    key = prng.PRNGKey(0)

    ground_friction: float = 0.1
    n_grid: int = 80
    dt: float = 3e-4
    primitive_action_steps = 1
    primitive_action_duration = 0.007  # seconds
    steps = int(primitive_action_duration / primitive_action_steps / dt)  # internal steps (= 23)
    E: float = 0.00005
    nu: float = 0.4999

    res: tuple = (n_grid // 3, n_grid // 4, n_grid // 3)

    dx, inv_dx = 1 / n_grid, float(n_grid)
    p_vol, p_rho = (dx * 0.5) ** 2, 1
    p_mass = p_vol * p_rho
    gravity = (0, -9.8, 0)

    # kernel option, not a reference field (include/unidom_hip.h): the liquid is sampled uniformly (no spatial order), the
    # kernels re-order it by grid cell internally at every step
    sort_particles = 1
    # the forward checkpoints the active grid cells for the backward (measured 0.97 cells per particle at rest); a step in
    # which the splashing liquid needs more than the pool holds falls back to recomputing the grid, silently
    grid_ckpt_cells = 4

    task = "pour_water"
    goal_path = f"{my_path}/goals/{task}/goal.npy"


PourWaterConfig = DefaultConf


class PourWaterEnv(MPMEnv):

    def __init__(self, batch_size, seed, max_steps=100, conf=None, aux_reward=False, device="cuda", **kwargs):
        conf = DefaultConf() if conf is None else conf
        self.conf = conf
        super().__init__(conf, batch_size, max_steps, seed, focus_computation=True, device=device)   # soft contact
        self.observation_size = 4281   # 702 * 6 + 23 * 3

    @staticmethod
    def get_primitive_actions(actions, state):   # :77-90 (vmapped over envs in the reference)
        actions = torch.cat([actions, torch.zeros_like(actions)], -1)          # second bowl: dummy action
        actions = torch.cat([actions[..., :6] * _R500, actions[..., 6:]], -1)  # `/ 500.0`: normalise translation and rotation of bowl 0 (XLA: A * (1 / Const))
        actions = actions + 1e-12
        actions = torch.cat([actions[..., :1], torch.zeros_like(actions[..., :1]), actions[..., 2:]], -1)   # no vertical motion
        return actions[None], state

    @staticmethod
    def process_pre_step_actions(actions, shift):
        return actions

    def auto_reset(self, state, state_new, key):   # :96-105: the first bowl back to its start, jittered in x and z
        key = prng.split(np.asarray(key, dtype=np.uint32))[..., 0, :]
        noise = torch.from_numpy(prng.normal_batch(key, 2) * np.float32(0.02)).pin_memory().to(state.x.device, non_blocking=True)   # [B,2]
        p = state.primitives[0]
        init_pos = torch.tensor([0.5, 0.2, 0.5], device=state.x.device).repeat(noise.shape[0], 1)
        init_pos[:, 0] = init_pos[:, 0] + noise[:, 0]
        init_pos[:, 2] = init_pos[:, 2] + noise[:, 1]
        position = torch.cat([init_pos[:, None, :], p.position[:, 1:]], 1)
        return state._replace(key=key, primitives=[p._replace(position=position)] + list(state.primitives[1:]))

    def reset(self, key):   # :114-133
        self.clean_up_b4_reset()
        set_sdf(container_sdf)
        state = self.simulator.add_box(conf=self.conf, state=None, hardness=1, size=[0.07, 0.07, 0.07], init_pos=[0.5, 0.2, 0.5],
                                       z_rotation_angle=0, material=0, density=4)
        box_size = np.array([[0.09, 0., 0.008], [0.08, 0., 0.008]])
        self.create_primitive(self.conf, state=state, friction=0.1, softness=666, color=[0.5, 0.5, 0.5],
                              size=box_size[0], init_pos=[0.5, 0.2, 0.5])
        self.create_primitive(self.conf, state=state, friction=0.1, softness=666, color=[0.5, 0.5, 0.5],
                              size=box_size[1], init_pos=[0.5, 0.06, 0.3])
        self.initialize_after_adding_particle_primitives(state)
        self.state = self.auto_reset(self.init_state, self.init_state, self.init_state.key)
        return self.get_obs(self.state), self.state
