"""shape_rope -- mirrors /root/reference/DaXBench/daxbench/core/envs/shape_rope_env.py:27-168
(DefaultConf :27-63, process_pre_step_actions :85-89, get_primitive_actions :91-121, random_push / random_policy
:123-151, reset :153-174).  Plastic rope (material 2, 582 particles on a 64x6x64 grid), one box pusher in soft-contact
mode (collide_batch, primitives.py:154-182), 30 macro sub-actions of 133 substeps per env.step."""
import os
from dataclasses import dataclass

import numpy as np
import torch

from ..engine.primitives.box import _sdf_batch as box_sdf
from ..engine.primitives.primitives import set_sdf
from ..utils import prng
from .basic.mpm_env import MPMEnv

my_path = os.path.dirname(os.path.abspath(__file__))


@dataclass
class DefaultConf:
    n_primitive = 1
    obs_type = MPMEnv.PARTICLE
    key = prng.PRNGKey(0)

    ground_friction: float = 0.9
    n_grid: int = 128
    dt: float = 0.5e-4
    primitive_action_steps = 30
    primitive_action_duration = 0.2  # seconds
    steps = int(primitive_action_duration / primitive_action_steps / dt)  # internal steps (= 133)
    E: float = 100
    nu: float = 0.1
    res: tuple = (n_grid // 2, 6, n_grid // 2)

    dx, inv_dx = 1 / n_grid, float(n_grid)
    p_vol, p_rho = (dx * 0.5) ** 2, 1
    p_mass = p_vol * p_rho
    gravity = (0, -9.8, 0)

    task = "shape_rope"
    goal_path = f"{my_path}/goals/{task}/goal.npy"

    # Rope property
    rope_width = [0.25, 0.006, 0.006]
    rope_init_pos = [0.5, 0.01, 0.5]
    rope_z_rotation_angle = 0
    rope_hardness = 1.0

    # kernel option, not a reference field (include/unidom_hip.h): the forward checkpoints the active grid cells (measured:
    # the rope touches 0.54 cells per particle) and the backward restores them instead of running p2g + grid op again
    grid_ckpt_cells = 2


ShapeRopeConfig = DefaultConf


class ShapeRopeEnv(MPMEnv):

    def __init__(self, batch_size, seed, max_steps=6, conf=None, aux_reward=False, device="cuda", **kwargs):
        conf = DefaultConf() if conf is None else conf
        self.conf = conf
        self.focus_computation = True
        super().__init__(conf, batch_size, max_steps, seed, self.focus_computation, device=device)   # soft contact
        self.aux_reward = aux_reward
        # the reference hard-codes 3540 (= 582*6 + 16*3, a legacy steps=16); with its own conf the observation is
        # 582*6 + steps*3.  Set from the state at reset.
        self.observation_size = 582 * 6 + conf.steps * 3

    def auto_reset(self, state, state_new, key):   # :79-83: "TODO complete auto reset" -> the new state unchanged
        return state_new

    @staticmethod
    def process_pre_step_actions(actions, shift):   # :85-89
        return torch.cat([actions[..., 0:3] + shift, actions[..., 3:] + shift], -1)

    @staticmethod
    def get_primitive_actions(actions, state):   # :91-121 (vmapped over envs in the reference)
        zero, y = torch.zeros_like(actions[:, :1]), torch.full_like(actions[:, :1], 0.01)
        start = torch.cat([actions[:, 0:1], y, actions[:, 2:3]], -1)
        end = torch.cat([actions[:, 3:4], y, actions[:, 5:6]], -1)
        norm = torch.linalg.norm(end - start, dim=-1, keepdim=True) + 1e-8
        vec = (end - start) / norm                     # max move length 0.3
        end = start + vec * norm.clamp(0.0, 0.3)
        p = state.primitives[0]
        position = torch.cat([start[:, None, :], p.position[:, 1:]], 1)
        state = state._replace(primitives=[p._replace(position=position)] + list(state.primitives[1:]))
        num_sub_steps = DefaultConf.primitive_action_steps
        act_push = (end - start) * float(np.float32(1.0) / np.float32(num_sub_steps))   # `/ num_sub_steps`, a literal under jit: A * (1 / Const)
        act_push = torch.cat([act_push[:, 0:1], zero, act_push[:, 2:3]], -1)
        n_primitive = DefaultConf.n_primitive
        sub = torch.cat([act_push, torch.zeros_like(act_push)] + [torch.zeros_like(act_push)] * (2 * (n_primitive - 1)), -1)
        return sub[None].expand(num_sub_steps, -1, -1), state

    def random_push(self, step=10):   # :123-130
        for _ in range(step):
            actions = self.random_policy(self.batch_size)
            actions[:, 1] = 0
            _, _, _, info = self.step_diff(torch.tensor(actions, dtype=torch.float32, device=self.device), self.state)
            self.state = info["state"]

    def random_policy(self, n_actions, radius=0.05):   # :132-151
        pc = self.state.x[0].detach().cpu().numpy()
        n_particles = pc.shape[0]
        p_ids = np.random.randint(0, n_particles, n_actions)
        end_list = pc[p_ids]
        angles = np.random.random((n_actions,)) * np.pi * 2
        end_list[:, 0] += np.cos(angles) * radius
        end_list[:, 2] += np.sin(angles) * radius
        start_list = pc[p_ids]
        start_list[:, 0] -= np.cos(angles) * radius
        start_list[:, 2] -= np.sin(angles) * radius
        return np.array([[*start_list[i], *end_list[i]] for i in range(n_actions)])

    def build_reset_state(self):
        set_sdf(box_sdf)
        state = self.simulator.add_box(conf=self.conf, state=None, hardness=self.conf.rope_hardness,
                                       size=self.conf.rope_width, init_pos=self.conf.rope_init_pos,
                                       z_rotation_angle=self.conf.rope_z_rotation_angle, material=2, density=3)
        state = self.create_primitive(self.conf, state=state, friction=0.1, color=[0.5, 0.5, 0.5],
                                      size=[0.015, 0.06, 0.015], init_pos=[0.5, 0.01, 0.45])
        self.initialize_after_adding_particle_primitives(state)
        self.observation_size = self.state.x.shape[1] * 6 + self.conf.steps * 3

    def reset(self, key):   # :153-174
        self.clean_up_b4_reset()
        self.build_reset_state()
        with torch.no_grad():
            self.random_push(step=2)
        return self.get_obs(self.state), self.state
