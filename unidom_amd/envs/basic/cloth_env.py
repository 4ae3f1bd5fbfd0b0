"""ClothEnv -- the env base directly above the cloth hot path, same surface as the reference's.

Mirrors /root/reference/DaXBench/daxbench/core/envs/basic/cloth_env.py (and cloth_env_para.py for the
parameter-aware observation):
    __init__                :25-65      get_obs             :94-132   (para: cloth_env_para.py:101-137)
    get_pnp_actions         :134-173    build_reset/reset   :178-187
    step_diff               :201-231    get_collision_func  :239-243
The reference jit-compiles step_diff around lax.scan(simulator.step_jax); here the scan is one kernel launch
(ClothSimulator.rollout) and the surrounding reward / observation arithmetic is torch on the same stream.
Rendering (pyrender) is out of scope: step_with_render / render raise NotImplementedError.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from ...engine.cloth_simulator import ClothSimulator, ClothState
from ...utils import prng
from . import _fused
from .step_info import _StepInfo


# `/ 3` and `/ 20` of get_pnp_actions as the reference executes them under jit: multiplications by the f32 reciprocal (XLA's
# A / Const => A * (1 / Const)); the recorded demos discriminate (csrc/env_glue.hip)
_R3, _R20 = float(np.float32(1) / np.float32(3)), float(np.float32(1) / np.float32(20))


class ClothEnv:
    PARTICLE = "PARTICLE"
    DEPTH = "DEPTH"
    RGB = "RGB"

    def __init__(self, conf, batch_size, max_steps, aux_reward=False, eval_min_max_stiff=None, device="cuda"):
        assert conf
        cloth_mask = self.create_cloth_mask(conf)
        collision_func = self.get_collision_func()
        simulator = ClothSimulator(conf, batch_size, collision_func, cloth_mask, device=device)

        self.conf = conf
        self.aux_reward = aux_reward
        self.eval_min_max_stiff = eval_min_max_stiff   # None = plain env; [lo,hi] = parameter-aware obs
        self.simulator = simulator
        self.cloth_mask = simulator.cloth_mask
        self.max_steps = max_steps
        self.batch_size = simulator.batch_size
        self.cur_step = 0
        self.action_size = 6
        self.device = simulator.device
        self.seed(conf.seed)                            # Q12: the env seed is conf.seed, not the CLI seed

        assert conf.goal_path
        self.goal_path = conf.goal_path
        num_p = int(self.cloth_mask.sum())
        self.observation_size = num_p * 6 + 8           # overwritten by the task envs, as in the reference
        self.cloth_state_shape = (num_p, 6)
        self.spec = None
        self.idx_i, self.idx_j = simulator.idx_i, simulator.idx_j
        self.step_diff = self.build_step_diff()
        self.reset = self.build_reset()

        if not os.path.exists(conf.goal_path):
            print("**************** Warning: goal file does not exist!")
            self.goal = torch.zeros((1, 3), device=self.device)
        else:
            self.goal = torch.tensor(np.load(conf.goal_path), dtype=torch.float32, device=self.device)

    def seed(self, seed):
        self.simulator.key_global = prng.PRNGKey(seed)
        np.random.seed(seed)

    # ------------------------------------------------------------------------------------------------
    def get_obs(self, state: ClothState, eval_min_max_stiff=None, obs_type=PARTICLE):
        if obs_type != ClothEnv.PARTICLE:
            raise NotImplementedError("only PARTICLE observations are on the hot path")
        lead = state.x.shape[:-2]
        parts = [state.x.reshape(lead + (-1,)), state.primitive0, state.primitive1]
        mm = eval_min_max_stiff if eval_min_max_stiff is not None else self.eval_min_max_stiff
        if mm is not None:   # cloth_env_para.py:130
            lo, hi = float(mm[0]), float(mm[1])
            parts.append((state.stiffness.to(torch.float32)[..., None] - lo) / (hi - lo))
        return torch.cat(parts, -1)

    @staticmethod
    def get_pnp_actions(actions, state: ClothState):
        """[B,6] (pick xyz, place xyz) -> [40,B,8] macro actions (cloth_env.py:134-173)."""
        B, dev = actions.shape[0], actions.device
        zero = torch.zeros_like(actions[:, :1])
        pick = torch.cat([actions[:, 0:1], zero, actions[:, 2:3]], -1)       # :145
        place = torch.cat([actions[:, 3:4], zero, actions[:, 5:6]], -1)      # :146
        one = torch.ones_like(zero)
        act_down = torch.cat([(pick - state.primitive0[:, :3]) * _R3, one], -1)                       # :148-151
        act_up = torch.tensor([0, 0.06, 0, 0], device=dev).repeat(B, 1)
        act_up = torch.cat([act_up[:, :3] / 10, act_up[:, 3:]], -1)                                  # :154-156
        move = place - pick
        move = torch.cat([move[:, 0:1], zero, move[:, 2:3]], -1)                                     # :159-160
        act_move = torch.cat([move * _R20, zero], -1)                                                  # :161-163
        act_release = torch.tensor([0.0, 0, 0, 1], device=dev).repeat(B, 1)                          # :166
        sub = torch.cat([act_down[None].expand(3, B, 4), act_up[None].expand(10, B, 4),
                         act_move[None].expand(20, B, 4), act_release[None].expand(7, B, 4)], 0)
        return torch.cat([sub, torch.zeros_like(sub)], -1)                                           # :170-171

    def get_x_grid(self, state):
        return self.simulator.get_x_grid(state.x)

    def build_reset(self):
        init_state = self.simulator.reset_jax()

        def reset(key):   # :181-185 ; key: uint32[2] (host)
            key = prng.split(np.asarray(key, dtype=np.uint32))[0]
            shift = prng.normal(key, 2) * np.float32(0.05)
            new_x = init_state.x.clone()
            new_x[..., 0] += float(shift[0])
            new_x[..., 2] += float(shift[1])
            state = init_state._replace(x=new_x)
            return self.get_obs(state), state

        return reset

    def build_step_diff(self):
        def step_diff(actions, state: ClothState, want_lists=None):
            want_lists = self.conf.use_substep_obs if want_lists is None else want_lists
            x_before = state.x
            # :206-210 contact_distance and get_pnp_actions, one kernel (the op-by-op form is get_pnp_actions above)
            macro, contact_distance = _fused.pnp_and_contact(actions, state.primitive0, state.x)
            state, state_list = self.simulator.rollout(state, macro, want_lists=want_lists)          # :211
            state = state._replace(cur_step=state.cur_step + 1)                                      # :213
            obs = self.get_obs(state)
            obs_list = self.get_obs(state_list) if want_lists else obs                               # :216-219
            done = state.cur_step >= self.max_steps
            chamfer_distance = _fused.chamfer(state.x, self.goal)                                    # :222
            reward = torch.exp(chamfer_distance * -10.0)                                             # :223  e ** (-10 d)
            if self.aux_reward:
                reward = reward + torch.exp(-contact_distance)                                       # :225

            def real_reward():   # :205, :226 -- nothing on the gradient path reads it (XLA drops it from the jitted
                with torch.no_grad():   # loss), so the second chamfer pass runs only when somebody asks
                    return _fused.chamfer(x_before, self.goal) - chamfer_distance + 0.1 * contact_distance

            info = _StepInfo({"state": state, "obs_list": obs_list, "state_list": state_list}, real_reward=real_reward)
            reward = reward * torch.pow(0.99, state.cur_step.to(torch.float32))                      # :228
            return obs, reward, done, info

        return step_diff

    def step_with_render(self, actions, state, visualize=True):
        raise NotImplementedError("rendering (pyrender) is outside the hot path; see DESIGN.md")

    def render(self, state, visualize=True):
        raise NotImplementedError("rendering (pyrender) is outside the hot path; see DESIGN.md")

    def create_cloth_mask(self, conf):
        raise NotImplementedError

    def get_collision_func(self):
        def collision_func(x, v, idx_i, idx_j):   # :239-243 identity
            return v

        return collision_func

    @staticmethod
    def get_random_fold_action(state: ClothState):   # :323-333
        B, P = state.x.shape[0], state.x.shape[1]
        st = np.random.randint(0, P, size=(B,))
        ed = np.random.randint(0, P, size=(B,))
        bi = torch.arange(B, device=state.x.device)
        return torch.cat((state.x[bi, torch.as_tensor(st, device=state.x.device)],
                          state.x[bi, torch.as_tensor(ed, device=state.x.device)]), -1)
