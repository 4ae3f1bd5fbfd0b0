"""MPMEnv -- the env base directly above the MPM hot path, same surface as the reference's.

Mirrors /root/reference/DaXBench/daxbench/core/envs/basic/mpm_env.py:
    __init__ :23-51    get_obs :57-76    reward_func :90-94    pre_step / post_step (focus shift) :99-125
    step_diff :130-167    initialize_after_adding_particle_primitives :195-198    create_primitive :207-217
The reference jit-compiles step_diff around lax.scan(simulator.step_jax); here each scanned `step` is one kernel
launch (SimpleMPMSimulator.step_jax); the focus shift before it and un-shift / nan_to_num / reward / observation after it
are one kernel each way (csrc/env_glue.hip), with the op-by-op torch form kept as `step_diff_unfused`.
Rendering (pyrender / trimesh) is out of scope.
"""
from __future__ import annotations

import math
import os
import sys
import weakref

import numpy as np
import torch

from ...engine.mpm_simulator import MPMState, SimpleMPMSimulator
from ...engine.primitives.primitives import create_primitive
from ...utils import prng
from ...utils.util import calc_l2
from . import _fused
from .step_info import _StepInfo


def _where_done(done, new, old, done_host=None):
    """jnp.where(right_broadcasting(done, x), y, x) over a (nested) state (mpm_env.py:161); `done_host` is the host copy
    of `done`, used for the leaves that live on the host (the PRNG keys) so that selecting them does not wait for the device.
    Leaves that auto_reset handed back untouched (the same object) select between two identical values: returned as is,
    which spares a launch for each of the ~20 constant primitive fields on every step."""
    if new is old:
        return old
    if torch.is_tensor(old):
        d = done.reshape(done.shape + (1,) * (old.dim() - done.dim()))
        return torch.where(d, new.to(old.dtype), old)
    if isinstance(old, np.ndarray):
        d = (done.detach().cpu().numpy() if done_host is None else np.asarray(done_host))
        return np.where(d.reshape(d.shape + (1,) * (old.ndim - d.ndim)), new, old)
    if isinstance(old, list):
        return [_where_done(done, n, o, done_host) for n, o in zip(new, old)]
    if isinstance(old, tuple):
        return type(old)(*[_where_done(done, n, o, done_host) for n, o in zip(new, old)])
    return old


def _detach(x):
    if torch.is_tensor(x):
        return x.detach()
    if isinstance(x, list):
        return [_detach(v) for v in x]
    if isinstance(x, tuple) and hasattr(x, "_fields"):
        return type(x)(*[_detach(v) for v in x])
    return x


class MPMEnv:
    PARTICLE = "PARTICLE"
    DEPTH = "DEPTH"
    RGB = "RGB"

    def __init__(self, conf, batch_size, max_steps, seed, focus_computation=False, use_position_control=False, device="cuda"):
        self.simulator = SimpleMPMSimulator(conf, batch_size, use_position_control, device=device)
        self.aux_reward = False
        self.max_steps = max_steps
        self.batch_size = batch_size
        self.cur_step = 0
        self.action_size = 6
        self.device = self.simulator.device
        self.seed(seed)
        self.goal_path = conf.goal_path
        self.conf = conf
        self.focus_computation = focus_computation
        self.observation_size = 0
        self.spec = None
        self.state = None
        self.init_state = None
        self._target_center = None
        self._step_mirror = {}
        self.step_diff = self.build_step_diff()
        self.step_diff_unfused = self.build_step_diff(fused=False)
        if not os.path.exists(conf.goal_path):
            print("**************** Warning: goal file does not exist!", file=sys.stderr)   # the reference prints it to stdout
            self.goal = torch.zeros((1, 3), device=self.device)
        else:
            self.goal = torch.tensor(np.load(conf.goal_path), dtype=torch.float32, device=self.device)

    def seed(self, seed):
        self.simulator.key_global = prng.PRNGKey(seed)
        np.random.seed(seed)

    @staticmethod
    def get_obs(state: MPMState, obs_type=PARTICLE):   # :57-76
        if obs_type != MPMEnv.PARTICLE:
            raise NotImplementedError
        lead = state.x.shape[:-2]
        return torch.cat([state.x.reshape(lead + (-1,)), state.v.reshape(lead + (-1,)),
                          state.primitives[0].position.reshape(lead + (-1,))], -1)

    @staticmethod
    def get_primitive_actions(actions, state):
        raise NotImplementedError

    @staticmethod
    def process_pre_step_actions(actions, shift):
        raise NotImplementedError

    def auto_reset(self, state, state_new, key):
        raise NotImplementedError

    @staticmethod
    def reward_func(state, goal):   # :90-94
        return math.e ** (-calc_l2(state.x, goal) * 10)

    # -- host mirror of cur_step ---------------------------------------------------------------------------
    # `done` decides whether auto_reset's values are selected; the reference traces it (jnp.where).  Here the env keeps
    # a host copy of every cur_step tensor it hands out, so a step on which no env finishes (all but one in max_steps)
    # skips auto_reset + the per-leaf select, and the host never waits for the device to learn `done`.
    def _host_steps(self, cur_step):
        ent = self._step_mirror.get(id(cur_step))
        if ent is not None and ent[0]() is cur_step:
            return ent[1]
        host = cur_step.detach().cpu().numpy().copy()      # a state this env has not produced: one sync, then mirrored
        self._remember_steps(cur_step, host)
        return host

    def _remember_steps(self, cur_step, host):
        if len(self._step_mirror) > 256:
            self._step_mirror = {k: e for k, e in self._step_mirror.items() if e[0]() is not None}
        self._step_mirror[id(cur_step)] = (weakref.ref(cur_step), host)

    def build_step_diff(self, fused=True):
        """fused=True: focus shift and the tail of step_diff (un-shift, nan_to_num, reward, obs) run as one kernel each way
        (csrc/env_glue.hip); fused=False is the same arithmetic op by op in torch, the readable form the tests compare with."""
        fused = fused and type(self).reward_func is MPMEnv.reward_func and type(self).get_obs is MPMEnv.get_obs

        def target_center():
            if self._target_center is None:   # constant of the conf: built once, not an H2D copy per step
                self._target_center = torch.tensor(self.conf.res, dtype=torch.float32, device=self.device) * 0.5 / self.conf.n_grid
            return self._target_center

        def pre_step(actions, state: MPMState):   # :99-114
            if fused:
                c = (np.asarray(self.conf.res, np.float32) * np.float32(0.5) / np.float32(self.conf.n_grid)).tolist()
                x, shift, pos = _fused.mpm_focus(c, state.x, [p.position for p in state.primitives])
                actions = self.process_pre_step_actions(actions, shift)
                prims = [p._replace(position=q) for p, q in zip(state.primitives, pos)]
                return actions, state._replace(x=x, primitives=prims), shift          # [B,3] here, [B,1,3] below
            state_center = state.x.mean(1)
            shift = target_center() - state_center
            shift = torch.cat([shift[:, 0:1], torch.zeros_like(shift[:, 0:1]), shift[:, 2:3]], -1)
            actions = self.process_pre_step_actions(actions, shift)
            shift = shift[:, None, :]
            prims = [p._replace(position=p.position + shift) for p in state.primitives]
            return actions, state._replace(x=state.x + shift, primitives=prims), shift

        def post_step(state, state_list, shift):   # :116-125
            state = state._replace(x=state.x - shift, primitives=[p._replace(position=p.position - shift) for p in state.primitives])
            return state, post_step_list(state_list, shift)

        def post_step_list(state_list, shift):
            return state_list._replace(
                x=state_list.x - shift[None], primitives=[p._replace(position=p.position - shift[None]) for p in state_list.primitives])

        def stack_states(states):
            first = states[0]
            if len(states) == 1:               # one scanned step (whip_rope, pour_water): a leading axis, not a copy
                lead = lambda v: v.unsqueeze(0) if torch.is_tensor(v) else np.asarray(v)[None]
                fields = {}
                for name in first._fields:
                    val = getattr(first, name)
                    if name == "primitives":
                        fields[name] = [type(q)(*[lead(t) for t in q]) for q in val]
                    else:
                        fields[name] = lead(val)
                return type(first)(**fields)
            fields = {}
            for name in first._fields:
                vals = [getattr(s, name) for s in states]
                if name == "primitives":
                    fields[name] = [type(vals[0][i])(*[torch.stack([v[i][j] for v in vals]) for j in range(len(vals[0][i]))])
                                    for i in range(len(vals[0]))]
                elif torch.is_tensor(vals[0]):
                    fields[name] = torch.stack(vals)
                else:
                    fields[name] = np.stack(vals)
            return type(first)(**fields)

        def step_diff(actions, state: MPMState):   # :130-167
            contact_distance = None
            if self.aux_reward:   # :131-132 (only the auxiliary reward reads it)
                pickup_place = actions[..., :3]
                contact_distance = torch.sqrt(((pickup_place[:, None, :] - state.x) ** 2).sum(-1)).min(-1).values
            steps_h = self._host_steps(state.cur_step) + 1
            shift = None
            if self.focus_computation:
                actions, state, shift = pre_step(actions, state)
            actions, state = self.get_primitive_actions(actions, state)      # [T,B,6]
            states = []
            for t in range(actions.shape[0]):                                # lax.scan(simulator.step_jax, ...)
                state, _ = self.simulator.step_jax(state, actions[t])
                states.append(state)
            state = state._replace(cur_step=state.cur_step + 1)
            done = state.cur_step >= self.max_steps
            if fused:
                # state_list / obs_list are built on first access: nothing on the gradient path reads them (XLA drops them
                # from the jitted loss), and for a single scanned step they repeat `state` / `obs`
                def make_state_list():
                    sl = stack_states(states)
                    return post_step_list(sl, shift[:, None, :]) if self.focus_computation else sl

                (x, v, Cm, F, J), reward, obs, pos = _fused.mpm_finish(
                    state.x, state.v, state.C, state.F, state.J, shift, self.goal,
                    [p.position for p in state.primitives])                                          # :116-125, :150-154, :90-94, :57-76
                state = state._replace(x=x, v=v, C=Cm, F=F, J=J, primitives=[p._replace(position=q) for p, q in zip(state.primitives, pos)])
            else:
                state_list = stack_states(states)
                if self.focus_computation:
                    state, state_list = post_step(state, state_list, shift)
                state = state._replace(x=torch.nan_to_num(state.x), v=torch.nan_to_num(state.v), C=torch.nan_to_num(state.C),
                                       F=torch.nan_to_num(state.F), J=torch.nan_to_num(state.J))   # :150-154
                reward = self.reward_func(state, self.goal)
                obs = None
            if self.aux_reward:
                reward = reward + math.e ** (-contact_distance)
            done_h = steps_h >= self.max_steps
            if done_h.any() or not fused:
                new_state = _detach(self.auto_reset(self.init_state, state, state.key))              # :159-160
                state = _where_done(done, new_state, state, done_h)
                steps_h = np.where(done_h, self._host_steps(self.init_state.cur_step), steps_h)
                obs = None
            self._remember_steps(state.cur_step, steps_h)
            if obs is None:
                obs = self.get_obs(state)
            if fused:
                info = _StepInfo({"state": state}, state_list=make_state_list, obs_list=lambda: self.get_obs(info["state_list"]))
            else:
                info = {"state": state, "state_list": state_list, "obs_list": self.get_obs(state_list)}
            return obs, reward, done, info

        return step_diff

    def step_with_render(self, actions, state, visualize=True):
        raise NotImplementedError("rendering (pyrender) is outside the hot path; see DESIGN.md")

    def render(self, state, visualize=False):
        raise NotImplementedError("rendering (pyrender) is outside the hot path; see DESIGN.md")

    def clean_up_b4_reset(self):
        self.state = None

    def initialize_after_adding_particle_primitives(self, state):   # :195-198
        self.state = self.simulator.reset_jax(state)
        self.init_state = self.state

    def create_primitive(self, conf, state, friction, color, size, init_pos, softness=666):   # :207-217
        p_state = create_primitive(conf, friction=friction, softness=softness, color=color, size=np.array(size),
                                   init_pos=np.array(init_pos), device=self.device)
        state.primitives.append(p_state)
        return state

    def reset(self, key):
        raise NotImplementedError
