"""Analytic-Policy-Gradient learner shared by the apg / apg_para / apg_no_para entry points and bench.py.

Counterpart of /root/reference/DaXBench/daxbench/algorithms/apg/apg.py (same update rule, different host):
    policy MLP obs->512->256->2*act, swish            apg.py:353-358  (brax make_model: lecun_uniform, zero bias)
    NormalTanhDistribution sample                      apg.py:98-100   (scale = softplus(raw)+0.001, tanh)
    do_one_step / loss = -mean(reward)                 apg.py:177-215
    nan_to_num -> clip_by_global_norm -> pmean -> Adam apg.py:217-267
Multi-device: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI); environments are
sharded `num_envs // world` per rank (apg.py:83-85), parameters/Adam state replicated, and exactly ONE
collective per update: the mean all-reduce of the already-clipped flat gradient (apg.py:234-235).
brax / flax / optax are third-party and absent; their semantics are restated from SURVEY.md Appendix B.
"""
from __future__ import annotations

import math
import time

import torch
import torch.distributed as dist
import torch.nn.functional as Fn

from ...envs.basic.cloth_env import ClothEnv


class Policy(torch.nn.Module):
    def __init__(self, obs_size, action_size, hidden=(512, 256), seed=0):
        super().__init__()
        sizes = [obs_size, *hidden, 2 * action_size]
        g = torch.Generator().manual_seed(seed)
        self.layers = torch.nn.ModuleList()
        for i in range(len(sizes) - 1):
            lin = torch.nn.Linear(sizes[i], sizes[i + 1])
            bound = math.sqrt(3.0 / sizes[i])               # lecun_uniform: U(-sqrt(3/fan_in), +)
            with torch.no_grad():
                lin.weight.copy_((torch.rand(lin.weight.shape, generator=g) * 2 - 1) * bound)
                lin.bias.zero_()
            self.layers.append(lin)

    def forward(self, obs):
        h = obs
        for i, lin in enumerate(self.layers):
            h = lin(h)
            if i < len(self.layers) - 1:
                h = Fn.silu(h)                               # flax.linen.swish
        return h


def sample_action(logits, eps, min_std=0.001):
    loc, raw = torch.chunk(logits, 2, dim=-1)
    return torch.tanh(loc + (Fn.softplus(raw) + min_std) * eps)


class GradSync:
    """The data-parallel half of the APG update (apg.py:233-258), device-agnostic:
    nan_to_num -> per-device global-norm clip -> ONE mean all-reduce of the flat gradient -> Adam.
    All parameter gradients are views into one flat buffer, so the collective is a single bucket."""

    def __init__(self, module, learning_rate, max_gradient_norm):
        self.params = [p for p in module.parameters()]
        self.n_params = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat_grad = torch.zeros(self.n_params, device=dev)
        self._bind()
        self.max_gradient_norm = max_gradient_norm
        self.optimizer = torch.optim.Adam(self.params, lr=learning_rate, betas=(0.9, 0.999), eps=1e-8)

    def _bind(self):
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_grad(self):
        self.flat_grad.zero_()

    def step(self):
        off = 0
        for p in self.params:   # autograd accumulates in place; re-bind defensively if it ever replaced a .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[off:off + 1].data_ptr():
                if p.grad is not None:
                    self.flat_grad[off:off + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        g = self.flat_grad
        torch.nan_to_num_(g)                                                 # apg.py:233
        g_norm = torch.linalg.vector_norm(g)
        raw_norm = g_norm.clone()
        scale = torch.where(g_norm < self.max_gradient_norm, torch.ones_like(g_norm),
                            self.max_gradient_norm / g_norm)                 # apg.py:260-267 (per device, BEFORE the mean)
        g.mul_(scale)
        if dist.is_initialized() and dist.get_world_size() > 1:              # apg.py:235 lax.pmean
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            g.div_(dist.get_world_size())
        self.optimizer.step()
        return raw_norm


def shard_envs(num_envs: int, world: int) -> int:
    """Environments per rank (apg.py:83-85)."""
    if num_envs % world:
        raise ValueError(f"num_envs={num_envs} is not divisible by the number of GPUs {world}")
    return num_envs // world


class APG:
    def __init__(self, env, episode_length, learning_rate=1e-4, max_gradient_norm=1e9, seed=0, truncation_length=None):
        self.env = env
        self.episode_length = episode_length
        self.max_gradient_norm = max_gradient_norm
        self.truncation_length = truncation_length
        self.device = env.device
        self.is_cloth = isinstance(env, ClothEnv)            # sigmoid on actions for non-MPM envs (apg.py:185)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.policy = Policy(env.observation_size, env.action_size, seed=seed).to(self.device)
        self.sync = GradSync(self.policy, learning_rate, max_gradient_norm)
        self.params, self.n_params, self.flat_grad, self.optimizer = (
            self.sync.params, self.sync.n_params, self.sync.flat_grad, self.sync.optimizer)
        self.gen = torch.Generator(device=self.device).manual_seed(seed * 1000 + self.rank)
        self.want_lists = False   # the loss never reads obs_list/state_list (XLA dead-code-eliminates them)

    # ------------------------------------------------------------------------------------------------
    def get_obs(self, state):
        return self.env.get_obs(state)

    def rollout(self, state, deterministic_noise=None):
        """do_one_step scanned episode_length times (apg.py:177-215). Returns (rewards[T,B], states, actions)."""
        rewards, actions_l, states = [], [], []
        for t in range(self.episode_length):
            obs = self.get_obs(state)
            logits = self.policy(obs)
            eps = (torch.randn(logits.shape[0], logits.shape[1] // 2, device=self.device, generator=self.gen)
                   if deterministic_noise is None else deterministic_noise[t])
            actions = sample_action(logits, eps)
            if self.is_cloth:
                actions = torch.sigmoid(actions)
            if self.is_cloth:
                _, reward, done, info = self.env.step_diff(actions, state, want_lists=self.want_lists)
            else:
                _, reward, done, info = self.env.step_diff(actions, state)
            state = info["state"]
            if self.truncation_length and (t + 1) % self.truncation_length == 0:
                state = type(state)(*[v.detach() if torch.is_tensor(v) else v for v in state])
            rewards.append(reward)
            actions_l.append(actions)
            states.append(state)
        return torch.stack(rewards), states, actions_l

    def loss(self, state, deterministic_noise=None):
        rewards, states, actions = self.rollout(state, deterministic_noise)
        return -rewards.mean(), (rewards, states, actions)

    def minimize(self, state, deterministic_noise=None):
        """One APG update (apg.py:217-258). Returns metrics dict (tensors; no host sync here)."""
        self.sync.zero_grad()
        loss, (rewards, _, _) = self.loss(state, deterministic_noise)
        loss.backward()
        raw_norm = self.sync.step()
        return {"grad_norm": raw_norm, "reward": rewards.detach(), "loss": loss.detach()}

    @torch.no_grad()
    def evaluate(self, state, steps):
        rewards = []
        for _ in range(steps):
            obs = self.get_obs(state)
            eps = torch.randn(obs.shape[0], self.env.action_size, device=self.device, generator=self.gen)
            actions = sample_action(self.policy(obs), eps)
            if self.is_cloth:
                actions = torch.sigmoid(actions)
                _, reward, _, info = self.env.step_diff(actions, state, want_lists=False)
            else:
                _, reward, _, info = self.env.step_diff(actions, state)
            state = info["state"]
            rewards.append(reward)
        return torch.stack(rewards)


def init_distributed(gpus: int):
    """One process per GPU. Under torchrun (RANK/WORLD_SIZE set) join the RCCL group; otherwise single GPU."""
    import os
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        return dist.get_rank(), dist.get_world_size(), torch.device("cuda", local)
    if gpus > 1:
        raise SystemExit(f"--gpus {gpus}: launch one process per GPU, e.g. "
                         f"python -m torch.distributed.run --nproc-per-node {gpus} --master-addr 127.0.0.1 -m <module> ...")
    return 0, 1, torch.device("cuda", 0)


class Timer:
    def __init__(self):
        self.t = time.time()

    def lap(self):
        t = time.time()
        d, self.t = t - self.t, t
        return d
