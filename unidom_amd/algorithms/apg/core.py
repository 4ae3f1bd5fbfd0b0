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

import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as Fn

from ...envs.basic.cloth_env import ClothEnv
from ...envs.basic.mpm_env import MPMEnv
from ...envs.shape_rope_env import ShapeRopeEnv
from ...utils import prng


def squashes_actions(env) -> bool:
    """The reference's predicate `not isinstance(core_env, MPMEnv) or isinstance(core_env, ShapeRopeEnv)`
    (apg.py:185, :297): these envs take sigmoid(actions) in [0,1] (pick / place points, pusher start / end points) and are
    reset from `split(key_env, devices)` every iteration; the others (whip_rope, pour_water, pour_soup) take the raw tanh
    sample and go through auto_reset (apg.py:90-91, :302-307)."""
    is_a = issubclass if isinstance(env, type) else isinstance         # an env class (registry entry) or an instance
    return (not is_a(env, MPMEnv)) or is_a(env, ShapeRopeEnv)


class Policy(torch.nn.Module):
    """brax 0.0.13 `networks.make_model([512, 256, 2*act], obs_size, activation=swish)` (apg.py:353-358): a flax MLP of
    Dense layers named hidden_0.. with lecun_uniform kernels and zero biases, initialised by `module.init(key_models, zeros)`
    (apg.py:107).  The kernels are drawn exactly as flax / jax draw them from `key` (utils/prng.py: the parameter key is
    fold_in(key, sha1("hidden_i")), the kernel [in, out] = uniform(k, (in, out), -1, 1) * sqrt(3 / fan_in) in f32) and
    stored transposed in torch's [out, in] layout.  No torch generator is involved."""

    def __init__(self, obs_size, action_size, hidden=(512, 256), seed=0, key=None):
        super().__init__()
        sizes = [obs_size, *hidden, 2 * action_size]
        key = prng.PRNGKey(seed) if key is None else np.asarray(key, np.uint32)
        self.layers = torch.nn.ModuleList()
        for i in range(len(sizes) - 1):
            fan_in, fan_out = sizes[i], sizes[i + 1]
            lin = torch.nn.Linear(fan_in, fan_out)
            k = prng.flax_param_key(key, (f"hidden_{i}",), 0)          # Dense's first make_rng('params') is the kernel's
            scale = np.sqrt(np.float32(3.0 / fan_in)).astype(np.float32)   # variance_scaling(1, "fan_in", "uniform")
            kernel = prng.uniform(k, fan_in * fan_out, -1.0, 1.0).reshape(fan_in, fan_out) * scale
            with torch.no_grad():
                lin.weight.copy_(torch.from_numpy(np.ascontiguousarray(kernel.T)))
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
    """brax NormalTanhDistribution(event_size).sample (apg.py:98-100, :184): loc, raw = split(logits, 2);
    tanh(loc + (softplus(raw) + min_std) * eps), eps = jax.random.normal(key_sample, loc.shape)."""
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
        self.profile = None    # a list: step() appends (start, end) around the all-reduce -- torch.cuda.Event pairs recorded on the current
                               # stream when the gradient lives on a GPU, perf_counter() seconds otherwise (bench.py: `allreduce_ms` per rank)

    def allreduce_ms(self):
        """milliseconds per all-reduce from `profile` (synchronise the device first), [] when nothing was recorded"""
        out = []
        for a, b in self.profile or []:
            out.append(a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3)
        return out

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
        if dist.is_initialized():                                            # apg.py:235 lax.pmean (a one-rank group reduces too: the identity)
            if self.profile is not None:
                cuda = g.device.type == "cuda"
                t0 = torch.cuda.Event(enable_timing=True) if cuda else time.perf_counter()
                if cuda:
                    t0.record(torch.cuda.current_stream(g.device))
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            if self.profile is not None:
                t1 = torch.cuda.Event(enable_timing=True) if cuda else time.perf_counter()
                if cuda:
                    t1.record(torch.cuda.current_stream(g.device))
                self.profile.append((t0, t1))
            g.div_(dist.get_world_size())
        self.optimizer.step()
        return raw_norm


def shard_envs(num_envs: int, world: int) -> int:
    """Environments per rank (apg.py:83-85)."""
    if num_envs % world:
        raise ValueError(f"num_envs={num_envs} is not divisible by the number of GPUs {world}")
    return num_envs // world


class APG:
    def __init__(self, env, episode_length, learning_rate=1e-4, max_gradient_norm=1e9, seed=0, truncation_length=None,
                 key=None, key_models=None):
        """`key` / `key_models`: the noise key carried in TrainingState and the model-init key (apg.py:76-80, :107, :276).
        Given only `seed` they are derived as the reference's train() derives them: key, key_models, _ = split(PRNGKey(seed), 3);
        key = split(key, process_count=1)[0].  Every device holds the SAME noise key (the reference replicates the
        TrainingState, apg.py:282-284), so every rank draws the same eps for its own environments."""
        self.env = env
        self.episode_length = episode_length
        self.max_gradient_norm = max_gradient_norm
        self.truncation_length = truncation_length
        self.device = env.device
        self.is_cloth = isinstance(env, ClothEnv)            # ClothEnv.step_diff takes want_lists
        self.squash = squashes_actions(env)                  # apg.py:185
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if key is None or key_models is None:
            k, km, _ = prng.split(prng.PRNGKey(seed), 3)
            key = prng.split(k, 1)[0] if key is None else key
            key_models = km if key_models is None else key_models
        self.key = np.asarray(key, np.uint32)
        self.policy = Policy(env.observation_size, env.action_size, key=key_models).to(self.device)
        self.sync = GradSync(self.policy, learning_rate, max_gradient_norm)
        self.params, self.n_params, self.flat_grad, self.optimizer = (
            self.sync.params, self.sync.n_params, self.sync.flat_grad, self.sync.optimizer)
        self.want_lists = False   # the loss never reads obs_list/state_list (XLA dead-code-eliminates them)

    # ------------------------------------------------------------------------------------------------
    def get_obs(self, state):
        return self.env.get_obs(state)

    def draw_noise(self, key, steps):
        """The noise of `steps` scanned do_one_step calls (apg.py:177-184): per step `key, key_sample = split(key)` and
        eps = jax.random.normal(key_sample, loc.shape = [B, action_size]).  The keys do not depend on the device state, so
        the whole episode's noise is drawn on the host up front and uploaded once.  Returns (key after the scan, eps [T,B,A])."""
        B, A = self.env.batch_size, self.env.action_size
        eps = np.empty((steps, B, A), np.float32)
        for t in range(steps):
            key, key_sample = prng.split(key)
            eps[t] = prng.normal(key_sample, B * A).reshape(B, A)
        host = torch.from_numpy(eps)
        if torch.device(self.device).type == "cuda":
            host = host.pin_memory()
        return key, host.to(self.device, non_blocking=True)

    def _act(self, state, eps):
        actions = sample_action(self.policy(self.get_obs(state)), eps)
        return torch.sigmoid(actions) if self.squash else actions              # apg.py:185-186

    def rollout(self, state, key=None, deterministic_noise=None):
        """do_one_step scanned episode_length times (apg.py:177-215). Returns (rewards[T,B], states, actions)."""
        if deterministic_noise is None:
            _, deterministic_noise = self.draw_noise(self.key if key is None else key, self.episode_length)
        rewards, actions_l, states = [], [], []
        for t in range(self.episode_length):
            actions = self._act(state, deterministic_noise[t])
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

    def loss(self, state, key=None, deterministic_noise=None):
        rewards, states, actions = self.rollout(state, key, deterministic_noise)
        return -rewards.mean(), (rewards, states, actions)

    def minimize(self, state, deterministic_noise=None):
        """One APG update (apg.py:217-258): key, key_grad = split(training_state.key); grad of loss(.., key_grad);
        nan_to_num, clip, mean over devices, Adam.  Returns metrics dict (tensors; no host sync here)."""
        self.key, key_grad = prng.split(self.key)
        self.sync.zero_grad()
        loss, (rewards, _, _) = self.loss(state, key_grad, deterministic_noise)
        loss.backward()
        raw_norm = self.sync.step()
        return {"grad_norm": raw_norm, "reward": rewards.detach(), "loss": loss.detach()}

    # ---- one update as ONE graph launch ----------------------------------------------------------------------------
    def capture(self, state, warmup=3):
        """Capture one APG update on `state` (policy, episode_length x step_diff, loss, backward, clip, Adam) into a HIP graph.
        An update of a small MPM env is ~200 launches, most of them a few microseconds of policy / reward / optimizer arithmetic
        between the simulator's kernels; replayed as one graph the GPU runs them back to back, the host issues a single launch.
        What makes it legal: every simulator call takes the capturing stream and allocates through torch (graph-private pool);
        the episode's noise is drawn on the host as before and copied into a static buffer before the replay; Adam runs in its
        capturable form (step count on the device).  `state` is bound at capture: minimize_captured() re-runs the update from THAT
        state (APG restarts every update from the reset state, apg.py:217-258).  Single process only (the gradient all-reduce
        stays eager).  Device-side status flags are written by every replay into the tensors recorded at capture."""
        if self.world > 1:
            raise RuntimeError("APG.capture: single-process only")
        dev = torch.device(self.device)
        T, B, A = self.episode_length, self.env.batch_size, self.env.action_size
        self._cap_noise = torch.zeros((T, B, A), device=dev)
        if not self.optimizer.defaults.get("capturable", False):
            sd = self.optimizer.state_dict()
            self.optimizer = self.sync.optimizer = torch.optim.Adam(self.params, lr=self.optimizer.defaults["lr"], betas=(0.9, 0.999), eps=1e-8, capturable=True)
            self.optimizer.load_state_dict({**sd, "param_groups": [{**g, "capturable": True} for g in sd["param_groups"]]})
        sim = getattr(self.env, "simulator", None)
        prof, had = (sim.profile, True) if sim is not None and hasattr(sim, "profile") else (None, False)
        if had:
            sim.profile = None                       # timing events cannot be queried from inside a graph
        # The capture runs on the CURRENT stream, which must be a non-default one and the stream every earlier eager update of this
        # learner ran on: autograd's per-parameter accumulator nodes remember the stream they were created on (they stay alive with
        # the last rollout's graph, e.g. through env state), and a backward that touches another stream inside a capture takes the
        # ROCm runtime down in hipStreamEndCapture rather than raising.  So: `with torch.cuda.stream(s):` around the learner's whole
        # life (bench.py does), capture() inside it.
        cur = torch.cuda.current_stream(dev)
        if cur == torch.cuda.default_stream(dev):
            raise RuntimeError("APG.capture: run the learner under a non-default stream (with torch.cuda.stream(s): ...), eager updates included")
        # The many-workgroup MPM path with a grid checkpoint hands a per-step overflow flag from the forward to the backward THROUGH THE
        # HOST (SimpleMPMSimulator._stage_flags: a side stream, pinned memory, an event the backward synchronises on): none of that can be
        # recorded into a graph -- the capture would fail, or bake in the flag of the capturing run.
        if sim is not None and getattr(sim, "_h_large", False) and getattr(sim, "grid_ckpt_cells", 0) > 0:
            raise RuntimeError("APG.capture: this env's simulator stages grid-checkpoint flags through the host per step "
                               "(many-workgroup MPM path with grid_ckpt_cells > 0); run it eagerly")
        # the warm-up updates below are real updates (with zero noise): parameters and Adam moments are put back afterwards, IN PLACE --
        # the graph holds the addresses of these very tensors
        saved_p = [p.detach().clone() for p in self.params]
        saved_s = [{k: v.detach().clone() for k, v in self.optimizer.state.get(p, {}).items() if torch.is_tensor(v)} for p in self.params]
        try:
            for _ in range(warmup):                  # eager warm-up: lazy initialisations, Adam state
                self._update(state, self._cap_noise)
            torch.cuda.synchronize(dev)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=cur):
                self._cap_out = self._update(state, self._cap_noise)
        finally:                                     # also when the capture raises: the learner keeps the parameters it came with
            with torch.no_grad():
                for p, sp, ss in zip(self.params, saved_p, saved_s):
                    p.copy_(sp)
                    for k, v in self.optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            v.copy_(ss[k]) if k in ss else v.zero_()     # no earlier state: a fresh Adam (zero moments, step 0)
            if had:
                sim.profile = prof
        return self

    def _update(self, state, noise):
        # torch.autograd.grad, not .backward(): the parameters' AccumulateGrad nodes were created by the eager updates on the default
        # stream and would run there, outside the capture (torch warns; hipStreamEndCapture then crashes).  The gradients land in the
        # flat buffer by copy -- the same values .backward() accumulates into its zeroed views.
        loss, (rewards, _, _) = self.loss(state, None, noise)
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        off = 0
        for p, g in zip(self.params, grads):
            dst = self.flat_grad[off:off + p.numel()]
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g.reshape(-1))
            off += p.numel()
        raw_norm = self.sync.step()
        return {"grad_norm": raw_norm, "reward": rewards.detach(), "loss": loss.detach()}

    def minimize_captured(self):
        """minimize() through the captured graph: same key schedule, same arithmetic; returns the (static) metric tensors."""
        self.key, key_grad = prng.split(self.key)
        _, noise = self.draw_noise(key_grad, self.episode_length)
        self._cap_noise.copy_(noise, non_blocking=True)
        self._graph.replay()
        return self._cap_out

    @torch.no_grad()
    def evaluate(self, state, steps, key=None):
        """run_eval (apg.py:127-149): `steps` scanned do_one_step_eval calls drawing their noise from `key` (key_debug)."""
        _, noise = self.draw_noise(self.key if key is None else key, steps)
        rewards = []
        for t in range(steps):
            actions = self._act(state, noise[t])
            if self.is_cloth:
                _, reward, _, info = self.env.step_diff(actions, state, want_lists=False)
            else:
                _, reward, _, info = self.env.step_diff(actions, state)
            state = info["state"]
            rewards.append(reward)
        return torch.stack(rewards)


def init_distributed(gpus: int):
    """One process per GPU. Under torchrun (RANK/WORLD_SIZE set) join the RCCL group; otherwise single GPU.
    UNIDOM_DIST_BACKEND=gloo (tests; rehearsals without GPUs) joins a gloo group on the CPU instead -- only host-side code
    (the launcher, the collective, the update rule) can run there, the simulators have no CPU path.  With
    UNIDOM_DIST_DEVICE=cuda:0 beside it every rank of the gloo group computes on that one GPU: the rehearsal of the whole
    N-rank flow on a one-GPU box (RCCL refuses two ranks on one device; gloo stages the gradient through the host)."""
    import os
    backend = os.environ.get("UNIDOM_DIST_BACKEND", "nccl")
    # UNIDOM_DIST_JOIN_SINGLE=1 (tests/test_z_rccl_gpu.py): join the group even when WORLD_SIZE is 1 -- the one-GPU box's way to
    # load RCCL, bind the device and run the update's collective once before the first multi-GPU run
    if "RANK" in os.environ and (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("UNIDOM_DIST_JOIN_SINGLE") == "1"):
        if backend == "gloo":
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo")
            dev = torch.device(os.environ.get("UNIDOM_DIST_DEVICE", "cpu"))
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
            return dist.get_rank(), dist.get_world_size(), dev
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        return dist.get_rank(), dist.get_world_size(), torch.device("cuda", local)
    if gpus > 1:
        raise SystemExit(f"--gpus {gpus}: launch one process per GPU, e.g. "
                         f"python -m torch.distributed.run --nproc-per-node {gpus} --master-addr 127.0.0.1 -m <module> ...")
    return 0, 1, (torch.device("cpu") if backend == "gloo" else torch.device("cuda", 0))


class Timer:
    def __init__(self):
        self.t = time.time()

    def lap(self):
        t = time.time()
        d, self.t = t - self.t, t
        return d
