"""ClothSimulator -- host-side mirror of the reference's mass-spring simulator over the HIP kernels.

Mirrors /root/reference/DaXBench/daxbench/core/engine/cloth_simulator.py:
    ClothState                     :13-23
    ClothSimulator.__init__        :26-70    (same constructor arguments and attributes)
    step_jax(state, action)        :68-70    one robot_step (50 substeps) for the whole batch
    reset_jax()                    :339-364
    get_x_grid(x)                  :366-368
    indices                        :72-103   (render triangles)
plus `rollout(state, actions[T,B,8])`, the fused equivalent of
`lax.scan(simulator.step_jax, state, actions)` (cloth_env.py:211): ONE kernel launch for all T macro steps.

All physics runs in libunidom_hip.so (csrc/cloth.hip); torch is used for device memory, streams and to hook
the kernels' forward / adjoint pair into autograd (torch.autograd.Function), nothing else.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import numpy as np
import torch

from .. import _lib
from ..utils import prng


# `x / 50.` (cloth_simulator.py:168-169) as the reference executes it: under jit XLA rewrites a division by a compile-time
# constant into a multiplication by its f32 reciprocal, and the recorded demos tell the two apart
_R50 = float(np.float32(1.0) / np.float32(50.0))


class ClothState(NamedTuple):  # cloth_simulator.py:13-23
    x: torch.Tensor            # [B,P,3]
    v: torch.Tensor            # [B,P,3]
    primitive0: torch.Tensor   # [B,4] (pos xyz, radius)
    primitive1: torch.Tensor   # [B,4]
    action0: torch.Tensor      # [B,4] (dxyz per substep, suction)
    action1: torch.Tensor      # [B,4]
    key: np.ndarray            # [B,2] uint32 (host; never differentiated)
    cur_step: torch.Tensor     # [B] int32
    stiffness: torch.Tensor    # [B]
    mu: torch.Tensor           # [B]


class _Rollout(torch.autograd.Function):
    """forward = ud_cloth_rollout_fwd, backward = ud_cloth_rollout_bwd (include/unidom_hip.h)."""

    @staticmethod
    def forward(ctx, sim, x, v, prim, stiffness, mu, actions, want_lists):
        L = _lib.lib()
        B, P = x.shape[0], sim.n_particles
        T = actions.shape[0]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        x, v, prim, stiffness, mu, actions = map(f32, (x, v, prim, stiffness, mu, actions))
        assert x.shape == (B, P, 3) and v.shape == (B, P, 3) and prim.shape == (B, 2, 4)
        assert stiffness.shape == (B,) and mu.shape == (B,) and actions.shape == (T, B, 8)
        dev = x.device
        xo, vo, po = torch.empty_like(x), torch.empty_like(v), torch.empty_like(prim)
        xl = torch.empty((T, B, P, 3), device=dev) if want_lists else None
        vl = torch.empty((T, B, P, 3), device=dev) if want_lists else None
        pl = torch.empty((T, B, 2, 4), device=dev) if want_lists else None
        need_grad = any(ctx.needs_input_grad)  # all False under torch.no_grad()
        ckpt = None
        if need_grad:
            nbytes = L.ud_cloth_ckpt_bytes(sim._h, C.c_int(B), C.c_int(T))
            ckpt = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev)
        grasp = torch.zeros((T, sim.substeps, B, 2, P), dtype=torch.uint8, device=dev) if sim.record_grasp else None
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ev = sim._prof_begin("fwd")
        _lib.check(L.ud_cloth_rollout_fwd(
            sim._h, C.c_int(B), C.c_int(T), _lib.ptr(x), _lib.ptr(v), _lib.ptr(prim), _lib.ptr(stiffness),
            _lib.ptr(mu), _lib.ptr(actions), _lib.ptr(xo), _lib.ptr(vo), _lib.ptr(po), _lib.ptr(xl),
            _lib.ptr(vl), _lib.ptr(pl), _lib.ptr(ckpt), _lib.ptr(grasp), stream), "ud_cloth_rollout_fwd")
        sim._prof_end(ev)
        sim.last_grasp = grasp
        ctx.sim, ctx.B, ctx.T, ctx.want_lists = sim, B, T, want_lists
        ctx.save_for_backward(ckpt, stiffness, mu, actions)
        if want_lists:
            return xo, vo, po, xl, vl, pl
        return xo, vo, po

    @staticmethod
    def backward(ctx, *g):
        L = _lib.lib()
        sim, B, T = ctx.sim, ctx.B, ctx.T
        ckpt, stiffness, mu, actions = ctx.saved_tensors
        if ckpt is None:
            raise _lib.UnidomError("cloth rollout backward without a checkpoint (forward ran under no_grad)")
        P, dev = sim.n_particles, stiffness.device
        z = lambda t, shape: (torch.zeros(shape, device=dev) if t is None else t.to(torch.float32).contiguous())
        gx, gv, gp = z(g[0], (B, P, 3)), z(g[1], (B, P, 3)), z(g[2], (B, 2, 4))
        o = lambda t: None if t is None else t.to(torch.float32).contiguous()
        gxl, gvl, gpl = (o(g[3]), o(g[4]), o(g[5])) if ctx.want_lists else (None, None, None)
        gx0, gv0, gp0 = torch.empty_like(gx), torch.empty_like(gv), torch.empty_like(gp)
        ga = torch.empty((T, B, 8), device=dev)
        gk, gmu = torch.empty((B,), device=dev), torch.empty((B,), device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ev = sim._prof_begin("bwd")
        _lib.check(L.ud_cloth_rollout_bwd(
            sim._h, C.c_int(B), C.c_int(T), _lib.ptr(ckpt), _lib.ptr(stiffness), _lib.ptr(mu), _lib.ptr(actions),
            _lib.ptr(gx), _lib.ptr(gv), _lib.ptr(gp), _lib.ptr(gxl), _lib.ptr(gvl), _lib.ptr(gpl),
            C.c_int(1 if sim.normalize_grad else 0), _lib.ptr(gx0), _lib.ptr(gv0), _lib.ptr(gp0), _lib.ptr(ga),
            _lib.ptr(gk), _lib.ptr(gmu), stream), "ud_cloth_rollout_bwd")
        sim._prof_end(ev)
        return None, gx0, gv0, gp0, gk, gmu, ga, None


class ClothSimulator:
    def __init__(self, conf, batch_size, collision_func, cloth_mask, device="cuda", mode=None):
        assert batch_size >= 1
        self.conf = conf
        self.batch_size = batch_size
        self.collision_func = collision_func  # the reference's is the identity (cloth_env.py:239-243)
        self.cloth_mask = np.asarray(cloth_mask.cpu() if torch.is_tensor(cloth_mask) else cloth_mask, dtype=np.float32)
        self.x_grid = None
        self.device = torch.device(device)

        self.N = conf.N
        self.cell_size = 1.0 / self.N
        self.gravity = conf.gravity
        self.stiffness = conf.stiffness
        self.damping = conf.damping
        self.dt = conf.dt
        self.max_v = conf.max_v
        self.small_num = conf.small_num
        self.mu = conf.mu
        self.seed = conf.seed
        self.key_global = prng.PRNGKey(self.seed)
        self.substeps = int(getattr(conf, "substeps", 50))   # cloth_simulator.py:176 (fori_loop bound; tests shorten it)
        self.normalize_grad = True           # live norm_grad, :182-196
        self.record_grasp = False            # tests: capture the gripper masks (Q3)
        self.last_grasp = None
        # kernel family (include/unidom_hip.h ud_cloth_conf.mode): 0 (default) forward in operation order "v2" (the reference's
        # formulas re-associated; bit-identical to the CPU restatement of the same order) + restructured adjoint, 1 forward
        # and adjoint in the reference's literal operation order, 2 fast-math v2 forward + restructured adjoint, 3 forward in
        # the reference's literal operation order (cloth_simulator.py:257-337 as written) + restructured adjoint
        self.mode = int(getattr(conf, "kernel_mode", 0) if mode is None else mode)
        self.profile = None                  # bench.py: {"fwd": [...], "bwd": [...]} lists of (start, end) events
        self._suction_col = torch.tensor([[False, False, False, True] * 2], device=self.device)   # columns robot_step leaves unscaled

        self.num_triangles = (self.N - 1) * (self.N - 1) * 2
        idx_i, idx_j = np.nonzero(self.cloth_mask)            # :52 (row-major)
        self.idx_i, self.idx_j = idx_i, idx_j
        self.grid_idx = np.stack([idx_i, idx_j], -1)
        self.n_particles = len(idx_i)
        self.set_indices()

        cc = _lib.ud_cloth_conf(N=self.N, gravity=float(conf.gravity), damping=float(conf.damping),
                                dt=float(conf.dt), max_v=float(conf.max_v), small_num=float(conf.small_num),
                                substeps=self.substeps, mode=self.mode, max_envs=int(batch_size),
                                one_workgroup_per_env=int(bool(getattr(conf, "one_workgroup_per_env", False))))
        mask_u8 = np.ascontiguousarray(self.cloth_mask != 0, dtype=np.uint8)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ud_cloth_create(C.byref(cc), mask_u8.ctypes.data_as(C.c_void_p), C.byref(self._h)),
                       "ud_cloth_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ud_cloth_destroy(self._h)
        except Exception:
            pass

    # -- :72-103 render triangles (a square is kept iff its whole 3x3 neighbourhood is cloth) ------------
    def set_indices(self):
        N, m = self.N, self.cloth_mask
        sq = []
        for i in range(N - 1):
            for j in range(N - 1):
                if np.all(m[[i - 1, i - 1, i - 1, i, i, i, i + 1, i + 1, i + 1],
                            [j - 1, j, j + 1, j - 1, j, j + 1, j - 1, j, j + 1]] != 0):
                    a, b_, c, d = i * N + j, (i + 1) * N + j, i * N + j + 1, (i + 1) * N + j + 1
                    sq.append((a, b_, c))
                    sq.append((d, c, b_))
        self.indices = np.asarray(sq, dtype=np.float32).reshape(-1, 3)

    def _prof_begin(self, kind):
        if self.profile is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream(self.device))   # same stream the kernel is launched on
        return (kind, a, b)

    def _prof_end(self, ev):
        if ev is not None:
            ev[2].record(torch.cuda.current_stream(self.device))
            self.profile[ev[0]].append((ev[1], ev[2]))

    def check_status(self):
        """Device-side failure flags of this handle (include/unidom_hip.h: ud_cloth_poll_timeouts): a part of the
        several-workgroup kernels that gave up waiting for a sibling.  Synchronises the current stream; raises UnidomError.
        apg.train calls it once per iteration (the update has synchronised by then); one-workgroup bodies return at once."""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        n = _lib.lib().ud_cloth_poll_timeouts(self._h, stream)
        if n != 0:
            raise _lib.UnidomError(f"cloth kernels (status {n}): {_lib.lib().ud_last_error().decode()}")

    def launch_envs(self, B=None):
        """envs per kernel launch of a call with B envs (ud_cloth_launch_envs)"""
        return int(_lib.lib().ud_cloth_launch_envs(self._h, C.c_int(self.batch_size if B is None else B)))

    # -- state helpers -------------------------------------------------------------------------------
    def reset_jax(self) -> ClothState:  # :339-364
        N, c = self.N, self.cell_size
        ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
        xg = np.stack([ii * c, np.zeros_like(ii, dtype=np.float64), (N - jj) * c], -1)
        self.x_grid = torch.tensor(xg, dtype=torch.float32, device=self.device)
        B, dev = self.batch_size, self.device
        x = self.x_grid[self.idx_i, self.idx_j]
        rep = lambda t: t[None].repeat((B,) + (1,) * t.dim()).contiguous()
        key = prng.split(self.key_global)[0]
        stiff = torch.tensor(self.conf.stiffness, device=dev,
                             dtype=torch.int32 if isinstance(self.conf.stiffness, int) else torch.float32)
        return ClothState(
            x=rep(x), v=rep(torch.zeros_like(x)),
            primitive0=rep(torch.tensor([0.5, 0.5, 0.5, 0.01], device=dev)),
            primitive1=rep(torch.tensor([1.0, 1.0, 1.0, 0.01], device=dev)),
            action0=torch.zeros((B, 4), device=dev), action1=torch.zeros((B, 4), device=dev),
            key=np.repeat(key[None], B, 0), cur_step=torch.zeros((B,), dtype=torch.int32, device=dev),
            stiffness=rep(stiff), mu=rep(torch.tensor(float(self.conf.mu), device=dev)))

    def get_x_grid(self, x):  # :366-368
        if self.x_grid is None:
            self.reset_jax()
        out = self.x_grid[None].repeat(x.shape[0], 1, 1, 1)
        out[:, self.idx_i, self.idx_j] = x
        return out

    # -- the hot path ----------------------------------------------------------------------------------
    def rollout(self, state: ClothState, actions, want_lists=True):
        """lax.scan(step_jax, state, actions[T,B,8]) in one launch -> (state, state_list)."""
        T = actions.shape[0]
        prim = torch.stack([state.primitive0, state.primitive1], 1)
        k = state.stiffness.to(torch.float32)
        out = _Rollout.apply(self, state.x, state.v, prim, k, state.mu, actions, want_lists)
        xo, vo, po = out[:3]
        a_last = actions[-1].detach()                                                # bookkeeping fields, never differentiated
        both = torch.where(self._suction_col, a_last, a_last.clamp(-2, 2) * _R50)    # :168-169 for both grippers at once
        act0, act1 = both[:, :4], both[:, 4:]
        key = prng.split_first(state.key, T)                                         # :172, once per robot_step
        new = state._replace(x=xo, v=vo, primitive0=po[:, 0], primitive1=po[:, 1], action0=act0, action1=act1, key=key)
        if not want_lists:
            return new, None
        xl, vl, pl = out[3:]
        a0l = torch.cat([actions[..., :3].clamp(-2, 2) * _R50, actions[..., 3:4]], -1)
        a1l = torch.cat([actions[..., 4:7].clamp(-2, 2) * _R50, actions[..., 7:8]], -1)
        keys, kk = [], state.key
        for _ in range(T):
            kk = prng.split_first(kk, 1)
            keys.append(kk)
        keys = np.stack(keys)
        ex = lambda t: t[None].expand((T,) + tuple(t.shape))
        state_list = ClothState(x=xl, v=vl, primitive0=pl[:, :, 0], primitive1=pl[:, :, 1], action0=a0l, action1=a1l,
                                key=keys, cur_step=ex(state.cur_step), stiffness=ex(state.stiffness), mu=ex(state.mu))
        return new, state_list

    def step_jax(self, state: ClothState, action):
        """One robot_step for the batch: (state, action[B,8]) -> (state, state)   (:68-70, :163-180)."""
        new, _ = self.rollout(state, action[None], want_lists=False)
        return new, new
