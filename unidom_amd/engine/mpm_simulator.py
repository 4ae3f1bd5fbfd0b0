"""SimpleMPMSimulator -- host-side mirror of the reference's MLS-MPM simulator over the HIP kernels.

Mirrors /root/reference/DaXBench/daxbench/core/engine/mpm_simulator.py:
    MPMState                        :13-24
    SimpleMPMSimulator.__init__     :27-63   (same arguments / attributes: material, h, n_particles, key_global)
    add_box / add_box_from_points   :65-145  (lattice seeding for material != 0; liquids use uniform sampling)
    reset_jax(state)                :152-172
    step_jax(state, action)         :61-63, :413-429  one `step` = conf.steps substeps for the whole batch
All physics runs in libunidom_hip.so (csrc/mpm.hip), one launch per `step`; torch provides device memory,
streams and the autograd hook (forward kernel + adjoint kernel as one torch.autograd.Function).
The reference's mutable attributes material / h / n_particles (set during add_box / reset) become constants of
the kernel handle, which is created at reset_jax().
"""
from __future__ import annotations

import ctypes as C
from typing import List, NamedTuple

import numpy as np
import torch

from .. import _lib
from ..utils import prng
from .primitives.primitives import PrimitiveState


class MPMState(NamedTuple):  # mpm_simulator.py:13-24
    x: torch.Tensor = None
    v: torch.Tensor = None
    C: torch.Tensor = None
    F: torch.Tensor = None
    J: torch.Tensor = None
    cur_step: torch.Tensor = None
    primitives: List[PrimitiveState] = []
    key: np.ndarray = None
    friction: torch.Tensor = None
    mu: torch.Tensor = None
    lamda: torch.Tensor = None


class _Step(torch.autograd.Function):
    """forward = ud_mpm_step_fwd, backward = ud_mpm_step_bwd (include/unidom_hip.h)."""

    @staticmethod
    def forward(ctx, sim, x, v, Cm, F, J, ppos, prot, psize, friction, mu, lamda, action):
        L = _lib.lib()
        B, N, S = x.shape[0], sim.n_particles, sim.conf.steps
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        x, v, Cm, F, J, ppos, prot, psize, friction, mu, lamda, action = map(
            f32, (x, v, Cm, F, J, ppos, prot, psize, friction, mu, lamda, action))
        assert x.shape == (B, N, 3) and Cm.shape == (B, N, 3, 3) and J.shape == (B, N)
        P = sim.n_primitive
        pa = (P,) if P > 1 else ()          # primitive axis after the env axis when there are several primitives
        assert ppos.shape == (B,) + pa + (S, 3) and prot.shape == (B,) + pa + (S, 4) and psize.shape == (B,) + pa + (3,)
        assert action.shape == (B, 6 * P)
        ctx.pshape = (tuple(friction.shape), tuple(mu.shape), tuple(lamda.shape))
        friction, mu, lamda = friction.reshape(B).contiguous(), mu.reshape(B).contiguous(), lamda.reshape(B).contiguous()
        dev = x.device
        xo, vo, Co, Fo, Jo = (torch.empty_like(t) for t in (x, v, Cm, F, J))
        ppo, pro = torch.empty_like(ppos), torch.empty_like(prot)
        pvo, pwo = torch.empty_like(ppos), torch.empty_like(ppos)
        ckpt = None
        if any(ctx.needs_input_grad):
            ckpt = torch.empty((L.ud_mpm_ckpt_bytes(sim._h, C.c_int(B)) // 4,), dtype=torch.float32, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)   # every entry is written by the call (include/unidom_hip.h)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ev = sim._prof_begin("fwd")
        _lib.check(L.ud_mpm_step_fwd(
            sim._h, C.c_int(B), *[_lib.ptr(t) for t in (x, v, Cm, F, J, ppos, prot, psize, friction, mu, lamda, action)],
            *[_lib.ptr(t) for t in (xo, vo, Co, Fo, Jo, ppo, pro, pvo, pwo)], _lib.ptr(ckpt), _lib.ptr(status), stream),
            "ud_mpm_step_fwd")
        sim._prof_end(ev)
        ctx.overflow = None
        if ckpt is not None and sim.grid_ckpt_cells > 0 and sim._h_large:
            # the only flag this path raises is "grid-checkpoint pool exhausted": not an error -- that step's backward recomputes
            # the grid instead.  The flags travel to pinned host memory on a side stream, so the backward reads them without a sync.
            ctx.overflow = sim._stage_flags(status)
        else:
            sim.status_log.append(status)
        ctx.sim, ctx.B = sim, B
        if sim.keep_last_ckpt:                 # measurement aid (active_cells_per_substep): off by default, a checkpoint can be gigabytes
            sim._last_ckpt = (ckpt, B)
        ctx.save_for_backward(ckpt, psize, friction, mu, lamda, action)
        if sim.use_position_control:
            ctx.mark_non_differentiable(Jo, pro, pvo, pwo)
        else:                                  # soft contact: the rotation array carries gradient (collide_batch)
            ctx.mark_non_differentiable(Jo, pvo, pwo)
        return xo, vo, Co, Fo, Jo, ppo, pro, pvo, pwo

    @staticmethod
    def backward(ctx, gx, gv, gC, gF, gJ, gppos, gprot, gpv, gpw):
        L = _lib.lib()
        sim, B = ctx.sim, ctx.B
        ckpt, psize, friction, mu, lamda, action = ctx.saved_tensors
        if ckpt is None:
            raise _lib.UnidomError("MPM step backward without a checkpoint (forward ran under no_grad)")
        N, S, dev = sim.n_particles, sim.conf.steps, psize.device
        z = lambda t, shape: (torch.zeros(shape, device=dev) if t is None else t.to(torch.float32).contiguous())
        gx, gv, gC, gF = z(gx, (B, N, 3)), z(gv, (B, N, 3)), z(gC, (B, N, 3, 3)), z(gF, (B, N, 3, 3))
        pa = (sim.n_primitive,) if sim.n_primitive > 1 else ()
        gppos, gprot = z(gppos, (B,) + pa + (S, 3)), z(gprot, (B,) + pa + (S, 4))
        ox, ov, oC, oF, opp, opr = (torch.empty_like(t) for t in (gx, gv, gC, gF, gppos, gprot))
        ofr, omu, ola = (torch.empty((B,), device=dev) for _ in range(3))
        oa = torch.empty((B, 6 * sim.n_primitive), device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)   # every entry is written by the call (include/unidom_hip.h)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ev = sim._prof_begin("bwd")
        _lib.check(L.ud_mpm_step_bwd(
            sim._h, C.c_int(B), _lib.ptr(ckpt), *[_lib.ptr(t) for t in (psize, friction, mu, lamda, action)],
            *[_lib.ptr(t) for t in (gx, gv, gC, gF, gppos, gprot)], C.c_int((1 if sim.clip_grad else 0) | (2 if sim._overflowed(ctx.overflow) else 0)),
            *[_lib.ptr(t) for t in (ox, ov, oC, oF, opp, opr, ofr, omu, ola, oa)], _lib.ptr(status), stream), "ud_mpm_step_bwd")
        sim._prof_end(ev)
        sim.status_log.append(status)
        fs, ms, ls = ctx.pshape
        return (None, ox, ov, oC, oF, None, opp, opr, None, ofr.reshape(fs), omu.reshape(ms), ola.reshape(ls), oa)


class SimpleMPMSimulator:
    # Kernel selection of the many-workgroup path (ud_mpm_conf.tune_*: lanes, cluster, cluster_part_lanes, cluster_envs, env_groups,
    # bwd_two_launch, collide_records; include/unidom_hip.h).  Empty = the library's choice by measurement.  Copied into `self.tuning` when a simulator
    # is built and fixed for a handle when it is created (_make_handle) -- there is no per-call switch; diagnostics and the tests set it.
    default_tuning: dict = {}

    def __init__(self, conf, batch_size, use_position_control=False, device="cuda"):
        self.conf = conf
        self.tuning = dict(type(self).default_tuning)
        self.keep_last_ckpt, self._last_ckpt = False, None
        self.key_global = None
        self.batch_size = batch_size
        self.ground_friction = conf.ground_friction
        self.res = tuple(conf.res)
        self.dt = conf.dt
        self.dx = conf.dx
        self.inv_dx = conf.inv_dx
        self.p_mass = conf.p_mass
        self.p_vol = conf.p_vol
        self.gravity = conf.gravity
        self.key = getattr(conf, "key", None)
        self.n_particles = 0
        self.n_grid = conf.n_grid
        self.use_position_control = use_position_control
        self.device = torch.device(device)
        self.material = None
        self.h = None
        self.clip_grad = True            # norm_grad_state / norm_grad (:375-411)
        self.prim_friction, self.prim_softness = 0.1, 666.0   # PrimitiveState.friction / .softness (collide_batch); set by create_primitive
        self.prim_friction_each, self.prim_softness_each = [], []   # per primitive, filled by reset_jax
        self.n_primitive, self.sdf_kind = 1, "box"             # fixed at reset_jax (state.primitives, primitives.set_sdf)
        self.grid_ckpt_cells = int(getattr(conf, "grid_ckpt_cells", 0))   # include/unidom_hip.h: 0 = recompute the grid in the backward
        self.sort_particles = int(getattr(conf, "sort_particles", 0))     # include/unidom_hip.h: internal spatial order (liquids)
        self.deterministic = int(getattr(conf, "deterministic", 0))       # include/unidom_hip.h: particle-order cell sums, IEEE arithmetic (test mode)
        self.profile = None
        self.status_log = []
        self._status_acc = None          # flags folded out of status_log (device scalar), read by check_status()
        self._staged = []                # (pinned flags, event) of recent forwards on the many-workgroup path
        self._h = None
        self._h_large = False            # the handle runs the many-workgroup path (N > 128 or soft contact)
        self._flag_stream = None
        self.grid_ckpt_overflows = 0     # steps whose backward fell back to recomputing the grid

    # -- particle seeding (:65-145) ----------------------------------------------------------------------
    def add_box(self, conf, state, size, init_pos, hardness=1, z_rotation_angle=0, material=0, density=1):
        assert density >= 1
        size, init_pos = np.asarray(size, np.float32), np.asarray(init_pos, np.float32)
        c, s = np.float32(np.cos(z_rotation_angle)), np.float32(np.sin(z_rotation_angle))
        rot = np.array([[c, -s], [s, c]], np.float32)
        if material == 0:                                            # :87-91 (uniform; jax sampler: utils/prng.uniform)
            n_points = int(size.prod() * conf.n_grid ** 3 * density)
            u = prng.uniform(np.asarray(conf.key, np.uint32), n_points * 3).reshape(n_points, 3)
            x_ = (u * 2 - 1) * (np.float32(0.5) * size)
        else:                                                        # :94-109 lattice
            n_grid = int(conf.n_grid * density)
            center = np.array([0.5, 0.01, 0.5], np.float32)
            lower = (np.zeros(3, np.float32) * 2 - 1) * (np.float32(0.5) * size) + center
            upper = (np.ones(3, np.float32) * 2 - 1) * (np.float32(0.5) * size) + center
            a, b, cc = np.indices((n_grid, n_grid, n_grid))
            grid_idx = np.stack([a, b, cc], -1).astype(np.float32) * np.float32(1.0) / np.float32(n_grid)
            mask = ((grid_idx <= upper) & (grid_idx >= lower)).all(-1)
            x_ = grid_idx[mask] - center
        x_ = x_.astype(np.float32)
        x_[:, [0, 2]] = x_[:, [0, 2]] @ rot.T
        x_ = x_ + init_pos
        return self._append(state, torch.tensor(x_, device=self.device), material, hardness)

    def add_box_from_points(self, conf, state, points, hardness=1, material=0):
        return self._append(state, torch.as_tensor(points, dtype=torch.float32, device=self.device), material, hardness)

    def _append(self, state, x_, material, hardness):
        n = x_.shape[0]
        mat, hh = np.full((n,), material, np.int32), np.full((n,), hardness, np.float32)
        if state is None:
            self.material, self.h = mat, hh
        else:
            x_ = torch.cat((state.x, x_), 0)
            self.material, self.h = np.concatenate((self.material, mat)), np.concatenate((self.h, hh))
        return MPMState(x=x_, primitives=[] if state is None else list(state.primitives))

    # -- reset (:152-172): also the point where the kernel handle is built ---------------------------------
    def reset_jax(self, state: MPMState) -> MPMState:
        self.n_particles = N = state.x.shape[0]
        B, dev = self.batch_size, self.device
        conf = self.conf
        E, nu = conf.E, conf.nu
        mu_0, lambda_0 = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
        rep = lambda t: t[None].repeat((B,) + (1,) * t.dim()).contiguous()
        prims = [PrimitiveState(*[rep(torch.as_tensor(f).to(dev)) for f in p]) for p in state.primitives]
        key_global = self.key_global if self.key_global is not None else prng.PRNGKey(0)
        out = MPMState(
            x=rep(state.x.to(dev)), v=torch.zeros((B, N, 3), device=dev), C=torch.zeros((B, N, 3, 3), device=dev),
            F=rep(torch.eye(3, device=dev)[None].repeat(N, 1, 1)), J=torch.ones((B, N), device=dev),
            cur_step=torch.zeros((B,), dtype=torch.int32, device=dev), primitives=prims,
            key=prng.split(key_global, B),
            friction=torch.full((B, 1), float(conf.ground_friction), device=dev),
            mu=torch.full((B, 1), float(mu_0), device=dev), lamda=torch.full((B, 1), float(lambda_0), device=dev))
        if state.primitives:                   # constants of collide_batch (primitives.py:154-182), fixed per handle
            self.prim_friction = float(state.primitives[0].friction.reshape(-1)[0])
            self.prim_softness = float(state.primitives[0].softness.reshape(-1)[0])
            # create_primitive passes friction / softness per primitive (mpm_env.py:201-217): one pair per primitive in the handle
            self.prim_friction_each = [float(q.friction.reshape(-1)[0]) for q in state.primitives]
            self.prim_softness_each = [float(q.softness.reshape(-1)[0]) for q in state.primitives]
            if len(state.primitives) > 4 or any(s <= 0 for s in self.prim_softness_each):
                raise NotImplementedError("at most four primitives, softness > 0")
            self.n_primitive = len(state.primitives)
        from .primitives.primitives import get_sdf_kind
        self.sdf_kind = get_sdf_kind()
        self._make_handle()
        return out

    def _make_handle(self):
        if self._h is not None:
            _lib.lib().ud_mpm_destroy(self._h)
        conf = self.conf
        g = [float(v) for v in np.asarray(conf.gravity, dtype=np.float64).reshape(3)]
        cc = _lib.ud_mpm_conf(n_particles=self.n_particles, n_grid=int(conf.n_grid), res=(C.c_int * 3)(*self.res),
                              steps=int(conf.steps), dt=float(conf.dt), p_mass=float(conf.p_mass), p_vol=float(conf.p_vol),
                              gravity=(C.c_float * 3)(*g), use_position_control=int(bool(self.use_position_control)),
                              prim_friction=float(self.prim_friction), prim_softness=float(self.prim_softness),
                              n_primitive=int(self.n_primitive), sdf_kind={"box": 0, "container": 1}[self.sdf_kind],
                              grid_ckpt_cells=int(self.grid_ckpt_cells), sort_particles=int(self.sort_particles),
                              prim_friction_each=(C.c_float * 4)(*(list(self.prim_friction_each) + [0.0] * 4)[:4]),
                              prim_softness_each=(C.c_float * 4)(*(list(self.prim_softness_each) + [0.0] * 4)[:4]),
                              deterministic=int(self.deterministic), max_envs=int(self.batch_size),
                              **{"tune_" + k: int(v) for k, v in self.tuning.items()})
        mat = np.ascontiguousarray(self.material, dtype=np.int32)
        hh = np.ascontiguousarray(self.h, dtype=np.float32)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ud_mpm_create(C.byref(cc), mat.ctypes.data_as(C.c_void_p),
                                                hh.ctypes.data_as(C.c_void_p), C.byref(self._h)), "ud_mpm_create")
        self._h_large = self.n_particles > 128 or not self.use_position_control

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ud_mpm_destroy(self._h)
        except Exception:
            pass

    # -- grid-checkpoint overflow flags (include/unidom_hip.h: status[] of the many-workgroup path) -------------------
    def _stage_flags(self, status):
        if self._flag_stream is None:
            self._flag_stream = torch.cuda.Stream(self.device)
        host = torch.empty(status.shape, dtype=status.dtype, pin_memory=True)
        done = torch.cuda.Event()
        self._flag_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._flag_stream):
            host.copy_(status, non_blocking=True)
            done.record(self._flag_stream)
        status.record_stream(self._flag_stream)
        self._staged.append((host, done))
        if len(self._staged) > 64:           # the oldest are long complete: look at them now, without a sync
            for h, d in self._staged[:32]:
                if d.query() and bool((h & 6).any()):
                    self._reset_after_failure(h)
                    raise _lib.UnidomError("MPM cluster kernels: device-side failure flags in an earlier step (spill list overflow / time-out)")
            self._staged = self._staged[32:]
        return host, done

    def _overflowed(self, staged):
        """status[] of a forward on the many-workgroup path (include/unidom_hip.h): bit 0 = the grid-checkpoint pool ran out (not an
        error: that step's backward recomputes the grid); bits 1-2 = a part of the persistent cluster kernel overflowed its cell
        table / gave up waiting for its siblings (the step's outputs are invalid: raise)."""
        if staged is None:
            return False
        host, done = staged
        done.synchronize()            # long complete by the time the backward of this step runs
        if bool((host & 6).any()):
            self._reset_after_failure(host)
            raise _lib.UnidomError(f"MPM cluster kernels: device-side failure flags {sorted(set(host.tolist()))} (2 = a part's spill list "
                                   "overflowed, 4 = a part gave up waiting for its siblings); outputs of this step are invalid")
        over = bool((host & 1).any())
        self.grid_ckpt_overflows += int(over)
        return over

    def _reset_after_failure(self, flags):
        """A part that gave up waiting (bit 4) skipped the zeroing of its cells: the handle's arenas go back to their rest state
        (ud_mpm_reset, asynchronous on the current stream) before anything else runs on it."""
        if bool((flags & 4).any()):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.lib().ud_mpm_reset(self._h, stream), "ud_mpm_reset")

    def _prof_begin(self, kind):
        if self.profile is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream(self.device))
        return (kind, a, b)

    def _prof_end(self, ev):
        if ev is not None:
            ev[2].record(torch.cuda.current_stream(self.device))
            self.profile[ev[0]].append((ev[1], ev[2]))

    def active_cells_per_substep(self):
        """Mean over envs and substeps of the active grid cells the last forward with gradients enabled recorded in its checkpoint
        (ud_mpm_ckpt_cells; set `keep_last_ckpt = True` before that forward); 0.0 for handles without a grid checkpoint.  Synchronises."""
        if self._last_ckpt is None or self._last_ckpt[0] is None:
            return 0.0
        ckpt, B = self._last_ckpt
        cells = torch.empty((B,), dtype=torch.int32, device=ckpt.device)
        stream = C.c_void_p(torch.cuda.current_stream(ckpt.device).cuda_stream)
        _lib.check(_lib.lib().ud_mpm_ckpt_cells(self._h, C.c_int(B), _lib.ptr(ckpt), _lib.ptr(cells), stream), "ud_mpm_ckpt_cells")
        return float(cells.double().mean().item()) / float(self.conf.steps)

    def launch_plan(self, B):
        """ud_mpm_launch_plan: 0 = one workgroup per env; bit 0 many-workgroup path, bit 1 persistent forward, bit 2 two-launch backward"""
        return int(_lib.lib().ud_mpm_launch_plan(self._h, C.c_int(B)))

    def check_status(self):
        """Host sync: raise if any launch since the last check overflowed its LDS cell table."""
        for host, done in self._staged:      # forwards whose flags travelled on the side stream (error bits only; bit 0 is the pool)
            done.synchronize()
            if bool((host & 6).any()):
                self._staged = []
                self._reset_after_failure(host)
                raise _lib.UnidomError(f"MPM cluster kernels: device-side failure flags {sorted(set(host.tolist()))} (2 = a part's spill "
                                       "list overflowed, 4 = a part gave up waiting for its siblings)")
        self._staged = []
        if self.status_log or self._status_acc is not None:
            self._fold_status(0)
            bad = self._status_acc.item()
            self._status_acc = None
            if bad and self._h_large:      # which bit it was is gone in the fold: a reset is cheap and always right
                stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                _lib.check(_lib.lib().ud_mpm_reset(self._h, stream), "ud_mpm_reset")
            if bad:
                raise _lib.UnidomError("MPM device-side capacity exceeded (UD_ERR_OVERFLOW): LDS cell table (one-workgroup path) or "
                                       "grid-checkpoint pool (conf.grid_ckpt_cells too small) -- particle cloud too spread out")

    def _fold_status(self, keep):
        """Fold all but the newest `keep` per-launch flag tensors into one persistent device counter (no host sync), so a
        long run that never calls check_status() cannot lose a flag by list truncation."""
        old = self.status_log[:len(self.status_log) - keep] if keep else self.status_log
        if old:
            s = torch.stack([t.ne(0).sum() for t in old]).sum()
            self._status_acc = s if self._status_acc is None else self._status_acc + s
        self.status_log = self.status_log[len(self.status_log) - keep:] if keep else []

    # -- the hot path --------------------------------------------------------------------------------------
    def step_jax(self, state: MPMState, action):
        """One `step` for the batch: (state, action[B, 6*n_primitive]) -> (state, state)   (:413-429)."""
        if self._h is None:
            raise _lib.UnidomError("reset_jax() must run before step_jax() (it fixes n_particles / material / h)")
        P = self.n_primitive
        if len(state.primitives) != P:
            raise _lib.UnidomError(f"state has {len(state.primitives)} primitives, the kernel handle was built for {P}")
        if len(self.status_log) > 256:
            self._fold_status(8)
        prims = state.primitives
        if P == 1:
            pos, rot, size = prims[0].position, prims[0].rotation, prims[0].size
        else:                                  # [B, P, ...]: primitive axis after the env axis (include/unidom_hip.h)
            pos, rot, size = (torch.stack([getattr(q, k) for q in prims], 1) for k in ("position", "rotation", "size"))
        xo, vo, Co, Fo, Jo, ppo, pro, pvo, pwo = _Step.apply(
            self, state.x, state.v, state.C, state.F, state.J, pos, rot, size, state.friction,
            state.mu, state.lamda, action[:, :6 * P])
        a = action[:, :6 * P].clamp(-1, 1)
        if P == 1:
            new_prims = [prims[0]._replace(position=ppo, rotation=pro, v=pvo, w=pwo, action_buffer=a)]
        else:
            new_prims = [q._replace(position=ppo[:, i], rotation=pro[:, i], v=pvo[:, i], w=pwo[:, i], action_buffer=a[:, 6 * i:6 * i + 6])
                         for i, q in enumerate(prims)]
        new = state._replace(x=xo, v=vo, C=Co, F=Fo, J=Jo, primitives=new_prims)
        return new, new
