"""PlbSimulator -- host mirror of GenORM's Taichi MPMSimulator + Primitives + Loss for the Torus task (float64).

Mirrors /root/reference/GenORM/policy/pbm/plb/engine/mpm_simulator.py (constants :14-32, step :438-449 in copy
mode, substep_grad :271-289), the two Sphere primitives of envs/torus.yml and engine/losses/loss.py:112-243; the
physics runs in libunidom_hip.so (csrc/plb.hip forward, csrc/plb_adj.hip adjoint + losses).  The reference holds one
env per process in Taichi fields and differentiates with ti.Tape; here B independent envs are batched in one call and
`step` / `compute_loss` are torch.autograd Functions over the forward / adjoint kernel pairs, so
`loss.backward()` plays the role of the tape: gradients reach the particle state, the action, the primitive position and
the parameter leaves E, nu, yield_stress (PlasticineLab/sim2sim/plb/engine/mpm_simulator.py:27-29,485-498,
get_parameter_grad) and the ground friction (optimize_ground_friction, :57-58; `PlbSimulator.ground_friction_grad`).
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import numpy as np
import torch

from .. import _lib


class PlbState(NamedTuple):
    x: torch.Tensor          # [B,N,3] float64
    v: torch.Tensor          # [B,N,3]
    C: torch.Tensor          # [B,N,3,3]
    F: torch.Tensor          # [B,N,3,3]
    prim_pos: torch.Tensor   # [B,n_prim,3]
    softness: torch.Tensor   # [B,n_prim]
    E: torch.Tensor          # [B]
    nu: torch.Tensor         # [B]
    yield_stress: torch.Tensor  # [B]


class PlbConf:
    """cfg.SIMULATOR of plb/config/default_config.py:10-24 overridden by envs/torus.yml."""
    dim = 3
    quality = 1.0
    n_particles = 1000
    E = 5e3
    nu = 0.35
    yield_stress = 1762.2
    gravity = (0.0, -0.4, 0.0)
    ground_friction = 0.5
    dtype = "float64"
    # SHAPES / PRIMITIVES of torus.yml
    box_width = (0.028, 0.5, 0.028)
    box_init_pos = (0.5, 0.3, 0.5)
    prim_radius = (0.025, 0.025)
    prim_init_pos = ((0.475, 0.05, 0.5), (0.5, 0.55, 0.5))
    lower_bound = (0.0, 0.0, 0.0)
    upper_bound = (1.0, 1.0, 1.0)


def _p(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _c(t):
    return t.detach().to(torch.float64).contiguous()


class _PlbStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sim, x, v, Cm, F, pp, so, action, E, nu, ys):
        L = _lib.lib()
        B = x.shape[0]
        x, v, Cm, F, pp, so, action, E, nu, ys = map(_c, (x, v, Cm, F, pp, so, action, E, nu, ys))
        xo, vo, Co, Fo, po = (torch.empty_like(t) for t in (x, v, Cm, F, pp))
        ckpt = None
        if any(ctx.needs_input_grad):
            ckpt = torch.empty((L.ud_plb_ckpt_bytes(sim._h, C.c_int(B)) // 8,), dtype=torch.float64, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        ev = sim._prof_begin("fwd")
        _lib.check(L.ud_plb_step_fwd(sim._h, C.c_int(B), _p(x), _p(v), _p(Cm), _p(F), _p(pp), _p(so), _p(action), _p(E), _p(nu),
                                     _p(ys), _p(xo), _p(vo), _p(Co), _p(Fo), _p(po), _p(ckpt), stream), "ud_plb_step_fwd")
        sim._prof_end(ev)
        ctx.sim, ctx.B = sim, B
        ctx.save_for_backward(ckpt, so, action, E, nu, ys)
        return xo, vo, Co, Fo, po

    @staticmethod
    def backward(ctx, gx, gv, gC, gF, gpp):
        L = _lib.lib()
        sim, B = ctx.sim, ctx.B
        ckpt, so, action, E, nu, ys = ctx.saved_tensors
        if ckpt is None:
            raise _lib.UnidomError("PLB step backward without a checkpoint (forward ran under no_grad)")
        N, P, dev = sim.n_particles, sim.n_primitive, so.device
        g = lambda t: None if t is None else _c(t)
        gx, gv, gC, gF, gpp = map(g, (gx, gv, gC, gF, gpp))
        mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        ox, ov, oC, oF, op, oa = mk(B, N, 3), mk(B, N, 3), mk(B, N, 3, 3), mk(B, N, 3, 3), mk(B, P, 3), mk(B, 3)
        oE, onu, oys, ofr = mk(B), mk(B), mk(B), mk(B)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ev = sim._prof_begin("bwd")
        _lib.check(L.ud_plb_step_bwd(sim._h, C.c_int(B), _p(ckpt), _p(so), _p(action), _p(E), _p(nu), _p(ys), _p(gx), _p(gv), _p(gC),
                                     _p(gF), _p(gpp), _p(ox), _p(ov), _p(oC), _p(oF), _p(op), _p(oa), _p(oE), _p(onu), _p(oys), _p(ofr),
                                     stream), "ud_plb_step_bwd")
        sim._prof_end(ev)
        sim.ground_friction_grad = ofr if sim.ground_friction_grad is None else sim.ground_friction_grad + ofr
        return None, ox, ov, oC, oF, op, None, oa, oE, onu, oys


class _PlbLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sim, x, pp, td, ts, wt, soft):
        L = _lib.lib()
        B = x.shape[0]
        x, pp = _c(x), _c(pp)
        loss = torch.empty((B,), dtype=torch.float64, device=x.device)
        parts = torch.empty((B, 3), dtype=torch.float64, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(L.ud_plb_loss_fwd(sim._h, C.c_int(B), _p(x), _p(pp), _p(td), _p(ts), _p(wt), C.c_int(int(soft)), _p(loss), _p(parts),
                                     stream), "ud_plb_loss_fwd")
        ctx.sim, ctx.B, ctx.soft = sim, B, soft
        ctx.save_for_backward(x, pp, td, ts, wt)
        ctx.mark_non_differentiable(parts)
        return loss, parts

    @staticmethod
    def backward(ctx, gl, _gparts):
        L = _lib.lib()
        x, pp, td, ts, wt = ctx.saved_tensors
        gl = _c(gl)
        gx, gpp = torch.empty_like(x), torch.empty_like(pp)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(L.ud_plb_loss_bwd(ctx.sim._h, C.c_int(ctx.B), _p(x), _p(pp), _p(td), _p(ts), _p(wt), C.c_int(int(ctx.soft)), _p(gl),
                                     _p(gx), _p(gpp), stream), "ud_plb_loss_bwd")
        return None, gx, gpp, None, None, None, None


class PlbSimulator:
    def __init__(self, cfg=None, batch_size=1, device="cuda"):
        cfg = PlbConf() if cfg is None else cfg
        assert cfg.dtype == "float64"                               # mpm_simulator.py:8
        self.cfg, self.batch_size, self.device = cfg, batch_size, torch.device(device)
        quality = cfg.quality * 0.5 if cfg.dim == 3 else cfg.quality   # :14-16
        self.n_particles = cfg.n_particles
        self.n_grid = int(128 * quality)
        self.dx, self.inv_dx = 1 / self.n_grid, float(self.n_grid)
        self.dt = 0.5e-4 / quality
        # :32 (float floor-division); `cfg.substeps` overrides it (ud_plb_conf.substeps is a handle constant): short horizons for tests
        self.substeps = int(getattr(cfg, "substeps", 0)) or int(2e-3 // self.dt)
        self.p_vol = (self.dx * 0.5) ** 2
        self.p_mass = self.p_vol * 1
        self.n_primitive = len(cfg.prim_radius)
        # include/unidom_hip.h: the checkpoint also keeps the touched grid cells and the adjoint restores them instead of running p2g
        # again (27 = every cell a substep can touch: never falls back; 36 B x 27 n_particles per env and substep of checkpoint)
        self.grid_ckpt_cells = int(getattr(cfg, "grid_ckpt_cells", 27))
        # kernel selection, fixed at create (include/unidom_hip.h): path 0 = the library's choice (one persistent launch per step call
        # where it fits), 1 = multi-kernel, 2 = persistent; lanes / sort_every: diagnostics of the multi-kernel path / the spatial order
        self.path, self.lanes, self.sort_every = (int(getattr(cfg, k, 0)) for k in ("path", "lanes", "sort_every"))
        cc = _lib.ud_plb_conf(
            n_particles=self.n_particles, n_grid=self.n_grid, substeps=self.substeps, dt=self.dt,
            gravity=(C.c_double * 3)(*cfg.gravity), ground_friction=float(cfg.ground_friction), n_primitives=self.n_primitive,
            radius=(C.c_double * 2)(*(list(cfg.prim_radius) + [0.0])[:2]),
            lower_bound=(C.c_double * 3)(*cfg.lower_bound), upper_bound=(C.c_double * 3)(*cfg.upper_bound),
            grid_ckpt_cells=int(self.grid_ckpt_cells), max_envs=int(batch_size), path=self.path, lanes=self.lanes, sort_every=self.sort_every)
        self._h = C.c_void_p()
        self.profile = None    # {"fwd": [], "bwd": []}: HIP-event pairs around every step call, on its stream (bench.py)
        self.ground_friction_grad = None   # [B], accumulated by backward() (optimize_ground_friction.grad); reset it by hand
        _lib.check(_lib.lib().ud_plb_create(C.byref(cc), C.byref(self._h)), "ud_plb_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ud_plb_destroy(self._h)
        except Exception:
            pass

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

    def launch_plan(self, B=None):
        """ud_plb_launch_plan: 1 = multi-kernel path, 2 = one persistent launch per step call and direction"""
        return int(_lib.lib().ud_plb_launch_plan(self._h, C.c_int(self.batch_size if B is None else B)))

    def check_status(self):
        """Synchronises the current stream; raises if a workgroup of the persistent kernels gave up waiting (outputs NaN); the handle is usable
        again afterwards (ud_plb_poll_timeouts resets its exchange arena)."""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        n = int(_lib.lib().ud_plb_poll_timeouts(self._h, stream))
        if n != 0:
            raise _lib.UnidomError(_lib.lib().ud_last_error().decode())

    def reset(self) -> PlbState:
        """Shapes.add_box with np.random.seed(0) (shape_maker.py:21-31,49-58) + primitive init (torus.yml)."""
        cfg, B, dev = self.cfg, self.batch_size, self.device
        st = np.random.get_state()
        np.random.seed(0)
        p = (np.random.random((cfg.n_particles, 3)) * 2 - 1) * (0.5 * np.array(cfg.box_width)) + np.array(cfg.box_init_pos)
        np.random.set_state(st)
        f64 = lambda a: torch.tensor(np.asarray(a, np.float64), device=dev)
        rep = lambda t: t[None].repeat((B,) + (1,) * t.dim()).contiguous()
        N = cfg.n_particles
        return PlbState(x=rep(f64(p)), v=torch.zeros((B, N, 3), dtype=torch.float64, device=dev),
                        C=torch.zeros((B, N, 3, 3), dtype=torch.float64, device=dev),
                        F=rep(torch.eye(3, dtype=torch.float64, device=dev)[None].repeat(N, 1, 1)),
                        prim_pos=rep(f64(cfg.prim_init_pos)),
                        softness=torch.full((B, self.n_primitive), 666.0, dtype=torch.float64, device=dev),   # set_softness()
                        E=torch.full((B,), float(cfg.E), dtype=torch.float64, device=dev),
                        nu=torch.full((B,), float(cfg.nu), dtype=torch.float64, device=dev),
                        yield_stress=torch.full((B,), float(cfg.yield_stress), dtype=torch.float64, device=dev))

    def step(self, state: PlbState, action) -> PlbState:
        """TaichiEnv.step(action) in copy mode: one call = `substeps` substeps for every env.  Differentiable: when any of
        the state tensors / the action / E / nu / yield_stress requires grad, the forward keeps a checkpoint and backward()
        runs the adjoint kernels (substep_grad)."""
        B = state.x.shape[0]
        action = torch.as_tensor(action, dtype=torch.float64, device=self.device).reshape(B, 3)
        xo, vo, Co, Fo, po = _PlbStep.apply(self, state.x, state.v, state.C, state.F, state.prim_pos, state.softness, action,
                                            state.E, state.nu, state.yield_stress)
        return state._replace(x=xo, v=vo, C=Co, F=Fo, prim_pos=po)

    def compute_loss(self, state: PlbState, target_density, target_sdf, weights=(1.0, 1.0, 1.0), soft_contact=True):
        """Loss.compute_loss_kernel (engine/losses/loss.py:190-214): (loss [B], parts [B,3] = contact, density, sdf).
        target_density / target_sdf: [n_grid, n_grid, n_grid] float64; weights = (contact, density, sdf) as in :158-162."""
        td = torch.as_tensor(target_density, dtype=torch.float64, device=self.device).reshape(-1).contiguous()
        ts = torch.as_tensor(target_sdf, dtype=torch.float64, device=self.device).reshape(-1).contiguous()
        assert td.numel() == self.n_grid ** 3 == ts.numel()
        wt = torch.as_tensor(weights, dtype=torch.float64, device=self.device).reshape(3).contiguous()
        return _PlbLoss.apply(self, state.x, state.prim_pos, td, ts, wt, bool(soft_contact))

    @staticmethod
    def set_softness1(state: PlbState, softness) -> PlbState:   # primitives.py: Primitives.set_softness1
        so = state.softness.clone()
        so[:, 0] = softness
        return state._replace(softness=so)
