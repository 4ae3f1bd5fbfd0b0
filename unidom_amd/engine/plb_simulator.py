"""PlbSimulator -- host mirror of GenORM's Taichi MPMSimulator + Primitives for the Torus task (float64, forward).

Mirrors /root/reference/GenORM/policy/pbm/plb/engine/mpm_simulator.py (constants :14-32, step :438-449 in copy
mode) and the two Sphere primitives of envs/torus.yml; the physics runs in libunidom_hip.so (csrc/plb.hip).
The reference holds one env per process in Taichi fields; here B independent envs are batched in one call.
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
        self.substeps = int(2e-3 // self.dt)
        self.p_vol = (self.dx * 0.5) ** 2
        self.p_mass = self.p_vol * 1
        self.n_primitive = len(cfg.prim_radius)
        cc = _lib.ud_plb_conf(
            n_particles=self.n_particles, n_grid=self.n_grid, substeps=self.substeps, dt=self.dt,
            gravity=(C.c_double * 3)(*cfg.gravity), ground_friction=float(cfg.ground_friction), n_primitives=self.n_primitive,
            radius=(C.c_double * 2)(*(list(cfg.prim_radius) + [0.0])[:2]),
            lower_bound=(C.c_double * 3)(*cfg.lower_bound), upper_bound=(C.c_double * 3)(*cfg.upper_bound))
        self._h = C.c_void_p()
        _lib.check(_lib.lib().ud_plb_create(C.byref(cc), C.byref(self._h)), "ud_plb_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ud_plb_destroy(self._h)
        except Exception:
            pass

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

    @torch.no_grad()
    def step(self, state: PlbState, action) -> PlbState:
        """TaichiEnv.step(action) in copy mode: one call = `substeps` substeps for every env."""
        B = state.x.shape[0]
        c = lambda t: t.to(torch.float64).contiguous()
        x, v, Cm, F, pp, so, E, nu, ys = map(c, (state.x, state.v, state.C, state.F, state.prim_pos, state.softness, state.E,
                                                 state.nu, state.yield_stress))
        action = c(torch.as_tensor(action, dtype=torch.float64, device=self.device).reshape(B, 3))
        xo, vo, Co, Fo, po = (torch.empty_like(t) for t in (x, v, Cm, F, pp))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().ud_plb_step_fwd(self._h, C.c_int(B), p(x), p(v), p(Cm), p(F), p(pp), p(so), p(action), p(E), p(nu),
                                              p(ys), p(xo), p(vo), p(Co), p(Fo), p(po), stream), "ud_plb_step_fwd")
        return state._replace(x=xo, v=vo, C=Co, F=Fo, prim_pos=po)

    @staticmethod
    def set_softness1(state: PlbState, softness) -> PlbState:   # primitives.py: Primitives.set_softness1
        so = state.softness.clone()
        so[:, 0] = softness
        return state._replace(softness=so)
