"""ORACLE (test infrastructure, never shipped, never measured as the product).

Literal NumPy (float64, dense n_grid^3 grid) restatement of the Taichi "PlasticineLab" MLS-MPM used by GenORM's
Torus task (BASELINE config 5), forward only.  Follows /root/reference/GenORM/policy/pbm/plb/:

  engine/mpm_simulator.py   constants :14-32, compute_F_tmp :91-94, svd :96-99, compute_von_mises :133-150,
                            p2g :166-195, grid_op :200-232, g2p :234-253, substep :256-268, step :438-449
  engine/primitive/primive_base.py   forward_kinematics :118-121, set_velocity :185-192
  engine/primitive/primitives.py     Sphere.sdf/normal/collider_v/collide :17-53, length :8-10 (+1e-14)
  engine/primitive/utils.py          length (+1e-8) :4-5, qrot :8-13, qmul :19-27, w2quat :30-41
  engine/shapes/shape_maker.py       add_box seeding with np.random.seed(0) :21-31, :49-58
  envs/torus.yml                     task constants

PARITY UNPINNED: taichi is not installed and the reference ships no recorded trajectory for this path, so
nothing ties this restatement to the reference beyond reading; ti.svd (third party) is taken as A = U sig V^T with
non-negative sig (valid for det F > 0).  tests/test_plb.py checks it against analytic known answers and the C++
restatement (oracle/csrc/plb_oracle.hpp).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class PlbConf:
    quality: float = 1.0                 # cfg.SIMULATOR.quality; 3-D halves it (:14-16)
    n_particles: int = 1000
    E: float = 5e3
    nu: float = 0.35
    yield_stress: float = 1762.2         # torus.yml
    gravity: tuple = (0.0, -0.4, 0.0)
    ground_friction: float = 0.5
    radius: tuple = (0.025, 0.025)       # two Sphere primitives
    lower_bound: tuple = (0.0, 0.0, 0.0)
    upper_bound: tuple = (1.0, 1.0, 1.0)

    @property
    def q3(self):
        return self.quality * 0.5

    @property
    def n_grid(self):
        return int(128 * self.q3)

    @property
    def dx(self):
        return 1 / self.n_grid

    @property
    def inv_dx(self):
        return float(self.n_grid)

    @property
    def dt(self):
        return 0.5e-4 / self.q3

    @property
    def substeps(self):
        return int(2e-3 // self.dt)

    @property
    def p_vol(self):
        return (self.dx * 0.5) ** 2

    @property
    def p_mass(self):
        return self.p_vol * 1


def torus_particles(n=1000, width=(0.028, 0.5, 0.028), init_pos=(0.5, 0.3, 0.5)):
    """shape_maker.py:21-31,49-58: np.random.seed(0); (random((n,3))*2-1)*0.5*width + pos (legacy NumPy RNG)."""
    state = np.random.get_state()
    np.random.seed(0)
    p = (np.random.random((n, 3)) * 2 - 1) * (0.5 * np.array(width)) + np.array(init_pos)
    np.random.set_state(state)
    return p


def _qrot(rot, v):
    qvec = rot[1:4]
    uv = np.cross(qvec, v)
    uuv = np.cross(qvec, uv)
    return v + 2 * (rot[0] * uv + uuv)


class PlbTwin:
    def __init__(self, conf: PlbConf):
        self.c = conf

    def substep(self, x, v, C, F, pos_f, pos_f1, softness, E, nu, yield_stress):
        """One substep for one env. pos_f / pos_f1: [n_prim,3] primitive positions at f and f+1 (rotations identity)."""
        c = self.c
        n, dt, dx, inv_dx = c.n_grid, c.dt, c.dx, c.inv_dx
        N = x.shape[0]
        grid_v = np.zeros((n, n, n, 3))
        grid_m = np.zeros((n, n, n))
        I3 = np.eye(3)
        F_tmp = (I3[None] + dt * C) @ F                                              # :91-94
        U, sig, Vh = np.linalg.svd(F_tmp)                                            # :96-99 (V = Vh^T)
        mu, lam = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))             # :168
        base = (x * inv_dx - 0.5).astype(np.int64)                                   # cast(int): truncation
        fx = x * inv_dx - base
        w = [0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1) ** 2, 0.5 * (fx - 0.5) ** 2]
        # compute_von_mises :133-150
        sg = np.maximum(sig, 0.05)
        eps = np.log(sg)
        eps_hat = eps - eps.sum(-1, keepdims=True) / 3
        eps_hat_norm = np.sqrt((eps_hat * eps_hat).sum(-1) + 1e-8)
        delta_gamma = eps_hat_norm - yield_stress / (2 * mu)
        yields = delta_gamma > 0
        eps_y = eps - (delta_gamma / eps_hat_norm)[:, None] * eps_hat
        F_y = (U * np.exp(eps_y)[:, None, :]) @ Vh
        new_F = np.where(yields[:, None, None], F_y, F_tmp)
        J = np.linalg.det(new_F)
        r = U @ Vh
        stress = 2 * mu * (new_F - r) @ new_F.transpose(0, 2, 1) + I3[None] * (lam * J * (J - 1))[:, None, None]
        stress = (-dt * c.p_vol * 4 * inv_dx * inv_dx) * stress
        affine = stress + c.p_mass * C
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    off = np.array([i, j, k])
                    dpos = (off - fx) * dx
                    weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                    idx = base + off
                    np.add.at(grid_v, (idx[:, 0], idx[:, 1], idx[:, 2]),
                              weight[:, None] * (c.p_mass * v + (affine @ dpos[..., None])[..., 0]))
                    np.add.at(grid_m, (idx[:, 0], idx[:, 1], idx[:, 2]), weight * c.p_mass)
        # grid_op :200-232
        out = np.zeros_like(grid_v)
        g = np.array(c.gravity)
        occ = np.argwhere(grid_m > 1e-12)
        for (a, b_, d_) in occ:
            Iv = np.array([a, b_, d_])
            vo = grid_v[a, b_, d_] / grid_m[a, b_, d_]
            vo = vo + dt * g * 30
            gp = Iv * dx
            for pi in range(pos_f.shape[0]):                                        # Sphere.collide :46-53
                dist = np.sqrt(((gp - pos_f[pi]) ** 2).sum() + 1e-14) - c.radius[pi]
                soft = softness[pi]
                influence = min(np.exp(-dist * soft), 1)
                if ((soft > 0 and influence > 0.1) or dist <= 0.001) and soft > 0:
                    vo = (pos_f1[pi] - pos_f[pi]) / dt                               # collider_v with identity rotations
            for d in range(3):
                if Iv[d] < 3 and vo[d] < 0:
                    if d != 1 or c.ground_friction == 0:
                        vo[d] = 0
                    elif c.ground_friction < 10:
                        normal = np.zeros(3)
                        normal[d] = 1.0
                        lin = vo.dot(normal) + 1e-30
                        vit = vo - lin * normal - Iv * 1e-30
                        lit = np.sqrt(vit.dot(vit) + 1e-8)
                        vo = max(1.0 + c.ground_friction * lin / lit, 0.0) * (vit + Iv * 1e-30)
                        vo[1] = 0
                    else:
                        vo = np.zeros(3)
                if Iv[d] > n - 3 and vo[d] > 0:
                    vo[d] = 0
            out[a, b_, d_] = vo
        # g2p :234-253
        new_v = np.zeros((N, 3))
        new_C = np.zeros((N, 3, 3))
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    off = np.array([i, j, k])
                    dpos = off - fx
                    weight = w[i][:, 0] * w[j][:, 1] * w[k][:, 2]
                    idx = base + off
                    g_v = out[idx[:, 0], idx[:, 1], idx[:, 2]]
                    new_v += weight[:, None] * g_v
                    new_C += 4 * inv_dx * weight[:, None, None] * (g_v[:, :, None] * dpos[:, None, :])
        new_x = np.maximum(np.minimum(x + dt * new_v, 1.0 - 3 * dx), 0.0)
        return new_x, new_v, new_C, new_F

    def step(self, x, v, C, F, prim_pos, action, softness, E=None, nu=None, yield_stress=None):
        """TaichiEnv.step in copy mode (:438-449): set_action (clip +-1, v = a*scale/substeps for the primitives that
        have an action; Torus: primitive 0 has 3 dims, primitive 1 none), `substeps` substeps, copy frame cur -> 0."""
        c = self.c
        E = c.E if E is None else E
        nu = c.nu if nu is None else nu
        ys = c.yield_stress if yield_stress is None else yield_stress
        S = c.substeps
        a = np.clip(np.asarray(action, np.float64).reshape(-1), -1, 1)
        pv = np.zeros_like(prim_pos)
        pv[0] = a[:3] * 1.0 / S
        lo, hi = np.array(c.lower_bound), np.array(c.upper_bound)
        pos = prim_pos.copy()
        for _ in range(S):
            pos1 = np.maximum(np.minimum(pos + pv, hi), lo)                          # forward_kinematics :118-121
            x, v, C, F = self.substep(x, v, C, F, pos, pos1, softness, E, nu, ys)
            pos = pos1
        return x, v, C, F, pos
