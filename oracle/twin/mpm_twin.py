"""ORACLE (test infrastructure, never shipped, never measured as the product).

Line-by-line CPU restatement of the reference's MLS-MPM simulator in torch (dense `res` grid exactly like the
source; torch.autograd stands in for jax.grad).  Follows, under /root/reference/DaXBench/daxbench/core/engine/:

  svd / _svd_bwd              <- svd_safe_batch.py:19-51, :65-102
  p2g_micro / g2p_micro       <- mpm_simulator.py:178-194, :196-221
  substep                     <- mpm_simulator.py:223-330
  norm_grad(_state)           <- mpm_simulator.py:375-411
  step / copy_frame           <- mpm_simulator.py:413-429, :365-373
  forward_kinematics, set_action/set_velocity, position_control_batch, collide_batch, sdf_batch, qmul, w2quat,
  qrot_batch, inv_trans_batch, length, normal_batch, collider_v_batch
                              <- primitives/primitives.py:68-239
  box _sdf_batch              <- primitives/box.py:6-18
  pre_step / post_step / get_primitive_actions (whip_rope) / reward
                              <- envs/basic/mpm_env.py:90-125, envs/whip_rope_env.py:108-115

Pinned by reference data: tests/test_oracle_mpm.py replays expert_demo/whip_rope/demo_0.pkl (69 recorded
transitions = 3450 substeps) with the legacy parameter overrides of SURVEY.md F3.
Third-party semantics assumed: JAX scatter drops / gather clamps out-of-bounds
indices (Q5), negative indices wrap (Q9), jnp.linalg.svd returns (U, S, Vh) with S descending.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, NamedTuple

import numpy as np
import torch


@dataclass
class MPMConf:                         # whip_rope_env.py:27-73
    n_grid: int = 64
    res: tuple = (32, 32, 32)
    dt: float = 1e-4
    steps: int = 70
    E: float = 100
    nu: float = 0.1
    ground_friction: float = 0.1
    gravity: tuple = (0, -9.8, 0)
    n_primitive: int = 1
    p_rho: float = 1

    @property
    def dx(self):
        return 1 / self.n_grid

    @property
    def inv_dx(self):
        return float(self.n_grid)

    @property
    def p_vol(self):
        return (self.dx * 0.5) ** 2

    @property
    def p_mass(self):
        return self.p_vol * self.p_rho


class Prim(NamedTuple):                # primitives.py:9-23 (fields the dynamics touch)
    size: torch.Tensor                 # [3]
    position: torch.Tensor             # [steps,3]
    rotation: torch.Tensor             # [steps,4]
    v: torch.Tensor                    # [steps,3]
    w: torch.Tensor                    # [steps,3]
    friction: torch.Tensor             # []
    softness: torch.Tensor             # []
    action_buffer: torch.Tensor        # [6]
    action_scale: torch.Tensor         # [6]


class MPMState(NamedTuple):            # mpm_simulator.py:13-24
    x: torch.Tensor
    v: torch.Tensor
    C: torch.Tensor
    F: torch.Tensor
    J: torch.Tensor
    primitives: List[Prim]
    friction: torch.Tensor             # [1]
    mu: torch.Tensor                   # [1]
    lamda: torch.Tensor                # [1]


def _clip(x, lo, hi):
    return torch.minimum(torch.maximum(x, torch.as_tensor(lo, dtype=x.dtype)), torch.as_tensor(hi, dtype=x.dtype))


# ---- svd_safe_batch.py ------------------------------------------------------------------------------
class _SVD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, eps):
        U, S, Vh = torch.linalg.svd(A, full_matrices=False)
        ctx.save_for_backward(U, S, Vh)
        ctx.eps = eps
        return U, S, Vh

    @staticmethod
    def backward(ctx, dU, dS, dVh):    # :65-102 (real case: conj = identity)
        U, S, Vh = ctx.saved_tensors
        eps = ctx.eps
        safe = lambda x: x / (x ** 2 + eps)
        Ut = U.transpose(1, 2)
        Vt = Vh                         # Cc(Vh)
        Vt_dV = Vt @ dVh.transpose(1, 2)
        S2 = S ** 2
        S_inv = safe(S)
        I = torch.eye(3, dtype=U.dtype)[None].repeat(S.shape[0], 1, 1)
        Fm = safe(S2[:, None, :] - S2[..., None])
        Fm = Fm - I * Fm
        J = Fm * (Ut @ dU)
        K = Fm * Vt_dV
        L = I * Vt_dV
        Pu = I - U @ Ut
        Pv = I - Vh.transpose(1, 2) @ Vt
        S_, dS_, Si_ = S[:, None, :], dS[:, None, :], S_inv[:, None, :]
        H = lambda a: a.transpose(1, 2)
        dA = (U * dS_) @ Vt + U @ ((J + H(J)) * S_) @ Vt + (U * S_) @ (K + H(K)) @ Vt \
            + .5 * ((U * Si_) @ (L - H(L)) @ Vt) + Pu @ (dU * Si_) @ Vt + (U * Si_) @ dVh @ Pv
        return dA, None


def svd(A, eps=1e-12):
    return _SVD.apply(A, eps)


# ---- primitives.py ------------------------------------------------------------------------------------
def length(x):                                       # :68-70  (Q11: +1e-12 inside the sqrt)
    return torch.sqrt((x * x).sum(-1) + 1e-12)


def qmul(q, r):                                      # :73-81
    t = torch.outer(r, q)
    w = t[0, 0] - t[1, 1] - t[2, 2] - t[3, 3]
    x = t[0, 1] + t[1, 0] - t[2, 3] + t[3, 2]
    y = t[0, 2] + t[1, 3] + t[2, 0] - t[3, 1]
    z = t[0, 3] - t[1, 2] + t[2, 1] + t[3, 0]
    out = torch.stack([w, x, y, z])
    return out / _clip(torch.sqrt(out.dot(out)), 1e-12, math.inf)


def w2quat(axis_angle):                              # :84-92
    w = torch.linalg.norm(axis_angle) + 1e-12
    v = (axis_angle / w) * torch.sin(w / 2)
    return torch.cat([torch.cos(w / 2)[None], v[:3]])


def qrot_batch(rot, v):                              # :95-102
    qvec = rot[1:4].expand_as(v)
    uv = torch.linalg.cross(qvec, v, dim=-1)
    uuv = torch.linalg.cross(qvec, uv, dim=-1)
    return v + 2 * (rot[0] * uv + uuv)


def inv_trans_batch(pos, position, rotation):        # :105-109
    inv_quat = torch.stack([rotation[0], -rotation[1], -rotation[2], -rotation[3]])
    inv_quat = inv_quat / (torch.linalg.norm(inv_quat) + 1e-12)
    return qrot_batch(inv_quat, pos - position)


def box_sdf(size, grid_pos):                         # box.py:6-18
    q = torch.abs(grid_pos) - size.reshape(3)
    q = _clip(q, 0., math.inf)
    out = length(q)
    tmp = torch.where(q[..., 1] > q[..., 2], q[..., 1], q[..., 2])
    tmp = torch.where(q[..., 0] > tmp, q[..., 0], tmp)
    tmp = _clip(tmp, -math.inf, 0.)
    return out + tmp


def container_sdf(size, p):                          # container.py:8-16 (cut hollow sphere, size = (r, h, t))
    r, h, t = size[0], size[1], size[2]
    w = torch.sqrt(r * r - h * h)
    q = torch.stack([length(torch.stack([p[..., 0], p[..., 2]], -1)), p[..., 1]], -1)
    mask = h * q[..., 0] < w * q[..., 1]
    val1 = length(q - torch.stack([w, h])) - t
    val2 = torch.abs(length(q) - r) - t
    return torch.where(mask, val1, val2)


_sdf_batch = box_sdf                                 # primitives.py:5-6, 26-28: one process-global SDF


def set_sdf(fn):
    global _sdf_batch
    _sdf_batch = fn


def _gather_clamp(arr, idx):
    """JAX gather out-of-bounds = clamp (Q5)."""
    return arr[min(max(idx, 0), arr.shape[0] - 1)]


def _set_drop(arr, idx, val):
    """JAX scatter out-of-bounds = drop (Q5)."""
    if 0 <= idx < arr.shape[0]:
        arr = arr.clone()
        arr[idx] = val
    return arr


def sdf_batch(f, grid_pos, p: Prim):                 # :112-114
    gp = inv_trans_batch(grid_pos, _gather_clamp(p.position, f), _gather_clamp(p.rotation, f))
    return _sdf_batch(p.size, gp)


def normal_batch(f, grid_pos, p: Prim):              # :117-141
    gp = inv_trans_batch(grid_pos, _gather_clamp(p.position, f), _gather_clamp(p.rotation, f))
    d = 1.e-6
    comps = []
    for a in range(3):
        e = torch.zeros(3, dtype=gp.dtype)
        e[a] = d
        comps.append((0.5 / d) * (_sdf_batch(p.size, gp + e) - _sdf_batch(p.size, gp - e)))
    n = torch.stack(comps, -1)
    n = n / length(n)[..., None]
    return qrot_batch(_gather_clamp(p.rotation, f), n)


def collider_v_batch(f, grid_pos, dt, p: Prim):      # :144-151
    rot_f = _gather_clamp(p.rotation, f)
    inv_quat = torch.stack([rot_f[0], -rot_f[1], -rot_f[2], -rot_f[3]])
    inv_quat = inv_quat / (torch.linalg.norm(inv_quat) + 1e-12)
    relative_pos = qrot_batch(inv_quat, grid_pos - _gather_clamp(p.position, f))
    new_pos = qrot_batch(_gather_clamp(p.rotation, f + 1), relative_pos) + _gather_clamp(p.position, f + 1)
    return (new_pos - grid_pos) / dt


def collide_batch(f, grid_pos, v_out, dt, p: Prim):  # :154-182
    dist = sdf_batch(f, grid_pos, p)
    influence = _clip(torch.exp(-dist * p.softness), -math.inf, 1)[..., None]
    D = normal_batch(f, grid_pos, p)
    cv = collider_v_batch(f, grid_pos, dt, p)
    input_v = v_out - cv
    normal_component = (input_v * D).sum(-1, keepdim=True)
    grid_v_t = input_v - _clip(normal_component, -math.inf, 0.) * D
    grid_v_t_norm = length(grid_v_t)[..., None]
    grid_v_t_friction = grid_v_t / grid_v_t_norm * _clip(grid_v_t_norm + normal_component * p.friction, 1e-12, math.inf)
    grid_v_t_dot = (grid_v_t * grid_v_t).sum(-1, keepdim=True)
    flag = ((normal_component < 0).to(torch.int32) * (torch.sqrt(grid_v_t_dot) > 1e-12).to(torch.int32)).detach() * 1.0
    flag = flag.to(v_out.dtype)
    grid_v_t = grid_v_t_friction * flag + grid_v_t * (1 - flag)
    return cv + input_v * (1 - influence) + grid_v_t * influence


def forward_kinematics(f, p: Prim):                  # :185-194
    position = _set_drop(p.position, f + 1, _gather_clamp(p.position, f) + _gather_clamp(p.v, f))
    position = _clip(position, -2, 2)
    rotation = _set_drop(p.rotation, f + 1, qmul(w2quat(_gather_clamp(p.w, f)), _gather_clamp(p.rotation, f)))
    return p._replace(position=position, rotation=rotation)


def set_action(n_substeps, action, p: Prim):         # :212-229
    p = p._replace(action_buffer=action)
    vv = (p.action_buffer[:3] * p.action_scale[:3] / n_substeps)[None].expand(n_substeps, 3)
    ww = (p.action_buffer[3:] * p.action_scale[3:] / n_substeps)[None].expand(n_substeps, 3)
    v = torch.cat([vv, p.v[n_substeps:]], 0)
    w = torch.cat([ww, p.w[n_substeps:]], 0)
    return p._replace(v=v, w=w)


def position_control_batch(f, grid_pos, v_out, dt, p: Prim):   # :232-239
    dist = sdf_batch(f, grid_pos, p)
    control_mask = dist < p.size[0] * 1.5
    return torch.where(control_mask[..., None], _gather_clamp(p.v, f).reshape(1, 1, 1, 3) / dt, v_out)


# ---- mpm_simulator.py ---------------------------------------------------------------------------------
class _NormGradTree(torch.autograd.Function):
    """norm_grad / norm_grad_state bwd (:389-394, :403-408): nan_to_num, then global-norm clip to 1."""
    @staticmethod
    def forward(ctx, *xs):
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        gs = [torch.nan_to_num(g + 0.0) for g in gs]
        g_norm = torch.sqrt(sum((g * g).sum() for g in gs))
        if bool(g_norm < 1.0):
            return tuple(gs)
        return tuple(g / g_norm for g in gs)


class MPMTwin:
    def __init__(self, conf: MPMConf, n_particles, material=1, hardness=1.0, use_position_control=True,
                 dtype=torch.float32, clip_grads=True):
        self.conf, self.dtype = conf, dtype
        self.n_particles = n_particles
        self.material = torch.full((n_particles,), material)
        self.h = torch.full((n_particles,), hardness, dtype=dtype)
        self.use_position_control = use_position_control
        self.clip_grads = clip_grads
        a, b, c = np.indices((3, 3, 3))
        self.idx = torch.as_tensor(np.stack([a, b, c], -1).reshape(-1, 3))
        a, b, c = np.indices(conf.res)
        self.grid_idx = torch.as_tensor(np.stack([a, b, c], -1).reshape(-1, 3))
        self.grid_idx_3d = self.grid_idx.reshape(tuple(conf.res) + (3,))
        self.gravity = torch.as_tensor(conf.gravity, dtype=dtype)

    def _wrap_drop(self, pos):
        """scatter index rule: negative wraps (Q9), out-of-bounds dropped (Q5). Returns (pos, keep_mask)."""
        res = torch.as_tensor(self.conf.res)
        pos = torch.where(pos < 0, pos + res, pos)
        keep = ((pos >= 0) & (pos < res)).all(-1)
        return pos, keep

    def _wrap_clamp(self, pos):
        res = torch.as_tensor(self.conf.res)
        pos = torch.where(pos < 0, pos + res, pos)
        return torch.minimum(torch.maximum(pos, torch.zeros_like(pos)), res - 1)

    def p2g_micro(self, v, grid_v, grid_m, fx, w, base, affine):                                    # :178-194
        c, N = self.conf, self.n_particles
        idx = self.idx
        i, j, k = idx[:, 0], idx[:, 1], idx[:, 2]
        offset = idx[:, None, :].expand(27, N, 3)
        dpos = (offset.to(self.dtype) - fx) * c.dx
        weight = w[i][:, :, 0] * w[j][:, :, 1] * w[k][:, :, 2]
        pos_in_grid = base + offset
        grid_v_vals = weight.reshape(27, N, 1) * (c.p_mass * v + (affine @ dpos[..., None]).squeeze(-1))
        pos, keep = self._wrap_drop(pos_in_grid.reshape(-1, 3))
        wf = (weight.flatten() * c.p_mass)
        vals = grid_v_vals.reshape(-1, 3)
        grid_m = grid_m.index_put((pos[keep, 0], pos[keep, 1], pos[keep, 2]), wf[keep], accumulate=True)
        grid_v = grid_v.index_put((pos[keep, 0], pos[keep, 1], pos[keep, 2]), vals[keep], accumulate=True)
        return grid_v, grid_m

    def g2p_micro(self, grid_v, fx, w, base):                                                       # :196-221
        c, N = self.conf, self.n_particles
        idx = self.idx
        i, j, k = idx[:, 0], idx[:, 1], idx[:, 2]
        offset = idx[:, None, :].expand(27, N, 3)
        dpos = (offset.to(self.dtype) - fx).reshape(-1, 3)
        weight = (w[i][:, :, 0] * w[j][:, :, 1] * w[k][:, :, 2]).flatten()
        pos = self._wrap_clamp((base + offset).reshape(-1, 3))
        g_v = grid_v[pos[:, 0], pos[:, 1], pos[:, 2]]
        new_v = (weight[:, None] * g_v).reshape(27, N, 3).sum(0)
        outer = torch.einsum('ij,ik->ijk', g_v, dpos)
        new_C = (4 * weight[:, None, None] * outer * c.inv_dx).reshape(27, N, 3, 3).sum(0)
        return new_v, new_C

    def substep(self, f, st: MPMState):                                                             # :223-330
        c, N, dt_ = self.conf, self.n_particles, self.dtype
        res = tuple(c.res)
        grid_v = torch.zeros(res + (3,), dtype=dt_)
        grid_m = torch.zeros(res, dtype=dt_)
        liquid_mask = self.material == 0
        plastic_mask = self.material == 2

        base = (st.x * c.inv_dx - 0.5).to(torch.int32)                                              # :233 truncation
        fx = st.x * c.inv_dx - base.to(dt_)
        w = torch.stack([0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1) ** 2, 0.5 * (fx - 0.5) ** 2])      # :235
        eye = torch.eye(3, dtype=dt_)
        F_ = (eye[None] + c.dt * st.C) @ st.F                                                       # :238
        h = _clip(self.h, 0.1, 5)
        mu, la = st.mu * h, st.lamda * h
        mu = torch.where(liquid_mask, torch.zeros_like(mu), mu)
        la = torch.where(liquid_mask, torch.ones_like(la), la)
        U, sig, V = svd(F_)                                                                         # :246 (V is Vh)
        sig_ = _clip(sig, 1 - 2.5e-2 * 10, 1 + 4.5e-3 * 100)
        sig = torch.where(plastic_mask[..., None], sig_, sig)
        J = torch.ones((N,), dtype=dt_) * sig.prod(-1)
        J = J[..., None, None]
        sig_m = eye[None] * sig[..., None]
        F_ = torch.where(plastic_mask[..., None, None], U @ sig_m @ V, F_)                          # :257
        stress = 2 * mu[..., None, None] * (F_ - U @ V) @ F_.transpose(1, 2) \
            + eye.reshape(1, 3, 3) * la[..., None, None] * J * (J - 1)                              # :265-266
        stress = (-c.dt * c.p_vol * 4) * stress / c.dx ** 2                                         # :267
        affine = stress + c.p_mass * st.C                                                           # :268
        rep = lambda t: t[None].expand((27,) + tuple(t.shape))
        grid_v, grid_m = self.p2g_micro(rep(st.v), grid_v, grid_m, rep(fx), w, rep(base), rep(affine))

        prims = [forward_kinematics(f, p) for p in st.primitives]                                   # :277-278

        grid_v_ = grid_v / grid_m[..., None]                                                        # :283 (Q7)
        grid_v = torch.where(grid_m[..., None] > 0, grid_v_, grid_v)
        grid_v = grid_v + c.dt * self.gravity                                                       # :285
        grid_pos = self.grid_idx_3d.to(dt_) * c.dx
        for p in prims:                                                                             # :289-294
            if self.use_position_control:
                grid_v = position_control_batch(f, grid_pos, grid_v, c.dt, p)
            else:
                grid_v = collide_batch(f, grid_pos, grid_v, c.dt, p)

        normal = torch.tensor([0, 1, 0], dtype=dt_)                                                 # :297-307
        gi = self.grid_idx_3d.to(dt_)
        lin = (grid_v * normal).sum(-1) + 1e-30
        vit = grid_v - lin[..., None] * normal.reshape(1, 1, 1, 3) - gi * 1e-30
        lit = torch.sqrt(((vit + 1e-12) ** 2).sum(-1))
        grid_v_f = _clip(1. + st.friction * lin[..., None] / lit[..., None], 0., math.inf) * (vit + gi * 1e-30)
        grid_v_f = torch.cat([grid_v_f[..., :1], torch.zeros_like(grid_v_f[..., :1]), grid_v_f[..., 2:]], -1)
        friction_mask = torch.zeros(res + (3,), dtype=torch.bool)
        friction_mask[:, :3, :, :] = True
        fric_speed_mask = grid_v[..., 1] <= 0
        grid_v = torch.where(friction_mask & fric_speed_mask[..., None], grid_v_f, grid_v)

        cond = ((self.grid_idx_3d < 3) & (grid_v < 0)) | ((self.grid_idx_3d > c.n_grid - 3) & (grid_v > 0))  # :310-313 (Q8)
        grid_v = torch.where(cond, torch.zeros_like(grid_v), grid_v)

        v_, C_ = self.g2p_micro(grid_v, rep(fx), w, rep(base))                                      # :318-324
        x_ = st.x + c.dt * v_
        tr = sum(C_[i, i, :] for i in range(min(3, N)))            # C_.trace() on [N,3,3]: axes (0,1)  (Q6)
        J_ = st.J * (1 + c.dt * tr.sum(-1))
        return st._replace(x=x_, v=v_, C=C_, F=F_, J=J_, primitives=prims)

    def copy_frame(self, source, target, st: MPMState):                                             # :365-373
        prims = []
        for p in st.primitives:
            position = _set_drop(p.position, target, _gather_clamp(p.position, source))
            rotation = _set_drop(p.rotation, target, _gather_clamp(p.rotation, source))
            prims.append(p._replace(position=position, rotation=rotation))
        return st._replace(primitives=prims)

    def step(self, st: MPMState, action):                                                           # :413-429
        c = self.conf
        if self.clip_grads:
            st = st._replace(x=torch.nan_to_num(st.x), v=torch.nan_to_num(st.v), C=torch.nan_to_num(st.C),
                             F=torch.nan_to_num(st.F), J=torch.nan_to_num(st.J))
            leaves = [st.x, st.v, st.C, st.F, st.J, st.friction, st.mu, st.lamda]
            for p in st.primitives:
                leaves += [p.position, p.rotation, p.v, p.w, p.size, p.action_buffer, p.action_scale, p.friction]
            o = _NormGradTree.apply(*leaves)
            prims = [p._replace(position=o[8 + 8 * i], rotation=o[9 + 8 * i], v=o[10 + 8 * i], w=o[11 + 8 * i], size=o[12 + 8 * i],
                                action_buffer=o[13 + 8 * i], action_scale=o[14 + 8 * i], friction=o[15 + 8 * i])
                     for i, p in enumerate(st.primitives)]
            st = st._replace(x=o[0], v=o[1], C=o[2], F=o[3], J=o[4], friction=o[5], mu=o[6], lamda=o[7], primitives=prims)
            action = _NormGradTree.apply(action)[0]
        action = _clip(action, -1, 1)
        prims = [set_action(c.steps, action[i * 6:(i + 1) * 6], p) for i, p in enumerate(st.primitives)]
        st = st._replace(primitives=prims)
        for f in range(c.steps):
            st = self.substep(f, st)
        return self.copy_frame(c.steps, 0, st)


def make_prim(conf: MPMConf, size, init_pos, dtype=torch.float32, friction=0.1, softness=666.0):   # :31-60
    steps = conf.steps
    position = torch.zeros((steps, 3), dtype=dtype)
    position[0] = torch.as_tensor(init_pos, dtype=dtype)
    rotation = torch.tensor([[1., 0., 0., 0.]], dtype=dtype).repeat(steps, 1)
    return Prim(size=torch.as_tensor(size, dtype=dtype), position=position, rotation=rotation,
                v=torch.zeros((steps, 3), dtype=dtype), w=torch.zeros((steps, 3), dtype=dtype),
                friction=torch.tensor(friction, dtype=dtype), softness=torch.tensor(softness, dtype=dtype),
                action_buffer=torch.zeros(6, dtype=dtype), action_scale=torch.ones(6, dtype=dtype))


# ---- env-level hooks (mpm_env.py:99-125, whip_rope_env.py:108-115) ---------------------------------------
def focus_shift(conf: MPMConf, x):
    """pre_step: shift so that mean(x,z) sits at res*0.5/n_grid; y untouched."""
    target = torch.as_tensor(conf.res, dtype=x.dtype) * 0.5 / conf.n_grid
    shift = target - x.mean(0)
    shift = torch.stack([shift[0], torch.zeros_like(shift[0]), shift[2]])
    return shift
