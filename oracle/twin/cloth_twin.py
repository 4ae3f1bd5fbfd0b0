"""ORACLE (test infrastructure, never shipped, never measured as the product).

Line-by-line CPU restatement of the reference's mass-spring cloth simulator in
torch (CPU tensors; torch.autograd plays the role jax.grad plays in the reference).
Follows /root/reference/DaXBench/daxbench/core/engine/cloth_simulator.py:

  tables          <- ClothSimulator.__init__      cloth_simulator.py:48-66
  NormGrad        <- live norm_grad (2nd def)     cloth_simulator.py:182-196
  gripper()       <- primitive_collision_func     cloth_simulator.py:198-226
  step()          <- step                         cloth_simulator.py:257-337
  robot_step()    <- robot_step                   cloth_simulator.py:163-180
  reset()         <- reset                        cloth_simulator.py:339-364

Parity status: the reference's cloth demos were recorded by an older cloth step
(SURVEY.md F3), so cloth x/v parity is UNPINNED by reference data; this twin is
pinned only by reading + analytic known answers (tests/test_oracle_cloth.py).
Third-party semantics assumed: jnp.clip = min(max(x,lo),hi)
whose gradient splits ties 0.5/0.5 (torch.maximum/minimum do the same), and
jnp.nan_to_num defaults (nan->0, +-inf->+-float max).

It is deliberately slow and literal (dense 80x80 scatter/gather like the source);
the fast C++ oracle (oracle/csrc) is validated against it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import NamedTuple

import numpy as np
import torch


@dataclass
class ClothConf:                      # fold_cloth1_env.py:15-33
    N: int = 80
    gravity: float = 0.5
    stiffness: float = 900
    damping: float = 2
    dt: float = 2e-3
    max_v: float = 2.0
    small_num: float = 1e-8
    mu: float = 0.5
    size: int = 16
    substeps: int = 50                # cloth_simulator.py:176


class ClothState(NamedTuple):         # cloth_simulator.py:13-23 (key/cur_step omitted: not differentiated)
    x: torch.Tensor
    v: torch.Tensor
    primitive0: torch.Tensor
    primitive1: torch.Tensor
    action0: torch.Tensor
    action1: torch.Tensor
    stiffness: torch.Tensor
    mu: torch.Tensor


def fold_cloth1_mask(conf: ClothConf) -> np.ndarray:   # fold_cloth1_env.py:48-53
    m = np.zeros((conf.N, conf.N), dtype=np.float32)
    s = conf.size
    m[s * 2:s * 3, s * 2:s * 4] = 1
    return m


def _recip(c, like):
    """x / c with a compile-time constant c, as XLA's AlgebraicSimplifier rewrites it under jit: x * (1 / c), the reciprocal
    rounded once in the array's dtype.  The reference's recorded primitive trajectories discriminate the two forms for / 3
    (get_pnp_actions) and / 50 (robot_step): tests/test_oracle_cloth.py::test_sibling_demo_*."""
    one = torch.ones((), dtype=like.dtype)
    return one / torch.tensor(float(c), dtype=like.dtype)


def _clip(x, lo, hi):
    """jnp.clip: minimum(hi, maximum(lo, x)); tie gradients split evenly."""
    lo_t = torch.as_tensor(lo, dtype=x.dtype)
    hi_t = torch.as_tensor(hi, dtype=x.dtype)
    return torch.minimum(torch.maximum(x, lo_t), hi_t)


class _NormGrad(torch.autograd.Function):   # cloth_simulator.py:182-196
    @staticmethod
    def forward(ctx, x, n_mask, enabled):
        ctx.n_mask = n_mask
        ctx.enabled = enabled
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if not ctx.enabled:
            return g, None, None
        g = g / torch.linalg.norm(g)
        g = torch.nan_to_num(g)
        g = g / ctx.n_mask
        return g, None, None


class ClothTwin:
    def __init__(self, conf: ClothConf, cloth_mask: np.ndarray, dtype=torch.float32, normalize=True):
        self.conf, self.dtype, self.normalize = conf, dtype, normalize
        N = conf.N
        self.cell_size = 1.0 / N
        self.mask = torch.as_tensor(cloth_mask, dtype=dtype)
        self.n_mask = float(cloth_mask.sum())
        links = np.array([[-1, 0], [1, 0], [0, -1], [0, 1], [-1, -1], [1, -1], [-1, 1], [1, 1]])  # :48
        idx_i, idx_j = np.nonzero(cloth_mask)                                                      # :52
        grid_idx = np.stack([idx_i, idx_j], -1)
        j_ = np.repeat(grid_idx.reshape(-1, 1, 2), 8, axis=1) + links[None]                      # :56-57
        j_ = np.clip(j_, 0, N - 1)                                                                 # :58
        i_ = np.repeat(grid_idx.reshape(-1, 1, 2), 8, axis=1)                                     # :60
        # jnp.linalg.norm on int32 -> float32; cell_size is a python float (weak) -> f32 product
        npdt = np.float64 if dtype == torch.float64 else np.float32   # f64 twin: everything in double
        nrm = np.sqrt(((j_ - i_).astype(npdt) ** 2).sum(-1))
        ol = (npdt(self.cell_size) * nrm)[..., None]                                              # :61
        self.ori_len_is_not_0 = torch.as_tensor((ol != 0).astype(np.float32), dtype=dtype)       # :62
        self.original_length = torch.as_tensor(np.clip(ol, 1e-12, np.inf), dtype=dtype)           # :63
        self.idx_i, self.idx_j = torch.as_tensor(idx_i), torch.as_tensor(idx_j)
        self.j_x, self.j_y = torch.as_tensor(j_.reshape(-1, 2)[:, 0]), torch.as_tensor(j_.reshape(-1, 2)[:, 1])
        self.i_x, self.i_y = torch.as_tensor(i_.reshape(-1, 2)[:, 0]), torch.as_tensor(i_.reshape(-1, 2)[:, 1])
        self.P = len(idx_i)

    # -- helpers ---------------------------------------------------------------------------
    def _t(self, a):
        return torch.as_tensor(a, dtype=self.dtype)

    def norm_grad(self, x):
        return _NormGrad.apply(x, self.n_mask, self.normalize)

    def reset(self) -> ClothState:                                                                 # :339-364
        N, c = self.conf.N, self.cell_size
        xg = np.zeros((N, N, 3))
        for i, j in np.ndindex((N, N)):
            xg[i, j] = np.array([i * c, 0, (N - j) * c])
        x = self._t(xg)[self.idx_i, self.idx_j]
        return ClothState(x=x, v=torch.zeros_like(x),
                          primitive0=self._t([0.5, 0.5, 0.5, 0.01]), primitive1=self._t([1, 1, 1, 0.01]),
                          action0=self._t([0, 0, 0, 0]), action1=self._t([0, 0, 0, 0]),
                          stiffness=self._t(self.conf.stiffness), mu=self._t(self.conf.mu))

    def gripper(self, x, v, action, ps):                                                           # :198-226
        pos, radius = ps[:3], ps[3]
        d_v = action[:3].reshape(1, 3)
        suction = action[-1]
        x_ = x - pos.reshape(1, 3)
        dist = torch.sqrt(x_[:, 0] * x_[:, 0] + x_[:, 1] * x_[:, 1] + x_[:, 2] * x_[:, 2])
        mask = (dist <= radius)[..., None].expand(-1, 3)
        v_ = torch.where(mask, suction * v, v)
        x_ = torch.where(mask, x + d_v * (1 - suction), x)
        x = self.norm_grad(x_)
        v = self.norm_grad(v_)
        return x, v, mask[:, 0]

    def step(self, st: ClothState, want_mask=False):                                               # :257-337
        c = self.conf
        N = c.N
        x, v = st.x, st.v
        v = v - self._t([0, c.gravity * c.dt, 0])                                                  # :259
        x_grid = torch.zeros((N, N, 3), dtype=self.dtype).index_put((self.idx_i, self.idx_j), x)   # :261
        relative_pos = x_grid[self.j_x, self.j_y] - x_grid[self.i_x, self.i_y]                     # :262
        sq = relative_pos ** 2
        # (rel**2).sum(-1): XLA:CPU reduces the minor dimension sequentially -> ((x2+y2)+z2)
        current_length = torch.sqrt(_clip(sq[:, 0] + sq[:, 1] + sq[:, 2], 1e-12, math.inf))         # :264
        current_length = current_length.reshape(-1, 8, 1)
        force = st.stiffness * relative_pos.reshape(-1, 8, 3) / current_length * (
            current_length - self.original_length) / self.original_length                          # :267-268
        force = force * self.ori_len_is_not_0                                                      # :273
        force = force * self.mask[self.j_x, self.j_y].reshape(-1, 8, 1)                            # :276
        fsum = force[:, 0]
        for l in range(1, 8):                      # force.sum(1): sequential over the 8 links
            fsum = fsum + force[:, l]
        force = fsum                                                                               # :277
        force = force + self._t([0, -c.gravity, 0])                                                # :278

        friction_mask = x[:, 1] <= c.small_num                                                     # :281
        muF = st.mu * _clip(force[:, 1], -math.inf, 0) * -1                                        # :282
        xV, yV = v[:, 0], v[:, 2]
        sV = torch.sqrt(xV ** 2 + yV ** 2 + c.small_num)                                           # :285
        dyn = (friction_mask & (sV > c.small_num)).to(self.dtype)                                  # :288
        f0 = force[:, 0] - dyn * muF * xV / sV                                                     # :289
        f2 = force[:, 2] - dyn * muF * yV / sV                                                     # :290
        force = torch.stack([f0, force[:, 1], f2], -1)
        static = friction_mask & (sV <= c.small_num)                                               # :293
        xF, yF = force[:, 0], force[:, 2]
        sF = torch.sqrt(xF ** 2 + yF ** 2 + c.small_num)                                           # :296
        zero_m = (static & (muF > sF)).to(self.dtype)                                              # :298
        f0 = 0 + (1.0 - zero_m) * force[:, 0]                                                      # :299
        f2 = 0 + (1.0 - zero_m) * force[:, 2]                                                      # :300
        nz = (static & (muF <= sF)).to(self.dtype)                                                 # :302-303
        R = 1.0 - muF / sF                                                                         # :304
        f0 = (R * xF) * nz + f0 * (1.0 - nz)                                                       # :305
        f2 = (R * yF) * nz + f2 * (1.0 - nz)                                                       # :306
        force = torch.stack([f0, force[:, 1], f2], -1)

        v = v + force * c.dt                                                                       # :308
        v = v * math.exp(-c.damping * c.dt) if self.dtype == torch.float64 else \
            v * float(np.float32(np.exp(np.float64(np.float32(-c.damping * c.dt)))))                                       # :309
        # :312 collision_func is the identity (cloth_env.py:239-243)
        x, v, m0 = self.gripper(x, v, st.action0, st.primitive0)                                   # :313
        x, v, m1 = self.gripper(x, v, st.action1, st.primitive1)                                   # :314
        ps0 = _clip(st.primitive0 + torch.cat([st.action0[:3], st.action0.new_zeros(1)]), 0, 1)    # :322
        ps1 = _clip(st.primitive1 + torch.cat([st.action1[:3], st.action1.new_zeros(1)]), 0, 1)    # :323
        x = _clip(x, 0, 1)                                                                         # :326
        v = _clip(v, -c.max_v, c.max_v)                                                            # :327
        x = x + c.dt * v                                                                           # :329
        x, v = self.norm_grad(x), self.norm_grad(v)                                                # :331-332
        ps0, ps1 = self.norm_grad(ps0), self.norm_grad(ps1)                                        # :333-334
        out = st._replace(x=x, v=v, primitive0=ps0, primitive1=ps1)
        return (out, m0, m1) if want_mask else out

    def robot_step(self, st: ClothState, action, record=None):                                     # :163-180
        a0 = torch.cat([_clip(action[:3], -2, 2) * _recip(50, action), action[3:4]])                             # :168
        a1 = torch.cat([_clip(action[4:7], -2, 2) * _recip(50, action), action[7:8]])                            # :169
        st = st._replace(action0=a0, action1=a1)
        for _ in range(self.conf.substeps):                                                        # :176
            if record is not None:
                st, m0, m1 = self.step(st, want_mask=True)
                record.append((m0.nonzero().flatten().tolist(), m1.nonzero().flatten().tolist()))
            else:
                st = self.step(st)
        return st
