"""ORACLE (test infrastructure, never shipped, never measured as the product).

torch (float64, dense n_grid^3 grid, batched over envs) twin of the Taichi "PlasticineLab" MLS-MPM substep, whose
`torch.autograd` plays the role of Taichi's reverse-mode kernels: what `substep_grad` computes
(/root/reference/GenORM/policy/pbm/plb/engine/mpm_simulator.py:271-289 = g2p.grad, grid_op.grad,
forward_kinematics.grad, p2g.grad, svd_grad :101-124, compute_F_tmp.grad) is the gradient of the same expressions
evaluated here, with the one hand-written piece -- `backward_svd` (:107-124, the clamp of :152-161) -- restated as a
custom autograd Function.  Also the loss kernels of engine/losses/loss.py:112-186 (density, SDF, soft / hard contact)
and the parameter leaves E, nu, yield_stress of PlasticineLab/sim2sim/plb/engine/mpm_simulator.py:27-29,490-495.

The forward arithmetic is oracle/twin/plb_twin.py's (the literal NumPy reading), vectorised; tests/test_plb.py holds the
two against each other.  PARITY UNPINNED: taichi is absent and the reference ships no recording or gradient of this path.
Differences of convention that cannot be settled without taichi (they only matter on measure-zero ties): the
sub-gradient of max / min / clamp at equality is torch's.
"""
from __future__ import annotations

import torch

from .plb_twin import PlbConf

DT = torch.float64


class _SvdRef(torch.autograd.Function):
    """ti.svd forward (A = U diag(sig) V^T, sig descending) with the reference's backward_svd (:107-124):
    F_ij = 1 / clamp(s_j^2 - s_i^2) off the diagonal (clamp :152-161: |.| >= 1e-6, sign kept), and
    gA = U ((F * (U^T gU - gU^T U)) sig) V^T + U (sig ((F * (V^T gV - gV^T V)) V^T)) + U gsig V^T."""

    @staticmethod
    def forward(ctx, A):
        U, S, Vh = torch.linalg.svd(A)
        V = Vh.transpose(-1, -2)
        ctx.save_for_backward(U, S, V)
        return U, S, V

    @staticmethod
    def backward(ctx, gU, gS, gV):
        U, S, V = ctx.saved_tensors
        sig = torch.diag_embed(S)
        Ut, Vt = U.transpose(-1, -2), V.transpose(-1, -2)
        s2 = S * S
        diff = s2[..., None, :] - s2[..., :, None]                    # [i, j] = s_j^2 - s_i^2
        cl = torch.where(diff >= 0, diff.clamp(min=1e-6), diff.clamp(max=-1e-6))
        Fm = 1.0 / cl
        eye = torch.eye(3, dtype=U.dtype, device=U.device)
        Fm = Fm * (1 - eye)
        gsig = torch.diag_embed(gS)
        u_term = U @ ((Fm * (Ut @ gU - gU.transpose(-1, -2) @ U)) @ sig) @ Vt
        v_term = U @ (sig @ ((Fm * (Vt @ gV - gV.transpose(-1, -2) @ V)) @ Vt))
        return u_term + v_term + U @ gsig @ Vt


def svd_ref(A):
    return _SvdRef.apply(A)


class PlbTorchTwin:
    def __init__(self, conf: PlbConf):
        self.c = conf

    # ---- one substep for B envs --------------------------------------------------------------------------
    def substep(self, x, v, C, F, pos_f, pos_f1, softness, E, nu, ys, fric):
        """x, v [B,N,3]; C, F [B,N,3,3]; pos_f, pos_f1 [B,np,3]; softness [B,np]; E, nu, ys, fric [B] (tensors)."""
        c = self.c
        n, dt, dx, inv_dx = c.n_grid, c.dt, c.dx, c.inv_dx
        B, N = x.shape[0], x.shape[1]
        I3 = torch.eye(3, dtype=DT)
        F_tmp = (I3 + dt * C) @ F                                                           # :91-94
        U, sig, V = svd_ref(F_tmp)                                                          # :96-99
        Vt = V.transpose(-1, -2)
        mu = (E / (2 * (1 + nu)))[:, None]                                                  # :168
        lam = (E * nu / ((1 + nu) * (1 - 2 * nu)))[:, None]
        base = (x.detach() * inv_dx - 0.5).to(torch.int64)                                  # cast(int): truncation, no gradient
        fx = x * inv_dx - base.to(DT)
        w = [0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1) ** 2, 0.5 * (fx - 0.5) ** 2]
        # compute_von_mises :133-150
        sg = torch.clamp(sig, min=0.05)
        eps = torch.log(sg)
        eps_hat = eps - eps.sum(-1, keepdim=True) / 3
        eps_hat_norm = torch.sqrt((eps_hat * eps_hat).sum(-1) + 1e-8)
        delta_gamma = eps_hat_norm - ys[:, None] / (2 * mu)
        yields = delta_gamma > 0
        eps_y = eps - (delta_gamma / eps_hat_norm)[..., None] * eps_hat
        F_y = (U * torch.exp(eps_y)[..., None, :]) @ Vt
        new_F = torch.where(yields[..., None, None], F_y, F_tmp)
        J = torch.linalg.det(new_F)
        r = U @ Vt
        stress = 2 * mu[..., None, None] * (new_F - r) @ new_F.transpose(-1, -2) + I3 * (lam * J * (J - 1))[..., None, None]
        stress = (-dt * c.p_vol * 4 * inv_dx * inv_dx) * stress
        affine = stress + c.p_mass * C
        G = n * n * n
        grid_v = torch.zeros((B, G, 3), dtype=DT)
        grid_m = torch.zeros((B, G), dtype=DT)
        lins, weights, dposs = [], [], []
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    off = torch.tensor([i, j, k], dtype=DT)
                    weight = w[i][..., 0] * w[j][..., 1] * w[k][..., 2]
                    idx = base + torch.tensor([i, j, k])
                    lin = (idx[..., 0] * n + idx[..., 1]) * n + idx[..., 2]
                    lins.append(lin); weights.append(weight); dposs.append(off - fx)
                    dpos = (off - fx) * dx
                    contrib = weight[..., None] * (c.p_mass * v + (affine @ dpos[..., None])[..., 0])
                    grid_v = grid_v.scatter_add(1, lin[..., None].expand(-1, -1, 3), contrib)
                    grid_m = grid_m.scatter_add(1, lin, weight * c.p_mass)
        # grid_op :200-232, dense and masked
        ar = torch.arange(n)
        Ig = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), -1).reshape(G, 3)       # [G,3] integer cell index
        gp = Ig.to(DT) * dx
        occ = grid_m > 1e-12
        safe_m = torch.where(occ, grid_m, torch.ones_like(grid_m))
        g30 = torch.tensor(c.gravity, dtype=DT) * dt * 30
        vo = grid_v / safe_m[..., None] + g30
        for pi in range(pos_f.shape[1]):                                                    # Sphere.collide (sticky) primitives.py:46-53
            d = gp[None] - pos_f[:, pi, None, :]
            dist = torch.sqrt((d * d).sum(-1) + 1e-14) - c.radius[pi]
            soft = softness[:, pi, None]
            infl = torch.clamp(torch.exp(-dist * soft), max=1.0)
            cond = (((soft > 0) & (infl > 0.1)) | (dist <= 0.001)) & (soft > 0)
            cv = ((pos_f1[:, pi] - pos_f[:, pi]) / dt)[:, None, :]                           # collider_v, identity rotations
            vo = torch.where(cond[..., None], cv.expand_as(vo), vo)
        Igf = Ig.to(DT)
        fr = fric[:, None]
        for d in range(3):
            lo = (Ig[None, :, d] < 3) & (vo[..., d] < 0)
            if d != 1:
                vo = torch.cat([torch.where(lo, torch.zeros_like(vo[..., e]), vo[..., e])[..., None] if e == d else vo[..., e:e + 1]
                                for e in range(3)], -1)
            else:
                lin_ = vo[..., 1] + 1e-30
                normal = torch.tensor([0.0, 1.0, 0.0], dtype=DT)
                vit = vo - lin_[..., None] * normal - Igf[None] * 1e-30
                lit = torch.sqrt((vit * vit).sum(-1) + 1e-8)
                sc = torch.clamp(1.0 + fr * lin_ / lit, min=0.0)
                vf = sc[..., None] * (vit + Igf[None] * 1e-30)
                vf = torch.cat([vf[..., 0:1], torch.zeros_like(vf[..., 1:2]), vf[..., 2:3]], -1)
                zero_all = torch.zeros_like(vo)
                only_y = torch.cat([vo[..., 0:1], torch.zeros_like(vo[..., 1:2]), vo[..., 2:3]], -1)
                branch = torch.where((fr == 0)[..., None], only_y, torch.where((fr < 10)[..., None], vf, zero_all))
                vo = torch.where(lo[..., None], branch, vo)
            hi = (Ig[None, :, d] > n - 3) & (vo[..., d] > 0)
            vo = torch.cat([torch.where(hi, torch.zeros_like(vo[..., e]), vo[..., e])[..., None] if e == d else vo[..., e:e + 1]
                            for e in range(3)], -1)
        out = torch.where(occ[..., None], vo, torch.zeros_like(vo))
        # g2p :234-253
        new_v = torch.zeros((B, N, 3), dtype=DT)
        new_C = torch.zeros((B, N, 3, 3), dtype=DT)
        for lin, weight, dpos in zip(lins, weights, dposs):
            g_v = out.gather(1, lin[..., None].expand(-1, -1, 3))
            new_v = new_v + weight[..., None] * g_v
            new_C = new_C + 4 * inv_dx * weight[..., None, None] * (g_v[..., :, None] * dpos[..., None, :])
        new_x = torch.clamp(x + dt * new_v, min=0.0, max=1.0 - 3 * dx)
        return new_x, new_v, new_C, new_F

    def step(self, x, v, C, F, prim_pos, action, softness, E, nu, ys, fric):
        """TaichiEnv.step in copy mode (:438-449): set_action (clip +-1, v = a * scale / substeps for primitive 0),
        `substeps` substeps, copy frame cur -> 0.  All arguments are tensors (requires_grad where a gradient is wanted)."""
        c = self.c
        S = c.substeps
        a = torch.clamp(action, -1, 1)
        pv = torch.zeros_like(prim_pos)
        pv = torch.cat([(a[:, :3] / S)[:, None, :], pv[:, 1:]], 1)
        lo, hi = torch.tensor(c.lower_bound, dtype=DT), torch.tensor(c.upper_bound, dtype=DT)
        pos = prim_pos
        for _ in range(S):
            pos1 = torch.maximum(torch.minimum(pos + pv, hi), lo)                            # forward_kinematics :118-121
            x, v, C, F = self.substep(x, v, C, F, pos, pos1, softness, E, nu, ys, fric)
            pos = pos1
        return x, v, C, F, pos

    # ---- losses (engine/losses/loss.py) ----------------------------------------------------------------------
    def grid_mass(self, x):
        """compute_grid_m_kernel (mpm_simulator.py:456-466): the p2g of the masses only."""
        c = self.c
        n, inv_dx = c.n_grid, c.inv_dx
        B = x.shape[0]
        base = (x.detach() * inv_dx - 0.5).to(torch.int64)
        fx = x * inv_dx - base.to(DT)
        w = [0.5 * (1.5 - fx) ** 2, 0.75 - (fx - 1) ** 2, 0.5 * (fx - 0.5) ** 2]
        gm = torch.zeros((B, n * n * n), dtype=DT)
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    idx = base + torch.tensor([i, j, k])
                    lin = (idx[..., 0] * n + idx[..., 1]) * n + idx[..., 2]
                    gm = gm.scatter_add(1, lin, w[i][..., 0] * w[j][..., 1] * w[k][..., 2] * c.p_mass)
        return gm

    def loss(self, x, prim_pos, target_density, target_sdf, weights, soft_contact=True):
        """compute_loss_kernel :190-214: density = sum |grid_mass - target_density| (:145-148), sdf = sum target_sdf * grid_mass
        (:150-153), contact = sum_primitives min_dist^2 (:137-140) with, soft (:126-135): min_dist = sum_i d_i w(d_i) / sum_i w(d_i),
        w(d) = 1 / (1 + 1e4 d^2), d_i = max(sdf(x_i), 0); hard (:120-124): min_dist = min_i d_i.  loss = sum of the three, weighted
        (:158-162).  weights = (contact, density, sdf).  Returns (loss [B], parts [B,3] = contact, density, sdf)."""
        c = self.c
        gm = self.grid_mass(x)
        density = (gm - target_density[None]).abs().sum(-1)
        sdf = (target_sdf[None] * gm).sum(-1)
        contact = torch.zeros_like(density)
        for pi in range(prim_pos.shape[1]):
            d = x - prim_pos[:, pi, None, :]
            dij = torch.clamp(torch.sqrt((d * d).sum(-1) + 1e-14) - c.radius[pi], min=0.0)
            if soft_contact:
                sw = 1 / (1 + dij * dij * 10000)
                md = (dij * sw / sw.sum(-1, keepdim=True)).sum(-1)
            else:
                md = dij.min(-1).values
            contact = contact + md ** 2
        total = contact * weights[0] + density * weights[1] + sdf * weights[2]
        return total, torch.stack([contact, density, sdf], -1)
