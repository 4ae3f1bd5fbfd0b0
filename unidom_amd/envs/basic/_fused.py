"""Fused device kernels for the cloth-env arithmetic around the rollout (libunidom_hip: csrc/env_glue.hip).

The reference jit-compiles step_diff, so XLA fuses calc_chamfer (core/utils/util.py:138-153), contact_distance
(cloth_env.py:206-209) and get_pnp_actions (cloth_env.py:134-173) into a few kernels; op by op they are ~180 tiny
launches and a dense [B,P,Q] autograd graph per step_diff.  These autograd Functions run one kernel forward and one
backward each; the op-by-op torch forms stay in utils/util.py / ClothEnv.get_pnp_actions as the readable
restatement the tests compare them with.
"""
from __future__ import annotations

import ctypes as C

import torch

from ... import _lib


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class ChamferFn(torch.autograd.Function):
    """calc_chamfer(x [B,P,3], y [Q,3]) -> [B]; gradient w.r.t. x only (y is the fixed goal cloud)."""

    @staticmethod
    def forward(ctx, x, y):
        x = x.detach().to(torch.float32).contiguous()
        y = y.detach().to(torch.float32).contiguous()
        B, P, Q = x.shape[0], x.shape[1], y.shape[0]
        out = torch.empty((B,), dtype=torch.float32, device=x.device)
        ixy = torch.empty((B, P), dtype=torch.int32, device=x.device)
        iyx = torch.empty((B, Q), dtype=torch.int32, device=x.device)
        _lib.check(_lib.lib().ud_chamfer_fwd(C.c_int(B), C.c_int(P), C.c_int(Q), _lib.ptr(x), _lib.ptr(y), _lib.ptr(out),
                                             _lib.ptr(ixy), _lib.ptr(iyx), _stream(x.device)), "ud_chamfer_fwd")
        ctx.save_for_backward(x, y, ixy, iyx)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, ixy, iyx = ctx.saved_tensors
        B, P, Q = x.shape[0], x.shape[1], y.shape[0]
        g = g.to(torch.float32).contiguous()
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().ud_chamfer_bwd(C.c_int(B), C.c_int(P), C.c_int(Q), _lib.ptr(x), _lib.ptr(y), _lib.ptr(ixy),
                                             _lib.ptr(iyx), _lib.ptr(g), _lib.ptr(gx), _stream(x.device)), "ud_chamfer_bwd")
        return gx, None


class PnpContactFn(torch.autograd.Function):
    """(actions [B,6], primitive0 [B,4], x [B,P,3]) -> (macro_actions [40,B,8], contact_distance [B])."""

    @staticmethod
    def forward(ctx, actions, primitive0, x):
        a = actions.detach().to(torch.float32).contiguous()
        p0 = primitive0.detach().to(torch.float32).contiguous()
        xc = x.detach().to(torch.float32).contiguous()
        B, P = xc.shape[0], xc.shape[1]
        macro = torch.empty((40, B, 8), dtype=torch.float32, device=xc.device)
        contact = torch.empty((B,), dtype=torch.float32, device=xc.device)
        idx = torch.empty((B,), dtype=torch.int32, device=xc.device)
        _lib.check(_lib.lib().ud_cloth_pnp_fwd(C.c_int(B), C.c_int(P), _lib.ptr(a), _lib.ptr(p0), _lib.ptr(xc), _lib.ptr(macro),
                                               _lib.ptr(contact), _lib.ptr(idx), _stream(xc.device)), "ud_cloth_pnp_fwd")
        ctx.save_for_backward(a, xc, contact, idx)
        return macro, contact

    @staticmethod
    def backward(ctx, g_macro, g_contact):
        a, xc, contact, idx = ctx.saved_tensors
        B, P = xc.shape[0], xc.shape[1]
        g_macro = (torch.zeros((40, B, 8), dtype=torch.float32, device=xc.device) if g_macro is None
                   else g_macro.to(torch.float32).contiguous())
        g_contact = None if g_contact is None else g_contact.to(torch.float32).contiguous()
        ga = torch.empty_like(a)
        gp = torch.empty((B, 4), dtype=torch.float32, device=xc.device)
        gx = torch.empty_like(xc)
        _lib.check(_lib.lib().ud_cloth_pnp_bwd(C.c_int(B), C.c_int(P), _lib.ptr(a), _lib.ptr(xc), _lib.ptr(contact), _lib.ptr(idx),
                                               _lib.ptr(g_macro), _lib.ptr(g_contact), _lib.ptr(ga), _lib.ptr(gp), _lib.ptr(gx),
                                               _stream(xc.device)), "ud_cloth_pnp_bwd")
        return ga, gp, gx


def chamfer(x, goal):
    return ChamferFn.apply(x, goal)


def pnp_and_contact(actions, primitive0, x):
    return PnpContactFn.apply(actions, primitive0, x)


# ---- MPM envs (csrc/env_glue.hip, second half) ---------------------------------------------------------------------
def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _ptr_array(tensors):
    """HOST array of device pointers (include/unidom_hip.h: prim_pos arguments); None entries become NULL."""
    arr = (C.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else _lib.ptr(t).value
    return arr


class FocusFn(torch.autograd.Function):
    """pre_step (mpm_env.py:99-114): (x [B,N,3], *primitive positions [B,S,3]) -> (x + shift, shift [B,3], *positions + shift)."""

    @staticmethod
    def forward(ctx, center, x, *pos):
        x = _f32(x)
        pos = [_f32(p) for p in pos]
        B, N, S = x.shape[0], x.shape[1], pos[0].shape[1] if pos else 1
        xo, po = torch.empty_like(x), [torch.empty_like(p) for p in pos]
        shift = torch.empty((B, 3), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ud_mpm_focus_fwd(C.c_int(B), C.c_int(N), C.c_int(len(pos)), C.c_int(S), C.c_float(center[0]),
                                               C.c_float(center[2]), _lib.ptr(x), _ptr_array(pos), _lib.ptr(xo), _ptr_array(po),
                                               _lib.ptr(shift), _stream(x.device)), "ud_mpm_focus_fwd")
        ctx.dims = (B, N, len(pos), S, x.device)
        ctx.set_materialize_grads(False)
        return (xo, shift, *po)

    @staticmethod
    def backward(ctx, g_xo, g_shift, *g_po):
        B, N, P, S, dev = ctx.dims
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_xo, g_shift, g_po = c(g_xo), c(g_shift), [c(g) for g in g_po]
        gx = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().ud_mpm_focus_bwd(C.c_int(B), C.c_int(N), C.c_int(P), C.c_int(S), _lib.ptr(g_xo), _ptr_array(g_po),
                                               _lib.ptr(g_shift), _lib.ptr(gx), _stream(dev)), "ud_mpm_focus_bwd")
        return (None, gx, *g_po)


class FinishFn(torch.autograd.Function):
    """Tail of step_diff (mpm_env.py:116-125, :150-154, :90-94, :57-76):
    (x, v, C, F, J, shift | None, goal [N,3], *positions) -> (x, v, C, F, J, reward [B], obs [B, 6N+3S], *positions)."""

    @staticmethod
    def forward(ctx, x, v, Cm, F, J, shift, goal, *pos):
        x, v, Cm, F, J, goal = map(_f32, (x, v, Cm, F, J, goal))
        shift = None if shift is None else _f32(shift)
        pos = [_f32(p) for p in pos]
        B, N, S = x.shape[0], x.shape[1], pos[0].shape[1]
        if goal.shape not in ((N, 3), (1, 3)):
            raise _lib.UnidomError(f"reward_func: goal {tuple(goal.shape)} does not broadcast against the {N} particles of the state")
        Q = goal.shape[0]
        xo, vo, Co, Fo, Jo = (torch.empty_like(t) for t in (x, v, Cm, F, J))
        po = [torch.empty_like(p) for p in pos]
        reward = torch.empty((B,), dtype=torch.float32, device=x.device)
        obs = torch.empty((B, 6 * N + 3 * S), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ud_mpm_finish_fwd(
            C.c_int(B), C.c_int(N), C.c_int(len(pos)), C.c_int(S), C.c_int(Q), *[_lib.ptr(t) for t in (x, v, Cm, F, J, shift)], _ptr_array(pos),
            _lib.ptr(goal), *[_lib.ptr(t) for t in (xo, vo, Co, Fo, Jo)], _ptr_array(po), _lib.ptr(reward), _lib.ptr(obs),
            _stream(x.device)), "ud_mpm_finish_fwd")
        ctx.save_for_backward(x, v, Cm, F, shift, goal, reward)
        ctx.dims = (B, N, len(pos), S)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(Jo)
        return (xo, vo, Co, Fo, Jo, reward, obs, *po)

    @staticmethod
    def backward(ctx, g_x, g_v, g_C, g_F, g_J, g_reward, g_obs, *g_po):
        x, v, Cm, F, shift, goal, reward = ctx.saved_tensors
        B, N, P, S = ctx.dims
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_x, g_v, g_C, g_F, g_reward, g_obs, g_po = c(g_x), c(g_v), c(g_C), c(g_F), c(g_reward), c(g_obs), [c(g) for g in g_po]
        ox, ov, oC, oF = (torch.empty_like(t) for t in (x, v, Cm, F))
        opos = [torch.empty((B, S, 3), dtype=torch.float32, device=x.device) for _ in range(P)]
        osh = None if shift is None else torch.empty((B, 3), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ud_mpm_finish_bwd(
            C.c_int(B), C.c_int(N), C.c_int(P), C.c_int(S), C.c_int(goal.shape[0]), *[_lib.ptr(t) for t in (x, v, Cm, F, shift, goal, reward)],
            *[_lib.ptr(t) for t in (g_x, g_v, g_C, g_F)], _ptr_array(g_po), _lib.ptr(g_reward), _lib.ptr(g_obs),
            *[_lib.ptr(t) for t in (ox, ov, oC, oF)], _ptr_array(opos), _lib.ptr(osh), _stream(x.device)), "ud_mpm_finish_bwd")
        return (ox, ov, oC, oF, None, osh, None, *opos)


def mpm_focus(center, x, positions):
    out = FocusFn.apply(tuple(float(c) for c in center), x, *positions)
    return out[0], out[1], list(out[2:])


def mpm_finish(x, v, Cm, F, J, shift, goal, positions):
    out = FinishFn.apply(x, v, Cm, F, J, shift, goal, *positions)
    return out[:5], out[5], out[6], list(out[7:])
