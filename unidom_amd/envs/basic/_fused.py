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
