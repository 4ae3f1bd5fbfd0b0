"""ORACLE loader (test infrastructure only): ctypes bindings of oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (unidom_amd) must never import anything from oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return so


def use_native():
    """bench.py's cpu_baseline leg only: (re)build the restatement with -O3 -march=native ON THIS MACHINE and
    bind this process to it.  Same sources, same -ffp-contract=off / no fast-math; only the instruction set differs."""
    global _LIB
    subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "native"])
    _LIB = None
    lib(os.path.join(_HERE, "_native", "liboracle.so"))


def lib(path=None):
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(path or build())
        _LIB.oc_cloth_create.restype = C.c_void_p
        if hasattr(_LIB, "oc_mpm_create"):
            _LIB.oc_mpm_create.restype = C.c_void_p
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _suf(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def damp_factor(damping, dt, dtype):
    """jnp.exp(-damping*dt): python-float product, evaluated in the array dtype (cloth_simulator.py:309)."""
    if np.dtype(dtype) == np.float32:
        # correctly rounded f32 exp of the f32 argument (numpy's own f32 exp is 1 ulp off here)
        return float(np.float32(np.exp(np.float64(np.float32(-damping * dt)))))
    return float(np.exp(-damping * dt))


class ClothOracle:
    """CPU restatement of ClothSimulator's robot_step rollout + adjoint (cloth_simulator.py:163-337)."""

    def __init__(self, mask, N=80, gravity=0.5, damping=2, dt=2e-3, max_v=2.0, small_num=1e-8, substeps=50, order=1):
        """order=1: the reference's operation order; order=2: the re-associated IEEE order of the default HIP forward."""
        self.N, self.S = N, substeps
        self.mask = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        L = lib()
        self.h = C.c_void_p(L.oc_cloth_create(
            C.c_int(N), _p(self.mask), C.c_double(gravity), C.c_double(dt),
            C.c_double(damp_factor(damping, dt, np.float32)), C.c_double(damp_factor(damping, dt, np.float64)),
            C.c_double(max_v), C.c_double(small_num), C.c_int(substeps)))
        self.P = L.oc_cloth_num_particles(self.h)
        L.oc_cloth_set_order(self.h, C.c_int(order))

    def __del__(self):
        try:
            lib().oc_cloth_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        nbr = np.zeros((self.P, 8), np.int32)
        L0 = np.zeros((self.P, 8), np.float32)
        lib().oc_cloth_tables(self.h, _p(nbr), _p(L0))
        return nbr, L0

    def rollout_fwd(self, x0, v0, prim0, k, mu, actions, want_lists=False, want_grasp=False, want_ckpt=False,
                    nthreads=1):
        dt = x0.dtype
        B, P = x0.shape[0], self.P
        T = actions.shape[0]
        c = lambda a: np.ascontiguousarray(a, dtype=dt)
        x0, v0, prim0, k, mu, actions = map(c, (x0, v0, prim0, k, mu, actions))
        assert x0.shape == (B, P, 3) and prim0.shape == (B, 2, 4) and actions.shape == (T, B, 8)
        xo, vo, po = np.empty_like(x0), np.empty_like(v0), np.empty_like(prim0)
        xl = np.empty((T, B, P, 3), dt) if want_lists else None
        vl = np.empty((T, B, P, 3), dt) if want_lists else None
        pl = np.empty((T, B, 2, 4), dt) if want_lists else None
        gr = np.zeros((T, self.S, B, 2, P), np.uint8) if want_grasp else None
        ck = np.empty((B, T, self.S, P * 6 + 8), dt) if want_ckpt else None
        getattr(lib(), "oc_cloth_rollout_fwd_" + _suf(dt))(
            self.h, C.c_int(B), C.c_int(T), _p(x0), _p(v0), _p(prim0), _p(k), _p(mu), _p(actions),
            _p(xo), _p(vo), _p(po), _p(xl), _p(vl), _p(pl), _p(gr), _p(ck), C.c_int(nthreads))
        out = dict(x=xo, v=vo, prim=po)
        if want_lists:
            out.update(x_list=xl, v_list=vl, prim_list=pl)
        if want_grasp:
            out["grasp"] = gr
        if want_ckpt:
            out["ckpt"] = ck
        return out

    def rollout_bwd(self, x0, v0, prim0, k, mu, actions, gx, gv, gprim, gx_list=None, gv_list=None,
                    gprim_list=None, normalize=True, nthreads=1):
        dt = x0.dtype
        B, P = x0.shape[0], self.P
        T = actions.shape[0]
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
        x0, v0, prim0, k, mu, actions, gx, gv, gprim, gx_list, gv_list, gprim_list = map(
            c, (x0, v0, prim0, k, mu, actions, gx, gv, gprim, gx_list, gv_list, gprim_list))
        gx0, gv0, gp0 = np.empty_like(x0), np.empty_like(v0), np.empty_like(prim0)
        ga = np.empty((T, B, 8), dt)
        gk, gmu = np.empty((B,), dt), np.empty((B,), dt)
        getattr(lib(), "oc_cloth_rollout_bwd_" + _suf(dt))(
            self.h, C.c_int(B), C.c_int(T), _p(x0), _p(v0), _p(prim0), _p(k), _p(mu), _p(actions),
            _p(gx), _p(gv), _p(gprim), _p(gx_list), _p(gv_list), _p(gprim_list), C.c_int(int(normalize)),
            _p(gx0), _p(gv0), _p(gp0), _p(ga), _p(gk), _p(gmu), C.c_int(nthreads))
        return dict(gx=gx0, gv=gv0, gprim=gp0, gactions=ga, gk=gk, gmu=gmu)


class MpmOracle:
    """CPU restatement of SimpleMPMSimulator.step (mpm_simulator.py:413-429) + adjoint, dense grid."""

    def __init__(self, N, n_grid=64, res=(32, 32, 32), steps=70, dt=1e-4, p_rho=1.0, gravity=(0, -9.8, 0),
                 position_control=True, material=None, hardness=None, prim_friction=0.1, prim_softness=666.0,
                 n_prim=1, sdf="box"):
        self.N, self.steps, self.res, self.n_grid, self.P = N, steps, tuple(res), n_grid, n_prim
        dx = 1 / n_grid
        p_vol = (dx * 0.5) ** 2
        p_mass = p_vol * p_rho
        mat = np.ascontiguousarray(np.full(N, 1) if material is None else material, dtype=np.int32)
        hd = np.ascontiguousarray(np.full(N, 1.0) if hardness is None else hardness, dtype=np.float64)
        r = np.ascontiguousarray(res, dtype=np.int32)
        g = np.ascontiguousarray(gravity, dtype=np.float64)
        self.h = C.c_void_p(lib().oc_mpm_create(
            C.c_int(N), C.c_int(n_grid), _p(r), C.c_int(steps), C.c_double(dt), C.c_double(p_mass), C.c_double(p_vol),
            _p(g), C.c_int(int(position_control)), _p(mat), _p(hd), C.c_double(float(np.ravel(prim_friction)[0])), C.c_double(float(np.ravel(prim_softness)[0])),
            C.c_int(n_prim), C.c_int({"box": 0, "container": 1}[sdf])))
        assert self.h.value, "oc_mpm_create refused the configuration"
        if not np.isscalar(prim_friction) or not np.isscalar(prim_softness):     # one value per primitive
            fr = np.ascontiguousarray(np.broadcast_to(np.asarray(prim_friction, np.float64), (n_prim,)))
            so = np.ascontiguousarray(np.broadcast_to(np.asarray(prim_softness, np.float64), (n_prim,)))
            lib().oc_mpm_set_prim_each(self.h, C.c_int(n_prim), _p(fr), _p(so))

    def __del__(self):
        try:
            lib().oc_mpm_destroy(self.h)
        except Exception:
            pass

    def _prep(self, st, dt):
        c = lambda a: np.ascontiguousarray(a, dtype=dt)
        return [c(st[k]) for k in ("x", "v", "C", "F", "J", "ppos", "prot", "psize", "friction", "mu", "lamda", "action")]

    def step_fwd(self, st, nthreads=1):
        """st: dict x[B,N,3] v C[B,N,3,3] F J[B,N] ppos[B,(P,)steps,3] prot[B,(P,)steps,4] psize[B,(P,)3] friction mu lamda[B]
        action[B,6P]; the primitive axis P is present when n_prim > 1."""
        dt = st["x"].dtype
        a = self._prep(st, dt)
        B, N, S = a[0].shape[0], self.N, self.steps
        pa = (self.P,) if self.P > 1 else ()
        o = dict(x=np.empty((B, N, 3), dt), v=np.empty((B, N, 3), dt), C=np.empty((B, N, 3, 3), dt),
                 F=np.empty((B, N, 3, 3), dt), J=np.empty((B, N), dt), ppos=np.empty((B,) + pa + (S, 3), dt),
                 prot=np.empty((B,) + pa + (S, 4), dt), pv=np.empty((B,) + pa + (S, 3), dt), pw=np.empty((B,) + pa + (S, 3), dt))
        getattr(lib(), "oc_mpm_step_fwd_" + _suf(dt))(
            self.h, C.c_int(B), *[_p(q) for q in a],
            *[_p(o[k]) for k in ("x", "v", "C", "F", "J", "ppos", "prot", "pv", "pw")], C.c_int(nthreads))
        return o

    def step_bwd(self, st, g, clip=True, nthreads=1):
        """g: dict gx gv gC gF gppos [gprot] (cotangents of the step outputs)."""
        dt = st["x"].dtype
        a = self._prep(st, dt)
        B, N, S = a[0].shape[0], self.N, self.steps
        c = lambda q: np.ascontiguousarray(q, dtype=dt)
        pa = (self.P,) if self.P > 1 else ()
        gin = [c(g[k]) for k in ("gx", "gv", "gC", "gF", "gppos")] + [c(g["gprot"]) if "gprot" in g else np.zeros((B,) + pa + (S, 4), dt)]
        o = dict(gx=np.empty((B, N, 3), dt), gv=np.empty((B, N, 3), dt), gC=np.empty((B, N, 3, 3), dt),
                 gF=np.empty((B, N, 3, 3), dt), gppos=np.empty((B,) + pa + (S, 3), dt), gprot=np.empty((B,) + pa + (S, 4), dt),
                 gfriction=np.empty((B,), dt), gmu=np.empty((B,), dt), glamda=np.empty((B,), dt), gaction=np.empty((B, 6 * self.P), dt))
        getattr(lib(), "oc_mpm_step_bwd_" + _suf(dt))(
            self.h, C.c_int(B), *[_p(q) for q in a], *[_p(q) for q in gin], C.c_int(int(clip)),
            *[_p(o[k]) for k in ("gx", "gv", "gC", "gF", "gppos", "gprot", "gfriction", "gmu", "glamda", "gaction")],
            C.c_int(nthreads))
        return o


def mpm_det_forward(st, N, n_grid=64, res=(32, 32, 32), steps=70, dt=1e-4, p_rho=1.0, gravity=(0, -9.8, 0), material=None, hardness=None,
                    position_control=True, n_prim=1, sdf_kind=0, prim_friction=None, prim_softness=None):
    """The deterministic MPM forward's own source (unidom_amd/csrc/mpm_det.h) compiled for the CPU (csrc/mpm_det_host.cpp): the
    same-order, same-arithmetic checker of ud_mpm_conf.deterministic.  f32 only; position control with one box primitive, or
    (position_control=False) collide_batch of n_prim box (sdf_kind 0) / container (1) primitives with their own friction / softness.
    st as MpmOracle.step_fwd takes it (primitive arrays [B, P, ...] when P > 1); returns x v C F after the step and the primitive rows
    forward kinematics leaves."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    x, v, Cm, F = f(st["x"]), f(st["v"]), f(st["C"]), f(st["F"])
    B = x.shape[0]
    dx = 1 / n_grid
    p_vol = (dx * 0.5) ** 2
    mat = np.ascontiguousarray(np.full(N, 1) if material is None else material, dtype=np.int32)
    hd = f(np.full(N, 1.0) if hardness is None else hardness)
    ppos, prot = f(st["ppos"]).copy(), f(st["prot"]).copy()
    o = dict(x=np.empty_like(x), v=np.empty_like(v), C=np.empty_like(Cm), F=np.empty_like(F))
    rc = lib().oc_mpm_det_forward(
        C.c_int(B), C.c_int(N), C.c_int(n_grid), _p(np.ascontiguousarray(res, dtype=np.int32)), C.c_int(steps), C.c_float(dt),
        C.c_float(p_vol * p_rho), C.c_float(p_vol), _p(f(gravity)), _p(mat), _p(hd), _p(x), _p(v), _p(Cm), _p(F), _p(ppos), _p(prot),
        _p(f(st["psize"])), _p(f(st["friction"])), _p(f(st["mu"])), _p(f(st["lamda"])), _p(f(st["action"])),
        _p(o["x"]), _p(o["v"]), _p(o["C"]), _p(o["F"]), C.c_int(1 if position_control else 0), C.c_int(n_prim), C.c_int(sdf_kind),
        _p(f(np.zeros(4) if prim_friction is None else np.resize(np.asarray(prim_friction, np.float32), 4))),
        _p(f(np.zeros(4) if prim_softness is None else np.resize(np.asarray(prim_softness, np.float32), 4))))
    assert rc == 0
    o["ppos_rows"], o["prot_rows"] = ppos, prot
    return o


def svd3(A):
    A = np.ascontiguousarray(A)
    U, S, Vh = np.empty_like(A), np.empty(3, A.dtype), np.empty_like(A)
    getattr(lib(), "oc_svd3_" + _suf(A.dtype))(_p(A), _p(U), _p(S), _p(Vh))
    return U, S, Vh


class PlbOracle:
    """CPU restatement (f64, dense grid) of the Taichi PlasticineLab forward step (GenORM Torus, BASELINE config 5).
    PARITY UNPINNED (see oracle/csrc/plb_oracle.hpp)."""

    def __init__(self, N=1000, n_grid=64, substeps=19, dt=1e-4, gravity=(0, -0.4, 0), ground_friction=0.5, radius=(0.025, 0.025)):
        self.N, self.n_prim = N, len(radius)
        lib().oc_plb_create.restype = C.c_void_p
        g = np.ascontiguousarray(gravity, np.float64)
        r = np.ascontiguousarray(radius, np.float64)
        self.h = C.c_void_p(lib().oc_plb_create(C.c_int(N), C.c_int(n_grid), C.c_int(substeps), C.c_double(dt), _p(g),
                                               C.c_double(ground_friction), C.c_int(self.n_prim), _p(r)))

    def __del__(self):
        try:
            lib().oc_plb_destroy(self.h)
        except Exception:
            pass

    def step(self, x, v, Cm, F, prim_pos, softness, action, E, nu, ys, nthreads=1):
        c = lambda a: np.ascontiguousarray(a, np.float64)
        x, v, Cm, F, prim_pos, softness, action, E, nu, ys = map(c, (x, v, Cm, F, prim_pos, softness, action, E, nu, ys))
        B = x.shape[0]
        xo, vo, Co, Fo, po = np.empty_like(x), np.empty_like(v), np.empty_like(Cm), np.empty_like(F), np.empty_like(prim_pos)
        lib().oc_plb_step(self.h, C.c_int(B), _p(x), _p(v), _p(Cm), _p(F), _p(prim_pos), _p(softness), _p(action), _p(E), _p(nu),
                          _p(ys), _p(xo), _p(vo), _p(Co), _p(Fo), _p(po), C.c_int(nthreads))
        return dict(x=xo, v=vo, C=Co, F=Fo, prim_pos=po)
