"""diagnostic: the PLB adjoint at bench shape (N=1000, n_grid 64, 8 envs); which envs / leaves are non-finite, under which lane mapping / checkpoint mode / batch"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.twin.plb_twin import torus_particles
from test_plb import _hip_sim


def case(B, N=1000, seed=21):
    rng = np.random.default_rng(seed)
    x = torus_particles(1000)[None].repeat(B, 0) + rng.normal(size=(B, N, 3)) * 1e-4
    v = rng.normal(size=(B, N, 3)) * 0.05
    Cm = rng.normal(size=(B, N, 3, 3)) * 0.5
    F = np.eye(3)[None, None] + rng.normal(size=(B, N, 3, 3)) * 0.02
    prim = np.stack([x[:, 7], np.repeat(np.array([[0.5, 0.55, 0.5]]), B, 0)], 1)
    soft = np.full((B, 2), 666.0)
    act = rng.uniform(-0.01, 0.01, size=(B, 3)) * np.array([1.0, 0.3, 1.0])
    E = rng.uniform(3e3, 6e3, size=B)
    nu = rng.uniform(0.25, 0.4, size=B)
    ys = np.array([1762.2, 30.0, 1762.2, 200.0, 1762.2, 50.0, 1762.2, 30.0])[:B]
    w = [rng.normal(size=s) for s in ((B, N, 3), (B, N, 3), (B, N, 3, 3), (B, N, 3, 3), (B, 2, 3))]
    return x, v, Cm, F, prim, soft, act, E, nu, ys, w


def run(B, K=None, sel=None, steps=1):
    x, v, Cm, F, prim, soft, act, E, nu, ys, w = case(8)
    if sel is not None:
        x, v, Cm, F, prim, soft, act, E, nu, ys = (np.ascontiguousarray(a[sel]) for a in (x, v, Cm, F, prim, soft, act, E, nu, ys))
        w = [np.ascontiguousarray(a[sel]) for a in w]
    B = x.shape[0]
    sim = _hip_sim(1000, B, quality=1.0, grid_ckpt_cells=K)
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
    hl = dict(x=T(x), v=T(v), C=T(Cm), F=T(F), prim=T(prim), act=T(act), E=T(E), nu=T(nu), ys=T(ys))
    s = sim.reset()._replace(x=hl["x"], v=hl["v"], C=hl["C"], F=hl["F"], prim_pos=hl["prim"], softness=T(soft, False), E=hl["E"], nu=hl["nu"], yield_stress=hl["ys"])
    for _ in range(steps):
        s = sim.step(s, hl["act"])
    sum((t * T(wi, False)).sum() for t, wi in zip((s.x, s.v, s.C, s.F, s.prim_pos), w)).backward()
    out = {}
    for k, t in hl.items():
        g = t.grad.cpu().numpy()
        out[k] = g
    fin = {k: [int((~np.isfinite(g[b])).sum()) for b in range(B)] for k, g in out.items()}
    return out, fin, {k: t.detach().cpu().numpy() for k, t in (("x", s.x), ("v", s.v), ("F", s.F))}


ref = None
for label, env, kw in (("default 8 envs", {}, {}), ("lanes 4", {"UD_PLB_LANES": "4"}, {}), ("lanes 1", {"UD_PLB_LANES": "1"}, {}),
                       ("K=0 (recompute)", {}, {"K": 0}), ("envs [0,1] alone", {}, {"sel": [0, 1]}), ("envs [2,7] alone", {}, {"sel": [2, 7]}),
                       ("default again", {}, {})):
    for k in ("UD_PLB_LANES",):
        os.environ.pop(k, None)
    os.environ.update(env)
    out, fin, st = run(8, **kw)
    print(label, "non-finite per env:", {k: v for k, v in fin.items() if any(v)}, "state finite", {k: bool(np.isfinite(a).all()) for k, a in st.items()}, flush=True)
    if ref is None:
        ref = out
    elif "sel" not in kw:
        print("   max rel diff vs first run:", {k: float(np.nanmax(np.abs(out[k] - ref[k])) / (np.nanmax(np.abs(ref[k])) + 1e-300)) for k in out})
