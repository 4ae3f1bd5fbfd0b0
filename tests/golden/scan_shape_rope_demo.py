#!/usr/bin/env python3
"""Bounded search for the legacy configuration that recorded expert_demo/shape_rope/demo_*.pkl (third bullet).

Test infrastructure only.  Those recordings exercise soft contact (collide_batch) and the plastic branch of the MPM substep; if
a configuration of the CURRENT arithmetic reproduced them they would pin both the way whip_rope/demo_0.pkl pins position
control.  What the data fixes by itself (read here as data, stub Unpickler, no reference code executed): steps = 16 (primitive
arrays [16,3]), 20 sub-actions per env.step (primitive.v = action_buffer / 16 and action_buffer = (end - start) / 20), gripper
size (0.02, 0.06, 0.02), recorded state.friction = 1.0, mu / lamda of E = 100, nu = 0.1.  Unknown: dt, the grid (n_grid, res),
which nu / ground friction the recorder really used (whip_rope's recorder ignored the state fields), the material branch.
This script replays recorded transitions with the CPU oracle (the restatement of the current code) over a grid of those
unknowns and writes the table to tests/golden/shape_rope_scan.csv:  max |dx| against the recorded next state, next to the
distance the rope actually moved in that step (a configuration "reproduces" at ~1e-5, not at 1e-3).

    python tests/golden/scan_shape_rope_demo.py [--quick]
"""
import glob
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg                     # noqa: E402  (the stub Unpickler)
from oracle.pyoracle import MpmOracle        # noqa: E402

F32 = np.float32
SUB_ACTIONS, STEPS = 20, 16


def transition(d, k):
    s0, s1 = d["state"][k], d["state"][k + 1]
    g = lambda s, f: np.asarray(getattr(s, f))[0].astype(F32)
    p = s0.primitives[0]
    return dict(x=g(s0, "x"), v=g(s0, "v"), C=g(s0, "C"), F=g(s0, "F"), J=g(s0, "J"), action=np.asarray(d["action"][k])[0].astype(F32),
                psize=np.asarray(p.size)[0].astype(F32), x1=g(s1, "x"), v1=g(s1, "v"),
                ppos1=np.asarray(s1.primitives[0].position)[0, 0].astype(F32))


def replay(tr, dt, nu, fric, n_grid, res, material, threads=2):
    """One legacy env.step: focus shift (mpm_env.py:99-114), get_primitive_actions with 20 sub-actions
    (shape_rope_env.py:93-121), 20 x simulator.step of 16 substeps in soft-contact mode, un-shift (:116-125)."""
    N = tr["x"].shape[0]
    x0 = tr["x"][None]
    shift = (np.array(res, F32) * F32(0.5) / F32(n_grid) - x0.mean(1, dtype=F32)).astype(F32)
    shift[:, 1] = 0
    start, end = tr["action"][None, :3] + shift, tr["action"][None, 3:] + shift
    start[:, 1] = end[:, 1] = 0.01
    nrm = np.linalg.norm(end - start, axis=-1, keepdims=True).astype(F32) + F32(1e-8)
    end = start + (end - start) / nrm * np.clip(nrm, 0, 0.3)
    push = ((end - start) * (F32(1) / F32(SUB_ACTIONS))).astype(F32)
    push[:, 1] = 0
    E = 100.0
    mu0, la0 = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    ppos = np.zeros((1, STEPS, 3), F32)
    ppos[:, :] = start[:, None]
    prot = np.zeros((1, STEPS, 4), F32)
    prot[..., 0] = 1
    st = dict(x=x0 + shift[:, None], v=tr["v"][None], C=tr["C"][None], F=tr["F"][None], J=tr["J"][None], ppos=ppos, prot=prot,
              psize=tr["psize"][None], friction=np.full(1, fric, F32), mu=np.full(1, mu0, F32), lamda=np.full(1, la0, F32),
              action=np.concatenate([push, np.zeros_like(push)], -1))
    orc = MpmOracle(N, n_grid=n_grid, res=res, steps=STEPS, dt=dt, position_control=False, material=np.full(N, material),
                    prim_friction=0.1, prim_softness=666.0)
    for _ in range(SUB_ACTIONS):
        o = orc.step_fwd(st, nthreads=threads)
        st.update(x=o["x"], v=o["v"], C=o["C"], F=o["F"], J=o["J"], ppos=o["ppos"], prot=o["prot"])
        if not np.isfinite(o["x"]).all():
            return np.inf, np.inf, np.inf
    x = st["x"][0] - shift
    ex = float(np.abs(x - tr["x1"]).max())
    ev = float(np.abs(st["v"][0] - tr["v1"]).max() / (np.abs(tr["v1"]).max() + 1e-30))
    ep = float(np.abs(st["ppos"][0, 0] - shift[0] - tr["ppos1"]).max())
    return ex, ev, ep


def main():
    quick = "--quick" in sys.argv
    demos = sorted(glob.glob(f"{mg.REF}/algorithms/expert_demo/shape_rope/demo_*.pkl"))
    d = mg.load(demos[0])
    trs = [transition(d, k) for k in ((0,) if quick else (0, 2))]
    grids = [(128, (64, 6, 64)), (64, (32, 6, 32)), (64, (32, 32, 32))]
    dts = [0.5e-4, 1e-4, 1.5e-4, 2e-4, 3e-4, 6.25e-4] if not quick else [1e-4, 3e-4]
    rows = []
    for (n_grid, res), dt, nu, fric, material in itertools.product(grids, dts, (0.1, 0.2), (0.1, 0.9, 1.0), (2, 1)):
        errs = [replay(tr, dt, nu, fric, n_grid, res, material) for tr in trs]
        moved = [float(np.abs(tr["x1"] - tr["x"]).max()) for tr in trs]
        rows.append((n_grid, "x".join(map(str, res)), dt, nu, fric, material, max(e[0] for e in errs), max(e[1] for e in errs),
                     max(e[2] for e in errs), max(moved)))
        print("n_grid %3d res %-8s dt %.2e nu %.1f fric %.1f mat %d : max|dx| %.2e  rel dv %.2e  prim %.1e  (rope moved %.2e)" % rows[-1], flush=True)
    rows.sort(key=lambda r: r[6])
    out = os.path.join(HERE, "shape_rope_scan.csv")
    with open(out, "w") as f:
        f.write("# expert_demo/shape_rope/demo_0.pkl, transitions %s replayed by the CPU oracle (current arithmetic, soft contact), sorted by max|dx|\n"
                % ("0" if quick else "0 and 2"))
        f.write("n_grid,res,dt,nu,ground_friction,material,max_abs_dx,rel_dv,primitive_abs_err,rope_moved\n")
        for r in rows:
            f.write("%d,%s,%.3g,%.1f,%.1f,%d,%.3e,%.3e,%.1e,%.3e\n" % r)
    print("best:", rows[0], "->", out)


if __name__ == "__main__":
    main()
