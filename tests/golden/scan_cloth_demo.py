#!/usr/bin/env python3
"""Bounded search for the legacy cloth step that recorded expert_demo/fold_cloth1/demo_*.pkl.

Test infrastructure only.  The reference ships 17 fold_cloth1 transitions (state_k, action_k) -> state_k+1, one env.step_diff
= 40 robot_steps x 50 substeps each (tests/golden/fold_cloth1_demos.npz, re-packed as data by make_golden.py).  The CURRENT
cloth step (cloth_simulator.py:257-337) does not reproduce their x / v: the recorded resting v_y is -0.249415, the terminal
velocity of gravity applied ONCE per substep, while the current code applies it twice (:259 and :278) -> -0.499.  If some
variant of the current arithmetic reproduced the recordings, a "legacy" switch in the oracle would pin the spring / friction /
damping arithmetic it shares with the current code.  This script replays the transitions with a vectorised f32 NumPy
restatement of the step (same formulas and operation order as oracle/twin/cloth_twin.py, all transitions batched) under

    gravity at :259 only | at :278 only | both      x  mu in {0.5, 0.9}          x  stiffness in {900, 100 (the commented :270)}
    damping in {2, 1, 4}                            x  friction formula variants x  gripper: hard mask (3-D | ground-plane distance |
    2 x radius) | the commented soft weight (:213-215)
    substeps per robot_step in {50, 25, 100} (primitive path kept: action / substeps)

and scores every variant, FIRST on quantities that do not depend on the discrete grasp -- the resting v_y fixed point, the
share of grounded particles, the cloth centroid -- THEN on the full x.  Output: tests/golden/cloth_scan.csv (sorted by max|dx|
over the transitions; a variant "reproduces" at ~1e-5, not at 1e-2) and a summary on stdout.

    python tests/golden/scan_cloth_demo.py [--quick]
"""
import itertools
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def tables(N=80, size=16):
    """cloth_simulator.py:48-66 for the fold_cloth1 mask (fold_cloth1_env.py:48-53): neighbour ids (self where missing), 1/L0, weight"""
    m = np.zeros((N, N), bool)
    m[size * 2:size * 3, size * 2:size * 4] = True
    ii, jj = np.nonzero(m)
    idx = -np.ones((N, N), int)
    idx[ii, jj] = np.arange(len(ii))
    links = np.array([[-1, 0], [1, 0], [0, -1], [0, 1], [-1, -1], [1, -1], [-1, 1], [1, 1]])
    P = len(ii)
    nbr = np.zeros((P, 8), int)
    w = np.zeros((P, 8), F)
    L0 = np.zeros((P, 8), F)
    for l, (di, dj) in enumerate(links):
        ni, nj = np.clip(ii + di, 0, N - 1), np.clip(jj + dj, 0, N - 1)
        j = idx[ni, nj]
        ok = j >= 0
        nbr[:, l] = np.where(ok, j, np.arange(P))
        w[:, l] = ok
        L0[:, l] = F(1.0 / N) * F(np.sqrt(F(di * di + dj * dj)))
    return nbr, w, L0


def pnp_actions(action6, prim0):
    """cloth_env.py:134-173, batched: [n,6], [n,4] -> [40,n,8] (pinned exactly by the recordings, tests/test_oracle_cloth.py)"""
    n = len(action6)
    pick, place = action6[:, :3].copy(), action6[:, 3:].copy()
    pick[:, 1] = 0
    place[:, 1] = 0
    a = np.zeros((40, n, 8), F)
    a[:3, :, :3] = ((pick - prim0[:, :3]) * (F(1) / F(3)))[None]
    a[:3, :, 3] = 1
    a[3:13, :, 1] = F(0.06 / 10)
    mv = place - pick
    mv[:, 1] = 0
    a[13:33, :, :3] = (mv * (F(1) / F(20)))[None]
    a[33:, :, 3] = 1
    return a


class Variant:
    def __init__(self, grav="both", mu=None, k=900.0, damping=2.0, friction="current", grip="hard", substeps=50, dt=2e-3, g=0.5):
        self.grav, self.mu, self.k, self.damping, self.friction, self.grip, self.substeps, self.dt, self.g = grav, mu, k, damping, friction, grip, substeps, dt, g

    def key(self):
        return (self.grav, self.mu if self.mu is not None else "state", self.k, self.damping, self.friction, self.grip, self.substeps)


def rollout(var, x, v, prim0, prim1, mu_state, actions, nbr, w, L0, small=F(1e-8), max_v=F(2.0)):
    """one env.step_diff for all transitions at once (f32, the twin's operation order)"""
    n, P, _ = x.shape
    dt, g = F(var.dt), F(var.g)
    damp = F(np.exp(np.float64(F(-var.damping * var.dt))))
    k = F(var.k)
    mu = (mu_state if var.mu is None else np.full(n, var.mu, F)).astype(F)[:, None]
    x, v = x.copy(), v.copy()
    ps = [prim0.copy(), prim1.copy()]
    S = var.substeps
    r = F(1) / F(S)
    for t in range(actions.shape[0]):
        a = actions[t]
        acts = [np.concatenate([np.clip(a[:, :3], -2, 2) * r, a[:, 3:4]], 1), np.concatenate([np.clip(a[:, 4:7], -2, 2) * r, a[:, 7:8]], 1)]
        for _ in range(S):
            if var.grav in ("both", "v"):
                v[:, :, 1] -= g * dt                                                   # :259
            rel = x[:, nbr] - x[:, :, None, :]                                         # [n,P,8,3]   :261-262
            cur = np.sqrt(np.maximum((rel * rel).sum(-1, dtype=F), F(1e-12)))[..., None]   # :264
            force = (k * rel / cur * (cur - L0[None, :, :, None]) / L0[None, :, :, None] * w[None, :, :, None]).sum(2, dtype=F)   # :267-277
            if var.grav in ("both", "f"):
                force[:, :, 1] -= g                                                    # :278
            fy = force[:, :, 1] if var.grav != "v" or var.friction != "normal_g" else force[:, :, 1] - g
            ground = x[:, :, 1] <= small                                               # :281
            muF = mu * np.minimum(fy, 0) * F(-1)                                       # :282
            xV, yV = v[:, :, 0], v[:, :, 2]
            sV = np.sqrt(xV * xV + yV * yV + small)
            if var.friction in ("current", "normal_g", "dynamic_only"):
                dyn = (ground & (sV > small)).astype(F)
                force[:, :, 0] -= dyn * muF * xV / sV                                  # :289
                force[:, :, 2] -= dyn * muF * yV / sV                                  # :290
            if var.friction in ("current", "normal_g"):
                stat = ground & (sV <= small)
                xF, yF = force[:, :, 0].copy(), force[:, :, 2].copy()
                sF = np.sqrt(xF * xF + yF * yF + small)
                zero = (stat & (muF > sF)).astype(F)
                force[:, :, 0] = (F(1) - zero) * force[:, :, 0]
                force[:, :, 2] = (F(1) - zero) * force[:, :, 2]
                nz = (stat & (muF <= sF)).astype(F)
                R = F(1) - muF / sF
                force[:, :, 0] = (R * xF) * nz + force[:, :, 0] * (F(1) - nz)
                force[:, :, 2] = (R * yF) * nz + force[:, :, 2] * (F(1) - nz)
            v += force * dt                                                            # :308
            if var.friction == "velocity":       # Coulomb friction as a velocity decrement: |v_t| -= mu N dt, stopping at 0
                vt = np.sqrt(v[:, :, 0] ** 2 + v[:, :, 2] ** 2 + small)
                sc = np.where(ground, np.maximum(F(1) - muF * dt / vt, 0), F(1)).astype(F)
                v[:, :, 0] *= sc
                v[:, :, 2] *= sc
            v *= damp                                                                  # :309
            for gi in range(2):                                                        # :313-314, :198-226
                pos, rad = ps[gi][:, None, :3], ps[gi][:, None, 3]
                d_v, suction = acts[gi][:, None, :3], acts[gi][:, None, 3:4]
                dd = x - pos
                dist = np.sqrt(dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1] + dd[..., 2] * dd[..., 2])
                if var.grip in ("hard", "hard_xz", "hard_r2"):
                    if var.grip == "hard_xz":                                          # distance in the ground plane only
                        dist = np.sqrt(dd[..., 0] * dd[..., 0] + dd[..., 2] * dd[..., 2])
                    m = (dist <= (rad * F(2) if var.grip == "hard_r2" else rad))[..., None]
                    v = np.where(m, suction * v, v)
                    x = np.where(m, x + d_v * (F(1) - suction), x)
                else:                                                                  # the commented soft gripper (:213-215)
                    wt = np.exp(F(-1) * (dist * F(20) - F(1)))[..., None].astype(F)
                    if var.grip == "soft_capped":
                        wt = np.minimum(wt, F(1))
                    v = v - wt * suction * v
                    x = x + d_v * wt
            for gi in range(2):
                ps[gi][:, :3] = np.clip(ps[gi][:, :3] + acts[gi][:, :3], 0, 1)         # :322-323
            x = np.clip(x, 0, 1)                                                       # :326
            v = np.clip(v, -max_v, max_v)                                              # :327
            x = x + dt * v                                                             # :329
    return x, v, ps[0], ps[1]


def score(x, v, ps0, d):
    x1, v1 = d["s1_x"], d["s1_v"]
    err = np.abs(x - x1).reshape(len(x), -1).max(1)
    moved = np.abs(x1 - d["s0_x"]).reshape(len(x), -1).max(1)
    rest_gold = F(-0.24941486)
    # grasp-independent statistics
    vy = v[:, :, 1]
    near = np.abs(vy - np.median(vy, axis=1, keepdims=True)) < 1e-5
    rest = np.array([np.median(vy[i][x[i, :, 1] <= 0]) if (x[i, :, 1] <= 0).any() else np.nan for i in range(len(x))])
    grounded = (x[:, :, 1] <= 0).mean(1)
    grounded_gold = (x1[:, :, 1] <= 0).mean(1)
    cen = np.abs(x.mean(1) - x1.mean(1)).max(1)
    return dict(max_dx=float(err.max()), med_dx=float(np.median(err)), rel=float((err / moved).max()), rest_vy=float(np.nanmedian(rest)),
                rest_err=float(abs(np.nanmedian(rest) - rest_gold)), grounded_err=float(np.abs(grounded - grounded_gold).max()),
                centroid_err=float(cen.max()), prim_err=float(np.abs(ps0 - d["s1_primitive0"]).max()))


def main():
    quick = "--quick" in sys.argv
    d = dict(np.load(os.path.join(HERE, "fold_cloth1_demos.npz")))
    sel = np.arange(len(d["demo"])) if not quick else np.array([0, 1, 2])
    d = {k: v[sel] for k, v in d.items()}
    nbr, w, L0 = tables()
    acts = pnp_actions(d["action"].astype(F), d["s0_primitive0"])
    base = (d["s0_x"].astype(F), d["s0_v"].astype(F), d["s0_primitive0"].astype(F), d["s0_primitive1"].astype(F), d["s0_mu"].astype(F), acts, nbr, w, L0)
    # sanity of the restatement: the current step must match the C++ oracle (reference operation order) on one transition
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    try:
        from oracle.pyoracle import ClothOracle
        m = np.zeros((80, 80), F)
        m[32:48, 32:64] = 1
        o = ClothOracle(m).rollout_fwd(d["s0_x"][:1], d["s0_v"][:1], np.stack([d["s0_primitive0"][:1], d["s0_primitive1"][:1]], 1),
                                       np.array([900], F), d["s0_mu"][:1].astype(F), acts[:, :1])
        cur = Variant()
        xs, vs, p0, _ = rollout(cur, *(a[:1] if isinstance(a, np.ndarray) and a.shape[0] == len(sel) else a for a in base[:5]), acts[:, :1], nbr, w, L0)
        print("restatement vs C++ oracle on transition 0 (2000 substeps, chaotic: same formulas, NumPy reduction order): max|dx| %.2e" % np.abs(xs - o["x"]).max(), flush=True)
        print("                         after 3 robot_steps: max|dx| %.2e" % np.abs(
            rollout(cur, d["s0_x"][:1], d["s0_v"][:1], d["s0_primitive0"][:1], d["s0_primitive1"][:1], d["s0_mu"][:1], acts[:3, :1], nbr, w, L0)[0]
            - ClothOracle(m).rollout_fwd(d["s0_x"][:1], d["s0_v"][:1], np.stack([d["s0_primitive0"][:1], d["s0_primitive1"][:1]], 1),
                                         np.array([900], F), d["s0_mu"][:1].astype(F), np.ascontiguousarray(acts[:3, :1]))["x"]).max(), flush=True)
    except Exception as e:   # the scan itself needs no oracle
        print("oracle cross-check skipped:", type(e).__name__, e)
    gravs = ("both", "v", "f")
    mus = (None, 0.5)                     # the recorded state says 0.9; conf.mu is 0.5
    ks = (900.0, 100.0)
    variants = []
    for grav, mu, k in itertools.product(gravs, mus, ks):
        variants.append(Variant(grav=grav, mu=mu, k=k))
    for grav in ("v", "f"):
        for fr in ("normal_g", "dynamic_only", "none", "velocity"):
            variants.append(Variant(grav=grav, friction=fr))
            variants.append(Variant(grav=grav, friction=fr, mu=0.5))
        for grip in ("soft", "soft_capped", "hard_xz", "hard_r2"):
            variants.append(Variant(grav=grav, grip=grip))
            variants.append(Variant(grav=grav, grip=grip, k=100.0))
        for damping in (1.0, 4.0):
            variants.append(Variant(grav=grav, damping=damping))
        for S in (25, 100):
            variants.append(Variant(grav=grav, substeps=S))
    if quick:
        variants = variants[:6]
    rows = []
    t0 = time.time()
    for i, var in enumerate(variants):
        x, v, p0, p1 = rollout(var, *base)
        s = score(x, v, p0, d)
        rows.append((var.key(), s))
        print("[%3d/%d %5.0fs] grav %-4s mu %-5s k %-5g damp %-3g fric %-12s grip %-11s S %-3d : max|dx| %.2e med %.2e  rest v_y %.5f (gold -0.24941)  "
              "grounded err %.2f  centroid err %.2e  prim err %.1e" % ((i + 1, len(variants), time.time() - t0) + var.key() + (
                  s["max_dx"], s["med_dx"], s["rest_vy"], s["grounded_err"], s["centroid_err"], s["prim_err"])), flush=True)
    rows.sort(key=lambda r: r[1]["max_dx"])
    out = os.path.join(HERE, "cloth_scan.csv")
    with open(out, "w") as f:
        f.write("# expert_demo/fold_cloth1/demo_*.pkl: %d transitions replayed by a NumPy f32 restatement of cloth_simulator.py:257-337 under legacy hypotheses, sorted by max|dx|\n" % len(sel))
        f.write("# gold: resting v_y = -0.24941486; the rope of numbers to beat is ~1e-5 (f32 round-off over 2000 substeps is ~1e-2 for this chaotic system only AFTER the grasp sets diverge)\n")
        f.write("gravity,mu,stiffness,damping,friction,gripper,substeps,max_abs_dx,median_abs_dx,max_dx_over_moved,rest_vy,rest_vy_err,grounded_share_err,centroid_err,primitive_err\n")
        for key, s in rows:
            f.write("%s,%s,%g,%g,%s,%s,%d,%.3e,%.3e,%.3e,%.6f,%.2e,%.3f,%.3e,%.1e\n" % (key + (
                s["max_dx"], s["med_dx"], s["rel"], s["rest_vy"], s["rest_err"], s["grounded_err"], s["centroid_err"], s["prim_err"])))
    print("best:", rows[0][0], rows[0][1], "->", out)


if __name__ == "__main__":
    main()
