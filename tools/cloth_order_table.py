"""Operation order 2 ("v2", the default GPU forward) against operation order 1 (the reference's literal order,
cloth_simulator.py:257-337) over one whole step_diff (40 macro actions x 50 substeps), on the CPU restatements of both
(oracle/csrc/cloth_oracle.hpp; each is what the matching GPU forward equals bit for bit).  Per env: max-norm relative
difference of x and v after 100 / 500 / 2000 substeps and the first substep at which the two orders' grasp sets differ.
Checker-side tool (imports the oracle); writes the table to stdout:  python tools/cloth_order_table.py > profiles/r05_cloth_order_table.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import fold_cloth1_mask, make_cloth_case  # noqa: E402
from oracle.pyoracle import ClothOracle  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def main():
    B, T, S = 8, 40, 50
    o1, o2 = ClothOracle(fold_cloth1_mask(), order=1), ClothOracle(fold_cloth1_mask(), order=2)
    print("# fold_cloth1 (P = 512, k = 900, dt = 2e-3), one step_diff = 40 x 50 substeps, f32; order 2 (v2) vs order 1 (reference, literal)")
    print("# inputs: tests/conftest.make_cloth_case(default_rng(seed), B=8, T=40, deform=5e-4, v_scale=0.01), actions x 0.2 -- the case of the GPU parity tests")
    print("# rel = max|a - b| / max|b| over the env's particles; 'first grasp diff' = first substep (0-based, of 2000) whose grasp set (either gripper) differs, '-' = none")
    print("seed env | rel dx @100   rel dv @100 | rel dx @500   rel dv @500 | rel dx @2000  rel dv @2000 | first grasp diff | grasped particle-substeps (order 1)")
    worst = {k: 0.0 for k in ("x100", "v100", "x500", "v500", "x2000", "v2000")}
    first_all = []
    for seed in (7, 8, 11):
        rng = np.random.default_rng(seed)
        x, v, prim, k, mu, actions = make_cloth_case(rng, B, T, deform=0.0005, v_scale=0.01)
        actions *= 0.2
        a = o1.rollout_fwd(x, v, prim, k, mu, actions, want_lists=True, want_grasp=True, nthreads=8)
        b = o2.rollout_fwd(x, v, prim, k, mu, actions, want_lists=True, want_grasp=True, nthreads=8)
        ga = a["grasp"].reshape(T * S, B, -1)
        gb = b["grasp"].reshape(T * S, B, -1)
        for e in range(B):
            row = []
            for n in (100, 500, 2000):
                t = n // S - 1
                dx, dv = rel(b["x_list"][t, e], a["x_list"][t, e]), rel(b["v_list"][t, e], a["v_list"][t, e])
                worst[f"x{n}"] = max(worst[f"x{n}"], dx)
                worst[f"v{n}"] = max(worst[f"v{n}"], dv)
                row.append(f"{dx:.2e}      {dv:.2e}")
            diff = np.nonzero((ga[:, e] != gb[:, e]).any(-1))[0]
            first = int(diff[0]) if diff.size else None
            first_all.append(first)
            print(f"{seed:4d} {e:3d} | " + " | ".join(row) + f" | {'-' if first is None else first:>16} | {int(ga[:, e].sum())}")
    print("worst    | " + " | ".join(f"{worst['x%d' % n]:.2e}      {worst['v%d' % n]:.2e}" for n in (100, 500, 2000)))
    hit = [f for f in first_all if f is not None]
    print(f"# envs whose grasp sets differ somewhere in the 2000 substeps: {len(hit)} of {len(first_all)}" + (f"; earliest at substep {min(hit)}" if hit else ""))
    print(f"# north_star's 1e-4 on velocities is exceeded between the two IEEE orders from 100 substeps on (worst rel dv @100 = {worst['v100']:.1e}); positions stay "
          f"within {worst['x2000']:.1e} over the whole step_diff")


if __name__ == "__main__":
    main()
