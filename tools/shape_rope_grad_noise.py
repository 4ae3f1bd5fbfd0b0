#!/usr/bin/env python3
"""shape_rope's env-level gradient against the number of scanned simulator.steps.

`MPMEnv.step_diff` (fused focus shift + tail) and `step_diff_unfused` (the same arithmetic op by op) hand the simulator shifts
that differ in their last bit; DESIGN.md 6 argues that 30 x 133 plastic, contacting substeps amplify that -- and a one-ulp move
of the cloud -- to an O(1) change of the adjoint.  This script MEASURES it: for T = 1 ... 30 scanned steps of 133 substeps
(the push per step is the default env's: 0.1 m over 30 steps) it prints, per differentiated leaf, the relative difference
    fused vs op-by-op        (max|g_f - g_u| / max(max|g_u|, 1e-3 x the largest leaf's))
    op-by-op vs op-by-op with the cloud moved by one ulp (2^-24)
    op-by-op vs op-by-op again, same inputs (run-to-run: the scatter order of the float atomics)
so that the growth from rounding noise with the horizon is a table, not an argument.  tests/test_envs_gpu.py compares the
gradients at the longest horizon where the noise stays below 5 %.

    python tools/shape_rope_grad_noise.py [--envs 3] > profiles/r03_shape_rope_grad_noise.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_env(T, B, device="cuda"):
    """shape_rope scanning T simulator.steps per env.step (the conf's own 133 substeps each)"""
    from unidom_amd.envs import shape_rope_env as sre
    sre.DefaultConf.primitive_action_steps = T          # get_primitive_actions reads the class attribute (like the reference)
    env = sre.ShapeRopeEnv(batch_size=B, seed=1, device=device)
    env.build_reset_state()
    st = env.state
    # The reset lattice IS the goal cloud: a particle that has not moved sits exactly on its goal point, where the reward's
    # sqrt(mean((x - goal)^2)) (util.py:156-159) has the gradient 0 * inf = NaN -- in the reference too.  The env's own reset
    # pushes the rope around first (random_push); here a seeded 0.2 mm jitter does, so that short horizons are measurable.
    jit = torch.randn(st.x.shape, generator=torch.Generator().manual_seed(11)) * 2e-4
    return env, st._replace(x=st.x + jit.to(st.x.device))


def push_action(st, T, device):
    """start 5 mm from the rope's middle particle, moving into it at the default env's pace (0.1 m per 30 steps)"""
    mid = st.x[:, st.x.shape[1] // 2].cpu()
    a = torch.cat([mid + torch.tensor([-0.004, 0.0, -0.02]), mid + torch.tensor([-0.004 + 0.014 * T / 30.0, 0.0, -0.02 + 0.1 * T / 30.0])], -1)
    a[:, [1, 4]] = 0
    return a.to(device)


def grads(env, st, act, fn, nudge=0.0, seed=3):
    g = torch.Generator(device=env.device).manual_seed(seed)
    a = act.clone().requires_grad_(True)
    leaves = {k: (getattr(st, k) + (nudge if k == "x" else 0.0)).clone().requires_grad_(True) for k in ("x", "v", "C", "F")}
    pos = [p.position.clone().requires_grad_(True) for p in st.primitives]
    s = st._replace(primitives=[p._replace(position=q) for p, q in zip(st.primitives, pos)], **leaves)
    obs, reward, done, info = fn(a, s)
    ns = info["state"]
    outs = {"reward": reward, "obs": obs, "x": ns.x, "v": ns.v, "C": ns.C, "F": ns.F}
    loss = sum((t * torch.randn(t.shape, device=env.device, generator=g)).sum() * (1.0 if k == "reward" else 1e-3) for k, t in outs.items())
    loss.backward()
    return {"a": a.grad, **{k: t.grad for k, t in leaves.items()}, "pos0": pos[0].grad}, {k: t.detach() for k, t in outs.items()}


def rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def measure(T, B):
    env, st = make_env(T, B)
    act = push_action(st, T, env.device)
    gf, of = grads(env, st, act, env.step_diff)
    gu, ou = grads(env, st, act, env.step_diff_unfused)
    gr, _ = grads(env, st, act, env.step_diff_unfused, nudge=2.0 ** -24)
    g2, _ = grads(env, st, act, env.step_diff_unfused)
    env.simulator.check_status()
    gmax = max(float(t.abs().max()) for t in gu.values())       # a leaf whose whole cotangent is tiny is judged on the adjoint's scale
    sc = {k: max(float(gu[k].abs().max()), 1e-3 * gmax) + 1e-30 for k in gu}
    rows = {k: tuple(float((g[k] - gu[k]).abs().max()) / sc[k] for g in (gf, gr, g2)) for k in gu}
    vals = {k: rel(of[k], ou[k]) for k in ("x", "v", "reward")}
    return rows, vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=3)
    ap.add_argument("--steps", type=int, nargs="*", default=[1, 2, 4, 8, 12, 16, 20, 30])
    args = ap.parse_args()
    print("# shape_rope: relative gradient differences vs scanned simulator.steps (133 substeps each), %d envs" % args.envs)
    print("# columns per leaf: fused-vs-unfused | one-ulp nudge of the cloud | same inputs again (atomics order)")
    print("%4s %8s  %s" % ("T", "substeps", "  ".join("%-26s" % k for k in ("a", "x", "v", "C", "F", "pos0"))))
    for T in args.steps:
        rows, vals = measure(T, args.envs)
        print("%4d %8d  %s   values: x %.1e v %.1e reward %.1e" % (
            T, T * 133, "  ".join("%.1e %.1e %.1e" % rows[k] for k in ("a", "x", "v", "C", "F", "pos0")), vals["x"], vals["v"], vals["reward"]),
            flush=True)


if __name__ == "__main__":
    main()
