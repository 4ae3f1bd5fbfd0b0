#!/usr/bin/env python3
"""Diagnostic: where does the NaN in shape_rope's short-horizon env gradient come from?  (tools/shape_rope_grad_noise.py, T <= 4)
Runs one env.step_diff with T scanned steps under torch's anomaly mode (names the first backward node that returns NaN), then
the simulator step alone with unit cotangents, clip on and off, printing finite / NaN / inf counts per output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shape_rope_grad_noise as srg   # noqa: E402


def census(name, t):
    t = t.detach().float()
    print(f"  {name:10s} shape {tuple(t.shape)} nan {int(torch.isnan(t).sum())} inf {int(torch.isinf(t).sum())} absmax(finite) "
          f"{float(t[torch.isfinite(t)].abs().max()) if torch.isfinite(t).any() else float('nan'):.3e}")


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    env, st = srg.make_env(T, 3)
    act = srg.push_action(st, T, env.device)
    print("T", T, "action", act[0].tolist(), "rope mid", st.x[0, st.x.shape[1] // 2].tolist(), "prim0 pos", st.primitives[0].position[0, 0].tolist())
    for fn_name in ("step_diff", "step_diff_unfused"):
        print(fn_name)
        try:
            with torch.autograd.set_detect_anomaly(True):
                g, o = srg.grads(env, st, act, getattr(env, fn_name))
            for k, t in g.items():
                census("grad " + k, t)
        except RuntimeError as e:
            print("  anomaly:", str(e).splitlines()[0][:300])
    # simulator.step alone
    from unidom_amd.engine.mpm_simulator import _Step
    sim = env.simulator
    p = st.primitives[0]
    for clip in (True, False):
        sim.clip_grad = clip
        leaves = [t.clone().requires_grad_(True) for t in (st.x, st.v, st.C, st.F)]
        pos = p.position.clone().requires_grad_(True)
        a = torch.zeros((3, 6), device=env.device)
        a[:, 2] = 0.1 / 30
        a.requires_grad_(True)
        out = _Step.apply(sim, *leaves, st.J, pos, p.rotation, p.size, st.friction, st.mu, st.lamda, a)
        (out[0].sum() + out[1].sum()).backward()
        print("simulator.step alone, clip", clip)
        for k, t in zip(("x", "v", "C", "F", "pos", "a"), (*leaves, pos, a)):
            census("grad " + k, t.grad)
    sim.clip_grad = True
    sim.check_status()


if __name__ == "__main__":
    main()
