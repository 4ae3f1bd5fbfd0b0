#!/usr/bin/env python3
"""Where does HIP-graph capture of an APG update break?  Runs growing slices of the whip_rope update under torch.cuda.graph, each in
its own child process (a crash in one does not hide the others), and prints which ones capture and replay.
    python tools/graph_probe.py            (parent: starts the children, touches no GPU itself)
"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGES = ["update_one_stream", "learner_capture"]


def child(stage):
    faulthandler.enable()
    import torch
    from unidom_amd.algorithms.apg.core import APG
    from unidom_amd.envs.registration import env_functions
    from unidom_amd.utils import prng
    dev = torch.device("cuda", 0)
    work = torch.cuda.Stream(dev)
    torch.cuda.set_stream(work)                      # everything below, eager updates included, on ONE non-default stream
    env = env_functions["whip_rope"](batch_size=32, seed=0, aux_reward=True, device=dev)
    _, state = env.reset(prng.PRNGKey(0))
    learner = APG(env, 3, learning_rate=1e-4, max_gradient_norm=0.3, seed=0)
    for _ in range(2):
        learner.minimize(state)
    env.simulator.check_status()
    if stage == "learner_capture":
        learner.capture(state)
        for _ in range(3):
            out = learner.minimize_captured()
        torch.cuda.synchronize(dev)
        print(stage, "replayed", float(out["loss"]), float(out["grad_norm"]), flush=True)
        eager = APG(env_functions["whip_rope"](batch_size=32, seed=0, aux_reward=True, device=dev), 3, learning_rate=1e-4, max_gradient_norm=0.3, seed=0)
        for _ in range(5):
            m = eager.minimize(state)
        print(stage, "eager after as many updates", float(m["loss"]), float(m["grad_norm"]), flush=True)
        return
    stage = "update"
    noise = torch.zeros((3, 32, env.action_size), device=dev)
    act = torch.zeros((32, env.action_size), device=dev, requires_grad=True)

    if stage.endswith("_rocblas"):
        torch.backends.cuda.preferred_blas_library("cublas")      # = rocBLAS on ROCm (the default picks hipBLASLt for these shapes)
    stage = stage.replace("_rocblas", "")

    def body():
        if stage == "policy_fwd":
            with torch.no_grad():
                return learner.policy(learner.get_obs(state)).square().mean()
        if stage == "policy_fwd_bwd":
            learner.sync.zero_grad()
            out = learner.policy(learner.get_obs(state)).square().mean()
            out.backward()
            return out.detach()
        if stage == "policy_grad":
            out = learner.policy(learner.get_obs(state)).square().mean()
            gs = torch.autograd.grad(out, learner.params)
            return sum(g.abs().sum() for g in gs)
        if stage == "adam_only":
            learner.flat_grad.fill_(1e-3)
            learner.sync.step()
            return learner.flat_grad.sum()
        if stage == "policy_adam":
            learner.sync.zero_grad()
            out = learner.policy(learner.get_obs(state)).square().mean()
            out.backward()
            learner.sync.step()
            return out.detach()
        if stage == "sim_fwd":
            with torch.no_grad():
                _, r, _, _ = env.step_diff(act, state)
            return r
        if stage == "sim_fwd_bwd":
            act.grad = None
            _, r, _, _ = env.step_diff(act, state)
            r.sum().backward()
            return act.grad
        if stage == "update_no_adam":
            learner.sync.zero_grad()
            loss, _ = learner.loss(state, None, noise)
            loss.backward()
            return loss.detach()
        return learner._update(state, noise)["loss"]

    if stage in ("policy_adam", "update", "adam_only"):
        sd = learner.optimizer.state_dict()
        learner.optimizer = learner.sync.optimizer = torch.optim.Adam(learner.params, lr=1e-4, capturable=True)
        learner.optimizer.load_state_dict({**sd, "param_groups": [{**g, "capturable": True} for g in sd["param_groups"]]})
    for _ in range(3):
        body()
    torch.cuda.synchronize(dev)
    print(stage, "warm-up ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=work):
        out = body()
    print(stage, "captured", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(dev)
    print(stage, "replayed", float(out.float().abs().sum()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for st in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), st], capture_output=True, text=True, timeout=300)
            tail = (r.stdout + r.stderr).strip().splitlines()
            keep = [l for l in tail if l.startswith(st.replace('_rocblas', '')) or "Error" in l or "error" in l or "File" in l or "Fatal" in l][-14:]
            print(f"== {st}: rc={r.returncode}")
            print("\n".join("   " + l for l in keep), flush=True)
