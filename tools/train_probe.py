"""Does APG learn on top of the kernels?  A short training run per env (mean reward of the first vs the last iterations).
Functional evidence only -- no reference learning curve exists to compare with.
usage (GPU box): PYTHONPATH=$PWD python tools/train_probe.py"""
import time

import numpy as np
import torch

from unidom_amd.algorithms.apg.core import APG
from unidom_amd.envs.registration import env_functions

import os
import sys

LONG = "--long" in sys.argv      # one long run on the headline env, curve written to gpurun_out/train_fold_cloth1.csv
RUNS = (("fold_cloth1", 4, 3, 1000, 1e-4),) if LONG else (("fold_cloth1", 4, 3, 150, 1e-4), ("whip_rope", 32, 3, 150, 1e-4), ("pour_water", 8, 3, 60, 1e-4),
                                                                 ("shape_rope", 4, 2, 20, 1e-4), ("fold_tshirt", 4, 2, 40, 1e-4))
for name, num_envs, ep_len, iters, lr in RUNS:
    torch.manual_seed(0)
    np.random.seed(0)
    env = env_functions[name](batch_size=num_envs, seed=0, aux_reward=True)
    _, st = env.reset(np.array([0, 3], np.uint32))
    learner = APG(env, ep_len, learning_rate=lr, max_gradient_norm=0.3, seed=0)
    rewards, t0 = [], time.time()
    for it in range(iters):
        m = learner.minimize(st)
        rewards.append(float(m["reward"].mean()))
        assert np.isfinite(rewards[-1]) and bool(torch.isfinite(m["grad_norm"]))
    if LONG:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/train_fold_cloth1.csv", "w") as f:
            f.write("iteration,mean_reward_over_ep_len_and_envs\n")
            f.writelines(f"{i},{r:.6f}\n" for i, r in enumerate(rewards))
    k = max(iters // 10, 1)
    print(f"{name}: {iters} APG iterations ({num_envs} envs, ep_len {ep_len}) in {time.time() - t0:.1f}s; mean reward first {k}: "
          f"{np.mean(rewards[:k]):.4f}  last {k}: {np.mean(rewards[-k:]):.4f}  (max {max(rewards):.4f})", flush=True)
