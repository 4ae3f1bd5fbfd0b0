"""Host enqueue cost vs GPU time of one many-workgroup MPM `step` (shape_rope sizes): is the path launch-bound?
usage (GPU box): PYTHONPATH=$PWD python tools/host_rate_probe.py"""
import time

import numpy as np
import torch

from unidom_amd.envs.registration import env_functions

env = env_functions["shape_rope"](batch_size=32, seed=0)
env.build_reset_state()
st, sim = env.state, env.simulator
act = torch.zeros((32, 6), device=env.device)
act[:, 0] = 0.001
host, total = [], []
with torch.no_grad():
    for _ in range(60):      # warm clocks
        sim.step_jax(st, act)
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.step_jax(st, act)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0)
        total.append(t2 - t0)
n = 133 * 4 + 6
print(f"launches/step {n}: host enqueue {np.median(host) * 1e3:.2f} ms ({np.median(host) / n * 1e6:.2f} us/launch), "
      f"enqueue + GPU {np.median(total) * 1e3:.2f} ms")
