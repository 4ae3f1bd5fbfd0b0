import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from unidom_amd.envs.registration import env_functions
from unidom_amd.utils import prng
env = env_functions["whip_rope"](batch_size=32, seed=0)
_, st = env.reset(prng.PRNGKey(0))
a = torch.zeros((32, 6), device=env.device); a[:, 0] = 0.5
with torch.no_grad():
    for _ in range(3): env.simulator.step_jax(st, a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): env.simulator.step_jax(st, a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(os.environ.get("UNIDOM_HIP_SO", "default")[-22:], "fwd step %.3f ms = %.2f us/substep" % (dt * 1e3, dt * 1e6 / 70))
