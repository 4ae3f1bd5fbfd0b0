"""Kernel-time probe for the cloth rollout kernels (not a test, not the bench): fold_cloth1 state, B envs, T macro
steps; prints the average launch time of forward (with / without checkpoints) and adjoint from HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidom_amd.envs.registration import env_functions
from unidom_amd.engine.cloth_simulator import ClothSimulator, _Rollout
from unidom_amd.utils import prng

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
env = env_functions["fold_cloth1"](batch_size=B, seed=0, aux_reward=True)
mask = env.cloth_mask.copy()
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # keep only the first `rows` lattice rows of the cloth (P = rows * 32)
if rows:
    import numpy as np
    r = np.where(mask.any(1))[0]
    mask[r[rows]:, :] = 0
sim = ClothSimulator(env.conf, B, env.get_collision_func(), mask, device=dev, mode=mode)
print("P =", int(mask.sum()))
st = sim.reset_jax()
g = torch.Generator(device=dev).manual_seed(0)
acts = (torch.rand((T, B, 8), device=dev, generator=g) * 2 - 1)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def fwd_nograd():
    with torch.no_grad():
        sim.rollout(st, acts, want_lists=False)

x = st.x.clone().requires_grad_(True)
def fwd_grad():
    global out
    out = sim.rollout(st._replace(x=x), acts, want_lists=False)
def bwd():
    torch.autograd.grad(out[0].x.sum(), x, retain_graph=True)

t0 = timed(fwd_nograd); t1 = timed(fwd_grad); fwd_grad(); t2 = timed(bwd)
n = T * 50
print(f"B={B} T={T} mode={mode}: fwd(no ckpt) {t0:.3f} ms ({t0*1e3/n:.3f} us/substep)  fwd(ckpt) {t1:.3f} ms ({t1*1e3/n:.3f})  bwd {t2:.3f} ms ({t2*1e3/n:.3f})")
