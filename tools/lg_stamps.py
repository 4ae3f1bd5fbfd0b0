"""Phase times inside lg_p2g<1> / lg_g2p_adj<1> / lg_p2g_adj<1> (diagnostic build -DUD_LG_STAMPS of mpm_large.hip, see
tools/lg_stamps.sh): run a few APG updates of a many-workgroup workload, then read the s_memtime sums of wave 0 of every block.
usage (GPU box): UNIDOM_HIP_SO=$PWD/gpurun_in/lib_stamps.so python tools/lg_stamps.py pour_soup"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidom_amd import _lib
from unidom_amd.algorithms.apg.core import APG
from unidom_amd.envs.registration import env_functions
from unidom_amd.utils import prng

name = sys.argv[1] if len(sys.argv) > 1 else "pour_soup"
env = env_functions[name](batch_size=32, seed=0, aux_reward=True, device="cuda:0")
_, state = env.reset(prng.PRNGKey(0))
learner = APG(env, 3, learning_rate=1e-4, max_gradient_norm=0.3, seed=0)
L = _lib.lib()
buf = (C.c_ulonglong * 32)()
for _ in range(2):
    learner.minimize(state)
torch.cuda.synchronize()
assert L.ud_debug_lg_stamps(buf, 1) == 0
for _ in range(3):
    learner.minimize(state)
torch.cuda.synchronize()
assert L.ud_debug_lg_stamps(buf, 0) == 0
t = np.array(list(buf), dtype=np.float64).reshape(4, 8)
names = {0: ("lg_p2g<1>", ["clear + state loads", "pre-pass (SVD, stress)", "window reduction + barriers", "27-cell walk", "barrier before flush", "flush atomics + bitmap OR", "list append"]),
         1: ("lg_g2p_adj<1>", ["clear + x loads", "window reduction + barriers (cotangent loads under them)", "scatter walk (table adds)", "gather walk (velocity loads)", "scratch stores", "barrier before flush", "flush atomics"]),
         2: ("lg_p2g_adj<1>", ["state loads", "pre-pass + adjoint extras", "cotangent loads", "27-cell gather", "particle adjoint + stores"]),
         3: ("lg_grid_adj (soft contact, per 256-cell tile)", ["list / checkpoint / cotangent loads", "collide chains + adjoints (all primitives)", "wave sums + barriers + block atomics (all primitives)", "head adjoint + store"])}
for k, (kn, ph) in names.items():
    n = t[k, 7] if k != 2 else None
    tot = t[k, :len(ph)].sum()
    blocks = t[k, 7] if t[k, 7] else 1
    print(f"{kn}: {int(t[k, 7])} blocks sampled; ticks per block (wave 0) {tot / blocks:.0f}")
    for i, p in enumerate(ph):
        print(f"    {p:32s} {t[k, i] / blocks:9.0f} ticks  {100 * t[k, i] / tot:5.1f} %")
