"""Phase times inside pcl_fwd_kernel / pcl_bwd_kernel (diagnostic build -DUD_PCL_STAMPS of plb_cluster.hip, see tools/pcl_stamps.sh):
the Torus bench shape (8 envs x 1000 particles, one env.step forward with checkpoint + adjoint), s_memtime sums of thread 0 of every part.
usage (GPU box): UNIDOM_HIP_SO=$PWD/gpurun_in/lib_pcl_stamps.so python tools/pcl_stamps.py [n_grid]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidom_amd import _lib
from unidom_amd.engine.plb_simulator import PlbConf, PlbSimulator

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = PlbConf()
cfg.quality = 2.0 if ng == 128 else 1.0
B = 8
sim = PlbSimulator(cfg, B)
st = sim.reset()
st = st._replace(prim_pos=st.prim_pos.clone())
st.prim_pos[:, 0] = st.x[:, 7]
act = torch.tensor(np.repeat(np.array([[-0.0014, 0.0013, 0.0]]), B, 0), dtype=torch.float64, device=sim.device, requires_grad=True)
L = _lib.lib()
buf = (C.c_ulonglong * 32)()


def one():
    s = st
    for _ in range(3):
        s = sim.step(s, act)
    (s.x.sum() + s.v.sum()).backward()


one()
torch.cuda.synchronize()
assert L.ud_debug_pcl_stamps(buf, 1) == 0
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(3):
    one()
t1.record()
torch.cuda.synchronize()
assert L.ud_debug_pcl_stamps(buf, 0) == 0
t = np.array(list(buf), dtype=np.float64).reshape(2, 16)
S = sim.substeps
names = {0: ("pcl_fwd_kernel", ["prologue", "clear + pre-pass + record stores", "p2g walk", "compaction", "flush atomics", "barrier", "read-back + grid op + zeroing", "g2p", "epilogue"]),
         1: ("pcl_bwd_kernel", ["prologue", "restore + loads + pre-pass", "grid op", "g2p adjoint walk", "flush atomics", "barrier", "read-back + grid-op adjoint + zeroing", "p2g adjoint + particle adjoint", "epilogue"])}
print(f"n_grid {sim.n_grid}, {S} substeps per call, {B} envs; 9 step calls each way in {t0.elapsed_time(t1):.2f} ms (stamped build)")
for k, (kn, ph) in names.items():
    parts = t[k, 15] or 1
    tot = t[k, :len(ph)].sum()
    print(f"{kn}: {int(parts)} parts sampled; ticks per part and launch {tot / parts:.0f} = {tot / parts / S:.1f} per substep")
    for i, nme in enumerate(ph):
        per = t[k, i] / parts / (S if 1 <= i <= 7 else 1)
        print(f"   {nme:45s} {per:9.1f} ticks {'per substep' if 1 <= i <= 7 else 'per launch'}   {100 * t[k, i] / tot:5.1f} %")
    extra = {0: ((9, "(phase 1, split off) table clear"), (10, "(phase 1, split off) weights + pre-pass")), 1: ((9, "arrive + pre-pass (between flush and wait)"),)}[k]
    for i, nme in extra:
        if t[k, i]:
            print(f"   {nme:45s} {t[k, i] / parts / S:9.1f} ticks per substep")
    print(f"   all phases: {t[k, :12].sum() / parts / S:.0f} ticks per substep")
