"""Phase times inside the persistent cluster forward kernel (diagnostic build -DUD_LG_STAMPS, tools/lg_stamps.sh): s_memtime sums of
lane 0 of every part (100 MHz ticks), per substep.   usage (GPU box): UNIDOM_HIP_SO=$PWD/gpurun_in/lib_stamps.so python tools/clm_stamps.py [T]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UD_MPM_CLUSTER"] = "1"
os.environ["UD_MPM_CLUSTER_T"] = sys.argv[1] if len(sys.argv) > 1 else "128"
import bench   # noqa: E402
from unidom_amd import _lib   # noqa: E402

class A: pass
a = A(); a.n_grid, a.envs, a.grid_ckpt, a.warmup, a.steps = 128, 32, 2, 1, 3
L = _lib.lib()
buf = (C.c_ulonglong * 32)()
import io, contextlib
assert L.ud_debug_lg_stamps(buf, 1) == 0
with contextlib.redirect_stdout(io.StringIO()):
    bench.bench_mpm_scaled(a, 0, 1, torch.device("cuda:0"))
torch.cuda.synchronize()
assert L.ud_debug_lg_stamps(buf, 0) == 0
t = np.array(list(buf), dtype=np.float64).reshape(4, 8)
ph = ["table clear + pre-pass", "window reduction", "27-cell walk", "flush atomics + drain", "arrival + poll", "read-back + grid op + zeroing", "g2p"]
n = t[0, 7]
print(f"T={os.environ['UD_MPM_CLUSTER_T']}: {int(n)} part-substeps sampled; s_memtime ticks per substep (lane 0 of a part) {t[0, :7].sum() / n:.0f}")
for i, p in enumerate(ph):
    print(f"    {p:32s} {t[0, i] / n:9.0f} ticks  {100 * t[0, i] / t[0, :7].sum():5.1f} %")
