#!/usr/bin/env python3
"""Cost of ud_mpm_conf.deterministic: forward ms per simulator.step, deterministic against default, 32 envs.
    python tools/det_cost.py > gpurun_out/det_cost.txt   (tools/final_regression.sh copies it to profiles/<TAG>_det_cost.txt)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_mpm_gpu import LegacyConf, ScaledConf, _scaled_case, run_hip
    from test_mpm_det import _det_sim
    B = 32
    print("# forward only (no checkpoint), %d envs, ms per ud_mpm_step_fwd call; then forward + backward per simulator.step pair; 1x MI355X" % B)
    for name in ("whip_rope N=67 res 32^3, 70 substeps", "rope at n_grid 128 N=798 res 64^3, 70 substeps"):
        row = []
        for det in (0, 1):
            if name.startswith("whip"):
                sim = _det_sim(70, B, det)
                rng = np.random.default_rng(0)
                N = 67
                x = np.stack([np.linspace(0.2, 0.4, N), np.full(N, 0.02), np.full(N, 0.25)], -1).astype(np.float32)
                ppos = np.zeros((70, 3), np.float32); ppos[0] = x[N // 2]
                prot = np.zeros((70, 4), np.float32); prot[:, 0] = 1
                st = dict(x=np.repeat(x[None], B, 0), v=np.zeros((B, N, 3), np.float32), C=np.zeros((B, N, 3, 3), np.float32),
                          F=np.repeat(np.eye(3, dtype=np.float32)[None, None], B, 0).repeat(N, 1), J=np.ones((B, N), np.float32),
                          ppos=np.repeat(ppos[None], B, 0), prot=np.repeat(prot[None], B, 0), psize=np.repeat(np.float32([[0.02, 0.06, 0.02]]), B, 0),
                          friction=np.full(B, 0.1, np.float32), mu=np.full(B, 41.7, np.float32), lamda=np.full(B, 27.8, np.float32),
                          action=np.repeat(np.float32([[0.4, 0.1, 0.3, 0, 0, 0]]) / 50, B, 0))
            else:
                class Conf(ScaledConf):
                    deterministic = det
                sim, st, _, N = _scaled_case(70, 0, B=B, conf_cls=Conf)
            rng = np.random.default_rng(1)
            g = dict(gx=rng.normal(size=(B, N, 3)), gv=rng.normal(size=(B, N, 3)), gC=rng.normal(size=(B, N, 3, 3)) * 0.01,
                     gF=rng.normal(size=(B, N, 3, 3)) * 0.1, gppos=rng.normal(size=(B, 70, 3)))
            for gg in (None, g):
                for _ in range(2):
                    run_hip(sim, st, g=gg)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    run_hip(sim, st, g=gg)
                torch.cuda.synchronize()
                row.append((time.perf_counter() - t0) / 5 * 1e3)
        print("%-55s default %8.2f ms   deterministic %8.2f ms   x%.1f" % (name, row[0], row[2], row[2] / row[0]))
        print("%-55s default %8.2f ms   deterministic %8.2f ms   x%.1f" % ("   ... forward with checkpoint + backward (host copies in both)", row[1], row[3], row[3] / row[1]))


if __name__ == "__main__":
    main()
