"""Diagnostic: which term of the norm_grad_state global norm differs between the many-workgroup adjoint and the oracle?
usage (GPU box): PYTHONPATH=$PWD:$PWD/tests python tools/debug_norm.py"""
import numpy as np

from oracle.pyoracle import MpmOracle
from test_mpm_gpu import LegacyConf, run_hip
from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator

rng = np.random.default_rng(11)
S, N, B = 4, 160, 2
conf = LegacyConf(); conf.steps = S
sim = SimpleMPMSimulator(conf, B, use_position_control=True)
sim.n_particles, sim.material, sim.h = N, np.full(N, 1, np.int32), np.ones(N, np.float32)
sim._make_handle()
x = (0.004 + rng.uniform(0, 0.06, size=(B, N, 3))).astype(np.float32)
ppos = np.zeros((B, S, 3), np.float32); ppos[:, 0] = x.mean(1) + rng.normal(size=(B, 3)).astype(np.float32) * 0.01
prot = np.zeros((B, S, 4), np.float32); prot[..., 0] = 1
mu0, la0 = 100 / (2 * 1.1), 100 * 0.1 / (1.1 * 0.8)
st = dict(x=x, v=(rng.normal(size=(B, N, 3)) * 0.3).astype(np.float32), C=(rng.normal(size=(B, N, 3, 3)) * 2).astype(np.float32),
          F=(np.eye(3) + rng.normal(size=(B, N, 3, 3)) * 0.03).astype(np.float32), J=np.ones((B, N), np.float32), ppos=ppos, prot=prot,
          psize=np.tile(np.float32([0.02, 0.02, 0.02]), (B, 1)), friction=np.full(B, 0.3, np.float32),
          mu=np.full(B, mu0, np.float32), lamda=np.full(B, la0, np.float32), action=(rng.normal(size=(B, 6)) * 0.01).astype(np.float32))
g = dict(gx=rng.normal(size=(B, N, 3)), gv=rng.normal(size=(B, N, 3)) * 0.01, gC=rng.normal(size=(B, N, 3, 3)) * 1e-4,
         gF=rng.normal(size=(B, N, 3, 3)) * 0.01, gppos=rng.normal(size=(B, S, 3)), gprot=rng.normal(size=(B, S, 4)))
g = {k: v.astype(np.float32) for k, v in g.items()}
orc = MpmOracle(N, steps=S, position_control=True)
st64, g64 = {k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}
raw_o, clp_o = orc.step_bwd(st64, g64, clip=False), orc.step_bwd(st64, g64, clip=True)
raw_h, clp_h = run_hip(sim, st, g=g, clip=False), run_hip(sim, st, g=g, clip=True)
keys = ("gx", "gv", "gC", "gF", "gppos")
rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
print("clip=False rel err:", {k: float("%.1e" % rel(raw_h[k], raw_o[k])) for k in keys + ("gaction", "gfriction", "gmu", "glamda")})
for b in range(B):
    sq = lambda d: {k: float((np.asarray(d[k][b], np.float64) ** 2).sum()) for k in keys}
    so, sh = sq(raw_o), sq(raw_h)
    # implied norms: raw / clipped, elementwise on the largest entry of gx
    i = np.unravel_index(np.abs(raw_o["gx"][b]).argmax(), raw_o["gx"][b].shape)
    print(f"env {b}: sum-sq oracle {so}\n        sum-sq hip    {sh}")
    print(f"        implied sn oracle {raw_o['gx'][b][i] / clp_o['gx'][b][i]:.6f}  hip {raw_h['gx'][b][i] / clp_h['gx'][b][i]:.6f}"
          f"   sqrt(sum state sq) oracle {np.sqrt(sum(so.values())):.6f} hip {np.sqrt(sum(sh.values())):.6f}")
    print(f"        gfriction/gmu/glamda oracle {raw_o['gfriction'][b]:.4e} {raw_o['gmu'][b]:.4e} {raw_o['glamda'][b]:.4e}  "
          f"hip {raw_h['gfriction'].reshape(-1)[b]:.4e} {raw_h['gmu'].reshape(-1)[b]:.4e} {raw_h['glamda'].reshape(-1)[b]:.4e}")
