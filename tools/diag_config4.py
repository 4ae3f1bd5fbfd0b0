"""diagnostic: per-env adjoint error of the one-workgroup MPM kernels vs the f64 oracle, as a function of the number of substeps"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import GOLDEN
from test_oracle_mpm import LA0, MU0, _adjoint_case
from test_mpm_gpu import make_sim, run_hip, _rel
from oracle.pyoracle import MpmOracle
demo = dict(np.load(os.path.join(GOLDEN, "whip_rope_demo0.npz")))
B = 32
for S in (3, 20, 50, 70):
    rng = np.random.default_rng(4)
    ks = rng.integers(0, 69, size=B)
    cases = [_adjoint_case(demo, S, int(k), 1, 100 + n, np.float32) for n, k in enumerate(ks)]
    st = {k: np.concatenate([c[0][k] for c in cases]) for k in cases[0][0]}
    g = {k: np.concatenate([c[1][k] for c in cases]) for k in cases[0][1]}
    st["action"] = (rng.uniform(-1, 1, size=(B, 6)) * np.float32([1, 1, 1, 0, 0, 0]) / 50).astype(np.float32)
    st["friction"] = rng.uniform(0.05, 0.3, size=B).astype(np.float32)
    st["mu"] = (MU0 * rng.uniform(0.7, 1.3, size=B)).astype(np.float32)
    st["lamda"] = (LA0 * rng.uniform(0.7, 1.3, size=B)).astype(np.float32)
    orc = MpmOracle(67, steps=S)
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}, clip=True, nthreads=8)
    ob32 = orc.step_bwd(st, g, clip=True, nthreads=8)
    obn = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}, clip=False, nthreads=8)
    oh = run_hip(make_sim(S, B), st, g=g, clip=True)
    ohn = run_hip(make_sim(S, B), st, g=g, clip=False)
    for key in ("gx", "gF", "gaction"):
        errs = np.array([_rel(oh[key][b], ob[key][b]) for b in range(B)])
        errn = np.array([_rel(ohn[key][b], obn[key][b]) for b in range(B)])
        gaps = np.array([_rel(ob32[key][b], ob[key][b]) for b in range(B)])
        w = int(errs.argmax())
        print(f"S={S:3d} {key:8s} clip: max {errs.max():.2e} (env {w}, demo state {ks[w]}), median {np.median(errs):.2e}; f32-oracle gap max {gaps.max():.2e}; no-clip: max {errn.max():.2e} (env {int(errn.argmax())})", flush=True)
    # magnitude of the raw (unclipped) cotangent norm vs the clip threshold 1.0
    nrm = np.sqrt(sum((obn[k].reshape(B, -1) ** 2).sum(1) for k in ("gx", "gv", "gC", "gF")))
    print(f"      unclipped state-cotangent norm: min {nrm.min():.2e} max {nrm.max():.2e}; env 4: {nrm[4]:.3e}")
