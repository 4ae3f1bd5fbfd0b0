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
for S in (1, 2, 3):
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
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}, clip=False, nthreads=8)
    ob32 = orc.step_bwd(st, g, clip=False, nthreads=8)
    oh = run_hip(make_sim(S, B), st, g=g, clip=False)
    b = 8
    for key in ("gF", "gC", "gx", "gv"):
        e = np.abs(oh[key][b] - ob[key][b]).reshape(67, -1).max(1)
        e32 = np.abs(ob32[key][b] - ob[key][b]).reshape(67, -1).max(1)
        p = int(e.argmax())
        print(f"S={S} env {b} {key}: worst particle {p}: kernel err {e[p]:.3e} (oracle-f32 err there {e32[p]:.3e}), scale {np.abs(ob[key][b]).max():.3e}, |ref at p| {np.abs(ob[key][b][p]).max():.3e}")
    F = st["F"][b]; C = st["C"][b]
    Ft = (np.eye(3)[None] + 1e-4 * C.astype(np.float64)) @ F.astype(np.float64)
    sv = np.linalg.svd(Ft, compute_uv=False)
    p = int(np.abs(oh["gF"][b] - ob["gF"][b]).reshape(67, -1).max(1).argmax())
    x = st["x"][b][p]
    print(f"   particle {p}: x={x}, x*64={x*64}, sing values {sv[p]}, gaps {sv[p][0]-sv[p][1]:.2e} {sv[p][1]-sv[p][2]:.2e}; min gap over particles {np.min(np.abs(np.diff(sv,axis=1))):.2e}")
