"""GPU parity: HIP MPM `step` (through the C ABI, via SimpleMPMSimulator) vs the reference's recorded
trajectory (whip_rope demo) and vs the CPU oracle (forward and adjoint).

p2g accumulates with LDS float atomics (summation order differs run to run) and the kernel's SVD is a
Jacobi iteration, so everything here is tolerance-based; each tolerance is written at its assert.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from test_oracle_mpm import LA0, MU0, S_LEGACY, _adjoint_case, legacy_state

pytestmark = pytest.mark.gpu


class LegacyConf:   # whip_rope_env.py:27-73 with the legacy overrides of SURVEY.md F3
    n_grid = 64
    res = (32, 32, 32)
    dt = 1e-4
    steps = S_LEGACY
    E, nu = 100, 0.2
    ground_friction = 0.1
    dx, inv_dx = 1 / 64, 64.0
    p_vol = (dx * 0.5) ** 2
    p_mass = p_vol * 1
    gravity = (0, -9.8, 0)
    n_primitive = 1


def make_sim(steps, B, material=1, N=67):
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    conf = LegacyConf()
    conf.steps = steps
    sim = SimpleMPMSimulator(conf, B, use_position_control=True)
    sim.n_particles = N
    sim.material = np.full(N, material, np.int32)
    sim.h = np.ones(N, np.float32)
    sim._make_handle()
    return sim


def run_hip(sim, st, g=None, clip=True):
    from unidom_amd.engine.mpm_simulator import _Step
    dev = sim.device
    rg = g is not None
    t = lambda a, r=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=r)
    X, V, Cm, F, PP, FR, MU, LA, AC = (t(st[k], rg) for k in ("x", "v", "C", "F", "ppos", "friction", "mu", "lamda", "action"))
    sim.clip_grad = clip
    out = _Step.apply(sim, X, V, Cm, F, t(st["J"]), PP, t(st["prot"]), t(st["psize"]), FR, MU, LA, AC)
    res = {k: o.detach().cpu().numpy() for k, o in zip(("x", "v", "C", "F", "J", "ppos", "prot", "pv", "pw"), out)}
    if rg:
        loss = (out[0] * t(g["gx"])).sum() + (out[1] * t(g["gv"])).sum() + (out[2] * t(g["gC"])).sum() + \
            (out[3] * t(g["gF"])).sum() + (out[5] * t(g["gppos"])).sum()
        loss.backward()
        res.update(gx=X.grad.cpu().numpy(), gv=V.grad.cpu().numpy(), gC=Cm.grad.cpu().numpy(), gF=F.grad.cpu().numpy(),
                   gppos=PP.grad.cpu().numpy(), gfriction=FR.grad.cpu().numpy(), gmu=MU.grad.cpu().numpy(),
                   glamda=LA.grad.cpu().numpy(), gaction=AC.grad.cpu().numpy())
    sim.check_status()
    return res


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.fixture(scope="module")
def demo():
    return np.load(os.path.join(GOLDEN, "whip_rope_demo0.npz"))


def test_golden_all_transitions_hip(demo):
    """The reference's own recording: 69 transitions x 50 substeps, batched as B=69 envs in one launch."""
    d = demo
    sts, shifts = zip(*[legacy_state(d, k) for k in range(69)])
    st = {k: np.concatenate([s[k] for s in sts]) for k in sts[0]}
    sim = make_sim(S_LEGACY, 69)
    o = run_hip(sim, st)
    for k in range(69):
        assert np.abs(o["x"][k] - shifts[k] - d["x"][k + 1]).max() < 1e-6        # oracle reaches 1.2e-7
        assert _rel(o["v"][k], d["v"][k + 1]) < 1e-4                              # north_star tolerance
        assert _rel(o["C"][k], d["C"][k + 1]) < 1e-3
        assert np.abs(o["F"][k] - d["F"][k + 1]).max() < 5e-5
        assert np.abs(o["ppos"][k, 0] - shifts[k] - d["prim_position"][k + 1, 0]).max() < 1e-7
        np.testing.assert_array_equal(o["J"][k], d["J"][k + 1])


@pytest.mark.parametrize("material", [1, 2])
def test_fwd_matches_oracle(demo, material):
    from oracle.pyoracle import MpmOracle
    S = 20
    st, g = _adjoint_case(demo, S, 30, material, 1, np.float32)
    orc = MpmOracle(67, steps=S, material=np.full(67, material))
    of = orc.step_fwd(st)
    oh = run_hip(make_sim(S, 1, material), st)
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4      # north_star: 1e-4 relative
    # F (FMA-contracted f32 products of (I + dt C) F with |C| ~ 1e3, Jacobi SVD in the plastic projection): 5e-5
    assert _rel(oh["C"], of["C"]) < 1e-3 and _rel(oh["F"], of["F"]) < 5e-5
    for key in ("ppos", "prot", "pv", "pw"):
        np.testing.assert_allclose(oh[key], of[key], rtol=0, atol=1e-7)


@pytest.mark.parametrize("clip,material,S,k", [(False, 1, 3, 40), (True, 1, 3, 40), (False, 2, 3, 20), (True, 1, 50, 40)])
def test_bwd_matches_oracle(demo, clip, material, S, k):
    """Adjoint kernel vs the f64 oracle adjoint on the same (f32-representable) inputs."""
    from oracle.pyoracle import MpmOracle
    st, g = _adjoint_case(demo, S, k, material, 0, np.float32)
    st64 = {kk: v.astype(np.float64) for kk, v in st.items()}
    g64 = {kk: v.astype(np.float64) for kk, v in g.items()}
    ob = MpmOracle(67, steps=S, material=np.full(67, material)).step_bwd(st64, g64, clip=clip)
    oh = run_hip(make_sim(S, 1, material), st, g=g, clip=clip)
    # f32 kernel (Jacobi SVD, safe-inverse SVD VJP, LDS atomics) vs f64 oracle: 2e-3 relative (max-norm);
    # S=50 accumulates 50 reverse substeps: 1e-2
    tol = 2e-3 if S <= 3 else 1e-2
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        assert _rel(oh[key], ob[key]) < tol, (key, _rel(oh[key], ob[key]))
    for key in ("gfriction", "gmu", "glamda"):
        assert _rel(oh[key].reshape(-1), ob[key]) < 5 * tol, (key, oh[key], ob[key])


def test_config4_default_shape_32_envs_70_substeps_matches_oracle(demo):
    """BASELINE config 4's default launch shape (whip_rope: 256 envs over 8 GPUs = 32 envs per GPU, N = 67, res 32^3, 70
    substeps per step): 32 different rope states / actions / parameters in one launch, one full `step` forward and adjoint,
    against the CPU oracle env by env.  Forward: the single-env tolerances.  Adjoint: every env within 2e-3 of the f64 oracle on
    every output (measured worst env: gx 1.9e-4, gF 2.3e-4, gaction 2.3e-5 -- the same figures as the restatement's own f32 run,
    i.e. what 70 substeps of f32 forward state cost).
    History, because the earlier criterion was wrong about the cause: until round 2 a few envs were off by percents (worst gF
    7e-4 / 3e-3 / 5e-3 / 4e-2 after 3 / 20 / 50 / 70 substeps, tools/diag_config4*.py) and the test asked for a median of 1e-3 and
    1e-1 at worst, blaming the 1 / (s_j^2 - s_i^2) of the reference's SVD cotangent (svd_safe_batch.py:65-102) for amplifying
    forward differences.  The factor is there, but what it amplified was the kernel's own rounding: the rotation's cotangent
    enters that VJP as two matrix products whose antisymmetric parts cancel only in exact arithmetic (mpm_device.h,
    particle_adjoint).  Evaluated in closed form -- (W - W^T)_ij (S_j - S_i) / (S_j^2 - S_i^2) from ONE product W -- the outliers
    are gone and the result no longer depends on how the compiler schedules the products."""
    from oracle.pyoracle import MpmOracle
    S, B = 70, 32
    rng = np.random.default_rng(4)
    ks = rng.integers(0, 69, size=B)
    cases = [_adjoint_case(demo, S, int(k), 1, 100 + n, np.float32) for n, k in enumerate(ks)]
    st = {k: np.concatenate([c[0][k] for c in cases]) for k in cases[0][0]}
    g = {k: np.concatenate([c[1][k] for c in cases]) for k in cases[0][1]}
    st["action"] = (rng.uniform(-1, 1, size=(B, 6)) * np.float32([1, 1, 1, 0, 0, 0]) / 50).astype(np.float32)   # whip_rope_env.py:108-115
    st["friction"] = rng.uniform(0.05, 0.3, size=B).astype(np.float32)
    st["mu"] = (MU0 * rng.uniform(0.7, 1.3, size=B)).astype(np.float32)
    st["lamda"] = (LA0 * rng.uniform(0.7, 1.3, size=B)).astype(np.float32)
    orc = MpmOracle(67, steps=S)
    of = orc.step_fwd(st, nthreads=8)
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}, clip=True, nthreads=8)
    ob32 = orc.step_bwd(st, g, clip=True, nthreads=8)
    oh = run_hip(make_sim(S, B), st, g=g, clip=True)
    for b in range(B):
        assert _rel(oh["x"][b], of["x"][b]) < 1e-5 and _rel(oh["v"][b], of["v"][b]) < 1e-4, b      # north_star: 1e-4 relative
        assert _rel(oh["C"][b], of["C"][b]) < 1e-3 and _rel(oh["F"][b], of["F"][b]) < 5e-5, b
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        errs = np.array([_rel(oh[key][b], ob[key][b]) for b in range(B)])
        assert errs.max() < (2e-4 if key == "gaction" else 2e-3), (key, np.median(errs), errs.max(), int(errs.argmax()))
    assert _rel(ob32["gF"], ob["gF"]) < 2e-3           # the restatement's own f32 adjoint, for the record (see the docstring)
    for key in ("gfriction", "gmu", "glamda"):
        assert _rel(oh[key].reshape(-1), ob[key]) < 5e-2, (key, oh[key].reshape(-1), ob[key])


def _nearly_equal_singular_values(rng, n, gaps=(1e-3, 1e-4, 3e-5)):
    """F = Q1 diag(s) Q2 with random rotations and singular values that differ by `gaps` (relative) -- where the SVD cotangent's
    1 / (s_j^2 - s_i^2) is 5e2 ... 2e4."""
    def rot(k):
        q, r = np.linalg.qr(rng.normal(size=(k, 3, 3)))
        q = q * np.sign(np.linalg.det(q))[:, None, None]
        return q
    base = rng.uniform(0.95, 1.05, size=n)
    gap = np.asarray(gaps)[rng.integers(0, len(gaps), size=(n, 2))]
    sv = np.stack([base, base * (1 + gap[:, 0]), base * (1 + gap[:, 0]) * (1 + gap[:, 1])], 1)
    return (rot(n) * sv[:, None, :]) @ rot(n)


@pytest.mark.parametrize("material", [1, 2])
def test_adjoint_with_nearly_equal_singular_values(demo, material):
    """Every particle's F has singular values 1e-3 ... 3e-5 apart.  The reference's SVD cotangent applied literally lost
    eps / gap there (two products whose antisymmetric parts cancel only in exact arithmetic, times 1 / (s_j^2 - s_i^2)): percent
    errors; the closed form of particle_adjoint does not.  One-workgroup kernels, f64 oracle."""
    from oracle.pyoracle import MpmOracle
    S = 3
    st, g = _adjoint_case(demo, S, 40, material, 0, np.float32)
    st["F"] = _nearly_equal_singular_values(np.random.default_rng(7), 67)[None].astype(np.float32)
    st64 = {kk: v.astype(np.float64) for kk, v in st.items()}
    ob = MpmOracle(67, steps=S, material=np.full(67, material)).step_bwd(st64, {kk: v.astype(np.float64) for kk, v in g.items()}, clip=False)
    oh = run_hip(make_sim(S, 1, material), st, g=g, clip=False)
    for key in ("gx", "gv", "gC", "gF", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        per = np.linalg.norm((oh[key][0] - ob[key][0]).reshape(len(ob[key][0]), -1), axis=1) / (np.linalg.norm(ob[key][0].reshape(len(ob[key][0]), -1), axis=1) + 1e-30) \
            if key != "gaction" else np.array([_rel(oh[key], ob[key])])
        assert _rel(oh[key], ob[key]) < 5e-4, (key, _rel(oh[key], ob[key]))
        if key in ("gC", "gF"):
            assert per.max() < 5e-3, (key, per.max(), int(per.argmax()))     # no particle left behind


def test_adjoint_with_nearly_equal_singular_values_many_workgroups(one_lane_per_particle):
    """The same on the many-workgroup path (N = 798, one lane per particle: lg_p2g_adj<1>, the kernel whose register pressure is
    the tightest and whose schedule the literal form's error depended on)."""
    from oracle.pyoracle import MpmOracle
    S = 3
    sim, st, g, N = _scaled_case(S, 5, B=2)
    st["F"] = np.stack([_nearly_equal_singular_values(np.random.default_rng(8 + b), N) for b in range(2)]).astype(np.float32)
    orc = MpmOracle(N, n_grid=128, res=(64, 64, 64), steps=S)
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}, clip=False, nthreads=2)
    oh = run_hip(sim, st, g=g, clip=False)
    for key in ("gx", "gv", "gC", "gF", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        assert _rel(oh[key], ob[key]) < 5e-4, (key, _rel(oh[key], ob[key]))
    per = np.linalg.norm((oh["gF"] - ob["gF"]).reshape(2, N, 9), axis=2) / (np.linalg.norm(ob["gF"].reshape(2, N, 9), axis=2) + 1e-30)
    assert per.max() < 5e-3, (per.max(), np.unravel_index(per.argmax(), per.shape))


@pytest.mark.parametrize("N", [30, 64, 96, 97, 128])
def test_particle_counts_cover_both_adjoint_kernels(demo, N):
    """N <= 96 runs the wave-specialised adjoint (particle waves + stencil waves), 97..128 the single-mapping one;
    64 / 96 / 128 are the exact-wave edges, 30 / 97 the ragged ones.  Bodies: the first N rope particles, or the rope
    plus a second strand shifted by 3 cells."""
    from oracle.pyoracle import MpmOracle
    S = 4
    st, g = _adjoint_case(demo, S, 40, 1, 3, np.float32)
    def body(a):   # [1,67,...] -> [1,N,...]
        if N <= 67:
            return a[:, :N].copy()
        extra = a[:, :N - 67].copy()
        return np.concatenate([a, extra], 1)
    st = dict(st); g = dict(g)
    for k in ("x", "v", "C", "F"):
        st[k] = body(st[k]); g["g" + k] = body(g["g" + k])
    if N > 67:
        st["x"][:, 67:, 2] += np.float32(3 / 64)
    st["J"] = st["J"][:, :1].repeat(N, 1) if st["J"].ndim == 2 else st["J"]
    st64 = {kk: v.astype(np.float64) for kk, v in st.items()}
    g64 = {kk: v.astype(np.float64) for kk, v in g.items()}
    orc = MpmOracle(N, steps=S)
    of, ob = orc.step_fwd(st), orc.step_bwd(st64, g64, clip=True)
    oh = run_hip(make_sim(S, 1, N=N), st, g=g, clip=True)
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4 and _rel(oh["F"], of["F"]) < 5e-5
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        assert _rel(oh[key], ob[key]) < 2e-3, (key, _rel(oh[key], ob[key]))


def test_batched_envs_with_different_parameters(demo):
    """B=5 envs with different states / actions / friction / Lame parameters in one launch == 5 oracle runs."""
    from oracle.pyoracle import MpmOracle
    S = 12
    cases = [_adjoint_case(demo, S, k, 1, k, np.float32)[0] for k in (5, 20, 33, 47, 60)]
    for n, st in enumerate(cases):
        st["friction"] = np.float32([0.05 + 0.1 * n]); st["mu"] = np.float32([MU0 * (0.6 + 0.2 * n)])
        st["lamda"] = np.float32([LA0 * (1.5 - 0.2 * n)]); st["action"] = np.float32([[0.2 * n - 0.4, 0.1, -0.3 + 0.1 * n, 0, 0, 0]])
    stB = {k: np.concatenate([s[k] for s in cases]) for k in cases[0]}
    oh = run_hip(make_sim(S, 5), stB)
    orc = MpmOracle(67, steps=S)
    for n, st in enumerate(cases):
        of = orc.step_fwd(st)
        # x: 1e-5, not the 5e-6 of the other forward tests -- measured 1.75e-6 here (round 1, gpurun_out/t12.log), run-to-run
        # variation from the order of the LDS atomics over 50 substeps with per-env parameters; v: north_star's 1e-4 relative
        assert _rel(oh["x"][n], of["x"][0]) < 1e-5 and _rel(oh["v"][n], of["v"][0]) < 1e-4
        assert _rel(oh["C"][n], of["C"][0]) < 1e-3 and _rel(oh["F"][n], of["F"][0]) < 5e-5


def test_liquid_material_forward(demo):
    """material 0: mu = 0, la = 1 regardless of E (Q10)."""
    from oracle.pyoracle import MpmOracle
    S = 10
    st, _ = _adjoint_case(demo, S, 30, 0, 2, np.float32)
    of = MpmOracle(67, steps=S, material=np.zeros(67, np.int32)).step_fwd(st)
    oh = run_hip(make_sim(S, 1, material=0), st)
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4 and _rel(oh["F"], of["F"]) < 5e-5


def test_out_of_range_index_semantics(demo):
    """Q5/Q9: particles whose 3x3x3 support leaves the `res` grid -- scatter drops, gather clamps, negative
    indices wrap -- against the dense oracle."""
    from oracle.pyoracle import MpmOracle
    S = 3
    st, _ = _adjoint_case(demo, S, 10, 1, 4, np.float32)
    x = st["x"].copy()
    x[0, :10, 2] += np.float32(0.27)      # z beyond res*dx = 0.5: upper out-of-range cells
    x[0, 10:20, 0] -= np.float32(0.245)   # x*inv_dx - 0.5 < 0: base 0 with negative quadratic weights (Q13)
    st["x"] = x
    of = MpmOracle(67, steps=S).step_fwd(st)
    oh = run_hip(make_sim(S, 1), st)
    assert np.isfinite(of["x"]).all()
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4 and _rel(oh["C"], of["C"]) < 1e-3


# ---------------------------------------------------------------------------------------------------------
# many-workgroup path (N > 128): the scaled whip_rope configuration n_grid=128 -> res 64^3, N=798
# ---------------------------------------------------------------------------------------------------------
class ScaledConf(LegacyConf):
    n_grid = 128
    res = (64, 64, 64)
    dx, inv_dx = 1 / 128, 128.0
    p_vol = (dx * 0.5) ** 2
    p_mass = p_vol * 1
    E, nu = 100, 0.1


class ScaledConf256(LegacyConf):
    """BASELINE config 4's stated size: "128^3 grid" = res 128^3 at n_grid 256 -> N = 6675."""
    n_grid = 256
    res = (128, 128, 128)
    dx, inv_dx = 1 / 256, 256.0
    p_vol = (dx * 0.5) ** 2
    p_mass = p_vol * 1
    E, nu = 100, 0.1


def _scaled_case(S, seed, B=1, grid_ckpt_cells=0, conf_cls=ScaledConf):
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    conf = conf_cls()
    conf.steps = S
    sim = SimpleMPMSimulator(conf, B, use_position_control=True)
    sim.grid_ckpt_cells = grid_ckpt_cells
    st0 = sim.add_box(conf, None, size=[0.38, 0.006, 0.006], init_pos=[0.25, 0.01, 0.25], z_rotation_angle=np.pi / 2,
                      material=1, density=2.75, hardness=1.0)
    x = st0.x.cpu().numpy()
    N = x.shape[0]
    sim.n_particles = N
    sim._make_handle()
    rng = np.random.default_rng(seed)
    mu0, la0 = 100 / (2 * 1.1), 100 * 0.1 / (1.1 * 0.8)
    ppos = np.zeros((S, 3), np.float32)
    ppos[0] = x[N // 3] + np.float32([0.0, 0.0, 0.0])
    prot = np.zeros((S, 4), np.float32); prot[:, 0] = 1
    st = dict(x=np.repeat(x[None], B, 0), v=(rng.normal(size=(B, N, 3)) * 0.05).astype(np.float32),
              C=(rng.normal(size=(B, N, 3, 3)) * 1.0).astype(np.float32),
              F=(np.eye(3, dtype=np.float32)[None, None] + rng.normal(size=(B, N, 3, 3)).astype(np.float32) * 0.01),
              J=np.ones((B, N), np.float32), ppos=np.repeat(ppos[None], B, 0), prot=np.repeat(prot[None], B, 0),
              psize=np.repeat(np.float32([[0.02, 0.02, 0.02]]), B, 0), friction=np.full(B, 0.1, np.float32),
              mu=np.full(B, mu0, np.float32), lamda=np.full(B, la0, np.float32),
              action=np.repeat(np.float32([[0.4, -0.1, 0.3, 0, 0, 0]]) / 50, B, 0))   # env scale: a/50 (whip_rope_env.py:112)
    g = dict(gx=rng.normal(size=(B, N, 3)), gv=rng.normal(size=(B, N, 3)), gC=rng.normal(size=(B, N, 3, 3)) * 0.01,
             gF=rng.normal(size=(B, N, 3, 3)) * 0.1, gppos=rng.normal(size=(B, S, 3)))
    return sim, st, {k: v.astype(np.float32) for k, v in g.items()}, N


def test_ckpt_cells_counts_the_active_cells_of_a_step(large_path):
    """ud_mpm_ckpt_cells (bench.py's G_act, tools/traffic_table.py): the grid-checkpoint records of a step call per env.  One substep of the
    rope at n_grid 128 from rest: the records are exactly the cells the 27-point stencils of its particles reach inside `res`
    (mpm_simulator.py:178-194: scatter drops what lies outside), counted on the host; a handle without a grid checkpoint reports 0."""
    import ctypes as C
    from unidom_amd import _lib
    S = 1
    sim, st, g, N = _scaled_case(S, 5, B=3, grid_ckpt_cells=2)
    sim.keep_last_ckpt = True
    run_hip(sim, st, g=g)
    conf = ScaledConf()
    base = (st["x"][0] * np.float32(conf.n_grid) - np.float32(0.5)).astype(np.int32)
    cells = {(b[0] + i, b[1] + j, b[2] + k) for b in base for i in range(3) for j in range(3) for k in range(3)}
    cells = {c for c in cells if all(0 <= c[d] < conf.res[d] for d in range(3))}
    assert sim.active_cells_per_substep() == float(len(cells)) and len(cells) > 300
    ckpt, B = sim._last_ckpt
    out = torch.full((B,), -1, dtype=torch.int32, device=ckpt.device)
    stream = C.c_void_p(torch.cuda.current_stream(ckpt.device).cuda_stream)
    _lib.check(_lib.lib().ud_mpm_ckpt_cells(sim._h, C.c_int(B), _lib.ptr(ckpt), _lib.ptr(out), stream), "ud_mpm_ckpt_cells")
    assert out.tolist() == [len(cells)] * B
    sim0, st0, g0, _ = _scaled_case(S, 5, B=3, grid_ckpt_cells=0)
    sim0.keep_last_ckpt = True
    run_hip(sim0, st0, g=g0)
    assert sim0.active_cells_per_substep() == 0.0


def _tune(monkeypatch, **kw):
    """Kernel selection of every SimpleMPMSimulator built from here on in this test (ud_mpm_conf.tune_*, fixed at ud_mpm_create)."""
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    monkeypatch.setattr(SimpleMPMSimulator, "default_tuning", dict(kw))


@pytest.fixture(params=["cluster_128", "cluster_64", "multi_kernel"])
def large_path(request, monkeypatch):
    """The many-workgroup path below 100 k particles per launch has two forwards: the persistent cluster kernel (csrc/mpm_cluster.h; one
    launch per step call, parts of 32 or 16 particles -- tune_cluster_part_lanes) and the multi-kernel forward (tune_cluster = -1; two
    launches per substep).  By default a one-primitive solid takes the cluster forward, which writes the grid checkpoint the
    multi-kernel backward restores from.  The choice is fixed when a handle is created; the fixture sets it for the handles of a test."""
    if request.param == "multi_kernel":
        _tune(monkeypatch, cluster=-1)
    else:
        _tune(monkeypatch, cluster=1, cluster_part_lanes=int(request.param.split("_")[1]))
    return request.param


@pytest.fixture
def multi_kernel_path(monkeypatch):
    """for the tests of the multi-kernel path's own machinery (active list, bitmap, grid checkpoint)"""
    _tune(monkeypatch, cluster=-1)


@pytest.fixture
def one_lane_per_particle(monkeypatch):
    """The many-workgroup kernels come in two lane mappings: 4 lanes per particle below 100 k particles per launch, 1 beyond --
    sizes the CPU oracle cannot follow.  tune_lanes = 1 (fixed at create) puts the one-lane kernels (the ones bench.py's n_grid-256 and
    pour_soup workloads run) before the oracle at test sizes."""
    _tune(monkeypatch, lanes=1)


@pytest.mark.parametrize("grid_ckpt_cells", [0, 6])
def test_large_path_one_lane_kernels_match_oracle_n798(grid_ckpt_cells, one_lane_per_particle):
    test_large_path_matches_oracle_n798(grid_ckpt_cells, None)


def test_collide_shape_rope_geometry_one_lane_kernels(one_lane_per_particle):
    test_collide_shape_rope_geometry_fwd_bwd(None)


@pytest.mark.parametrize("case", ["body_at_the_domain_corner", "across_the_upper_grid_edge", "negative_weights", "four_box_primitives"])
def test_mpm_step_edge_cases_one_lane_kernels(demo, case, one_lane_per_particle):
    test_mpm_step_edge_cases(demo, case, None)


def test_full_batch_launch_agrees_with_the_oracle_checked_small_one():
    """Size-independent property at bench size: envs are independent, so env b of a 128-env launch (102 k particles -- past the
    100 k threshold, i.e. the one-lane kernels, four env groups under position control: what bench.py's n_grid-256 workload runs)
    must equal the same env stepped in a 2-env launch (four-lane kernels, one group -- the configuration
    test_large_path_matches_oracle_n798 pins to the oracle).  Forward and adjoint, grid checkpoint on."""
    S = 5
    simL, stL, gL, N = _scaled_case(S, 3, B=128, grid_ckpt_cells=6)
    assert 128 * N >= 100000
    pick = [0, 77]
    simS, _, _, _ = _scaled_case(S, 3, B=2, grid_ckpt_cells=6)
    stS = {k: np.ascontiguousarray(v[pick]) for k, v in stL.items()}
    gS = {k: np.ascontiguousarray(v[pick]) for k, v in gL.items()}
    big, small = run_hip(simL, stL, g=gL, clip=True), run_hip(simS, stS, g=gS, clip=True)
    for key in ("x", "v", "C", "F"):
        assert _rel(big[key][pick], small[key]) < 2e-5, (key, _rel(big[key][pick], small[key]))
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(big[key]).all(), key
        assert _rel(big[key][pick], small[key]) < 2e-3, (key, _rel(big[key][pick], small[key]))


@pytest.mark.parametrize("grid_ckpt_cells", [0, 2])
def test_baseline_config4_size_n_grid_256_matches_oracle(grid_ckpt_cells):
    """BASELINE config 4 at its stated size (whip_rope, "128^3 grid": n_grid 256, res 128^3, N = 6675).  A
    16-env launch = 106 800 particles, past the 100 k threshold, so the one-lane-per-particle kernels and a single env group
    run -- the regime bench.py --n-grid 256 measures, no lane override.  Two of its envs are followed by the CPU oracle on the
    dense 2 M-cell grid over 3 substeps (dt = 1e-4 at dx = 1/256 is 4x the default CFL number: a short window), forward in
    f32 and adjoint in f64, with the backward both recomputing the grid (0) and restoring it from the checkpoint (2, the
    bench's setting)."""
    from oracle.pyoracle import MpmOracle
    S, B, pick = 3, 16, [0, 9]
    sim, st, g, N = _scaled_case(S, 11, B=B, grid_ckpt_cells=grid_ckpt_cells, conf_cls=ScaledConf256)
    assert N == 6675 and B * N >= 100000 and _no_path_override()
    st["action"][9] = np.float32([-0.3, 0.2, 0.1, 0, 0, 0]) / 50
    sub = lambda d: {k: np.ascontiguousarray(v[pick]) for k, v in d.items()}
    orc = MpmOracle(N, n_grid=256, res=(128, 128, 128), steps=S)
    of = orc.step_fwd(sub(st), nthreads=2)
    assert all(np.isfinite(of[k]).all() for k in ("x", "v", "C", "F"))
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in sub(st).items()},
                      {k: v.astype(np.float64) for k, v in sub(g).items()}, clip=True, nthreads=2)
    oh = run_hip(sim, st, g=g, clip=True)
    sim.check_status()
    assert _rel(oh["x"][pick], of["x"]) < 5e-6 and _rel(oh["v"][pick], of["v"]) < 1e-4     # north_star: 1e-4 relative
    assert _rel(oh["C"][pick], of["C"]) < 1e-3 and _rel(oh["F"][pick], of["F"]) < 5e-5
    np.testing.assert_allclose(oh["J"][pick], of["J"], rtol=1e-5)
    for key in ("ppos", "prot", "pv", "pw"):
        np.testing.assert_allclose(oh[key][pick], of[key], rtol=0, atol=1e-7)
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        # 5e-4: measured 1e-5 ... 2e-5.  With the SVD cotangent evaluated literally this test sat at 9e-5 and jumped to 4e-2 (gC, gF:
        # one particle in a hundred off by up to 30 %) when an unrelated edit changed the compiler's schedule -- see particle_adjoint
        assert _rel(oh[key][pick], ob[key]) < 5e-4, (key, _rel(oh[key][pick], ob[key]))
    for key in ("gfriction", "gmu", "glamda"):
        assert _rel(oh[key].reshape(-1)[pick], ob[key]) < 2e-2, (key, oh[key].reshape(-1)[pick], ob[key])


@pytest.mark.parametrize("grid_ckpt_cells", [0, 6])
def test_large_path_matches_oracle_n798(grid_ckpt_cells, large_path):
    """grid_ckpt_cells = 0: the backward recomputes p2g + grid op; 6: it restores the grid from the forward's checkpoint."""
    from oracle.pyoracle import MpmOracle
    S = 5   # dt=1e-4 at dx=1/128 is 4x the CFL number of the default config: keep the window short
    sim, st, g, N = _scaled_case(S, 0, B=2, grid_ckpt_cells=grid_ckpt_cells)
    assert N == 798
    st["action"][1] = np.float32([-0.3, 0.2, 0.1, 0, 0, 0]) / 50
    orc = MpmOracle(N, n_grid=128, res=(64, 64, 64), steps=S)
    of = orc.step_fwd(st, nthreads=2)
    assert all(np.isfinite(of[k]).all() for k in ("x", "v", "C", "F"))
    st64 = {k: v.astype(np.float64) for k, v in st.items()}
    ob = orc.step_bwd(st64, {k: v.astype(np.float64) for k, v in g.items()}, clip=True, nthreads=2)
    oh = run_hip(sim, st, g=g, clip=True)
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4     # north_star: 1e-4 relative
    assert _rel(oh["C"], of["C"]) < 1e-3 and _rel(oh["F"], of["F"]) < 5e-5
    np.testing.assert_allclose(oh["J"], of["J"], rtol=1e-5)
    for key in ("ppos", "prot", "pv", "pw"):
        np.testing.assert_allclose(oh[key], of[key], rtol=0, atol=1e-7)
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        assert _rel(oh[key], ob[key]) < 5e-3, (key, _rel(oh[key], ob[key]))
    for key in ("gfriction", "gmu", "glamda"):
        assert _rel(oh[key].reshape(-1), ob[key]) < 2e-2, (key, oh[key], ob[key])
    # a second call on the same handle (persistent grid arena must be back to all-zero)
    oh2 = run_hip(sim, st)
    assert _rel(oh2["x"], of["x"]) < 5e-6 and _rel(oh2["v"], of["v"]) < 1e-4


@pytest.mark.parametrize("lanes", ["4", "1"])
def test_active_list_bitmap_rows_at_every_word_offset(lanes, monkeypatch):
    """First toucher of a cell = whoever sets its bit in the env's bitmap; a block window marks a row of eight z-consecutive cells
    with one OR, or two when the row straddles a 32-bit word (z offset of the window & 31 > 24).  The rope is moved cell by cell
    in z so that the windows start at z = 25 ... 32: forward vs the oracle each time, twice on the same handle (a cell left
    marked, or a list entry lost, shows in the second call: the grid would not be all-zero again)."""
    from oracle.pyoracle import MpmOracle
    _tune(monkeypatch, cluster=-1, lanes=int(lanes))
    S = 3
    sim, st, _, N = _scaled_case(S, 3, B=2)
    orc = MpmOracle(N, n_grid=128, res=(64, 64, 64), steps=S)
    z0 = st["x"][..., 2].copy()
    for cells in (-6, -5, -4, -3, -2, -1, 0, 1):
        st["x"][..., 2] = z0 + np.float32(cells / 128.0)
        st["ppos"][..., 2] = st["x"][:, N // 3, 2][:, None]
        base_z = int(np.floor(st["x"][..., 2].min() * 128 - 0.5))
        of = orc.step_fwd(st, nthreads=2)
        for _ in range(2):
            oh = run_hip(sim, st)
            assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4, (cells, base_z & 31)
            assert _rel(oh["C"], of["C"]) < 1e-3 and _rel(oh["F"], of["F"]) < 5e-5, (cells, base_z & 31)


def test_grid_checkpoint_pool_overflow_falls_back_to_recompute(multi_kernel_path):
    """A pool of 1 record per particle and substep holds the compact rope (measured 0.58 active cells per particle) but not the
    same particles scattered through the volume (up to 27 cells each): the forward flags the env in status[], the host mirror
    sees the flag without a sync and asks that step's backward to recompute the grid (clip bit 1) -- same gradients as a
    handle that never checkpoints the grid."""
    sim, st, g, N = _scaled_case(3, 0, B=2, grid_ckpt_cells=1)
    run_hip(sim, st, g=g, clip=True)                                   # the rope fits
    assert sim.grid_ckpt_overflows == 0
    st["x"][1] = np.random.default_rng(2).uniform(0.1, 0.4, size=st["x"][1].shape).astype(np.float32)   # env 1 scattered
    got = run_hip(sim, st, g=g, clip=True)
    assert sim.grid_ckpt_overflows == 1
    ref_sim, _, _, _ = _scaled_case(3, 0, B=2, grid_ckpt_cells=0)
    ref = run_hip(ref_sim, st, g=g, clip=True)
    for key in ("x", "v", "gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(got[key]).all() and _rel(got[key], ref[key]) < 2e-5, (key, _rel(got[key], ref[key]))


# ---- soft contact (collide_batch, primitives.py:154-182) -- the mode shape_rope / pour_* use --------------------------
def make_collide_sim(steps, B, material=1, N=67):
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    conf = LegacyConf()
    conf.steps = steps
    sim = SimpleMPMSimulator(conf, B, use_position_control=False)
    sim.n_particles = N
    sim.material = np.full(N, material, np.int32)
    sim.h = np.ones(N, np.float32)
    sim._make_handle()
    return sim


def run_hip_collide(sim, st, g, clip):
    from unidom_amd.engine.mpm_simulator import _Step
    dev = sim.device
    t = lambda a, r=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=r)
    X, V, Cm, F, PP, PR, FR, MU, LA, AC = (t(st[k], True) for k in ("x", "v", "C", "F", "ppos", "prot", "friction", "mu", "lamda", "action"))
    sim.clip_grad = clip
    out = _Step.apply(sim, X, V, Cm, F, t(st["J"]), PP, PR, t(st["psize"]), FR, MU, LA, AC)
    res = {k: o.detach().cpu().numpy() for k, o in zip(("x", "v", "C", "F", "J", "ppos", "prot", "pv", "pw"), out)}
    loss = (out[0] * t(g["gx"])).sum() + (out[1] * t(g["gv"])).sum() + (out[2] * t(g["gC"])).sum() + \
        (out[3] * t(g["gF"])).sum() + (out[5] * t(g["gppos"])).sum() + (out[6] * t(g["gprot"])).sum()
    loss.backward()
    res.update(gx=X.grad.cpu().numpy(), gv=V.grad.cpu().numpy(), gC=Cm.grad.cpu().numpy(), gF=F.grad.cpu().numpy(),
               gppos=PP.grad.cpu().numpy(), gprot=PR.grad.cpu().numpy(), gfriction=FR.grad.cpu().numpy(),
               gmu=MU.grad.cpu().numpy(), glamda=LA.grad.cpu().numpy(), gaction=AC.grad.cpu().numpy())
    sim.check_status()
    return res


@pytest.mark.parametrize("clip,material,S,k", [(False, 1, 3, 40), (True, 1, 3, 40), (False, 2, 3, 20), (True, 1, 20, 40)])
def test_collide_matches_oracle(demo, clip, material, S, k):
    """Soft contact with a rotated, turning box: forward vs the f32 oracle, adjoint vs the f64 oracle."""
    from oracle.pyoracle import MpmOracle
    from test_oracle_mpm import _collide_case
    st, g = _collide_case(demo, S, k, material, 0, np.float32)
    st64 = {kk: v.astype(np.float64) for kk, v in st.items()}
    g64 = {kk: v.astype(np.float64) for kk, v in g.items()}
    orc = MpmOracle(67, steps=S, material=np.full(67, material), position_control=False)
    of, ob = orc.step_fwd(st), orc.step_bwd(st64, g64, clip=clip)
    oh = run_hip_collide(make_collide_sim(S, 1, material), st, g, clip)
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4, (_rel(oh["x"], of["x"]), _rel(oh["v"], of["v"]))
    assert _rel(oh["C"], of["C"]) < 1e-3 and _rel(oh["F"], of["F"]) < 5e-5
    for key in ("ppos", "prot", "pv", "pw"):
        np.testing.assert_allclose(oh[key], of[key], rtol=0, atol=2e-7)
    tol = 2e-3 if S <= 3 else 1e-2
    for key in ("gx", "gv", "gC", "gF", "gppos", "gprot", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        assert _rel(oh[key], ob[key]) < tol, (key, _rel(oh[key], ob[key]))
    for key in ("gfriction", "gmu", "glamda"):
        assert _rel(oh[key].reshape(-1), ob[key]) < 5 * tol, (key, oh[key], ob[key])
    assert np.abs(oh["gaction"][0, 3:]).min() > 0 and np.abs(oh["gprot"]).max() > 0


def test_collide_zero_rotation_action_laundered(demo):
    """w = 0 (every shape_rope action): NaN from d|w|/dw, zeroed by the step-boundary nan_to_num like the reference."""
    from test_oracle_mpm import _collide_case
    st, g = _collide_case(demo, 2, 40, 1, 1, np.float32, w=(0, 0, 0))
    raw = run_hip_collide(make_collide_sim(2, 1), st, g, clip=False)
    assert np.isnan(raw["gaction"][0, 3:]).all() and np.isfinite(raw["gaction"][0, :3]).all()
    assert np.isfinite(raw["gx"]).all() and np.isfinite(raw["gprot"]).all()
    lau = run_hip_collide(make_collide_sim(2, 1), st, g, clip=True)
    assert (lau["gaction"][0, 3:] == 0).all() and np.isfinite(lau["gaction"]).all()


def _shape_rope_case(S, B):
    """shape_rope's sizes (shape_rope_env.py: 582 plastic particles seeded like the goal lattice, 64x6x64 grid, dt 0.5e-4, ground
    friction 0.9, soft contact) with a pusher overlapping the rope; env b > 1 repeats env b % 2's geometry moved by a fraction of a
    cell, with its own v / C / F / action."""
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    from unidom_amd.envs.shape_rope_env import DefaultConf
    conf = DefaultConf()
    conf.steps = S
    N = 582
    sim = SimpleMPMSimulator(conf, B, use_position_control=False)
    sim.n_particles, sim.material, sim.h = N, np.full(N, 2, np.int32), np.ones(N, np.float32)
    assert sim.grid_ckpt_cells == conf.grid_ckpt_cells == 2          # the env runs with the grid checkpoint
    sim._make_handle()
    rng = np.random.default_rng(5)
    x0 = np.load(conf.goal_path).astype(np.float32)
    x = np.stack([x0, x0 + np.float32(0.003)]).astype(np.float32)
    x[..., 1] = x0[:, 1]
    ppos = np.zeros((2, S, 3), np.float32)
    ppos[:, 0] = x[:, 291] + np.array([[0.002, 0.0, -0.012], [-0.003, 0.001, 0.011]], np.float32)
    prot = np.zeros((2, S, 4), np.float32)
    q = np.array([[0.95, 0.05, 0.3, -0.02], [1, 0, 0, 0]], np.float32)
    prot[:] = (q / np.linalg.norm(q, axis=1, keepdims=True))[:, None]
    action = np.array([[0.002, 0, 0.012, 0.2, -0.1, 0.3], [-0.004, 0, -0.01, 0.1, 0.2, -0.2]], np.float32)
    if B > 2:
        rep = lambda a: np.ascontiguousarray(np.concatenate([a] * ((B + 1) // 2), 0)[:B])
        x, ppos, prot, action = rep(x), rep(ppos), rep(prot), rep(action)
        off = np.zeros((B, 1, 3), np.float32)
        off[2:, 0, 0] = rng.uniform(-0.004, 0.004, size=B - 2)      # up to half a cell (dx = 1/128) in x and z: other base cells,
        off[2:, 0, 2] = rng.uniform(-0.004, 0.004, size=B - 2)      # other weights, the same contact geometry
        x = x + off
        ppos[:, :1] += off
        action[2:] *= rng.uniform(0.5, 1.5, size=(B - 2, 6)).astype(np.float32)
    mu0, la0 = conf.E / (2 * (1 + conf.nu)), conf.E * conf.nu / ((1 + conf.nu) * (1 - 2 * conf.nu))
    st = dict(x=x, v=rng.normal(size=(B, N, 3)).astype(np.float32) * 0.05, C=rng.normal(size=(B, N, 3, 3)).astype(np.float32) * 2,
              F=(np.eye(3) + rng.normal(size=(B, N, 3, 3)) * 0.05).astype(np.float32), J=np.ones((B, N), np.float32), ppos=ppos, prot=prot,
              psize=np.tile(np.array([0.015, 0.06, 0.015], np.float32), (B, 1)), friction=np.full(B, 0.9, np.float32),
              mu=np.full(B, mu0, np.float32), lamda=np.full(B, la0, np.float32), action=action)
    g = dict(gx=rng.normal(size=(B, N, 3)), gv=rng.normal(size=(B, N, 3)) * 1e-3, gC=rng.normal(size=(B, N, 3, 3)) * 1e-6,
             gF=rng.normal(size=(B, N, 3, 3)) * 1e-2, gppos=rng.normal(size=(B, S, 3)), gprot=rng.normal(size=(B, S, 4)))
    return sim, conf, st, {k: v.astype(np.float32) for k, v in g.items()}, N


def test_collide_shape_rope_geometry_fwd_bwd(large_path):
    """shape_rope's sizes (582 plastic particles, 64x6x64 grid, dt 0.5e-4, ground friction 0.9; 4 lanes per particle on
    the many-workgroup path): a pusher overlapping the rope -- forward vs the f32 oracle, adjoint vs the f64 oracle."""
    from oracle.pyoracle import MpmOracle
    S, B = 4, 2
    sim, conf, st, g, N = _shape_rope_case(S, B)
    orc = MpmOracle(N, n_grid=128, res=(64, 6, 64), steps=S, dt=conf.dt, position_control=False, material=np.full(N, 2))
    of = orc.step_fwd(st, nthreads=2)
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()},
                      clip=False, nthreads=2)
    oh = run_hip_collide(sim, st, g, clip=False)
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4, (_rel(oh["x"], of["x"]), _rel(oh["v"], of["v"]))
    assert _rel(oh["C"], of["C"]) < 1e-3 and _rel(oh["F"], of["F"]) < 5e-5
    for key in ("gx", "gv", "gC", "gF", "gppos", "gprot", "gaction"):
        assert np.isfinite(oh[key]).all(), key
        assert _rel(oh[key], ob[key]) < 5e-3, (key, _rel(oh[key], ob[key]))
    assert np.abs(ob["gprot"]).max() > 0 and np.abs(ob["gaction"][:, 3:]).min() > 0


def _no_path_override():
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    return SimpleMPMSimulator.default_tuning == {}


def test_bench_launch_shape_rope_n_grid_128_32_envs_default_kernels_match_oracle():
    """The launch bench.py --workload whip_rope --n-grid 128 times: 32 envs x 798 particles in ONE step call, no switch set -- the
    persistent cluster forward (32 envs x 25 parts of 32 particles = 800 workgroups that must all be resident and meet at the
    inter-workgroup barrier every substep: what a 2-env launch cannot exercise) writing the grid checkpoint, and the two-launch
    backward restoring from it.  Every env has its own v / C / F and a position moved by a fraction of a cell; two of them are
    followed by the CPU oracle (forward f32, adjoint f64) over 5 substeps; check_status() clean."""
    from oracle.pyoracle import MpmOracle
    assert _no_path_override()
    S, B, pick = 5, 32, [5, 30]
    sim, st, g, N = _scaled_case(S, 17, B=B, grid_ckpt_cells=2)       # bench.py's --grid-ckpt default
    assert N == 798 and sim.launch_plan(B) == 7                         # many-workgroup, persistent forward, two-launch backward
    rng = np.random.default_rng(3)
    off = np.zeros((B, 1, 3), np.float32)
    off[:, 0, 0], off[:, 0, 2] = rng.uniform(-0.004, 0.004, size=B), rng.uniform(-0.004, 0.004, size=B)
    st["x"] = st["x"] + off
    st["ppos"][:, :1] += off
    st["action"] = (st["action"] * rng.uniform(-1.5, 1.5, size=(B, 6))).astype(np.float32)
    sub = lambda d: {k: np.ascontiguousarray(v[pick]) for k, v in d.items()}
    orc = MpmOracle(N, n_grid=128, res=(64, 64, 64), steps=S)
    of = orc.step_fwd(sub(st), nthreads=2)
    assert all(np.isfinite(of[k]).all() for k in ("x", "v", "C", "F"))
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in sub(st).items()},
                      {k: v.astype(np.float64) for k, v in sub(g).items()}, clip=True, nthreads=2)
    oh = run_hip(sim, st, g=g, clip=True)
    sim.check_status()
    assert sim.grid_ckpt_overflows == 0
    assert _rel(oh["x"][pick], of["x"]) < 5e-6 and _rel(oh["v"][pick], of["v"]) < 1e-4     # north_star: 1e-4 relative
    assert _rel(oh["C"][pick], of["C"]) < 1e-3 and _rel(oh["F"][pick], of["F"]) < 5e-5
    for key in ("ppos", "prot", "pv", "pw"):
        np.testing.assert_allclose(oh[key][pick], of[key], rtol=0, atol=1e-7)
    for key in ("x", "v", "C", "F", "gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(oh[key]).all(), key
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert _rel(oh[key][pick], ob[key]) < 5e-3, (key, _rel(oh[key][pick], ob[key]))   # tolerance of the 2-env test
    for key in ("gfriction", "gmu", "glamda"):
        assert _rel(oh[key].reshape(-1)[pick], ob[key]) < 2e-2, (key, oh[key].reshape(-1)[pick], ob[key])
    # the same handle again: rotating grids, owner stamps and barrier words must be back in their initial state
    oh2 = run_hip(sim, st)
    assert _rel(oh2["x"][pick], of["x"]) < 5e-6 and _rel(oh2["v"][pick], of["v"]) < 1e-4


def test_bench_launch_shape_shape_rope_32_envs_default_kernels_match_oracle():
    """The launch bench.py --workload shape_rope times: 32 envs x 582 plastic particles, soft contact, grid checkpoint 2, no switch
    set -- cluster forward (32 x 19 parts) + two-launch backward in four env groups.  Two envs followed by the CPU oracle."""
    from oracle.pyoracle import MpmOracle
    assert _no_path_override()
    S, B, pick = 5, 32, [3, 28]
    sim, conf, st, g, N = _shape_rope_case(S, B)
    assert sim.launch_plan(B) == 7
    sub = lambda d: {k: np.ascontiguousarray(v[pick]) for k, v in d.items()}
    orc = MpmOracle(N, n_grid=128, res=(64, 6, 64), steps=S, dt=conf.dt, position_control=False, material=np.full(N, 2))
    of = orc.step_fwd(sub(st), nthreads=2)
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in sub(st).items()}, {k: v.astype(np.float64) for k, v in sub(g).items()},
                      clip=False, nthreads=2)
    oh = run_hip_collide(sim, st, g, clip=False)
    sim.check_status()
    assert sim.grid_ckpt_overflows == 0
    assert _rel(oh["x"][pick], of["x"]) < 5e-6 and _rel(oh["v"][pick], of["v"]) < 1e-4
    assert _rel(oh["C"][pick], of["C"]) < 1e-3 and _rel(oh["F"][pick], of["F"]) < 5e-5
    for key in ("x", "v", "C", "F", "gx", "gv", "gC", "gF", "gppos", "gprot"):
        assert np.isfinite(oh[key]).all(), key
    for key in ("gx", "gv", "gC", "gF", "gppos", "gprot", "gaction"):
        assert _rel(oh[key][pick], ob[key]) < 5e-3, (key, _rel(oh[key][pick], ob[key]))


@pytest.mark.parametrize("clip", [False, True])
def test_two_container_primitives_match_oracle(demo, clip, large_path):
    """n_primitive = 2, container SDF, liquid material (pour_water's configuration): forward vs the f32 oracle, adjoint
    (every primitive's position / rotation rows, the 12 action components) vs the f64 oracle."""
    from oracle.pyoracle import MpmOracle
    from test_oracle_mpm import _two_bowl_case
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator, _Step
    S, N = 3, 67
    st, g = _two_bowl_case(demo, S, 40, 0, np.float32)
    conf = LegacyConf()
    conf.steps = S
    sim = SimpleMPMSimulator(conf, 1, use_position_control=False)
    sim.n_particles, sim.material, sim.h = N, np.zeros(N, np.int32), np.ones(N, np.float32)
    sim.n_primitive, sim.sdf_kind = 2, "container"
    sim._make_handle()
    orc = MpmOracle(N, steps=S, material=np.zeros(N), position_control=False, n_prim=2, sdf="container")
    st64, g64 = {k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}
    of, of64 = orc.step_fwd(st), orc.step_fwd(st64)
    ob, ob32 = orc.step_bwd(st64, g64, clip=clip), orc.step_bwd(st, g, clip=clip)
    dev = sim.device
    t = lambda a, r=False: torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=r)
    X, V, Cm, F, PP, PR, FR, MU, LA, AC = (t(st[k], True) for k in ("x", "v", "C", "F", "ppos", "prot", "friction", "mu", "lamda", "action"))
    sim.clip_grad = clip
    out = _Step.apply(sim, X, V, Cm, F, t(st["J"]), PP, PR, t(st["psize"]), FR, MU, LA, AC)
    oh = {k: o.detach().cpu().numpy() for k, o in zip(("x", "v", "C", "F", "J", "ppos", "prot", "pv", "pw"), out)}
    loss = (out[0] * t(g["gx"])).sum() + (out[1] * t(g["gv"])).sum() + (out[2] * t(g["gC"])).sum() + \
        (out[3] * t(g["gF"])).sum() + (out[5] * t(g["gppos"])).sum() + (out[6] * t(g["gprot"])).sum()
    loss.backward()
    sim.check_status()
    # The finite-difference normal (d = 1e-6, primitives.py:119-134) of the container's nested square roots carries ~1 %
    # round-off noise in f32, and the turning bowl's pose comes from sinf / cosf that differ by an ulp between libraries:
    # the f32 restatement itself sits this far from the f64 one.  Criterion: the kernel is as close to f64 as the f32 oracle.
    base = dict(x=5e-6, v=1e-4, C=1e-3, F=5e-5)
    for key in ("x", "v", "C", "F"):
        gap = _rel(of[key], of64[key])
        assert _rel(oh[key], of64[key]) < 3 * gap + base[key], (key, _rel(oh[key], of64[key]), gap)
    for key in ("ppos", "prot", "pv", "pw"):
        assert oh[key].shape == of[key].shape == (1, 2, S, 4 if key == "prot" else 3)
        np.testing.assert_allclose(oh[key], of[key], rtol=0, atol=2e-7)
    got = dict(gx=X.grad, gv=V.grad, gC=Cm.grad, gF=F.grad, gppos=PP.grad, gprot=PR.grad, gaction=AC.grad)
    for key, val in got.items():
        val = val.cpu().numpy()
        assert np.isfinite(val).all(), key
        gap = _rel(ob32[key], ob[key])
        assert _rel(val, ob[key]) < 3 * gap + 2e-3, (key, _rel(val, ob[key]), gap)
    assert _rel(FR.grad.cpu().numpy().reshape(-1), ob["gfriction"]) < 3 * _rel(ob32["gfriction"], ob["gfriction"]) + 1e-2
    assert np.abs(ob["gaction"]).min() > 0 and np.abs(got["gprot"].cpu().numpy()[0, 1]).max() > 0


def test_two_primitives_with_their_own_friction_and_softness(demo, multi_kernel_path):
    """ud_mpm_conf.prim_friction_each / prim_softness_each: one pair per primitive (create_primitive passes them per primitive,
    mpm_env.py:201-217).  Upright, translating bowls with friction 0.1 / 0.45 and softness 666 / 120 against the oracle with the same
    pairs, forward (f32) and adjoint (f64); the uniform handle gives a different answer."""
    from oracle.pyoracle import MpmOracle
    from test_oracle_mpm import _two_bowl_case
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    S, N = 3, 67
    st, g = _two_bowl_case(demo, S, 40, 0, np.float32, turning=False)
    fr_each, so_each = [0.1, 0.45], [666.0, 120.0]

    def make(each):
        conf = LegacyConf()
        conf.steps = S
        sim = SimpleMPMSimulator(conf, 1, use_position_control=False)
        sim.n_particles, sim.material, sim.h = N, np.zeros(N, np.int32), np.ones(N, np.float32)
        sim.n_primitive, sim.sdf_kind = 2, "container"
        if each:
            sim.prim_friction_each, sim.prim_softness_each = fr_each, so_each
        sim._make_handle()
        return sim
    orc = MpmOracle(N, steps=S, material=np.zeros(N), position_control=False, n_prim=2, sdf="container", prim_friction=fr_each, prim_softness=so_each)
    st64, g64 = {k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}
    of, of64 = orc.step_fwd(st), orc.step_fwd(st64)
    ob, ob32 = orc.step_bwd(st64, g64, clip=False), orc.step_bwd(st, g, clip=False)
    got = run_hip_collide(make(True), st, g, False)
    uni = run_hip_collide(make(False), st, g, False)
    base = dict(x=5e-6, v=1e-4, C=1e-3, F=5e-5)
    for key in ("x", "v", "C", "F"):
        gap = _rel(of[key], of64[key])
        assert _rel(got[key], of64[key]) < 3 * gap + base[key], (key, _rel(got[key], of64[key]), gap)
    assert _rel(uni["v"], of64["v"]) > 10 * (_rel(got["v"], of64["v"]) + 1e-7)       # the pairs matter
    lin = [0, 1, 2, 6, 7, 8]          # the bowls do not turn: d|w|/dw at w = 0 is NaN without the clip, here as in the reference (primitives.py:86)
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        a, b, b32 = (v[:, lin] if key == "gaction" else v for v in (got[key], ob[key], ob32[key]))
        gap = _rel(b32, b)
        assert np.isfinite(a).all() and _rel(a, b) < 3 * gap + 2e-3, (key, _rel(a, b), gap)
    assert np.isnan(got["gaction"][:, [3, 4, 5, 9, 10, 11]]).all() and np.isnan(ob["gaction"][:, [3, 4, 5, 9, 10, 11]]).all()


def test_two_upright_containers_forward_tracks_f32_oracle(demo):
    """Same two bowls, upright and only translating: no library sin / cos in the poses, so the SDF + finite-difference
    normal path is operation-for-operation the oracle's and the forward agrees at the usual tolerances."""
    from oracle.pyoracle import MpmOracle
    from test_oracle_mpm import _two_bowl_case
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator, _Step
    S, N = 3, 67
    st, _ = _two_bowl_case(demo, S, 40, 0, np.float32, turning=False)
    conf = LegacyConf()
    conf.steps = S
    sim = SimpleMPMSimulator(conf, 1, use_position_control=False)
    sim.n_particles, sim.material, sim.h = N, np.zeros(N, np.int32), np.ones(N, np.float32)
    sim.n_primitive, sim.sdf_kind = 2, "container"
    sim._make_handle()
    of = MpmOracle(N, steps=S, material=np.zeros(N), position_control=False, n_prim=2, sdf="container").step_fwd(st)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=sim.device)
    with torch.no_grad():
        out = _Step.apply(sim, *[t(st[k]) for k in ("x", "v", "C", "F", "J", "ppos", "prot", "psize", "friction", "mu", "lamda", "action")])
    oh = {k: o.cpu().numpy() for k, o in zip(("x", "v", "C", "F"), out)}
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4, (_rel(oh["x"], of["x"]), _rel(oh["v"], of["v"]))
    assert _rel(oh["C"], of["C"]) < 1e-3 and _rel(oh["F"], of["F"]) < 5e-5


@pytest.mark.parametrize("case", ["one_substep", "n129_just_over_the_one_workgroup_limit", "body_at_the_domain_corner", "four_box_primitives",
                                  "one_workgroup_one_substep", "one_workgroup_domain_corner",
                                  "across_the_upper_grid_edge", "one_workgroup_across_the_upper_grid_edge",
                                  "negative_weights", "one_workgroup_negative_weights"])
def test_mpm_step_edge_cases(demo, case, large_path):
    """Edges vs the oracle (forward f32, adjoint f64), on the many-workgroup path and (one_workgroup_*) the one-workgroup
    path: a single substep per step (copy_frame and the primitive recurrences degenerate, Q5), the smallest particle count
    that takes the many-workgroup path, a body pressed into the domain corner (base truncates to 0, every cell sits in the
    friction and boundary bands of all three axes), and the maximum of four primitives in soft-contact mode."""
    from oracle.pyoracle import MpmOracle
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    rng = np.random.default_rng(11)
    S, N, P, pc = 4, 160, 1, True
    if case.startswith("one_workgroup"):                             # N <= 128, position control: the one-workgroup kernels
        N = 100
        if large_path not in (None, "cluster_128"):
            pytest.skip("the one-workgroup path has no cluster / multi-kernel variants")
    if case.endswith("one_substep"):
        S = 1
    elif case == "n129_just_over_the_one_workgroup_limit":
        N = 129
    elif case == "four_box_primitives":
        P, pc = 4, False
    conf = LegacyConf()
    conf.steps = S
    sim = SimpleMPMSimulator(conf, 2, use_position_control=pc)
    sim.n_particles, sim.material, sim.h = N, np.full(N, 1, np.int32), np.ones(N, np.float32)
    sim.n_primitive = P
    sim.grid_ckpt_cells = 0 if case.endswith(("domain_corner", "negative_weights")) else 8
    sim._make_handle()
    B = 2
    lo = 0.004 if case.endswith("domain_corner") else 0.15            # dx = 1/64: x * inv_dx < 0.5 below 0.0078
    x = (lo + rng.uniform(0, 0.06, size=(B, N, 3))).astype(np.float32)
    if case.endswith("upper_grid_edge"):      # res = 32 cells of dx = 1/64 end at 0.5: stencil cells beyond it are dropped by the
        x[..., 0] += np.float32(0.30)          # scatter and clamped by the gather (Q5 / Q9); x in [0.45, 0.51]
    if case.endswith("negative_weights"):     # x * inv_dx < 0.134: w[1] = 0.75 - (fx - 1)^2 < 0, cells with m < 0 (Q13)
        x[..., 1] = (0.0002 + rng.uniform(0, 0.004, size=(B, N))).astype(np.float32)
    pa = (P,) if P > 1 else ()
    ppos = np.zeros((B,) + pa + (S, 3), np.float32)
    ppos[..., 0, :] = (x.mean(1)[:, None] if P > 1 else x.mean(1)) + rng.normal(size=(B,) + pa + (3,)).astype(np.float32) * 0.01
    prot = np.zeros((B,) + pa + (S, 4), np.float32)
    prot[..., 0] = 1
    mu0, la0 = 100 / (2 * 1.1), 100 * 0.1 / (1.1 * 0.8)
    st = dict(x=x, v=(rng.normal(size=(B, N, 3)) * 0.3).astype(np.float32), C=(rng.normal(size=(B, N, 3, 3)) * 2).astype(np.float32),
              F=(np.eye(3) + rng.normal(size=(B, N, 3, 3)) * 0.03).astype(np.float32), J=np.ones((B, N), np.float32), ppos=ppos, prot=prot,
              psize=np.tile(np.float32([0.02, 0.02, 0.02]), (B,) + pa + (1,)), friction=np.full(B, 0.3, np.float32),
              mu=np.full(B, mu0, np.float32), lamda=np.full(B, la0, np.float32),
              action=(rng.normal(size=(B, 6 * P)) * 0.01).astype(np.float32))
    g = dict(gx=rng.normal(size=(B, N, 3)), gv=rng.normal(size=(B, N, 3)) * 0.01, gC=rng.normal(size=(B, N, 3, 3)) * 1e-4,
             gF=rng.normal(size=(B, N, 3, 3)) * 0.01, gppos=rng.normal(size=(B,) + pa + (S, 3)), gprot=rng.normal(size=(B,) + pa + (S, 4)))
    g = {k: v.astype(np.float32) for k, v in g.items()}
    if pc:
        del g["gprot"]        # position control: no rotation cotangent crosses the boundary (run_hip does not feed one either)
    orc = MpmOracle(N, steps=S, position_control=pc, n_prim=P)
    of = orc.step_fwd(st)
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}, clip=True)
    oh = run_hip_collide(sim, st, g, clip=True) if not pc else run_hip(sim, st, g=g, clip=True)
    assert _rel(oh["x"], of["x"]) < 5e-6 and _rel(oh["v"], of["v"]) < 1e-4, (_rel(oh["x"], of["x"]), _rel(oh["v"], of["v"]))
    assert _rel(oh["C"], of["C"]) < 1e-3 and _rel(oh["F"], of["F"]) < 5e-5
    for key in ("ppos", "prot", "pv", "pw"):
        np.testing.assert_allclose(oh[key], of[key], rtol=0, atol=2e-7)
    keys = ("gx", "gv", "gC", "gF", "gppos", "gaction") + (() if pc else ("gprot",))
    for key in keys:
        assert np.isfinite(oh[key]).all(), key
        assert _rel(oh[key], ob[key]) < 5e-3, (key, _rel(oh[key], ob[key]))
    if case.endswith("domain_corner"):
        assert (x * 64 < 0.5).any()                                      # base truncates to 0 with fx < 0.5 for some particles
    if case.endswith("upper_grid_edge"):
        assert ((x[..., 0] * 64 - 0.5).astype(np.int32) + 2 >= 32).any() and ((x[..., 0] * 64 - 0.5).astype(np.int32) + 2 < 32).any()
    if case.endswith("negative_weights"):
        assert (x[..., 1] * 64 < 0.134).any()


@pytest.mark.parametrize("grid_ckpt_cells", [0, 8])
def test_internal_spatial_order_is_invisible(demo, grid_ckpt_cells):
    """ud_mpm_conf.sort_particles: a body whose particles arrive shuffled (mixed materials and hardness, so the per-particle
    tables and the Q6 trace over the caller's particles 0..2 must follow the permutation) gives the same step and the same
    gradients, in the caller's order, with and without the internal re-ordering -- and both match the oracle."""
    from oracle.pyoracle import MpmOracle
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    rng = np.random.default_rng(21)
    S, N, B = 4, 300, 2
    mat = rng.integers(1, 3, size=N).astype(np.int32)                  # elastic and plastic particles interleaved
    hard = rng.uniform(0.5, 2.0, size=N).astype(np.float32)
    x = (0.2 + rng.uniform(0, 0.09, size=(B, N, 3))).astype(np.float32)   # no spatial order at all
    ppos = np.zeros((B, S, 3), np.float32)
    ppos[:, 0] = x.mean(1)
    prot = np.zeros((B, S, 4), np.float32)
    prot[..., 0] = 1
    mu0, la0 = 100 / (2 * 1.1), 100 * 0.1 / (1.1 * 0.8)
    st = dict(x=x, v=(rng.normal(size=(B, N, 3)) * 0.2).astype(np.float32), C=(rng.normal(size=(B, N, 3, 3)) * 2).astype(np.float32),
              F=(np.eye(3) + rng.normal(size=(B, N, 3, 3)) * 0.03).astype(np.float32), J=np.ones((B, N), np.float32), ppos=ppos, prot=prot,
              psize=np.tile(np.float32([0.02, 0.02, 0.02]), (B, 1)), friction=np.full(B, 0.3, np.float32),
              mu=np.full(B, mu0, np.float32), lamda=np.full(B, la0, np.float32), action=(rng.normal(size=(B, 6)) * 0.01).astype(np.float32))
    g = dict(gx=rng.normal(size=(B, N, 3)), gv=rng.normal(size=(B, N, 3)) * 0.01, gC=rng.normal(size=(B, N, 3, 3)) * 1e-4,
             gF=rng.normal(size=(B, N, 3, 3)) * 0.01, gppos=rng.normal(size=(B, S, 3)))
    g = {k: v.astype(np.float32) for k, v in g.items()}
    outs = []
    for sort in (0, 1):
        conf = LegacyConf()
        conf.steps = S
        sim = SimpleMPMSimulator(conf, B, use_position_control=True)
        sim.n_particles, sim.material, sim.h = N, mat, hard
        sim.grid_ckpt_cells, sim.sort_particles = grid_ckpt_cells, sort
        sim._make_handle()
        outs.append(run_hip(sim, st, g=g, clip=True))
        outs.append(run_hip(sim, st))                                   # the no-checkpoint forward keeps its order in the handle
    plain, plain_nograd, srt, srt_nograd = outs
    for key in ("x", "v", "C", "F", "J", "gx", "gv", "gC", "gF", "gppos", "gaction", "gmu", "glamda", "gfriction"):
        assert _rel(srt[key], plain[key]) < 2e-5, (key, _rel(srt[key], plain[key]))     # only the atomics' summation order differs
    for key in ("x", "v", "C", "F", "J"):
        assert _rel(srt_nograd[key], plain_nograd[key]) < 2e-5, key
    orc = MpmOracle(N, steps=S, material=mat, hardness=hard)
    of = orc.step_fwd(st)
    ob = orc.step_bwd({k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}, clip=True)
    assert _rel(srt["x"], of["x"]) < 5e-6 and _rel(srt["v"], of["v"]) < 1e-4 and _rel(srt["F"], of["F"]) < 5e-5
    np.testing.assert_allclose(srt["J"], of["J"], rtol=1e-5)           # Q6: the trace runs over the caller's particles 0, 1, 2
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert _rel(srt[key], ob[key]) < 5e-3, (key, _rel(srt[key], ob[key]))


def test_cluster_call_cut_into_several_launches(monkeypatch):
    """The parts of a cluster launch wait for each other, so a call with more envs than fit on the chip at once is cut into
    launches on the caller's stream (csrc/mpm_large.hip: clm_envs_per_launch).  tune_cluster_envs caps the envs per launch:
    5 envs as 2 + 2 + 1 must give what one launch gives, forward and adjoint, whichever launch an env was in, and a second call
    on the same handle must find the rotating grids at rest."""
    _tune(monkeypatch, cluster=1)
    sim, st, g, N = _scaled_case(4, 7, B=5, grid_ckpt_cells=2)
    for b in range(5):
        st["action"][b] = np.float32([0.1 * b - 0.2, 0.05 * b, 0.3 - 0.1 * b, 0, 0, 0]) / 50
    one = run_hip(sim, st, g=g, clip=True)
    _tune(monkeypatch, cluster=1, cluster_envs=2)
    sim2, _, _, _ = _scaled_case(4, 7, B=5, grid_ckpt_cells=2)
    assert sim.launch_plan(5) == sim2.launch_plan(5) == 7
    for _ in range(2):
        cut = run_hip(sim2, st, g=g, clip=True)
        for key in ("x", "v", "C", "F", "J"):
            assert _rel(cut[key], one[key]) < 2e-6, (key, _rel(cut[key], one[key]))
        for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
            assert np.isfinite(cut[key]).all() and _rel(cut[key], one[key]) < 1e-4, (key, _rel(cut[key], one[key]))


@pytest.mark.parametrize("part_lanes", [64, 128])
@pytest.mark.parametrize("grid_ckpt_cells", [0, 2])
def test_cluster_scattered_cloud_outgrows_no_table_or_spills_and_stays_right(part_lanes, grid_ckpt_cells, monkeypatch):
    """A rope whose particles are thrown uniformly through the volume (a plastic body that tears, a whip that scatters).  Parts of 16
    particles (64 lanes) touch at most 16 x 27 = 432 cells: their 512-slot tables cannot overflow.  Parts of 32 particles (128 lanes,
    the default for solids) can touch 864: what finds no slot goes to the env's HBM grid directly, is read back and put through the grid
    op in the gather, and is zeroed from the part's spill list (csrc/mpm_cluster.h::clm_scatter) -- the step stays VALID; with a grid
    checkpoint the env is flagged (its checkpoint misses cells) and that step's backward recomputes the grid.  Result = the multi-kernel
    path's, which sends what does not fit its tables to HBM atomics too; twice on the same handle (everything back at rest)."""
    _tune(monkeypatch, cluster=-1)
    ref_sim, st, g, N = _scaled_case(3, 0, B=2, grid_ckpt_cells=grid_ckpt_cells)
    st["x"][1] = np.random.default_rng(2).uniform(0.1, 0.4, size=st["x"][1].shape).astype(np.float32)   # env 1 scattered
    ref = run_hip(ref_sim, st, g=g, clip=True)
    _tune(monkeypatch, cluster=1, cluster_part_lanes=part_lanes)
    sim, _, _, _ = _scaled_case(3, 0, B=2, grid_ckpt_cells=grid_ckpt_cells)
    assert sim.launch_plan(2) & 2
    for rep in range(2):
        got = run_hip(sim, st, g=g, clip=True)          # run_hip ends with check_status(): no failure flag
        for key in ("x", "v", "C", "F"):
            assert _rel(got[key], ref[key]) < 2e-6, (rep, key, _rel(got[key], ref[key]))
        for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
            assert np.isfinite(got[key]).all() and _rel(got[key], ref[key]) < 1e-4, (rep, key, _rel(got[key], ref[key]))
    if grid_ckpt_cells and part_lanes == 128:
        assert sim.grid_ckpt_overflows == 2               # both calls: env 1 spilled, its backward recomputed the grid


def test_cluster_forward_grid_checkpoint_overflow_falls_back_to_recompute(monkeypatch):
    """The cluster forward writes the grid checkpoint the multi-kernel backward restores from (one record per active cell, by the
    part that owns the cell, at a position drawn from the env's record counter).  A pool of 1 record per particle and substep
    holds the compact rope but not the same particles scattered through the volume: the env is flagged in status[] (bit 0), the
    host mirror sees it without a sync and asks that step's backward to recompute the grid (clip bit 1) -- same gradients as a
    handle that never checkpoints the grid.  (Parts of 16 particles: a scattered cloud cannot outgrow their cell tables.)"""
    _tune(monkeypatch, cluster=1, cluster_part_lanes=64)
    sim, st, g, N = _scaled_case(3, 0, B=2, grid_ckpt_cells=1)
    run_hip(sim, st, g=g, clip=True)                                   # the rope fits
    assert sim.grid_ckpt_overflows == 0
    st["x"][1] = np.random.default_rng(2).uniform(0.1, 0.4, size=st["x"][1].shape).astype(np.float32)   # env 1 scattered
    got = run_hip(sim, st, g=g, clip=True)
    assert sim.grid_ckpt_overflows == 1
    ref_sim, _, _, _ = _scaled_case(3, 0, B=2, grid_ckpt_cells=0)
    ref = run_hip(ref_sim, st, g=g, clip=True)
    for key in ("x", "v", "gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(got[key]).all() and _rel(got[key], ref[key]) < 2e-5, (key, _rel(got[key], ref[key]))


@pytest.mark.parametrize("S", [1, 2, 5, 8])
@pytest.mark.parametrize("forward", ["cluster", "multi_kernel"])
def test_two_launch_backward_is_the_four_kernel_backward(S, forward, monkeypatch):
    """With the grid checkpoint the backward runs two launches per reverse substep where four lanes work on a particle and one primitive
    touches the grid (lg_gadj_restore, lg_padj_gadj; the cotangent grids of odd and even substeps in two arrays), the four-kernel
    sequence elsewhere (tune_bwd_two_launch = -1 forces it).  Same arithmetic per particle and per cell -- the float atomics' order is
    the only difference -- for odd and even numbers of substeps, a single one included, behind either forward; each handle runs twice,
    so each form must hand both cotangent arrays back all-zero."""
    cl = 1 if forward == "cluster" else -1
    _tune(monkeypatch, cluster=cl, bwd_two_launch=-1)
    sim4, st, g, N = _scaled_case(S, 4, B=3, grid_ckpt_cells=6)
    _tune(monkeypatch, cluster=cl)
    sim2, _, _, _ = _scaled_case(S, 4, B=3, grid_ckpt_cells=6)
    assert sim4.launch_plan(3) & 4 == 0 and sim2.launch_plan(3) & 4 == 4
    ref = run_hip(sim4, st, g=g, clip=True)
    for name, sim in (("two-launch", sim2), ("two-launch again", sim2), ("four-kernel again", sim4)):
        r = run_hip(sim, st, g=g, clip=True)
        for key in ("gx", "gv", "gC", "gF", "gppos", "gaction", "gfriction", "gmu", "glamda"):
            assert np.isfinite(r[key]).all() and _rel(r[key], ref[key]) < 2e-5, (key, name, _rel(r[key], ref[key]))


@pytest.mark.parametrize("forward", ["cluster", "multi_kernel"])
@pytest.mark.parametrize("material", [1, 2])
def test_backward_reads_the_svd_factors_the_forward_checkpointed(forward, material, monkeypatch):
    """In the four-lane regime every history record of the many-workgroup path carries the SVD factors (U, S, Vh) of its substep's F beside
    the state, and the backward's pre-pass reads them instead of running the Jacobi iteration again; the one-lane kernels (tune_lanes = 1:
    the regime of full launches) keep no factors and iterate.  Same code on the same inputs produced the factors, so the per-particle
    arithmetic is the same; only the lane mapping and the float atomics' order separate the two runs.  Plastic material too (the clamp
    uses the raw singular values), behind either forward."""
    def build(**kw):
        _tune(monkeypatch, **kw)
        sim, st, g, N = _scaled_case(6, 5, B=2, grid_ckpt_cells=6)
        if material == 2:
            sim.material = np.full(N, 2, np.int32)
            sim._make_handle()
        return sim, st, g
    sim1, st, g = build(cluster=-1, lanes=1)
    ref = run_hip(sim1, st, g=g, clip=True)
    sim4, _, _ = build(cluster=1 if forward == "cluster" else -1)
    got = run_hip(sim4, st, g=g, clip=True)
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction", "gmu", "glamda"):
        assert np.isfinite(got[key]).all() and _rel(got[key], ref[key]) < 2e-5, (key, _rel(got[key], ref[key]))


def test_launch_plan_reports_the_kernels_a_call_runs(monkeypatch):
    """ud_mpm_launch_plan (for logs and bench labels): 0 = one workgroup per env; bit 0 many-workgroup path, bit 1 persistent forward,
    bit 2 two-launch backward -- the library's own rules, or what ud_mpm_conf.tune_* fixed at create."""
    assert _no_path_override()
    assert make_sim(5, 2).launch_plan(2) == 0                               # whip_rope's 67 particles
    sim, _, _, _ = _scaled_case(3, 0, B=2, grid_ckpt_cells=2)               # rope at n_grid 128: solid, one primitive, four lanes
    assert sim.launch_plan(2) == 1 | 2 | 4
    assert sim.launch_plan(3) == -1                                          # more envs than the handle was created for
    _tune(monkeypatch, cluster=-1)
    assert _scaled_case(3, 0, B=2, grid_ckpt_cells=2)[0].launch_plan(2) == 1 | 4
    _tune(monkeypatch, cluster=-1, bwd_two_launch=-1)
    assert _scaled_case(3, 0, B=2, grid_ckpt_cells=2)[0].launch_plan(2) == 1
    _tune(monkeypatch)
    sim0, _, _, _ = _scaled_case(3, 0, B=2, grid_ckpt_cells=0)              # no grid checkpoint: the backward recomputes (six kernels)
    assert sim0.launch_plan(2) == 1 | 2
    big, _, _, _ = _scaled_case(3, 0, B=200, grid_ckpt_cells=2)
    assert big.launch_plan(200) == 1                                         # 200 x 798 particles: one lane per particle, neither rule applies


@pytest.mark.parametrize("grid_ckpt_cells", [0, 2])
def test_reset_leaves_a_healthy_handle_as_it_was(grid_ckpt_cells):
    """ud_mpm_reset (what the mirror calls after status bit 4, a part that gave up at the env barrier) rewrites the handle's rest state --
    barrier words, exchange grids, record counters, bitmaps.  On a healthy handle that state is already there: the step and its gradients
    before and after a reset agree to the float-atomics spread of two plain runs (test_large_path_matches_oracle_n798 pins this handle
    shape to the oracle)."""
    import ctypes as C
    from unidom_amd import _lib
    sim, st, g, N = _scaled_case(3, 0, B=2, grid_ckpt_cells=grid_ckpt_cells)
    assert sim.launch_plan(2) & 2                                            # the persistent cluster forward: the path with barrier words
    a = run_hip(sim, st, g)
    stream = C.c_void_p(torch.cuda.current_stream(sim.device).cuda_stream)
    for _ in range(2):                                                       # idempotent
        _lib.check(_lib.lib().ud_mpm_reset(sim._h, stream), "ud_mpm_reset")
    b = run_hip(sim, st, g)
    for k in a:
        scale = max(1e-6, float(np.abs(a[k]).max()))
        assert np.abs(a[k] - b[k]).max() <= 2e-4 * scale, k


def test_more_envs_than_the_handle_was_created_for_is_refused():
    """Every arena is sized at ud_mpm_create (max_envs): a call with more envs returns UD_ERR_INVALID instead of growing anything."""
    from unidom_amd._lib import UnidomError
    sim, st, g, N = _scaled_case(2, 0, B=2)
    big = {k: np.concatenate([v, v[:1]]) for k, v in st.items()}
    with pytest.raises(UnidomError):
        run_hip(sim, big)
    run_hip(sim, st)                                                         # the handle is still good
