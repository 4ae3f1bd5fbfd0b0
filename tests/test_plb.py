"""PlasticineLab-style f64 MPM (GenORM Torus, BASELINE config 5).  PARITY UNPINNED by reference data (taichi is
absent, no recording ships): the C++ restatement is checked against the literal NumPy twin and analytic known
answers (CPU), and the HIP kernels against the C++ restatement (GPU)."""
import numpy as np
import pytest

from oracle.pyoracle import PlbOracle
from oracle.twin.plb_twin import PlbConf, PlbTwin, torus_particles


def _case(N, seed=0):
    rng = np.random.default_rng(seed)
    x = torus_particles(1000)[:N].copy()
    v = rng.normal(size=(N, 3)) * 0.01
    Cm = rng.normal(size=(N, 3, 3)) * 0.1
    F = np.eye(3)[None] + rng.normal(size=(N, 3, 3)) * 0.002
    prim = np.array([x[3], [0.5, 0.55, 0.5]])
    return x, v, Cm, F, prim, np.array([666.0, 666.0])


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


def test_constants_of_the_3d_quality_1_config():
    c = PlbConf()
    assert (c.n_grid, c.substeps) == (64, 19) and abs(c.dt - 1e-4) < 1e-18     # mpm_simulator.py:14-32
    c2 = PlbConf(quality=2.0)
    assert (c2.n_grid, c2.substeps) == (128, 39) and abs(c2.dt - 5e-5) < 1e-18


@pytest.mark.parametrize("ys,scaleF", [(1762.2, 1.0), (5.0, 1.05)])
def test_cpp_restatement_matches_numpy_twin(ys, scaleF):
    N = 150
    x, v, Cm, F, prim, soft = _case(N)
    r = PlbTwin(PlbConf(n_particles=N)).step(x, v, Cm, F * scaleF, prim, [0.3, -0.2, 0.1], soft, yield_stress=ys)
    o = PlbOracle(N=N).step(x[None], v[None], Cm[None], (F * scaleF)[None], prim[None], soft[None], [[0.3, -0.2, 0.1]], [5e3], [0.35], [ys])
    assert _rel(o["x"][0], r[0]) < 1e-12 and _rel(o["v"][0], r[1]) < 1e-11
    assert _rel(o["C"][0], r[2]) < 1e-10 and _rel(o["F"][0], r[3]) < 1e-11
    np.testing.assert_allclose(o["prim_pos"][0], r[4], rtol=0, atol=1e-15)


def test_free_fall_known_answer():
    """No primitive contact, F = I, v = 0: after one step every particle moves with v_y = 30 * g_y * dt * substeps
    (gravity is multiplied by 30 in grid_op, :205) -- a uniform field is reproduced exactly by the B-spline transfer."""
    N = 200
    x = torus_particles(1000)[:N]
    z3, z9 = np.zeros((1, N, 3)), np.zeros((1, N, 3, 3))
    F = np.eye(3)[None, None].repeat(N, 1)
    prim = np.array([[[0.1, 0.9, 0.1], [0.9, 0.9, 0.9]]])
    o = PlbOracle(N=N).step(x[None], z3, z9, F, prim, [[666.0, 666.0]], [[0, 0, 0]], [5e3], [0.35], [1762.2])
    np.testing.assert_allclose(o["v"][0, :, 1], 30 * -0.4 * 1e-4 * 19, rtol=1e-7)   # cells with m <= 1e-12 are skipped (:202)
    np.testing.assert_allclose(o["v"][0][:, [0, 2]], 0, atol=1e-9)
    np.testing.assert_allclose(o["F"][0], np.eye(3)[None].repeat(N, 0), atol=1e-9)


def test_sticky_sphere_moves_material_with_it():
    """Sphere.collide (primitives.py:46-53): cells within the soft shell take the sphere's velocity a/substeps/dt."""
    N = 150
    x, v, Cm, F, prim, soft = _case(N)
    a = np.array([0.02, 0.0, 0.01])
    eye = np.eye(3)[None, None].repeat(N, 1)
    o = PlbOracle(N=N).step(x[None], 0 * v[None], 0 * Cm[None], eye, prim[None], soft[None], [a], [5e3], [0.35], [1762.2])
    np.testing.assert_allclose(o["prim_pos"][0, 0], prim[0] + a, atol=1e-12)      # 19 increments of a/19
    near = np.linalg.norm(x - prim[0], axis=1) < 0.01
    assert near.sum() >= 1
    np.testing.assert_allclose(o["v"][0][near], (a / 19 / 1e-4)[None].repeat(near.sum(), 0), rtol=0.05, atol=0.5)
    released = PlbOracle(N=N).step(x[None], 0 * v[None], 0 * Cm[None], eye, prim[None], [[0.0, 666.0]], [a], [5e3], [0.35], [1762.2])
    assert np.abs(released["v"][0][near][:, 0]).max() < 1.0                        # softness 0: the sphere lets go


@pytest.mark.gpu
def test_hip_matches_restatement_full_torus_state():
    import torch
    from unidom_amd.engine.plb_simulator import PlbSimulator
    sim = PlbSimulator(batch_size=3)
    st = sim.reset()
    assert st.x.shape == (3, 1000, 3) and (sim.n_grid, sim.substeps) == (64, 19)
    rng = np.random.default_rng(1)
    x0 = st.x.cpu().numpy()
    v0 = rng.normal(size=x0.shape) * 0.01
    C0 = rng.normal(size=(3, 1000, 3, 3)) * 0.1
    F0 = np.eye(3)[None, None] + rng.normal(size=(3, 1000, 3, 3)) * 0.002
    prim = st.prim_pos.cpu().numpy().copy()
    prim[:, 0] = x0[:, 7]                      # sphere 0 on the rope
    soft = np.array([[666.0, 666.0], [0.0, 666.0], [666.0, 666.0]])
    E, nu, ys = np.array([5e3, 3e3, 5e3]), np.array([0.35, 0.3, 0.35]), np.array([1762.2, 1762.2, 2.0])
    act = np.array([[0.004, 0.003, 0.0], [0.002, -0.001, 0.001], [-0.003, 0.002, 0.002]])
    T = lambda a: torch.tensor(a, dtype=torch.float64, device=sim.device)
    s = st._replace(v=T(v0), C=T(C0), F=T(F0), prim_pos=T(prim), softness=T(soft), E=T(E), nu=T(nu), yield_stress=T(ys))
    orc = PlbOracle(N=1000)
    cur = dict(x=x0, v=v0, C=C0, F=F0, prim_pos=prim)
    for step in range(3):                      # three env steps = 57 substeps
        s = sim.step(s, act)
        cur = orc.step(cur["x"], cur["v"], cur["C"], cur["F"], cur["prim_pos"], soft, act, E, nu, ys, nthreads=3)
    # f64 on both sides; differences = atomics summation order and Jacobi-vs-Jacobi round-off: 1e-9 relative
    for key, t in (("x", s.x), ("v", s.v), ("C", s.C), ("F", s.F)):
        assert _rel(t.cpu().numpy(), cur[key]) < 1e-9, (key, _rel(t.cpu().numpy(), cur[key]))
    np.testing.assert_allclose(s.prim_pos.cpu().numpy(), cur["prim_pos"], rtol=0, atol=1e-14)
    # repeated call on the same handle (grid arena back to all-zero)
    s2 = sim.step(s, act * 0)
    cur2 = orc.step(cur["x"], cur["v"], cur["C"], cur["F"], cur["prim_pos"], soft, act * 0, E, nu, ys, nthreads=3)
    assert _rel(s2.x.cpu().numpy(), cur2["x"]) < 1e-9 and _rel(s2.v.cpu().numpy(), cur2["v"]) < 1e-8


@pytest.mark.gpu
def test_hip_matches_restatement_quality_2():
    """The same path at quality 2 (n_grid 128, dt 0.5e-4 / 1 -> 40 substeps per step, bench.py --workload torus --n-grid 128):
    ~1 particle per cell, the regime where the internal spatial order and the four-lane mapping matter most."""
    import torch
    from unidom_amd.engine.plb_simulator import PlbConf, PlbSimulator
    cfg = PlbConf()
    cfg.quality = 2
    sim = PlbSimulator(cfg, batch_size=2)
    st = sim.reset()
    assert sim.n_grid == 128
    rng = np.random.default_rng(3)
    x0 = st.x.cpu().numpy()
    v0 = rng.normal(size=x0.shape) * 0.01
    C0 = rng.normal(size=(2, 1000, 3, 3)) * 0.1
    F0 = np.eye(3)[None, None] + rng.normal(size=(2, 1000, 3, 3)) * 0.002
    prim = st.prim_pos.cpu().numpy().copy()
    prim[:, 0] = x0[:, 11]
    soft = np.full((2, 2), 666.0)
    E, nu, ys = np.array([5e3, 4e3]), np.array([0.35, 0.3]), np.array([1762.2, 20.0])
    act = np.array([[0.004, 0.003, 0.0], [-0.002, 0.001, 0.002]])
    T = lambda a: torch.tensor(a, dtype=torch.float64, device=sim.device)
    s = st._replace(v=T(v0), C=T(C0), F=T(F0), prim_pos=T(prim), softness=T(soft), E=T(E), nu=T(nu), yield_stress=T(ys))
    orc = PlbOracle(N=1000, n_grid=128, substeps=sim.substeps, dt=sim.dt)
    cur = dict(x=x0, v=v0, C=C0, F=F0, prim_pos=prim)
    for step in range(2):
        s = sim.step(s, act)
        cur = orc.step(cur["x"], cur["v"], cur["C"], cur["F"], cur["prim_pos"], soft, act, E, nu, ys, nthreads=2)
    for key, t in (("x", s.x), ("v", s.v), ("C", s.C), ("F", s.F)):
        assert _rel(t.cpu().numpy(), cur[key]) < 1e-9, (key, _rel(t.cpu().numpy(), cur[key]))


@pytest.mark.gpu
def test_hip_one_lane_kernels_match_restatement(monkeypatch):
    """plb_p2g / plb_g2p come in two lane mappings (4 lanes per particle below 100 k particles per launch, 1 beyond); UD_PLB_LANES,
    read at every step call, forces one lane per particle so that mapping meets the restatement too."""
    monkeypatch.setenv("UD_PLB_LANES", "1")
    test_hip_matches_restatement_full_torus_state()
