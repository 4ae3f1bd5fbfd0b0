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
@pytest.mark.parametrize("path", [2, 1])
def test_hip_matches_restatement_full_torus_state(path, lanes=0):
    """path 2: ONE persistent launch per step call (csrc/plb_cluster.hip; what ud_plb_create picks for the Torus sizes); path 1: the
    multi-kernel path (2 launches per substep)."""
    import torch
    from unidom_amd.engine.plb_simulator import PlbConf, PlbSimulator
    cfg = PlbConf()
    cfg.path, cfg.lanes = path, lanes
    sim = PlbSimulator(cfg, batch_size=3)
    assert sim.launch_plan() == path
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
    sim.check_status()


@pytest.mark.gpu
@pytest.mark.parametrize("path", [2, 1])
def test_hip_matches_restatement_quality_2(path):
    """The same paths at quality 2 (n_grid 128, dt 0.5e-4 / 1 -> 39 substeps per step (int(2e-3 // 5e-5), float floor division), bench.py --workload torus --n-grid 128):
    ~1 particle per cell, the regime where the internal spatial order and the four-lane mapping matter most."""
    import torch
    from unidom_amd.engine.plb_simulator import PlbConf, PlbSimulator
    cfg = PlbConf()
    cfg.quality = 2
    cfg.path = path
    sim = PlbSimulator(cfg, batch_size=2)
    st = sim.reset()
    assert sim.n_grid == 128 and sim.launch_plan() == path
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
@pytest.mark.parametrize("lanes", [1, 4, 8])
def test_hip_other_lane_mappings_match_restatement(lanes):
    """The multi-kernel path's particle kernels come in three lane mappings: 8 lanes per particle up to 16 k particles per launch, 4 below
    100 k, 1 beyond; ud_plb_conf.lanes (fixed at create) puts each of them in front of the restatement at the tests' sizes."""
    test_hip_matches_restatement_full_torus_state(1, lanes)


@pytest.mark.gpu
def test_hip_persistent_path_several_launches_per_call_and_odd_part_sizes():
    """The persistent path cuts a call into launches whose workgroups are all resident at once (envs per launch = CUs x occupancy / parts per
    env: 8 envs of 32 parts on 256 CUs): 11 envs = a launch of 8 and one of 3 on the same exchange arena; N = 1000 leaves a last part of 8
    particles, N = 37 a second part of 5, N = 5 a single part.  Forward with checkpoint vs the restatement, twice on the same handle."""
    import torch
    from unidom_amd.engine.plb_simulator import PlbConf, PlbSimulator
    for N, B in ((1000, 11), (37, 3), (5, 2)):
        cfg = PlbConf()
        cfg.n_particles, cfg.path = N, 2
        sim = PlbSimulator(cfg, batch_size=B)
        rng = np.random.default_rng(N)
        x0 = sim.reset().x.cpu().numpy() + rng.normal(size=(B, N, 3)) * 1e-3
        v0 = rng.normal(size=x0.shape) * 0.01
        C0 = rng.normal(size=(B, N, 3, 3)) * 0.1
        F0 = np.eye(3)[None, None] + rng.normal(size=(B, N, 3, 3)) * 0.002
        prim = np.stack([x0[:, 3], np.repeat(np.array([[0.5, 0.75, 0.5]]), B, 0)], 1)
        soft = np.full((B, 2), 666.0)
        E, nu, ys = rng.uniform(3e3, 5e3, size=B), rng.uniform(0.3, 0.35, size=B), np.where(np.arange(B) % 2 == 0, 1762.2, 5.0)
        act = rng.uniform(-0.004, 0.004, size=(B, 3))
        T = lambda a, r=False: torch.tensor(a, dtype=torch.float64, device=sim.device, requires_grad=r)
        s = sim.reset()._replace(x=T(x0, True), v=T(v0), C=T(C0), F=T(F0), prim_pos=T(prim), softness=T(soft), E=T(E), nu=T(nu), yield_stress=T(ys))
        cur = dict(x=x0, v=v0, C=C0, F=F0, prim_pos=prim)
        orc = PlbOracle(N=N)
        for step in range(2):
            s = sim.step(s, act)
            cur = orc.step(cur["x"], cur["v"], cur["C"], cur["F"], cur["prim_pos"], soft, act, E, nu, ys, nthreads=4)
        sim.check_status()
        for key, t in (("x", s.x), ("v", s.v), ("C", s.C), ("F", s.F)):
            assert _rel(t.detach().cpu().numpy(), cur[key]) < 1e-9, (N, key, _rel(t.detach().cpu().numpy(), cur[key]))
        np.testing.assert_allclose(s.prim_pos.detach().cpu().numpy(), cur["prim_pos"], rtol=0, atol=1e-14)


# ---------------------------------------------------------------------------------------------------------------------
# adjoint, losses, parameter gradients.  The checker is torch.autograd through the torch twin, whose
# forward is held against the literal NumPy twin and whose gradient is held against central differences here.
# ---------------------------------------------------------------------------------------------------------------------
def _small_case(B, N, seed, n_grid_quality=0.5, low=True):
    """A body resting on / falling onto the floor (friction branch), sphere 0 inside it (sticky contact), sphere 1 away."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(B, N, 3)) * np.array([0.25, 0.12 if low else 0.2, 0.25]) + np.array([0.38, 0.035 if low else 0.2, 0.38])
    v = rng.normal(size=(B, N, 3)) * 0.3 + np.array([0.0, -0.8, 0.0])
    Cm = rng.normal(size=(B, N, 3, 3)) * 2.0
    F = np.eye(3)[None, None] + rng.normal(size=(B, N, 3, 3)) * 0.12
    prim = np.array([[0.5, 0.1 if low else 0.3, 0.5], [0.2, 0.7, 0.2]])[None].repeat(B, 0) + rng.normal(size=(B, 2, 3)) * 0.01
    soft = np.array([[666.0, 666.0]]).repeat(B, 0)
    act = rng.uniform(-0.9, 0.9, size=(B, 3)) * np.array([1.0, 0.3, 1.0])
    E = np.array([5e3, 3e3, 4e3])[:B]
    nu = np.array([0.35, 0.3, 0.25])[:B]
    ys = np.array([1762.2, 30.0, 200.0])[:B]          # env 1 yields almost everywhere, env 0 hardly
    return x, v, Cm, F, prim, soft, act, E, nu, ys


def _twin_step(conf, case, fric, w):
    import torch
    from oracle.twin.plb_twin_torch import PlbTorchTwin
    x, v, Cm, F, prim, soft, act, E, nu, ys = case
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), requires_grad=r)
    leaves = dict(x=T(x), v=T(v), C=T(Cm), F=T(F), prim=T(prim), act=T(act), E=T(E), nu=T(nu), ys=T(ys), fric=T(np.full(len(E), fric)))
    tw = PlbTorchTwin(conf)
    out = tw.step(leaves["x"], leaves["v"], leaves["C"], leaves["F"], leaves["prim"], leaves["act"], T(soft, False), leaves["E"],
                  leaves["nu"], leaves["ys"], leaves["fric"])
    loss = sum((o * torch.tensor(wi)).sum() for o, wi in zip(out, w))
    return leaves, out, loss


def test_torch_twin_is_the_numpy_twin_and_its_autograd_is_the_derivative():
    import torch
    torch.set_num_threads(4)
    conf = PlbConf(quality=0.25, n_particles=60)            # n_grid 16, 4 substeps
    case = _small_case(1, 60, 0)
    x, v, Cm, F, prim, soft, act, E, nu, ys = case
    r = PlbTwin(conf).step(x[0], v[0], Cm[0], F[0], prim[0], act[0], soft[0], E=E[0], nu=nu[0], yield_stress=ys[0])
    rng = np.random.default_rng(5)
    w = [rng.normal(size=s) for s in ((1, 60, 3), (1, 60, 3), (1, 60, 3, 3), (1, 60, 3, 3), (1, 2, 3))]
    leaves, out, loss = _twin_step(conf, case, conf.ground_friction, w)
    for o, ref in zip(out, r):
        assert _rel(o.detach().numpy()[0], ref) < 1e-11
    loss.backward()

    def f(**kw):
        c2 = list(case)
        names = ["x", "v", "C", "F", "prim", "soft", "act", "E", "nu", "ys"]
        for k, val in kw.items():
            c2[names.index(k)] = val
        return float(_twin_step(conf, tuple(c2), kw.get("fric", conf.ground_friction), w)[2].detach())

    h = 1e-6
    for name, idx in (("x", (0, 7, 1)), ("v", (0, 3, 0)), ("C", (0, 11, 2, 1)), ("F", (0, 20, 0, 1)), ("act", (0, 0)), ("prim", (0, 0, 2))):
        names = ["x", "v", "C", "F", "prim", "soft", "act", "E", "nu", "ys"]
        base = case[names.index(name)]
        up, dn = base.copy(), base.copy()
        up[idx] += h; dn[idx] -= h
        fd = (f(**{name: up}) - f(**{name: dn})) / (2 * h)
        an = float(leaves[name].grad[idx])
        assert abs(fd - an) < 1e-5 * max(1.0, abs(an)), (name, fd, an)
    for name, val in (("E", E), ("nu", nu), ("ys", ys)):
        hh = 1e-6 * val[0]
        fd = (f(**{name: val + hh}) - f(**{name: val - hh})) / (2 * hh)
        an = float(leaves[name].grad[0])
        assert abs(fd - an) < 1e-5 * max(1e-3, abs(an)), (name, fd, an)


def test_loss_twin_known_answers():
    """density: a grid mass equal to the target costs 0; sdf: linear in the mass; contact (soft): one particle at distance d from
    a sphere gives min_dist = d; hard: the nearest particle's distance."""
    import torch
    from oracle.twin.plb_twin_torch import PlbTorchTwin
    conf = PlbConf(quality=0.25, n_particles=5)
    tw = PlbTorchTwin(conf)
    x = torch.tensor([[[0.5, 0.5, 0.5], [0.52, 0.5, 0.5], [0.6, 0.55, 0.5], [0.3, 0.3, 0.3], [0.7, 0.3, 0.4]]], dtype=torch.float64)
    gm = tw.grid_mass(x)
    assert abs(float(gm.sum()) - 5 * conf.p_mass) < 1e-15                      # the B-spline weights sum to 1
    prim = torch.tensor([[[0.5, 0.5, 0.56], [0.9, 0.9, 0.9]]], dtype=torch.float64)
    zero = torch.zeros(conf.n_grid ** 3, dtype=torch.float64)
    total, parts = tw.loss(x, prim, gm[0].detach(), zero, (1.0, 1.0, 1.0), soft_contact=False)
    d0 = np.sqrt(0.06 ** 2 + 1e-14) - 0.025
    d1 = np.sqrt(0.3 ** 2 + 0.35 ** 2 + 0.4 ** 2 + 1e-14) - 0.025           # particle 2 is the nearest to sphere 1
    assert abs(float(parts[0, 1])) < 1e-18 and abs(float(parts[0, 0]) - (d0 ** 2 + d1 ** 2)) < 1e-12
    sdf = torch.full((conf.n_grid ** 3,), 2.0, dtype=torch.float64)
    total, parts = tw.loss(x, prim, zero, sdf, (0.0, 0.0, 1.0))
    assert abs(float(total) - 2.0 * 5 * conf.p_mass) < 1e-15
    one = x[:, :1]
    total, parts = tw.loss(one, prim[:, :1], zero, zero, (1.0, 0.0, 0.0), soft_contact=True)
    assert abs(float(total) - d0 ** 2) < 1e-15


def _hip_sim(N, B, quality=0.5, grid_ckpt_cells=None, path=0, lanes=0, sort_every=0, substeps=0):
    from unidom_amd.engine.plb_simulator import PlbConf as HipConf, PlbSimulator
    cfg = HipConf()
    cfg.quality = quality
    cfg.substeps = substeps
    cfg.n_particles = N
    cfg.path, cfg.lanes, cfg.sort_every = path, lanes, sort_every
    if grid_ckpt_cells is not None:
        cfg.grid_ckpt_cells = grid_ckpt_cells
    return PlbSimulator(cfg, batch_size=B)


@pytest.mark.gpu
@pytest.mark.parametrize("path", [2, 1])
@pytest.mark.parametrize("low", [True, False])
def test_hip_step_adjoint_matches_torch_twin(low, path, lanes=0):
    """ud_plb_step_fwd (with checkpoint) + ud_plb_step_bwd (path 2: one persistent launch each; path 1: the multi-kernel path) against torch.autograd through the twin: three envs with different
    E / nu / yield stress (one yields almost everywhere: the return-mapping adjoint; one hardly: the SVD adjoint alone), a body
    on the floor (ground-friction branch and the boundary zeroing) or in the air, sphere 0 inside it (sticky contact -> action
    and primitive-position cotangents).  n_grid 32, 9 substeps.  f64 on both sides; what differs is summation order (atomics)
    and the SVD (Jacobi vs LAPACK), amplified by the clamped 1 / (s_j^2 - s_i^2) of backward_svd: 1e-6 relative."""
    import torch
    torch.set_num_threads(8)
    B, N = 3, 300
    sim = _hip_sim(N, B, path=path, lanes=lanes)
    assert (sim.n_grid, sim.substeps) == (32, 9) and sim.launch_plan() == path
    conf = PlbConf(quality=0.5, n_particles=N)
    case = _small_case(B, N, 1, low=low)
    x, v, Cm, F, prim, soft, act, E, nu, ys = case
    rng = np.random.default_rng(9)
    w = [rng.normal(size=s) for s in ((B, N, 3), (B, N, 3), (B, N, 3, 3), (B, N, 3, 3), (B, 2, 3))]
    leaves, out, loss = _twin_step(conf, case, conf.ground_friction, w)
    loss.backward()
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
    st = sim.reset()
    hl = dict(x=T(x), v=T(v), C=T(Cm), F=T(F), prim=T(prim), act=T(act), E=T(E), nu=T(nu), ys=T(ys))
    s = st._replace(x=hl["x"], v=hl["v"], C=hl["C"], F=hl["F"], prim_pos=hl["prim"], softness=T(soft, False), E=hl["E"], nu=hl["nu"],
                    yield_stress=hl["ys"])
    s1 = sim.step(s, hl["act"])
    for o, t, name in zip(out, (s1.x, s1.v, s1.C, s1.F, s1.prim_pos), "xvCFp"):
        assert _rel(t.detach().cpu().numpy(), o.detach().numpy()) < 1e-9, name
    hloss = sum((t * T(wi, False)).sum() for t, wi in zip((s1.x, s1.v, s1.C, s1.F, s1.prim_pos), w))
    sim.ground_friction_grad = None
    hloss.backward()
    sim.check_status()
    for name in ("x", "v", "C", "F", "prim", "act", "E", "nu", "ys"):
        got, ref = hl[name].grad.cpu().numpy(), leaves[name].grad.numpy()
        assert np.isfinite(got).all(), name
        assert _rel(got, ref) < 1e-6, (name, _rel(got, ref))
    gfr = sim.ground_friction_grad.cpu().numpy()
    ref = leaves["fric"].grad.numpy()
    assert _rel(gfr, ref) < 1e-6 or np.abs(ref).max() < 1e-12, (gfr, ref)
    if low:
        assert np.abs(ref).max() > 0          # the friction branch really ran
    assert np.abs(leaves["act"].grad.numpy()).max() > 0 and np.abs(leaves["ys"].grad.numpy()[1]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("path", [2, 1])
def test_hip_negative_yield_stress_follows_the_return_mapping(path):
    """yield_stress <= 0 (g_ys is an exposed gradient: an optimiser can push it there): dg = |dev log-strain| - ys / 2 mu is then positive
    for every particle and the reference's return mapping (mpm_simulator.py:133-150) yields everywhere.  The kernels decide "cannot yield"
    from a squared bound without taking logarithms; that shortcut must not fire for a negative limit.  Forward 1e-9 / every leaf 1e-6
    against the twin, which has no such shortcut."""
    import torch
    torch.set_num_threads(8)
    B, N = 3, 300
    sim = _hip_sim(N, B, path=path)
    conf = PlbConf(quality=0.5, n_particles=N)
    x, v, Cm, F, prim, soft, act, E, nu, ys = _small_case(B, N, 3, low=True)
    ys = np.array([-20.0, 1762.2, -500.0])
    F = np.eye(3)[None, None] + (F - np.eye(3)) * 0.05          # nearly undeformed: the bound of the shortcut is tiny, only the sign of ys makes these yield
    case = (x, v, Cm, F, prim, soft, act, E, nu, ys)
    rng = np.random.default_rng(10)
    w = [rng.normal(size=s) for s in ((B, N, 3), (B, N, 3), (B, N, 3, 3), (B, N, 3, 3), (B, 2, 3))]
    leaves, out, loss = _twin_step(conf, case, conf.ground_friction, w)
    loss.backward()
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
    hl = dict(x=T(x), v=T(v), C=T(Cm), F=T(F), prim=T(prim), act=T(act), E=T(E), nu=T(nu), ys=T(ys))
    s = sim.reset()._replace(x=hl["x"], v=hl["v"], C=hl["C"], F=hl["F"], prim_pos=hl["prim"], softness=T(soft, False), E=hl["E"], nu=hl["nu"],
                             yield_stress=hl["ys"])
    s1 = sim.step(s, hl["act"])
    for o, t, name in zip(out, (s1.x, s1.v, s1.C, s1.F, s1.prim_pos), "xvCFp"):
        assert _rel(t.detach().cpu().numpy(), o.detach().numpy()) < 1e-9, name
    sum((t * T(wi, False)).sum() for t, wi in zip((s1.x, s1.v, s1.C, s1.F, s1.prim_pos), w)).backward()
    sim.check_status()
    for name in ("x", "v", "C", "F", "prim", "act", "E", "nu", "ys"):
        got, ref = hl[name].grad.cpu().numpy(), leaves[name].grad.numpy()
        assert np.isfinite(got).all(), name
        assert _rel(got, ref) < 1e-6, (name, _rel(got, ref))
    assert (np.abs(leaves["ys"].grad.numpy()[[0, 2]]) > 0).all()      # the negative-limit envs really went through the mapping


@pytest.mark.gpu
def test_hip_step_adjoint_at_the_benched_configuration():
    """bench.py --workload torus --plb-grad as it is measured: N = 1000, n_grid 64 (quality 1, 19 substeps per env.step), 8 envs in
    one launch, the kernels ud_plb_create picks by itself (the persistent path: 8 x 32 workgroups that must all be resident and meet
    at an inter-workgroup barrier every substep) -- the body of the
    Torus task (shape_maker.py:49-58) with sphere 0 on it (sticky contact: action and primitive-position cotangents), every env with
    its own v / C / F, action, E, nu and yield stress.  Two of the eight envs are followed by torch.autograd through the twin over
    the whole env.step: forward 1e-9, every leaf 1e-6 relative (the tolerances of the n_grid-32 test above)."""
    import torch
    torch.set_num_threads(8)
    B, N, pick = 8, 1000, [2, 7]
    sim = _hip_sim(N, B, quality=1.0)
    assert (sim.n_grid, sim.substeps) == (64, 19) and sim.launch_plan() == 2     # the library's choice at this shape: the persistent path
    conf = PlbConf(quality=1.0, n_particles=N)
    rng = np.random.default_rng(21)
    x = torus_particles(1000)[None].repeat(B, 0) + rng.normal(size=(B, N, 3)) * 1e-4
    v = rng.normal(size=(B, N, 3)) * 0.05
    Cm = rng.normal(size=(B, N, 3, 3)) * 0.5
    F = np.eye(3)[None, None] + rng.normal(size=(B, N, 3, 3)) * 0.02
    prim = np.stack([x[:, 7], np.repeat(np.array([[0.5, 0.55, 0.5]]), B, 0)], 1)
    soft = np.full((B, 2), 666.0)
    # the sphere moves action / substeps / dt per substep: 0.01 per env.step is 5 m/s, the scripted rollout's order of magnitude
    # (solver.py:299-302: 0.0015 per step); 0.5 would be 1.6 cells per substep and the forward itself blows up
    act = rng.uniform(-0.01, 0.01, size=(B, 3)) * np.array([1.0, 0.3, 1.0])
    E = rng.uniform(3e3, 6e3, size=B)
    nu = rng.uniform(0.25, 0.4, size=B)
    ys = np.array([1762.2, 30.0, 1762.2, 200.0, 1762.2, 50.0, 1762.2, 30.0])      # env 7 yields almost everywhere, env 2 hardly
    w = [rng.normal(size=s) for s in ((B, N, 3), (B, N, 3), (B, N, 3, 3), (B, N, 3, 3), (B, 2, 3))]
    sub = lambda a: np.ascontiguousarray(np.asarray(a)[pick])
    case = tuple(sub(a) for a in (x, v, Cm, F, prim, soft, act, E, nu, ys))
    leaves, out, loss = _twin_step(conf, case, conf.ground_friction, [sub(wi) for wi in w])
    loss.backward()
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
    hl = dict(x=T(x), v=T(v), C=T(Cm), F=T(F), prim=T(prim), act=T(act), E=T(E), nu=T(nu), ys=T(ys))
    s = sim.reset()._replace(x=hl["x"], v=hl["v"], C=hl["C"], F=hl["F"], prim_pos=hl["prim"], softness=T(soft, False), E=hl["E"],
                             nu=hl["nu"], yield_stress=hl["ys"])
    s1 = sim.step(s, hl["act"])
    for o, t, name in zip(out, (s1.x, s1.v, s1.C, s1.F, s1.prim_pos), "xvCFp"):
        assert torch.isfinite(t).all(), name
        assert _rel(t.detach().cpu().numpy()[pick], o.detach().numpy()) < 1e-9, name
    sim.ground_friction_grad = None
    sum((t * T(wi, False)).sum() for t, wi in zip((s1.x, s1.v, s1.C, s1.F, s1.prim_pos), w)).backward()
    sim.check_status()
    for name in ("x", "v", "C", "F", "prim", "act", "E", "nu", "ys"):
        got, ref = hl[name].grad.cpu().numpy(), leaves[name].grad.numpy()
        assert np.isfinite(got).all(), name
        assert _rel(got[pick], ref) < 1e-6, (name, _rel(got[pick], ref))
    assert np.abs(leaves["act"].grad.numpy()).max() > 0 and np.abs(leaves["ys"].grad.numpy()[1]) > 0


@pytest.mark.gpu
def test_hip_step_adjoint_at_n_grid_128_the_other_benched_configuration():
    """bench.py --workload torus --plb-grad --n-grid 128 (BASELINE config 5's "128^3 grid": quality 2, dt 5e-5): N = 1000, 8 envs in one
    launch, the kernels ud_plb_create picks by itself (persistent path: 8 x 32 parts, 1024-slot cell tables at about one particle per
    cell, per-part cell records, the three-buffer ring of exchange grids).  The twin's tape of 39 dense 128^3 substeps does not fit in
    memory, and the launch shape -- not the horizon -- is what this grid adds, so ud_plb_conf.substeps is cut to 4 on both sides
    (set_action divides the action by the same count: G/engine/primitive/primive_base.py:185-192).  Two of the eight envs are followed
    by torch.autograd through the twin: forward 1e-9, every leaf 1e-6 relative -- the tolerances of the n_grid-32 / 64 tests."""
    import torch
    torch.set_num_threads(8)
    B, N, S, pick = 8, 1000, 4, [1, 6]
    sim = _hip_sim(N, B, quality=2.0, substeps=S)
    assert (sim.n_grid, sim.substeps) == (128, S) and abs(sim.dt - 5e-5) < 1e-18 and sim.launch_plan() == 2

    class Short(PlbConf):
        substeps = S
    conf = Short(quality=2.0, n_particles=N)
    assert (conf.n_grid, conf.substeps) == (128, S)
    rng = np.random.default_rng(22)
    x = torus_particles(1000)[None].repeat(B, 0) + rng.normal(size=(B, N, 3)) * 1e-4
    x[:, :, 1] -= x[:, :, 1].min(1, keepdims=True) - 1.2 / 128      # foot of the column on the floor: ground friction + boundary zeroing
    v = rng.normal(size=(B, N, 3)) * 0.05
    v[:, :, 1] -= 0.2
    Cm = rng.normal(size=(B, N, 3, 3)) * 0.5
    F = np.eye(3)[None, None] + rng.normal(size=(B, N, 3, 3)) * 0.02
    prim = np.stack([x[:, 7], np.repeat(np.array([[0.5, 0.55, 0.5]]), B, 0)], 1)
    soft = np.full((B, 2), 666.0)
    act = rng.uniform(-0.002, 0.002, size=(B, 3)) * np.array([1.0, 0.3, 1.0])   # over 4 substeps instead of 39: same per-substep travel as 0.01-0.02 per env.step
    E = rng.uniform(3e3, 6e3, size=B)
    nu = rng.uniform(0.25, 0.4, size=B)
    ys = np.array([1762.2, 30.0, 1762.2, 200.0, 1762.2, 50.0, 30.0, 1762.2])      # env 1 and 6 yield almost everywhere / in part
    w = [rng.normal(size=s) for s in ((B, N, 3), (B, N, 3), (B, N, 3, 3), (B, N, 3, 3), (B, 2, 3))]
    sub = lambda a: np.ascontiguousarray(np.asarray(a)[pick])
    case = tuple(sub(a) for a in (x, v, Cm, F, prim, soft, act, E, nu, ys))
    leaves, out, loss = _twin_step(conf, case, conf.ground_friction, [sub(wi) for wi in w])
    loss.backward()
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
    hl = dict(x=T(x), v=T(v), C=T(Cm), F=T(F), prim=T(prim), act=T(act), E=T(E), nu=T(nu), ys=T(ys))
    s = sim.reset()._replace(x=hl["x"], v=hl["v"], C=hl["C"], F=hl["F"], prim_pos=hl["prim"], softness=T(soft, False), E=hl["E"],
                             nu=hl["nu"], yield_stress=hl["ys"])
    s1 = sim.step(s, hl["act"])
    for o, t, name in zip(out, (s1.x, s1.v, s1.C, s1.F, s1.prim_pos), "xvCFp"):
        assert torch.isfinite(t).all(), name
        assert _rel(t.detach().cpu().numpy()[pick], o.detach().numpy()) < 1e-9, name
    sim.ground_friction_grad = None
    sum((t * T(wi, False)).sum() for t, wi in zip((s1.x, s1.v, s1.C, s1.F, s1.prim_pos), w)).backward()
    sim.check_status()
    for name in ("x", "v", "C", "F", "prim", "act", "E", "nu", "ys"):
        got, ref = hl[name].grad.cpu().numpy(), leaves[name].grad.numpy()
        assert np.isfinite(got).all(), name
        assert _rel(got[pick], ref) < 1e-6, (name, _rel(got[pick], ref))
    gfr, rfr = sim.ground_friction_grad.cpu().numpy()[pick], leaves["fric"].grad.numpy()
    assert np.abs(rfr).max() > 0 and _rel(gfr, rfr) < 1e-6, (gfr, rfr)          # the friction branch really ran
    assert np.abs(leaves["act"].grad.numpy()).max() > 0 and np.abs(leaves["ys"].grad.numpy()).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("K", [0, 1, 2])
def test_hip_adjoint_with_and_without_the_grid_checkpoint(K):
    """ud_plb_conf.grid_ckpt_cells: by default (27 cells per particle: every cell a substep can touch) the adjoint restores each
    substep's (m, mv) cells from the checkpoint and never launches p2g again -- that is the configuration the twin comparison above
    ran.  Here the other modes against it: K = 0 (no grid checkpoint: p2g recomputed every substep), K = 1 and 2 (a pool that the
    substeps of this body overflow, some or all: those fall back to recomputing on the device, env by env).  The restored cells are
    the forward's own sums, the recomputed ones a second run of the same atomics: 1e-8 relative."""
    import torch
    B, N = 3, 300
    case = _small_case(B, N, 1, low=True)
    x, v, Cm, F, prim, soft, act, E, nu, ys = case
    rng = np.random.default_rng(9)
    w = [rng.normal(size=s) for s in ((B, N, 3), (B, N, 3), (B, N, 3, 3), (B, N, 3, 3), (B, 2, 3))]

    def run(k):
        sim = _hip_sim(N, B, grid_ckpt_cells=k, path=1)     # the multi-kernel path's own machinery
        assert sim.grid_ckpt_cells == k
        T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
        hl = dict(x=T(x), v=T(v), C=T(Cm), F=T(F), prim=T(prim), act=T(act), E=T(E), nu=T(nu), ys=T(ys))
        s = sim.reset()._replace(x=hl["x"], v=hl["v"], C=hl["C"], F=hl["F"], prim_pos=hl["prim"], softness=T(soft, False), E=hl["E"],
                                 nu=hl["nu"], yield_stress=hl["ys"])
        s1 = sim.step(sim.step(s, hl["act"]), hl["act"])                 # two env.steps chained
        sum((t * T(wi, False)).sum() for t, wi in zip((s1.x, s1.v, s1.C, s1.F, s1.prim_pos), w)).backward()
        return {k2: t.grad.cpu().numpy() for k2, t in hl.items()}, sim.ground_friction_grad.cpu().numpy()

    ref, ref_f = run(27)
    got, got_f = run(K)
    for name in ref:
        assert np.isfinite(got[name]).all() and _rel(got[name], ref[name]) < 1e-8, (name, _rel(got[name], ref[name]))
    assert _rel(got_f, ref_f) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("path", [2, 1])
def test_hip_two_steps_chain_and_forward_without_checkpoint(path):
    """Two env.steps chained through autograd (the tape of solver.py:41-54) equal the twin's; a forward under no_grad (no
    checkpoint) gives the same state as the checkpointing forward."""
    import torch
    torch.set_num_threads(8)
    B, N = 2, 200
    sim = _hip_sim(N, B, path=path)
    conf = PlbConf(quality=0.5, n_particles=N)
    case = _small_case(B, N, 4)
    x, v, Cm, F, prim, soft, act, E, nu, ys = case
    from oracle.twin.plb_twin_torch import PlbTorchTwin
    Tc = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), requires_grad=r)
    tl = dict(x=Tc(x), act=Tc(act), E=Tc(E))
    tw = PlbTorchTwin(conf)
    fr = Tc(np.full(B, conf.ground_friction), False)
    o1 = tw.step(tl["x"], Tc(v, False), Tc(Cm, False), Tc(F, False), Tc(prim, False), tl["act"], Tc(soft, False), tl["E"], Tc(nu, False), Tc(ys, False), fr)
    o2 = tw.step(*o1[:4], o1[4], tl["act"] * 0.5, Tc(soft, False), tl["E"], Tc(nu, False), Tc(ys, False), fr)
    (o2[0].sum() + o2[1][..., 1].sum()).backward()
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
    hl = dict(x=T(x), act=T(act), E=T(E))
    st = sim.reset()._replace(x=hl["x"], v=T(v, False), C=T(Cm, False), F=T(F, False), prim_pos=T(prim, False), softness=T(soft, False),
                              E=hl["E"], nu=T(nu, False), yield_stress=T(ys, False))
    s1 = sim.step(st, hl["act"])
    s2 = sim.step(s1, hl["act"] * 0.5)
    assert _rel(s2.x.detach().cpu().numpy(), o2[0].detach().numpy()) < 1e-9
    (s2.x.sum() + s2.v[..., 1].sum()).backward()
    for name in ("x", "act", "E"):
        assert _rel(hl[name].grad.cpu().numpy(), tl[name].grad.numpy()) < 1e-6, name
    with torch.no_grad():
        n1 = sim.step(st, hl["act"])
    assert _rel(n1.x.cpu().numpy(), s1.x.detach().cpu().numpy()) < 1e-12 and _rel(n1.F.cpu().numpy(), s1.F.detach().cpu().numpy()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("soft_contact", [True, False])
def test_hip_losses_match_torch_twin(soft_contact):
    """ud_plb_loss_fwd / bwd (density, SDF, contact; engine/losses/loss.py) against the twin and its autograd, at the Torus
    task's own size (N = 1000, n_grid 64), with a target density that is another body's grid mass and a random SDF field."""
    import torch
    from oracle.twin.plb_twin_torch import PlbTorchTwin
    B, N = 2, 1000
    sim = _hip_sim(N, B, quality=1.0)
    conf = PlbConf(n_particles=N)
    rng = np.random.default_rng(2)
    x = np.stack([torus_particles(1000), torus_particles(1000) + rng.normal(size=(1000, 3)) * 0.003])
    prim = np.array([[[0.47, 0.28, 0.5], [0.55, 0.62, 0.5]], [[0.5, 0.35, 0.52], [0.5, 0.1, 0.5]]])
    tw = PlbTorchTwin(conf)
    other = torch.tensor((torus_particles(1000) + np.array([0.004, -0.01, 0.0]))[None])
    td = tw.grid_mass(other)[0]
    ts = torch.tensor(rng.normal(size=conf.n_grid ** 3))
    wts = (3.0, 0.7, 1.3)
    tx, tp = torch.tensor(x, requires_grad=True), torch.tensor(prim, requires_grad=True)
    total, parts = tw.loss(tx, tp, td, ts, wts, soft_contact=soft_contact)
    gl = torch.tensor([1.0, -2.5], dtype=torch.float64)
    (total * gl).sum().backward()
    T = lambda a, r=True: torch.tensor(np.asarray(a, np.float64), device=sim.device, requires_grad=r)
    hx, hp = T(x), T(prim)
    st = sim.reset()._replace(x=hx, prim_pos=hp)
    hloss, hparts = sim.compute_loss(st, td.numpy(), ts.numpy(), wts, soft_contact)
    assert _rel(hloss.detach().cpu().numpy(), total.detach().numpy()) < 1e-11
    assert _rel(hparts.cpu().numpy(), parts.detach().numpy()) < 1e-11
    (hloss * T(gl.numpy(), False)).sum().backward()
    assert _rel(hx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-9 and _rel(hp.grad.cpu().numpy(), tp.grad.numpy()) < 1e-9
    assert np.abs(tp.grad.numpy()).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [1, 4])
def test_hip_step_adjoint_other_lane_mappings(lanes):
    """The multi-kernel adjoint kernels have the forward's lane mappings (8 / 4 / 1 lanes per particle by launch size); ud_plb_conf.lanes
    puts the instantiations the tests' sizes do not get by default in front of the twin too."""
    test_hip_step_adjoint_matches_torch_twin(True, 1, lanes)


@pytest.mark.gpu
@pytest.mark.parametrize("path", [2, 1])
def test_hip_spatial_order_is_reused_and_still_valid_after_the_state_changes(path):
    """The spatial order is computed on the first call and every eighth after it (ud_plb_conf.sort_every); in between a call runs on
    the previous order -- any permutation is valid, only its locality ages.  A handle that sorted for one state and is then given a
    completely different one (the body mirrored and shuffled) still matches a handle that sorts at every call."""
    from unidom_amd.engine.plb_simulator import PlbConf, PlbSimulator
    import torch
    cfg = PlbConf()
    cfg.path = path
    sim = PlbSimulator(cfg, batch_size=2)
    cfg1 = PlbConf()
    cfg1.path, cfg1.sort_every = path, 1
    sim1 = PlbSimulator(cfg1, batch_size=2)
    st = sim.reset()
    act = torch.tensor([[0.3, -0.2, 0.1]] * 2, dtype=torch.float64, device=sim.device)
    s1 = sim.step(st, act)                                  # sorts for this state
    rng = np.random.default_rng(5)
    perm = torch.tensor(rng.permutation(sim.n_particles), device=sim.device)
    x2 = st.x[:, perm].clone()
    x2[..., 0] = 1.0 - x2[..., 0]                           # elsewhere in the grid, in another particle order
    st2 = st._replace(x=x2)
    got = sim.step(st2, act)                                # reuses the order computed for `st`
    ref = sim1.step(st2, act)                               # sorts for st2
    for a, b, name in ((got.x, ref.x, "x"), (got.v, ref.v, "v"), (got.C, ref.C, "C"), (got.F, ref.F, "F")):
        assert _rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-9, name
    assert torch.isfinite(s1.x).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", [2, 1])
def test_hip_smaller_batch_after_larger_stays_inside_the_callers_arrays(path):
    """One handle, a call with B = 8 and then one with B = 3 (ud_plb_step_fwd / _bwd take B per call; the handle's arenas are sized
    for max_envs = 8 at create): every [B] / [B, ...] array of the second call sits between canaries, and its results equal those of a
    fresh handle that has only ever seen B = 3.  (The per-env guards of plb_prologue / plb_adj_epilogue once used the arena's
    B: envs 3..7 of the epilogue's 64-thread block then wrote g_action, g_E, ... past the caller's arrays.)"""
    import ctypes as C
    import torch
    from unidom_amd import _lib
    N = 200
    L = _lib.lib()
    dev = torch.device("cuda")
    CANARY = -7.25e11

    def run(sim, B, seed):
        x, v, Cm, F, prim, soft, act, E, nu, ys = _small_case(3, N, seed)
        rep = lambda a: np.concatenate([a] * 3, 0)[:B]               # B envs out of the 3-env case, cyclically
        ins = [rep(a) for a in (x, v, Cm, F, prim, soft, act, E, nu, ys)]
        PAD = 4096
        bufs = []

        def guarded(shape, fill=None):
            n = int(np.prod(shape))
            buf = torch.full((n + 2 * PAD,), CANARY, dtype=torch.float64, device=dev)
            t = buf[PAD:PAD + n].view(*shape)
            if fill is not None:
                t.copy_(torch.tensor(np.asarray(fill, np.float64), device=dev))
            bufs.append((buf, n))
            return t

        gi = [guarded(a.shape, a) for a in ins]
        xo, vo, Co, Fo, po = (guarded(a.shape) for a in ins[:5])
        ck = guarded((L.ud_plb_ckpt_bytes(sim._h, C.c_int(B)) // 8,))
        p = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.ud_plb_step_fwd(sim._h, C.c_int(B), *(p(t) for t in gi), p(xo), p(vo), p(Co), p(Fo), p(po), p(ck), st), "fwd")
        rng = np.random.default_rng(5)
        gouts = [guarded(a.shape, rng.normal(size=a.shape)) for a in ins[:5]]
        og = [guarded(a.shape) for a in ins[:5]] + [guarded((B, 3)), guarded((B,)), guarded((B,)), guarded((B,)), guarded((B,))]
        _lib.check(L.ud_plb_step_bwd(sim._h, C.c_int(B), p(ck), p(gi[5]), p(gi[6]), p(gi[7]), p(gi[8]), p(gi[9]),
                                     *(p(t) for t in gouts), *(p(t) for t in og), st), "bwd")
        torch.cuda.synchronize()
        for buf, n in bufs:
            assert (buf[:PAD] == CANARY).all() and (buf[PAD + n:] == CANARY).all(), "a kernel wrote outside the caller's array"
        return [t.cpu().numpy().copy() for t in (xo, vo, Co, Fo, po, *og)]

    big = _hip_sim(N, 8, path=path)
    run(big, 8, 2)
    got = run(big, 3, 3)
    ref = run(_hip_sim(N, 3, path=path), 3, 3)
    with pytest.raises(_lib.UnidomError):                      # more envs than the handle was created for: refused, nothing allocated
        run(_hip_sim(N, 2, path=path), 3, 3)
    for g, r in zip(got, ref):
        assert np.isfinite(g).all() and _rel(g, r) < 1e-9
