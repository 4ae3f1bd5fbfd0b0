"""CPU: pin the MPM oracle.

Primary pin = the reference's OWN recorded trajectory expert_demo/whip_rope/demo_0.pkl (re-packed as data in
tests/golden/whip_rope_demo0.npz): 69 (state, action) -> state' transitions = 3450 substeps, replayed with the
legacy parameter overrides found in SURVEY.md F3 (steps=50, gripper size (.02,.06,.02), Lame parameters from
E=100, nu=0.2, ground friction 0.1, recorded action fed to simulator.step unchanged, position control, focus shift).
It pins the forward substep (M1-M5, M8, E3 focus shift).  Not pinned by reference data (-> validated against
torch.autograd on the line-by-line twin + f64 finite differences): every adjoint, the plastic branch.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.pyoracle import MpmOracle, svd3
from oracle.twin import mpm_twin as tw

torch.set_num_threads(2)
E, NU = 100, 0.2
MU0, LA0 = E / (2 * (1 + NU)), E * NU / ((1 + NU) * (1 - 2 * NU))
S_LEGACY = 50


@pytest.fixture(scope="module")
def demo():
    return np.load(os.path.join(GOLDEN, "whip_rope_demo0.npz"))


def legacy_state(d, k, dt=np.float32, S=S_LEGACY):
    """demo state k -> step input with the focus shift of mpm_env.py:99-114 applied."""
    x = d["x"][k].astype(dt)
    target = np.array([32, 32, 32], dt) * dt(0.5) / dt(64)
    shift = (target - x.mean(0, dtype=dt)).astype(dt)
    shift[1] = 0
    ppos = np.zeros((S, 3), dt)
    ppos[0] = d["prim_position"][k, 0]
    ppos = ppos + shift
    prot = np.zeros((S, 4), dt)
    prot[:, 0] = 1
    st = dict(x=(x + shift)[None], v=d["v"][k][None].astype(dt), C=d["C"][k][None].astype(dt), F=d["F"][k][None].astype(dt),
              J=d["J"][k][None].astype(dt), ppos=ppos[None], prot=prot[None], psize=np.array([[0.02, 0.06, 0.02]], dt),
              friction=np.array([0.1], dt), mu=np.array([MU0], dt), lamda=np.array([LA0], dt),
              action=d["action"][k][None].astype(dt))
    return st, shift


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_golden_all_transitions_cpp(demo):
    """All 69 recorded transitions, one step (50 substeps) ahead."""
    d = demo
    orc = MpmOracle(67, steps=S_LEGACY)
    for k in range(69):
        st, shift = legacy_state(d, k)
        o = orc.step_fwd(st)
        assert np.abs(o["x"][0] - shift - d["x"][k + 1]).max() < 5e-7          # measured 1.2e-7
        assert _rel(o["v"][0], d["v"][k + 1]) < 3e-5                            # measured 7.4e-6
        assert _rel(o["C"][0], d["C"][k + 1]) < 1e-4                            # measured 2.7e-5
        assert np.abs(o["F"][0] - d["F"][k + 1]).max() < 2e-5                   # measured 4.6e-6
        np.testing.assert_array_equal(o["ppos"][0, 0] - shift, d["prim_position"][k + 1, 0])   # Q5: steps-1 increments
        np.testing.assert_array_equal(o["J"][0], d["J"][k + 1])


def test_golden_chained_rollout_cpp(demo):
    """20 chained steps = 1000 substeps from state 0 (errors may accumulate)."""
    d = demo
    orc = MpmOracle(67, steps=S_LEGACY)
    st, _ = legacy_state(d, 0)
    cur = {k: st[k] for k in ("v", "C", "F", "J")}
    cur["x"] = d["x"][0][None].astype(np.float32)
    ppos0 = d["prim_position"][0, 0].astype(np.float32)
    for k in range(20):
        x = cur["x"][0]
        shift = (np.float32([0.25, 0, 0.25]) - x.mean(0, dtype=np.float32)).astype(np.float32)
        shift[1] = 0
        ppos = np.zeros((S_LEGACY, 3), np.float32)
        ppos[0] = ppos0
        s = dict(st, x=(x + shift)[None], v=cur["v"], C=cur["C"], F=cur["F"], J=cur["J"], ppos=(ppos + shift)[None],
                 action=d["action"][k][None])
        o = orc.step_fwd(s)
        cur = dict(x=o["x"] - shift, v=o["v"], C=o["C"], F=o["F"], J=o["J"])
        ppos0 = o["ppos"][0, 0] - shift
    assert np.abs(cur["x"][0] - d["x"][20]).max() < 2e-6                        # measured 4e-7
    assert _rel(cur["v"][0], d["v"][20]) < 5e-5
    assert np.abs(ppos0 - d["prim_position"][20, 0]).max() < 1e-6


def test_golden_twin(demo):
    """The literal torch twin (dense jnp-style code, LAPACK svd) reproduces the recording too."""
    d = demo
    conf = tw.MPMConf(steps=S_LEGACY)
    sim = tw.MPMTwin(conf, 67, clip_grads=False)
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    with torch.no_grad():
        for k in (1, 40):
            p = tw.make_prim(conf, [0.02, 0.06, 0.02], d["prim_position"][k, 0])
            st = tw.MPMState(t(d["x"][k]), t(d["v"][k]), t(d["C"][k]), t(d["F"][k]), t(d["J"][k]), [p], t([0.1]), t([MU0]), t([LA0]))
            shift = tw.focus_shift(conf, st.x)
            st = st._replace(x=st.x + shift, primitives=[p._replace(position=p.position + shift)])
            st = sim.step(st, t(d["action"][k]))
            assert np.abs((st.x - shift).numpy() - d["x"][k + 1]).max() < 5e-7
            assert _rel(st.v.numpy(), d["v"][k + 1]) < 3e-5
            assert _rel(st.C.numpy(), d["C"][k + 1]) < 1e-4


def test_svd3_properties():
    rng = np.random.default_rng(0)
    for dt, tol in ((np.float64, 1e-13), (np.float32, 5e-6)):
        for _ in range(50):
            A = (np.eye(3) + rng.normal(size=(3, 3)) * 0.3).astype(dt)
            U, S, Vh = svd3(A)
            assert np.abs(U @ np.diag(S) @ Vh - A).max() < tol
            assert np.abs(U.T @ U - np.eye(3)).max() < tol and np.abs(Vh @ Vh.T - np.eye(3)).max() < tol
            assert S[0] >= S[1] >= S[2] >= 0
            np.testing.assert_allclose(S, np.linalg.svd(A.astype(np.float64), compute_uv=False), rtol=0, atol=tol * 10)
    U, S, Vh = svd3(np.eye(3, dtype=np.float32))   # degenerate: triple singular value
    np.testing.assert_array_equal(S, [1, 1, 1])
    np.testing.assert_array_equal(U @ Vh, np.eye(3, dtype=np.float32))


def _adjoint_case(d, S, k, material, seed, dtn=np.float64):
    rng = np.random.default_rng(seed)
    st, shift = legacy_state(d, k, dtn, S)
    st["ppos"][0, 0] = st["x"][0, 5]                     # gripper on the rope: control touches occupied cells
    st["F"] = st["F"] + rng.normal(size=st["F"].shape) * 0.05
    st["action"] = np.array([[0.3, -0.2, 0.5, 0, 0, 0]], dtn)
    g = dict(gx=rng.normal(size=(1, 67, 3)), gv=rng.normal(size=(1, 67, 3)), gC=rng.normal(size=(1, 67, 3, 3)) * 0.01,
             gF=rng.normal(size=(1, 67, 3, 3)) * 0.1, gppos=rng.normal(size=(1, S, 3)))
    return st, {kk: v.astype(dtn) for kk, v in g.items()}


@pytest.mark.parametrize("clip,material,S,k", [(False, 1, 3, 40), (True, 1, 3, 40), (False, 2, 3, 20)])
def test_adjoint_matches_autograd_through_twin_f64(demo, clip, material, S, k):
    """Hand-derived adjoint (incl. _svd_bwd, friction/boundary/control masks, set_action, copy_frame, the
    step-boundary nan_to_num + global-norm clip) == torch.autograd through the literal twin, f64."""
    st, g = _adjoint_case(demo, S, k, material, 0)
    orc = MpmOracle(67, steps=S, material=np.full(67, material))
    of, ob = orc.step_fwd(st), orc.step_bwd(st, g, clip=clip)
    conf = tw.MPMConf(steps=S)
    sim = tw.MPMTwin(conf, 67, material=material, dtype=torch.float64, clip_grads=clip)
    L = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    xt, vt, Ct, Ft, pp = L(st["x"][0]), L(st["v"][0]), L(st["C"][0]), L(st["F"][0]), L(st["ppos"][0])
    fr, mu, la, ac = L(st["friction"]), L(st["mu"]), L(st["lamda"]), L(st["action"][0])
    p = tw.make_prim(conf, [0.02, 0.06, 0.02], [0, 0, 0], torch.float64)._replace(position=pp)
    s2 = sim.step(tw.MPMState(xt, vt, Ct, Ft, torch.tensor(st["J"][0]), [p], fr, mu, la), ac)
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    loss = (s2.x * T(g["gx"][0])).sum() + (s2.v * T(g["gv"][0])).sum() + (s2.C * T(g["gC"][0])).sum() + \
        (s2.F * T(g["gF"][0])).sum() + (s2.primitives[0].position * T(g["gppos"][0])).sum()
    loss.backward()
    for key, ref in (("x", s2.x), ("v", s2.v), ("C", s2.C), ("F", s2.F)):
        assert _rel(of[key][0], ref.detach().numpy()) < 1e-12
    for key, ref in (("gx", xt), ("gv", vt), ("gC", Ct), ("gF", Ft), ("gppos", pp)):
        assert _rel(ob[key][0], ref.grad.numpy()) < 1e-10, key
    for key, ref in (("gfriction", fr), ("gmu", mu), ("glamda", la)):
        assert _rel(ob[key], ref.grad.numpy()) < 1e-9, key
    assert _rel(ob["gaction"][0], ac.grad.numpy()) < 1e-10
    assert np.abs(ob["gaction"][0, :3]).max() > 0


def test_adjoint_vs_finite_differences_f64(demo):
    S = 2
    st, g = _adjoint_case(demo, S, 40, 1, 3)
    orc = MpmOracle(67, steps=S)

    def Lf(s):
        o = orc.step_fwd(s)
        return (o["x"] * g["gx"]).sum() + (o["v"] * g["gv"]).sum() + (o["C"] * g["gC"]).sum() + \
            (o["F"] * g["gF"]).sum() + (o["ppos"] * g["gppos"]).sum()

    b = orc.step_bwd(st, g, clip=False)
    rng = np.random.default_rng(0)
    for name, gname in (("x", "gx"), ("v", "gv"), ("C", "gC"), ("F", "gF"), ("mu", "gmu"), ("lamda", "glamda"), ("friction", "gfriction")):
        for _ in range(4):
            a = st[name]
            idx = tuple(rng.integers(0, s) for s in a.shape)
            h = 1e-6 * max(1.0, abs(a[idx]))
            sp, sm = dict(st), dict(st)
            sp[name], sm[name] = a.copy(), a.copy()
            sp[name][idx] += h
            sm[name][idx] -= h
            fd = (Lf(sp) - Lf(sm)) / (2 * h)
            an = b[gname][idx]
            assert abs(fd - an) <= 2e-5 * max(1.0, abs(fd), abs(an)), (name, idx, fd, an)


def _collide_case(d, S, k, material, seed, dtn=np.float64, w=(0.4, -0.3, 0.2)):
    """Soft-contact (collide_batch) case: the gripper sits in the rope, rotated, and turns during the step."""
    st, g = _adjoint_case(d, S, k, material, seed, dtn)
    rng = np.random.default_rng(seed + 100)
    q = np.array([0.9, 0.2, -0.3, 0.1], dtn)
    st["prot"][0, :] = q / np.linalg.norm(q)
    st["ppos"][0, 0] = st["x"][0, 5] + np.array([0.004, 0.002, -0.003], dtn)
    st["action"] = np.array([[0.3, -0.2, 0.5, *w]], dtn)
    g["gprot"] = rng.normal(size=(1, S, 4)).astype(dtn)
    return st, g


@pytest.mark.parametrize("clip,material,S,k", [(False, 1, 3, 40), (True, 1, 3, 40), (False, 2, 2, 20)])
def test_collide_adjoint_matches_autograd_through_twin_f64(demo, clip, material, S, k):
    """collide_batch (primitives.py:154-182: sdf, finite-difference normal, collider velocity, friction flag) and the
    rotation chain (qmul / w2quat / forward_kinematics) -- hand-derived adjoint == torch.autograd through the twin, f64."""
    st, g = _collide_case(demo, S, k, material, 0)
    orc = MpmOracle(67, steps=S, material=np.full(67, material), position_control=False)
    of, ob = orc.step_fwd(st), orc.step_bwd(st, g, clip=clip)
    conf = tw.MPMConf(steps=S)
    sim = tw.MPMTwin(conf, 67, material=material, dtype=torch.float64, clip_grads=clip, use_position_control=False)
    L = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    xt, vt, Ct, Ft, pp, pr = L(st["x"][0]), L(st["v"][0]), L(st["C"][0]), L(st["F"][0]), L(st["ppos"][0]), L(st["prot"][0])
    fr, mu, la, ac = L(st["friction"]), L(st["mu"]), L(st["lamda"]), L(st["action"][0])
    p = tw.make_prim(conf, [0.02, 0.06, 0.02], [0, 0, 0], torch.float64)._replace(position=pp, rotation=pr)
    p = p._replace(size=p.size.clone().requires_grad_(True), friction=p.friction.clone().requires_grad_(True))
    s2 = sim.step(tw.MPMState(xt, vt, Ct, Ft, torch.tensor(st["J"][0]), [p], fr, mu, la), ac)
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    loss = (s2.x * T(g["gx"][0])).sum() + (s2.v * T(g["gv"][0])).sum() + (s2.C * T(g["gC"][0])).sum() + \
        (s2.F * T(g["gF"][0])).sum() + (s2.primitives[0].position * T(g["gppos"][0])).sum() + \
        (s2.primitives[0].rotation * T(g["gprot"][0])).sum()
    loss.backward()
    for key, ref in (("x", s2.x), ("v", s2.v), ("C", s2.C), ("F", s2.F)):
        assert _rel(of[key][0], ref.detach().numpy()) < 1e-11, key
    assert _rel(of["prot"][0], s2.primitives[0].rotation.detach().numpy()) < 1e-13
    # the finite-difference normal amplifies round-off by 0.5/d = 5e5: 1e-16 * 5e5 ~ 5e-11 relative in every cotangent
    for key, ref in (("gx", xt), ("gv", vt), ("gC", Ct), ("gF", Ft), ("gppos", pp), ("gprot", pr)):
        assert _rel(ob[key][0], ref.grad.numpy()) < 2e-7, key
    for key, ref in (("gfriction", fr), ("gmu", mu), ("glamda", la)):
        assert _rel(ob[key], ref.grad.numpy()) < 2e-7, key
    assert _rel(ob["gaction"][0], ac.grad.numpy()) < 2e-7
    assert np.abs(ob["gaction"][0, 3:]).min() > 0 and np.abs(ob["gprot"]).max() > 0


def test_collide_zero_rotation_action_is_nan_then_laundered(demo):
    """w = 0: d|w|/dw = 0.5/0 * 0 -> NaN in the reference's chain rule (jnp.linalg.norm, primitives.py:86); the
    step-boundary nan_to_num (mpm_simulator.py:403-408) turns it into 0."""
    st, g = _collide_case(demo, 2, 40, 1, 1, w=(0, 0, 0))
    orc = MpmOracle(67, steps=2, position_control=False)
    raw = orc.step_bwd(st, g, clip=False)
    assert np.isnan(raw["gaction"][0, 3:]).all() and np.isfinite(raw["gaction"][0, :3]).all()
    assert np.isfinite(raw["gx"]).all() and np.isfinite(raw["gprot"]).all()
    lau = orc.step_bwd(st, g, clip=True)
    assert (lau["gaction"][0, 3:] == 0).all() and np.isfinite(lau["gaction"]).all()


def _two_bowl_case(d, S, k, seed, dtn=np.float64, turning=True):
    """pour_water-like: liquid particles, two container primitives (cut hollow spheres), the first one moving and turning."""
    st, g = _adjoint_case(d, S, k, 0, seed, dtn)
    rng = np.random.default_rng(seed + 200)
    c = st["x"][0].mean(0)
    ppos = np.zeros((1, 2, S, 3), dtn)
    ppos[0, 0, 0] = c + np.array([0.01, 0.03, -0.005], dtn)          # bowl under / around the body
    ppos[0, 1, 0] = c + np.array([-0.03, -0.02, 0.06], dtn)
    prot = np.zeros((1, 2, S, 4), dtn)
    q = np.array([[0.96, 0.2, -0.1, 0.05], [1, 0, 0, 0]], dtn)
    prot[0] = (q / np.linalg.norm(q, axis=1, keepdims=True))[:, None]
    st.update(ppos=ppos, prot=prot, psize=np.array([[[0.05, 0.0, 0.008], [0.04, 0.01, 0.008]]], dtn),
              action=np.array([[0.3, -0.2, 0.5, 0.4, -0.3, 0.2, 0.1, 0.05, -0.1, 0.2, 0.1, -0.3]], dtn) * dtn(0.02))   # ~10 m/s colliders
    if not turning:                                                     # upright bowls that only translate
        st["prot"][:] = 0
        st["prot"][..., 0] = 1
        st["action"][0, [3, 4, 5, 9, 10, 11]] = 0
    g["gppos"] = rng.normal(size=(1, 2, S, 3)).astype(dtn)
    g["gprot"] = rng.normal(size=(1, 2, S, 4)).astype(dtn)
    return st, g


@pytest.mark.parametrize("clip", [False, True])
def test_two_container_primitives_adjoint_matches_autograd_f64(demo, clip):
    """n_primitive = 2 with the container SDF (container.py:8-16), collide_batch applied primitive after primitive
    (mpm_simulator.py:292-294), liquid material: forward and hand-derived adjoint == torch.autograd through the twin."""
    S = 3
    st, g = _two_bowl_case(demo, S, 40, 0)
    orc = MpmOracle(67, steps=S, material=np.zeros(67), position_control=False, n_prim=2, sdf="container")
    of, ob = orc.step_fwd(st), orc.step_bwd(st, g, clip=clip)
    conf = tw.MPMConf(steps=S, n_primitive=2)
    tw.set_sdf(tw.container_sdf)
    try:
        sim = tw.MPMTwin(conf, 67, material=0, dtype=torch.float64, clip_grads=clip, use_position_control=False)
        L = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
        xt, vt, Ct, Ft = L(st["x"][0]), L(st["v"][0]), L(st["C"][0]), L(st["F"][0])
        fr, mu, la, ac = L(st["friction"]), L(st["mu"]), L(st["lamda"]), L(st["action"][0])
        pps, prs, prims = [], [], []
        for i in range(2):
            pp, pr = L(st["ppos"][0, i]), L(st["prot"][0, i])
            p = tw.make_prim(conf, st["psize"][0, i], [0, 0, 0], torch.float64)._replace(position=pp, rotation=pr)
            prims.append(p._replace(size=p.size.clone().requires_grad_(True), friction=p.friction.clone().requires_grad_(True)))
            pps.append(pp)
            prs.append(pr)
        s2 = sim.step(tw.MPMState(xt, vt, Ct, Ft, torch.tensor(st["J"][0]), prims, fr, mu, la), ac)
    finally:
        tw.set_sdf(tw.box_sdf)
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    loss = (s2.x * T(g["gx"][0])).sum() + (s2.v * T(g["gv"][0])).sum() + (s2.C * T(g["gC"][0])).sum() + (s2.F * T(g["gF"][0])).sum()
    for i in range(2):
        loss = loss + (s2.primitives[i].position * T(g["gppos"][0, i])).sum() + (s2.primitives[i].rotation * T(g["gprot"][0, i])).sum()
    loss.backward()
    for key, ref in (("x", s2.x), ("v", s2.v), ("C", s2.C), ("F", s2.F)):
        assert _rel(of[key][0], ref.detach().numpy()) < 1e-11, key
    for i in range(2):
        assert _rel(of["ppos"][0, i], s2.primitives[i].position.detach().numpy()) < 1e-13
        assert _rel(of["prot"][0, i], s2.primitives[i].rotation.detach().numpy()) < 1e-13
    for key, ref in (("gx", xt), ("gv", vt), ("gC", Ct), ("gF", Ft)):
        assert _rel(ob[key][0], ref.grad.numpy()) < 2e-7, key
    for i in range(2):
        assert _rel(ob["gppos"][0, i], pps[i].grad.numpy()) < 2e-7, i
        assert _rel(ob["gprot"][0, i], prs[i].grad.numpy()) < 2e-7, i
        assert np.abs(ob["gprot"][0, i]).max() > 0
    assert _rel(ob["gfriction"], fr.grad.numpy()) < 2e-7
    assert _rel(ob["gaction"][0], ac.grad.numpy()) < 2e-7 and np.abs(ob["gaction"][0]).min() > 0


def test_two_primitives_with_their_own_friction_and_softness_f64(demo):
    """create_primitive passes friction / softness PER primitive (mpm_env.py:201-217; every reference env gives all of them 0.1 /
    666): the restatement with one pair per primitive == the twin whose two PrimitiveStates carry different values, forward and
    adjoint; and it is not the uniform case (the second bowl's values matter)."""
    S = 3
    st, g = _two_bowl_case(demo, S, 40, 0)
    fr_each, so_each = [0.1, 0.45], [666.0, 120.0]
    orc = MpmOracle(67, steps=S, material=np.zeros(67), position_control=False, n_prim=2, sdf="container",
                    prim_friction=fr_each, prim_softness=so_each)
    of, ob = orc.step_fwd(st), orc.step_bwd(st, g, clip=False)
    uni = MpmOracle(67, steps=S, material=np.zeros(67), position_control=False, n_prim=2, sdf="container").step_fwd(st)
    assert _rel(of["v"], uni["v"]) > 1e-6
    conf = tw.MPMConf(steps=S, n_primitive=2)
    tw.set_sdf(tw.container_sdf)
    try:
        sim = tw.MPMTwin(conf, 67, material=0, dtype=torch.float64, clip_grads=False, use_position_control=False)
        L = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
        xt, vt, Ct, Ft = L(st["x"][0]), L(st["v"][0]), L(st["C"][0]), L(st["F"][0])
        fr, mu, la, ac = L(st["friction"]), L(st["mu"]), L(st["lamda"]), L(st["action"][0])
        prims, pps = [], []
        for i in range(2):
            pp, pr = L(st["ppos"][0, i]), L(st["prot"][0, i])
            p = tw.make_prim(conf, st["psize"][0, i], [0, 0, 0], torch.float64, friction=fr_each[i], softness=so_each[i])
            prims.append(p._replace(position=pp, rotation=pr))
            pps.append(pp)
        s2 = sim.step(tw.MPMState(xt, vt, Ct, Ft, torch.tensor(st["J"][0]), prims, fr, mu, la), ac)
    finally:
        tw.set_sdf(tw.box_sdf)
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    loss = (s2.x * T(g["gx"][0])).sum() + (s2.v * T(g["gv"][0])).sum() + (s2.C * T(g["gC"][0])).sum() + (s2.F * T(g["gF"][0])).sum()
    for i in range(2):
        loss = loss + (s2.primitives[i].position * T(g["gppos"][0, i])).sum() + (s2.primitives[i].rotation * T(g["gprot"][0, i])).sum()
    loss.backward()
    for key, ref in (("x", s2.x), ("v", s2.v), ("C", s2.C), ("F", s2.F)):
        assert _rel(of[key][0], ref.detach().numpy()) < 1e-11, key
    for key, ref in (("gx", xt), ("gv", vt), ("gC", Ct), ("gF", Ft)):
        assert _rel(ob[key][0], ref.grad.numpy()) < 2e-7, key
    for i in range(2):
        assert _rel(ob["gppos"][0, i], pps[i].grad.numpy()) < 2e-7, i
    assert _rel(ob["gaction"][0], ac.grad.numpy()) < 2e-7


@pytest.mark.parametrize("sdf,n_prim", [("box", 1), ("container", 2)])
def test_collide_adjoint_vs_finite_differences_f64(demo, sdf, n_prim):
    """Soft contact, independent of autograd: central differences of the oracle's own f64 forward against its hand-derived
    adjoint, for particle state, material parameters, primitive position / rotation rows and the action (translation and
    rotation parts) -- box with one primitive, container with two."""
    S = 2
    if sdf == "box":
        st, g = _collide_case(demo, S, 40, 1, 3)
        orc = MpmOracle(67, steps=S, position_control=False)
    else:
        st, g = _two_bowl_case(demo, S, 40, 3)
        orc = MpmOracle(67, steps=S, material=np.zeros(67), position_control=False, n_prim=2, sdf="container")

    def Lf(s):
        o = orc.step_fwd(s)
        return (o["x"] * g["gx"]).sum() + (o["v"] * g["gv"]).sum() + (o["C"] * g["gC"]).sum() + (o["F"] * g["gF"]).sum() + \
            (o["ppos"] * g["gppos"]).sum() + (o["prot"] * g["gprot"]).sum()

    b = orc.step_bwd(st, g, clip=False)
    rng = np.random.default_rng(0)
    # The forward contains a finite-difference normal of step 1e-6 (primitives.py:119-134) that amplifies f64 round-off by 5e5:
    # an outer difference of step h carries ~5e-11 / h of noise (5e-5 at h = 1e-6), while a larger h starts crossing the kinks of
    # the SDFs (measured: 1.5e-4 at h = 1e-5).  So this check is a coarse, autograd-independent one at 2e-4; the tight one is
    # the comparison with the twin above (2e-7).
    tol = 2e-4
    names = [("x", "gx"), ("v", "gv"), ("C", "gC"), ("F", "gF"), ("friction", "gfriction"), ("action", "gaction")]
    if sdf == "box":
        names += [("mu", "gmu"), ("lamda", "glamda")]          # the liquid of the container case has mu = 0, la = 1 whatever the state says
    for name, gname in names:
        for _ in range(4):
            a = st[name]
            idx = tuple(rng.integers(0, n) for n in a.shape)
            h = 1e-6 * max(1.0, abs(a[idx]))
            sp, sm = dict(st), dict(st)
            sp[name], sm[name] = a.copy(), a.copy()
            sp[name][idx] += h
            sm[name][idx] -= h
            fd = (Lf(sp) - Lf(sm)) / (2 * h)
            an = b[gname][idx]
            assert abs(fd - an) <= tol * max(1.0, abs(fd), abs(an)), (name, idx, fd, an)
    # primitive rows: only row 0 of position / rotation is an input that matters (rows >= 1 are overwritten by FK)
    for name, gname, width in (("ppos", "gppos", 3), ("prot", "gprot", 4)):
        for ip in range(n_prim):
            for d in range(width):
                idx = (0, ip, 0, d) if n_prim > 1 else (0, 0, d)
                a = st[name]
                h = 1e-6
                sp, sm = dict(st), dict(st)
                sp[name], sm[name] = a.copy(), a.copy()
                sp[name][idx] += h
                sm[name][idx] -= h
                fd = (Lf(sp) - Lf(sm)) / (2 * h)
                an = b[gname][idx]
                assert abs(fd - an) <= tol * max(1.0, abs(fd), abs(an)), (name, idx, fd, an)
