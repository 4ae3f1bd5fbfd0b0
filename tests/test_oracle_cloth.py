"""CPU: pin the cloth oracle.

The reference's cloth demos do NOT pin cloth x/v (legacy cloth step) -> "parity unpinned" for x/v;
what they do pin exactly is checked here (primitive kinematics through get_pnp_actions + robot_step's action
scaling + the primitive update).  Beyond that: analytic known answers, the C++ restatement against the
line-by-line torch twin (forward and autograd adjoint), and the adjoint against f64 finite differences.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, cloth_reset_x, fold_cloth1_mask, make_cloth_case
from oracle.pyoracle import ClothOracle
from oracle.twin.cloth_twin import ClothConf, ClothState, ClothTwin

torch.set_num_threads(1)


def pnp_actions(action6, prim0):
    """cloth_env.py:134-173 in numpy (single env)."""
    pick, place = action6[:3].copy(), action6[3:].copy()
    pick[1] = 0
    place[1] = 0
    rows = []
    down = np.ones(4, np.float32)
    down[:3] = (pick - prim0[:3]) * (np.float32(1) / np.float32(3))     # "/ 3" under jit = * f32(1/3): see the sibling-demo test
    rows += [down] * 3
    rows += [np.array([0, 0.06 / 10, 0, 0], np.float32)] * 10
    mv = place - pick
    mv[1] = 0
    rows += [np.concatenate([mv * (np.float32(1) / np.float32(20)), [0]]).astype(np.float32)] * 20
    rows += [np.array([0, 0, 0, 1], np.float32)] * 7
    a = np.stack(rows)
    return np.concatenate([a, np.zeros_like(a)], 1)[:, None, :]  # [40,1,8]


def test_demo_primitive_trajectory_exact():
    """Pinned by reference data: primitive0/1 after one step_diff reproduce the recorded values exactly."""
    d = np.load(os.path.join(GOLDEN, "fold_cloth1_demos.npz"))
    orc = ClothOracle(fold_cloth1_mask())
    for i in range(len(d["demo"])):
        prim = np.stack([d["s0_primitive0"][i], d["s0_primitive1"][i]])[None]
        acts = pnp_actions(d["action"][i].astype(np.float32), d["s0_primitive0"][i])
        o = orc.rollout_fwd(d["s0_x"][i][None], d["s0_v"][i][None], prim, np.array([900], np.float32),
                            d["s0_mu"][i][None].astype(np.float32), acts.astype(np.float32))
        np.testing.assert_array_equal(o["prim"][0, 0], d["s1_primitive0"][i])
        np.testing.assert_array_equal(o["prim"][0, 1], d["s1_primitive1"][i])
        assert d["s1_cur_step"][i] == d["s0_cur_step"][i] + 1


def test_sibling_demo_primitive_trajectories_keys_and_steps_exact():
    """The same exact pins from the reference's other cloth recordings: expert_demo/{fold_cloth3, unfold_cloth1, unfold_cloth3,
    fold_tshirt}/demo_*.pkl, 226 transitions (tests/golden/cloth_sibling_demos.npz).  primitive0/1 after one step_diff are a
    function of (action, primitive0, primitive1) only -- get_pnp_actions (cloth_env.py:134-173), robot_step's action scaling
    (cloth_simulator.py:166-171) and the per-substep primitive update + clip (:322-323) never read the cloth -- so the oracle is
    run on a small stand-in cloth; the recorded values are reproduced bit for bit, as are the 40-split key chain and cur_step.
    (The cloth x/v of these legacy recordings are not reproducible; an attempt to also pin unfold_cloth1's reset
    -- lattice + normal(split(key)[0]) * 1e-4 folded once from np.random.seed(1) picks -- against the recorded first primitive
    found no key / seed hypothesis that matches: the recorder's reset differed, like its cloth step.)"""
    from unidom_amd.utils import prng
    d = np.load(os.path.join(GOLDEN, "cloth_sibling_demos.npz"))
    assert {t: int((d["task"] == t).sum()) for t in np.unique(d["task"])} == {
        "fold_cloth3": 30, "fold_tshirt": 39, "unfold_cloth1": 55, "unfold_cloth3": 102}
    mask = np.zeros((80, 80), np.float32)
    mask[38:42, 38:42] = 1
    orc = ClothOracle(mask)
    ii, jj = np.nonzero(mask)
    x = np.stack([ii / 80, np.zeros(16), (80 - jj) / 80], -1).astype(np.float32)[None]
    n = len(d["task"])
    prim = np.stack([d["s0_primitive0"], d["s0_primitive1"]], 1).astype(np.float32)            # [n,2,4]
    acts = np.concatenate([pnp_actions(d["action"][i].astype(np.float32), d["s0_primitive0"][i]) for i in range(n)], 1)
    o = orc.rollout_fwd(np.repeat(x, n, 0), np.zeros((n, 16, 3), np.float32), prim, np.full(n, 900, np.float32),
                        np.full(n, 0.9, np.float32), acts.astype(np.float32), nthreads=4)
    np.testing.assert_array_equal(o["prim"][:, 0], d["s1_primitive0"])
    np.testing.assert_array_equal(o["prim"][:, 1], d["s1_primitive1"])
    np.testing.assert_array_equal(prng.split_first(d["s0_key"], 40), d["s1_key"])
    np.testing.assert_array_equal(d["s1_cur_step"], d["s0_cur_step"] + 1)


def test_terminal_velocity_known_answer():
    """Flat cloth on the ground: gravity enters twice per substep (Q1, :259 and :278) ->
    v_y* = -(2 g dt) e^{-c dt} / (1 - e^{-c dt}) ~= -0.499 (g=0.5, dt=2e-3, c=2)."""
    orc = ClothOracle(fold_cloth1_mask())
    x = cloth_reset_x()[None]
    v = np.zeros_like(x)
    prim = np.array([[[0.5, 0.5, 0.5, 0.01], [1, 1, 1, 0.01]]], np.float32)
    acts = np.zeros((40, 1, 8), np.float32)
    o = orc.rollout_fwd(x, v, prim, np.array([900], np.float32), np.array([0.5], np.float32), acts)
    e = np.exp(-2 * 2e-3)
    expect = -(2 * 0.5 * 2e-3) * e / (1 - e)
    assert abs(expect + 0.499) < 1e-3
    np.testing.assert_allclose(o["v"][0, :, 1], expect, rtol=2e-3)
    # resting lattice: in-plane forces are f32 round-off of the lattice spacing times k/L0 = 72000
    np.testing.assert_allclose(o["v"][0, :, [0, 2]], 0, atol=2e-3)


def test_gripped_particle_follows_gripper():
    """suction=0 grasp: the gripped particle is displaced by the per-substep action (x += d_v, v = 0)."""
    orc = ClothOracle(fold_cloth1_mask(), substeps=1)
    x = cloth_reset_x()[None]
    v = np.zeros_like(x)
    p = 200
    prim = np.array([[[*x[0, p], 0.01], [1, 1, 1, 0.01]]], np.float32)
    acts = np.zeros((1, 1, 8), np.float32)
    acts[0, 0, :3] = [0.0, 0.5, 0.0]       # /50 per substep
    o = orc.rollout_fwd(x, v, prim, np.array([900], np.float32), np.array([0.5], np.float32), acts, want_grasp=True)
    assert o["grasp"][0, 0, 0, 0].nonzero()[0].tolist() == [p]          # radius 0.01 < spacing 0.0125 (Q3)
    np.testing.assert_allclose(o["x"][0, p], x[0, p] + np.float32([0, 0.01, 0]), atol=1e-7)
    np.testing.assert_array_equal(o["v"][0, p], 0)
    np.testing.assert_allclose(o["prim"][0, 0, :3], x[0, p] + np.float32([0, 0.01, 0]), atol=1e-7)


def _twin_run(dtype_np, dtype_t, normalize, S=4, T=2, seed=0):
    conf = ClothConf()
    conf.substeps = S
    mask = fold_cloth1_mask()
    rng = np.random.default_rng(seed)
    x, v, prim, k, mu, actions = [a.astype(dtype_np) for a in make_cloth_case(rng, 1, T)]
    tw = ClothTwin(conf, mask, dtype=dtype_t, normalize=normalize)
    leaf = lambda a: torch.tensor(a, requires_grad=True)
    xt, vt, p0, p1, kt, mut, at = leaf(x[0]), leaf(v[0]), leaf(prim[0, 0]), leaf(prim[0, 1]), leaf(k[0]), leaf(mu[0]), leaf(actions[:, 0])
    s = ClothState(xt, vt, p0, p1, torch.zeros(4, dtype=dtype_t), torch.zeros(4, dtype=dtype_t), kt, mut)
    rec = []
    for t in range(T):
        s = tw.robot_step(s, at[t], record=rec)
    g = [rng.normal(size=a.shape).astype(dtype_np) for a in (x[0], v[0], prim[0])]
    loss = (s.x * torch.tensor(g[0])).sum() + (s.v * torch.tensor(g[1])).sum() + \
           (s.primitive0 * torch.tensor(g[2][0])).sum() + (s.primitive1 * torch.tensor(g[2][1])).sum()
    loss.backward()
    orc = ClothOracle(mask, substeps=S)
    of = orc.rollout_fwd(x, v, prim, k, mu, actions, want_grasp=True)
    ob = orc.rollout_bwd(x, v, prim, k, mu, actions, g[0][None], g[1][None], g[2][None], normalize=normalize)
    twin = dict(x=s.x.detach().numpy(), v=s.v.detach().numpy(), gx=xt.grad.numpy(), gv=vt.grad.numpy(),
                gp0=p0.grad.numpy(), ga=at.grad.numpy(), gk=kt.grad.numpy(), gmu=mut.grad.numpy(), rec=rec)
    return twin, of, ob


@pytest.mark.parametrize("normalize", [True, False])
def test_cpp_oracle_matches_torch_twin_f64(normalize):
    """C++ restatement == line-by-line twin (forward) and == torch.autograd through the twin (adjoint, incl. the
    norm_grad custom VJP and the 0.5 tie-split clip gradients), f64, 1e-10 relative."""
    tw, of, ob = _twin_run(np.float64, torch.float64, normalize)
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-300)
    assert rel(of["x"][0], tw["x"]) < 1e-12 and rel(of["v"][0], tw["v"]) < 1e-10
    S = 4
    for i, (g0, g1) in enumerate(tw["rec"]):
        assert of["grasp"][i // S, i % S, 0, 0].nonzero()[0].tolist() == g0
    assert sum(len(r[0]) for r in tw["rec"]) > 0
    for a, b in ((ob["gx"][0], tw["gx"]), (ob["gv"][0], tw["gv"]), (ob["gprim"][0, 0], tw["gp0"]),
                 (ob["gactions"][:, 0], tw["ga"]), (ob["gk"][0], tw["gk"]), (ob["gmu"][0], tw["gmu"])):
        assert rel(a, b) < 1e-9, rel(a, b)


def test_cpp_oracle_matches_torch_twin_f32():
    """f32: torch's CPU sqrt is not correctly rounded, so only f32 round-off agreement is expected here."""
    tw, of, ob = _twin_run(np.float32, torch.float32, True)
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    assert rel(of["x"][0], tw["x"]) < 1e-6 and rel(of["v"][0], tw["v"]) < 2e-4
    assert rel(ob["gx"][0], tw["gx"]) < 1e-3 and rel(ob["gactions"][:, 0], tw["ga"]) < 1e-3


def test_adjoint_vs_finite_differences_f64():
    """True adjoint (normalize=False) against central differences of the f64 forward."""
    mask = fold_cloth1_mask()
    S, T = 3, 2
    orc = ClothOracle(mask, substeps=S)
    rng = np.random.default_rng(5)
    x, v, prim, k, mu, actions = [a.astype(np.float64) for a in make_cloth_case(rng, 1, T)]
    actions[..., 3] = 0.3   # partial suction keeps the action gradient non-trivial
    prim[:, 1, :3] = 0.9    # away from the clip bound 1.0 (ties have one-sided derivatives)
    gx, gv, gp = [rng.normal(size=a.shape) for a in (x, v, prim)]

    def L(x_, v_, prim_, k_, mu_, a_):
        o = orc.rollout_fwd(x_, v_, prim_, k_, mu_, a_)
        return (o["x"] * gx).sum() + (o["v"] * gv).sum() + (o["prim"] * gp).sum()

    b = orc.rollout_bwd(x, v, prim, k, mu, actions, gx, gv, gp, normalize=False)
    args = [x, v, prim, k, mu, actions]
    names = ["gx", "gv", "gprim", "gk", "gmu", "gactions"]
    for ai, nm in enumerate(names):
        a = args[ai]
        for _ in range(6):
            idx = tuple(rng.integers(0, s) for s in a.shape)
            if nm == "gprim" and idx[-1] == 3:
                continue  # radius: the grasp mask is not differentiable
            h = 1e-6 * max(1.0, abs(a[idx]))
            ap, am = [q.copy() for q in args], [q.copy() for q in args]
            ap[ai][idx] += h
            am[ai][idx] -= h
            fd = (L(*ap) - L(*am)) / (2 * h)
            an = b[nm][idx]
            assert abs(fd - an) <= 1e-5 * max(1.0, abs(fd), abs(an)), (nm, idx, fd, an)


def test_order_v2_matches_reference_order_short_horizon():
    """The re-associated IEEE order "v2" (what the default HIP forward computes) against the reference's literal
    operation order: f32 round-off per substep (this stiff system amplifies it over long horizons);
    in f64 the two orders agree to 1e-9 over 50 substeps."""
    mask = fold_cloth1_mask()
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    for S, tol_x, tol_v in ((1, 5e-7, 1e-4), (5, 1e-6, 2e-4)):
        o1, o2 = ClothOracle(mask, substeps=S), ClothOracle(mask, substeps=S, order=2)
        rng = np.random.default_rng(3)
        x, v, prim, k, mu, a = make_cloth_case(rng, 2, 2, deform=0.0005, v_scale=0.01)
        r1, r2 = o1.rollout_fwd(x, v, prim, k, mu, a, want_grasp=True), o2.rollout_fwd(x, v, prim, k, mu, a, want_grasp=True)
        np.testing.assert_array_equal(r1["grasp"], r2["grasp"])
        np.testing.assert_array_equal(r1["prim"], r2["prim"])
        assert rel(r2["x"], r1["x"]) < tol_x and rel(r2["v"], r1["v"]) < tol_v
    o1, o2 = ClothOracle(mask), ClothOracle(mask, order=2)
    x, v, prim, k, mu, a = [q.astype(np.float64) for q in make_cloth_case(np.random.default_rng(3), 1, 1, deform=0.0005, v_scale=0.01)]
    r1, r2 = o1.rollout_fwd(x, v, prim, k, mu, a), o2.rollout_fwd(x, v, prim, k, mu, a)
    assert rel(r2["x"], r1["x"]) < 1e-9 and rel(r2["v"], r1["v"]) < 1e-7
