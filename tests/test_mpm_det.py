"""Deterministic MPM mode (ud_mpm_conf.deterministic, include/unidom_hip.h).

The reference's p2g is a scatter-add that XLA's CPU backend applies in the order of the flattened update array, which is
[27 offsets][N particles] (mpm_simulator.py:178-194, :259-274): a cell receives its contributions ordered by (offset, particle index).
The fast kernels sum in the arrival order of atomics, so every other MPM comparison carries a tolerance.  In this mode each cell is summed
in that order, in IEEE f32 -- a group of 32 lanes per cell on the device (each lane fetches one offset's bucket, the running sums pass
through the lanes in order), one thread on the host: the same additions in the same order -- and the tests below are bit-for-bit:
  * two GPU runs of the same step are identical,
  * x, v, C, F after 70 substeps equal the CPU build of the same per-element source (oracle/csrc/mpm_det_host.cpp),
and, on the CPU, that source is held against the independent restatement of the reference (oracle/csrc/mpm_oracle.hpp) within the
tolerance their different SVDs leave -- the same-order checker pins order and arithmetic, the independent one the algorithm.
Round 4: soft contact in the same mode (collide_batch compiled into both builds), and a reproducible BACKWARD in both contact modes (two calls,
the same bits in every gradient; checked against the oracle by tolerance -- its arithmetic is the fast kernels')."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_mpm import S_LEGACY, _rel, legacy_state


@pytest.fixture(scope="module")
def demo():
    return np.load(os.path.join(GOLDEN, "whip_rope_demo0.npz"))


def _batch(demo, ks, S):
    sts = [legacy_state(demo, k, S=S)[0] for k in ks]
    return {key: np.concatenate([s[key] for s in sts]) for key in sts[0]}


def test_same_order_source_follows_the_reference_restatement(demo):
    """whip_rope's recorded states, 70 substeps: the deterministic mode's source (CPU build) against the independent oracle."""
    from oracle.pyoracle import MpmOracle, mpm_det_forward
    st = _batch(demo, [0, 20, 40], 70)
    ref = MpmOracle(67, steps=70).step_fwd(st)
    got = mpm_det_forward(st, 67, steps=70)
    assert _rel(got["x"], ref["x"]) < 2e-6 and _rel(got["v"], ref["v"]) < 1e-4, (_rel(got["x"], ref["x"]), _rel(got["v"], ref["v"]))
    assert _rel(got["C"], ref["C"]) < 1e-3 and _rel(got["F"], ref["F"]) < 5e-5, (_rel(got["C"], ref["C"]), _rel(got["F"], ref["F"]))
    # the recorded next states themselves (50 substeps, the legacy step of the recording: SURVEY.md F3)
    st50 = _batch(demo, [5], S_LEGACY)
    o = mpm_det_forward(st50, 67, steps=S_LEGACY)
    shift = legacy_state(demo, 5)[1]
    assert np.abs(o["x"][0] - shift - demo["x"][6]).max() < 5e-7
    assert _rel(o["v"][0], demo["v"][6]) < 3e-5


def _det_sim(steps, B, deterministic, N=67, conf_cls=None):
    from test_mpm_gpu import LegacyConf
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    conf = (conf_cls or LegacyConf)()
    conf.steps = steps
    conf.deterministic = deterministic
    sim = SimpleMPMSimulator(conf, B, use_position_control=True)
    sim.n_particles = N
    sim.material = np.full(N, 1, np.int32)
    sim.h = np.ones(N, np.float32)
    sim._make_handle()
    return sim


@pytest.mark.gpu
def test_deterministic_forward_is_bit_identical_to_the_same_order_checker_over_70_substeps(demo):
    from oracle.pyoracle import mpm_det_forward
    from test_mpm_gpu import run_hip
    st = _batch(demo, [0, 20, 40, 60], 70)
    sim = _det_sim(70, 4, 1)
    a = run_hip(sim, st)
    b = run_hip(sim, st)
    for key in ("x", "v", "C", "F", "J", "ppos", "prot"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key + ": two runs differ")
    chk = mpm_det_forward(st, 67, steps=70)
    for key in ("x", "v", "C", "F"):
        np.testing.assert_array_equal(a[key], chk[key], err_msg=key + ": GPU != CPU build of the same source")
    # and it is the same physics as the default kernels, to their tolerance
    fast = run_hip(_det_sim(70, 4, 0), st)
    assert _rel(a["x"], fast["x"]) < 5e-6 and _rel(a["v"], fast["v"]) < 1e-4 and _rel(a["F"], fast["F"]) < 5e-5
    np.testing.assert_array_equal(a["J"].shape, fast["J"].shape)


@pytest.mark.gpu
def test_deterministic_mode_on_a_many_workgroup_body(demo):
    """rope seeded at n_grid 128 (N = 798, res 64^3): the shape the many-workgroup kernels serve by default"""
    from oracle.pyoracle import mpm_det_forward
    from test_mpm_gpu import ScaledConf, _scaled_case, run_hip

    class DetScaled(ScaledConf):
        deterministic = 1
    S = 6
    sim, st, g, N = _scaled_case(S, 3, B=2, conf_cls=DetScaled)
    assert sim.deterministic == 1
    a = run_hip(sim, st, g=g, clip=True)
    b = run_hip(sim, st)
    conf = ScaledConf()
    chk = mpm_det_forward(st, N, n_grid=conf.n_grid, res=conf.res, steps=S, dt=conf.dt)
    for key in ("x", "v", "C", "F"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key + ": two runs differ")
        np.testing.assert_array_equal(a[key], chk[key], err_msg=key + ": GPU != CPU build of the same source")
    # the backward of a deterministic handle (position control) is deterministic too: ordered sums wherever the default kernels take atomics --
    # two calls on the same inputs return the same bits, and the gradients are the default backward's to its tolerance
    a2 = run_hip(sim, st, g=g, clip=True)
    sim0, _, _, _ = _scaled_case(S, 3, B=2)
    ref = run_hip(sim0, st, g=g, clip=True)
    for key in ("gx", "gv", "gC", "gF", "gppos", "gfriction", "gmu", "glamda", "gaction"):
        np.testing.assert_array_equal(a[key], a2[key], err_msg=key + ": two backward runs differ")
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(a[key]).all() and _rel(a[key], ref[key]) < 2e-4, (key, _rel(a[key], ref[key]))
    for key in ("gfriction", "gmu", "glamda"):
        assert _rel(a[key].reshape(-1), ref[key].reshape(-1)) < 1e-3, (key, a[key], ref[key])


@pytest.mark.gpu
@pytest.mark.parametrize("clip", [True, False])
def test_deterministic_backward_is_reproducible_and_matches_the_oracle(demo, clip):
    """whip_rope's recorded states, 4 envs x 70 substeps, deterministic handle: forward + backward twice -> every gradient bit-identical;
    against the f64 oracle adjoint at the tolerance of the default kernels' test (test_bwd_matches_oracle)."""
    from oracle.pyoracle import MpmOracle
    from test_mpm_gpu import run_hip
    from test_oracle_mpm import _adjoint_case
    S = 70
    cases = [_adjoint_case(demo, S, k, 1, k, np.float32) for k in (0, 20, 40, 60)]
    st = {key: np.concatenate([c[0][key] for c in cases]) for key in cases[0][0]}
    g = {key: np.concatenate([c[1][key] for c in cases]) for key in cases[0][1]}
    sim = _det_sim(S, 4, 1)
    a, b = run_hip(sim, st, g=g, clip=clip), run_hip(sim, st, g=g, clip=clip)
    keys = ("gx", "gv", "gC", "gF", "gppos", "gfriction", "gmu", "glamda", "gaction")
    for key in keys:
        np.testing.assert_array_equal(a[key], b[key], err_msg=key + ": two backward runs differ")
    st64, g64 = {k: v.astype(np.float64) for k, v in st.items()}, {k: v.astype(np.float64) for k, v in g.items()}
    ob = MpmOracle(67, steps=S).step_bwd(st64, g64, clip=clip)
    # f32 over 70 reverse substeps against f64: the default kernels' own test takes 1e-2 at 50 substeps (test_bwd_matches_oracle), this one 2e-2 at 70;
    # gC is the smallest cotangent by orders of magnitude and carries the most round-off
    for key in ("gx", "gv", "gC", "gF", "gppos", "gaction"):
        assert np.isfinite(a[key]).all() and _rel(a[key], ob[key]) < (3e-2 if key == "gC" else 2e-2), (key, _rel(a[key], ob[key]))
    fast = run_hip(_det_sim(S, 4, 0), st, g=g, clip=clip)                   # and the default kernels' gradients, much closer (same arithmetic)
    for key in ("gx", "gv", "gF", "gppos", "gaction"):
        assert _rel(a[key], fast[key]) < 2e-3, (key, _rel(a[key], fast[key]))


def _soft_cases(demo):
    """soft contact (collide_batch): a rotated box sitting in the rope (elastic and plastic rope, 20 substeps) and two upright bowls
    (container SDF) around a liquid body -- no primitive turns during the step, so no library sin / cos enters the poses"""
    from test_oracle_mpm import _collide_case, _two_bowl_case
    out = []
    for material in (1, 2):
        st, _ = _collide_case(demo, 20, 40, material, 0, np.float32, w=(0.0, 0.0, 0.0))
        out.append((f"box_material{material}", st, dict(steps=20, material=np.full(67, material), n_prim=1, sdf_kind=0)))
    st, _ = _two_bowl_case(demo, 3, 40, 0, np.float32, turning=False)
    out.append(("two_bowls_liquid", st, dict(steps=3, material=np.zeros(67, np.int64), n_prim=2, sdf_kind=1)))
    return out


def _soft_checker(st, kw):
    from oracle.pyoracle import mpm_det_forward
    P = kw["n_prim"]
    return mpm_det_forward(st, 67, steps=kw["steps"], material=kw["material"], position_control=False, n_prim=P, sdf_kind=kw["sdf_kind"],
                           prim_friction=[0.1] * P, prim_softness=[666.0] * P)


def test_same_order_source_follows_the_reference_restatement_in_soft_contact(demo):
    """collide_batch in the deterministic mode's source (mpm_collide.h compiled for the host, its exp the plain-IEEE ud_expf) against the
    independent oracle: the usual forward tolerances of the soft-contact tests."""
    from oracle.pyoracle import MpmOracle
    for name, st, kw in _soft_cases(demo):
        ref = MpmOracle(67, steps=kw["steps"], material=kw["material"], position_control=False, n_prim=kw["n_prim"],
                        sdf="container" if kw["sdf_kind"] else "box").step_fwd(st)
        got = _soft_checker(st, kw)
        assert np.isfinite(got["x"]).all(), name
        assert _rel(got["x"], ref["x"]) < 5e-6 and _rel(got["v"], ref["v"]) < 1e-4, (name, _rel(got["x"], ref["x"]), _rel(got["v"], ref["v"]))
        assert _rel(got["C"], ref["C"]) < 1e-3 and _rel(got["F"], ref["F"]) < 5e-5, (name, _rel(got["C"], ref["C"]), _rel(got["F"], ref["F"]))


@pytest.mark.gpu
def test_deterministic_soft_contact_is_bit_identical_to_the_same_order_checker(demo):
    """ud_mpm_conf.deterministic with use_position_control = 0 (round 4): box and container primitives, elastic / plastic / liquid bodies.
    Two GPU runs identical; x, v, C, F bit for bit the CPU build of the same source; the default kernels' physics to their tolerance."""
    import torch
    from test_mpm_gpu import LegacyConf
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator, _Step

    def run(st, kw, deterministic):
        conf = LegacyConf()
        conf.steps = kw["steps"]
        conf.deterministic = deterministic
        sim = SimpleMPMSimulator(conf, 1, use_position_control=False)
        sim.n_particles, sim.material, sim.h = 67, np.asarray(kw["material"], np.int32), np.ones(67, np.float32)
        if kw["n_prim"] > 1:
            sim.n_primitive, sim.sdf_kind = kw["n_prim"], "container"
        sim._make_handle()
        t = lambda a: torch.tensor(np.asarray(a, np.float32), device=sim.device)
        outs = []
        for _ in range(2):
            with torch.no_grad():
                out = _Step.apply(sim, *[t(st[k]) for k in ("x", "v", "C", "F", "J", "ppos", "prot", "psize", "friction", "mu", "lamda", "action")])
            sim.check_status()
            outs.append({k: o.cpu().numpy() for k, o in zip(("x", "v", "C", "F"), out)})
        return outs

    for name, st, kw in _soft_cases(demo):
        a, b = run(st, kw, 1)
        chk = _soft_checker(st, kw)
        for key in ("x", "v", "C", "F"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{name} {key}: two runs differ")
            np.testing.assert_array_equal(a[key], chk[key], err_msg=f"{name} {key}: GPU != CPU build of the same source")
        fast = run(st, kw, 0)[0]
        assert _rel(a["x"], fast["x"]) < 5e-6 and _rel(a["v"], fast["v"]) < 2e-4 and _rel(a["F"], fast["F"]) < 5e-5, \
            (name, _rel(a["x"], fast["x"]), _rel(a["v"], fast["v"]), _rel(a["F"], fast["F"]))


@pytest.mark.gpu
@pytest.mark.parametrize("clip", [True, False])
def test_deterministic_soft_contact_backward_is_reproducible(demo, clip):
    """The backward of a deterministic handle under soft contact: the collide adjoint's per-primitive cotangents (position, rotation, size,
    friction) take the same route as the position-control ones -- one row per listed cell, the list sorted, added up in a fixed order.
    Two forward + backward calls: every gradient bit-identical; the default kernels' gradients to their tolerance."""
    from test_mpm_gpu import LegacyConf, run_hip_collide
    from test_oracle_mpm import _collide_case, _two_bowl_case
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator

    def make(kw, deterministic):
        conf = LegacyConf()
        conf.steps = kw["steps"]
        conf.deterministic = deterministic
        sim = SimpleMPMSimulator(conf, 1, use_position_control=False)
        sim.n_particles, sim.material, sim.h = 67, np.asarray(kw["material"], np.int32), np.ones(67, np.float32)
        if kw["n_prim"] > 1:
            sim.n_primitive, sim.sdf_kind = kw["n_prim"], "container"
        sim._make_handle()
        return sim

    cases = []
    for material in (1, 2):
        st, g = _collide_case(demo, 20, 40, material, 0, np.float32, w=(0.0, 0.0, 0.0))
        cases.append((f"box_material{material}", st, g, dict(steps=20, material=np.full(67, material), n_prim=1)))
    st, g = _two_bowl_case(demo, 3, 40, 0, np.float32, turning=False)
    cases.append(("two_bowls_liquid", st, g, dict(steps=3, material=np.zeros(67, np.int64), n_prim=2)))
    keys = ("x", "v", "C", "F", "gx", "gv", "gC", "gF", "gppos", "gprot", "gfriction", "gmu", "glamda", "gaction")
    for name, st, g, kw in cases:
        sim = make(kw, 1)
        a, b = run_hip_collide(sim, st, g, clip), run_hip_collide(sim, st, g, clip)
        for key in keys:
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{name} {key}: two runs differ")
        fast = run_hip_collide(make(kw, 0), st, g, clip)
        for key in ("gx", "gv", "gF", "gppos"):
            fin = np.isfinite(fast[key])
            assert (np.isfinite(a[key]) == fin).all(), (name, key)
            assert _rel(np.where(fin, a[key], 0), np.where(fin, fast[key], 0)) < 5e-3, (name, key, _rel(np.where(fin, a[key], 0), np.where(fin, fast[key], 0)))
