"""GPU: the reference-shaped env / APG surface over the kernels (ClothEnv, MPMEnv, env_functions, APG update)."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_registry_names():
    from unidom_amd.envs.registration import env_functions
    assert {"fold_cloth1", "fold_cloth1_para", "fold_cloth3", "unfold_cloth1", "unfold_cloth3", "fold_tshirt", "whip_rope", "shape_rope",
            "push_rope", "shape_rope_hard", "push_rope_hard", "pour_water", "pour_soup"} == set(env_functions)   # registration.py:13-27, every key
    assert env_functions["push_rope"] is env_functions["shape_rope"] and env_functions["push_rope_hard"] is env_functions["shape_rope_hard"]


def test_fold_cloth1_step_diff_reproduces_recorded_primitives_and_keys():
    """Reference data through the whole env path (get_pnp_actions -> 40 robot_steps -> state): the recorded
    primitive0/1, cur_step and PRNG key after each step_diff are reproduced exactly (cloth x/v are not pinned by
    these demos -- legacy cloth step)."""
    from unidom_amd.envs.registration import env_functions
    d = np.load(os.path.join(GOLDEN, "fold_cloth1_demos.npz"))
    n = len(d["demo"])
    env = env_functions["fold_cloth1"](batch_size=n, aux_reward=True)
    _, st = env.reset(np.array([0, 1], np.uint32))
    dev = env.device
    t = lambda a: torch.tensor(a, device=dev)
    st = st._replace(x=t(d["s0_x"]), v=t(d["s0_v"]), primitive0=t(d["s0_primitive0"]), primitive1=t(d["s0_primitive1"]),
                     key=d["s0_key"], cur_step=t(d["s0_cur_step"]), mu=t(d["s0_mu"]))
    obs, reward, done, info = env.step_diff(t(d["action"]), st)
    s1 = info["state"]
    np.testing.assert_array_equal(s1.primitive0.cpu().numpy(), d["s1_primitive0"])
    np.testing.assert_array_equal(s1.primitive1.cpu().numpy(), d["s1_primitive1"])
    np.testing.assert_array_equal(s1.key, d["s1_key"])
    np.testing.assert_array_equal(s1.cur_step.cpu().numpy(), d["s1_cur_step"])
    assert obs.shape == (n, 1544) and info["obs_list"].shape == (40, n, 1544)
    assert torch.isfinite(reward).all() and reward.shape == (n,)
    assert info["state_list"].x.shape == (40, n, 512, 3)


@pytest.mark.parametrize("task,count", [("fold_cloth3", 30), ("unfold_cloth1", 55), ("unfold_cloth3", 102), ("fold_tshirt", 39)])
def test_sibling_cloth_envs_step_diff_reproduce_recorded_primitives_and_keys(task, count):
    """The same pin for the reference's other cloth recordings (expert_demo/{fold_cloth3, unfold_cloth1, unfold_cloth3,
    fold_tshirt}, tests/golden/cloth_sibling_demos.npz): every recorded transition as one env of one launch of that task's own
    env (its mask, stiffness, dt, max_steps); the cloth starts from the env's reset state (the recorded cloth x/v are legacy and
    not kept -- the primitives, the key chain and cur_step do not depend on them)."""
    from unidom_amd.envs.registration import env_functions
    d = np.load(os.path.join(GOLDEN, "cloth_sibling_demos.npz"))
    sel = np.nonzero(d["task"] == task)[0]
    assert len(sel) == count
    if task == "fold_tshirt":
        sel = sel[:12]                 # 3573 particles x 2000 substeps per env: a dozen transitions are plenty here
    n = len(sel)
    env = env_functions[task](batch_size=n, aux_reward=True)
    np.random.seed(0)                  # the unfold envs fold the cloth at reset with np.random picks (unfold_cloth1_env.py:56-66)
    _, st = env.reset(np.array([0, 1], np.uint32))
    t = lambda a: torch.tensor(a, device=env.device)
    st = st._replace(primitive0=t(d["s0_primitive0"][sel]), primitive1=t(d["s0_primitive1"][sel]), key=d["s0_key"][sel],
                     cur_step=t(d["s0_cur_step"][sel]))
    _, reward, _, info = env.step_diff(t(d["action"][sel]), st, want_lists=False)
    s1 = info["state"]
    np.testing.assert_array_equal(s1.primitive0.cpu().numpy(), d["s1_primitive0"][sel])
    np.testing.assert_array_equal(s1.primitive1.cpu().numpy(), d["s1_primitive1"][sel])
    np.testing.assert_array_equal(s1.key, d["s1_key"][sel])
    np.testing.assert_array_equal(s1.cur_step.cpu().numpy(), d["s1_cur_step"][sel])
    assert torch.isfinite(reward).all()


def test_fold_cloth1_para_obs_and_grad():
    from unidom_amd.envs.registration import env_functions
    env = env_functions["fold_cloth1_para"](batch_size=2, aux_reward=True, stiffness=1200, eval_min_max_stiff=[10, 1800])
    obs, st = env.reset(np.array([0, 7], np.uint32))
    assert obs.shape == (2, 1545)
    np.testing.assert_allclose(obs[:, -1].cpu().numpy(), (1200 - 10) / (1800 - 10), rtol=1e-6)
    a = torch.full((2, 6), 0.5, device=env.device, requires_grad=True)
    st = st._replace(stiffness=st.stiffness.to(torch.float32).requires_grad_(True))
    _, reward, _, info = env.step_diff(a, st, want_lists=False)
    reward.sum().backward()
    assert torch.isfinite(a.grad).all() and a.grad.abs().sum() > 0
    assert torch.isfinite(st.stiffness.grad).all()


def test_whip_rope_reset_and_step():
    from unidom_amd.envs.registration import env_functions
    d = np.load(os.path.join(GOLDEN, "whip_rope_demo0.npz"))
    env = env_functions["whip_rope"](batch_size=3, seed=1)
    obs, st = env.reset(np.array([0, 5], np.uint32))
    assert obs.shape == (3, 612) and st.x.shape == (3, 67, 3)
    # lattice seeding (add_box, mpm_simulator.py:94-109) reproduces the recorded rope up to the reset shift
    x0 = st.x[0].cpu().numpy()
    np.testing.assert_allclose(x0 - x0[0], d["x"][0] - d["x"][0][0], atol=2e-7)
    np.testing.assert_allclose(x0[:, 1], d["x"][0][:, 1], atol=1e-7)
    a = torch.tensor([[0.5, 0.0, -0.3, 0, 0, 0]] * 3, device=env.device, requires_grad=True)
    obs2, reward, done, info = env.step_diff(a, st)
    env.simulator.check_status()
    assert obs2.shape == (3, 612) and info["obs_list"].shape == (1, 3, 612)
    assert torch.isfinite(reward).all() and not bool(done.any())
    # primitive advanced steps-1 increments of a/50/steps (Q5)
    p0 = st.primitives[0].position[:, 0].cpu().numpy()
    p1 = info["state"].primitives[0].position[:, 0].detach().cpu().numpy()
    np.testing.assert_allclose(p1 - p0, np.float32((np.array([0.5, 0, -0.3]) + 1e-12) / 50 / 70 * 69)[None].repeat(3, 0),
                               atol=3e-6)   # 69 f32 increments onto a position of ~0.5
    reward.sum().backward()
    assert torch.isfinite(a.grad).all() and a.grad[:, :3].abs().sum() > 0 and a.grad[:, 3:].abs().sum() == 0


@pytest.mark.parametrize("name,num_envs,ep_len", [("fold_cloth1", 2, 2), ("whip_rope", 4, 2), ("pour_water", 3, 2)])
def test_apg_update_runs(name, num_envs, ep_len):
    from unidom_amd.algorithms.apg.core import APG
    from unidom_amd.envs.registration import env_functions
    env = env_functions[name](batch_size=num_envs, seed=0, aux_reward=True)
    _, st = env.reset(np.array([0, 3], np.uint32))
    learner = APG(env, ep_len, learning_rate=1e-4, max_gradient_norm=0.3, seed=0)
    w0 = learner.policy.layers[0].weight.detach().clone()
    m = learner.minimize(st)
    assert torch.isfinite(m["grad_norm"]) and float(m["grad_norm"]) > 0
    assert not torch.equal(w0, learner.policy.layers[0].weight)
    assert m["reward"].shape == (ep_len, num_envs)


def test_apg_update_captured_as_one_hip_graph_is_the_eager_update():
    """APG.capture replays one update (policy, 2 x step_diff, loss, backward, clip, Adam) as ONE HIP graph.  Same key schedule and
    arithmetic as minimize(): after three updates from the same seed the losses agree and the parameters have moved alike (Adam's
    first steps are ~lr * sign(g): an element whose gradient is at the float atomics' noise level may land on the other side, so
    the comparison allows a per-mille of the elements to differ by more than a tenth of one step).  The learner lives on one
    non-default stream, eager updates included (capture()'s docstring says why); capture() leaves parameters and moments untouched."""
    from unidom_amd.algorithms.apg.core import APG
    from unidom_amd.envs.registration import env_functions
    dev = torch.device("cuda", 0)
    work = torch.cuda.Stream(dev)
    lr = 1e-4
    with torch.cuda.stream(work):
        out = {}
        for mode in ("eager", "graph"):
            env = env_functions["whip_rope"](batch_size=4, seed=0, aux_reward=True)
            _, st = env.reset(np.array([0, 3], np.uint32))
            learner = APG(env, 2, learning_rate=lr, max_gradient_norm=0.3, seed=0)
            w0 = [p.detach().clone() for p in learner.params]
            if mode == "graph":
                learner.capture(st)
                assert all(torch.equal(a, b) for a, b in zip(w0, learner.params))
            ms = [(learner.minimize_captured() if mode == "graph" else learner.minimize(st)) for _ in range(3)]
            torch.cuda.synchronize(dev)
            env.simulator.check_status()
            out[mode] = (float(ms[-1]["loss"]), float(ms[-1]["grad_norm"]), torch.cat([(p.detach() - a).reshape(-1) for p, a in zip(learner.params, w0)]))
    (le, ge, de), (lg, gg, dg) = out["eager"], out["graph"]
    assert math.isfinite(le) and abs(le - lg) < 1e-5 * max(1.0, abs(le)) and abs(ge - gg) < 1e-3 * ge, (le, lg, ge, gg)
    assert float(de.abs().max()) > lr                                    # the parameters did move
    assert float(((de - dg).abs() > 0.1 * lr).float().mean()) < 1e-3, float(((de - dg).abs() > 0.1 * lr).float().mean())
    with pytest.raises(RuntimeError):                                    # on the default stream capture refuses (instead of crashing the runtime)
        learner.capture(st)


def test_apg_entry_point_cli(tmp_path):
    """`python -m unidom_amd.algorithms.apg.apg_no_para` with the reference's flags (train_no_para.sh), 2 iterations."""
    import json
    from unidom_amd.algorithms.apg import apg_no_para
    logdir = str(tmp_path / "log")
    apg_no_para.main(["--env", "fold_cloth1", "--ep_len", "1", "--num_envs", "2", "--lr", "1e-4", "--gpus", "1",
                      "--max_grad_norm", "0.3", "--seed", "0", "--eval_freq", "100", "--max_it", "1",
                      "--train_min_stiff", "1000", "--train_max_stiff", "1600", "--eval_min_stiff", "10",
                      "--eval_max_stiff", "1800", "--logdir", logdir])
    recs = [json.loads(l) for l in open(os.path.join(logdir, "log.jsonl"))]
    assert [r["iter"] for r in recs] == [0, 1]
    np.random.seed(0)
    assert abs(recs[0]["core_env_stiffness"] - np.random.uniform(1000, 1600)) < 1e-9    # apg_para.py:326-329
    assert all(np.isfinite(r["train_reward"]) and np.isfinite(r["grad_norm"]) for r in recs)
    assert os.path.exists(os.path.join(logdir, "apg_fold_cloth1_0.pt"))


# ---- fused env arithmetic (csrc/env_glue.hip) against the op-by-op torch restatement ---------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("P,Q,B", [(512, 512, 4), (333, 700, 3), (64, 17, 1)])
def test_fused_chamfer_matches_op_by_op(P, Q, B):
    from unidom_amd.envs.basic import _fused
    from unidom_amd.utils.util import calc_chamfer
    g = torch.Generator(device="cuda").manual_seed(P + Q)
    x = torch.rand((B, P, 3), device="cuda", generator=g).requires_grad_(True)
    y = torch.rand((Q, 3), device="cuda", generator=g)
    w = torch.rand((B,), device="cuda", generator=g) + 0.5
    ref = calc_chamfer(x, y)
    (gref,) = torch.autograd.grad((ref * w).sum(), x)
    out = _fused.chamfer(x, y)
    (gout,) = torch.autograd.grad((out * w).sum(), x)
    # f32 on both sides; differences = summation order of the two means and of the argmin scatter
    torch.testing.assert_close(out, ref, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(gout, gref, rtol=2e-5, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("bad", ["nan", "inf", "all_nan", "all_inf"])
def test_fused_chamfer_and_contact_with_non_finite_particles(bad):
    """One NaN / inf particle (or a whole non-finite cloud) must not turn into an out-of-range argmin: the index the
    forward stores is always a valid one, and the non-finite value flows through min / sqrt as in jnp / torch
    (min propagates NaN; the cotangent goes to the first NaN / the first minimum)."""
    from unidom_amd.envs.basic import _fused
    from unidom_amd.utils.util import calc_chamfer
    B, P, Q = 2, 300, 257
    g = torch.Generator(device="cuda").manual_seed(7)
    x0 = torch.rand((B, P, 3), device="cuda", generator=g)
    y = torch.rand((Q, 3), device="cuda", generator=g)
    val = float("nan") if "nan" in bad else float("inf")
    if bad.startswith("all"):
        x0[0] = val
    else:
        x0[0, 17, 1] = val
        x0[1, 211] = val
    x = x0.clone().requires_grad_(True)
    ref = calc_chamfer(x, y)
    (gref,) = torch.autograd.grad(ref.sum(), x)
    xf = x0.clone().requires_grad_(True)
    out = _fused.chamfer(xf, y)
    (gout,) = torch.autograd.grad(out.sum(), xf)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=2e-6, atol=1e-7, equal_nan=True)
    # finite entries agree; the non-finite ones sit in the same rows (inf * 0 vs NaN may differ in kind, not in place)
    assert torch.equal(torch.isfinite(gout), torch.isfinite(gref))
    fin = torch.isfinite(gref)
    torch.testing.assert_close(gout[fin], gref[fin], rtol=2e-5, atol=1e-8)
    # pick-and-place contact distance with the same clouds
    prim0 = torch.rand((B, 4), device="cuda", generator=g)
    a = torch.rand((B, 6), device="cuda", generator=g).requires_grad_(True)
    xc = x0.clone().requires_grad_(True)
    macro, contact = _fused.pnp_and_contact(a, prim0, xc)
    (contact.sum() + macro.sum()).backward()
    torch.cuda.synchronize()
    cref = torch.linalg.norm(a.detach()[:, None, :3] - x0, dim=-1).min(-1).values
    torch.testing.assert_close(contact.detach(), cref, rtol=2e-6, atol=1e-7, equal_nan=True)
    assert xc.grad.shape == x0.shape


@pytest.mark.gpu
@pytest.mark.parametrize("aux", [True, False])
def test_fused_pnp_and_contact_match_op_by_op(aux):
    from unidom_amd.envs.basic import _fused
    from unidom_amd.envs.basic.cloth_env import ClothEnv
    from unidom_amd.engine.cloth_simulator import ClothState
    B, P = 4, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    actions = torch.rand((B, 6), device="cuda", generator=g).requires_grad_(True)
    prim0 = torch.rand((B, 4), device="cuda", generator=g).requires_grad_(True)
    x = torch.rand((B, P, 3), device="cuda", generator=g).requires_grad_(True)
    st = ClothState(x=x, v=None, primitive0=prim0, primitive1=None, action0=None, action1=None, key=None, cur_step=None,
                    stiffness=None, mu=None)
    macro_ref = ClothEnv.get_pnp_actions(actions, st)
    contact_ref = torch.sqrt(((actions[..., :3][:, None, :] - x) ** 2).sum(-1)).min(-1).values
    macro, contact = _fused.pnp_and_contact(actions, prim0, x)
    # the kernel divides (as jnp does); torch's tensor / scalar multiplies by the rounded reciprocal: 1 ulp apart
    torch.testing.assert_close(macro, macro_ref, rtol=3e-7, atol=0)
    torch.testing.assert_close(contact, contact_ref, rtol=1e-6, atol=0)
    wm = torch.rand(macro.shape, device="cuda", generator=g)
    wc = torch.rand((B,), device="cuda", generator=g) if aux else torch.zeros((B,), device="cuda")
    loss = lambda m, c: (m * wm).sum() + ((c * wc).sum() if aux else 0.0)
    gref = torch.autograd.grad(loss(macro_ref, contact_ref), (actions, prim0, x), allow_unused=True)
    gout = torch.autograd.grad(loss(macro, contact), (actions, prim0, x), allow_unused=True)
    for a, r in zip(gout, gref):
        r = torch.zeros_like(a) if r is None else r
        torch.testing.assert_close(a, r, rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
def test_step_diff_real_reward_is_lazy_and_right():
    from unidom_amd.envs.registration import env_functions
    from unidom_amd.utils.util import calc_chamfer
    from unidom_amd.utils import prng
    env = env_functions["fold_cloth1"](batch_size=2, seed=0, aux_reward=True)
    _, state = env.reset(prng.PRNGKey(0))
    actions = torch.tensor([[0.4, 0.0, 0.45, 0.6, 0.0, 0.5], [0.5, 0.0, 0.5, 0.45, 0.0, 0.6]], device="cuda")
    _, reward, _, info = env.step_diff(actions, state)
    assert "real_reward" in info and "real_reward" not in dict.keys(info)   # not computed yet
    old = calc_chamfer(state.x, env.goal)
    new = calc_chamfer(info["state"].x, env.goal)
    contact = torch.sqrt(((actions[:, None, :3] - state.x) ** 2).sum(-1)).min(-1).values
    torch.testing.assert_close(info["real_reward"], old - new + 0.1 * contact, rtol=1e-5, atol=1e-6)   # cloth_env.py:226
    expect = (math.e ** (-new * 10) + math.e ** (-contact)) * 0.99 ** info["state"].cur_step
    torch.testing.assert_close(reward, expect, rtol=1e-5, atol=1e-6)


# ---- the sibling cloth envs: same mask, same kernels, other confs ---------------------------
def test_sibling_cloth_envs_confs():
    from unidom_amd.envs.registration import env_functions
    e3 = env_functions["fold_cloth3"](batch_size=2)
    assert (e3.max_steps, e3.conf.stiffness, e3.conf.mu, e3.conf.use_substep_obs) == (4, 900, 0.5, True)   # fold_cloth3_env.py:23-47
    u1, u3 = env_functions["unfold_cloth1"](batch_size=2), env_functions["unfold_cloth3"](batch_size=2)
    for u, n in ((u1, 1), (u3, 3)):
        assert (u.max_steps, u.conf.mu, u.conf.use_substep_obs, u.random_fold_steps) == (15, 3, False, n)   # unfold_cloth1_env.py:25-44,75
    assert u3.conf.task == "unfold_cloth3" and u3.goal.shape == (512, 3) and e3.goal.shape == (512, 3)


def test_unfold_cloth1_reset_folds_and_step_matches_oracle():
    """reset = particle jitter + one random pick-and-place through step_diff (unfold_cloth1_env.py:68-79); then one
    step_diff of the mu = 3 cloth against the CPU restatement (order v2: bit-exact positions)."""
    from oracle.pyoracle import ClothOracle
    from unidom_amd.envs.registration import env_functions
    np.random.seed(3)
    env = env_functions["unfold_cloth1"](batch_size=2, aux_reward=True)
    flat = env.simulator.reset_jax()
    obs, st = env.reset(np.array([0, 11], np.uint32))
    assert obs.shape == (2, 1544) and int(st.cur_step[0]) == 1
    assert float((st.x - flat.x).abs().max()) > 1e-3                    # the fold moved the cloth
    assert float(st.mu[0]) == 3.0
    actions = torch.tensor([[0.45, 0.0, 0.5, 0.6, 0.0, 0.55], [0.5, 0.0, 0.45, 0.4, 0.0, 0.6]], device=env.device)
    _, reward, _, info = env.step_diff(actions, st)
    from unidom_amd.envs.basic import _fused
    macro = _fused.pnp_and_contact(actions, st.primitive0, st.x)[0].cpu().numpy()   # the macro actions step_diff fed the kernel
    orc = ClothOracle(np.asarray(env.cloth_mask), order=2)
    prim = torch.stack([st.primitive0, st.primitive1], 1).cpu().numpy()
    ref = orc.rollout_fwd(st.x.cpu().numpy(), st.v.cpu().numpy(), prim, st.stiffness.float().cpu().numpy(),
                          st.mu.cpu().numpy(), macro, nthreads=2)
    np.testing.assert_array_equal(info["state"].x.cpu().numpy(), ref["x"])      # friction-heavy conf, still bit-exact
    np.testing.assert_array_equal(info["state"].v.cpu().numpy(), ref["v"])
    assert torch.isfinite(reward).all()


def test_fold_cloth1_step_diff_in_the_reference_operation_order_matches_its_oracle():
    """The env surface with `conf.kernel_mode = 3` (what bench.py's `reference_order` leg runs): one full step_diff of fold_cloth1 -- the env's own
    reset state, pick-and-place macro actions from get_pnp_actions, 40 x 50 substeps -- against the CPU restatement in the reference's literal
    operation order (cloth_simulator.py:257-337 as written): final x / v bit for bit; the reward's gradient reaches the action."""
    from oracle.pyoracle import ClothOracle
    from unidom_amd.envs.basic import _fused
    from unidom_amd.envs.fold_cloth1_env import DefaultConf
    from unidom_amd.envs.registration import env_functions
    conf = DefaultConf()
    conf.kernel_mode = 3
    env = env_functions["fold_cloth1"](batch_size=3, conf=conf, seed=0, aux_reward=True)
    assert env.simulator.mode == 3
    _, st = env.reset(np.array([0, 5], np.uint32))
    actions = torch.tensor([[0.45, 0.0, 0.5, 0.6, 0.0, 0.55], [0.5, 0.0, 0.45, 0.4, 0.0, 0.6], [0.55, 0.0, 0.6, 0.45, 0.0, 0.4]],
                           device=env.device, requires_grad=True)
    _, reward, _, info = env.step_diff(actions, st)
    reward.sum().backward()
    assert torch.isfinite(actions.grad).all() and float(actions.grad.abs().sum()) > 0
    macro = _fused.pnp_and_contact(actions.detach(), st.primitive0, st.x)[0].cpu().numpy()
    orc = ClothOracle(np.asarray(env.cloth_mask), order=1)
    prim = torch.stack([st.primitive0, st.primitive1], 1).cpu().numpy()
    ref = orc.rollout_fwd(st.x.cpu().numpy(), st.v.cpu().numpy(), prim, st.stiffness.float().cpu().numpy(), st.mu.cpu().numpy(), macro, nthreads=3)
    np.testing.assert_array_equal(info["state"].x.detach().cpu().numpy(), ref["x"])
    np.testing.assert_array_equal(info["state"].v.detach().cpu().numpy(), ref["v"])
    # ... and the default order gives ANOTHER trajectory from the same inputs (the two orders are two samples of a chaotic system: DESIGN.md 3.1)
    ref2 = ClothOracle(np.asarray(env.cloth_mask), order=2).rollout_fwd(st.x.cpu().numpy(), st.v.cpu().numpy(), prim, st.stiffness.float().cpu().numpy(),
                                                                        st.mu.cpu().numpy(), macro, nthreads=3)
    assert not np.array_equal(ref2["x"], ref["x"])


def test_shape_rope_seeding_and_push_matches_oracle():
    """shape_rope (shape_rope_env.py:153-174): 582 plastic particles reproduce the reference's goal.npy lattice; one
    env.step pushes the rope through collide_batch -- the first scanned `step`s agree with the CPU oracle driven with
    the same shifted state and sub-actions, and the reward carries a finite gradient back to the push action."""
    from oracle.pyoracle import MpmOracle
    from unidom_amd.envs.registration import env_functions
    env = env_functions["shape_rope"](batch_size=2, seed=1)
    env.build_reset_state()                                     # reset() without the random pushes
    st = env.state
    conf = env.conf
    assert st.x.shape == (2, 582, 3) and conf.steps == 133 and tuple(conf.res) == (64, 6, 64)
    goal = np.load(conf.goal_path)          # the reference's recorded goal: the same 582-point lattice, settled by < 2.1e-5
    np.testing.assert_allclose(st.x[0].cpu().numpy(), goal, atol=3e-5)
    mid = st.x[0, 291].cpu().numpy()
    acts = np.array([[mid[0] - 0.004, 0, mid[2] - 0.02, mid[0] + 0.01, 0, mid[2] + 0.08],      # 5 mm from the rope, pushing across it
                     [mid[0] + 0.06, 0, mid[2] + 0.022, mid[0] + 0.04, 0, mid[2] - 0.08]], np.float32)
    a = torch.tensor(acts, device=env.device, requires_grad=True)
    obs, reward, done, info = env.step_diff(a, st)
    env.simulator.check_status()
    assert obs.shape == (2, env.observation_size) and info["obs_list"].shape == (30, 2, env.observation_size)
    assert torch.isfinite(reward).all() and int(info["state"].cur_step[0]) == 1
    # the pusher moved (end - start) * 29/30 ... in 30 steps of 132/133 increments (Q5), y pinned to 0.01
    # ---- oracle, first K scanned steps --------------------------------------------------------------------------
    K = 3
    x0 = st.x.cpu().numpy()
    shift = (np.array(conf.res, np.float32) * np.float32(0.5) / np.float32(conf.n_grid) - x0.mean(1, dtype=np.float32)).astype(np.float32)
    shift[:, 1] = 0
    start = acts[:, :3] + shift
    end = acts[:, 3:] + shift
    start[:, 1] = 0.01
    end[:, 1] = 0.01
    nrm = np.linalg.norm(end - start, axis=-1, keepdims=True).astype(np.float32) + np.float32(1e-8)
    end = start + (end - start) / nrm * np.clip(nrm, 0, 0.3)
    push = ((end - start) / np.float32(30)).astype(np.float32)
    push[:, 1] = 0
    S = conf.steps
    ppos = np.zeros((2, S, 3), np.float32)
    ppos[:, 0] = start
    prot = np.zeros((2, S, 4), np.float32)
    prot[..., 0] = 1
    mu0, la0 = conf.E / (2 * (1 + conf.nu)), conf.E * conf.nu / ((1 + conf.nu) * (1 - 2 * conf.nu))
    ost = dict(x=x0 + shift[:, None], v=np.zeros_like(x0), C=np.zeros((2, 582, 3, 3), np.float32),
               F=np.tile(np.eye(3, dtype=np.float32), (2, 582, 1, 1)), J=np.ones((2, 582), np.float32), ppos=ppos, prot=prot,
               psize=np.tile(np.array([0.015, 0.06, 0.015], np.float32), (2, 1)), friction=np.full(2, 0.9, np.float32),
               mu=np.full(2, mu0, np.float32), lamda=np.full(2, la0, np.float32),
               action=np.concatenate([push, np.zeros_like(push)], -1))
    orc = MpmOracle(582, n_grid=128, res=(64, 6, 64), steps=S, dt=conf.dt, position_control=False, material=np.full(582, 2),
                    prim_friction=0.1, prim_softness=666.0)
    sl = info["state_list"]
    npy = lambda t: t.detach().cpu().numpy()
    for k in range(K):
        o = orc.step_fwd(ost, nthreads=4)
        xk = npy(sl.x[k]) + shift[:, None]                              # post_step removed the focus shift
        err_x = np.abs(xk - o["x"]).max()
        err_v = np.abs(npy(sl.v[k]) - o["v"]).max() / (np.abs(o["v"]).max() + 1e-30)
        # 133 substeps of plastic contact in f32: the oracle's own f32 and f64 runs differ by 4e-7 / 1.3e-4 here
        assert err_x < 2e-6 and err_v < 1e-3, (k, err_x, err_v)
        np.testing.assert_allclose(npy(sl.primitives[0].position[k, :, 0]) + shift, o["ppos"][:, 0], atol=2e-7)
        # next step from the kernel's own state (each comparison = one `step` from identical inputs)
        ost.update(x=xk, v=npy(sl.v[k]), C=npy(sl.C[k]), F=npy(sl.F[k]), J=npy(sl.J[k]),
                   ppos=npy(sl.primitives[0].position[k]) + shift[:, None], prot=npy(sl.primitives[0].rotation[k]))
    assert np.abs(o["v"]).max() > 1e-3                                # the pusher reached the rope
    reward.sum().backward()
    assert torch.isfinite(a.grad).all() and a.grad.abs().sum() > 0


def test_shape_rope_hard_reset_runs():
    from unidom_amd.envs.registration import env_functions
    env = env_functions["shape_rope_hard"](batch_size=2, seed=0)
    obs, st = env.reset(None)
    env.simulator.check_status()
    assert obs.shape == (2, env.observation_size) and torch.isfinite(obs).all()
    assert int(st.cur_step[0]) == 10 and env.max_steps == 20        # 2 + 8 random pushes advance cur_step (shape_rope_env.py:123-130)
    assert (st.x[0] - st.x[1]).abs().max() > 1e-4                     # each env got its own random pushes


def test_pour_water_reset_step_matches_oracle_and_grad():
    """pour_water (pour_water_env.py:114-133): 702 liquid particles seeded like the reference's goal.npy count, two bowls
    with the container SDF.  One env.step (23 substeps, both bowls colliding in turn) agrees with the CPU oracle driven
    with the same shifted state; the reward's gradient reaches the translation and the tilt of the first bowl."""
    from oracle.pyoracle import MpmOracle
    from unidom_amd.envs.registration import env_functions
    env = env_functions["pour_water"](batch_size=2, seed=1)
    obs, st = env.reset(np.array([0, 7], np.uint32))
    conf = env.conf
    N = st.x.shape[1]
    assert N == 702 == np.load(conf.goal_path).shape[0] and conf.steps == 23 and obs.shape == (2, 4281) == (2, env.observation_size)
    assert len(st.primitives) == 2 and env.simulator.n_primitive == 2 and env.simulator.sdf_kind == "container"
    p0 = st.primitives[0].position[:, 0].cpu().numpy()
    assert np.abs(p0[:, [0, 2]] - 0.5).max() < 0.1 and np.abs(p0[0] - p0[1]).max() > 0 and np.all(p0[:, 1] == np.float32(0.2))
    a = torch.tensor([[0.8, 0.3, -0.5, 0.6, -0.2, 0.3], [-0.4, 0.0, 0.9, -0.5, 0.4, 0.1]], device=env.device, requires_grad=True)
    obs2, reward, done, info = env.step_diff(a, st)
    env.simulator.check_status()
    assert obs2.shape == (2, 4281) and info["obs_list"].shape == (1, 2, 4281) and torch.isfinite(reward).all()
    # ---- oracle ---------------------------------------------------------------------------------------------------------
    S = conf.steps
    npy = lambda t: t.detach().cpu().numpy()
    x0 = npy(st.x)
    shift = (np.array(conf.res, np.float32) * np.float32(0.5) / np.float32(conf.n_grid) - x0.mean(1, dtype=np.float32)).astype(np.float32)
    shift[:, 1] = 0
    act = np.concatenate([npy(a), np.zeros((2, 6), np.float32)], -1)
    act[:, :6] = act[:, :6] / np.float32(500.0)
    act = act + np.float32(1e-12)
    act[:, 1] = 0
    ppos = np.stack([npy(p.position) for p in st.primitives], 1) + shift[:, None, None]
    prot = np.stack([npy(p.rotation) for p in st.primitives], 1)
    psize = np.stack([npy(p.size) for p in st.primitives], 1)
    ost = dict(x=x0 + shift[:, None], v=npy(st.v), C=npy(st.C), F=npy(st.F), J=npy(st.J), ppos=ppos, prot=prot, psize=psize,
               friction=npy(st.friction).reshape(2), mu=npy(st.mu).reshape(2), lamda=npy(st.lamda).reshape(2), action=act)
    orc = MpmOracle(N, n_grid=conf.n_grid, res=conf.res, steps=S, dt=conf.dt, position_control=False, material=np.zeros(N),
                    n_prim=2, sdf="container")
    o32 = orc.step_fwd(ost, nthreads=2)
    o64 = orc.step_fwd({k: v.astype(np.float64) for k, v in ost.items()}, nthreads=2)
    s1 = info["state"]
    xk = npy(s1.x) + shift[:, None]
    gx, gv = np.abs(o32["x"] - o64["x"]).max(), np.abs(o32["v"] - o64["v"]).max() / np.abs(o64["v"]).max()
    ex, ev = np.abs(xk - o64["x"]).max(), np.abs(npy(s1.v) - o64["v"]).max() / np.abs(o64["v"]).max()
    # tilting bowls: the f32 finite-difference normal is noisy (see test_two_container_primitives_match_oracle); the kernel
    # has to be as close to the f64 restatement as the f32 restatement is
    assert ex < 3 * gx + 2e-6 and ev < 3 * gv + 1e-4, (ex, gx, ev, gv)
    for i in range(2):
        np.testing.assert_allclose(npy(s1.primitives[i].position[:, 0]) + shift, o32["ppos"][:, i, 0], atol=2e-7)
        np.testing.assert_allclose(npy(s1.primitives[i].rotation[:, 0]), o32["prot"][:, i, 0], atol=2e-7)
    assert np.abs(npy(s1.primitives[0].rotation[:, 0, 1:])).max() > 1e-5 and np.abs(npy(s1.primitives[1].rotation[:, 0, 1:])).max() < 1e-10   # bowl 1: action = 0 + 1e-12
    reward.sum().backward()
    assert torch.isfinite(a.grad).all() and a.grad[:, [0, 2]].abs().min() > 0 and a.grad[:, 3:].abs().sum() > 0
    assert (a.grad[:, 1] == 0).all()                                    # vertical motion is overwritten (pour_water_env.py:88)


def test_fold_tshirt_step_matches_oracle_and_grad():
    """fold_tshirt (fold_cloth_tshirt_env.py): 3573 particles from the recovered T-shirt mask, observation = every tenth
    particle + grippers (1082).  One step_diff (2000 substeps at k = 5000, dt = 0.5e-3) is bit-exact against the CPU
    restatement in the operation order of the default forward ("v2"); the reward's gradient reaches the pick-and-place action."""
    from oracle.pyoracle import ClothOracle
    from unidom_amd.envs.basic import _fused
    from unidom_amd.envs.registration import env_functions
    env = env_functions["fold_tshirt"](batch_size=2, aux_reward=True)
    obs, st = env.reset(np.array([0, 5], np.uint32))
    conf = env.conf
    assert st.x.shape == (2, 3573, 3) and obs.shape == (2, 1082) == (2, env.observation_size) and env.max_steps == 5
    assert tuple(env.goal.shape) == (3573, 3)          # the reference's goals/fold_tshirt/goal.npy (fold_cloth_tshirt_env.py:36-37)
    xm = st.x[0].mean(0).cpu().numpy()
    a = torch.tensor([[xm[0] - 0.08, 0.0, xm[2] + 0.05, xm[0] + 0.1, 0.0, xm[2] - 0.02],
                      [xm[0] + 0.1, 0.0, xm[2] - 0.1, xm[0] - 0.05, 0.0, xm[2] + 0.08]], device=env.device, requires_grad=True)
    obs2, reward, done, info = env.step_diff(a, st)
    assert obs2.shape == (2, 1082) and torch.isfinite(reward).all()
    macro = _fused.pnp_and_contact(a.detach(), st.primitive0, st.x)[0].cpu().numpy()
    orc = ClothOracle(np.asarray(env.cloth_mask), N=conf.N, gravity=conf.gravity, damping=conf.damping, dt=conf.dt,
                      max_v=conf.max_v, small_num=conf.small_num, order=2)     # 7 parts of 512 particles: operation order "v2"
    prim = torch.stack([st.primitive0, st.primitive1], 1).cpu().numpy()
    ref = orc.rollout_fwd(st.x.cpu().numpy(), st.v.cpu().numpy(), prim, st.stiffness.float().cpu().numpy(),
                          st.mu.cpu().numpy(), macro, nthreads=2)
    np.testing.assert_array_equal(info["state"].x.detach().cpu().numpy(), ref["x"])
    np.testing.assert_array_equal(info["state"].v.detach().cpu().numpy(), ref["v"])
    assert float((info["state"].x.detach() - st.x).abs().max()) > 1e-3        # the pick-and-place moved the shirt
    # reward on the oracle's state, restated in f64 (cloth_env.py:222-228, util.py:138-153, aux_reward): the chamfer distance
    # to the reference's goal cloud, the contact term, and the 0.99^cur_step discount
    goal = env.goal.cpu().numpy().astype(np.float64)
    cur = info["state"].cur_step.cpu().numpy()
    for b in range(2):
        d = np.sqrt(((ref["x"][b].astype(np.float64)[:, None, :] - goal[None]) ** 2).mean(-1))
        chamfer = d.min(1).mean() + d.min(0).mean()
        contact = np.linalg.norm(a.detach().cpu().numpy()[b, :3].astype(np.float64) - st.x[b].cpu().numpy().astype(np.float64), axis=-1).min()
        expect = (np.exp(-10 * chamfer) + np.exp(-contact)) * 0.99 ** cur[b]
        assert abs(float(reward[b].detach()) - expect) < 2e-5 * expect, (b, float(reward[b].detach()), expect)
    reward.sum().backward()
    assert torch.isfinite(a.grad).all() and a.grad.abs().sum() > 0


@pytest.mark.parametrize("name,steps", [("pour_water", 60), ("shape_rope", 4), ("whip_rope", 80)])
def test_long_rollouts_stay_finite_and_in_the_domain(name, steps):
    """Many env.steps in a row with random actions (through auto_reset where the episode is shorter): states stay finite, particles
    stay inside the unit box, no device-side capacity flag is left standing (the grid-checkpoint fallback absorbs pool overflows)."""
    from unidom_amd.envs.registration import env_functions
    torch.manual_seed(0)
    np.random.seed(0)
    env = env_functions[name](batch_size=3, seed=2)
    _, st = env.reset(np.array([0, 9], np.uint32))
    for k in range(steps):
        if name == "shape_rope":
            a = torch.tensor(env.random_policy(3), dtype=torch.float32, device=env.device)
            a[:, 1] = 0
        else:
            a = torch.rand((3, 6), device=env.device) * 2 - 1
        with torch.no_grad():
            obs, reward, done, info = env.step_diff(a, st)
        st = info["state"]
        env.state = st
    env.simulator.check_status()
    for t in (st.x, st.v, st.C, st.F, obs, reward):
        assert torch.isfinite(t).all()
    assert float(st.x.min()) > -1e-3 and float(st.x.max()) < 1.0 + 1e-3
    assert int(st.cur_step.max()) <= env.max_steps


def _mpm_env_and_state(name, B, **kw):
    from unidom_amd.envs.registration import env_functions
    env = env_functions[name](batch_size=B, seed=1, **kw)
    if name.startswith("shape_rope"):
        env.build_reset_state()                                 # reset() without the random pushes
        return env, env.state
    _, st = env.reset(np.array([0, 11], np.uint32))
    return env, st


def _mpm_actions(name, env, st, B):
    g = torch.Generator().manual_seed(5)
    if name.startswith("shape_rope"):
        mid = st.x[:, 291].cpu()
        a = torch.cat([mid + torch.tensor([-0.004, 0.0, -0.02]), mid + torch.tensor([0.01, 0.0, 0.08])], -1)
        a[:, [1, 4]] = 0
        return a.to(env.device)
    return (torch.rand((B, 6), generator=g) * 2 - 1).to(env.device)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["whip_rope", "pour_water", "shape_rope"])
def test_mpm_step_diff_fused_matches_op_by_op(name):
    """MPMEnv.step_diff runs the focus shift and its tail (un-shift, nan_to_num, reward, observation; mpm_env.py:99-125, :150-154,
    :90-94, :57-76) as one kernel each way; step_diff_unfused is the same arithmetic op by op in torch.  Same simulator kernels in
    between, so values agree to rounding and every gradient (actions, x, v, C, F, primitive trajectories) to the scatter-order noise
    of the simulator's adjoint -- with cotangents on every output, not only the reward."""
    B = 3
    env, st = _mpm_env_and_state(name, B)
    act = _mpm_actions(name, env, st, B)
    g = torch.Generator(device=env.device).manual_seed(3)
    w = {}

    def weight(key, like):
        if key not in w:
            w[key] = torch.randn(like.shape, device=env.device, generator=g)
        return w[key]

    def run(fn, nudge=0.0):
        a = act.clone().requires_grad_(True)
        leaves = {k: (getattr(st, k) + (nudge if k == "x" else 0.0)).clone().requires_grad_(True) for k in ("x", "v", "C", "F")}
        pos = [p.position.clone().requires_grad_(True) for p in st.primitives]
        s = st._replace(primitives=[p._replace(position=q) for p, q in zip(st.primitives, pos)], **leaves)
        obs, reward, done, info = fn(a, s)
        ns = info["state"]
        outs = {"reward": reward, "obs": obs, "x": ns.x, "v": ns.v, "C": ns.C, "F": ns.F, "J": ns.J}
        for i, p in enumerate(ns.primitives):
            outs[f"pos{i}"] = p.position
        loss = sum((t * weight(k, t)).sum() * (0.0 if k == "J" else 1.0 if k == "reward" else 1e-3) for k, t in outs.items())
        loss.backward()
        grads = {"a": a.grad, **{k: t.grad for k, t in leaves.items()}, **{f"pos{i}": q.grad for i, q in enumerate(pos)}}
        lists = {"obs_list": info["obs_list"], "x_list": info["state_list"].x, "pos_list": info["state_list"].primitives[0].position}
        return ({k: t.detach() for k, t in outs.items()}, grads, {k: t.detach() for k, t in lists.items()}, done, ns)

    out_f, grad_f, list_f, done_f, ns_f = run(env.step_diff)
    out_u, grad_u, list_u, done_u, ns_u = run(env.step_diff_unfused)
    out_r, grad_r, list_r, _, _ = run(env.step_diff_unfused, nudge=2.0 ** -24)   # the op-by-op path again, the cloud moved by one ulp
    env.simulator.check_status()
    assert torch.equal(done_f, done_u) and torch.equal(ns_f.cur_step, ns_u.cur_step)
    np.testing.assert_array_equal(np.asarray(ns_f.key), np.asarray(ns_u.key))

    bad, skipped = [], []

    def close(a, b, r, k, floor, min_scale=0.0):
        # The two paths hand the simulator a shift that differs in its last bit (reduction order of the mean); 23-4000 substeps with
        # contact and friction branches amplify that, in the adjoint most of all.  So the bar is the path's measured sensitivity to a
        # one-ulp move of the cloud (which also contains its run-to-run scatter-order noise), not a fixed epsilon: what this test pins
        # is the wiring (every output and cotangent in its place); the kernels' arithmetic is pinned by the two tests below.
        scale = max(float(b.abs().max()), min_scale) + 1e-30
        err, noise = float((a - b).abs().max()) / scale, float((r - b).abs().max()) / scale
        if noise > 0.05:          # shape_rope's adjoint over 30 x 133 plastic, contacting substeps: a one-ulp move changes it by O(1),
            skipped.append(k)     # so it carries no information about the wiring either
        elif not err <= 10 * noise + floor:
            bad.append((k, err, noise, scale, int((a - b).abs().argmax())))

    for k in out_u:
        close(out_f[k], out_u[k], out_r[k], k, 1e-3 if k in ("C", "F") else 1e-4)
    for k in list_u:
        assert list_f[k].shape == list_u[k].shape
        close(list_f[k], list_u[k], list_r[k], k, 1e-4)
    gmax = max(float(t.abs().max()) for t in grad_u.values())   # the input C's cotangent is ~1e3 x smaller than the others (g2p overwrites
    for k in grad_u:                                             # C every substep): judged on the scale of the adjoint it was carved out of
        assert grad_u[k] is not None and grad_f[k] is not None, k
        assert torch.isfinite(grad_f[k]).all() and grad_f[k].shape == grad_u[k].shape
        if name == "shape_rope":
            # 30 x 133 plastic, contacting substeps: the adjoint is chaotic with a heavy tail -- a one-ulp move of the cloud changes it
            # by 3 % in one run and by O(1) in the next (friction / clip branches flipping), so a single measured sensitivity is not
            # a bar: with the SVD cotangent in closed form it sometimes came out under the 5 % cut while fused and op-by-op (whose
            # shifts differ in the last bit) were O(1) apart.  Its gradients carry no information about the wiring: values only.
            skipped.append("grad " + k)
            continue
        close(grad_f[k], grad_u[k], grad_r[k], "grad " + k, 2e-3, min_scale=1e-3 * gmax)
    assert not bad, bad
    assert name == "shape_rope" or not ({"grad a", "grad x", "grad v", "grad pos0"} & set(skipped)), skipped


SHAPE_ROPE_WIRING_STEPS = 6   # scanned simulator.steps (133 substeps each) of the gradient wiring check below: a horizon at which the
                              # measured noise of the adjoint (one-ulp nudge, run-to-run) stays under 5 % for the action, x, v and C
                              # cotangents -- 1e-3 ... 1e-2 up to 8 steps, O(1) from 12 on (profiles/r03_shape_rope_grad_noise.txt); the
                              # onset moves with the float atomics' order from build to build: at 8 a round-4 build (same per-cell
                              # arithmetic, another schedule) measured 5 % on the action and x cotangents, so two steps of margin.
                              # The F cotangent is the exception: 0.3 ... 1 at EVERY horizon, one step included (a plastic body's F enters
                              # only through the SVD of (I + dt C) F, with 1 / (s_j^2 - s_i^2) factors on an almost isotropic F); it is
                              # compared wherever its noise allows and reported otherwise.


@pytest.mark.gpu
def test_shape_rope_step_diff_gradient_wiring_short_horizon(monkeypatch):
    """The full shape_rope env.step (30 x 133 plastic, contacting substeps) has a chaotic adjoint, so the test above compares its
    values only.  The cotangent ROUTES of the fused step_diff on that env -- shift -> actions (process_pre_step_actions adds it to
    both end points), primitive position, the push action through 2 scanned steps, x / v / C / F -- are checked here at a horizon
    where the adjoint is well conditioned (tools/shape_rope_grad_noise.py measures the noise per horizon): fused against op-by-op,
    every gradient, against the measured sensitivity to a one-ulp move of the cloud."""
    import os
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import shape_rope_grad_noise as srg
    from unidom_amd.envs import shape_rope_env as sre
    monkeypatch.setattr(sre.DefaultConf, "primitive_action_steps", sre.DefaultConf.primitive_action_steps)   # restored afterwards
    T, B = SHAPE_ROPE_WIRING_STEPS, 3
    env, st = srg.make_env(T, B)
    act = srg.push_action(st, T, env.device)
    gf, of = srg.grads(env, st, act, env.step_diff)
    gu, ou = srg.grads(env, st, act, env.step_diff_unfused)
    gr, _ = srg.grads(env, st, act, env.step_diff_unfused, nudge=2.0 ** -24)
    env.simulator.check_status()
    for k in ("x", "v", "reward"):
        assert srg.rel(of[k], ou[k]) < 1e-4, (k, srg.rel(of[k], ou[k]))
    gmax = max(float(t.abs().max()) for t in gu.values())
    bad = []
    noisy = []
    for k in gu:
        assert torch.isfinite(gf[k]).all(), k
        assert k == "pos0" or float(gu[k].abs().max()) > 0, k     # (get_primitive_actions overwrites position[0]: the start point gets it)
        scale = max(float(gu[k].abs().max()), 1e-3 * gmax)
        err, noise = float((gf[k] - gu[k]).abs().max()) / scale, float((gr[k] - gu[k]).abs().max()) / scale
        if noise >= 0.05:
            noisy.append(k)
        elif not err <= 10 * noise + 2e-3:
            bad.append((k, err, noise))
    assert not bad, bad
    assert set(noisy) <= {"F"}, (noisy, "the horizon is too long for a wiring check: lower SHAPE_ROPE_WIRING_STEPS")
    assert float(gf["a"][:, [0, 2, 3, 5]].abs().min()) > 0        # both end points of the push receive a gradient


@pytest.mark.gpu
@pytest.mark.parametrize("n_prim", [1, 2])
def test_mpm_focus_kernel_matches_torch(n_prim):
    """ud_mpm_focus_* against pre_step's torch expressions (mpm_env.py:99-114): the shift, the moved cloud and trajectories, and
    the gradient through the mean (x and z only) with cotangents on the shift itself too (shape_rope adds it to the actions)."""
    from unidom_amd.envs.basic import _fused
    dev = torch.device("cuda", 0)
    B, N, S = 3, 777, 5
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g)
    x, pos = rnd(B, N, 3) * 0.1 + 0.4, [rnd(B, S, 3) * 0.1 + 0.5 for _ in range(n_prim)]
    center = (0.25, 0.1, 0.375)
    w = [rnd(B, N, 3), rnd(B, 3)] + [rnd(B, S, 3) for _ in range(n_prim)]

    def run(fusedp):
        xx, lp = x.clone().requires_grad_(True), [p.clone().requires_grad_(True) for p in pos]
        if fusedp:
            xs, shift, ps = _fused.mpm_focus(center, xx, lp)
        else:
            sh = torch.tensor(center, device=dev) - xx.mean(1)
            shift = torch.cat([sh[:, 0:1], torch.zeros_like(sh[:, 0:1]), sh[:, 2:3]], -1)
            xs, ps = xx + shift[:, None], [p + shift[:, None] for p in lp]
        outs = [xs, shift] + list(ps)
        sum((t * wt).sum() for t, wt in zip(outs, w)).backward()
        return [t.detach() for t in outs], [xx.grad] + [p.grad for p in lp]

    (of, gf), (ou, gu) = run(True), run(False)
    for a, b in zip(of + gf, ou + gu):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-6)
    assert float(of[1][:, 1].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n_prim,focus", [(1, True), (2, True), (1, False)])
def test_mpm_finish_kernel_matches_torch_with_nonfinite_states(n_prim, focus):
    """ud_mpm_finish_* against the torch expressions it replaces (mpm_env.py:116-125, :150-154, :90-94, :57-76) on a state holding
    NaN and +-inf: nan_to_num's values (0, +-FLT_MAX) and gradient mask, reward and observation, the shift's gradient."""
    from unidom_amd.envs.basic import _fused
    from unidom_amd.utils.util import calc_l2
    dev = torch.device("cuda", 0)
    B, N, S = 3, 300, 7
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g)
    x, v, Cm, F, J = rnd(B, N, 3) * 0.1 + 0.4, rnd(B, N, 3), rnd(B, N, 3, 3), rnd(B, N, 3, 3), rnd(B, N)
    x[0, 5, 1], x[1, 7, 0], v[2, 3, 2], Cm[0, 1, 2, 2], F[1, 2, 0, 1], J[0, 0] = (float("nan"), float("inf"), -float("inf"),
                                                                                  float("nan"), float("inf"), float("nan"))
    pos = [rnd(B, S, 3) * 0.1 + 0.5 for _ in range(n_prim)]
    goal = rnd(N, 3) * 0.1 + 0.4
    sh0 = rnd(B, 3) * 0.05
    wts = {}

    def run(fusedp):
        lv = [t.clone().requires_grad_(True) for t in (x, v, Cm, F)] + [p.clone().requires_grad_(True) for p in pos]
        xx, vv, CC, FF = lv[:4]
        lp = lv[4:]
        shift = sh0.clone().requires_grad_(True) if focus else None
        if fusedp:
            (xo, vo, Co, Fo, Jo), reward, obs, po = _fused.mpm_finish(xx, vv, CC, FF, J, shift, goal, lp)
        else:
            xo, po = (xx - shift[:, None], [p - shift[:, None] for p in lp]) if focus else (xx, lp)
            # jnp.nan_to_num is a chain of selects, so a replaced entry passes NO cotangent; torch.nan_to_num's backward multiplies by
            # isfinite(x) instead and lets a non-finite cotangent through as NaN (here 0 * 2 * FLT_MAX from the +inf entry's distance)
            n2n = lambda t: torch.where(torch.isfinite(t), t, torch.nan_to_num(t.detach()))
            xo, vo, Co, Fo, Jo = (n2n(t) for t in (xo, vv, CC, FF, J))
            reward = math.e ** (-calc_l2(xo, goal) * 10)
            obs = torch.cat([xo.reshape(B, -1), vo.reshape(B, -1), po[0].reshape(B, -1)], -1)
        outs = {"x": xo, "v": vo, "C": Co, "F": Fo, "reward": reward, "obs": obs, **{f"p{i}": p for i, p in enumerate(po)}}
        loss = 0
        for k, t in outs.items():
            if k not in wts:
                wts[k] = torch.randn(t.shape, device=dev, generator=g)
            loss = loss + (t * wts[k]).sum()
        loss.backward()
        return {**{k: t.detach() for k, t in outs.items()}, "J": Jo.detach()}, [t.grad for t in lv] + ([shift.grad] if focus else [])

    of, gf = run(True)
    ou, gu = run(False)
    for k in ou:
        torch.testing.assert_close(of[k], ou[k], rtol=1e-5, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
    assert float(of["x"][0, 5, 1]) == 0.0 and float(of["x"][1, 7, 0]) == torch.finfo(torch.float32).max and float(of["J"][0, 0]) == 0.0
    for i, (a, b) in enumerate(zip(gf, gu)):
        assert torch.isfinite(a).all(), i
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5, msg=lambda m, i=i: f"grad {i}: {m}")
    assert float(gf[0][0, 5, 1]) == 0.0 and float(gf[0][1, 7, 0]) == 0.0 and float(gf[1][2, 3, 2]) == 0.0   # masked by nan_to_num


@pytest.mark.gpu
def test_mpm_env_done_mirror_and_auto_reset():
    """The env tracks cur_step on the host so that steps on which nobody finishes skip auto_reset (and never wait for the device);
    a short episode exercises the other branch: same `done`, keys, reset positions and cur_step as the op-by-op step_diff at every
    step, through two episode ends, and a state the env has not seen (cur_step tensor made elsewhere) is picked up by value."""
    B = 3
    env, st0 = _mpm_env_and_state("whip_rope", B, max_steps=2)
    act = _mpm_actions("whip_rope", env, st0, B)
    sf = su = st0
    dones = []
    with torch.no_grad():
        for k in range(5):
            of, rf, df, inf_f = env.step_diff(act, sf)
            ou, ru, du, inf_u = env.step_diff_unfused(act, su)
            sf, su = inf_f["state"], inf_u["state"]
            assert torch.equal(df, du) and torch.equal(sf.cur_step, su.cur_step), k
            np.testing.assert_array_equal(np.asarray(sf.key), np.asarray(su.key))
            torch.testing.assert_close(sf.x, su.x, rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(sf.primitives[0].position, su.primitives[0].position, rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(of, ou, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(rf, ru, rtol=1e-4, atol=1e-6)
            dones.append(bool(df.all()))
        assert dones == [False, True, False, True, False]
        foreign = sf._replace(cur_step=torch.ones((B,), dtype=torch.int32, device=env.device))
        _, _, d, info = env.step_diff(act, foreign)
        assert bool(d.all()) and int(info["state"].cur_step.max()) == 0


@pytest.mark.gpu
def test_pour_soup_reset_step_matches_oracle_and_grad():
    """pour_soup (pour_soup_env.py:110-183): 2877 liquid + 2 x 343 tofu + 4068 vegetable particles = 7631, the count behind the
    reference's hard-coded observation_size 45861; mixed material / hardness per particle, two container bowls, n_grid 128 with
    res 128x64x128.  One env.step (25 substeps) agrees with the CPU oracle driven with the same shifted state; the missing goal
    file scores against the origin (mpm_env.py:46-48) and the reward's gradient reaches the bowl's translation and tilt."""
    from oracle.pyoracle import MpmOracle
    from unidom_amd.envs.registration import env_functions
    env = env_functions["pour_soup"](batch_size=2, seed=1)
    obs, st = env.reset(np.array([0, 7], np.uint32))
    conf, sim = env.conf, env.simulator
    N, S = st.x.shape[1], conf.steps
    assert env.size_list == [2877, 343, 343, 4068] and N == 7631 and S == 25 and obs.shape == (2, 45861) == (2, env.observation_size)
    assert tuple(env.goal.shape) == (1, 3) and float(env.goal.abs().sum()) == 0.0 and env.max_steps == 120
    mat, hard = np.asarray(sim.material), np.asarray(sim.h)
    assert (mat[:2877] == 0).all() and (mat[2877:] == 1).all() and (hard[:2877] == 1).all() and np.allclose(hard[2877:], 0.3)
    veg = st.x[0, 2877 + 686:].cpu().numpy()
    np.testing.assert_allclose(veg.mean(0), [0.55, 0.2, 0.5], atol=1e-6)         # (veg - mean) / 400 + (0.55, 0.2, 0.5)
    assert np.ptp(veg, 0).max() < 21 / 400 and len(st.primitives) == 2 and sim.sdf_kind == "container" and sim._h_large
    a = torch.tensor([[0.8, 0.3, -0.5, 0.6, -0.2, 0.3], [-0.4, 0.0, 0.9, -0.5, 0.4, 0.1]], device=env.device, requires_grad=True)
    obs2, reward, done, info = env.step_diff(a, st)
    sim.check_status()
    s1 = info["state"]
    assert obs2.shape == (2, 45861) and torch.isfinite(obs2).all() and not bool(done.any())
    npy = lambda t: t.detach().cpu().numpy()
    want_r = np.exp(-10 * np.sqrt((npy(s1.x).astype(np.float64) ** 2).mean(-1)).mean(-1))
    np.testing.assert_allclose(npy(reward), want_r, rtol=2e-6)
    # ---- oracle -----------------------------------------------------------------------------------------------------------
    x0 = npy(st.x)
    shift = (np.array(conf.res, np.float32) * np.float32(0.5) / np.float32(conf.n_grid) - x0.mean(1, dtype=np.float32)).astype(np.float32)
    shift[:, 1] = 0
    act = np.concatenate([npy(a), np.zeros((2, 6), np.float32)], -1)
    act[:, :6] = act[:, :6] / np.float32(500.0)
    act = act + np.float32(1e-12)
    act[:, 1] = 0
    ppos = np.stack([npy(p.position) for p in st.primitives], 1) + shift[:, None, None]
    prot = np.stack([npy(p.rotation) for p in st.primitives], 1)
    psize = np.stack([npy(p.size) for p in st.primitives], 1)
    ost = dict(x=x0 + shift[:, None], v=npy(st.v), C=npy(st.C), F=npy(st.F), J=npy(st.J), ppos=ppos, prot=prot, psize=psize,
               friction=npy(st.friction).reshape(2), mu=npy(st.mu).reshape(2), lamda=npy(st.lamda).reshape(2), action=act)
    orc = MpmOracle(N, n_grid=conf.n_grid, res=conf.res, steps=S, dt=conf.dt, position_control=False, material=mat, hardness=hard,
                    n_prim=2, sdf="container")
    o32 = orc.step_fwd(ost, nthreads=2)
    o64 = orc.step_fwd({k: v.astype(np.float64) for k, v in ost.items()}, nthreads=2)
    xk = npy(s1.x) + shift[:, None]
    gx, gv = np.abs(o32["x"] - o64["x"]).max(), np.abs(o32["v"] - o64["v"]).max() / np.abs(o64["v"]).max()
    ex, ev = np.abs(xk - o64["x"]).max(), np.abs(npy(s1.v) - o64["v"]).max() / np.abs(o64["v"]).max()
    assert ex < 3 * gx + 2e-6 and ev < 3 * gv + 1e-4, (ex, gx, ev, gv)        # as close to f64 as the f32 restatement is
    for i in range(2):
        np.testing.assert_allclose(npy(s1.primitives[i].position[:, 0]) + shift, o32["ppos"][:, i, 0], atol=2e-7)
    reward.sum().backward()
    assert torch.isfinite(a.grad).all() and a.grad[:, [0, 2]].abs().min() > 0 and (a.grad[:, 1] == 0).all()


@pytest.mark.gpu
def test_pour_water_gradient_with_and_without_collide_records(monkeypatch):
    """ud_mpm_conf.tune_collide_records: by default the forward's grid op leaves exp(-dist * softness) and the finite-difference normal of
    every cell and primitive beside the grid checkpoint and the grid-op adjoint (lg_grid_adj_rec) re-runs the collide chain from them; with
    -1 the adjoint evaluates the seven SDFs per cell and primitive again (lg_grid_adj).  The records are the bits the SDFs would give, so
    one env.step of pour_water (two container bowls, 23 substeps) must return the same action gradient either way -- up to the order of the
    float atomics both paths share (1e-5 relative); the forward does not read the records (two forwards differ only by the arrival order of
    the p2g atomics: an ulp here and there)."""
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    from unidom_amd.envs.registration import env_functions
    got = {}
    for rec in (0, -1):
        monkeypatch.setattr(SimpleMPMSimulator, "default_tuning", {"collide_records": rec})
        env = env_functions["pour_water"](batch_size=2, seed=1)
        _, st = env.reset(np.array([0, 7], np.uint32))
        a = torch.tensor([[0.8, 0.3, -0.5, 0.6, -0.2, 0.3], [-0.4, 0.0, 0.9, -0.5, 0.4, 0.1]], device=env.device, requires_grad=True)
        _, reward, _, info = env.step_diff(a, st)
        reward.sum().backward()
        env.simulator.check_status()
        assert env.simulator.tuning == {"collide_records": rec}
        got[rec] = (info["state"].x.detach().cpu().numpy(), a.grad.cpu().numpy())
    np.testing.assert_allclose(got[0][0], got[-1][0], rtol=0, atol=2e-7)
    g0, g1 = got[0][1], got[-1][1]
    assert np.isfinite(g0).all() and np.abs(g0).max() > 0
    assert np.abs(g0 - g1).max() <= 1e-5 * np.abs(g1).max(), (g0, g1)


def test_pour_soup_step_matches_oracle_with_one_lane_kernels(monkeypatch):
    """bench.py's pour_soup workload (32 envs x 7631 particles) runs the one-lane-per-particle kernels of the many-workgroup path;
    at the two envs the oracle can follow the four-lane ones would run.  tune_lanes = 1 (ud_mpm_conf, fixed at create) puts the same
    env.step through the one-lane kernels: internal spatial order, block window / hash staging, soft contact, mixed materials."""
    from unidom_amd.engine.mpm_simulator import SimpleMPMSimulator
    monkeypatch.setattr(SimpleMPMSimulator, "default_tuning", {"lanes": 1})
    test_pour_soup_reset_step_matches_oracle_and_grad()
