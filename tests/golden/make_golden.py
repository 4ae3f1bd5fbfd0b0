#!/usr/bin/env python3
"""Re-pack the reference's recorded *data* fixtures into small .npz files.

Test infrastructure only. Reads (as data, with a stub Unpickler -- no reference
code is imported or executed, jax is not needed):

  /root/reference/DaXBench/daxbench/algorithms/expert_demo/whip_rope/demo_0.pkl
  /root/reference/DaXBench/daxbench/algorithms/expert_demo/fold_cloth1/demo_*.pkl
  /root/reference/DaXBench/daxbench/core/envs/goals/{fold_cloth1,whip_rope}/goal.npy

and writes tests/golden/{whip_rope_demo0.npz,fold_cloth1_demos.npz} plus the two
goal arrays (also copied into unidom_amd/envs/goals/, where the envs load them
exactly like the reference does: cloth_env.py:60-65, mpm_env.py:46-51).

The pickles hold (state_k, action_k, obs_k) trajectories recorded by the
reference's own step_diff (cloth_env.py:306-318, whip_rope_env.py:159-165); see
SURVEY.md section 0/F3 for what they pin.  /root/reference does not exist on the
GPU box, so only the committed outputs of this script are used by the tests.
"""
import collections
import glob
import os
import pickle
import shutil
import sys

import numpy as np

REF = "/root/reference/DaXBench/daxbench"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

ClothState = collections.namedtuple(
    "ClothState", "x v primitive0 primitive1 action0 action1 key cur_step stiffness mu")
MPMState = collections.namedtuple(
    "MPMState", "x v C F J cur_step primitives key friction mu lamda")
PrimitiveState = collections.namedtuple(
    "PrimitiveState", "size dim friction softness color position rotation v w xyz_limit "
                      "action_buffer action_scale min_dist dist_norm")


def _reconstruct_device_array(fun, args, arr_state, aval_state):
    arr = fun(*args)
    arr.__setstate__(arr_state)
    return np.asarray(arr)


class _Stub(pickle.Unpickler):
    def find_class(self, module, name):
        if name == "ClothState":
            return ClothState
        if name == "MPMState":
            return MPMState
        if name == "PrimitiveState":
            return PrimitiveState
        if module.startswith("jax") and name == "reconstruct_device_array":
            return _reconstruct_device_array
        if module.startswith("numpy.core"):
            module = module.replace("numpy.core", "numpy._core")
        if module.startswith("numpy"):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load {module}.{name}")


def load(path):
    with open(path, "rb") as f:
        return _Stub(f).load()


def main():
    # ---- whip_rope (MPM forward golden) -------------------------------------------------
    d = load(f"{REF}/algorithms/expert_demo/whip_rope/demo_0.pkl")
    st = d["state"]
    out = dict(
        x=np.stack([np.asarray(s.x)[0] for s in st]).astype(np.float32),
        v=np.stack([np.asarray(s.v)[0] for s in st]).astype(np.float32),
        C=np.stack([np.asarray(s.C)[0] for s in st]).astype(np.float32),
        F=np.stack([np.asarray(s.F)[0] for s in st]).astype(np.float32),
        J=np.stack([np.asarray(s.J)[0] for s in st]).astype(np.float32),
        cur_step=np.stack([np.asarray(s.cur_step)[0] for s in st]),
        friction=np.stack([np.asarray(s.friction)[0] for s in st]),
        mu=np.stack([np.asarray(s.mu)[0] for s in st]),
        lamda=np.stack([np.asarray(s.lamda)[0] for s in st]),
        prim_position=np.stack([np.asarray(s.primitives[0].position)[0] for s in st]).astype(np.float32),
        prim_rotation=np.stack([np.asarray(s.primitives[0].rotation)[0] for s in st]).astype(np.float32),
        prim_v=np.stack([np.asarray(s.primitives[0].v)[0] for s in st]).astype(np.float32),
        prim_size=np.asarray(st[0].primitives[0].size)[0].astype(np.float32),
        action=np.stack([np.asarray(a)[0] for a in d["action"]]).astype(np.float32),
        obs=np.stack([np.asarray(o)[0] for o in d["obs"]]).astype(np.float32),
    )
    np.savez_compressed(f"{HERE}/whip_rope_demo0.npz", **out)
    print("whip_rope:", {k: v.shape for k, v in out.items()})

    # ---- fold_cloth1 (primitive kinematics + PRNG split chain; NOT cloth x/v, see F3) ----
    recs = collections.defaultdict(list)
    for p in sorted(glob.glob(f"{REF}/algorithms/expert_demo/fold_cloth1/demo_*.pkl")):
        d = load(p)
        demo_id = int(os.path.basename(p).split("_")[1].split(".")[0])
        for k in range(len(d["state"]) - 1):
            s0, s1, a = d["state"][k], d["state"][k + 1], d["action"][k]
            recs["demo"].append(demo_id)
            recs["k"].append(k)
            recs["action"].append(np.asarray(a)[0])
            for nm, s in (("s0", s0), ("s1", s1)):
                recs[nm + "_x"].append(np.asarray(s.x)[0])
                recs[nm + "_v"].append(np.asarray(s.v)[0])
                recs[nm + "_primitive0"].append(np.asarray(s.primitive0)[0])
                recs[nm + "_primitive1"].append(np.asarray(s.primitive1)[0])
                recs[nm + "_key"].append(np.asarray(s.key)[0])
                recs[nm + "_cur_step"].append(np.asarray(s.cur_step)[0])
                recs[nm + "_stiffness"].append(np.asarray(s.stiffness)[0])
                recs[nm + "_mu"].append(np.asarray(s.mu)[0])
    recs = {k: np.stack(v) for k, v in recs.items()}
    np.savez_compressed(f"{HERE}/fold_cloth1_demos.npz", **recs)
    print("fold_cloth1:", {k: (v.shape, v.dtype) for k, v in recs.items()})

    # ---- goals (inputs of the reward) -----------------------------------------------------
    for task in ("fold_cloth1", "whip_rope"):
        src = f"{REF}/core/envs/goals/{task}/goal.npy"
        shutil.copyfile(src, f"{HERE}/goal_{task}.npy")
        dst = f"{REPO}/unidom_amd/envs/goals/{task}"
        os.makedirs(dst, exist_ok=True)
        shutil.copyfile(src, f"{dst}/goal.npy")
        print(task, "goal", np.load(src).shape)
    # ---- fold_tshirt cloth mask (input of the simulator's constructor) ---------------------------------------------
    # The reference builds it from others/t-shirt.jpg with cv2 (fold_cloth_tshirt_env.py:50-67); cv2 is not available here, but
    # the recorded reset state of expert_demo/fold_tshirt/demo_0.pkl (y = 0, lattice positions) holds the same information:
    # particle p sits at lattice point (round(x / cell), N - round(z / cell)), in row-major order of the mask (cloth_simulator.py:52).
    class _Legacy(_Stub):                       # these demos predate ClothState.stiffness / .mu: 8 fields
        def find_class(self, module, name):
            if name == "ClothState":
                return type("LegacyClothState", (tuple,), {"__new__": lambda cls, *a: tuple.__new__(cls, a)})
            return super().find_class(module, name)
    with open(f"{REF}/algorithms/expert_demo/fold_tshirt/demo_0.pkl", "rb") as f:
        x0 = np.asarray(_Legacy(f).load()["state"][0][0])[0]
    N = 180
    assert x0.shape == (3573, 3) and np.all(x0[:, 1] == 0)
    ii, jj = np.round(x0[:, 0] * N).astype(int), N - np.round(x0[:, 2] * N).astype(int)
    tmask = np.zeros((N, N), np.uint8)
    tmask[ii, jj] = 1
    assert tmask.sum() == 3573 and np.array_equal(np.stack(np.nonzero(tmask), 1), np.stack([ii, jj], 1))
    os.makedirs(f"{REPO}/unidom_amd/envs/others", exist_ok=True)
    np.save(f"{REPO}/unidom_amd/envs/others/tshirt_mask.npy", tmask)
    print("fold_tshirt mask", tmask.shape, int(tmask.sum()))

    # ---- pour_soup vegetable point cloud (input of reset: pour_soup_env.py:152-159) -----------------------------------
    # core/engine/pyrender/models/veg/model.pcd, PCD v0.7 binary: 8161 points x {rgb, normal xyz, x y z, 4 pad bytes}.  The env reads
    # it with open3d; here the xyz columns are kept as a data file and the env restates open3d's voxel_down_sample in numpy.
    raw = open(f"{REF}/core/engine/pyrender/models/veg/model.pcd", "rb").read()
    start = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    assert b"FIELDS rgb normal_x normal_y normal_z x y z _" in raw[:start] and b"POINTS 8161" in raw[:start]
    rec = np.dtype([("rgb", "<f4"), ("n", "<f4", 3), ("xyz", "<f4", 3), ("pad", "u1", 4)])
    veg = np.frombuffer(raw[start:start + 8161 * rec.itemsize], dtype=rec)["xyz"].copy()
    np.save(f"{REPO}/unidom_amd/envs/others/veg_points.npy", veg)
    print("pour_soup veg cloud", veg.shape, veg.min(0), veg.max(0))

    # ---- sibling cloth demos (fold_cloth3, unfold_cloth1/3, fold_tshirt): what they pin exactly, like fold_cloth1's, is the
    # primitive kinematics (get_pnp_actions + robot_step's action scaling + the primitive update), the 40-split key chain per
    # step_diff and cur_step -- none of which reads the cloth's x/v (which these legacy recordings do not pin), so
    # only those small fields are kept (fold_tshirt's x alone would be 43 KB per state).  The fold_tshirt pickles predate
    # ClothState.stiffness / .mu (8 fields).
    recs = collections.defaultdict(list)
    for task in ("fold_cloth3", "unfold_cloth1", "unfold_cloth3", "fold_tshirt"):
        for p in sorted(glob.glob(f"{REF}/algorithms/expert_demo/{task}/demo_*.pkl")):
            with open(p, "rb") as f:
                d = _Legacy(f).load()
            demo_id = int(os.path.basename(p).split("_")[1].split(".")[0])
            for k in range(len(d["state"]) - 1):
                s0, s1 = d["state"][k], d["state"][k + 1]
                recs["task"].append(task)
                recs["demo"].append(demo_id)
                recs["k"].append(k)
                recs["action"].append(np.asarray(d["action"][k])[0])
                for nm, st in (("s0", s0), ("s1", s1)):
                    recs[nm + "_primitive0"].append(np.asarray(st[2])[0])
                    recs[nm + "_primitive1"].append(np.asarray(st[3])[0])
                    recs[nm + "_key"].append(np.asarray(st[6])[0])
                    recs[nm + "_cur_step"].append(np.asarray(st[7])[0])
    recs = {k: np.stack(v) for k, v in recs.items()}
    np.savez_compressed(f"{HERE}/cloth_sibling_demos.npz", **recs)
    print("sibling cloth demos:", {t: int((recs["task"] == t).sum()) for t in np.unique(recs["task"])})

    # goals of the sibling envs on the same kernels (data files: inputs of their reward)
    for task in ("fold_cloth3", "unfold_cloth1", "unfold_cloth3", "fold_tshirt", "shape_rope", "pour_water"):
        src = f"{REF}/core/envs/goals/{task}/goal.npy"
        dst = f"{REPO}/unidom_amd/envs/goals/{task}"
        os.makedirs(dst, exist_ok=True)
        shutil.copyfile(src, f"{dst}/goal.npy")
        print(task, "goal", np.load(src).shape)


if __name__ == "__main__":
    sys.exit(main())
