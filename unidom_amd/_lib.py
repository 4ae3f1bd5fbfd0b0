"""ctypes binding of libunidom_hip.so (the C ABI declared in include/unidom_hip.h).

The HIP library IS the product path: there is no CPU or eager-PyTorch fallback. If the shared object is
missing or a call fails, an exception is raised (UnidomError) -- nothing silently routes elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("UNIDOM_HIP_SO", os.path.join(CSRC, "libunidom_hip.so"))   # override: diagnostic builds only

# every symbol include/unidom_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "ud_last_error", "ud_version",
    "ud_cloth_create", "ud_cloth_destroy", "ud_cloth_num_particles", "ud_cloth_ckpt_bytes", "ud_cloth_launch_envs", "ud_cloth_poll_timeouts",
    "ud_cloth_rollout_fwd", "ud_cloth_rollout_bwd",
    "ud_mpm_create", "ud_mpm_destroy", "ud_mpm_ckpt_bytes", "ud_mpm_ckpt_cells", "ud_mpm_launch_plan", "ud_mpm_reset", "ud_mpm_step_fwd", "ud_mpm_step_bwd",
    "ud_plb_create", "ud_plb_destroy", "ud_plb_launch_plan", "ud_plb_poll_timeouts", "ud_plb_step_fwd", "ud_plb_ckpt_bytes", "ud_plb_step_bwd", "ud_plb_loss_fwd", "ud_plb_loss_bwd",
    "ud_chamfer_fwd", "ud_chamfer_bwd", "ud_cloth_pnp_fwd", "ud_cloth_pnp_bwd",
    "ud_mpm_focus_fwd", "ud_mpm_focus_bwd", "ud_mpm_finish_fwd", "ud_mpm_finish_bwd",
]


class UnidomError(RuntimeError):
    pass


class ud_cloth_conf(C.Structure):
    _fields_ = [("N", C.c_int), ("gravity", C.c_float), ("damping", C.c_float), ("dt", C.c_float),
                ("max_v", C.c_float), ("small_num", C.c_float), ("substeps", C.c_int), ("mode", C.c_int), ("max_envs", C.c_int),
                ("one_workgroup_per_env", C.c_int)]


class ud_mpm_conf(C.Structure):
    _fields_ = [("n_particles", C.c_int), ("n_grid", C.c_int), ("res", C.c_int * 3), ("steps", C.c_int),
                ("dt", C.c_float), ("p_mass", C.c_float), ("p_vol", C.c_float), ("gravity", C.c_float * 3),
                ("use_position_control", C.c_int), ("prim_friction", C.c_float), ("prim_softness", C.c_float),
                ("n_primitive", C.c_int), ("sdf_kind", C.c_int), ("grid_ckpt_cells", C.c_int), ("sort_particles", C.c_int),
                ("prim_friction_each", C.c_float * 4), ("prim_softness_each", C.c_float * 4), ("deterministic", C.c_int),
                ("max_envs", C.c_int), ("tune_lanes", C.c_int), ("tune_cluster", C.c_int), ("tune_cluster_part_lanes", C.c_int),
                ("tune_cluster_envs", C.c_int), ("tune_env_groups", C.c_int), ("tune_bwd_two_launch", C.c_int),
                ("tune_collide_records", C.c_int)]


class ud_plb_conf(C.Structure):
    _fields_ = [("n_particles", C.c_int), ("n_grid", C.c_int), ("substeps", C.c_int), ("dt", C.c_double),
                ("gravity", C.c_double * 3), ("ground_friction", C.c_double), ("n_primitives", C.c_int),
                ("radius", C.c_double * 2), ("lower_bound", C.c_double * 3), ("upper_bound", C.c_double * 3),
                ("grid_ckpt_cells", C.c_int), ("max_envs", C.c_int), ("path", C.c_int), ("lanes", C.c_int), ("sort_every", C.c_int)]


def build(force: bool = False) -> str:
    """Compile libunidom_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "unidom_hip.h"))
    stale = (not os.path.exists(SO_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "libunidom_hip.so"] + (["-B"] if force else []))
    return SO_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise UnidomError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {CSRC}`). There is no fallback path.")
        # torch first: its wheel carries its own libamdhip64 / libhsa-runtime64, the library links /opt/rocm's.  Loaded in this order the
        # library's HIP calls resolve to the runtime torch has already brought in (one runtime per process: device pointers and streams
        # are torch's); loaded the other way round the process holds two runtimes and the second one finds "no ROCm-capable device".
        import torch  # noqa: F401
        L = C.CDLL(SO_PATH)
        L.ud_last_error.restype = C.c_char_p
        L.ud_version.restype = C.c_char_p
        L.ud_cloth_ckpt_bytes.restype = C.c_size_t
        L.ud_mpm_ckpt_bytes.restype = C.c_size_t
        L.ud_plb_ckpt_bytes.restype = C.c_size_t
        for name in SYMBOLS:
            if not hasattr(L, name):
                raise UnidomError(f"{SO_PATH} does not export {name}")
        _LIB = L
    return _LIB


def check(rc: int, what: str):
    if rc != 0:
        raise UnidomError(f"{what} failed (status {rc}): {lib().ud_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous float32/uint8 CUDA(HIP) tensor, or NULL."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "libunidom_hip takes contiguous device tensors"
    return C.c_void_p(t.data_ptr())
