import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def run_rccl_child(timeout=420):
    """One rank of `python -m torch.distributed.run ... bench.py --workload selftest` with the nccl (= RCCL) backend as a CHILD process:
    the driver's own N-rank command line at N = 1.  Must be started by a process that has not initialised the GPU (a GPU-initialised
    parent must not fork + exec on this pool) -- pytest_sessionstart below does it before any test runs."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "UNIDOM_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["UNIDOM_DIST_JOIN_SINGLE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "selftest"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        return {"rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr[-4000:]}
    except subprocess.TimeoutExpired as e:
        return {"rc": None, "stdout": (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""),
                "stderr": f"timed out after {timeout}s"}


def pytest_sessionstart(session):
    """-m gpu runs only: start the RCCL child of tests/test_z_rccl_gpu.py now, while this process has not touched the GPU
    (torch.cuda.device_count() does not initialise it on this image; is_available() would)."""
    session.config._rccl_child = None
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    session.config._rccl_child = run_rccl_child()


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def fold_cloth1_mask(N=80, size=16):
    """fold_cloth1_env.py:48-53"""
    m = np.zeros((N, N), dtype=np.float32)
    m[size * 2:size * 3, size * 2:size * 4] = 1
    return m


def cloth_reset_x(N=80, mask=None):
    """cloth_simulator.py:339-353"""
    mask = fold_cloth1_mask(N) if mask is None else mask
    c = 1.0 / N
    ii, jj = np.nonzero(mask)
    return np.stack([ii * c, np.zeros_like(ii, dtype=np.float64), (N - jj) * c], -1).astype(np.float32)


def make_cloth_case(rng, B, T, P_x=None, deform=0.002, v_scale=0.05, grasp=True):
    """Seeded synthetic cloth states/actions: deformed lattice, gripper 0 placed on a particle (so the
    discrete grasp event fires), macro actions that lift/move it."""
    x0 = cloth_reset_x() if P_x is None else P_x
    P = x0.shape[0]
    x = np.repeat(x0[None], B, 0).astype(np.float32)
    x[:, :, [0, 2]] += (rng.normal(size=(B, 1, 2)) * 0.05).astype(np.float32)
    x += (rng.normal(size=x.shape) * deform).astype(np.float32)
    x = np.abs(x).astype(np.float32)
    v = (rng.normal(size=x.shape) * v_scale).astype(np.float32)
    prim = np.zeros((B, 2, 4), np.float32)
    prim[:, 1] = [1, 1, 1, 0.01]
    for b in range(B):
        p = rng.integers(0, P)
        prim[b, 0, :3] = x[b, p] + (np.array([0, 0.002, 0]) if grasp else np.array([0, 0.2, 0]))
        prim[b, 0, 3] = 0.01
    actions = (rng.normal(size=(T, B, 8)) * 0.3).astype(np.float32)
    actions[..., 3] = rng.uniform(0, 1, size=(T, B)) < 0.3
    actions[..., 4:] = 0
    k = rng.uniform(600, 1500, size=B).astype(np.float32)
    mu = rng.uniform(0.3, 0.9, size=B).astype(np.float32)
    return x, v, prim, k, mu, actions
