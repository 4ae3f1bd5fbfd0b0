"""GPU: RCCL entered once on the pool's image before the first multi-GPU run.

The N-rank path (one process per GPU, `torch.distributed` backend "nccl" = RCCL, ONE mean all-reduce of the already-clipped flat
policy gradient per update: apg.py:233-235, :269-271) is covered on the CPU by gloo tests (tests/test_distributed_cpu.py,
tests/test_bench_contract.py).  What gloo cannot show -- the RCCL library loads, the rank binds its device, the group comes up
with HSA_ENABLE_IPC_MODE_LEGACY=0, a collective on device memory completes and the group shuts down -- is what this test adds, with
the driver's own command line at N = 1: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --workload selftest`.
The child is started by tests/conftest.py::pytest_sessionstart BEFORE this process touches the GPU (a GPU-initialised process must
not fork + exec on this pool).  It proves nothing about N > 1 scaling.  (The file name sorts last on purpose: under `pytest -x` a
failure here -- the one test that depends on the box's RCCL set-up rather than on this repo's kernels -- must not hide the parity tests.)"""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_group_runs_the_update_collective(request):
    child = getattr(request.config, "_rccl_child", None)
    if child is None:
        import torch
        if torch.cuda.device_count() < 1:
            pytest.skip("no GPU")
        if torch.cuda.is_initialized():
            pytest.skip("run with -m gpu: the RCCL child must be started before this process initialises the GPU (tests/conftest.py)")
        from conftest import run_rccl_child
        child = run_rccl_child()
    assert child["rc"] == 0, (child["rc"], child["stderr"][-3000:], child["stdout"][-1000:])
    lines = [ln for ln in child["stdout"].splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, child["stdout"][-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["n_ranks_seen"] == 1 and line["steps"] == 3 and line["value"] > 0
    c = line["collective"]
    assert c["backend"] == "nccl" and c["device"] == "cuda:0", c                  # RCCL, bound to the rank's device
    assert c["known_answer_allreduce_ok"] is True                                  # an all-reduce on device memory returned the known sum
    assert c["gradient_allreduces_timed"] == 3 and c["gradient_allreduce_ms_mean"] > 0   # GradSync.step's collective, once per update
    assert line["replicas_identical"] is True
