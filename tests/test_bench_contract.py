"""CPU: the bench line contract.  bench.py itself needs a GPU; what can be checked here is that it parses its arguments the
way the driver calls it and that the committed headline line (profiles/, produced by `python bench.py` on the box) carries every
key the contract names, with the right types -- a schema regression would otherwise only show up at round end."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_cli_accepts_the_driver_arguments():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in out.stdout


def test_committed_headline_line_has_the_contract_keys():
    line = json.load(open(os.path.join(ROOT, "profiles", "r05b_bench_line_fold_cloth1.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["scaling"] == "weak" and line["n_gpus"] == 1 and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["unit"] == base.get("unit", line["unit"])
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["traffic"] is None or roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    assert line["value"] > cpu["value"]          # not a target, but a GPU line slower than the CPU port would be a regression
    assert cpu["one_thread"]["cores"] == 1 and cpu["all_cores"]["cores"] >= cpu["cores"] and "-O3" in cpu["sample"]
    assert line["n_ranks_seen"] == 1 and line["allreduce_bytes_per_update"] == 0 and line["config"]["order"].startswith("v2")
    # round 5: the same update in the reference's literal operation order beside `value`, and the CPU restatement timed in BOTH orders
    # (`value` = the order the GPU leg ran: like for like)
    ro = line["reference_order"]
    assert ro["kernel_mode"] == 3 and 0 < ro["value"] < line["value"] and ro["kernel_ms"]["fwd"] > roof["kernel_ms"]["fwd"]
    assert cpu["order"] == 2 and cpu["reference_order"]["order"] == 1 and 0 < cpu["reference_order"]["value"] < ro["value"]
    assert "order 2" in cpu["value_is"] and "literal operation order" in cpu["reference_order"]["sample"]


def test_bench_gpus_2_launches_its_own_ranks_gloo():
    """`python bench.py --gpus 2` with no RANK in the environment starts the two ranks itself (torch.distributed.run,
    127.0.0.1), rank 0 prints ONE JSON line, and the parent relays it.  Driven here with gloo on the CPU
    (UNIDOM_DIST_BACKEND=gloo) on the simulator-free selftest workload: the launcher, init_distributed, the barrier-bracketed
    timed region, the max-over-ranks timing and the gradient all-reduce are the ones every workload uses."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["UNIDOM_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--workload", "selftest"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["allreduce_bytes_per_update"] == 4 * (32 * 64 + 64 + 64 * 32 + 32 + 32 * 12 + 12)
    assert line["replicas_identical"] is True and line["value"] > 0
    # the N-rank line explains itself: per-rank wall time (max = the time `value` is computed from), per-rank kernel times (none here: the
    # selftest has no simulator) and the per-rank time of the one collective per update, measured around dist.all_reduce in GradSync.step
    r = line["ranks"]
    assert len(r["time_s"]["per_rank"]) == 2 and r["time_s"]["max"] >= r["time_s"]["min"] > 0
    assert abs(r["time_s"]["max"] - line["ms_per_step"] * 3 / 1e3) < 1e-6
    assert r["kernel_ms"] == {"fwd": [None, None], "bwd": [None, None]}
    assert len(r["allreduce_ms"]["mean"]) == 2 and all(v is not None and v > 0 for v in r["allreduce_ms"]["mean"] + r["allreduce_ms"]["max"])


def test_bench_self_launch_propagates_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["UNIDOM_DIST_BACKEND"] = "gloo"
    # the cloth workload cannot run without a GPU: every rank raises, the parent must exit non-zero and print no result line
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-saturation"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
