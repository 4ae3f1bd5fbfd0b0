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
    line = json.load(open(os.path.join(ROOT, "profiles", "r01h_bench_line_fold_cloth1.json")))
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
