#!/bin/bash
# every bench line once with its cpu_baseline (validates the baseline legs); output under gpurun_out/r03d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
run() { name=$1; shift; SECONDS=0; timeout -k 10 400 python bench.py "$@" 2>$O/$name.err | tail -n 1 > $O/line_$name.json; echo "$name: $SECONDS s wall"
  python - <<PY
import json
try:
    d = json.load(open("$O/line_$name.json"))
    r = d["roofline"]
    print("$name", round(d["value"]), "%.2f ms" % d["ms_per_step"], "frac %.4f" % r["frac"], "issue", r.get("issue"), "traffic", r.get("traffic"), "cpu", d.get("cpu_baseline"))
    if d.get("saturation"): print("   saturation", d["saturation"])
except Exception as e:
    print("$name FAILED", e); print(open("$O/$name.err").read()[-1500:])
PY
}
run fold_cloth1
run whip_rope --workload whip_rope
run whip_rope_ngrid128 --workload whip_rope --n-grid 128
run shape_rope --workload shape_rope
run pour_water --workload pour_water
run pour_soup --workload pour_soup
run torus_ngrid64 --workload torus --n-grid 64
run torus_grad_ngrid64 --workload torus --n-grid 64 --plb-grad
