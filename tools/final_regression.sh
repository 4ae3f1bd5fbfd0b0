#!/bin/bash
# end-of-round regression on the GPU box: the GPU test suite, then every bench line of profiles/README.md, then (KS=1) one
# rocprofv3 kernel-stats CSV per bench workload.   usage: TAG=r05 KS=1 [PART=1|2|3] bash tools/final_regression.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05}; O=gpurun_out/final; mkdir -p $O
# PART (gpurun caps a call at 20 minutes): 1 = GPU suite + deterministic-mode cost + cloth / one-workgroup MPM lines, 2 = many-workgroup MPM lines,
# 3 = PlasticineLab lines; unset = everything
want() { [ -z "$PART" ] || [ "$PART" = "$1" ]; }
if want 1; then
timeout -k 10 1100 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/${TAG}_gpu_tests.log; tail -2 $O/${TAG}_gpu_tests.log
timeout -k 10 300 python tools/det_cost.py > $O/${TAG}_det_cost.txt 2>$O/det_cost.err; cat $O/${TAG}_det_cost.txt
fi
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" 2>$O/$name.err | tail -n 1 > $O/${TAG}_bench_line_$name.json
  python - <<PY
import json
d = json.load(open("$O/${TAG}_bench_line_$name.json"))
i = d["roofline"].get("issue") or {}
i = i.get("frac") if "frac" in i else {k: v.get("frac") for k, v in i.items() if isinstance(v, dict)}
ro = d.get("reference_order")
if ro: print("   reference_order", round(ro["value"]), ro["kernel_ms"], "cpu order1 / order2", (d.get("cpu_baseline") or {}).get("reference_order", {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
print("$name", round(d["value"]), "%.2f ms" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"], d["roofline"].get("kernel_ms"), "traffic", d["roofline"].get("traffic"), "issue", i, "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
  if [ -n "$KS" ]; then
    rm -rf $O/ks_$name; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ks_$name -o p -f csv -- python3 bench.py "$@" --no-cpu-baseline --no-saturation > $O/ks_$name.log 2>&1 \
      && cp $O/ks_$name/p_kernel_stats.csv $O/${TAG}_kernel_stats_$name.csv && rm -rf $O/ks_$name
  fi
}
if want 1; then
run fold_cloth1
run fold_cloth1_para_32envs --workload fold_cloth1_para
run fold_tshirt --workload fold_tshirt
run whip_rope --workload whip_rope
run whip_rope_eager --workload whip_rope --no-graph --no-cpu-baseline
run whip_rope_256envs --workload whip_rope --envs 256 --no-cpu-baseline
fi
if want 2; then
run whip_rope_ngrid128 --workload whip_rope --n-grid 128
run whip_rope_ngrid256 --workload whip_rope --n-grid 256
run shape_rope --workload shape_rope
run pour_water --workload pour_water
run pour_soup --workload pour_soup
fi
if want 3; then
run torus_ngrid64 --workload torus --n-grid 64
run torus_ngrid128 --workload torus --n-grid 128
run torus_grad_ngrid64 --workload torus --n-grid 64 --plb-grad
run torus_grad_ngrid128 --workload torus --n-grid 128 --plb-grad
fi
