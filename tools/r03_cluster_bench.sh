#!/bin/bash
# the three workloads of the many-workgroup MPM path below 100 k particles per launch, on the persistent cluster kernels
# (parts of 16 / 32 particles) and on the multi-kernel path they replace.   usage: bash tools/r03_cluster_bench.sh [out_dir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r03b}; mkdir -p $O
run() { tag=$1; shift
  for v in "1 128" "1 64" "0 64"; do set -- $v "$@"; c=$1; t=$2; shift 2
    UD_MPM_CLUSTER=$c UD_MPM_CLUSTER_T=$t timeout -k 10 240 python bench.py "$@" --no-cpu-baseline 2>$O/$tag.err | tail -n 1 > $O/line_${tag}_c${c}_t${t}.json
    python - <<PY
import json
d = json.load(open("$O/line_${tag}_c${c}_t${t}.json"))
print("$tag cluster=$c T=$t", round(d["value"]), "substeps/s  %.2f ms/step" % d["ms_per_step"], d["roofline"].get("kernel_ms"))
PY
  done
}
run shape_rope --workload shape_rope
run rope128 --workload whip_rope --n-grid 128
run pour_water --workload pour_water
