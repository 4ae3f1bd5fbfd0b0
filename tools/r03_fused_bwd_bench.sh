#!/bin/bash
# backward of the many-workgroup MPM path: two launches per reverse substep (default) against the four-kernel sequence
# (UD_LG_FUSED_BWD=0), behind the default forward of each workload.   usage: bash tools/r03_fused_bwd_bench.sh [out_dir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r03c}; mkdir -p $O
run() { tag=$1; shift
  for fu in 1 0; do
    UD_LG_FUSED_BWD=$fu timeout -k 10 240 python bench.py "$@" --no-cpu-baseline 2>$O/$tag.err | tail -n 1 > $O/line_${tag}_fused${fu}.json
    python - <<PY
import json
d = json.load(open("$O/line_${tag}_fused${fu}.json"))
print("$tag fused_bwd=$fu", round(d["value"]), "substeps/s  %.2f ms/step" % d["ms_per_step"], d["roofline"].get("kernel_ms"))
PY
  done
}
run shape_rope --workload shape_rope
run rope128 --workload whip_rope --n-grid 128
run pour_water --workload pour_water
