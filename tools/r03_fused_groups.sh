#!/bin/bash
# two-launch backward x env groups (streams): substeps/s and kernel ms per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r03c}; mkdir -p $O
run() { tag=$1; shift
  for gr in 1 2 4; do for fu in 1 0; do
    UD_LG_GROUPS=$gr UD_LG_FUSED_BWD=$fu timeout -k 10 240 python bench.py "$@" --no-cpu-baseline 2>$O/$tag.err | tail -n 1 > $O/l.json
    python - <<PY
import json
d = json.load(open("$O/l.json"))
print("$tag groups=$gr fused_bwd=$fu", round(d["value"]), "substeps/s", d["roofline"].get("kernel_ms"))
PY
  done; done
}
run shape_rope --workload shape_rope
run rope128 --workload whip_rope --n-grid 128
run pour_water --workload pour_water
