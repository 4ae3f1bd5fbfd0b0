#!/bin/bash
# lanes per particle on the multi-kernel path at the four-lane sizes: 4 (default) against 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in "--workload whip_rope --n-grid 128" "--workload shape_rope" "--workload pour_water"; do
  for l in 4 1; do
    UD_MPM_CLUSTER=0 UD_LG_LANES=$l timeout -k 10 240 python bench.py $w --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w lanes=$l', round(d['value']), d['roofline']['kernel_ms'])"
  done
done
