#!/bin/bash
# the MPM bench lines in one go (no CPU baseline): value and kernel ms
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in "--workload whip_rope" "--workload whip_rope --n-grid 128" "--workload whip_rope --n-grid 256" "--workload shape_rope" "--workload pour_water" "--workload pour_soup"; do
  timeout -k 10 300 python bench.py $w --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['value']), '%.2f ms' % d['ms_per_step'], d['roofline']['kernel_ms'])"
done
