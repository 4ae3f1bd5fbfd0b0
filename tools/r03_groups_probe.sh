#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 4 6 8; do
  UD_LG_GROUPS=$g timeout -k 10 240 python bench.py --workload shape_rope --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shape_rope groups=$g', round(d['value']), d['roofline']['kernel_ms'])"
done
for g in 1; do
  UD_LG_GROUPS=$g timeout -k 10 240 python bench.py --workload whip_rope --n-grid 128 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rope128 groups=$g', round(d['value']), d['roofline']['kernel_ms'])"
done
