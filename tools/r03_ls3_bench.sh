#!/bin/bash
# multi-kernel MPM forward: two launches per substep (default: grid op that retires, g2p -> p2g in one launch), three (UD_LG_FUSED_FWD=0),
# four (UD_LG_CLEAR_LAUNCH=1: lg_clear_fk in front, the state of rounds 1-2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in "--workload pour_water" "--workload pour_soup" "--workload whip_rope --n-grid 256"; do
  for m in "two" "three" "four"; do
    case $m in two) E="";; three) E="UD_LG_FUSED_FWD=0";; four) E="UD_LG_CLEAR_LAUNCH=1";; esac
    env $E timeout -k 10 300 python bench.py $w --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w launches/substep=$m', round(d['value']), '%.2f ms' % d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
