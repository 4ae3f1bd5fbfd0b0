#!/bin/bash
# kernel launches and GPU-busy time per APG update of a bench workload (rocprofv3 kernel trace)
# usage (GPU box): W="whip_rope" bash tools/count_launches.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${W:-whip_rope}
rm -rf gpurun_out/prof_cnt && mkdir -p gpurun_out/prof_cnt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cnt -o p -f csv -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_cnt/log 2>&1
tail -n 1 gpurun_out/prof_cnt/log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms_per_step %.3f' % d['ms_per_step'])"
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_cnt/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
ud=sum(float(r["TotalDurationNs"]) for r in rows if "ud::" in r["Name"])
print("per update (25): %.2f ms GPU busy (%.2f ms in ud:: kernels), %d launches" % (tot/1e6/25, ud/1e6/25, calls/25))
PY
