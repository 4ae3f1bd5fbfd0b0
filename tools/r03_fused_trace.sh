#!/bin/bash
# kernel trace of the rope at n_grid 128 with the two-launch backward and with the four-kernel one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r03c}; mkdir -p $O
for fu in 1 0; do
  export UD_LG_FUSED_BWD=$fu
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_fused$fu -o t -f csv -- python3 bench.py --workload whip_rope --n-grid 128 --steps 10 --warmup 3 --no-cpu-baseline > $O/trace_fused$fu.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = glob.glob("$O/trace_fused$fu/**/*kernel_stats.csv", recursive=True)[0]
print("fused=$fu")
for r in list(csv.DictReader(open(f)))[:9]:
    print("  %-60s %6s %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  python3 tools/trace_timeline.py $O/trace_fused$fu
done
