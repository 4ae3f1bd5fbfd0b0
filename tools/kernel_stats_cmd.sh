#!/bin/bash
# per-kernel time of any bench command line: usage (GPU box): OUT=prof_plb bash tools/kernel_stats_cmd.sh --workload torus --plb-grad --steps 3 --warmup 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUT:-prof_cmd}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT -o p -f csv -- python3 bench.py "$@" --no-cpu-baseline > $OUT/log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/summary.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_us,percent\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        f.write('"%s",%s,%.3f,%.2f,%.2f\n' % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                            float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print(open("$OUT/summary.csv").read())
PY
