#!/bin/bash
# per-kernel time of a bench workload (rocprofv3 kernel trace): usage (GPU box): W=pour_soup STEPS=3 bash tools/kernel_stats.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${W:-pour_soup}; STEPS=${STEPS:-3}
rm -rf gpurun_out/prof_ks && mkdir -p gpurun_out/prof_ks
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ks -o p -f csv -- python3 bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline > gpurun_out/prof_ks/log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_ks/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/prof_ks/summary.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_us,percent\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        f.write('"%s",%s,%.3f,%.2f,%.2f\n' % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                            float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print(open("gpurun_out/prof_ks/summary.csv").read())
PY
