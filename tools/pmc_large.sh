#!/bin/bash
# rocprofv3 counter passes over a large-path workload (one counter group per pass, kernel trace only), summarised per kernel.
# usage (GPU box): W=pour_soup bash tools/pmc_large.sh      |  W=whip_rope ARGS="--n-grid 256" NAME=whip_rope_ngrid256 bash tools/pmc_large.sh
# then: python3 tools/pmc_large_traffic.py gpurun_out/pmc_large_summary.csv $NAME   (per-step-call HBM bytes -> profiles/pmc_traffic.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${W:-pour_soup}; NAME=${NAME:-$W}
ARGS="$ARGS --tune env_groups=1"   # one launch per kernel and substep for the whole batch (env groups split it; the bytes are the same)
i=0
PASSES=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE")
[ -n "$QUICK" ] && PASSES=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE")   # QUICK=1: what pmc_traffic.json needs
rm -rf gpurun_out/pmc_[0-9]*
for G in "${PASSES[@]}"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$i; mkdir -p gpurun_out/pmc_$i
  timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace -d gpurun_out/pmc_$i -o p -f csv -- python3 bench.py --workload $W $ARGS --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$i/log 2>&1 || echo "pass $i ($G) failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ud::", "")
        if k.startswith("lg_") or k.startswith("clm_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE"]
with open("gpurun_out/pmc_large_summary.csv", "w") as o:
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        o.write(k + "," + str(n) + "," + ",".join("%.4g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in names) + "\n")
print(open("gpurun_out/pmc_large_summary.csv").read())
PY
cp gpurun_out/pmc_large_summary.csv gpurun_out/pmc_large_summary_$NAME.csv
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_$NAME.json
python3 tools/pmc_large_traffic.py gpurun_out/pmc_large_summary_$NAME.csv $NAME gpurun_out/pmc_traffic_$NAME.json
rm -rf gpurun_out/pmc_[0-9]*
