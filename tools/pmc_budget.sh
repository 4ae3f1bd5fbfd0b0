#!/bin/bash
# Instruction budget of a many-workgroup MPM workload, per kernel: instruction mix (VALU / SALU / LDS / VMEM / SMEM
# wave-instructions) and where the wave cycles go (active / waiting on an instruction's issue / parked at s_waitcnt or a barrier), one
# counter group per rocprofv3 pass, kernel trace only.
# usage (GPU box): W=whip_rope ARGS="--n-grid 256" NAME=whip_rope_ngrid256 bash tools/pmc_budget.sh  -> gpurun_out/pmc_budget_$NAME.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${W:-whip_rope}; NAME=${NAME:-$W}
ARGS="$ARGS --tune env_groups=1"
PASSES=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM")
rm -rf gpurun_out/pmcb_[0-9]*
i=0
for G in "${PASSES[@]}"; do
  i=$((i+1)); mkdir -p gpurun_out/pmcb_$i
  timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace -d gpurun_out/pmcb_$i -o p -f csv -- python3 bench.py --workload $W $ARGS --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcb_$i/log 2>&1 || echo "pass $i ($G) failed: $(tail -n 2 gpurun_out/pmcb_$i/log | head -c 300)"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = []
for f in sorted(glob.glob("gpurun_out/pmcb_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ud::", "")
        if k.startswith("lg_") or k.startswith("clm_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] not in names:
                names.append(r["Counter_Name"])
with open("gpurun_out/pmc_budget_$NAME.csv", "w") as o:
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        o.write(k + "," + str(n) + "," + ",".join("%.5g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in names) + "\n")
print(open("gpurun_out/pmc_budget_$NAME.csv").read())
PY
rm -rf gpurun_out/pmcb_[0-9]*
