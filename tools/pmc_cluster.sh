#!/bin/bash
# rocprofv3 counter passes over the persistent cluster kernels (rope at n_grid 128, 32 envs): one counter group per pass, kernel trace only.
# usage (GPU box): [T=128] bash tools/pmc_cluster.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export UD_MPM_CLUSTER=1 UD_MPM_CLUSTER_T=${T:-128}
O=gpurun_out/pmc_cluster; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SMEM" "SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); mkdir -p $O/p$i
  timeout -k 10 200 rocprofv3 --pmc $G --kernel-trace -d $O/p$i -o p -f csv -- python3 bench.py --workload whip_rope --n-grid 128 --steps 1 --warmup 1 --no-cpu-baseline > $O/p$i/log 2>&1 || echo "pass $i ($G) failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ud::", "")
        if k.startswith("clm_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-26s %14.4g  (%d launches)" % (c, sum(acc[k][c]) / len(acc[k][c]), len(acc[k][c])))
PY
