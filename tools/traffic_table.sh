#!/bin/bash
# counter passes of one many-workgroup workload (tools/pmc_large.sh) + the per-kernel byte accounting of its backward (tools/traffic_table.py)
# usage (GPU box): W=shape_rope bash tools/traffic_table.sh   -> gpurun_out/traffic_table_<W>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${W:-shape_rope}
QUICK=1 W=$W NAME=$W bash tools/pmc_large.sh > gpurun_out/pmc_large_$W.log 2>&1; tail -n 3 gpurun_out/pmc_large_$W.log
timeout -k 10 300 python tools/traffic_table.py $W gpurun_out/pmc_large_summary_$W.csv > gpurun_out/traffic_table_$W.txt 2> gpurun_out/traffic_table_$W.err || tail -5 gpurun_out/traffic_table_$W.err
cat gpurun_out/traffic_table_$W.txt
