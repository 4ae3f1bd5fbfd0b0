#!/bin/bash
# kernel stats + HBM traffic of a cloth bench workload: rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in a pass of
# their own each (MI355X_MICROARCH.md, HBM / rocprofv3 section), merged into profiles-style files under gpurun_out/.
# usage (GPU box): W=fold_tshirt TAG=r02b [EXTRA=--no-graph] bash tools/pmc_cloth.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${W:-fold_cloth1}; TAG=${TAG:-r02}; O=gpurun_out/pmc_cloth_$W; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ks -o p -f csv -- python3 bench.py --workload $W $EXTRA --steps 5 --warmup 2 --no-cpu-baseline --no-saturation > $O/${TAG}_bench_line_${W}_under_rocprof.json 2> $O/ks.err || echo "kernel-stats pass failed"
cp $O/ks/p_kernel_stats.csv $O/${TAG}_kernel_stats_$W.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $O/$C -o p -f csv -- python3 bench.py --workload $W $EXTRA --steps 2 --warmup 1 --no-cpu-baseline --no-saturation > $O/$C.log 2>&1 || echo "pass $C failed"
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $O/INSTS -o p -f csv -- python3 bench.py --workload $W $EXTRA --steps 2 --warmup 1 --no-cpu-baseline --no-saturation > $O/INSTS.log 2>&1 || echo "pass INSTS failed"
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python3 tools/pmc_summary.py $O/FETCH_SIZE/p_counter_collection.csv $O/WRITE_SIZE/p_counter_collection.csv $O/pmc_traffic.json $O/INSTS/p_counter_collection.csv | grep -i "cloth\|mpm_step"
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE $O/INSTS $O/ks
head -6 $O/${TAG}_kernel_stats_$W.csv
