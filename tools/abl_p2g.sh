#!/bin/bash
# timing-only ablation of lg_p2g (diagnostic builds in gpurun_in/, never shipped): average kernel duration per variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for A in 0 1 64 128 193; do
  if [ $A = 0 ]; then unset UNIDOM_HIP_SO; else export UNIDOM_HIP_SO=$GRAFT_REPO_ROOT/gpurun_in/lib_abl$A.so; fi
  rm -rf gpurun_out/abl_$A; mkdir -p gpurun_out/abl_$A
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/abl_$A -o p -f csv -- python3 bench.py --workload whip_rope --n-grid 128 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/abl_$A/log 2>&1
  echo "ABLATE=$A $(grep -E 'lg_p2g<4>|lg_g2p<4>' gpurun_out/abl_$A/p_kernel_stats.csv | cut -d, -f1,4 | tr '\n' ' ')"
done
