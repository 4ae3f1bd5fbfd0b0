#!/bin/bash
# timing-only ablation of the persistent cluster forward kernel (diagnostic builds in gpurun_in/, tools/build_abl.sh 4096 8192 16384 28672 61440 61441):
# UD_MPM_ABLATE bits 4096 no read-back (own LDS sums instead), 8192 no zeroing, 16384 no flush atomics, 32768 no inter-part barrier, 1 no SVD
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for T in 128 64; do for A in ${ABL:-0 4096 8192 16384 28672 61440 61441}; do
  if [ $A = 0 ]; then unset UNIDOM_HIP_SO; else export UNIDOM_HIP_SO=$GRAFT_REPO_ROOT/gpurun_in/lib_abl$A.so; fi
  echo "T=$T ABLATE=$A $(UD_MPM_CLUSTER=1 UD_MPM_CLUSTER_T=$T timeout -k 10 120 python bench.py --workload whip_rope --n-grid 128 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -n 1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done; done
