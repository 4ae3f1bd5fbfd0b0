#!/bin/bash
# timing-only ablation of lg_p2g (diagnostic builds in gpurun_in/, never shipped): kernel totals per variant.
# usage (GPU box): NG=128|256 bash tools/abl_p2g.sh  |  W=pour_soup bash tools/abl_p2g.sh     (builds: bash tools/build_abl.sh 1 64 128 193)
# ABLATE bits: 1 no SVD, 64 one stencil cell instead of 27, 128 no flush, 256 no LDS atomics in the walk, 512 no slot probing   (ABL="0 256 512" selects)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NG=${NG:-128}
W=${W:-whip_rope}
if [ $W = whip_rope ]; then WARGS="--workload whip_rope --n-grid $NG"; else WARGS="--workload $W"; fi
for A in ${ABL:-0 1 64 128 193}; do
  if [ $A = 0 ]; then unset UNIDOM_HIP_SO; else export UNIDOM_HIP_SO=$GRAFT_REPO_ROOT/gpurun_in/lib_abl$A.so; fi
  rm -rf gpurun_out/abl_$A; mkdir -p gpurun_out/abl_$A
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/abl_$A -o p -f csv -- python3 bench.py $WARGS --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/abl_$A/log 2>&1
  echo "$WARGS ABLATE=$A $(python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/abl_$A/p_kernel_stats.csv")):
    if "lg_p2g<" in r["Name"] or "lg_g2p<" in r["Name"]:
        print(r["Name"].split("(")[0].replace("void ud::",""), "avg_us=%.1f" % (float(r["AverageNs"])/1e3), end="  ")
PY
)"
done
