#!/bin/bash
# Every counter pass profiles/pmc_traffic.json is built from, on the final tree (its entries carry the hash of the kernel sources):
# cloth (one workgroup per env; several), the one-workgroup MPM kernels, the many-workgroup MPM step calls, the PlasticineLab path.
# usage (GPU box): TAG=r05 [PART=1|2|3] bash tools/pmc_all.sh  ->  gpurun_out/pmc_all/{pmc_traffic.json, <TAG>_pmc_large_*.csv, <TAG>_kernel_stats_*}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r05}; A=gpurun_out/pmc_all; mkdir -p $A
# PART (gpurun caps a call at 20 minutes; copy gpurun_out/pmc_all/pmc_traffic.json to profiles/ between the calls): 1 = cloth + one-workgroup MPM +
# shape_rope + rope at n_grid 128, 2 = the other many-workgroup workloads, 3 = PlasticineLab; unset = everything
want() { [ -z "$PART" ] || [ "$PART" = "$1" ]; }
keep() { cp $1 profiles/pmc_traffic.json; }     # the box's copy of the repo is scratch: the next tool starts from the merged file
large() { name=$1; w=$2; shift 2
  QUICK=1 W=$w NAME=$name ARGS="$*" bash tools/pmc_large.sh > $A/large_$name.log 2>&1; keep gpurun_out/pmc_traffic_$name.json
  cp gpurun_out/pmc_large_summary_$name.csv $A/${TAG}_pmc_large_$name.csv; tail -n 2 $A/large_$name.log; }
if want 1; then
W=fold_cloth1 TAG=$TAG bash tools/pmc_cloth.sh > $A/cloth.log 2>&1; keep gpurun_out/pmc_cloth_fold_cloth1/pmc_traffic.json; tail -n 3 $A/cloth.log
W=fold_tshirt TAG=$TAG bash tools/pmc_cloth.sh > $A/tshirt.log 2>&1; keep gpurun_out/pmc_cloth_fold_tshirt/pmc_traffic.json; tail -n 3 $A/tshirt.log
W=whip_rope TAG=$TAG EXTRA=--no-graph bash tools/pmc_cloth.sh > $A/whip.log 2>&1; keep gpurun_out/pmc_cloth_whip_rope/pmc_traffic.json; tail -n 3 $A/whip.log
large shape_rope shape_rope
large whip_rope_ngrid128 whip_rope --n-grid 128
fi
if want 2; then
large whip_rope_ngrid256 whip_rope --n-grid 256
large pour_water pour_water
large pour_soup pour_soup
fi
if want 3; then
bash tools/pmc_plb.sh > $A/plb.log 2>&1; keep gpurun_out/pmc_plb/pmc_traffic.json; grep "GB per" $A/plb.log
fi
cp profiles/pmc_traffic.json $A/pmc_traffic.json
