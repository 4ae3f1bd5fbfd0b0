#!/bin/bash
# the many-workgroup part of tools/pmc_all.sh alone (starts from profiles/pmc_traffic.json)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03}; A=gpurun_out/pmc_all; mkdir -p $A
keep() { cp $1 profiles/pmc_traffic.json; }
large() { name=$1; w=$2; shift 2
  QUICK=1 W=$w NAME=$name ARGS="$*" bash tools/pmc_large.sh > $A/large_$name.log 2>&1; keep gpurun_out/pmc_traffic_$name.json
  cp gpurun_out/pmc_large_summary_$name.csv $A/${TAG}_pmc_large_$name.csv; tail -n 2 $A/large_$name.log; }
large shape_rope shape_rope
large whip_rope_ngrid128 whip_rope --n-grid 128
large whip_rope_ngrid256 whip_rope --n-grid 256
large pour_water pour_water
large pour_soup pour_soup
cp profiles/pmc_traffic.json $A/pmc_traffic.json
