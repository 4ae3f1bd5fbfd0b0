#!/bin/bash
# re-collect the MPM entries of profiles/pmc_traffic.json after a change to the MPM sources (one-workgroup kernels + many-workgroup step calls)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03}; A=gpurun_out/pmc_all; mkdir -p $A
W=whip_rope TAG=$TAG EXTRA=--no-graph bash tools/pmc_cloth.sh > $A/whip.log 2>&1; cp gpurun_out/pmc_cloth_whip_rope/pmc_traffic.json profiles/pmc_traffic.json; tail -n 3 $A/whip.log
TAG=$TAG bash tools/pmc_all_large.sh
