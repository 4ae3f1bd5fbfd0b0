#!/bin/bash
# PlasticineLab entries of profiles/pmc_traffic.json on the current sources
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_all
bash tools/pmc_plb.sh > gpurun_out/pmc_all/plb.log 2>&1; cp gpurun_out/pmc_plb/pmc_traffic.json profiles/pmc_traffic.json; grep "GB per" gpurun_out/pmc_all/plb.log
cp profiles/pmc_traffic.json gpurun_out/pmc_all/pmc_traffic.json
