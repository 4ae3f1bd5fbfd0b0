#!/bin/bash
# diagnostic builds for tools/abl_p2g.sh: libunidom_hip.so with -DUD_MPM_ABLATE=<bits> in mpm_large.hip, into gpurun_in/ (git-ignored,
# travels with gpurun).  Timing only -- the ablated kernels compute wrong results.   usage: bash tools/build_abl.sh 1 64 128 193
set -e
cd "$(dirname "$0")/../unidom_amd/csrc"
mkdir -p ../../gpurun_in build/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt"
for A in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DUD_MPM_ABLATE=$A -c mpm_large.hip -o build/abl/mpm_large_$A.o &&
    /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../gpurun_in/lib_abl$A.so $(ls build/*.o | grep -v "build/mpm_large.o") build/abl/mpm_large_$A.o && echo built $A ) &
done
wait
