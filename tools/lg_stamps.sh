#!/bin/bash
# diagnostic build of libunidom_hip.so with s_memtime phase stamps in lg_p2g / lg_g2p_adj / lg_p2g_adj (-DUD_LG_STAMPS), into gpurun_in/
# (git-ignored, travels with gpurun).  Then on the GPU box:  UNIDOM_HIP_SO=$PWD/gpurun_in/lib_stamps.so python tools/lg_stamps.py pour_soup
set -e
cd "$(dirname "$0")/../unidom_amd/csrc"
make -s libunidom_hip.so
mkdir -p ../../gpurun_in build/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt"
/opt/rocm/bin/hipcc $FLAGS -DUD_LG_STAMPS -c mpm_large.hip -o build/abl/mpm_large_stamps.o
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../gpurun_in/lib_stamps.so $(ls build/*.o | grep -v "build/mpm_large.o") build/abl/mpm_large_stamps.o
echo built gpurun_in/lib_stamps.so
