#!/bin/bash
# diagnostic build of libunidom_hip.so with s_memtime phase stamps in pcl_fwd_kernel / pcl_bwd_kernel (-DUD_PCL_STAMPS), into gpurun_in/
# (git-ignored, travels with gpurun).  Then on the GPU box:  UNIDOM_HIP_SO=$PWD/gpurun_in/lib_pcl_stamps.so python tools/pcl_stamps.py [128]
set -e
cd "$(dirname "$0")/../unidom_amd/csrc"
make -s libunidom_hip.so
mkdir -p ../../gpurun_in build/abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DUD_PCL_STAMPS $PCL_EXTRA -c plb_cluster.hip -o build/abl/plb_cluster_stamps.o
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../gpurun_in/lib_pcl_stamps.so $(ls build/*.o | grep -v "build/plb_cluster.o") build/abl/plb_cluster_stamps.o
echo built gpurun_in/lib_pcl_stamps.so
