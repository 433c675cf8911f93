#!/bin/bash
# An experiment build of the library with other compile-time switches for the r = 7 CRF kernel (timing experiments only):
#   tools/build_variant.sh <name> -DFS_SKIPMAIN=1 ...   ->  imagesegmentation_amd/libfs_hip_<name>.so
# Select it at run time with FS_HIP_LIB=<path>.  The objects of the other files are the product build's.
set -e
name=$1; shift
cd "$(dirname "$0")/../imagesegmentation_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -c k_crf_fused_r7.hip -o /tmp/k_crf_fused_r7_$name.o
objs=$(ls *.o | grep -v cwalk | grep -v k_crf_fused_r7.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfs_hip_$name.so $objs /tmp/k_crf_fused_r7_$name.o
echo ../libfs_hip_$name.so
