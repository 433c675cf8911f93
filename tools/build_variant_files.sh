#!/bin/bash
# An experiment build of the library in which the named source files are compiled with other switches:
#   tools/build_variant_files.sh <name> "k_crf_roll_u.hip k_crf_roll_m.hip" -DFS_ASM_STEP=0   ->  imagesegmentation_amd/libfs_hip_<name>.so
# (tools/build_variant.sh is the same for the r = 7 instance file alone).  Select it at run time with FS_HIP_LIB=<path>.
set -e
name=$1; files=$2; shift 2
cd "$(dirname "$0")/../imagesegmentation_amd/csrc"
make -j8 > /dev/null
objs=""
skip=""
for f in $files; do
  o=/tmp/${f%.hip}_$name.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -c $f -o $o &
  objs="$objs $o"; skip="$skip ${f%.hip}.o"
done
wait
rest=$(ls *.o | grep -v cwalk | grep -v -x -F "$(echo $skip | tr ' ' '\n')")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfs_hip_$name.so $rest $objs
echo ../libfs_hip_$name.so
