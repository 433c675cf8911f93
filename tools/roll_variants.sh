#!/bin/bash
# Times builds of the library (tools/build_variant_files.sh) on the 1280x960 workloads, side by side in one call:
#   VARIANTS="base rollc" bash tools/roll_variants.sh            ("base" = the product build)
mkdir -p gpurun_out/roll_var
for v in ${VARIANTS:-base}; do
  s=_$v; [ "$v" = base ] && s=
  lib=$PWD/imagesegmentation_amd/libfs_hip$s.so
  for cfg in "mosaic1280 7 32" "mosaic1280 10 32" "mosaic1280 15 64" "dense1280 7 0" "dense1280 10 0" "dense1280 15 0"; do
    set -- $cfg
    lab=""; [ $1 = mosaic1280 ] && lab="--labels $3"
    FS_HIP_LIB=$lib timeout -k 10 170 python bench.py --workload $1 $lab --kernel-size $2 --frames 16 --steps 5 --warmup 2 --no-host-leg --no-cpu-baseline --no-stats-leg \
        > gpurun_out/roll_var/${v}_$1_r$2.json 2> gpurun_out/roll_var/${v}_$1_r$2.err || { echo "$v $cfg failed"; tail -3 gpurun_out/roll_var/${v}_$1_r$2.err; continue; }
    python -c "
import json
d=json.load(open('gpurun_out/roll_var/${v}_$1_r$2.json'));print('variant $v $1 r=$2', round(d['value']),'images/s  avg_us',round(d['roofline']['avg_us'],1))"
  done
done
