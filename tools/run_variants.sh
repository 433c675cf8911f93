#!/bin/bash
# Times experiment builds of the library (tools/build_variant.sh) with the headline workload, HBM-resident:
#   VARIANTS="base skipq ..." bash tools/run_variants.sh      ("base" = the product build)
for v in ${VARIANTS:-base}; do
  s=_$v; [ "$v" = base ] && s=
  lib=imagesegmentation_amd/libfs_hip$s.so
  FS_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-host-leg --no-cpu-baseline --steps 20 > gpurun_out/r3_var_$v.json 2> gpurun_out/r3_var_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/r3_var_$v.err; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r3_var_$v.json'));print('variant $v',round(d['value']),'images/s  avg_us',round(d['roofline']['avg_us'],2),'ms/step',round(d['ms_per_step'],2))"
done
