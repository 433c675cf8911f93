#!/bin/bash
# One frame per call with the early launch of the updates switched on and off (run on the GPU box).
set -e
out=gpurun_out/one_frame_ab
mkdir -p $out
B="python bench.py --frames 1 --steps 400 --warmup 20 --no-host-leg --no-cpu-baseline --no-kernel-timing"
timeout -k 10 120 $B > $out/spec.json
FS_NO_SPECULATION=1 timeout -k 10 120 $B > $out/nospec.json
for f in spec nospec; do python - $out/$f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "ms/call", round(d["ms_per_step"], 4))
PY
done
