# rolling-window kernels: mosaic and dense 1280x960 bench lines at r = 10 and 15 -> gpurun_out/roll_ab/
set -e
mkdir -p gpurun_out/roll_ab
for r in 10 15; do
L=32; [ $r = 15 ] && L=64
timeout -k 10 170 python bench.py --workload mosaic1280 --labels $L --kernel-size $r --steps 6 --warmup 2 --no-host-leg --no-cpu-baseline --no-stats-leg > gpurun_out/roll_ab/mosaic_r$r.json
timeout -k 10 170 python bench.py --workload dense1280 --frames 16 --kernel-size $r --steps 6 --warmup 2 --no-host-leg --no-cpu-baseline --no-stats-leg > gpurun_out/roll_ab/dense_r$r.json
done
grep -o '"avg_us": [0-9.]*' gpurun_out/roll_ab/*.json
