# dense 1280x960 frames at r = 7, 10, 15: bench lines into gpurun_out/dense_frames/ (FRAMES frames per step, default 16)
set -e
mkdir -p gpurun_out/dense_frames
for r in ${RADII:-7 10 15}; do for f in ${FRAMES:-16}; do
timeout -k 10 170 python bench.py --workload dense1280 --kernel-size $r --frames $f --steps 6 --warmup 2 --no-host-leg --no-cpu-baseline --no-stats-leg > gpurun_out/dense_frames/r${r}_f${f}.json
done; done
grep -o '"ms_per_step": [0-9.]*\|"avg_us": [0-9.]*' gpurun_out/dense_frames/*.json
