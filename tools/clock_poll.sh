#!/bin/bash
# Shader and memory clocks under the headline workload: rocm-smi sampled four times a second during 400 steps of the bench
# (run on the GPU box) -> gpurun_out/clkpoll.txt, gpurun_out/clk_bench.json
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/clkpoll.txt &
P=$!
timeout -k 10 120 python bench.py --no-host-leg --no-cpu-baseline --steps 400 > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err
wait $P
python - <<'PY'
import json
d=json.load(open('gpurun_out/clk_bench.json')); print('images/s',round(d['value']),'avg_us',round(d['roofline']['avg_us'],2))
PY
