#!/bin/bash
# Kernel trace + HBM traffic counters of the bench workload.  Usage (on the GPU box):
#   tools/profile_round.sh gpurun_out/prof_r1
# Writes <out>/stats (rocprofv3 --kernel-trace --stats of `python bench.py`), <out>/fetch and
# <out>/write (one --pmc pass each on a 16-frame batch; counters never share a run with a trace).
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/stats.log" || { echo "stats run failed"; tail -5 "$out/stats.log"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p "$out/$c"
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out/$c/pass1" -- python bench.py --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$out/$c/pass1.log" 2>&1 || { echo "$c pass failed"; tail -5 "$out/$c/pass1.log"; exit 1; }
  python tools/pmc_summary.py "$out/$c" > /dev/null
done
find "$out" -name '*kernel_stats.csv' | head -3
