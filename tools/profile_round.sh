#!/bin/bash
# Kernel traces and counter passes of the round's workloads.  Usage (on the GPU box):
#   tools/profile_round.sh gpurun_out/prof_r2
# Writes per workload <out>/<name>/ (rocprofv3 --kernel-trace --stats of `python bench.py ...` with the bench line of the
# same run), and for the OSD workload the FETCH_SIZE / WRITE_SIZE / SQ passes on a 16-frame batch (counters never
# share a run with a trace; one --pmc pass per run).
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
trace() {   # name, bench args...
  local name=$1; shift
  mkdir -p "$out/$name"
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$name/stats" -- python bench.py "$@" --no-cpu-baseline \
      > "$out/$name/bench_under_rocprof.json" 2> "$out/$name/stats.log" || { echo "$name: stats run failed"; tail -5 "$out/$name/stats.log"; exit 1; }
  echo "$name done"
}
trace osd --steps 5 --warmup 2 --no-host-leg
trace osd_hostbuf --host-buffers --steps 5 --warmup 2
trace osd_one_frame --frames 1 --steps 200 --warmup 5 --no-host-leg --no-kernel-timing
trace m1280_s32_r7 --workload mosaic1280 --labels 32 --kernel-size 7 --steps 3 --warmup 1 --no-host-leg
trace m1280_s32_r10 --workload mosaic1280 --labels 32 --kernel-size 10 --steps 3 --warmup 1 --no-host-leg
trace m1280_s64_r15 --workload mosaic1280 --labels 64 --kernel-size 15 --steps 3 --warmup 1 --no-host-leg
trace stats --stage stats --steps 5 --warmup 2 --no-host-leg
pmc() {     # name, counters, bench args...
  local name=$1 ctrs=$2; shift 2
  mkdir -p "$out/pmc_$name"
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc_$name/pass1" -- python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-host-leg \
      > "$out/pmc_$name/pass1.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$out/pmc_$name/pass1.log"; exit 1; }
  python tools/pmc_summary.py "$out/pmc_$name" > /dev/null
  echo "pmc $name done"
}
pmc osd_fetch FETCH_SIZE --frames 16
pmc osd_write WRITE_SIZE --frames 16
pmc osd_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --frames 16
pmc osd_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" --frames 16
pmc r10_fetch FETCH_SIZE --workload mosaic1280 --labels 32 --kernel-size 10 --frames 4
pmc r10_write WRITE_SIZE --workload mosaic1280 --labels 32 --kernel-size 10 --frames 4
pmc r10_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --workload mosaic1280 --labels 32 --kernel-size 10 --frames 4
pmc r10_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" --workload mosaic1280 --labels 32 --kernel-size 10 --frames 4
pmc r15_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" --workload mosaic1280 --labels 64 --kernel-size 15 --frames 4
pmc r15_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" --workload mosaic1280 --labels 64 --kernel-size 15 --frames 4
find "$out" -name '*kernel_stats.csv' | head -8
