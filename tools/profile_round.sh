#!/bin/bash
# Kernel traces and counter passes of the round's workloads.  Usage (on the GPU box):
#   tools/profile_round.sh gpurun_out/prof_r3 [traces|pmc|all]
# Writes per workload <out>/<name>/ (rocprofv3 --kernel-trace --stats of `python bench.py ...` with the bench line of the
# same run; ONLY="name ..." in the environment restricts the run to those workloads), and per CRF workload the FETCH_SIZE / WRITE_SIZE / SQ / GRBM passes (counters never share a run with a
# trace; one --pmc pass per run).  tools/collect_profiles.py copies the summaries into profiles/.
out=$1
what=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
trace() {   # name, bench args...
  local name=$1; shift
  [[ -n "$ONLY" && " $ONLY " != *" $name "* ]] && return
  mkdir -p "$out/$name"
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$name/stats" -- python bench.py "$@" --no-cpu-baseline --no-stats-leg \
      > "$out/$name/bench_under_rocprof.json" 2> "$out/$name/stats.log" || { echo "$name: stats run failed"; tail -5 "$out/$name/stats.log"; exit 1; }
  echo "$name done"
}
pmc() {     # name, counters, bench args...
  local name=$1 ctrs=$2; shift 2
  mkdir -p "$out/pmc_$name"
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc_$name/pass1" -- python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-host-leg --no-stats-leg \
      > "$out/pmc_$name/pass1.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$out/pmc_$name/pass1.log"; exit 1; }
  python tools/pmc_summary.py "$out/pmc_$name" > /dev/null
  echo "pmc $name done"
}
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
passes() {  # prefix, bench args...: the four counter passes of one workload
  local pre=$1; shift
  [[ -n "$ONLY" && " $ONLY " != *" $pre "* ]] && return
  pmc ${pre}_fetch FETCH_SIZE "$@"
  pmc ${pre}_write WRITE_SIZE "$@"
  pmc ${pre}_sq1 "$SQ" "$@"
  pmc ${pre}_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" "$@"
}
if [ "$what" != pmc ]; then
  trace osd --steps 5 --warmup 2 --no-host-leg
  trace osd_hostbuf --host-buffers --steps 5 --warmup 2
  trace osd_one_frame --frames 1 --steps 200 --warmup 5 --no-host-leg --no-kernel-timing
  trace m1280_s32_r7 --workload mosaic1280 --labels 32 --kernel-size 7 --steps 3 --warmup 1 --no-host-leg
  trace m1280_s32_r10 --workload mosaic1280 --labels 32 --kernel-size 10 --steps 3 --warmup 1 --no-host-leg
  trace m1280_s64_r15 --workload mosaic1280 --labels 64 --kernel-size 15 --steps 3 --warmup 1 --no-host-leg
  trace dense1280_r7 --workload dense1280 --kernel-size 7 --steps 3 --warmup 1 --no-host-leg
  trace dense1280_r10 --workload dense1280 --kernel-size 10 --steps 3 --warmup 1 --no-host-leg
  trace dense1280_r15 --workload dense1280 --kernel-size 15 --steps 3 --warmup 1 --no-host-leg
  trace stats --stage stats --steps 5 --warmup 2 --no-host-leg
fi
if [ "$what" != traces ]; then
  passes osd --frames 16
  passes m7 --workload mosaic1280 --labels 32 --kernel-size 7 --frames 4
  passes m10 --workload mosaic1280 --labels 32 --kernel-size 10 --frames 4
  passes m15 --workload mosaic1280 --labels 64 --kernel-size 15 --frames 4
  passes d7 --workload dense1280 --kernel-size 7 --frames 4
  passes d10 --workload dense1280 --kernel-size 10 --frames 4
  passes d15 --workload dense1280 --kernel-size 15 --frames 4
fi
find "$out" -name '*kernel_stats.csv' | head -12
