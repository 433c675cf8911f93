#!/bin/bash
# one rocprofv3 --pmc pass: tools/pmc_one.sh <outdir> "<counters>" [bench args]
out=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass1" -- python bench.py --frames 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > "$out/pass1.log" 2>&1 || { echo "pass failed"; tail -5 "$out/pass1.log"; exit 1; }
python tools/pmc_summary.py "$out" > /dev/null
