#!/bin/bash
# SQ instruction counters of the CRF update for an experiment build of the library:
#   tools/pmc_variant.sh <library.so> <outdir>
lib=$1; out=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
export FS_HIP_LIB=$lib
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$out/pass1" -- python bench.py --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$out/pass1.log" 2>&1 || { echo "pass failed"; tail -5 "$out/pass1.log"; exit 1; }
python tools/pmc_summary.py "$out" | grep -A 9 "crf_iter_fused"
