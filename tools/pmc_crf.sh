#!/bin/bash
# SQ counter passes for the CRF update kernel on a 16-frame batch: tools/pmc_crf.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python bench.py --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done
python tools/pmc_summary.py "$out" | grep -A 20 "crf_iter_fused"
