# two SQ counter passes of the dense 1280x960 bench (4 frames) at radius $1 -> gpurun_out/pmc_dense_r$1/summary.txt
R=$1
out=gpurun_out/pmc_dense_r$R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_VALU"
timeout -s KILL 150 rocprofv3 --pmc $A --output-format csv -d $out/pass1 -- python3 bench.py --workload dense1280 --kernel-size $R --frames 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-host-leg --no-stats-leg > $out/pass1.log 2>&1 || { tail -5 $out/pass1.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc $B --output-format csv -d $out/pass2 -- python3 bench.py --workload dense1280 --kernel-size $R --frames 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-host-leg --no-stats-leg > $out/pass2.log 2>&1 || { tail -5 $out/pass2.log; exit 1; }
python3 tools/pmc_summary.py $out | grep -A 17 crf_iter_roll
