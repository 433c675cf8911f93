# per-kernel times of the dense 1280x960 bench at radius $1 (rocprofv3 kernel trace) -> gpurun_out/dense_prof_r$1/
set -e
cd /tmp && export TMPDIR=/tmp
R=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/dense_prof_r$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 bench.py --workload dense1280 --kernel-size $R --frames 16 --steps 3 --warmup 1 --no-host-leg --no-cpu-baseline --no-stats-leg --no-kernel-timing > $OUT/bench.json
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:8]:
    print(row["Name"][:110], row["Calls"], row["TotalDurationNs"], row["AverageNs"])
PY
