#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3c
timeout 900 python -m pytest tests/test_hip_round3.py -q -m gpu -k "frame_per_lane" > $O/tests_tile.log 2>&1; echo "tile tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests_tile.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_profile > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head -3
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print('%-70s %5s avg %8.1f us min %8.1f max %8.1f'%(r['Name'][:70],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3))
PY
