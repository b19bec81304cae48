#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3g
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/scripts/train.py --steps 10 --bs_train ${BS:-256} --json > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - "$O/prof/t_kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:26]:
    print('%-78s %5s tot %8.1f us avg %8.2f %5.1f%%'%(r['Name'][:78],r['Calls'],float(r['TotalDurationNs'])/1e3,float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
print('total ms', tot/1e6)
PY
