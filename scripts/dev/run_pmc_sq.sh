#!/bin/bash
# SQ / GRBM counter pass (own run, kernel-trace only) over the bench workload: effective clock (GRBM_GUI_ACTIVE / wall),
# wave-cycle buckets and matrix-core busy cycles per kernel.  BENCH_ARGS="--workload vertices" selects another workload,
# PMC_CMD="python $PWD/scripts/train.py --steps 3 --no_graph --json" another command (the training step).
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
OUT=$R/gpurun_out/pmc_sq_$TAG
rm -rf $OUT
( cd /tmp && timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $OUT -o pmc -- ${PMC_CMD:-python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile --no_secondary --no_traffic ${BENCH_ARGS:-}} > $OUT.log 2>&1 )
tail -3 $OUT.log
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter csv', glob.glob(out + '/**/*', recursive=True)[:10]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:70]
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    did = row['Dispatch_Id']
    if did not in seen:
        seen.add(did); n[k] += 1
        if 'Start_Timestamp' in row and row.get('End_Timestamp'):
            dur[k] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
names = sorted({c for k in acc for c in acc[k]})
print('kernel,launches,avg_us,' + ','.join(names))
for k in sorted(acc, key=lambda k: -acc[k].get('GRBM_GUI_ACTIVE', 0))[:12]:
    print(k, n[k], '%.1f' % (dur[k] / max(n[k], 1) / 1e3), *['%.4g' % (acc[k][c] / n[k]) for c in names], sep=',')
PY
