#!/bin/bash
# Dev: SQ instruction-mix / LDS counters of the SMPL evaluation kernels alone (scripts/dev/bench_chain.py), two passes.
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
for PASS in 1 2; do
  OUT=$R/gpurun_out/chainpmc_$PASS
  rm -rf $OUT; mkdir -p $OUT
  if [ $PASS = 1 ]; then CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR";
  else CNT="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"; fi
  ( cd /tmp && timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o pmc -- env PYTHONPATH=$R python $R/scripts/dev/bench_chain.py > $OUT.log 2>&1 )
  tail -2 $OUT.log
  python - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no csv'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:50]
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    if row['Dispatch_Id'] not in seen:
        seen.add(row['Dispatch_Id']); n[k] += 1
for k in acc:
    print(k, n[k], {a: '%.4g' % (v / n[k]) for a, v in sorted(acc[k].items())})
PY
done
