#!/bin/bash
# SQ counters of one lab variant: lab_pmc.sh <shape> <variant>
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/labpmc_$1_$2
rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d $OUT -o pmc -- /tmp/gemm_lab $1 $2 > $OUT.log 2>&1 )
python - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:60]
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    if row['Dispatch_Id'] not in seen:
        seen.add(row['Dispatch_Id']); n[k] += 1
for k in acc:
    if 'gemm' not in k: continue
    c = {a: v / n[k] for a, v in acc[k].items()}
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    print(k, 'launches', n[k])
    print('  cycles/launch %.0f  mfma_util %.3f  waves/simd %.2f' % (cyc, c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), c['SQ_WAVE_CYCLES'] * 4 / (cyc * 1024)))
    w = c['SQ_WAVE_CYCLES']
    print('  wave-cycle split: wait_any %.3f  wait_inst_any %.3f  active %.3f  (wait_inst_lds %.3f)  lds_conflict/cyc %.4f' % (
        c['SQ_WAIT_ANY'] / w, c['SQ_WAIT_INST_ANY'] / w, c['SQ_ACTIVE_INST_ANY'] / w, c['SQ_WAIT_INST_LDS'] / w, c['SQ_LDS_BANK_CONFLICT'] / (cyc * 256)))
PY
