#!/bin/bash
# PMC passes (own runs, kernel-trace only): HBM read/write bytes per kernel for the bench workload.
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/pmc_${TAG}_$C
  rm -rf $OUT
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile > $OUT.log 2>&1 )
  ls $OUT | head
done
python scripts/dev/make_pmc_json.py $TAG
