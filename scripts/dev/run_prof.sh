#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/dev/run_prof.sh <tag>
# Runs tests, the bench, and a rocprofv3 kernel-trace of the bench; leaves summaries in gpurun_out/.
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/gpu_tests_$TAG.log
timeout 900 python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lgd -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no_cpu_baseline --no_profile > $OUT.log 2>&1 )
find $OUT -name "*kernel_stats*" | head -3
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/kernel_stats_$TAG.csv
ls -la gpurun_out | tail -12
