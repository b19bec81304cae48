#!/bin/bash
# kernel statistics of the training step at the reference's batch (12 windows)
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r3m
OUT=$R/gpurun_out/r3m/prof
rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/scripts/train.py --steps 10 --bs_train 12 --json > $OUT.log 2>&1 )
cp $OUT/t_kernel_stats.csv gpurun_out/r3m/train_kernel_stats_bs12.csv
rm -rf $OUT
tail -1 gpurun_out/r3m/prof.log | cut -c1-300
