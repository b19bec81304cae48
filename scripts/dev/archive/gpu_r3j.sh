#!/bin/bash
# kernel stats of the training step at 256 windows with the fused train-mode layer forced on
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r3j
OUT=$R/gpurun_out/r3j/prof
rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/scripts/train.py --steps 10 --bs_train 256 --json --option train_fused=2 > $OUT.log 2>&1 )
cp $OUT/t_kernel_stats.csv gpurun_out/r3j/train_fused_kernel_stats_bs256.csv
rm -rf $OUT
tail -1 gpurun_out/r3j/prof.log | cut -c1-300
