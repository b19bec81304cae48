#!/bin/bash
# rocprofv3 kernel stats of the training step: prof_train.sh <tag> <train.py args...>  -> gpurun_out/<dir>/train_stats_<tag>.csv
tag=$1; shift
out=${PROF_OUT:-gpurun_out/r04c}
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o t --output-format csv -- python $GRAFT_REPO_ROOT/scripts/train.py --steps 8 --warmup 4 --json "$@" > /tmp/prof_$tag.log 2>&1
cp /tmp/prof_$tag/t_kernel_stats.csv $GRAFT_REPO_ROOT/$out/train_stats_$tag.csv
tail -1 /tmp/prof_$tag.log
