#!/bin/bash
# Dev: where does a step of lstm_rows_x3_kernel spend its time?  Variants of the library with parts of the K loop compiled
# out (results are then wrong; only the timing matters), timed on the LSTM alone (B = 1024, F = 32, 2 x 512).
#   LR_LAB_NOBARRIER  no hand-over barrier per k-step
#   LR_LAB_NOGLOAD    no global loads in the loop (the prologue's operands are multiplied again and again)
#   LR_LAB_NOLDS      no LDS traffic in the loop
# usage (container): bash scripts/dev/lstm_rows_lab.sh build ; (GPU box): bash scripts/dev/lstm_rows_lab.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin
V=${VARIANTS:-BASE LR_LAB_NOBARRIER LR_LAB_NOGLOAD LR_LAB_NOLDS LR_LAB_NOGLOAD+LR_LAB_NOLDS LR_LAB_NOGLOAD+LR_LAB_NOLDS+LR_LAB_NOBARRIER}
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for v in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $(echo $v | sed "s/+/ -D/g; s/^/-D/; s/_EQ_/=/g") -c $C/lstm_rows_x3.hip -o /tmp/lstm_rows_$v.o 2>/dev/null || exit 1
    objs=$(ls $C/*.o | grep -v lstm_rows_x3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_$v.so $objs /tmp/lstm_rows_$v.o || exit 1
    echo built $v
  done
else
  for v in $V; do
    echo "== $v"; EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_$v.so python $R/scripts/dev/bench_lstm.py 2>&1 | grep "LSTM B="
  done
  echo "== lstm_x3 = 1 (K-split kernel)"; EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_BASE.so EMPOSE_LSTM_X3=1 python $R/scripts/dev/bench_lstm.py 2>&1 | grep "LSTM B="
fi
