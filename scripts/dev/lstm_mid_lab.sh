#!/bin/bash
# Dev: how many k-steps of fragments should lstm_mid_x3_kernel keep in flight?  Variants of the library with other ring
# depths (LM3_RING1: launches of at most 32 rows, LM3_RING2: 33..64 rows per workgroup), timed on the stand-alone LSTM
# (2 x 512, input 60; scripts/dev/bench_lstm_mid.py).
#   8_6+LM3_LAB_NOK   one k-step per wave: what a launch costs without its K loop (results are wrong)
#   8_6+LM3_LAB_TWICE_WGS   twice the workgroups, each with every other half of its waves' k-steps (results are wrong):
#                           what a 16-column tile on all 256 CUs would gain in the K loop
# usage (container): bash scripts/dev/lstm_mid_lab.sh build ; (GPU box): bash scripts/dev/lstm_mid_lab.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin
V=${VARIANTS:-3_3 5_4 8_6 12_8 16_10}
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for v in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DLM3_RING1=${v%%_*} -DLM3_RING2=$(echo ${v#*_} | sed 's/+.*//') $(echo $v | grep -o '+.*' | sed 's/+/ -D/g') -c $C/lstm_mid_x3.hip -o /tmp/lstm_mid_$v.o 2>/dev/null || exit 1
    objs=$(ls $C/*.o | grep -v lstm_mid_x3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_RING_$v.so $objs /tmp/lstm_mid_$v.o || exit 1
    echo built $v
  done
else
  for v in $V; do
    echo "== ring depth $v (<= 32 rows _ 33..64 rows)"; EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_RING_$v.so python $R/scripts/dev/bench_lstm_mid.py 2>&1 | grep "B=" | head -${ROWS:-5}
  done
fi
