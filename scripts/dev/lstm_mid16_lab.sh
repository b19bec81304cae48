#!/bin/bash
# Dev: ring depths of lstm_mid16_x3_kernel (LH3_RING2: launches of at most 32 rows, LH3_RING4: 33..64 rows), timed on the
# stand-alone LSTM (2 x 512, input 60; scripts/dev/bench_lstm_mid.py).  Variants <ring 2>_<ring 4>.
# usage (container): bash scripts/dev/lstm_mid_lab.sh build ; (GPU box): bash scripts/dev/lstm_mid_lab.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin
V=${VARIANTS:-4_3 6_4 8_4 8_6 12_6}
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for v in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DLH3_RING2=${v%%_*} -DLH3_RING4=$(echo ${v#*_} | sed 's/+.*//') $(echo $v | grep -o '+.*' | sed 's/+/ -D/g') -c $C/lstm_mid16_x3.hip -o /tmp/lstm_mid16_$v.o 2>/dev/null || exit 1
    objs=$(ls $C/*.o | grep -v lstm_mid16_x3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_RING16_$v.so $objs /tmp/lstm_mid16_$v.o || exit 1
    echo built $v
  done
else
  for v in $V; do
    echo "== ring depth $v (<= 32 rows _ 33..64 rows)"; EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_RING16_$v.so python $R/scripts/dev/bench_lstm_mid.py 2>&1 | grep "B=" | head -${ROWS:-5}
  done
fi
