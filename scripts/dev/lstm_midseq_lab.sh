#!/bin/bash
# Dev: where does a step of lstm_midseq_x3_kernel (whole sequence of a medium batch in one launch) spend its time?  Variants
# of the library <ring 1>_<ring 2>[+FLAG...] (A fragments in flight at <= 32 / 33..64 rows) with parts of the hand-over
# compiled out (results are then wrong; only the timing matters), timed on the stand-alone LSTM (2 x 512, input 60;
# scripts/dev/bench_lstm_mid.py):
#   LQ3_LAB_NOPOLL      no wait for the progress counters
#   LQ3_LAB_NOACK       no wait for the acknowledgement of the plane stores before the counter is raised
#   LQ3_LAB_SAMEPLANES  every step reads the planes of slot 0 again (cache hits instead of fresh lines)
# usage (container): bash scripts/dev/lstm_midseq_lab.sh build ; (GPU box): bash scripts/dev/lstm_midseq_lab.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin
V=${VARIANTS:-6_4 6_4+LQ3_LAB_NOPOLL 6_4+LQ3_LAB_NOACK 6_4+LQ3_LAB_SAMEPLANES 6_4+LQ3_LAB_NOPOLL+LQ3_LAB_NOACK 6_4+LQ3_LAB_NOPOLL+LQ3_LAB_NOACK+LQ3_LAB_SAMEPLANES}
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for v in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DLQ3_RING1=${v%%_*} -DLQ3_RING2=$(echo ${v#*_} | sed 's/+.*//') $(echo $v | grep -o '+.*' | sed 's/+/ -D/g') -c $C/lstm_midseq_x3.hip -o /tmp/lstm_midseq_$v.o 2>/dev/null || exit 1
    objs=$(ls $C/*.o | grep -v lstm_midseq_x3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_MIDSEQ_$v.so $objs /tmp/lstm_midseq_$v.o || exit 1
    echo built $v
  done
else
  for v in $V; do
    echo "== $v"; EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_MIDSEQ_$v.so python $R/scripts/dev/bench_lstm_mid.py 2>&1 | grep "B=" | head -${ROWS:-5}
  done
fi
