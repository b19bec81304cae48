#!/bin/bash
# Dev: where does a step of lstm_persist_kernel (B = 1) spend its time?  Builds variants of the library with parts of the
# step compiled out (results are then wrong; only the timing matters) and times the LSTM alone.
#   PL_NOWAIT   polls are issued once and taken as they are (no retry)
#   PL_NOPOLL   no exchange loads at all (hidden states read as zeros)
#   PL_NOREDUCE no wave reduction / activations (the dot products still run)
#   PL_NOX      no staging of the stored input
# The PL_* switches are NOT in the tree: the lab patched csrc/lstm.hip locally (#ifdef around the poll loads, the retry
# loop, the input staging, the wave reduction, the barriers, a second row buffer), built one library per switch and
# threw the patch away.  What it measured, B = 1, F = 256, per wavefront step (round 3):
#   as shipped 3.17 us | polls taken as they come (no retry) 3.05 | no exchange loads at all 2.32 | no wave reduction
#   2.63 | no staging of the stored input 2.81 | barriers that wait for LDS only 3.17 | the stored input fetched a step
#   ahead 3.79 (worse) | 7 instead of 17 shuffles for a single row 2.95-3.20 (inside the noise) | two row buffers, one
#   barrier per step 3.20.
# So the exchange itself is ~0.85 us of a step (a tagged word takes 0.5-0.6 us one way between any two workgroups, same
# XCD or not: scripts/dev/xcd_pingpong.hip) and no single local piece is the rest; the step did not get shorter.
# usage (container): bash scripts/dev/persist_lab.sh build ;  (GPU box): bash scripts/dev/persist_lab.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for v in ${VARIANTS:-BASE PL_NOWAIT PL_NOPOLL PL_NOREDUCE PL_NOX}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $(echo $v | sed "s/+/ -D/g; s/^/-D/") -c $C/lstm.hip -o /tmp/lstm_$v.o 2>/dev/null || exit 1
    objs=$(ls $C/*.o | grep -v lstm.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_$v.so $objs /tmp/lstm_$v.o || exit 1
    echo built $v
  done
else
  for v in ${VARIANTS:-BASE PL_NOWAIT PL_NOPOLL PL_NOREDUCE PL_NOX}; do
    echo "== $v"; EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_$v.so python $R/scripts/dev/bench_lstm_small.py 2>&1 | grep "B= 1 \|B= 4 \|B=12"
  done
fi
