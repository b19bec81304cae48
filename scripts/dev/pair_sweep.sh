#!/bin/bash
# Dev: build variants of chain_pair_kernel (threads, pairs per tile, waves per SIMD) into /tmp and time each with
# scripts/dev/bench_chain.py.  Usage: pair_sweep.sh "NT NPAIR WAVES" ...
R=$PWD
mkdir -p /tmp/pv
for f in em_pose_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  [ $b = smpl_pair ] && continue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -c $f -o /tmp/pv/$b.o 2>/dev/null &
done
wait
for cfg in "$@"; do
  set -- $cfg
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DEMPOSE_PAIR_NT=$1 -DEMPOSE_PAIR_NPAIR=$2 -DEMPOSE_PAIR_WAVES=$3 -DEMPOSE_PAIR_TRACE \
    -c em_pose_amd/csrc/smpl_pair.hip -o /tmp/pv/smpl_pair.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|Spill|Occupancy" | tr '\n' ' '
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pv/libempose_hip.so /tmp/pv/*.o
  echo; echo "cfg NT=$1 NPAIR=$2 WAVES=$3:"; EMPOSE_LIB_PATH=/tmp/pv/libempose_hip.so python scripts/dev/bench_chain.py 2>&1 | tail -3
done
