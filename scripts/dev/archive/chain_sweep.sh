#!/bin/bash
# Dev: build variants of smpl.hip (-D switches) into /tmp and time each with scripts/dev/bench_chain.py.
# Usage: chain_sweep.sh "-DEMPOSE_CP_WAVES=3 -DEMPOSE_CP_PAIRS=2" ...
mkdir -p /tmp/pv
for f in em_pose_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  [ $b = smpl ] && continue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -c $f -o /tmp/pv/$b.o 2>/dev/null &
done
wait
for cfg in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DEMPOSE_CHAIN_TRACE $cfg \
    -c em_pose_amd/csrc/smpl.hip -o /tmp/pv/smpl.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "chain_pairs_kernel" | grep -E "VGPRs:|VGPRs Spill|Occupancy" | sed 's/.*remark: *//; s/ \[-R.*//' | tr '\n' ';'
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pv/libempose_hip.so /tmp/pv/*.o
  echo; echo "cfg $cfg:"; EMPOSE_LIB_PATH=/tmp/pv/libempose_hip.so python scripts/dev/bench_chain.py 2>&1 | tail -3
done
