#!/bin/bash
# Dev: per-phase shader-clock stamps of chain_sensors_kernel. Builds a traced copy of the library in /tmp (the in-tree
# library is untouched) and runs the SMPL entry point at T=32768.
set -e
R=$PWD
mkdir -p /tmp/trace_build
OBJS=""
for f in em_pose_amd/csrc/*.hip; do
  o=/tmp/trace_build/$(basename $f .hip).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DEMPOSE_CHAIN_TRACE -c $f -o $o &
  OBJS="$OBJS $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/trace_build/libempose_hip.so $OBJS
EMPOSE_LIB_PATH=/tmp/trace_build/libempose_hip.so python scripts/dev/bench_chain.py
