#!/bin/bash
# Dev: chain_sensors_kernel with other workgroup shapes (temporary builds in /tmp, the in-tree library is untouched).
# usage: chain_threads_sweep.sh "4 256" "3 192" ...   (frames per workgroup, threads)
set -e
R=$PWD
for v in "$@"; do
  set -- $v; fr=$1; n=$2
  D=/tmp/chain_v${fr}_$n; rm -rf $D; mkdir -p $D; mkdir -p $D/em_pose_amd/csrc $D/include; cp em_pose_amd/csrc/*.hip em_pose_amd/csrc/*.h $D/em_pose_amd/csrc/; cp include/*.h $D/include/; D0=$D; D=$D/em_pose_amd/csrc
  sed -i "s/return launch_chain_cfg<4, 256>(a, stream);/return launch_chain_cfg<$fr, $n>(a, stream);/" $D/smpl.hip
  OBJS=""
  for f in $D/*.hip; do o=${f%.hip}.o; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -I$R/include -c $f -o $o 2>/dev/null & OBJS="$OBJS $o"; done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libempose_hip.so $OBJS
  echo "frames=$fr threads=$n"; EMPOSE_LIB_PATH=$D/libempose_hip.so python scripts/dev/bench_chain.py 2>&1 | grep "^stop"
done
