#!/bin/bash
# Dev: the frame-per-lane SMPL kernel with the reverse pass on 8 waves (two per SIMD) instead of 4: build a variant of
# the library (-DTL_BWD_WAVES=8) and time the SMPL evaluation alone.   usage: build | run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for v in ${VARIANTS:-TL_BWD_WAVES=4 TL_BWD_WAVES=8}; do
    f=$(echo $v | tr '=+' '__')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $(echo $v | sed "s/+/ -D/g; s/^/-D/") -c $C/smpl_tile.hip -o /tmp/smpl_tile_$f.o 2>/dev/null || exit 1
    objs=$(ls $C/*.o | grep -v smpl_tile.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_$f.so $objs /tmp/smpl_tile_$f.o || exit 1
    echo built $f
  done
else
  for so in $R/scripts/dev/bin/libempose_TL_*.so; do
    echo "== $(basename $so)"; EMPOSE_LIB_PATH=$so python $R/scripts/dev/bench_chain.py 2>&1 | grep "us/launch"
  done
fi
