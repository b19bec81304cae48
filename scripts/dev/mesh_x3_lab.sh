#!/bin/bash
# Dev: what in the overlapped full-mesh kernel (mesh_x3.hip, option mesh_x3 = 1) corrupts the accumulators?  Lab builds:
#   MX_LAB_DUMMY            the real skinning runs one after the other; under the K loop runs a DUMMY skinning (no stores)
#   MX_LAB_DUMMY+MX_LAB_NOLDS   ... whose bone transforms come from registers instead of LDS (vector arithmetic only)
#   MX_LAB_NOK / MX_LAB_NOSKIN / MX_LAB_NOFLUSH   (MODE=time OPTS=2) the sequential kernel without its K loop / skinning / stores
# usage (container): bash scripts/dev/mesh_x3_lab.sh build ; (GPU box): bash scripts/dev/mesh_x3_lab.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin
V=${VARIANTS:-BASE MX_LAB_DUMMY MX_LAB_DUMMY+MX_LAB_NOLDS}
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  for v in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $(echo $v | sed "s/+/ -D/g; s/^/-D/; s/_EQ_/=/g") -c $C/mesh_x3.hip -o /tmp/mesh_x3_$v.o 2>/dev/null || exit 1
    objs=$(ls $C/*.o | grep -v mesh_x3.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_$v.so $objs /tmp/mesh_x3_$v.o || exit 1
    echo built $v
  done
else
  for v in $V; do
    echo "== $v"
    if [ "${MODE:-check}" = time ]; then
      EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_$v.so python $R/scripts/dev/time_mesh.py ${T:-16384} ${OPTS:-1,2} 2>&1 | grep mesh_x3=
    else
      EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_$v.so python $R/scripts/dev/dbg_mesh_x3.py ${T:-4096} 2>&1 | grep -v amdgpu.ids | grep "opt 1"
    fi
  done
fi
