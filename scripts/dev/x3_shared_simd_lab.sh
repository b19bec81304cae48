#!/bin/bash
# Dev: the library built WITHOUT X3_EXCLUSIVE_SIMD (csrc/bf16x3.h) -- kernels on v_mfma_f32_32x32x16_bf16 then share their SIMD
# with whatever another stream runs -- against the shipped build, on the check of tests/test_hip_round6.py::
# test_three_piece_kernels_keep_their_bits_beside_a_storing_kernel_on_another_stream (scripts/dev/x3_shared_simd.py).
# usage (container): bash scripts/dev/x3_shared_simd_lab.sh build ; (GPU box): bash scripts/dev/x3_shared_simd_lab.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/em_pose_amd/csrc
mkdir -p $R/scripts/dev/bin /tmp/x3lab
if [ "${1:-build}" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  objs=""
  for f in $C/*.hip; do
    o=/tmp/x3lab/$(basename $f .hip).o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DX3_LAB_SHARED_SIMD -c $f -o $o 2>/dev/null &
    objs="$objs $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/dev/bin/libempose_X3_LAB_SHARED_SIMD.so $objs || exit 1
  echo built
else
  echo "== shipped build"; python $R/scripts/dev/x3_shared_simd.py
  echo "== without X3_EXCLUSIVE_SIMD"; EMPOSE_LIB_PATH=$R/scripts/dev/bin/libempose_X3_LAB_SHARED_SIMD.so python $R/scripts/dev/x3_shared_simd.py
  # the scenario that found it: a 64-window training step (side streams on, three-piece layer products), five runs of the
  # same three steps -- do their printed losses agree to the last digit?
  for lib in "" $R/scripts/dev/bin/libempose_X3_LAB_SHARED_SIMD.so; do
    echo "== training, 64 windows, 3 steps x 5 runs: ${lib:-shipped build}"
    for i in 1 2 3 4 5; do EMPOSE_LIB_PATH=$lib python $R/scripts/train.py --steps 3 --warmup 1 --bs_train 64 2>/dev/null | grep "^\[TRAIN" | sed 's/ elapsed.*//' | md5sum; done | sort | uniq -c
  done
fi
