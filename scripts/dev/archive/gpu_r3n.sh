#!/bin/bash
# round-3 fuzz passes (bounded) + training kernel statistics at 12 windows
set -u
mkdir -p gpurun_out/r3n
bash scripts/dev/gpu_r3m.sh > gpurun_out/r3n/m.log 2>&1
cp gpurun_out/r3m/train_kernel_stats_bs12.csv gpurun_out/r3n/ 2>/dev/null
( timeout 300 python tests/fuzz/fuzz_lgd.py 3301 150 2>&1 | grep -v amdgpu.ids | tail -4 ) > gpurun_out/r3n/fuzz_lgd.txt
( timeout 400 python tests/fuzz/fuzz_train.py 3302 200 2>&1 | grep -v amdgpu.ids | tail -4 ) > gpurun_out/r3n/fuzz_train.txt
( timeout 200 python tests/fuzz/fuzz_lstm.py 3303 100 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/r3n/fuzz_lstm.txt
( timeout 200 python tests/fuzz/fuzz_linear_mesh.py 3304 100 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r3n/fuzz_linear_mesh.txt
cat gpurun_out/r3n/fuzz_*.txt
