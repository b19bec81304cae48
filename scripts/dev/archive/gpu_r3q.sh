#!/bin/bash
# final fuzz passes of round 3
set -u
mkdir -p gpurun_out/r3q
( timeout 400 python tests/fuzz/fuzz_lstm.py 3401 240 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/r3q/fuzz_lstm.txt
( timeout 400 python tests/fuzz/fuzz_lgd.py 3402 240 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r3q/fuzz_lgd.txt
( timeout 400 python tests/fuzz/fuzz_train.py 3403 240 2>&1 | grep -v amdgpu.ids | tail -4 ) > gpurun_out/r3q/fuzz_train.txt
cat gpurun_out/r3q/fuzz_*.txt
