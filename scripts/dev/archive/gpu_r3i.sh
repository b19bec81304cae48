#!/bin/bash
# full GPU suite + bench after the blend-GEMM fusion
set -u
mkdir -p gpurun_out/r3i
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|FAILED" | tail -8 > gpurun_out/r3i/tests.log
cat gpurun_out/r3i/tests.log
python bench.py --steps 20 --warmup 3 --no_cpu_baseline 2>/dev/null | tail -1 | cut -c1-400
python scripts/evaluate_real.py --synthetic --repeat 2 --json 2>/dev/null | tail -1 | cut -c1-300
