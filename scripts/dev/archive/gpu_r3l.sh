#!/bin/bash
# pipelined streaming driver: tests, timing of the one-recording-at-a-time evaluation
set -u
mkdir -p gpurun_out/r3l
python -m pytest tests -x -q -m gpu -k "streaming or evaluate or sequential or driver or suppression or masked or golden" 2>&1 | tail -8 > gpurun_out/r3l/tests.log
tail -4 gpurun_out/r3l/tests.log
for i in 1 2; do python scripts/evaluate_real.py --synthetic --sequential --repeat 3 --json 2>/dev/null | tail -1 | cut -c1-260; done
python scripts/evaluate_real.py --synthetic --repeat 2 --json 2>/dev/null | tail -1 | cut -c1-200
