#!/bin/bash
# wavefront BPTT: tests, A/B of the training step
set -u
mkdir -p gpurun_out/r3k
python -m pytest tests/test_hip_round3.py tests/test_hip_parity.py -x -q -m gpu -k "lstm_training or reverse_lstm or training_step or training_gradients" 2>&1 | tail -15 > gpurun_out/r3k/tests.log
tail -6 gpurun_out/r3k/tests.log
for bs in 12 256; do for v in 1 0 1 0; do
  echo -n "bs $bs bptt_wave=$v: "; python scripts/train.py --steps 30 --bs_train $bs --graph --json --option bptt_wave=$v 2>/dev/null | tail -1 | cut -c1-200
done; done
