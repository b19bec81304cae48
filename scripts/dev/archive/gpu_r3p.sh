#!/bin/bash
# whole-sequence LSTM on large batches: tests, A/B of the headline step
set -u
timeout 600 python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "whole_sequence_lstm_on_large or full_width or headline" 2>&1 | tail -3
for v in 1 0 1 0; do
  echo -n "lstm_seq=$v: "; python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_traffic --option lstm_seq=$v 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items() if k in ('lstm_step','mlp_fused','sum_of_kernels')})"
done
