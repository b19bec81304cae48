#!/bin/bash
# round 3: blend GEMMs with the update / Rodrigues reverse folded in -- tile tests, A/B bench
set -u
mkdir -p gpurun_out/r3h
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "frame_per_lane or headline" 2>&1 | tail -15 > gpurun_out/r3h/tests.log
tail -5 gpurun_out/r3h/tests.log
for v in 1 0 1 0; do
  echo "smpl_fuse=$v"; python bench.py --steps 20 --warmup 3 --no_cpu_baseline --option smpl_fuse=$v 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()})"
done
