#!/bin/bash
# K-loop tail + fast Rodrigues in the blend GEMMs: tests that pin bit-identity / parity, then the bench
set -u
mkdir -p gpurun_out/r3o
python -m pytest tests -x -q -m gpu -k "update_nets or fused or blend or frame_per_lane or golden or headline or mlp or linear" 2>&1 | tail -4
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()})"; done
