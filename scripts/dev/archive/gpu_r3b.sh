#!/bin/bash
# round 3: frame-per-lane SMPL path -- parity tests, then the headline bench with it on and off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3b
timeout 900 python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "frame_per_lane" > $O/tests_tile.log 2>&1; echo "tile tests rc=$?" | tee -a $O/summary.txt
tail -30 $O/tests_tile.log
timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/bench_tile.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
grep "^{" $O/bench_tile.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d['breakdown_ms_per_step'], indent=0))"
