#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3e
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3e
timeout 3000 python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -25 $O/tests_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1
grep "^{" $O/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d['breakdown_ms_per_step'])); print(d['cpu_baseline']['max_abs_diff_pose_shape_joints'], d['cpu_baseline']['mpjpe_hip_vs_oracle_mm'])"
