#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3f
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3f
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -q -m gpu -k "training or train or graphed or amass or validation" > $O/tests_train.log 2>&1; echo "train tests rc=$?" | tee -a $O/summary.txt
grep "passed\|failed\|FAILED" $O/tests_train.log | head -20
for bs in 12 64 256; do for g in "" "--graph"; do echo -n "bs_train $bs $g: "; timeout 300 python scripts/train.py --steps 20 --bs_train $bs $g --json 2>/dev/null | tail -1; done; done | tee $O/train_batch_scaling.txt
