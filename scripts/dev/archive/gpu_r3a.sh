#!/bin/bash
# round 3, first GPU pass: new tests, the whole GPU suite, RCCL single-rank self-tests, bench, LSTM small-batch profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3a
timeout 1500 python -m pytest tests/test_hip_round3.py -x -q -m gpu > $O/tests_round3.log 2>&1; echo "round3 tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/tests_round3.log
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_hip_round3.py > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -8 $O/tests_gpu.log
timeout 600 python bench.py --force_dist --steps 10 --warmup 3 > $O/bench_force_dist_nccl_world1.log 2>&1; echo "bench force_dist rc=$?" | tee -a $O/summary.txt
tail -2 $O/bench_force_dist_nccl_world1.log | cut -c1-1500
timeout 600 python scripts/train.py --force_dist --steps 10 --warmup 3 --bs_train 64 --json > $O/train_force_dist_nccl_world1.log 2>&1; echo "train force_dist rc=$?" | tee -a $O/summary.txt
tail -3 $O/train_force_dist_nccl_world1.log | cut -c1-800
timeout 600 python scripts/evaluate_real.py --synthetic --force_dist --json > $O/evaluate_real_force_dist_nccl_world1.log 2>&1; echo "evaluate_real force_dist rc=$?" | tee -a $O/summary.txt
tail -2 $O/evaluate_real_force_dist_nccl_world1.log | cut -c1-800
timeout 300 python scripts/dev/bench_lstm_small.py > $O/lstm_small_batch_per_step.txt 2>&1; echo "lstm small rc=$?" | tee -a $O/summary.txt
cat $O/lstm_small_batch_per_step.txt
