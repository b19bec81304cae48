#!/bin/bash
# Collect the numbers committed under profiles/ for one round (run on the GPU box from the repo root).
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
if [ -z "${SKIP_TESTS:-}" ]; then
  python -m pytest tests -m gpu -q -s 2>&1 | grep -E "passed|failed|error|fuzz slice|vs float64|fingerprints" | tail -60 > gpurun_out/${TAG}_gpu_tests.log
fi
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_lgdrnn12_b1024.json       # incl. `secondary`
python bench.py --steps 20 --warmup 3 --no_rnn --batch 256 --no_traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_lgd12_b256_config1.json   # with its own parity sample (cpu_baseline)
python bench.py --gpus 1 --force_dist --steps 20 --warmup 3 --no_cpu_baseline --no_traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_lgdrnn12_b1024_spawned_rank_nccl.json
python bench.py --steps 20 --warmup 3 --n_markers 6 --no_cpu_baseline --no_traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_lgdrnn6_b1024.json
python bench.py --workload vertices --batch 512 --frames 32 --steps 40 --warmup 8 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_vertices_t16384.json   # three-piece kernel, live PMC traffic
python bench.py --workload vertices --batch 512 --frames 32 --steps 40 --warmup 8 --no_traffic --option mesh_x3=0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_vertices_fp32_mfma_t16384.json
python bench.py --workload vertices --arith bf16x3 --batch 512 --frames 32 --steps 10 --warmup 2 --no_traffic 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_vertices_bf16x3_t16384.json
python scripts/dev/time_mesh.py 16384 0,1,2,3 2>&1 | grep mesh_x3= > gpurun_out/${TAG}_mesh_variants_t16384.txt
python scripts/evaluate_real.py --synthetic --repeat 10 --json 2>/dev/null | tail -1 > gpurun_out/${TAG}_evaluate_real_synthetic_batched.json
python scripts/evaluate_real.py --synthetic --sequential --repeat 6 --json 2>/dev/null | tail -1 > gpurun_out/${TAG}_evaluate_real_synthetic_sequential.json
python scripts/train.py --steps 30 --json 2>/dev/null | tail -1 > gpurun_out/${TAG}_train_step_bs12.json          # default: replayed as a HIP graph
python scripts/train.py --steps 30 --no_graph --json 2>/dev/null | tail -1 > gpurun_out/${TAG}_train_step_bs12_eager.json
( for o in "train_cols=0" "lstm_fewrows=0" "train_cols=0 --option lstm_fewrows=0"; do echo -n "$o: "; python scripts/train.py --steps 30 --json --option $o 2>/dev/null | tail -1; done ) > gpurun_out/${TAG}_train_step_bs12_ablations.txt
( for bs in 12 64 256; do for g in "--no_graph" "--graph" "--single_stream --no_graph"; do echo -n "bs_train $bs $g: "; python scripts/train.py --steps 20 --bs_train $bs $g --json 2>/dev/null | tail -1; done; done ) > gpurun_out/${TAG}_train_batch_scaling.txt
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/scripts/train.py --steps 10 --bs_train 256 --json > $OUT.log 2>&1 )
cp $OUT/t_kernel_stats.csv gpurun_out/${TAG}_train_kernel_stats_bs256.csv
rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $R/scripts/train.py --steps 10 --no_graph --json > $OUT.log 2>&1 )
cp $OUT/t_kernel_stats.csv gpurun_out/${TAG}_train_kernel_stats_bs12.csv
rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lgd -- python $R/bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_traffic --no_fp32_line --no_secondary > $OUT.log 2>&1 )
cp $OUT/lgd_kernel_stats.csv gpurun_out/${TAG}_rocprofv3_kernel_stats_bench_lgdrnn12_b1024.csv
rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o v -- python $R/bench.py --workload vertices --batch 512 --frames 32 --steps 5 --warmup 1 --no_traffic > $OUT.log 2>&1 )
cp $OUT/v_kernel_stats.csv gpurun_out/${TAG}_rocprofv3_kernel_stats_bench_vertices_t16384.csv
rm -rf $OUT
python scripts/dev/bench_lstm_small.py > gpurun_out/${TAG}_lstm_small_batch_per_step.txt 2>&1
python scripts/dev/bench_lstm_mid.py > gpurun_out/${TAG}_lstm_medium_batch_per_step.txt 2>&1
python scripts/dev/prof_seq.py > gpurun_out/${TAG}_streaming_forward_b1_f256.txt 2>&1
bash scripts/dev/train_trace.sh 12 2>&1 | grep -v Segm > gpurun_out/${TAG}_train_step_bs12_trace.txt
ls -la gpurun_out | grep $TAG
