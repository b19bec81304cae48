#!/bin/bash
# full GPU suite + split-bf16 vertices bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02f_gpu_tests.txt 2>&1; tail -5 gpurun_out/r02f_gpu_tests.txt
python bench.py --workload vertices --batch 512 --frames 32 --steps 20 --warmup 3 > gpurun_out/r02f_vertices_f32.json 2>&1; tail -1 gpurun_out/r02f_vertices_f32.json
python bench.py --workload vertices --arith bf16x3 --batch 512 --frames 32 --steps 20 --warmup 3 > gpurun_out/r02f_vertices_bf16x3.json 2>&1; tail -1 gpurun_out/r02f_vertices_bf16x3.json
