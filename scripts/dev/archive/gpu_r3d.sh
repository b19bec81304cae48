#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d
export HSA_ENABLE_IPC_MODE_LEGACY=0
TL_WAVES=${TL_WAVES:-4} bash scripts/dev/chain_trace.sh > gpurun_out/r3d/tile_trace.txt 2>&1
tail -12 gpurun_out/r3d/tile_trace.txt
