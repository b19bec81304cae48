#!/bin/bash
# Builds and runs scripts/dev/bf16_hazard_repro.hip with one and two waves per SIMD (run on the GPU box from the repo root).
set -u
F="--offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc -Wno-unused-result"
for nw in 8 4; do
  hipcc $F -DMB_NW=$nw scripts/dev/bf16_hazard_repro.hip -o /tmp/repro$nw 2>/dev/null || { echo "build failed (MB_NW=$nw)"; exit 1; }
done
echo "== two waves per SIMD"; /tmp/repro8 16384 20
echo "== one wave per SIMD (what the tree ships)"; /tmp/repro4 16384 20
echo "== two waves per SIMD, smaller launch"; /tmp/repro8 2048 40
# Variants that separate "two waves per SIMD" from "the two-wave build spills registers" (ring 4: 52 bytes of scratch per
# lane; ring 2 / 3: none; the one-wave build has a 512-register budget and never spills):
for ring in 2 3; do
  hipcc $F -DMB_NW=8 -DMB_RING=$ring scripts/dev/bf16_hazard_repro.hip -o /tmp/repro8r$ring 2>/dev/null && { echo "== two waves per SIMD, coefficient ring of $ring (no scratch)"; /tmp/repro8r$ring 16384 20; }
done
