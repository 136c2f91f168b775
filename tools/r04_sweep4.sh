#!/bin/bash
# Round-4 experiment 4: row alignment of the activation planes (PADX 4 = 16-byte, 8 = 32-byte, 16 = 64-byte, 32 = 128-byte aligned rows)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4d
mkdir -p $OUT
export WINO_DYNAMIC=0
run() { B=$ROOT/tools/$1
  $B 90 294 64 128 0 50 240
  WINO_BATCH=4 $B 90 294 64 128 0 50 240
  $B 90 294 128 128 1 50 240
  $B 45 147 128 512 0 50 240
  $B 180 588 64 64 0 50 228
  $B 180 588 64 64 1 50 228
  WINO_DYNAMIC=1 $B 360 1176 64 64 1 20 244
}
{
for B in wino_bench4 wino_bench4_padx8 wino_bench4_padx16 wino_bench4_padx32 wino_bench4_abl16 wino_bench4_padx32_abl16; do echo "=== $B"; run $B; done
} > $OUT/sweep4.log 2>&1
