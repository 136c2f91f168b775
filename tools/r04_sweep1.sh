#!/bin/bash
# Round-4 experiment 1: separate the F(4x4) kernel's per-launch, per-tile and per-item costs on the deep layers' shapes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4a
mkdir -p $OUT
B=$ROOT/tools/wino_bench4
export WINO_DYNAMIC=0
{
echo "== K sweep at 90x294 (240 tiles, 1 per WG), cout 128"
for cin in 8 16 32 64 128 256; do $B 90 294 $cin 128 0 50; done
echo "== K sweep at 90x294 pooled"
for cin in 16 64 128; do $B 90 294 $cin 128 1 50; done
echo "== batch sweep 90x294 64->128 (tiles per WG 1,2,3,4) grid 240"
for b in 2 4 6 8; do WINO_BATCH=$b $B 90 294 64 128 0 50 240; done
echo "== batch sweep 90x294 128->128 pool grid 240"
for b in 2 4 6 8; do WINO_BATCH=$b $B 90 294 128 128 1 50 240; done
echo "== batch sweep 45x147 128->512 grid 240"
for b in 2 4; do WINO_BATCH=$b $B 45 147 128 512 0 50 240; done
echo "== batch sweep 180x588 64->64 grid 228"
for b in 2 4; do WINO_BATCH=$b $B 180 588 64 64 0 50 228; done
echo "== 180x588 64->64 grid 256 dynamic"
WINO_DYNAMIC=1 $B 180 588 64 64 0 50 256
WINO_DYNAMIC=1 WINO_BATCH=4 $B 180 588 64 64 0 50 256
WINO_DYNAMIC=1 WINO_BATCH=4 $B 180 588 64 64 0 50 228
echo "== conv4a F(4x4) 45x147 128->128: 60 tiles"
$B 45 147 128 128 0 50
WINO_BATCH=4 $B 45 147 128 128 0 50
WINO_BATCH=8 $B 45 147 128 128 0 50
echo "== conv1b"
WINO_DYNAMIC=1 $B 360 1176 64 64 1 20 244
} > $OUT/sweep1.log 2>&1
cd /tmp && python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err
