#!/bin/bash
# would three stereo pairs per trunk launch (six images) pay?  the F(4x4) layers stand-alone at 2 / 4 / 6 / 8 images per launch
O=gpurun_out/r5y; mkdir -p $O
for shape in "360 1176 64 64 1" "180 588 64 64 0" "180 588 64 64 1" "90 294 64 128 0" "90 294 128 128 1" "45 147 128 512 0"; do
  for b in 2 4 6 8; do
    echo -n "$shape batch $b: "; WINO_DYNAMIC=1 WINO_BATCH=$b tools/wino_bench4 $shape 60 | grep -o "[0-9.]* us" | head -1
  done
done | tee $O/batch6.log
