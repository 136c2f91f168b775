cd $GRAFT_REPO_ROOT/tools
for B in 2 4; do
echo "== batch $B, conv4a shape"
WINO_BATCH=$B ./wino_bench4 45 147 128 128 0 50
WINO_BATCH=$B ./wino_bench4_tb1 45 147 128 128 0 50
WINO_BATCH=$B ./wino_bench2n 45 147 128 128 0 50
echo "== batch $B, convPa+Da shape"
WINO_BATCH=$B ./wino_bench4 45 147 128 512 0 50
WINO_BATCH=$B ./wino_bench4_tb1 45 147 128 512 0 50
echo "== batch $B, conv3a / conv3b"
WINO_BATCH=$B ./wino_bench4 90 294 64 128 0 50
WINO_BATCH=$B ./wino_bench4_tb1 90 294 64 128 0 50
WINO_BATCH=$B ./wino_bench4 90 294 128 128 1 50
WINO_BATCH=$B ./wino_bench4_tb1 90 294 128 128 1 50
echo "== batch $B, conv2a / conv1b"
WINO_BATCH=$B ./wino_bench4 180 588 64 64 0 30
WINO_BATCH=$B ./wino_bench4_tb1 180 588 64 64 0 30
WINO_BATCH=$B ./wino_bench4 360 1176 64 64 1 20
WINO_BATCH=$B ./wino_bench4_tb1 360 1176 64 64 1 20
done
