#!/bin/bash
# device time of the NMS stage (HIP events, un-contended: the synchronous call sequence) for the in-tree library at nms_first = 3, 4, 5 and for variants/rounds4 (four round launches, no finishing kernel)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
P=$ROOT/superpoint-stereo-visual-odometry_amd
cd /tmp
for w in "" sp_squeeze; do
for v in 3 4 5; do echo "in-tree nms_first $v $w: $(SPVO_TUNE_NMS_FIRST=$v python3 $ROOT/tools/sync_breakdown.py $w 2>/dev/null | grep 'device stages' | cut -c1-260)"; done
mkdir -p $P/variants/base; cp $P/libspvo.so $P/libspvo_host.so $P/variants/base/
cp $P/variants/rounds4/libspvo.so $P/libspvo.so; cp $P/variants/rounds4/libspvo_host.so $P/libspvo_host.so
echo "rounds4 $w: $(python3 $ROOT/tools/sync_breakdown.py $w 2>/dev/null | grep 'device stages' | cut -c1-260)"
cp $P/variants/base/libspvo.so $P/libspvo.so; cp $P/variants/base/libspvo_host.so $P/libspvo_host.so
done
