#!/bin/bash
# synchronous leg (tools/sync_leg.py) of library variants on one box, interleaved: tools/r04_sync_ab.sh <variant> ... ("base" = in-tree)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
P=$ROOT/superpoint-stereo-visual-odometry_amd
mkdir -p $P/variants/base; cp $P/libspvo.so $P/libspvo_host.so $P/variants/base/
cd /tmp
for r in 1 2 3; do for V in "$@"; do
  cp $P/variants/$V/libspvo.so $P/libspvo.so; cp $P/variants/$V/libspvo_host.so $P/libspvo_host.so
  echo "$V r$r: $(python3 $ROOT/tools/sync_leg.py 300 0 2>/dev/null | grep 'depth 0' | cut -c1-200)"
done; done
cp $P/variants/base/libspvo.so $P/libspvo.so; cp $P/variants/base/libspvo_host.so $P/libspvo_host.so
