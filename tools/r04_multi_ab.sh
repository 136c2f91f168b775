#!/bin/bash
# A/B/C... of library builds on one box, interleaved twice: tools/r04_multi_ab.sh <variant> <variant> ...  ("base" = the in-tree build)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mab
mkdir -p $OUT
P=$ROOT/superpoint-stereo-visual-odometry_amd
mkdir -p $P/variants/base; cp $P/libspvo.so $P/libspvo_host.so $P/variants/base/
cd /tmp
for round in 1 2; do
for V in "$@"; do
  cp $P/variants/$V/libspvo.so $P/libspvo.so; cp $P/variants/$V/libspvo_host.so $P/libspvo_host.so
  python3 $ROOT/tools/layer_roofline_json.py $OUT/lr_${V}_$round.json > /dev/null 2> $OUT/lr_${V}_$round.err
  python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 > $OUT/bench_${V}_$round.log 2> $OUT/bench_${V}_$round.err
done
done
cp $P/variants/base/libspvo.so $P/libspvo.so; cp $P/variants/base/libspvo_host.so $P/libspvo_host.so
python3 - $OUT "$@" <<'PY'
import json,sys,glob
out=sys.argv[1]; vs=sys.argv[2:]
names=None
for r in (1,2):
    for v in vs:
        try:
            d=json.load(open(f"{out}/lr_{v}_{r}.json"))
            b=json.loads(open(f"{out}/bench_{v}_{r}.log").read().strip().splitlines()[-1])
            print(f"{v:10s} r{r} " + " ".join(f"{l['duration_us']:6.1f}" for l in d['layers']) + f" | sum {d['conv_stack']['sum_of_layers_us']:6.1f} fwd {d['forward_pass_us']:6.1f} | bench {b['value']:7.1f} f/s k={b['roofline']['avg_kernel_ms']*1e3:5.1f}us")
        except Exception as e: print(v, r, 'ERR', e)
PY
