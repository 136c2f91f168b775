#!/bin/bash
# Round 4, after the NMS change: the pipeline-level artefacts again (the kernels themselves did not change: tools/collect_profiles_r04.sh's per-layer and PMC files stand)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err   # the driver's command
python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_200.log 2> $OUT/bench_200.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.log 2>&1
STEP_TRACE=400:470 SPVO_TUNE_TRUNK_TIMING=230 python3 $ROOT/tools/step_breakdown.py > $OUT/sb.log 2> $OUT/sb.err
python3 $ROOT/tools/trace_merge.py $OUT/sb.err 400 > $OUT/trace_after.log
grep -A12 "host time per call" $OUT/sb.err > $OUT/step_breakdown.log
python3 $ROOT/tools/fwd_batch.py > $OUT/fwd_batch.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 0 > $OUT/sync_leg.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 2 >> $OUT/sync_leg.log 2>&1
find $OUT -name "*.csv" -size +3M -delete
du -sh $OUT
