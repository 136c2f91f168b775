#!/bin/bash
# round 6, same box: one tail chain for both pairs of a group (SPVO_TUNE_TAIL_BATCH=1) against one chain per pair
O=gpurun_out/r6t; mkdir -p $O
# correctness first: the pipeline / host / long-sequence tests with the switch on (side build with -DSPVO_TAIL_BATCH_DEFAULT=1:
# make -C superpoint-stereo-visual-odometry_amd BUILD=build_tb OUT=variants/tb EXTRA=-DSPVO_TAIL_BATCH_DEFAULT=1)
SPVO_LIB_DIR=$PWD/superpoint-stereo-visual-odometry_amd/variants/tb timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_host.py tests/test_gpu_long_sequence.py tests/test_gpu_post.py -x -q -p no:cacheprovider > $O/pytest_batch.log 2>&1
tail -3 $O/pytest_batch.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_host.py tests/test_gpu_post.py tests/test_gpu_match.py -x -q -p no:cacheprovider > $O/pytest_default.log 2>&1
tail -3 $O/pytest_default.log
for rep in 1 2; do
  for mode in 1 0; do
    for cfg in 3 5; do
      SPVO_TUNE_TAIL_BATCH=$mode python bench.py --config $cfg --no-cpu-baseline --no-extras --no-profile > $O/c${cfg}_b${mode}_$rep.json 2> $O/c${cfg}_b${mode}_$rep.err
    done
  done
done
for mode in 1 0; do SPVO_TUNE_TAIL_BATCH=$mode python bench.py --no-cpu-baseline --no-extras --no-profile > $O/c2_b${mode}_1.json 2> $O/c2_b${mode}_1.err; done
python - <<'PY'
import json, glob
for cfg in (3, 5, 2):
    for mode in (1, 0):
        v = []
        for f in sorted(glob.glob("gpurun_out/r6t/c%d_b%d_*.json" % (cfg, mode))):
            try:
                d = json.loads(open(f).read().strip().splitlines()[-1]); v.append((d["value"], d["spread_pct"], d["latency_ms"]["p50"]))
            except Exception as e:
                v.append(("ERR", open(f.replace(".json", ".err")).read()[-300:]))
        if v: print("config", cfg, {1: "one chain per group", 0: "one chain per pair "}[mode], v)
PY
