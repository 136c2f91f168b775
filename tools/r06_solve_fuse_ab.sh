#!/bin/bash
# round 6, one box: the solver's tail kernel in one launch with the next frame's hypotheses (SPVO_TUNE_SOLVE_KEEP=2: two solves stay pending behind a submit)
# against one chain per frame (SPVO_TUNE_SOLVE_KEEP=1: every tail alone), configs 3 and 5 and the headline
O=gpurun_out/r6s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_odometry.py tests/test_gpu_host.py tests/test_gpu_long_sequence.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2 3; do
  for keep in 2 1; do
    for cfg in 3 5; do
      SPVO_TUNE_SOLVE_KEEP=$keep SPVO_TUNE_TRUNK_TIMING=$((rep == 3)) python bench.py --config $cfg --no-cpu-baseline --no-extras --no-profile > $O/c${cfg}_k${keep}_$rep.json 2> $O/c${cfg}_k${keep}_$rep.err
    done
  done
done
for keep in 2 1; do SPVO_TUNE_SOLVE_KEEP=$keep python bench.py --no-cpu-baseline --no-extras --no-profile > $O/c2_k${keep}_1.json 2> $O/c2_k${keep}_1.err; done
python - <<'PY'
import json, glob
for cfg in (3, 5, 2):
    for keep in (1, 2):
        v = []
        for f in sorted(glob.glob("gpurun_out/r6s/c%d_k%d_*.json" % (cfg, keep))):
            try:
                r = json.loads(open(f).read().strip().splitlines()[-1]); v.append((r["value"], r["spread_pct"], r["latency_ms"]["p50"]))
            except Exception as e:
                v.append(("ERR", open(f.replace(".json", ".err")).read()[-300:]))
        if v: print("config", cfg, "solves kept pending", keep, v)
for cfg in (3, 5):
    for keep in (1, 2):
        err = [l.strip() for l in open("gpurun_out/r6s/c%d_k%d_3.err" % (cfg, keep)) if "[spvo]" in l]
        for key in ("trunk timing", "tail stream", "host:"):
            for l in [l for l in err if key in l][-1:]: print("   config", cfg, "keep", keep, l[:300])
        for l in [l for l in err if "since the previous launch" in l][4:8]: print("      ", l[:250])
PY
