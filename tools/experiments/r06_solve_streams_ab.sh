#!/bin/bash
# round 6, one box: one solver stream against two (SPVO_TUNE_SOLVE_STREAMS=2: consecutive solves overlap), at look-ahead depths 4 and 6
O=gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_odometry.py tests/test_gpu_pipeline.py tests/test_gpu_host.py tests/test_gpu_long_sequence.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2; do
  for ss in 2 1; do
    for d in 4 6; do
      for cfg in 3 5; do
        SPVO_TUNE_SOLVE_STREAMS=$ss SPVO_TUNE_TRUNK_TIMING=$((rep == 2)) python bench.py --config $cfg --depth $d --no-cpu-baseline --no-extras --no-profile > $O/c${cfg}_s${ss}_d${d}_$rep.json 2> $O/c${cfg}_s${ss}_d${d}_$rep.err
      done
    done
  done
done
for ss in 2 1; do SPVO_TUNE_SOLVE_STREAMS=$ss python bench.py --no-cpu-baseline --no-extras --no-profile > $O/c2_s${ss}_d4_1.json 2> $O/c2_s${ss}_d4_1.err; done
python - <<'PY'
import json, glob
for cfg in (3, 5, 2):
    for ss in (1, 2):
        for d in (4, 6):
            v = []
            for f in sorted(glob.glob("gpurun_out/r6w/c%d_s%d_d%d_*.json" % (cfg, ss, d))):
                try:
                    r = json.loads(open(f).read().strip().splitlines()[-1]); v.append((r["value"], r["spread_pct"], r["latency_ms"]["p50"]))
                except Exception as e:
                    v.append(("ERR", open(f.replace(".json", ".err")).read()[-300:]))
            if v: print("config", cfg, "solver streams", ss, "depth", d, v)
for cfg in (3, 5):
    for ss in (1, 2):
        for d in (4, 6):
            err = [l.strip() for l in open("gpurun_out/r6w/c%d_s%d_d%d_2.err" % (cfg, ss, d)) if "[spvo]" in l]
            for key in ("trunk timing", "tail stream", "host:"):
                for l in [l for l in err if key in l][-1:]: print("   config", cfg, "streams", ss, "depth", d, l[:300])
PY
