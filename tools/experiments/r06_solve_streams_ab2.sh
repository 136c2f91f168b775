#!/bin/bash
# round 6, one box: one solver stream against two, second stream created with the context (hardware queue of its own)
O=gpurun_out/r6z2; mkdir -p $O
for rep in 1 2 3; do
  for ss in 3 1; do
    for cfg in 3 5; do
      SPVO_TUNE_SOLVE_STREAMS=$ss SPVO_TUNE_TRUNK_TIMING=$((rep == 3)) python bench.py --config $cfg --depth 4 --no-cpu-baseline --no-extras --no-profile > $O/c${cfg}_s${ss}_$rep.json 2> $O/c${cfg}_s${ss}_$rep.err
    done
  done
done
python - <<'PY'
import json, glob
for cfg in (3, 5):
    for ss in (1, 3):
        v = []
        for f in sorted(glob.glob("gpurun_out/r6z2/c%d_s%d_*.json" % (cfg, ss))):
            try:
                r = json.loads(open(f).read().strip().splitlines()[-1]); v.append((r["value"], r["spread_pct"], r["latency_ms"]["p50"]))
            except Exception as e:
                v.append(("ERR", open(f.replace(".json", ".err")).read()[-300:]))
        if v: print("config", cfg, "solver streams", ss, v)
for cfg in (3, 5):
    for ss in (1, 3):
        err = [l.strip() for l in open("gpurun_out/r6z2/c%d_s%d_3.err" % (cfg, ss)) if "[spvo]" in l]
        for key in ("trunk timing", "tail stream", "host:"):
            for l in [l for l in err if key in l][-1:]: print("   config", cfg, "streams", ss, l[:300])
        for l in [l for l in err if "since the previous launch" in l][4:8]: print("      ", l[:250])
PY
