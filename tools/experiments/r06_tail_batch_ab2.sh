#!/bin/bash
# round 6, one box, on top of the fused solver launch: one tail chain per group (SPVO_TUNE_TAIL_BATCH=1) against one per pair
O=gpurun_out/r6b2; mkdir -p $O
for rep in 1 2 3; do
  for mode in 1 0; do
    SPVO_TUNE_TAIL_BATCH=$mode SPVO_TUNE_TRUNK_TIMING=$((rep == 3)) python bench.py --config 3 --no-cpu-baseline --no-extras --no-profile > $O/c3_b${mode}_$rep.json 2> $O/c3_b${mode}_$rep.err
  done
done
python - <<'PY'
import json, glob
for mode in (1, 0):
    v = []
    for f in sorted(glob.glob("gpurun_out/r6b2/c3_b%d_*.json" % mode)):
        try:
            r = json.loads(open(f).read().strip().splitlines()[-1]); v.append((r["value"], r["spread_pct"], r["latency_ms"]["p50"]))
        except Exception as e:
            v.append(("ERR", open(f.replace(".json", ".err")).read()[-300:]))
    print("config 3", {1: "one chain per group", 0: "one chain per pair "}[mode], v)
    err = [l.strip() for l in open("gpurun_out/r6b2/c3_b%d_3.err" % mode) if "[spvo]" in l]
    for key in ("trunk timing", "tail stream", "host:"):
        for l in [l for l in err if key in l][-1:]: print("   ", l[:300])
    for l in [l for l in err if "since the previous launch" in l][4:8]: print("      ", l[:250])
PY
