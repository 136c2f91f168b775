#!/bin/bash
# round 6: config 3 (and 5) after the two-solves-in-flight change: frame rate beside the pipeline's own diagnostics (trunk_timing, solve_timing)
O=gpurun_out/r6c; mkdir -p $O
for cfg in 3 5; do
for i in 1 2; do
  SPVO_TUNE_TRUNK_TIMING=1 SPVO_TUNE_SOLVE_TIMING=1 python bench.py --config $cfg --no-cpu-baseline --no-extras > $O/b_${cfg}_$i.json 2> $O/b_${cfg}_$i.err
  python - $cfg $i <<'PY'
import json, sys
cfg, i = sys.argv[1], sys.argv[2]
d = json.loads(open("gpurun_out/r6c/b_%s_%s.json" % (cfg, i)).read().strip().splitlines()[-1])
err = [l.strip() for l in open("gpurun_out/r6c/b_%s_%s.err" % (cfg, i)) if "[spvo]" in l or "timing]" in l]
print("config", cfg, "run", i, d["value"], d["ms_per_step_min"], d["ms_per_step_max"], d["latency_ms"]["p50"])
for key in ("trunk timing", "tail stream", "host:", "pairs per launch", "solve timing", "host solve timing"):
    for l in [l for l in err if key in l][-1:]: print("    ", l[:330])
PY
done
done
python tools/step_breakdown.py --config 3 --py-loop --no-cpu-baseline --no-extras > $O/sb_cfg3.json 2> $O/sb_cfg3.err
grep -A16 "host time per call" $O/sb_cfg3.err | cut -c1-220
