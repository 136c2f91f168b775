#!/bin/bash
# config 3 (FP16 192x640): where the host spends a step, and how busy the streams are
O=gpurun_out/r5k; mkdir -p $O
python tools/step_breakdown.py --config 3 --py-loop --no-cpu-baseline --no-extras > $O/sb_cfg3.json 2> $O/sb_cfg3.err
SPVO_TUNE_TRUNK_TIMING=1 SPVO_TUNE_SOLVE_TIMING=1 python bench.py --config 3 --no-cpu-baseline --no-extras > $O/tt_cfg3.json 2> $O/tt_cfg3.err
grep -A14 "host time per call" $O/sb_cfg3.err | cut -c1-220
grep "trunk timing\|tail stream\|host:\|solve" $O/tt_cfg3.err | tail -8 | cut -c1-330
python - <<'PY'
import json
for f in ("gpurun_out/r5k/sb_cfg3.json", "gpurun_out/r5k/tt_cfg3.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
