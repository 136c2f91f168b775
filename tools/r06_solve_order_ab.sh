#!/bin/bash
# round 6, same box: the solve of frame k submitted BEFORE frame k - 1 is collected (two in flight; default) against rounds 1-5's order
# (collect, then submit: SPVO_TUNE_SOLVE_COLLECT_FIRST=1), three alternating runs each, configs 3 and 5 and the headline
O=gpurun_out/r6e; mkdir -p $O
for rep in 1 2 3; do
  for mode in 0 1; do
    for cfg in 3 5; do
      SPVO_TUNE_SOLVE_COLLECT_FIRST=$mode python bench.py --config $cfg --no-cpu-baseline --no-extras > $O/c${cfg}_m${mode}_$rep.json 2> $O/c${cfg}_m${mode}_$rep.err
    done
  done
done
for mode in 0 1; do SPVO_TUNE_SOLVE_COLLECT_FIRST=$mode python bench.py --no-cpu-baseline --no-extras > $O/c2_m${mode}_1.json 2> $O/c2_m${mode}_1.err; done
python - <<'PY'
import json, glob
for cfg in (3, 5, 2):
    for mode in (0, 1):
        v = []
        for f in sorted(glob.glob("gpurun_out/r6e/c%d_m%d_*.json" % (cfg, mode))):
            d = json.loads(open(f).read().strip().splitlines()[-1]); v.append((d["value"], d["spread_pct"], d["latency_ms"]["p50"]))
        print("config", cfg, "collect_first" if mode else "submit_first ", v)
PY
