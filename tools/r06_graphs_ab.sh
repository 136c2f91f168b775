#!/bin/bash
# round 6, same box: launch segments replayed from HIP graphs (default for FP16 / INT8 engines) against plain launches (SPVO_TUNE_GRAPHS=0)
O=gpurun_out/r6h; mkdir -p $O
for rep in 1 2 3; do
  for mode in 1 0; do
    for cfg in 3 5; do
      SPVO_TUNE_GRAPHS=$mode python bench.py --config $cfg --no-cpu-baseline --no-extras --no-profile > $O/c${cfg}_g${mode}_$rep.json 2> $O/c${cfg}_g${mode}_$rep.err
    done
  done
done
for mode in 2 0; do SPVO_TUNE_GRAPHS=$mode python bench.py --no-cpu-baseline --no-extras > $O/c2_g${mode}_1.json 2> $O/c2_g${mode}_1.err; done
python - <<'PY'
import json, glob
for cfg in (3, 5, 2):
    for mode in (1, 2, 0):
        v = []
        for f in sorted(glob.glob("gpurun_out/r6h/c%d_g%d_*.json" % (cfg, mode))):
            try:
                d = json.loads(open(f).read().strip().splitlines()[-1]); v.append((d["value"], d["spread_pct"], d["latency_ms"]["p50"]))
            except Exception as e:
                v.append(("ERR", open(f.replace(".json", ".err")).read()[-300:]))
        if v: print("config", cfg, {1: "graphs      ", 2: "graphs (all)", 0: "plain       "}[mode], v)
PY
