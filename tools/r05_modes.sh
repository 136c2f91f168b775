#!/bin/bash
# config 3 with two tail streams, process by process: frame rate beside the pipeline's own diagnostics (trunk_timing)
O=gpurun_out/r5m2; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  SPVO_TUNE_TRUNK_TIMING=1 python bench.py --config 3 --no-cpu-baseline --no-extras > $O/b_$i.json 2> $O/b_$i.err
  python - $i <<'PY'
import json, sys
i = sys.argv[1]
d = json.loads(open("gpurun_out/r5m2/b_%s.json" % i).read().strip().splitlines()[-1])
err = [l.strip() for l in open("gpurun_out/r5m2/b_%s.err" % i) if "[spvo]" in l]
tt = [l for l in err if "trunk timing" in l][-1:]; tl = [l for l in err if "tail stream" in l][-1:]; ho = [l for l in err if "host:" in l][-1:]; pat = [l for l in err if "pairs per launch" in l][-1:]
print("run", i, d["value"], d["ms_per_step_min"], d["ms_per_step_max"])
for l in tt + tl + ho: print("    ", l[:250])
for l in pat: print("    ", l[:200])
PY
done
