#!/bin/bash
# config 3 (192x640): how many NMS round launches go out with a submission before the one-workgroup finishing kernel
O=gpurun_out/r5v; mkdir -p $O
for rep in 1 2; do
  for v in 4 3 2; do
    SPVO_TUNE_NMS_FIRST=$v python bench.py --config 3 --no-cpu-baseline --no-extras > $O/cfg3_nf${v}_$rep.json 2> /dev/null
    SPVO_TUNE_NMS_FIRST=$v python bench.py --config 5 --no-cpu-baseline --no-extras > $O/cfg5_nf${v}_$rep.json 2> /dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5v/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("nms_host_continuations"))
PY
