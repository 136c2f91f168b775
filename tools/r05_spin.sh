#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "preprocess_inside or failed_group or pairing" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
  for v in 0 1; do
    SPVO_TUNE_SPIN_WAIT=$v python bench.py --config 3 --no-cpu-baseline --no-extras > $O/cfg3_spin${v}_$rep.json 2> /dev/null
    SPVO_TUNE_SPIN_WAIT=$v python tools/sync_leg.py 300 0 2> /dev/null | tail -1 > $O/sync_spin${v}_$rep.log
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5n/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
for f in sorted(glob.glob("gpurun_out/r5n/sync*.log")): print(f, open(f).read()[:120])
PY
