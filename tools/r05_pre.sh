#!/bin/bash
# preprocess fused into the first layer: the GPU tests that cover it, then the headline with and without it on the same box
O=gpurun_out/r5j; mkdir -p $O
python -m pytest tests/test_gpu_host.py tests/test_gpu_pipeline.py tests/test_gpu_post.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2 3; do
  for v in 1 0; do
    SPVO_TUNE_PREPROCESS_FUSED=$v python bench.py --no-cpu-baseline --legs host > $O/head_pre${v}_$rep.json 2> $O/head_pre${v}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5j/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    hi = d.get("host_interface") or {}
    print(f, d["value"], d["ms_per_step"], {k: (v.get("value") if isinstance(v, dict) else v) for k, v in hi.items() if k in ("synchronous", "lookahead")}, d.get("stages_ms", {}).get("net"))
PY
