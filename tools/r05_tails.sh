#!/bin/bash
# one or two tail streams: the tests, then configs 3 / 5 and the headline with either (same box, alternating)
O=gpurun_out/r5w; mkdir -p $O
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_host.py tests/test_gpu_match.py tests/test_gpu_post.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
  for v in 2 1; do
    SPVO_TUNE_TAIL_STREAMS=$v python bench.py --config 3 --no-cpu-baseline --no-extras > $O/cfg3_ts${v}_$rep.json 2> /dev/null
    SPVO_TUNE_TAIL_STREAMS=$v python bench.py --config 5 --no-cpu-baseline --no-extras > $O/cfg5_ts${v}_$rep.json 2> /dev/null
    SPVO_TUNE_TAIL_STREAMS=$v python bench.py --no-cpu-baseline --legs host,trained > $O/head_ts${v}_$rep.json 2> /dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5w/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    hi = d.get("host_interface") or {}
    print(f, d["value"], d["ms_per_step"], d.get("spread_pct"), (d.get("latency_ms") or {}).get("p50"), {k: (v.get("value") if isinstance(v, dict) else v) for k, v in hi.items() if k in ("synchronous", "lookahead")}, (d.get("trained_workload") or {}).get("value"))
PY
