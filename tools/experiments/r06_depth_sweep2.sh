#!/bin/bash
# round 6, one box, on top of the fused solver launch: look-ahead depth 4 / 6 / 8 (twelve buffer sets, ten submissions in flight)
O=gpurun_out/r6d2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_host.py tests/test_gpu_long_sequence.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for rep in 1 2 3; do
  for d in 4 6 8; do
    for cfg in 3 5; do
      SPVO_TUNE_TRUNK_TIMING=$((rep == 3)) python bench.py --config $cfg --depth $d --no-cpu-baseline --no-extras --no-profile > $O/c${cfg}_d${d}_$rep.json 2> $O/c${cfg}_d${d}_$rep.err
    done
  done
done
for d in 4 6; do python bench.py --depth $d --no-cpu-baseline --no-extras --no-profile > $O/c2_d${d}_1.json 2> $O/c2_d${d}_1.err; done
python - <<'PY'
import json, glob
for cfg in (3, 5, 2):
    for d in (4, 6, 8):
        v = []
        for f in sorted(glob.glob("gpurun_out/r6d2/c%d_d%d_*.json" % (cfg, d))):
            try:
                r = json.loads(open(f).read().strip().splitlines()[-1]); v.append((r["value"], r["spread_pct"], r["latency_ms"]["p50"]))
            except Exception as e:
                v.append(("ERR", open(f.replace(".json", ".err")).read()[-300:]))
        if v: print("config", cfg, "depth", d, v)
for d in (4, 6, 8):
    err = [l.strip() for l in open("gpurun_out/r6d2/c3_d%d_3.err" % d) if "[spvo]" in l]
    for key in ("trunk timing", "tail stream", "host:"):
        for l in [l for l in err if key in l][-1:]: print("   config 3 depth", d, l[:280])
    for l in [l for l in err if "since the previous launch" in l][4:7]: print("      ", l[:250])
PY
