#!/bin/bash
# two tail streams with more hardware queues (GPU_MAX_HW_QUEUES): configs 3 / 5, and every other leg with one tail stream under 4 / 8 queues
O=gpurun_out/r5z; mkdir -p $O
for rep in 1 2; do
for q in 4 8; do
  for v in 1 2; do
    [ $q = 4 ] && [ $v = 2 ] && continue
    GPU_MAX_HW_QUEUES=$q SPVO_TUNE_TAIL_STREAMS=$v python bench.py --config 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 queues $q tails $v', d['value'], d['ms_per_step'], d['latency_ms']['p50'], d.get('spread_pct'))"
    GPU_MAX_HW_QUEUES=$q SPVO_TUNE_TAIL_STREAMS=$v python bench.py --config 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 queues $q tails $v', d['value'], d['ms_per_step'], d['latency_ms']['p50'], d.get('spread_pct'))"
  done
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --legs host,trained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_interface']; print('head queues $q tails 1', d['value'], 'sync', h['synchronous']['value'], 'look', h['lookahead']['value'], 'trained', d['trained_workload']['value'])"
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style queues $q', d['value'])"
done
done | tee $O/tails_q.log
