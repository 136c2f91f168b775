mkdir -p gpurun_out/r4B; cd /tmp
for v in 3 4 5; do
SPVO_TUNE_NMS_FIRST=$v SPVO_TUNE_TRUNK_TIMING=1 python3 /root/repo/bench.py --no-cpu-baseline --steps 200 --warmup 20 > /root/repo/gpurun_out/r4B/b_$v.log 2> /root/repo/gpurun_out/r4B/b_$v.err
python3 - <<PY
import json
d=json.loads(open('/root/repo/gpurun_out/r4B/b_$v.log').read().strip().splitlines()[-1])
hi=d.get('host_interface',{})
print('nms_first $v: headline', d['value'], '| sync', hi.get('synchronous',{}).get('frames_per_s'), 'lookahead', hi.get('lookahead',{}).get('frames_per_s'), '| trained', d.get('trained_workload',{}).get('frames_per_s'), '| others', [(o.get('workload','')[:20], o.get('value')) for o in d.get('other_configs',[])])
PY
grep "NMS continuations" /root/repo/gpurun_out/r4B/b_$v.err | tail -1
done
