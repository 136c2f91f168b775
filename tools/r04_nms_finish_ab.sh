cd /tmp
for r in 1 2; do for v in 3 4 5; do
echo "nms_first $v: $(SPVO_TUNE_NMS_FIRST=$v python3 /root/repo/tools/sync_leg.py 300 0 2>/dev/null | grep 'depth 0' | cut -c1-120)"
done; done
for r in 1 2; do for v in 3 4; do
SPVO_TUNE_NMS_FIRST=$v python3 /root/repo/bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 > /tmp/b.log 2>/dev/null
python3 -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('nms_first $v:', d['value'], d['nms_host_continuations'], d['stages_ms']['nms'])"
done; done
