mkdir -p gpurun_out/r4A; cd /tmp
for r in 1 2; do for v in 3 4 6; do
SPVO_TUNE_NMS_FIRST=$v SPVO_TUNE_TRUNK_TIMING=1 python3 /root/repo/bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 > /root/repo/gpurun_out/r4A/b_${v}_$r.log 2> /root/repo/gpurun_out/r4A/b_${v}_$r.err
python3 -c "
import json;d=json.loads(open('/root/repo/gpurun_out/r4A/b_${v}_$r.log').read().strip().splitlines()[-1]);print('nms_first $v:', d['value'], d['roofline']['avg_kernel_ms'])"
grep -A2 "trunk timing" /root/repo/gpurun_out/r4A/b_${v}_$r.err | tail -3 | cut -c1-200
done; done
