# A/B of one tuning switch on one box, interleaved: tools/r04_ab_env.sh NAME value value ... (bench.py headline leg, 200-step blocks)
N=$1; shift
mkdir -p gpurun_out/abenv; cd /tmp
for r in 1 2 3; do for v in "$@"; do
env SPVO_TUNE_$N=$v SPVO_TUNE_TRUNK_TIMING=1 python3 /root/repo/bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 > /root/repo/gpurun_out/abenv/b_${v}_$r.log 2> /root/repo/gpurun_out/abenv/b_${v}_$r.err
python3 -c "
import json;d=json.loads(open('/root/repo/gpurun_out/abenv/b_${v}_$r.log').read().strip().splitlines()[-1]);print('$N $v:', d['value'], d['roofline']['avg_kernel_ms'])"
grep -A0 "trunk timing" /root/repo/gpurun_out/abenv/b_${v}_$r.err | tail -1 | cut -c1-200
done; done
