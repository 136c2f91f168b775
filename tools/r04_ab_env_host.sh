# A/B of one tuning switch on the host-image legs (bench.py --legs host), interleaved: tools/r04_ab_env_host.sh NAME value value ...
N=$1; shift
cd /tmp
for r in 1 2 3; do for v in "$@"; do
env SPVO_TUNE_$N=$v python3 /root/repo/bench.py --no-cpu-baseline --legs host --steps 200 --warmup 20 > /tmp/b.log 2> /tmp/b.err
python3 -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);hi=d['host_interface'];print('$N $v: headline', d['value'], 'sync', hi['synchronous']['value'], 'lookahead', hi['lookahead']['value'])"
done; done
