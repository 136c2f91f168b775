# In-pipeline tuning of the conv tile variants: bench.py with SPVO_CONV_FORCE="op:wr,wc,ck;..." per layer.
# Usage on the GPU box: bash tools/tune_variants.sh
run() { # label, force
  SPVO_CONV_FORCE="$2" python bench.py --steps 150 --warmup 30 --no-cpu-baseline --dump-ops 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['net_ops_ms']
print('$1'.ljust(26), d['value'], d['stages_ms']['net'], ' '.join('%s=%.1f'%(k.split(':')[1], v*1e3) for k,v in o.items()))"
}
run base ""
run "1:2,2,8" "1:2,2,8"

run "1:2,1,8" "1:2,1,8"

run "2:2,2,8 3:2,2,8" "2:2,2,8;3:2,2,8"

run "2:1,2,8 3:2,1,8" "2:1,2,8;3:2,1,8"

run "2:2,1,8 " "2:2,1,8"
run "4:2,2,8 5:2,2,8" "4:2,2,8;5:2,2,8"
run "4:1,2,8 5:2,1,8" "4:1,2,8;5:2,1,8"


run "6..9:1,1,8" "6:1,1,8;7:1,1,8;8:1,1,8;9:1,1,8"

run "6,7:1,2,8 8,9:2,1,8" "6:1,2,8;7:1,2,8;8:2,1,8;9:2,1,8"
run base2 ""
