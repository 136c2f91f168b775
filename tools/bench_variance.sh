# bench.py twice from a cold start and once after seconds of host work: the frame rate and the stage times side by side
show='import sys,json; d=json.loads(sys.stdin.read()); s=d["stages_ms"]; print(sys.argv[1], d["value"], "conv1b", d["roofline"]["avg_kernel_ms"], {k: s[k] for k in ("net","nms","match","solve","detect")})'
python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "$show" first
python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "$show" second
rm -f /tmp/spvo_synth_*.npz
python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "$show" no-cache
python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "$show" cache
python -c "
import time, numpy as np
t=time.time()
while time.time()-t < 4: np.random.rand(1000,1000) @ np.random.rand(1000,1000)
" 
python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "$show" after-other-process-busy
SPVO_BENCH_BURN=4 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "$show" burn-in-process
