#!/bin/bash
# round 5: FP16 engines with the merged sibling layers and the fused heads -- parity, then config 3 twice
set -x
O=gpurun_out/r5g; mkdir -p $O
python -m pytest tests/test_gpu_network.py tests/test_gpu_random_plans.py tests/test_gpu_pipeline.py tests/test_gpu_host.py -x -q -m gpu -k "fp16 or FP16 or pairing or int8 or INT8" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --config 3 --dump-ops --no-cpu-baseline --no-extras > $O/cfg3.json 2> $O/cfg3.err
python bench.py --config 3 --no-cpu-baseline --no-extras > $O/cfg3_b.json 2> $O/cfg3_b.err
python bench.py --config 5 --no-cpu-baseline --no-extras > $O/cfg5.json 2> $O/cfg5.err
tail -c 1500 $O/cfg3.json; tail -c 400 $O/cfg3_b.json $O/cfg5.json
