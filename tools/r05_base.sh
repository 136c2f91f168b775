#!/bin/bash
# round 5, first GPU call: where the time goes today (headline, 2-CPU host budget, configs 3 and 5 per op)
set -x
O=gpurun_out/r5a; mkdir -p $O
python bench.py --no-cpu-baseline --no-extras --dump-ops > $O/head.json 2> $O/head.err
python bench.py --no-cpu-baseline --no-extras --cpus 2 > $O/head_2cpu.json 2> $O/head_2cpu.err
python bench.py --no-cpu-baseline --no-extras --cpus 1 > $O/head_1cpu.json 2> $O/head_1cpu.err
python bench.py --no-cpu-baseline --no-extras > $O/head_b.json 2> $O/head_b.err
python bench.py --config 5 --dump-ops --no-cpu-baseline --no-extras > $O/cfg5.json 2> $O/cfg5.err
python bench.py --config 5 --dump-ops --no-cpu-baseline --no-extras --depth 2 > $O/cfg5_d2.json 2> $O/cfg5_d2.err
python bench.py --config 3 --dump-ops --no-cpu-baseline --no-extras > $O/cfg3.json 2> $O/cfg3.err
python bench.py --config 3 --dump-ops --no-cpu-baseline --no-extras --depth 2 > $O/cfg3_d2.json 2> $O/cfg3_d2.err
python tools/perop_int8.py mbv1 > $O/perop_int8.log 2>&1
python tools/perop.py vgg FP16 192x640 > $O/perop_f16.log 2>&1 || true
tail -c 600 $O/*.json
