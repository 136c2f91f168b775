#!/bin/bash
# gpurun_out/prof_r06 (tools/collect_profiles_r06.sh) -> profiles/r06_*; the JSON files get the hash of the csrc/ they were collected on
set -u
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r06
for f in bench.log bench_200step.log bench_2cpu.log bench_under_rocprof.log layer_roofline.json layer_roofline_4img.json layer_roofline_fp16_192x640.json layer_roofline_int8.json \
         pmc_layers.json pmc_layers_fp16_192x640.json pmc_layers_int8.json pmc.json heads_bench.log sync_leg.log solve_chain_cfg3.log solve_order_ab.log long_sequence.log ate.log; do
  [ -s $S/$f ] && cp $S/$f profiles/r06_$f
done
st=$(find $S/stats -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" profiles/r06_kernel_stats.csv
python3 tools/csrc_hash.py profiles/r06_*.json
