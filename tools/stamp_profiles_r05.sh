#!/bin/bash
# gpurun_out/prof_r05 (tools/collect_profiles_r05.sh) -> profiles/r05_*; the JSON files get the hash of the csrc/ they were collected on
set -u
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r05
for f in bench.log bench_200step.log bench_2cpu.log bench_under_rocprof.log layer_roofline.json layer_roofline_4img.json layer_roofline_fp16_192x640.json layer_roofline_int8.json \
         pmc_layers.json pmc_layers_fp16_192x640.json pmc_layers_int8.json pmc.json heads_bench.log sync_leg.log solve_chain_cfg3.log; do
  [ -s $S/$f ] && cp $S/$f profiles/r05_$f
done
st=$(find $S/stats -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" profiles/r05_kernel_stats.csv
python3 tools/csrc_hash.py profiles/r05_*.json
