#!/bin/bash
# Round-6 profile collection on the GPU box (writes under gpurun_out/prof_r06; tools/stamp_profiles_r06.sh copies the summaries into profiles/r06_*
# and stamps the JSON files with the hash of csrc/).  PMC passes are separate from each other and carry only --kernel-trace
# (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err      # the driver's command
python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_200step.log 2> $OUT/bench_200step.err
python3 $ROOT/bench.py --no-cpu-baseline --no-extras --cpus 2 > $OUT/bench_2cpu.log 2> $OUT/bench_2cpu.err   # one rank's share of an 8-rank node's CPUs
# per-layer timing (HIP events, nothing else on the chip)
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline.json 360x1176 FP32 > /dev/null 2> $OUT/lr.err
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline_4img.json 360x1176 FP32 4 > /dev/null 2>> $OUT/lr.err
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline_fp16_192x640.json 192x640 FP16 > /dev/null 2>> $OUT/lr.err
python3 $ROOT/tools/layer_roofline_int8.py $OUT/layer_roofline_int8.json mbv1 360x1176 2 > $OUT/lr_int8.log 2>> $OUT/lr.err
# per-kernel time of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.log 2>&1
# HBM traffic of every layer: forward-only loop, FETCH_SIZE and WRITE_SIZE in separate passes
for cfg in "vgg FP32 360x1176" "vgg FP16 192x640" "mbv1 INT8 360x1176"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$tag -o p -- python3 $ROOT/tools/forward_loop.py $cfg 30 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$tag -o p -- python3 $ROOT/tools/forward_loop.py $cfg 30 > /dev/null 2>&1
done
python3 $ROOT/tools/pmc_layers.py $OUT/pmc_fetch_vgg_FP32_360x1176 $OUT/pmc_write_vgg_FP32_360x1176 $OUT/layer_roofline.json $OUT/pmc_layers.json 30 > $OUT/pmc_layers.log 2>&1
python3 $ROOT/tools/pmc_layers.py $OUT/pmc_fetch_vgg_FP16_192x640 $OUT/pmc_write_vgg_FP16_192x640 $OUT/layer_roofline_fp16_192x640.json $OUT/pmc_layers_fp16_192x640.json 30 >> $OUT/pmc_layers.log 2>&1
python3 $ROOT/tools/pmc_layers.py $OUT/pmc_fetch_mbv1_INT8_360x1176 $OUT/pmc_write_mbv1_INT8_360x1176 $OUT/layer_roofline_int8.json $OUT/pmc_layers_int8.json 30 >> $OUT/pmc_layers.log 2>&1
# the dominant kernel, the matcher and the copy calibration (tools/pmc_summary.py reads these directories)
export WINO_DYNAMIC=1
W="$ROOT/tools/wino_bench4 360 1176 64 64 1 20 244"
M="$ROOT/tools/match_bench 1000 2 50"
C="$ROOT/tools/copy_bench 1024 3"
for prog in wino match copy; do
  case $prog in wino) CMD=$W;; match) CMD=$M;; copy) CMD=$C;; esac
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$prog -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$prog -o p -- $CMD > /dev/null 2>&1
done
for prog in wino match; do
  case $prog in wino) CMD=$W;; match) CMD=$M;; esac
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_$prog -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2_$prog -o p -- $CMD > /dev/null 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT $OUT/pmc.json > $OUT/pmc_summary.log 2>&1
# the heads kernel alone (two / four images), self-checked against a float64 host evaluation
$ROOT/tools/heads_bench 45 147 2 > $OUT/heads_bench.log 2>&1
$ROOT/tools/heads_bench 45 147 4 >> $OUT/heads_bench.log 2>&1
bash $ROOT/tools/r05_cfg3_trace.sh 3 > /dev/null 2>&1; cp $ROOT/gpurun_out/r5p/solve_chain.log $OUT/solve_chain_cfg3.log   # the solver chain inside config 3's loop (kernel trace)
cd $ROOT && bash tools/r06_solve_order_ab.sh > $OUT/solve_order_ab.log 2>&1; cd /tmp   # two solves in flight against rounds 1-5's order, same box
cd $ROOT && python3 -m pytest tests/test_gpu_long_sequence.py -q -m gpu -s > $OUT/long_sequence_pytest.log 2>&1; cp gpurun_out/long_sequence.log $OUT/long_sequence.log; cd /tmp
python3 $ROOT/tools/ate_eval.py 40 > $OUT/ate.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 0 > $OUT/sync_leg.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 2 >> $OUT/sync_leg.log 2>&1
find $OUT -name "*.csv" -size +3M -delete   # raw traces stay on the box; the summaries above are what travels
du -sh $OUT
