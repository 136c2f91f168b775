#!/bin/bash
# Round-4 profile collection on the GPU box (writes under gpurun_out/prof_r04; the summaries are copied into profiles/r04_* by hand).
# PMC passes are separate from each other and carry only --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err
# per-layer timing (HIP events, nothing else on the chip): FP32 at both sizes of SURVEY 8d, FP16 at config 3's size and at the headline size
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline.json 360x1176 FP32 > /dev/null 2> $OUT/lr.err
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline_240x784.json 240x784 FP32 > /dev/null 2>> $OUT/lr.err
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline_fp16_192x640.json 192x640 FP16 > /dev/null 2>> $OUT/lr.err
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline_fp16_360x1176.json 360x1176 FP16 > /dev/null 2>> $OUT/lr.err
# per-kernel time of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.log 2>&1
# HBM traffic of every layer: forward-only loop, FETCH_SIZE and WRITE_SIZE in separate passes
for cfg in "vgg FP32 360x1176" "vgg FP16 192x640" "vgg FP16 360x1176"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$tag -o p -- python3 $ROOT/tools/forward_loop.py $cfg 30 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$tag -o p -- python3 $ROOT/tools/forward_loop.py $cfg 30 > /dev/null 2>&1
done
python3 $ROOT/tools/pmc_layers.py $OUT/pmc_fetch_vgg_FP32_360x1176 $OUT/pmc_write_vgg_FP32_360x1176 $OUT/layer_roofline.json $OUT/pmc_layers.json 30 > $OUT/pmc_layers.log 2>&1
python3 $ROOT/tools/pmc_layers.py $OUT/pmc_fetch_vgg_FP16_192x640 $OUT/pmc_write_vgg_FP16_192x640 $OUT/layer_roofline_fp16_192x640.json $OUT/pmc_layers_fp16_192x640.json 30 >> $OUT/pmc_layers.log 2>&1
python3 $ROOT/tools/pmc_layers.py $OUT/pmc_fetch_vgg_FP16_360x1176 $OUT/pmc_write_vgg_FP16_360x1176 $OUT/layer_roofline_fp16_360x1176.json $OUT/pmc_layers_fp16_360x1176.json 30 >> $OUT/pmc_layers.log 2>&1
# the dominant kernel, the matcher and the copy calibration as in round 3 (tools/pmc_summary.py reads these directories)
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
# the deep layers' shapes on the stand-alone kernel: matrix-pipe busy / wave-cycle split / LDS conflicts, and the compile-time ablation per shape
declare -A SHAPE=( [conv2a]="180 588 64 64 0 30 228" [conv2b]="180 588 64 64 1 30 228" [conv3a]="90 294 64 128 0 30 240" [conv3b]="90 294 128 128 1 30 240" [convPaDa]="45 147 128 512 0 30 240" )
for L in conv2a conv2b conv3a conv3b convPaDa; do
  dyn=0; case $L in conv2a|conv2b) dyn=1;; esac
  export WINO_DYNAMIC=$dyn
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/shape_sq_$L -o p -- $ROOT/tools/wino_bench4 ${SHAPE[$L]} > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/shape_sq2_$L -o p -- $ROOT/tools/wino_bench4 ${SHAPE[$L]} > /dev/null 2>&1
  (for a in 0 1 2 4 6 15 16 31; do echo "conv_wino4 ablation $a (1 no input transform, 2 no filter staging, 4 no raw staging, 8 operands read once, 16 no stores)"; $ROOT/tools/wino_bench4_abl$a ${SHAPE[$L]}; done) > $OUT/wino4_ablation_$L.log 2>&1
done
python3 $ROOT/tools/pmc_shapes.py $OUT $OUT/pmc_shapes.json > $OUT/pmc_shapes.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT $OUT/pmc.json > $OUT/pmc_summary.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 0 > $OUT/sync_leg.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 2 >> $OUT/sync_leg.log 2>&1
find $OUT -name "*.csv" -size +3M -delete   # raw traces stay on the box; the summaries above are what travels
du -sh $OUT
