#!/bin/bash
# Round-3 profile collection on the GPU box (writes under gpurun_out/prof_r03; tools/pmc_summary.py turns it into profiles/r03_*).
# PMC passes are separate from each other and carry only --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err
python3 $ROOT/tools/layer_roofline_json.py $OUT/layer_roofline.json > /dev/null 2> $OUT/layer_roofline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.log 2>&1
export WINO_DYNAMIC=1   # tools/wino_bench4: tiles through the XCD-banded counters, as the library launches the layer
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
$ROOT/tools/mfma_coissue > $OUT/mfma_coissue.log 2>&1
(for a in 0_f0 3_f0 11_f0; do echo "conv_wino64 ablation $a"; $ROOT/tools/wino_bench64_st$a 360 1176 64 64 1 20 247; done; echo "conv_wino2 (F(2x2) 8-wave form)"; $ROOT/tools/wino_bench2_st 360 1176 64 64 1 20 238) > $OUT/wino_stamps.log 2>&1
(for a in 0 1 2 4 6 15 31; do echo "conv_wino4 ablation $a (1 no input transform, 2 no filter staging, 4 no raw staging, 8 operands read once, 16 no stores)"; $ROOT/tools/wino_bench4_abl$a 360 1176 64 64 1 20 244; done) > $OUT/wino4_ablation.log 2>&1


$ROOT/tools/mfma_coissue16 > $OUT/mfma_coissue16.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 0 > $OUT/sync_leg.log 2>&1
python3 $ROOT/tools/sync_leg.py 300 2 >> $OUT/sync_leg.log 2>&1
du -sh $OUT
