#!/bin/bash
# Round-2 profile collection on the GPU box (writes under gpurun_out/prof_r02; tools/pmc_summary.py turns it into profiles/r02_*).
# PMC passes are separate from each other and carry only --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench.log 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.log 2>&1
export WINO_DYNAMIC=1   # tools/wino_bench2: tiles through the XCD-banded counters, as the library launches the layer
W="$ROOT/tools/wino_bench2 360 1176 64 64 1 20 238"
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
find $OUT -name "*.csv" | head -40
du -sh $OUT
