#!/bin/bash
# Re-runs only the calibration passes of tools/collect_profiles.sh (copy kernel + LDS-DMA read kernel) into gpurun_out/prof_r02.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="$ROOT/tools/copy_bench 1024 3"
rm -rf $OUT/pmc_fetch_copy $OUT/pmc_write_copy
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_copy -o p -- $C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_copy -o p -- $C > /dev/null 2>&1
ls $OUT/pmc_fetch_copy
