#!/bin/bash
O=gpurun_out/r5b; mkdir -p $O
{
for b in 2 4; do
  ./tools/heads_bench 45 147 $b 200
  for a in 1 32; do echo "ABL $a:"; ./tools/heads_bench_abl$a 45 147 $b 200; done
done
./tools/heads_bench 3 5 1 20
./tools/heads_bench 30 98 2 100
./tools/heads_bench 47 155 2 100
./tools/heads_bench 15 49 2 100
./tools/heads_bench 24 80 4 100
} 2>&1 | tee $O/heads3.log
