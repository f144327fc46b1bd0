#!/bin/bash
# run_lu_variants.sh "A B C" : the timing configurations with every variant binary
for v in $1; do
  B=scripts/ubench/_build/lu_tiled_bench_$v
  echo "== variant $v"
  for cfg in "320 4096 3 dense" "384 4096 3 dense" "448 4096 3 dense" "512 4096 3 dense" "962 256 3 dense" "1024 512 3 dense"; do
    timeout 300 $B $cfg | grep -v "inside\|workgroups" || echo "   ^^^ FAILED ($cfg) rc=$?"
  done
done 2>&1 | tee gpurun_out/lu_variants.log
