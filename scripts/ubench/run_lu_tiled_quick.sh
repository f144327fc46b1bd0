#!/bin/bash
# quick correctness + timing of the matrix-core dense LU (needs scripts/ubench/_build/lu_tiled_bench)
B=${B:-scripts/ubench/_build/lu_tiled_bench}
mkdir -p gpurun_out
{
for cfg in "65 8 1 dense" "100 8 1 dense" "257 8 1 dense" "300 8 1 sing" "512 8 1 dense" "513 8 1 dense" "962 8 1 dense" "1000 8 1 dd"; do
  timeout 120 $B $cfg || echo "   ^^^ FAILED ($cfg) rc=$?"
done
for cfg in "512 4096 3 dense" "320 4096 3 dense" "962 256 3 dense" "1024 512 3 dense"; do
  timeout 300 $B $cfg || echo "   ^^^ FAILED ($cfg) rc=$?"
done
} 2>&1 | tee gpurun_out/lu_tiled_quick.log
