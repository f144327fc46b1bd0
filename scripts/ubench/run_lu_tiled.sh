#!/bin/bash
# correctness sweep + timing of the matrix-core dense LU (needs scripts/ubench/_build/lu_tiled_bench)
B=scripts/ubench/_build/lu_tiled_bench
mkdir -p gpurun_out
{
for cfg in "65 8 1 dense" "96 8 1 dense" "100 8 1 dense" "128 64 1 dense" "130 8 1 dense" "200 8 1 dd" "257 8 1 dense" "300 8 1 sing" "448 8 1 dense" "512 8 1 dense" "512 64 1 dd" \
           "513 8 1 dense" "640 8 1 dense" "962 8 1 dense" "1000 8 1 dd" "1024 8 1 dense"; do
  timeout 120 $B $cfg || echo "   ^^^ FAILED ($cfg) rc=$?"
done
for cfg in "512 4096 3 dense" "512 4096 3 dd" "256 4096 3 dense" "128 16384 3 dense" "962 256 3 dense" "1024 512 3 dense"; do
  timeout 300 $B $cfg || echo "   ^^^ FAILED ($cfg) rc=$?"
done
} 2>&1 | tee gpurun_out/lu_tiled.log
