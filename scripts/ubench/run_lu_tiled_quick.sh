#!/bin/bash
B=scripts/ubench/_build/lu_tiled_bench
mkdir -p gpurun_out
{
for cfg in "100 8 1 dense" "130 8 1 dense" "300 8 1 sing" "512 8 1 dense" "513 8 1 dense" "962 8 1 dense"; do
  timeout 120 $B $cfg || echo "   ^^^ FAILED ($cfg) rc=$?"
done
for cfg in "512 4096 3 dense" "512 4096 3 dd" "128 16384 3 dense" "962 256 3 dense"; do
  timeout 300 $B $cfg || echo "   ^^^ FAILED ($cfg) rc=$?"
done
} 2>&1 | tee gpurun_out/lu_tiled_quick.log
