#!/bin/bash
# builds variants of scripts/ubench/lu_tiled_bench with different layout parameters: build_lu_variants.sh name "flags" [name "flags" ...]
cd "$(dirname "$0")/../.."
mkdir -p scripts/ubench/_build
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-pass-failed $2 scripts/ubench/lu_tiled_bench.hip -o scripts/ubench/_build/lu_tiled_bench_$1 -save-temps=obj 2>&1 | grep -E "error" 
  echo "$1: $2"; grep -E "^\s+\.(vgpr_spill_count|private_segment_fixed_size)|\.name:" scripts/ubench/_build/lu_tiled_bench-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | sed 's/  */ /g' | grep factor
  shift 2
done
