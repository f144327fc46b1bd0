#!/bin/bash
# builds scripts/ubench/_build/lu_tiled_bench (gfx950) and prints the register / scratch usage of the kernels
set -e
cd "$(dirname "$0")/../.."
mkdir -p scripts/ubench/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function -Wno-pass-failed scripts/ubench/lu_tiled_bench.hip -o scripts/ubench/_build/lu_tiled_bench -save-temps=obj
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|vgpr_spill_count|group_segment_fixed_size)|\.name:" scripts/ubench/_build/lu_tiled_bench-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - | sed 's/  */ /g'
