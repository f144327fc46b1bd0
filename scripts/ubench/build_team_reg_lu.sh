#!/bin/bash
# builds scripts/ubench/_build/team_reg_lu_bench_<NL> (gfx950, the library's flags; NL = the compile-time bound on n: 128, 120, 96, 72) and prints the register /
# scratch usage of the two kernels
set -e
cd "$(dirname "$0")/../.."
mkdir -p scripts/ubench/_build
for NL in ${TRG_NLS:-128 120 96 72}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-value -Wno-pass-failed -Iinclude -DTRG_NL=$NL \
    scripts/ubench/team_reg_lu_bench.hip -o scripts/ubench/_build/team_reg_lu_bench_$NL -save-temps=obj
  echo "NL=$NL"; grep -E "^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|codeLenInByte)" scripts/ubench/_build/team_reg_lu_bench-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - | sed 's/  */ /g'
done
