#!/bin/bash
# correctness sweep + timing of the register-resident workgroup LU against the LDS-resident one (needs scripts/ubench/build_team_reg_lu.sh)
B=scripts/ubench/_build/team_reg_lu_bench
mkdir -p gpurun_out
{
for cfg in "128 120 64 dense" "128 128 64 dense" "128 121 64 dense" "128 127 32 ties" "128 125 33 sing" "128 128 32 dd" "120 120 64 dense" "120 113 64 dense" "120 117 64 ties" "120 119 33 sing" \
           "96 96 64 dense" "96 89 64 dense" "96 90 32 ties" "96 93 33 sing" "72 72 64 dense" "72 65 64 dense" "72 66 32 ties" "72 70 33 sing"; do
  set -- $cfg
  [ -x ${B}_$1 ] && { timeout 120 ${B}_$1 $2 $3 $4 || echo "   ^^^ FAILED ($cfg) rc=$?"; }
done
for cfg in "120 120 256 dd 8" "120 120 512 dd 8" "120 120 4096 dd 8" "128 128 512 dense 8" "128 128 4096 dense 8" "96 90 512 dd 8" "96 90 4096 dd 8" "72 72 4096 dd 8"; do
  set -- $cfg
  [ -x ${B}_$1 ] && { timeout 300 ${B}_$1 $2 $3 $4 $5 || echo "   ^^^ FAILED ($cfg) rc=$?"; }
done
} 2>&1 | tee gpurun_out/team_reg_lu.log
