#!/bin/bash
# every compile-time bound NL = 40 .. 128 of the register-resident workgroup LU at every n it serves (NL - 7 .. NL), matrix kinds in rotation (dense: an interchange per
# pivot; ties: small integers, equal magnitudes and exact zeros; sing: zero columns; dd: no interchange): bits against the LDS-resident form
# (needs TRG_NLS="40 48 56 64 72 80 88 96 104 112 120 128" scripts/ubench/build_team_reg_lu.sh)
B=scripts/ubench/_build/team_reg_lu_bench
mkdir -p gpurun_out
kinds=(dense ties sing dd)
i=0
{
for NL in 40 48 56 64 72 80 88 96 104 112 120 128; do
  [ -x ${B}_$NL ] || { echo "missing ${B}_$NL"; continue; }
  for n in $(seq $((NL - 7)) $NL); do
    k=${kinds[$((i % 4))]}; i=$((i + 1))
    out=$(timeout 120 ${B}_$NL $n 24 $k 3 2>&1 | grep "same bits")
    echo "NL=$NL n=$n $k: $out"
  done
done
} 2>&1 | tee gpurun_out/team_reg_lu_sweep.log | grep -c "factors yes, permutation yes, singular flags yes.*solutions yes"
