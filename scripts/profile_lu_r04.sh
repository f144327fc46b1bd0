#!/bin/bash
# Counters (rounds 4, 5: RND=r05) of the matrix-core dense LU (k_lu_factor_tiled, dsh_lu_tiled.hpp) at 512 x 4096 and 962 x 256: kernel trace + separate --pmc passes
# (SQ wait / active split, instruction mix, FETCH_SIZE, WRITE_SIZE).  Run through gpurun from the repo root; outputs under gpurun_out/r04/lu_*.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${RND:-r04}
mkdir -p $OUT
for cfg in "512 4096" "962 256"; do
  set -- $cfg; N=$1; NB=$2; TAG=lu_${N}
  P="python scripts/lu_bench.py $N $NB 3 dense"
  export DSH_LU_STRUCTURE=dense
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$TAG -o trace -- $P > $OUT/trace_$TAG.log 2>&1 < /dev/null
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc1_$TAG -o pmc -- $P > $OUT/pmc1_$TAG.log 2>&1 < /dev/null
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS -d $OUT/pmc2_$TAG -o pmc -- $P > $OUT/pmc2_$TAG.log 2>&1 < /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3_$TAG -o pmc -- $P > $OUT/pmc3_$TAG.log 2>&1 < /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4_$TAG -o pmc -- $P > $OUT/pmc4_$TAG.log 2>&1 < /dev/null
  TR=$(ls $OUT/trace_$TAG/*/*_results.db $OUT/trace_$TAG/*_results.db 2>/dev/null | head -1)
  python scripts/pmc_summary.py --match k_lu_factor_tiled --match k_lu_stage --trace "$TR" --out $OUT/pmc_$TAG.json "$OUT/pmc1_$TAG/*.db" "$OUT/pmc1_$TAG/*/*.db" "$OUT/pmc2_$TAG/*.db" "$OUT/pmc2_$TAG/*/*.db" "$OUT/pmc3_$TAG/*.db" "$OUT/pmc3_$TAG/*/*.db" "$OUT/pmc4_$TAG/*.db" "$OUT/pmc4_$TAG/*/*.db" > $OUT/summary_$TAG.log 2>&1 < /dev/null
  tail -2 $OUT/trace_$TAG.log; tail -3 $OUT/summary_$TAG.log; tail -2 $OUT/pmc2_$TAG.log
done
