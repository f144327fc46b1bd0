#!/bin/bash
# PMC passes (each under its own timeout) over k_lu_band_solve_wide<1,8> alone, n = 512 x 4096 (scripts/ubench/band_wide_bench.hip):  bash scripts/profile_band_wide.sh
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r02
mkdir -p $OUT
P="scripts/ubench/_build/band_wide_bench 4096"
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/bw_trace -o trace -- $P > $OUT/bw_trace.log 2>&1 < /dev/null
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $OUT/bw_pmc1 -o pmc -- $P > $OUT/bw_pmc1.log 2>&1 < /dev/null
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD -d $OUT/bw_pmc2 -o pmc -- $P > $OUT/bw_pmc2.log 2>&1 < /dev/null
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/bw_pmc3 -o pmc -- $P > $OUT/bw_pmc3.log 2>&1 < /dev/null
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/bw_pmc4 -o pmc -- $P > $OUT/bw_pmc4.log 2>&1 < /dev/null
TR=$(ls $OUT/bw_trace/*/*_results.db $OUT/bw_trace/*_results.db 2>/dev/null | head -1)
python scripts/pmc_summary.py --match k_lu_band_solve_wide --trace "$TR" --out $OUT/pmc_band_wide.json "$OUT/bw_pmc1/*.db" "$OUT/bw_pmc1/*/*.db" "$OUT/bw_pmc2/*.db" "$OUT/bw_pmc2/*/*.db" "$OUT/bw_pmc3/*.db" "$OUT/bw_pmc3/*/*.db" "$OUT/bw_pmc4/*.db" "$OUT/bw_pmc4/*/*.db" > $OUT/bw_summary.log 2>&1 < /dev/null
tail -3 $OUT/bw_summary.log
cat $OUT/pmc_band_wide.json 2>/dev/null | head -60
