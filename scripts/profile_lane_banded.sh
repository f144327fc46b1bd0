#!/bin/bash
# PMC passes (every one under its own timeout: a pass that over-subscribes a counter block aborts and then hangs) over the banded lane-per-member BDF (config 4: spm n = 42, 262 144 members): bash scripts/profile_lane_banded.sh <tag> [env assignments]
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${ROUND:-r03}
mkdir -p $OUT
TAG=${1:-a}
P="python scripts/spm_resident_once.py 262144 ${MODEL:-}"  # MODEL=dae: the singular-mass formulation
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/lb_trace_$TAG -o trace -- $P > $OUT/lb_trace_$TAG.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT -d $OUT/lb_pmc1_$TAG -o pmc -- $P > $OUT/lb_pmc1_$TAG.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_INSTS_LDS -d $OUT/lb_pmc2_$TAG -o pmc -- $P > $OUT/lb_pmc2_$TAG.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/lb_pmc3_$TAG -o pmc -- $P > $OUT/lb_pmc3_$TAG.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/lb_pmc4_$TAG -o pmc -- $P > $OUT/lb_pmc4_$TAG.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $OUT/lb_pmc5_$TAG -o pmc -- $P > $OUT/lb_pmc5_$TAG.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/lb_pmc6_$TAG -o pmc -- $P > $OUT/lb_pmc6_$TAG.log 2>&1 < /dev/null
TR=$(ls $OUT/lb_trace_$TAG/*/*_results.db $OUT/lb_trace_$TAG/*_results.db 2>/dev/null | head -1)
python scripts/pmc_summary.py --match k_bdf_lane_banded --match k_bdf_adaptive --trace "$TR" --out $OUT/pmc_lane_banded_$TAG.json "$OUT/lb_pmc1_$TAG/*.db" "$OUT/lb_pmc1_$TAG/*/*.db" "$OUT/lb_pmc2_$TAG/*.db" "$OUT/lb_pmc2_$TAG/*/*.db" "$OUT/lb_pmc3_$TAG/*.db" "$OUT/lb_pmc3_$TAG/*/*.db" "$OUT/lb_pmc4_$TAG/*.db" "$OUT/lb_pmc4_$TAG/*/*.db" "$OUT/lb_pmc5_$TAG/*.db" "$OUT/lb_pmc5_$TAG/*/*.db" "$OUT/lb_pmc6_$TAG/*.db" "$OUT/lb_pmc6_$TAG/*/*.db" > $OUT/lb_summary_$TAG.log 2>&1
tail -3 $OUT/lb_summary_$TAG.log
