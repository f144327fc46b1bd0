#!/bin/bash
# Round-5 profiling of bench.py's timed region on the GPU box (run through gpurun from the repo root):
#   kernel trace of the bench command + separate --pmc passes over its one kernel (k_bdf_adaptive, wavefront lock-step).
# Outputs under gpurun_out/r05/ ; the summaries are copied to profiles/ by hand after inspection.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
TAG=${1:-a}
MODE=${MODE:-auto}  # MODE=member: the per-member kernel (every member its own step sizes) instead of the bench's wavefront lock-step groups
P="python scripts/bench_kernel_once.py 100000 3 $MODE"
if [ "$MODE" = auto ]; then rocprofv3 --kernel-trace --stats -d $OUT/trace_$TAG -o trace -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $OUT/trace_$TAG.log 2>&1 < /dev/null
else rocprofv3 --kernel-trace --stats -d $OUT/trace_$TAG -o trace -- $P > $OUT/trace_$TAG.log 2>&1 < /dev/null; fi
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1_$TAG -o pmc -- $P > $OUT/pmc1_$TAG.log 2>&1 < /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $OUT/pmc2_$TAG -o pmc -- $P > $OUT/pmc2_$TAG.log 2>&1 < /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3_$TAG -o pmc -- $P > $OUT/pmc3_$TAG.log 2>&1 < /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4_$TAG -o pmc -- $P > $OUT/pmc4_$TAG.log 2>&1 < /dev/null
TR=$(ls $OUT/trace_$TAG/*/*_results.db $OUT/trace_$TAG/*_results.db 2>/dev/null | head -1)
python scripts/pmc_summary.py --match k_bdf_adaptive --match k_bdf_member --trace "$TR" --out $OUT/pmc_resident_$TAG.json "$OUT/pmc1_$TAG/*.db" "$OUT/pmc1_$TAG/*/*.db" "$OUT/pmc2_$TAG/*.db" "$OUT/pmc2_$TAG/*/*.db" "$OUT/pmc3_$TAG/*.db" "$OUT/pmc3_$TAG/*/*.db" "$OUT/pmc4_$TAG/*.db" "$OUT/pmc4_$TAG/*/*.db" > $OUT/summary_$TAG.log 2>&1 < /dev/null
# the merge back from the GPU box is capped at 64 MiB: keep the summaries, drop the raw databases unless asked (KEEP_RAW=1)
python scripts/top_kernels.py "$TR" 6 > $OUT/kernel_stats_$TAG.md 2>/dev/null
if [ "${KEEP_RAW:-0}" != 1 ]; then rm -rf $OUT/trace_$TAG $OUT/pmc1_$TAG $OUT/pmc2_$TAG $OUT/pmc3_$TAG $OUT/pmc4_$TAG; fi
tail -5 $OUT/trace_$TAG.log
