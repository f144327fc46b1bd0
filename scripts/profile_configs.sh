#!/bin/bash
# PMC passes (round tag RND, default r05) over the device-resident kernels of BASELINE configs 4 and 5 (bench.py's `configs` rooflines):
#   bash scripts/profile_configs.sh c4_ode|c4_dae|c5_per_member|c5_group64 [nb]
# separate rocprofv3 --pmc passes (every one under its own timeout) + a kernel trace; summary in gpurun_out/r04/pmc_<cfg>.json; scripts/publish_configs_profile.py
# merges the summaries into profiles/r04_pmc_configs.json (stamped with the kernel-source hash bench.py checks).
set -u
export TMPDIR=/tmp
CFG=$1
NB=${2:-}
RND=${RND:-r05}
OUT=$PWD/gpurun_out/$RND
mkdir -p $OUT
P="python scripts/config_once.py $CFG $NB"
W=/tmp/prof_$CFG
rm -rf $W; mkdir -p $W
timeout 300 rocprofv3 --kernel-trace --stats -d $W/trace -o trace -- $P > $OUT/cfg_trace_$CFG.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $W/pmc1 -o pmc -- $P > $W/pmc1.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $W/pmc2 -o pmc -- $P > $W/pmc2.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W/pmc3 -o pmc -- $P > $W/pmc3.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $W/pmc4 -o pmc -- $P > $W/pmc4.log 2>&1 < /dev/null
TR=$(ls $W/trace/*/*_results.db $W/trace/*_results.db 2>/dev/null | head -1)
python scripts/pmc_summary.py --match k_bdf_lane_banded --match k_sdirk_resident --match k_bdf_adaptive --trace "$TR" --out $OUT/pmc_$CFG.json "$W/pmc1/*.db" "$W/pmc1/*/*.db" "$W/pmc2/*.db" "$W/pmc2/*/*.db" "$W/pmc3/*.db" "$W/pmc3/*/*.db" "$W/pmc4/*.db" "$W/pmc4/*/*.db" > $OUT/cfg_summary_$CFG.log 2>&1 < /dev/null
python scripts/top_kernels.py "$TR" 6 > $OUT/cfg_kernel_stats_$CFG.md 2>/dev/null
tail -2 $OUT/cfg_trace_$CFG.log
