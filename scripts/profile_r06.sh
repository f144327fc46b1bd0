#!/bin/bash
# Round-6 profiling on the GPU box (run through gpurun from the repo root): everything bench.py's rooflines read, in one call.
#   bash scripts/profile_r06.sh [headline] [member] [configs]        (no argument: all three)
#   headline: kernel trace of the bench command + separate --pmc passes over its one kernel (k_bdf_adaptive, wavefront lock-step)   -> gpurun_out/r06/pmc_resident_a.json
#   member:   the same passes over the per-member kernel (bench.py's `per_member` extra)                                            -> gpurun_out/r06/pmc_resident_m.json
#   configs:  scripts/profile_configs.sh for c4_ode / c4_dae (262144 members) and c5_per_member / c5_group64 (65536)                -> gpurun_out/r06/pmc_<cfg>.json
# Afterwards, HERE (the summaries are stamped with the hash of the local kernel sources — the same tree):
#   python scripts/publish_profile.py a r06; python scripts/publish_profile.py m r06 member
#   RND=r06 python scripts/publish_configs_profile.py c4_ode:262144 c4_dae:262144 c5_per_member:65536 c5_group64:65536
# tests/test_profiles_fresh.py fails the CPU tier while any of the published summaries is older than the kernel sources it describes.
set -u
export TMPDIR=/tmp
export RND=r06
OUT=$PWD/gpurun_out/$RND
mkdir -p $OUT
WHAT="${*:-headline member configs}"
passes() {  # tag mode
  local TAG=$1 MODE=$2
  local P="python scripts/bench_kernel_once.py 100000 3 $MODE"
  local W=/tmp/prof_$TAG; rm -rf $W; mkdir -p $W
  if [ "$MODE" = auto ]; then timeout 600 rocprofv3 --kernel-trace --stats -d $W/trace -o trace -- python bench.py --no-cpu-baseline --no-extras --no-configs --steps 5 --warmup 2 > $OUT/trace_$TAG.log 2>&1 < /dev/null
  else timeout 300 rocprofv3 --kernel-trace --stats -d $W/trace -o trace -- $P > $OUT/trace_$TAG.log 2>&1 < /dev/null; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $W/pmc1 -o pmc -- $P > $W/pmc1.log 2>&1 < /dev/null
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $W/pmc2 -o pmc -- $P > $W/pmc2.log 2>&1 < /dev/null
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W/pmc3 -o pmc -- $P > $W/pmc3.log 2>&1 < /dev/null
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $W/pmc4 -o pmc -- $P > $W/pmc4.log 2>&1 < /dev/null
  local TR=$(ls $W/trace/*/*_results.db $W/trace/*_results.db 2>/dev/null | head -1)
  python scripts/pmc_summary.py --match k_bdf_adaptive --match k_bdf_member --trace "$TR" --out $OUT/pmc_resident_$TAG.json "$W/pmc1/*.db" "$W/pmc1/*/*.db" "$W/pmc2/*.db" "$W/pmc2/*/*.db" "$W/pmc3/*.db" "$W/pmc3/*/*.db" "$W/pmc4/*.db" "$W/pmc4/*/*.db" > $OUT/summary_$TAG.log 2>&1 < /dev/null
  python scripts/top_kernels.py "$TR" 6 > $OUT/kernel_stats_$TAG.md 2>/dev/null
  tail -2 $OUT/trace_$TAG.log | cut -c1-600
}
for w in $WHAT; do
  case $w in
    headline) passes a auto ;;
    member) passes m member ;;
    configs) for c in c5_per_member c5_group64 c4_ode c4_dae; do bash scripts/profile_configs.sh $c; done ;;
  esac
done
ls -la $OUT | tail -30
