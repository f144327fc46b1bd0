#!/bin/bash
# rocprofv3 kernel statistics of BASELINE config 3 (heat1d n = 512 x 4096, TR-BDF2, host-driven):  bash scripts/profile_c3.sh  -> gpurun_out/r04_c3_kernel_stats.md
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 -- python $R/scripts/run_configs.py --only heat > /tmp/c3.log 2>&1 < /dev/null
db=$(find /tmp/prof_c3 -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/scripts/top_kernels.py "$db" 16 > $R/gpurun_out/r04_c3_kernel_stats.md; cat $R/gpurun_out/r04_c3_kernel_stats.md; else echo "no database"; tail -5 /tmp/c3.log; fi
