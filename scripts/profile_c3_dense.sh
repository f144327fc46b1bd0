#!/bin/bash
# rocprofv3 kernel statistics of BASELINE config 3 in its literal "banded-as-dense LU" mode (DSH_LU_STRUCTURE=dense): bash scripts/profile_c3_dense.sh -> gpurun_out/r04_c3_dense_kernel_stats.md
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3d -o c3 -- python $R/scripts/run_configs.py --only heat_dense > /tmp/c3d.log 2>&1 < /dev/null
db=$(find /tmp/prof_c3d -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/scripts/top_kernels.py "$db" 16 > $R/gpurun_out/r04_c3_dense_kernel_stats.md; cat $R/gpurun_out/r04_c3_dense_kernel_stats.md; else echo "no database"; fi; tail -2 /tmp/c3d.log | cut -c1-400
