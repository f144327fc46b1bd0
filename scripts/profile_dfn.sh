#!/bin/bash
# rocprofv3 kernel statistics of the DFN solve (scripts/dfn_gpu.py gpu):  bash scripts/profile_dfn.sh <model.diffsl> <ref.npz> <nbatch>  -> gpurun_out/r02_dfn_kernel_stats_<nbatch>.md
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dfn
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_dfn -o dfn -- python $R/scripts/dfn_gpu.py gpu $R/$1 $R/$2 $3 > /tmp/dfn.log 2>&1 < /dev/null
tail -1 /tmp/dfn.log | cut -c1-300
db=$(find /tmp/prof_dfn -name "*.db" | head -1)
if [ -n "$db" ]; then python $R/scripts/top_kernels.py "$db" 14 > $R/gpurun_out/r04_dfn_kernel_stats_$3.md; cat $R/gpurun_out/r04_dfn_kernel_stats_$3.md; else echo "no database"; fi
