#!/bin/bash
# rocprofv3 kernel statistics of BASELINE config 3, bitwise (banded) and dense routes:  bash scripts/profile_c3_r05.sh  -> gpurun_out/r05/c3_{banded,dense}_kernel_stats.md
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
for route in banded dense; do
  rm -rf /tmp/prof_c3_$route
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3_$route -o c3 -- python $R/scripts/c3_once.py $route > /tmp/c3_$route.log 2>&1 < /dev/null
  tail -1 /tmp/c3_$route.log
  db=$(find /tmp/prof_c3_$route -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/scripts/top_kernels.py "$db" 16 > $R/gpurun_out/r05/c3_${route}_kernel_stats.md; head -12 $R/gpurun_out/r05/c3_${route}_kernel_stats.md; else echo "no database"; fi
done
