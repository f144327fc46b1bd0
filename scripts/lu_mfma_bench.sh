#!/bin/bash
# blocked dense LU at config 3's shape with and without the matrix-core trailing update, with the phase profile of workgroup 0
for m in 0 1; do
  DSH_LU_STRUCTURE=dense DSH_LU_MFMA=$m timeout 200 python scripts/lu_bench.py 512 4096 3 dense | tail -1
  DSH_LU_STRUCTURE=dense DSH_LU_MFMA=$m DSH_LU_PHASE_PROFILE=1 timeout 200 python scripts/lu_bench.py 512 64 1 dense 2>&1 | grep "phase us" | tail -1
done
