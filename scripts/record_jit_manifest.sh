#!/bin/bash
# On a GPU box (gpurun): run bench.py and the `-m gpu` tests with DSH_JIT_RECORD set, so that every hiprtc module they request is written to a manifest.
# Copy gpurun_out/jit_manifest/*.rec to diffsol_amd/jit_manifest/ afterwards: __graft_entry__.build() replays them into the in-tree cache (no GPU needed).
#   gpurun --timeout 1500 -- 'bash scripts/record_jit_manifest.sh [bench|tests|all]'
set -u
what=${1:-all}
mkdir -p gpurun_out/jit_manifest
if [ "$what" = bench ] || [ "$what" = all ]; then
  rm -f gpurun_out/jit_manifest/bench.rec
  t0=$(date +%s.%N)
  DSH_JIT_RECORD=$PWD/gpurun_out/jit_manifest/bench.rec python bench.py --steps 20 --warmup 5 > gpurun_out/bench_record.out 2> gpurun_out/bench_record.err
  echo "bench rc=$? wall=$(echo "$(date +%s.%N) - $t0" | bc) s"; tail -1 gpurun_out/bench_record.out | head -c 4200; echo
fi
if [ "$what" = tests ] || [ "$what" = all ]; then
  rm -f gpurun_out/jit_manifest/tests.rec
  DSH_JIT_RECORD=$PWD/gpurun_out/jit_manifest/tests.rec timeout 1300 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
  echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
fi
ls -la gpurun_out/jit_manifest
