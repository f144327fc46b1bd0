#!/bin/bash
# Round-4 occupancy experiment of the headline kernel: variants of libdiffsol_hip.so whose device-resident BDF is compiled for 3 / 4
# wavefronts per SIMD with D's swap partner in per-lane memory instead of LDS.   build:  bash scripts/occupancy_experiment.sh build
#                                                                                  run:    bash scripts/occupancy_experiment.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
CS=diffsol_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function"
if [ "$1" = build ]; then
  for v in "2 1" "3 1" "4 1"; do
    set -- $v
    d=diffsol_amd/lib_exp/w$1_p$2; mkdir -p $d
    ( /opt/rocm/bin/hipcc $FL -DDSH_ADAPTIVE_WAVES_PER_EU=$1 -DDSH_ADAPTIVE_DT_PRIVATE=$2 -c $CS/dsh_adaptive.hip -o $d/dsh_adaptive.o 2> $d/build.log &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdiffsol_hip.so $d/dsh_adaptive.o $(ls diffsol_amd/lib/obj/*.o | grep -v "/dsh_adaptive.o") -L/opt/rocm/lib -lhiprtc -ldl &&
      cp diffsol_amd/lib/libdiffsol_hip_host.so $d/ && rm $d/dsh_adaptive.o && echo "built $d" ) &
  done
  wait
else
  mkdir -p gpurun_out/r04
  for d in "" diffsol_amd/lib_exp/w2_p1 diffsol_amd/lib_exp/w3_p1 diffsol_amd/lib_exp/w4_p1; do
    echo "== ${d:-default (2 waves per SIMD, swap partner in LDS)}"
    DSH_LIB_DIR=${d:+$PWD/$d} python bench.py --no-cpu-baseline --no-configs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'per_member_ms': d['per_member']['ms_per_step'], 'large_ensemble_steps_per_s': d['large_ensemble']['ode_steps_per_sec'], 'large_kernel_ms': d['large_ensemble']['kernel_ms'], 'failed': d['checks']['failed_members']}))"
  done 2>&1 | tee gpurun_out/r04/occupancy.log
fi
