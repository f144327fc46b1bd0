#!/bin/bash
# config 4 at one GPU's share of the 8-GPU job (32768 members: 2 wavefronts per CU, bound by the chain of one wavefront): the banded lane kernel's tuning knobs
NB=${NB:-32768}
for cfg in c4_ode c4_dae; do
for k in "3 4 1" "1 4 1" "2 4 1" "1 8 1" "1 4 2" "2 8 2" "1 8 2"; do
  set -- $k
  echo "== $cfg waves_per_eu=$1 unroll=$2 chunk_scale=$3"
  DSH_BANDED_WAVES_PER_EU=$1 DSH_LANE_BANDED_UNROLL=$2 DSH_LANE_BANDED_CHUNK_SCALE=$3 timeout 600 python scripts/config_once.py $cfg $NB 3 2>&1 | grep wall | tail -2 | cut -c1-120
done; done
