#!/bin/bash
# A/B of the round-5 pieces of the matrix-core LU: lu_tiled_old = r04 (chunked trailing phase, stage from W, direct finish), lu_tiled_bench = all three on,
# lu_tiled_nofuse / lu_tiled_nofin = one of them off
D=scripts/ubench/_build
for bin in ${BINS:-lu_tiled_old lu_tiled_bench lu_tiled_nofuse lu_tiled_nofin}; do
  echo "######## $bin"
  for cfg in "65 8 1 dense" "100 8 1 dense" "257 8 1 dense" "300 8 1 sing" "513 8 1 dense" "962 8 1 dense" "1000 8 1 dd"; do
    timeout 120 $D/$bin $cfg | grep -E "^n=|mismatch" || echo "   ^^^ FAILED ($cfg)"
  done
  for lay in 1 2; do
    echo "== layout $lay"
    DSH_LU_TILED_LAYOUT=$lay timeout 200 $D/$bin 512 4096 3 dense | grep -v "^layout"
    DSH_LU_TILED_LAYOUT=$lay timeout 200 $D/$bin 320 4096 3 dense | grep -E "^n=|phases"
  done
  timeout 200 $D/$bin 962 256 3 dense | grep -E "^n=|phases"
  timeout 200 $D/$bin 1024 512 3 dense | grep -E "^n=|phases"
done
