#!/bin/bash
# two builds of scripts/ubench/lu_tiled_bench.hip on the SAME box (box-to-box spread is +-3 %): correctness at odd sizes, then 512 x 4096 / x 256 and 320 / 448 x 4096
D=scripts/ubench/_build
for b in ${BINS}; do echo "## $b"
for cfg in "65 8 1 dense" "100 8 1 dense" "130 8 1 dense" "257 8 1 dense" "300 8 1 sing" "400 64 1 dense" "496 8 1 dense" "512 64 1 dd" "512 8 1 dense"; do timeout 120 $D/$b $cfg | grep -E "^n=|mismatch" | sed 's/stage.*pivots/pivots/'; done
for rep in 1 2; do for nb in 4096 256; do timeout 200 $D/$b 512 $nb 3 dense | grep -v "^layout" | sed 's/pivots wrong/pw/; s/(with staging)//'; done; done
for n in 320 448; do timeout 200 $D/$b $n 4096 3 dense | grep -E "^n=" | sed 's/pivots wrong/pw/'; done
done
