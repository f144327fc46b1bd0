# the two workgroups of a CU started (b & 3) x T us apart (TL_STAGGER_US): profiles/r05_lu_stagger.log
D=scripts/ubench/_build
for t in 0 150 300 450 600; do echo "== TL_STAGGER_US=$t"; TL_STAGGER_US=$t timeout 200 $D/lu_tiled_bench 512 4096 3 dense | grep -E "^n=|phases" | sed 's/pivots wrong.*//'; done
