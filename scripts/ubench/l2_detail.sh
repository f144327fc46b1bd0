# phases of the two-workgroup layout of the matrix-core LU, chunked against column-dealt trailing phase, 4096 / 512 / 256 systems (profiles/r05_lu_layout2_detail.log)
D=scripts/ubench/_build
for nb in 4096 512 256; do
for bin in lu_tiled_old lu_tiled_ct3; do echo "#### $bin layout 2 nb=$nb"; DSH_LU_TILED_LAYOUT=2 timeout 200 $D/$bin 512 $nb 3 dense | grep -v "^layout"; done; done
echo "#### ct3 layout 1 nb=256"; DSH_LU_TILED_LAYOUT=1 timeout 200 $D/lu_tiled_ct3 512 256 3 dense
