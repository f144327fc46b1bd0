# pitch of the row-major working copy of the matrix-core LU + 0 .. 80 doubles: memory-channel conflicts of a 4 KB pitch? (profiles/r05_lu_ldw_pad.log: none)
B=scripts/ubench/_build/lu_tiled_bench
for pad in 0 16 32 64 80; do echo "== pad $pad"; DSH_TL_LDW_PAD=$pad timeout 200 $B 512 4096 3 dense | head -3; done
for pad in 0 16; do echo "== pad $pad 962"; DSH_TL_LDW_PAD=$pad timeout 200 $B 962 256 3 dense | head -3;  DSH_TL_LDW_PAD=$pad timeout 200 $B 1024 512 3 dense | head -2; done
