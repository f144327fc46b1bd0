D=scripts/ubench/_build
for b in ${BINS}; do echo "## $b"; 
for cfg in "65 8 1 dense" "100 8 1 dense" "128 8 1 dense" "130 8 1 dense" "192 8 1 dense" "200 8 1 dd" "257 8 1 dense" "300 8 1 sing" "320 8 1 dense" "384 8 1 dense" "400 64 1 dense" "448 8 1 dense" "512 64 1 dd" "512 8 1 dense"; do timeout 120 $D/$b $cfg | grep -E "^n=|mismatch"; done
for nb in 4096 256; do timeout 200 $D/$b 512 $nb 3 dense | grep -v "^layout"; done; 
timeout 200 $D/$b 320 4096 3 dense | grep -E "^n=|phases"
timeout 200 $D/$b 448 4096 3 dense | grep -E "^n=|phases"
done
