D=scripts/ubench/_build
for b in ${BINS}; do echo "## $b"; 
for cfg in "65 8 1 dense" "100 8 1 dense" "257 8 1 dense" "300 8 1 sing" "400 64 1 dense" "512 64 1 dd" "513 8 1 dense" "962 8 1 dense" "1000 8 1 dd"; do timeout 120 $D/$b $cfg | grep -E "^n=|mismatch"; done
for nb in 4096; do timeout 200 $D/$b 512 $nb 3 dense | grep -v "^layout"; done; 
timeout 200 $D/$b 320 4096 3 dense | grep -E "^n=|phases"
timeout 200 $D/$b 962 256 3 dense | grep -E "^n=|phases"
timeout 200 $D/$b 1024 512 3 dense | grep -E "^n=|phases"
done
