D=scripts/ubench/_build
for b in ${BINS}; do echo "## $b"; for cfg in "300 8 1 sing" "400 64 1 dense" "512 8 1 dense"; do timeout 120 $D/$b $cfg | grep -E "^n=|mismatch" | sed 's/stage.*pivots/pivots/'; done; for nb in 4096 256; do timeout 200 $D/$b 512 $nb 3 dense | grep -v "^layout" | sed 's/(with staging) //'; done; done
