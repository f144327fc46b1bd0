# round 4: tests of the dense LU paths + library-level timing of the matrix-core kernel against the exact one (needs a GPU)
cd /root/repo
python -m pytest tests/test_gpu_lu_models.py -q -x -m gpu -k "matrix_core or bitwise or singular" 2>&1 | tail -3
python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "config3" 2>&1 | tail -3
for args in "320 4096 3" "512 4096 3" "768 1024 3" "962 256 3" "1024 512 3"; do
  DSH_LU_STRUCTURE=dense python scripts/lu_bench.py $args dense | tail -1
  DSH_LU_STRUCTURE=dense DSH_LU_EXACT=1 python scripts/lu_bench.py $args dense | tail -1
done
DSH_LU_STRUCTURE=dense python scripts/lu_bench.py 512 4096 3 tri | tail -1
