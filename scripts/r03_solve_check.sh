# round 3: the streaming dense triangular solve (k_lu_solve_stream) against k_lu_solve_blocked: bitwise tests, then timing by ring depth (needs a GPU)
cd /root/repo
python -m pytest tests/test_gpu_lu_models.py tests/test_gpu_la_matrix.py -q -x -m gpu 2>&1 | tail -3
for args in "962 256 3" "512 4096 3" "200 4096 3" "1024 512 3" "100 16384 3"; do
  for d in 0 2 4 6; do
    echo -n "DSH_LU_STREAM_SOLVE=$d  "; DSH_LU_STREAM_SOLVE=$d DSH_LU_STRUCTURE=dense python scripts/lu_bench.py $args dense | tail -1
  done
done
