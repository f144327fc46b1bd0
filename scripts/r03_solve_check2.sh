cd /root/repo
for args in "962 256 3" "512 256 3"; do
  for cfg in "0 8" "2 8" "2 16" "3 16" "4 16" "2 32"; do
    set -- $cfg
    echo -n "DSH_LU_STREAM_SOLVE=$1 B=$2  "; DSH_LU_STREAM_SOLVE=$1 DSH_LU_STREAM_B=$2 DSH_LU_STRUCTURE=dense python scripts/lu_bench.py $args dense | tail -1
  done
done
DSH_LU_STREAM_B=16 python -m pytest tests/test_gpu_lu_models.py -q -x -m gpu 2>&1 | tail -2
DSH_LU_STREAM_B=32 DSH_LU_STREAM_SOLVE=2 python -m pytest tests/test_gpu_lu_models.py -q -x -m gpu 2>&1 | tail -2
