cd /root/repo
for args in "962 256 3" "1024 256 3" "512 256 3"; do
  for cfg in "512 2" "1024 2" "1024 3" "1024 4"; do
    set -- $cfg
    echo -n "threads=$1 depth=$2  "; DSH_LU_STREAM_THREADS=$1 DSH_LU_STREAM_SOLVE=$2 DSH_LU_STRUCTURE=dense python scripts/lu_bench.py $args dense | tail -1
  done
done
DSH_LU_STREAM_THREADS=1024 DSH_LU_STREAM_SOLVE=3 python -m pytest tests/test_gpu_lu_models.py -q -x -m gpu 2>&1 | tail -2
