export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05last3; mkdir -p $O
timeout 130 python -m pytest tests/test_gpu_knn.py tests/test_gpu_build.py -q -m gpu -x > $O/knn_build.log 2>&1; tail -n 2 $O/knn_build.log
