export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05last; mkdir -p $O
timeout 280 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
timeout 150 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/parity.log 2>&1; tail -n 2 $O/parity.log
