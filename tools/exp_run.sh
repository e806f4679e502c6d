export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05nd3v; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_coop.py tests/test_gpu_boundary.py tests/test_gpu_lookup_layouts.py tests/test_gpu_fuzz.py -q -m gpu -x > $O/coop_suite.log 2>&1; tail -n 2 $O/coop_suite.log
SGPU_TEST_HOOKS=1 timeout 400 python tools/latency_probe.py 8800000 > $O/latency.txt 2>&1; grep -E "nq=|sequential" $O/latency.txt
