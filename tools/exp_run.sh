export SGPU_TEST_HOOKS=1   # (the SGPU_* knobs these runs set are test hooks)
O=gpurun_out/r05smoke; mkdir -p $O
timeout 45 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/parity.log 2>&1; tail -n 1 $O/parity.log
