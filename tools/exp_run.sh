export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05proxy; mkdir -p $O
SGPU_TEST_HOOKS=1 timeout 700 python tools/proxy_probe.py $O/proxy.npz > $O/proxy.log 2>&1; tail -n 3 $O/proxy.log
