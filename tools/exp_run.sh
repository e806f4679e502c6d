export SGPU_INDEX_CACHE=/tmp
O=gpurun_out/r05soak2; mkdir -p $O
SGPU_TEST_HOOKS=1 timeout 900 python tools/soak.py 446 646 fuzz-only > $O/soak_plain.log 2>&1; tail -n 1 $O/soak_plain.log
SGPU_TEST_HOOKS=1 SGPU_COOP=force timeout 600 python tools/soak.py 646 746 fuzz-only > $O/soak_coop.log 2>&1; tail -n 1 $O/soak_coop.log
