#!/bin/bash
# r06: a chunk followed by its own call's next chunk takes its queries in input order (no plan kernels ahead of it)
O=gpurun_out/r06_entry6; mkdir -p $O
export SGPU_TEST_HOOKS=1
(timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -4) > $O/tests.log
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 300 python tools/e2e_probe.py 2>&1 | grep qps >> $O/probe.txt; }
run A=0
run SGPU_PLAN_IDENTITY=0
run A=1
run SGPU_PLAN_IDENTITY=0
(timeout 400 python tools/shard_probe.py 2>&1 | grep -E "request thread|host phases") > $O/shard_probe.txt
(SGPU_PLAN_IDENTITY=0 timeout 400 python tools/shard_probe.py 2>&1 | grep -E "request thread|host phases") > $O/shard_probe_planned.txt
cat $O/tests.log $O/probe.txt
