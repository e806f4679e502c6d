#!/bin/bash
# r06: streaming stores for the staging copy (SGPU_STAGE_NT, a test hook), 24 batches in rotation (cold in the host's caches)
O=gpurun_out/r06_entry8; mkdir -p $O
export SGPU_TEST_HOOKS=1 E2E_BATCHES=24
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 200 python tools/e2e_probe.py 2>&1 | grep qps | sed 's/.*threads/threads/' >> $O/probe.txt; }
run SGPU_STAGE_NT=1
run SGPU_STAGE_NT=0
run SGPU_STAGE_NT=1
run SGPU_STAGE_NT=0
(timeout 300 python -m pytest tests/test_gpu_boundary.py -q -x 2>&1 | tail -2) > $O/tests.log
cat $O/probe.txt $O/tests.log
