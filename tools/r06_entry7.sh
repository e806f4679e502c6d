#!/bin/bash
# r06: with the first chunk unplanned, a smaller first chunk (shorter head of a one-thread call)?
O=gpurun_out/r06_entry7; mkdir -p $O
export SGPU_TEST_HOOKS=1
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 300 python tools/e2e_probe.py 2>&1 | grep qps >> $O/probe.txt; }
run A=0
run SGPU_CHUNK_FIRST=250
run SGPU_CHUNK_FIRST=350
run A=1
run SGPU_CHUNK_FIRST=250
run SGPU_CHUNK_FIRST=350
run SGPU_CHUNK_FIRST=150
cat $O/probe.txt
