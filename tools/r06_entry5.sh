#!/bin/bash
# r06: the closing run of the entry-point work: full GPU suite, request threads, shards, the bench line.
O=gpurun_out/r06_entry5; mkdir -p $O
(timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6) > $O/suite.log
export SGPU_TEST_HOOKS=1
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 300 python tools/e2e_probe.py 2>&1 | grep qps >> $O/probe.txt; }
run A=0
run A=1
(timeout 400 python tools/shard_probe.py 2>&1 | tail -30) > $O/shard_probe.txt
unset SGPU_TEST_HOOKS
(timeout 400 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
cat $O/suite.log $O/probe.txt
