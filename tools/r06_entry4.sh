#!/bin/bash
# r06: the entry point with the rewritten plan kernels (cost: eight loads in flight, registers by query_cut) and eight free
# slots left by a chunk that is followed by another: tests, request threads, shards, the bench line.
O=gpurun_out/r06_entry4; mkdir -p $O
export SGPU_TEST_HOOKS=1
(timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_fuzz.py tests/test_gpu_bench_multirank.py -q -x 2>&1 | tail -4) > $O/tests.log
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 300 python tools/e2e_probe.py 2>&1 | grep qps >> $O/probe.txt; }
run A=0
run SGPU_GRID_SPARE=0
run SGPU_CHUNK_MAX=3
run A=1
(timeout 400 python tools/shard_probe.py 2>&1 | tail -30) > $O/shard_probe.txt
unset SGPU_TEST_HOOKS
(timeout 400 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
cd /tmp && export TMPDIR=/tmp
for n in 10000 1250; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$n -- python /root/repo/tools/e2e_timeline_run.py $n > /root/repo/$O/timeline_run_$n.txt 2>&1
  f=$(find /tmp/tl_$n -name '*kernel_trace.csv' | head -1)
  python /root/repo/tools/e2e_timeline.py $f 24 > /root/repo/$O/timeline_$n.txt 2>&1
done
cat /root/repo/$O/tests.log /root/repo/$O/probe.txt
