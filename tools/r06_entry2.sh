#!/bin/bash
# r06: the plan kernels rewritten (cost: registers + one atomic per wavefront; rank instead of a one-workgroup bitonic sort) and
# two workgroup slots left free by a device-planned chunk: tests, one / two request threads, shards, device timeline.
O=gpurun_out/r06_entry2; mkdir -p $O
export SGPU_TEST_HOOKS=1
(timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -4) > $O/tests.log
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 300 python tools/e2e_probe.py 2>&1 | grep qps >> $O/probe.txt; }
run A=0
run SGPU_GRID_SPARE=0
run SGPU_GRID_SPARE=4
run SGPU_CHUNK_FIRST=350
run SGPU_CHUNK_MAX=3
run A=1
(timeout 400 python tools/shard_probe.py 2>&1 | tail -30) > $O/shard_probe.txt
cd /tmp && export TMPDIR=/tmp
for n in 10000 1250; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$n -- python /root/repo/tools/e2e_timeline_run.py $n > /root/repo/$O/timeline_run_$n.txt 2>&1
  f=$(find /tmp/tl_$n -name '*kernel_trace.csv' | head -1)
  python /root/repo/tools/e2e_timeline.py $f 24 > /root/repo/$O/timeline_$n.txt 2>&1
done
cat /root/repo/$O/tests.log /root/repo/$O/probe.txt
