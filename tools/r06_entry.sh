#!/bin/bash
# r06: what is left between a one-thread sgpu_batch_search call and its kernel: free slots for the next chunk's plan kernels
# (SGPU_GRID_SPARE), a smaller first chunk (SGPU_CHUNK_FIRST), and the device timeline of the calls (rocprofv3 --kernel-trace).
O=gpurun_out/r06_entry; mkdir -p $O
export SGPU_TEST_HOOKS=1
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 300 python tools/e2e_probe.py 2>&1 | grep qps >> $O/probe.txt; }
run A=0
run SGPU_GRID_SPARE=1
run SGPU_GRID_SPARE=2
run SGPU_GRID_SPARE=4
run SGPU_GRID_SPARE=2 SGPU_CHUNK_FIRST=300
run SGPU_GRID_SPARE=2 SGPU_CHUNK_FIRST=400
run SGPU_CHUNK_FIRST=300
run A=1
cd /tmp && export TMPDIR=/tmp
for v in 0 2; do
  SGPU_GRID_SPARE=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -- python /root/repo/tools/e2e_timeline_run.py > /root/repo/$O/timeline_run_$v.txt 2>&1
  f=$(find /tmp/tl_$v -name '*kernel_trace.csv' | head -1)
  python /root/repo/tools/e2e_timeline.py $f 24 > /root/repo/$O/timeline_spare_$v.txt 2>&1
done
SGPU_GRID_SPARE=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_1250 -- python /root/repo/tools/e2e_timeline_run.py 1250 > /root/repo/$O/timeline_run_1250.txt 2>&1
f=$(find /tmp/tl_1250 -name '*kernel_trace.csv' | head -1)
python /root/repo/tools/e2e_timeline.py $f 16 > /root/repo/$O/timeline_1250.txt 2>&1
cat /root/repo/$O/probe.txt
