#!/bin/bash
# r06: one free workgroup slot per XCD (workgroups go to XCDs round-robin: the plan kernels' blocks need room on every XCD)
O=gpurun_out/r06_entry3; mkdir -p $O
export SGPU_TEST_HOOKS=1
run() { echo "== $*" >> $O/probe.txt; env "$@" timeout 300 python tools/e2e_probe.py 2>&1 | grep qps >> $O/probe.txt; }
run A=0
run SGPU_GRID_SPARE=8
run SGPU_GRID_SPARE=16
run SGPU_GRID_SPARE=8 SGPU_CHUNK_FIRST=300
run SGPU_GRID_SPARE=8 SGPU_CHUNK_MAX=4
run SGPU_GRID_SPARE=8
cd /tmp && export TMPDIR=/tmp
for n in 8 16; do
  SGPU_GRID_SPARE=$n timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$n -- python /root/repo/tools/e2e_timeline_run.py 10000 > /root/repo/$O/timeline_run_$n.txt 2>&1
  f=$(find /tmp/tl_$n -name '*kernel_trace.csv' | head -1)
  python /root/repo/tools/e2e_timeline.py $f 24 > /root/repo/$O/timeline_spare_$n.txt 2>&1
done
cat /root/repo/$O/probe.txt; tail -7 /root/repo/$O/timeline_spare_8.txt
