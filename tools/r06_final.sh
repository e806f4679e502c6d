#!/bin/bash
# r06: the driver's round-end commands at the final library: GPU suite, smoke(), bench line; then further fuzz seeds.
O=gpurun_out/r06_final; mkdir -p $O
(timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6) > $O/suite.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $O/smoke.log
(timeout 400 python bench.py 2>$O/bench.err | tail -1) > $O/bench.json
(timeout 400 python tools/soak.py 326 526 fuzz-only 2>&1 | tail -3) > $O/soak.log
cat $O/suite.log $O/smoke.log $O/soak.log
