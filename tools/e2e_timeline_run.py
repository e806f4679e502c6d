"""A handful of 10 000-query sgpu_batch_search calls from one thread (for rocprofv3 --kernel-trace; tools/e2e_timeline.py reads the csv)."""
import os, sys, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")
import numpy as np
sys.path.insert(0, '/root/repo')
from seismic_amd import _native
n = 8800000
docs = _native.synth(n, 30000, 42, 0)
path = "/tmp/lat_%d.idx" % n
ix = _native.NativeIndex.load(path)
ix.upload(0)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
q_off, qc, qv = _native.synth(nq, 30000, 43, 1, docs)
out = (np.zeros((nq, 10), np.float32), np.zeros((nq, 10), np.uint64), np.zeros(nq, np.uint32))
ts = []
for i in range(8):
    t0 = time.perf_counter(); ix.batch_search(q_off.astype(np.uint64), qc, qv, 10, 4, 1.0, False, out=out); ts.append((time.perf_counter() - t0) * 1e6)
    time.sleep(0.002)
print("call us:", " ".join("%.0f" % t for t in ts), flush=True)
