#!/usr/bin/env python3
"""What predicts a query's time in the search kernel? Dumps, per query of the 8.8M-document shape, the kernel's own
clocks and work counters (profiling build) next to what the host knows BEFORE the launch (postings / blocks / weights of the
lists the query will walk, its length), once inside a 10 000-query launch and once inside 1250-query launches.
The launch plan orders queries by an a-priori cost (make_plan in device_index.hip); this is the data that cost is fitted on.
usage: proxy_probe.py OUT.npz [--docs N]"""
import os
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SGPU_LIB", os.path.join(ROOT, "seismic_amd", "libseismic_hip_prof.so"))
from seismic_amd import _native  # noqa: E402
from seismic_amd._abi import BuildConfig  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[sys.argv.index("--docs") + 1]) if "--docs" in sys.argv else 8800000
NQ, CUT, K = 10000, 4, 10
docs = _native.synth(n, 30000, 42, 0)
path = "/tmp/lat_%d.idx" % n
if os.path.exists(path):
    ix = _native.NativeIndex.load(path)
else:
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(
        n_postings=max(1, 2000 * n // 1000000) if n < 1000000 else 2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=1))
    ix.save(path)
ix.upload(0)
q_off, qc, qv = _native.synth(NQ, 30000, 43, 1, docs)
d = ix.desc
lbs = np.ctypeslib.as_array(d.list_block_start, shape=(d.dim + 1,)).astype(np.int64)
bps = np.ctypeslib.as_array(d.block_post_start, shape=(d.n_blocks + 1,)).astype(np.int64)
list_np = bps[lbs[1:]] - bps[lbs[:-1]]
list_nb = lbs[1:] - lbs[:-1]
feat = np.zeros((NQ, 3 * CUT + 2))   # per walked list (heaviest first): postings, blocks, weight; then query length, sum of weights
for i in range(NQ):
    c = qc[q_off[i]:q_off[i + 1]]
    v = qv[q_off[i]:q_off[i + 1]]
    top = np.argsort(-v, kind="stable")[:CUT]
    m = len(top)
    feat[i, 0:m] = list_np[c[top]]
    feat[i, CUT:CUT + m] = list_nb[c[top]]
    feat[i, 2 * CUT:2 * CUT + m] = v[top]
    feat[i, 3 * CUT] = len(c)
    feat[i, 3 * CUT + 1] = v.sum()


def run(lo, hi, reps=3):
    b = _native.DeviceBatch(ix, q_off[lo:hi + 1] - q_off[lo], qc[q_off[lo]:q_off[hi]], qv[q_off[lo]:q_off[hi]], K)
    ms = [b.run(K, CUT, 1.0, False).kernel_ms for _ in range(reps)]
    return b.fetch_stats().copy(), min(ms)


big, ms_big = run(0, NQ)
mids, ms_mid = [], []
for r in range(8):
    st, ms = run(r * 1250, (r + 1) * 1250)
    mids.append(st)
    ms_mid.append(ms)
print("10 000-query launch %.3f ms (profiling build); 1250-query launches %s us" % (ms_big, [int(x * 1e3) for x in ms_mid]))
np.savez_compressed(out, feat=feat, big=big, mid=np.concatenate(mids), ms_big=ms_big, ms_mid=np.array(ms_mid))
