#!/usr/bin/env python3
"""Times Knn::new on the GPU (every document searched as a query through the search kernel)."""
import os, sys, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nknn = int(sys.argv[2]) if len(sys.argv) > 2 else 10
docs = _native.synth(n, 30000, 42, 0)
ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=max(1, 2000 * n // 1000000), centroid_fraction=0.2,
                                                                      summary_energy=0.5, max_fraction=6.0)).upload(0)
t = time.time()
ix.build_knn(nknn)
dt = time.time() - t
nb, kd = ix.get_knn()
print("Knn::new on GPU: %d docs, nknn=%d: %.2f s (%.0f doc-queries/s), %d neighbour ids" % (n, nknn, dt, n / dt, len(nb)))
q = _native.synth(1000, 30000, 43, 1, docs)
b = _native.DeviceBatch(ix, *q, 10)
for nk in (0, 5, 10):
    b.run(10, 4, 1.0, False, n_knn=nk)
    ms = min(b.run(10, 4, 1.0, False, n_knn=nk).kernel_ms for _ in range(5))
    print("search 1000 queries n_knn=%d: %.3f ms" % (nk, ms))
