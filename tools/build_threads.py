"""Host build time against the number of host threads: the default (hardware threads capped by the container's CPU quota,
common.hpp host_threads) against explicit counts.   python tools/build_threads.py [n_docs] [counts, e.g. 0,256,64,16]"""
import os
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
counts = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 256, 64, 16]
t = time.time()
docs = _native.synth(n, 30000, 42, 0)
print("synth %d documents: %.1f s" % (n, time.time() - t), flush=True)
cfg = dict(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=1)
for nt in counts:
    t = time.time()
    _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(num_threads=nt, **cfg))
    print("num_threads %3d%s: total %.1f s" % (nt, " (default)" if nt == 0 else "", time.time() - t), flush=True)
