"""Host build time against the number of host threads (SGPU_DEBUG=1 prints the phases)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
docs = _native.synth(2_000_000, 30000, 42, 0)
cfg = dict(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=1)
for nt in (256, 128, 64, 32):
    t = time.time()
    _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(num_threads=nt, **cfg))
    print("threads", nt, "total %.1f s" % (time.time() - t), flush=True)
