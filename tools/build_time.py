import os
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import desc_equal
docs=_native.synth(8_800_000, 30000, 42, 0)
cfg=dict(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0)
for dev in (0, 1):
    t=time.time()
    ix=_native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(use_device=dev, **cfg))
    print("use_device", dev, "total %.1f s" % (time.time()-t), flush=True)
    if dev == 0: host = ix
desc_equal(host.desc, ix.desc)
print("byte-identical at 8.8M docs")
