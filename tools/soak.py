import sys, os, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
import importlib
fz = importlib.import_module("test_gpu_fuzz")

class MP:
    def __init__(self): self.saved = {}
    def setenv(self, k, v):
        self.saved.setdefault(k, os.environ.get(k)); os.environ[k] = v
    def undo(self):
        for k, v in self.saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
bad = []
t0 = time.time()
# the committed test derives everything from the seed: run it for many more seeds
orig = np.random.default_rng
LO, HI = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (26, 226)
for seed in range(LO, HI):
    mp = MP()
    try:
        fz.test_differential(seed, "plain" if seed % 2 else "default", mp)   # (odd seeds: plain launches - stage 2 as a stream)
    except Exception as e:
        bad.append((seed, repr(e)[:300]))
    finally:
        mp.undo()
print("fuzz seeds %d..%d: %d failures in %.0f s" % (LO, HI - 1, len(bad), time.time() - t0), bad[:5], flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "fuzz-only":
    sys.exit(0)
# config 5 at full size vs the oracle
dim, n_docs, nq = 200_000, 5_000_000, 200
docs = _native.synth(n_docs, dim, 42, 0)
ix = _native.NativeIndex.build(4, dim, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.1, summary_energy=0.4,
                                                                   max_fraction=4.0, min_cluster_size=10, use_device=1))
ix.upload(0)
q = _native.synth(nq, dim, 43, 1, docs)
for hf in (0.7, 0.9, 1.0):
    g = ix.batch_search(*q, 100, 10, hf, False)
    c = orc.batch_search(ix.desc, *q, 100, 10, hf, False)[:3]
    ok = np.array_equal(g[2], c[2]) and np.array_equal(g[1], c[1]) and np.array_equal(g[0].view(np.uint32), c[0].view(np.uint32))
    print("config 5 at 5M docs, heap_factor %.1f: identical to the oracle: %s" % (hf, ok), flush=True)
