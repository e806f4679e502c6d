#!/usr/bin/env python3
"""Where the Python binding's per-call time comes from: the same 200 single-query sgpu_search calls through ctypes, timed
(1) in a fresh process, (2) after `import torch`, (3) after torch has initialised the device, (4) after request threads
ran batch calls (bench.py's order), (5) after the oracle's OpenMP team ran. Native loop (sgpu_search_sequential) beside each."""
import os, sys, time, threading
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
TORCH_FIRST = os.environ.get("PROBE_TORCH_FIRST", "0") == "1"
if TORCH_FIRST:   # bench.py's order: torch owns the device before the library is loaded
    import torch
    torch.cuda.set_device(0)
    torch.zeros(8, device="cuda").sum().item()
    torch.cuda.synchronize()
from seismic_amd import _native
from seismic_amd._abi import BuildConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8800000
docs = _native.synth(n, 30000, 42, 0)
path = os.path.join(os.environ.get("SGPU_INDEX_CACHE", "/tmp"), "lat_%d.idx" % n)
if os.path.exists(path):
    ix = _native.NativeIndex.load(path)
else:
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5,
                                                                          max_fraction=6.0, use_device=1))
    ix.save(path)
ix.upload(0)
q_off, qc, qv = _native.synth(10000, 30000, 43, 1, docs)
qs = [(qc[int(q_off[i]):int(q_off[i + 1])], qv[int(q_off[i]):int(q_off[i + 1])]) for i in range(200)]


def probe(tag):
    for c, v in qs[:10]:
        ix.search(c, v, 10, 4, 1.0, False)
    per = []
    for rep in range(3):
        t0 = time.perf_counter()
        for c, v in qs:
            ix.search(c, v, 10, 4, 1.0, False)
        per.append((time.perf_counter() - t0) * 1e6 / len(qs))
    nat = ix.search_sequential(q_off[:201], qc, qv, 10, 4, 1.0, False)[3]
    print("%-44s python binding %6.1f us/query (passes %s)   native loop %6.1f us   threads in process %d"
          % (tag, min(per), " ".join("%.0f" % p for p in per), nat, len(os.listdir("/proc/self/task"))), flush=True)


probe("torch initialised the device first" if TORCH_FIRST else "fresh process, no torch")
outs = [(np.zeros((10000, 10), np.float32), np.zeros((10000, 10), np.uint64), np.zeros(10000, np.uint32)) for _ in range(2)]
def worker(t):
    for _ in range(4):
        ix.batch_search(q_off, qc, qv, 10, 4, 1.0, False, out=outs[t])
th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
[x.start() for x in th]; [x.join() for x in th]
probe("after two request threads ran batch calls")
import orc
orc.batch_search(ix.desc, q_off[:257], qc, qv, 10, 4, 1.0, False, num_threads=16, tuned=True)
probe("after the oracle's OpenMP team ran")
time.sleep(1.0)
probe("one second later")
