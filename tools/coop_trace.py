#!/usr/bin/env python3
"""Timeline of cooperative single-query launches. Needs a trace build of the library:
  make -C seismic_amd/csrc OUT=../libseismic_hip_dbg.so BUILD=build_dbg EXTRA=-DSGPU_COOP_TRACE
  SGPU_LIB=seismic_amd/libseismic_hip_dbg.so SGPU_COOP_TRACE=1 python tools/coop_trace.py [n_docs] [n_queries]
Event times are 10 ns ticks of the constant 100 MHz counter, printed in microseconds after the owner took its query."""
import ctypes as C, os, sys
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SGPU_COOP_TRACE", "1")
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
docs = _native.synth(n, 30000, 42, 0)
path = "/tmp/lat_%d.idx" % n
if os.path.exists(path):
    ix = _native.NativeIndex.load(path)
else:
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2, summary_energy=0.5,
                                                                          max_fraction=6.0, use_device=1))
    ix.save(path)
ix.upload(0)
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8
q_off, qc, qv = _native.synth(max(nq, 16), 30000, 43, 1, docs)
L = _native.lib()
L.sgpu_debug_coop_trace.restype = C.c_uint32
L.sgpu_debug_coop_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
OWNER = ["took query", "wide: enter", "wide: published", "wide: own work done", "wide: all chunks done", "wide: replayed",
         "query finished", "summary dots done"]
for i in range(nq):
    ix.search(qc[q_off[i]:q_off[i + 1]], qv[q_off[i]:q_off[i + 1]], 10, 4, 1.0, False)
    buf = np.zeros(512 * 16, np.uint64)
    got = L.sgpu_debug_coop_trace(ix.h, buf.ctypes.data_as(C.c_void_p), len(buf))
    ev = buf[:got].reshape(-1, 16).astype(np.int64)
    owners = np.nonzero(ev[:, 0])[0]
    if len(owners) == 0:
        print("query %d: no trace (is this a trace build? SGPU_COOP=1?)" % i)
        continue
    o = owners[0]
    t0 = ev[o, 0]
    us = lambda t: (t - t0) / 100.0
    line = "query %2d (nnz %3d) owner wg %3d: " % (i, q_off[i + 1] - q_off[i], o)
    line += "  ".join("%s %.1f" % (OWNER[e], us(ev[o, e])) for e in (7, 1, 2, 3, 4, 5, 6) if ev[o, e])
    print(line)
    h = ev[ev[:, 8] != 0]
    if len(h):
        def rng(e):
            v = h[:, e][h[:, e] != 0]
            return "%.1f..%.1f (%d)" % (us(v.min()), us(v.max()), len(v)) if len(v) else "-"
        print("      helpers: attach %s  first claim %s  query in LDS %s  chunk filtered %s  last chunk scored %s" % (
            rng(8), rng(9), rng(10), rng(11), rng(12)))
