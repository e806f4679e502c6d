#!/usr/bin/env python3
"""What one rank of an N-GPU strong-scaling run does (bench.py --gpus N, BASELINE configs[3]): sgpu_batch_search calls of
10000 / N queries, issued by T request threads (bench.py's --host-threads, default 2). Reports microseconds per call at
steady state against the pro-rata share of a 10 000-query call: the predicted strong-scaling efficiency at N GPUs.
  python tools/shard_probe.py [--docs 8800000] [--shards 1,2,4,8] [--threads 1,2,3]"""
import argparse, os, sys, threading, time
os.environ.setdefault("SGPU_TEST_HOOKS", "1")   # (the SGPU_* knobs and sgpu_debug_* entry points this tool drives are test hooks)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seismic_amd import _native  # noqa: E402
from seismic_amd._abi import BuildConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=8_800_000)
ap.add_argument("--n-postings", type=int, default=2000)
ap.add_argument("--max-fraction", type=float, default=6.0)
ap.add_argument("--query-cut", type=int, default=4)
ap.add_argument("--heap-factor", type=float, default=1.0)
ap.add_argument("--shards", default="1,2,4,8")
ap.add_argument("--threads", default="1,2,3,4")
ap.add_argument("--batches", type=int, default=6)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
docs = _native.synth(a.docs, 30000, 42, 0)
ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(
    n_postings=a.n_postings, centroid_fraction=0.2, summary_energy=0.5, max_fraction=a.max_fraction, use_device=1))
ix.upload(0)
allq = _native.synth(10000 * a.batches, 30000, 43, 1, docs)
del docs
off, qc, qv = allq


def shard(b, n, r):
    lo = b * 10000 + 10000 * r // n
    hi = b * 10000 + 10000 * (r + 1) // n
    o0, o1 = int(off[lo]), int(off[hi])
    return (off[lo:hi + 1] - off[lo]).astype(np.uint64), qc[o0:o1], qv[o0:o1]


import ctypes
L = _native.lib()
L.sgpu_debug_call_timing.argtypes = [ctypes.c_void_p]
PH = ["validate+plan", "staging", "enqueue_h2d", "configure+launch", "enqueue_d2h", "wait", "copy_out"]
base = None
for n in [int(x) for x in a.shards.split(",")]:
    # rank 0's shard of every batch (a rank sees one shard per step)
    calls_in = [shard(b, n, 0) for b in range(a.batches)]
    nq = len(calls_in[0][0]) - 1
    outs = [(np.zeros((nq, 10), np.float32), np.zeros((nq, 10), np.uint64), np.zeros(nq, np.uint32)) for _ in calls_in]
    # host-side phases of one call (single thread, summed over the call's chunks)
    buf = (ctypes.c_double * 8)()
    L.sgpu_debug_call_timing(ctypes.addressof(buf))
    for j in range(2 * len(calls_in)):
        if j == len(calls_in):
            for i in range(8):
                buf[i] = 0.0
        ix.batch_search(*calls_in[j % len(calls_in)], 10, a.query_cut, a.heap_factor, False, out=outs[j % len(calls_in)])
    L.sgpu_debug_call_timing(None)
    print("shards %d: host phases per call (us): %s" % (n, ", ".join("%s %.0f" % (PH[i], buf[i] / len(calls_in)) for i in range(7))), flush=True)
    for T in [int(x) for x in a.threads.split(",")]:
        best = None
        for rep in range(a.reps + 1):
            n_calls = 8 * len(calls_in)

            def worker(t):
                for i in range(t, n_calls, T):
                    j = i % len(calls_in)
                    ix.batch_search(*calls_in[j], 10, a.query_cut, a.heap_factor, False, out=outs[j] if T == 1 else None)
            th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            us = (time.perf_counter() - t0) * 1e6 / n_calls
            if rep:
                best = us if best is None else min(best, us)
        if n == 1 and (base is None or best < base):
            base = best
        eff = (base / n) / best if base else float("nan")
        print("shards %d (%5d queries per call)  %d request thread(s): %8.1f us per call  -> %.1f %% of the pro-rata 1-GPU call (%.1f us)"
              % (n, nq, T, best, 100 * eff, (base or 0) / n), flush=True)
