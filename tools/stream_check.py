#!/usr/bin/env python3
"""Plain (streamed stage 2) launches against the oracle on small random collections; prints what differs.
Usage: [SGPU_LIB=...] python tools/stream_check.py [n_docs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SGPU_TEST_HOOKS", "1")
os.environ["SGPU_COOP"] = "0"
import orc  # noqa: E402
from seismic_amd import _native  # noqa: E402
from seismic_amd._abi import BuildConfig  # noqa: E402
from util import random_dataset, random_queries  # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dim = 512
off, comps, vals = random_dataset(5, n_docs, dim, nnz_lo=8, nnz_hi=120)
ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=100, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0))
ix.upload(0)
q = random_queries(6, 64, dim, 5, 50)
bad_total = 0
for block in (() if os.environ.get("SGPU_TRACE_QUERY") else ("1024", "512")):
    os.environ["SGPU_BLOCK"] = block
    for k, qcut, hf, srt in ((10, 4, 1.0, False), (10, 4, 1.0, True), (1, 2, 0.8, False), (100, 8, 0.9, False), (10, 1, 1.0, False)):
        gs, gi, gn = ix.batch_search(*q, k, qcut, hf, srt)
        os_, oi, on, _, _, _ = orc.batch_search(ix.desc, *q, k, qcut, hf, srt)
        bad = [i for i in range(len(gn)) if gn[i] != on[i] or not np.array_equal(gi[i, :gn[i]], oi[i, :on[i]]) or
               not np.array_equal(gs[i, :gn[i]].view(np.uint32), os_[i, :on[i]].view(np.uint32))]
        bad_total += len(bad)
        print("block %s k %d qcut %d hf %g sorted %d: %d of %d queries differ %s" % (block, k, qcut, hf, srt, len(bad), len(gn), bad[:8]))
        if bad:
            i = bad[0]
            print("   query %d: gpu n %d ids %s scores %s" % (i, gn[i], gi[i, :min(10, gn[i])], gs[i, :min(10, gn[i])]))
            print("             cpu n %d ids %s scores %s" % (on[i], oi[i, :min(10, on[i])], os_[i, :min(10, on[i])]))
print("library:", _native.build_info(), "| differing:", bad_total)


def where(qi, k, qcut, hf, srt):
    """Where the documents the GPU missed sit in the traversal of query qi."""
    gs, gi, gn = ix.batch_search(*q, k, qcut, hf, srt)
    os_, oi, on, _, _, _ = orc.batch_search(ix.desc, *q, k, qcut, hf, srt)
    a = orc.desc_arrays(ix.desc)
    q_off, qc, qv = q
    c = qc[q_off[qi]:q_off[qi + 1]]
    v = qv[q_off[qi]:q_off[qi + 1]]
    sel = c[np.argsort(-v, kind="stable")[:qcut]]
    missing = [d for d in oi[qi, :on[qi]] if d not in set(gi[qi, :gn[qi]])]
    extra = [d for d in gi[qi, :gn[qi]] if d not in set(oi[qi, :on[qi]])]
    print("query %d: lists %s missing %s extra %s kth cpu %.5f gpu %.5f" % (qi, sel, missing, extra, os_[qi, on[qi] - 1], gs[qi, gn[qi] - 1]))
    seq = 0
    for li, lst in enumerate(sel):
        b0, b1 = int(a["list_block_start"][lst]), int(a["list_block_start"][lst + 1])
        dots = orc.summary_distances(ix.desc, int(lst), c, v)
        order = np.arange(b1 - b0)
        if srt and li == 0 and b1 - b0 > 1:
            order = np.lexsort((np.arange(b1 - b0), -dots))
        for pos, b in enumerate(order):
            p0, p1 = int(a["block_post_start"][b0 + b]), int(a["block_post_start"][b0 + b + 1])
            docs = a["post_doc"][p0:p1]
            for d in missing + extra:
                if d in docs:
                    j = int(np.nonzero(docs == d)[0][0])
                    print("   doc %d: list %d (#%d, %d blocks) position %d block %d dot %.5f posting %d of %d; items before (all blocks live) %d" % (
                        d, lst, li, b1 - b0, pos, b, dots[b], j, p1 - p0, seq + j))
            seq += p1 - p0


def trace(qi, k, qcut, hf, srt):
    """One query alone (a library built with -DSGPU_STREAM_PRINTF prints its item stream) + the expected traversal."""
    q_off, qc, qv = q
    one = (np.array([0, q_off[qi + 1] - q_off[qi]], q_off.dtype), qc[q_off[qi]:q_off[qi + 1]].copy(), qv[q_off[qi]:q_off[qi + 1]].copy())
    gs, gi, gn = ix.batch_search(*one, k, qcut, hf, srt)
    os_, oi, on, _, _, _ = orc.batch_search(ix.desc, *one, k, qcut, hf, srt)
    print("TRACE gpu", gi[0, :gn[0]], gs[0, :gn[0]])
    print("TRACE cpu", oi[0, :on[0]], os_[0, :on[0]])
    a = orc.desc_arrays(ix.desc)
    c, v = one[1], one[2]
    sel = c[np.argsort(-v, kind="stable")[:qcut]]
    seq = 0
    for li, lst in enumerate(sel):
        b0, b1 = int(a["list_block_start"][lst]), int(a["list_block_start"][lst + 1])
        dots = orc.summary_distances(ix.desc, int(lst), c, v)
        order = np.arange(b1 - b0)
        if srt and li == 0 and b1 - b0 > 1:
            order = np.lexsort((np.arange(b1 - b0), -dots))
        for pos, b in enumerate(order):
            p0, p1 = int(a["block_post_start"][b0 + b]), int(a["block_post_start"][b0 + b + 1])
            print("E list#%d pos %d block %d dot %.7g seq %d docs %s" % (li, pos, b, dots[b], seq, list(a["post_doc"][p0:p1])))
            seq += p1 - p0
            if seq > 200:
                return


if os.environ.get("SGPU_TRACE_QUERY"):
    os.environ["SGPU_BLOCK"] = "512"
    trace(int(os.environ["SGPU_TRACE_QUERY"]), 10, 4, 1.0, True)
elif bad_total:
    os.environ["SGPU_BLOCK"] = "512"
    where(2, 10, 4, 1.0, True)
    where(6, 10, 4, 1.0, True)
    where(11, 100, 8, 0.9, False)

