"""Pins the CPU oracle to the reference's OWN known-answer tests for this path.

  - test_empty_vectors           reference src/inverted_index.rs:716-772
  - test_distances_iter          reference src/quantized_summary.rs:519-598
  - docs example                 reference docs/RustUsage.md:138-157
Inputs/expected values are restated as data (see tests/golden/README.md).
"""
import json
import os

import numpy as np
import pytest

import orc
from seismic_amd._abi import BuildConfig

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _build(vectors, dim, comp_width=2, **cfg):
    off, comps, vals = orc.csr(vectors)
    return orc.OracleIndex(comp_width, dim, off, comps, vals, BuildConfig.defaults(**cfg))


def test_empty_vectors_kat():
    g = json.load(open(os.path.join(GOLD, "kat_empty_vectors.json")))
    ix = _build([(d["components"], d["values"]) for d in g["docs"]], g["dim"])
    assert ix.desc.n_docs == 4 and ix.desc.dim == 5 and ix.desc.nnz == 7
    q = g["query"]
    for order in (orc.ORDER_LANES16, orc.ORDER_SEQ):
        sc, ids = orc.search(ix.desc, q["components"], q["values"], q["k"], q["query_cut"],
                             q["heap_factor"], q["first_sorted"], order)
        assert ids.tolist() == g["expected_ids"]          # [3, 0]: empty docs never retrieved
        assert sc.tolist() == g["expected_scores"]        # 30, 7 (exact small integers)


def test_rust_usage_doc_example():
    g = json.load(open(os.path.join(GOLD, "kat_rust_usage.json")))
    ix = _build([(d["components"], d["values"]) for d in g["docs"]], g["dim"])
    q = g["query"]
    sc, ids = orc.search(ix.desc, q["components"], q["values"], q["k"], q["query_cut"],
                         q["heap_factor"], False)
    assert ids.tolist() == g["expected_ids"] and sc.tolist() == g["expected_scores"]


def test_convert_dataset_preserves_postings_property():
    # reference src/inverted_index.rs:774-807: every posting doc id < len, postings non-empty
    ix = _build([([0, 2], [1.0, 2.0]), ([1, 3], [3.0, 4.0])], 4)
    a = orc.desc_arrays(ix.desc)
    assert len(a["post_doc"]) > 0 and (a["post_doc"] < 2).all()


def _merge_dot(qc, qv, dc, dv):
    i = j = 0
    r = np.float32(0)
    while i < len(qc) and j < len(dc):
        if qc[i] == dc[j]:
            r = np.float32(r + np.float32(qv[i]) * np.float32(dv[j]))
            i += 1
            j += 1
        elif qc[i] < dc[j]:
            i += 1
        else:
            j += 1
    return r


@pytest.mark.parametrize("comp_width", [4])
def test_distances_iter_property(comp_width):
    """All-ones vectors => quantisation is lossless (min==max -> quant 0 -> NaN -> code 0
    -> dequant == min), so distances() must equal the exact merge dot within 1e-5."""
    rng = np.random.default_rng(142)
    n_vecs = int(rng.integers(50, 101))
    dim = int(rng.integers(100_000, 140_001))
    vecs = []
    for _ in range(n_vecs):
        nnz = int(rng.integers(300, 501))
        c = np.sort(rng.choice(dim, nnz, replace=False)).astype(np.uint32)
        vecs.append((c, np.ones(nnz, np.float32)))
    # Put all vectors into ONE posting block each: build a one-list "index" by hand is
    # overkill; instead check the QuantizedSummary arithmetic through a synthetic desc:
    import ctypes as C
    from seismic_amd._abi import IndexDesc
    # summary dataset == the vectors themselves; list 0 owns all n_vecs blocks
    mins, quants, rows = [], [], {}
    for b, (c, v) in enumerate(vecs):
        mn, qt, codes = orc.quantize(v)
        assert qt == 0.0 and (codes == 0).all() and mn == 1.0
        mins.append(mn)
        quants.append(qt)
        for ci, code in zip(c, codes):
            rows.setdefault(int(ci), []).append((b, int(code)))
    row_comp = np.array(sorted(rows), np.uint32)
    row_ptr = np.zeros(len(row_comp) + 1, np.uint64)
    bid, code = [], []
    for i, c in enumerate(row_comp):
        for b, cd in rows[int(c)]:
            bid.append(b)
            code.append(cd)
        row_ptr[i + 1] = len(bid)
    bid = np.array(bid, np.uint16)
    code = np.array(code, np.uint8)
    mins = np.array(mins, np.float32)
    quants = np.array(quants, np.float32)
    lbs = np.zeros(dim + 1, np.uint64)
    lbs[1:] = n_vecs
    lrs = np.zeros(dim + 1, np.uint64)
    lrs[1:] = len(row_comp)
    bps = np.zeros(n_vecs + 1, np.uint64)
    fo = np.zeros(1, np.uint64)
    d = IndexDesc(comp_width=4, n_docs=0, dim=dim, nnz=0, n_blocks=n_vecs, n_postings=0,
                  n_rows=len(row_comp), n_entries=len(bid))
    keep = [lbs, lrs, bps, fo, row_comp, row_ptr, bid, code, mins, quants]
    d.fwd_offsets = fo.ctypes.data_as(type(d.fwd_offsets))
    d.list_block_start = lbs.ctypes.data_as(type(d.list_block_start))
    d.block_post_start = bps.ctypes.data_as(type(d.block_post_start))
    d.blk_min = mins.ctypes.data_as(type(d.blk_min))
    d.blk_quant = quants.ctypes.data_as(type(d.blk_quant))
    d.list_row_start = lrs.ctypes.data_as(type(d.list_row_start))
    d.row_comp = row_comp.ctypes.data_as(C.c_void_p)
    d.row_ptr = row_ptr.ctypes.data_as(type(d.row_ptr))
    d.sum_bid = bid.ctypes.data_as(type(d.sum_bid))
    d.sum_code = code.ctypes.data_as(type(d.sum_code))
    queries = []
    for _ in range(100):
        nnz = int(rng.integers(300, 501))
        c = np.sort(rng.choice(dim, nnz, replace=False)).astype(np.uint32)
        queries.append((c, rng.random(nnz, dtype=np.float32)))
    queries += vecs
    for qc, qv in queries:
        got = orc.summary_distances(d, 0, qc, qv)
        assert len(got) == n_vecs
        exp = np.array([_merge_dot(qc, qv, c, v) for c, v in vecs], np.float32)
        assert np.abs(got - exp).max() < 1e-5
    del keep


def test_quantize_traps():
    # reference src/utils.rs:68-90 — half-away-from-zero rounding, saturating cast, NaN -> 0
    mn, qt, codes = orc.quantize([0.0, 255.0, 127.5, 0.5, 254.5])
    assert (mn, qt) == (0.0, 1.0) and codes.tolist() == [0, 255, 128, 1, 255]
    mn, qt, codes = orc.quantize([2.5, 2.5, 2.5])
    assert mn == 2.5 and qt == 0.0 and codes.tolist() == [0, 0, 0]
    mn, qt, codes = orc.quantize([1.0])
    assert codes.tolist() == [0]


def test_f16_roundtrip_matches_numpy():
    L = orc.lib()
    allh = np.arange(65536, dtype=np.uint16)
    asf = allh.view(np.float16).astype(np.float32)
    for h in list(range(0, 65536, 7)) + [0x7bff, 0xfbff, 0x0001, 0x03ff, 0x0400]:
        f = L.orc_f16_to_f32(h)
        if np.isnan(asf[h]):
            assert np.isnan(f)
        else:
            assert f == asf[h]
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.normal(0, 3, 5000), rng.uniform(-70000, 70000, 2000),
                         rng.uniform(-1e-5, 1e-5, 2000), [65504, 65519.9, 65520, 1e9, -1e9, 0.0, -0.0,
                                                          2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25]]).astype(np.float32)
    for x in xs:
        got = L.orc_f32_to_f16(float(x))
        with np.errstate(over="ignore"):
            ref = np.float32(x).astype(np.float16)
        if np.isinf(ref):  # saturating, not IEEE overflow-to-inf
            assert got == (0x7bff if x > 0 else 0xfbff)
        else:
            assert got == int(ref.view(np.uint16)), (x, got)


def test_knn_hand_computed():
    """Knn::new / Knn::refine on five integer-valued documents, derived by hand.

    Pairwise dot products: d0.d1=14 d0.d2=5 d0.d3=0 d0.d4=4 | d1.d2=10 d1.d3=0 d1.d4=3 | d2.d3=4 d2.d4=0 |
    d3.d4=6; self products 17, 13, 26, 20, 10. Every posting list has at most three postings, hence one
    block (max(1, floor(0.1 * len)) centroids). Knn::new searches each document with k = nknn + 1 = 3,
    query_cut 10, heap_factor 0.7, lists in descending weight:
      d0 (c0:4, c1:1): list 0 -> d0 17, d1 14, d4 4 (full, 3rd = 4); list 1 (summary dot >= 0.7*4): d2 5 > 4
                       replaces d4 -> [d0, d1, d2] -> neighbours [1, 2]
      d1 (c0:3, c1:2): list 0 -> d0 14, d1 13, d4 3; list 1: d2 10 replaces d4 -> [d0, d1, d2] -> [0, 2]
      d2 (c1:5, c2:1): list 1 -> d2 26, d1 10, d0 5; list 2: d3 4 < 5 -> [d2, d1, d0] -> [1, 0]
      d3 (c2:4, c3:2): list 2 -> d3 20, d2 4; list 3: d4 6 -> [d3, d4, d2] -> [4, 2]
      d4 (c3:3, c0:1): list 3 -> d4 10, d3 6; list 0: d0 4 fills the heap, d1 3 < 4 -> [d4, d3, d0] -> [3, 0]
    refine, query (c0: 0.1, c2: 1.0), k 3, query_cut 1 (list 2 only): d3 = 4.0, d2 = 1.0 (heap not full).
      Snapshot [d3, d2]; d3's neighbours [4, 2]: d4 = 0.1 pushed (heap full, 3rd = 0.1), d2 visited;
      d2's neighbours [1, 0]: d1 = 0.1*3 replaces d4, d0 = 0.1*4 replaces d1 -> ids [3, 2, 0]."""
    g = json.load(open(os.path.join(GOLD, "kat_knn_hand.json")))
    ix = _build([(d["components"], d["values"]) for d in g["docs"]], g["dim"], **g["build"])
    a = orc.desc_arrays(ix.desc)
    assert np.diff(a["list_block_start"].astype(np.int64)).tolist() == [1, 1, 1, 1]
    nb = orc.knn_build(ix.desc, g["nknn"])
    assert nb.tolist() == g["expected_neighbours"]
    q = g["refine_query"]
    s0, i0 = orc.search(ix.desc, q["components"], q["values"], q["k"], q["query_cut"], q["heap_factor"], False)
    assert i0.tolist() == g["expected_without_refine"]["ids"] and s0.tolist() == g["expected_without_refine"]["scores"]
    orc.knn_attach(nb, g["nknn"])
    try:
        s1, i1 = orc.search(ix.desc, q["components"], q["values"], q["k"], q["query_cut"], q["heap_factor"], False,
                            n_knn=q["n_knn"])
    finally:
        orc.knn_attach(None, 0)
    assert i1.tolist() == g["expected_with_refine"]["ids"]
    w = np.float32(0.1)
    assert s1.tolist() == [4.0, 1.0, float(w * np.float32(4.0))]


def test_lossy_summary_distances_against_a_straight_line_loop():
    """QuantizedSummary::distances (src/quantized_summary.rs:64-118) on LOSSY summaries
    (real-valued data, u8 codes), checked against a scalar restatement written straight from the
    source: for each query component in ascending order, for each entry of that component's row,
    acc[block] += (f32(code) * quant[block] + min[block]) * qv - every operation rounded to f32."""
    rng = np.random.default_rng(7)
    dim, n_docs = 120, 2500
    vecs = []
    for _ in range(n_docs):
        n = int(rng.integers(5, 40))
        c = np.sort(rng.choice(dim, n, replace=False))
        vecs.append((c, (rng.exponential(0.45, n) + 0.02).astype(np.float32)))
    ix = _build(vecs, dim, n_postings=200, centroid_fraction=0.2, summary_energy=0.6, max_fraction=4.0)
    a = orc.desc_arrays(ix.desc)
    assert len(np.unique(a["sum_code"])) > 100          # the quantisation is genuinely lossy
    lists = np.argsort(np.diff(a["list_block_start"].astype(np.int64)))[-5:]
    f32 = np.float32
    for t in range(6):
        n = int(rng.integers(4, 50))
        qc = np.sort(rng.choice(dim, n, replace=False)).astype(np.uint32)
        qv = (rng.exponential(0.5, n) + 0.01).astype(np.float32)
        for l in lists:
            b0, b1 = int(a["list_block_start"][l]), int(a["list_block_start"][l + 1])
            acc = [f32(0)] * (b1 - b0)
            rows = {int(a["row_comp"][r]): r for r in range(int(a["list_row_start"][l]), int(a["list_row_start"][l + 1]))}
            for c, v in zip(qc.tolist(), qv):
                r = rows.get(c)
                if r is None:
                    continue
                for e in range(int(a["row_ptr"][r]), int(a["row_ptr"][r + 1])):
                    blk = int(a["sum_bid"][e])
                    deq = f32(f32(f32(a["sum_code"][e]) * a["blk_quant"][b0 + blk]) + a["blk_min"][b0 + blk])
                    acc[blk] = f32(acc[blk] + f32(deq * v))
            got = orc.summary_distances(ix.desc, int(l), qc, qv)
            assert np.array_equal(np.asarray(acc, np.float32).view(np.uint32), got.view(np.uint32)), (t, l)


# ---- the tuned CPU path (cpu_baseline) is the restatement, bit for bit -------------------------
@pytest.mark.parametrize("comp_width,fixedu8", [(2, False), (4, False), (2, True)])
def test_tuned_cpu_path_is_bit_identical_to_the_restatement(comp_width, fixedu8):
    """bench.py times orc_batch_search_tuned (AVX2 + F16C scorer, hash-set visited set, pinned threads) as
    cpu_baseline; it must return what the plain restatement returns: same ids, same score bits, same work
    counters - for every document length class (tails of 1..7 elements, > 128, > 256 elements), both
    value types, both component widths, sorted first list, kNN refinement, several thread counts."""
    from util import random_dataset, random_queries
    dim = 700 if comp_width == 2 else 70000
    off, comps, vals = random_dataset(11 + comp_width, 3000, dim, nnz_lo=1, nnz_hi=300, empty_every=97)
    vals = vals.copy()
    vals[::13] = 0.0            # stored zeros
    ix = orc.OracleIndex(comp_width, dim, off, comps, vals,
                         BuildConfig.defaults(n_postings=400, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0))
    if fixedu8:
        ix = ix.convert_fixedu8()
    q_off, qc, qv = random_queries(5, 60, dim, 3, 60)
    qv = qv.copy()
    qv[::7] *= -1.0             # negative weights
    # the scorer alone, document by document
    for doc in list(range(0, 3000, 37)) + [96]:
        for q in (0, 7, 31):
            c, v = qc[int(q_off[q]):int(q_off[q + 1])], qv[int(q_off[q]):int(q_off[q + 1])]
            a = np.float32(orc.score_doc(ix.desc, doc, c, v)).view(np.uint32)
            b = np.float32(orc.score_doc_tuned(ix.desc, doc, c, v)).view(np.uint32)
            assert a == b, (doc, q)
    knn = orc.knn_build(ix.desc, 4) if not fixedu8 else None
    try:
        if knn is not None:
            orc.knn_attach(knn, 4)
        for k, cut, hf, srt, nk in ((10, 4, 1.0, False, 0), (100, 10, 0.7, True, 0), (5, 3, 0.9, True, 2 if knn is not None else 0)):
            ref = orc.batch_search(ix.desc, q_off, qc, qv, k, cut, hf, srt, num_threads=1, n_knn=nk)
            for nt in (1, 3):
                got = orc.batch_search(ix.desc, q_off, qc, qv, k, cut, hf, srt, num_threads=nt, n_knn=nk, tuned=True)
                assert np.array_equal(ref[2], got[2]) and np.array_equal(ref[1], got[1])
                assert np.array_equal(ref[0].view(np.uint32), got[0].view(np.uint32))
                assert ref[3] == got[3]          # work counters / algorithmic bytes
    finally:
        orc.knn_attach(None, 0)


def test_tuned_baseline_pin_plan_gives_every_thread_its_own_cpu():
    """The tuned batch search pins thread t to pin_order[pin_slot(t)] (physical cores first): distinct, allowed CPUs for
    every team size up to the CPUs this process may use."""
    import os
    allowed = sorted(os.sched_getaffinity(0))
    for nt in sorted({1, 2, 3, len(allowed) // 2 or 1, len(allowed)}):
        cores, cpus = orc.pin_plan(nt)
        assert 1 <= cores <= len(allowed)
        assert len(set(cpus.tolist())) == nt and set(cpus.tolist()) <= set(allowed)
