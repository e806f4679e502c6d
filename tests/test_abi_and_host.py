"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the header
declares; host logic (marshalling, file formats, validation, exact search); loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import orc
import seismic_amd
from seismic_amd import _native
from seismic_amd._abi import BuildConfig, IndexDesc
from seismic_amd.index import _resolve
from util import random_dataset, random_queries

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(_native.LIB_PATH)
    for header, at_least in (("seismic_hip.h", 20), ("seismic_hip_testing.h", 6)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        names = set(re.findall(r"\b(sgpu_[a-z_0-9]+)\s*\(", hdr))
        assert len(names) >= at_least, header
        missing = [n for n in sorted(names) if not hasattr(L, n)]
        assert not missing, (header, missing)
    assert L.sgpu_abi_version() == 4


def test_the_library_is_the_build_of_this_tree():
    """sgpu_build_info() carries a fingerprint of the sources the binary was compiled from; it must be the fingerprint of
    the sources next to it (the prebuilt .so travels to the GPU box with the tree: a stale one would run there unnoticed).
    The profiling build is checked when present."""
    want = _native.source_fingerprint()
    info = _native.build_info()
    assert info.startswith("sources %s arch gfx950 extra [] " % want), (info, want)
    prof = os.path.join(os.path.dirname(_native.__file__), "libseismic_hip_prof.so")
    if os.path.exists(prof) and os.path.abspath(prof) != os.path.abspath(_native.LIB_PATH):
        import ctypes
        L = ctypes.CDLL(prof)
        L.sgpu_build_info.restype = ctypes.c_char_p
        assert L.sgpu_build_info().decode().startswith("sources %s arch gfx950 extra [-DSGPU_PROF]" % want)


def test_test_hooks_are_inert_without_the_switch(monkeypatch):
    """The sgpu_debug_* entry points (include/seismic_hip_testing.h) and the undocumented environment names only work
    while SGPU_TEST_HOOKS=1 is set - the suite's conftest sets it; a deployment does not."""
    L = ctypes.CDLL(_native.LIB_PATH)
    L.sgpu_debug_host_threads.restype = ctypes.c_uint32
    assert L.sgpu_debug_host_threads() >= 1
    monkeypatch.delenv("SGPU_TEST_HOOKS")
    assert L.sgpu_debug_host_threads() == 0
    need = ctypes.c_uint64(0)
    assert L.sgpu_debug_pack_forward(None, None, 0, None, ctypes.byref(need)) == 1   # SGPU_EINVAL: "is a test hook"
    assert b"test hook" in ctypes.cast(L.sgpu_last_error, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_struct_layouts_match_header_sizes():
    # sizes implied by the header's field lists (natural alignment)
    assert ctypes.sizeof(IndexDesc) == 8 + 7 * 8 + 13 * 8 + 8
    assert ctypes.sizeof(BuildConfig) == 40
    assert ctypes.sizeof(_native.SearchParams) == 20
    assert ctypes.sizeof(_native.LaunchStats) == 20


def test_search_without_gpu_fails_loudly():
    if _native.device_count() > 0:
        pytest.skip("a GPU is present")
    off, comps, vals = random_dataset(1, 50, 32)
    ix = _native.NativeIndex.build(2, 32, off, comps, vals)
    with pytest.raises(_native.SeismicHipError) as e:
        ix.upload(0)
    assert e.value.status == 2   # SGPU_EDEVICE
    with pytest.raises(_native.SeismicHipError) as e:   # not uploaded -> no silent CPU path
        ix.search([1], [1.0], 10, 5, 0.7)
    assert e.value.status == 2


def test_desc_validation_rejects_corrupt_indexes():
    off, comps, vals = random_dataset(2, 200, 64)
    ix = _native.NativeIndex.build(2, 64, off, comps, vals, BuildConfig.defaults(n_postings=20))
    a = orc.desc_arrays(ix.desc)
    for field, mutate in [("post_doc", lambda x: x.__setitem__(0, 10 ** 6)),
                          ("sum_bid", lambda x: x.__setitem__(0, 65535)),
                          ("fwd_offsets", lambda x: x.__setitem__(1, 10 ** 9))]:
        d = IndexDesc.from_buffer_copy(ix.desc)
        arr = a[field].copy()
        mutate(arr)
        setattr(d, field, arr.ctypes.data_as(type(getattr(d, field))))
        with pytest.raises(_native.SeismicHipError) as e:
            _native.NativeIndex.from_desc(d)
        assert e.value.status == 1


def test_exact_search_matches_oracle_bruteforce():
    dim = 200
    off, comps, vals = random_dataset(3, 3000, dim, nnz_lo=5, nnz_hi=60)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=10))
    q_off, qc, qv = random_queries(4, 20, dim, 5, 40)
    sc, ids, n = ix.exact_search(q_off, qc, qv, 10)
    for i in range(20):
        c = qc[q_off[i]:q_off[i + 1]]
        v = qv[q_off[i]:q_off[i + 1]]
        es, ei = orc.exact_search(ix.desc, c, v, 10, orc.ORDER_SEQ)
        assert np.array_equal(ids[i, :n[i]], ei)
        assert np.array_equal(sc[i, :n[i]].view(np.uint32), es.view(np.uint32))


def test_resolve_query_tokens_semantics():
    tm = {"a": 3, "b": 1, "c": 2}
    c, v = _resolve(["a", "zzz", "b", "c"], [1.0, 9.0, 2.0, 3.0], tm)   # unknown dropped, sorted by id
    assert c.tolist() == [1, 2, 3] and v.tolist() == [2.0, 3.0, 1.0]
    assert seismic_amd.get_seismic_string() == "U30"


def test_toy_dataset_plumbing_and_golden(tmp_path):
    """BASELINE config 1: toy dataset indexed with the Python-default parameters; the oracle's
    answers on the product-built index equal the committed golden file."""
    import json
    ix = seismic_amd.SeismicIndex.build(os.path.join(GOLD, "toy", "documents.jsonl"), upload=False)
    exp = json.load(open(os.path.join(GOLD, "toy", "expected.json")))
    assert (ix.dim, ix.len) == (exp["dim"], exp["n_docs"]) == (1396, 20)
    assert ix.get_doc_ids_in_postings(0) is not None
    ids, vecs, _ = seismic_amd.index.read_jsonl(os.path.join(GOLD, "toy", "queries.jsonl"))
    for row, qv in zip(exp["queries"], vecs):
        qc, qw = _resolve(list(qv.keys()), list(qv.values()), ix._tm)
        for srt in (False, True):
            s, i = orc.search(ix._ix.desc, qc, qw, 10, 10, 0.7, srt)
            got = [[ix._doc_ids[int(x)], float(y)] for y, x in zip(s, i)]
            assert got == row["seismic_sorted_%s" % srt]
        # the empty document (id 18) is never retrieved
        assert all(d != "18" for d, _ in row["seismic_sorted_True"])
    # get(id): the stored document (reference src/pylib/mod.rs:157-165): component ids and f16 values as f32
    _, dvecs, _ = seismic_amd.index.read_jsonl(os.path.join(GOLD, "toy", "documents.jsonl"))
    for doc in (0, 7, 18, 19):
        gc, gv = ix.get(doc)
        want = sorted((ix._tm[t], float(np.float32(np.float16(v)))) for t, v in dvecs[doc].items())
        assert list(zip(gc, gv)) == want
    assert ix.get(18) == ([], []) and not ix.is_empty
    with pytest.raises(IndexError):
        ix.get(20)
    # persistence of the string-keyed wrapper
    ix.save(str(tmp_path / "toy"))
    jx = seismic_amd.SeismicIndex.load(str(tmp_path / "toy"), upload=False)
    assert jx._tm == ix._tm and jx._doc_ids == ix._doc_ids and jx.nnz == ix.nnz


def test_golden_synth_small_oracle():
    import json
    g = json.load(open(os.path.join(GOLD, "synth_small.json")))
    off, c, v = orc.csr([(d[0], d[1]) for d in g["docs"]])
    for build in ("oracle", "product"):
        if build == "oracle":
            desc = orc.OracleIndex(2, g["dim"], off, c, v, BuildConfig.defaults(**g["build"]))
            d = desc.desc
        else:
            ix = _native.NativeIndex.build(2, g["dim"], off, c, v, BuildConfig.defaults(**g["build"]))
            d = ix.desc
        for r in g["results"]:
            for (qc, qv), e in zip(g["queries"], r["per_query"]):
                s, i = orc.search(d, qc, qv, r["k"], r["query_cut"], r["heap_factor"], r["first_sorted"])
                assert [int(x) for x in i] == e["ids"]
                assert [int(x) for x in s.view(np.uint32)] == e["score_bits"]


def test_inner_format_roundtrip(tmp_path):
    off, comps, vals = random_dataset(5, 100, 500)
    p = str(tmp_path / "documents.bin")
    seismic_amd.write_inner_format(p, off, comps, vals)
    o2, c2, v2 = seismic_amd.read_inner_format(p)
    assert np.array_equal(off, o2) and np.array_equal(comps, c2) and np.array_equal(vals, v2)


def test_seismic_dataset_exact_search():
    ds = seismic_amd.SeismicDataset()
    ds.add_document("d0", ["x", "y"], [1.0, 2.0])
    ds.add_document("d1", ["y", "z"], [4.0, 5.0])
    ds.add_document("d2", [], [])
    assert ds.len == 3
    r = ds.search("q", np.array(["y", "nope"], dtype="U30"), np.array([2.0, 1.0], np.float32), 2)
    assert r == [("q", 8.0, "d1"), ("q", 4.0, "d0")]


def test_score_order_tolerance():
    """The canonical 16-lane accumulation order vs a plain left-to-right sum: the stated
    tolerance |d| <= 1e-5 * max(1, |s|) (vectorium's true order is not in the reference tree)."""
    dim = 300
    off, comps, vals = random_dataset(6, 2000, dim, nnz_lo=50, nnz_hi=280)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=5))
    q_off, qc, qv = random_queries(7, 8, dim, 20, 90)
    worst = 0.0
    for i in range(8):
        c = qc[q_off[i]:q_off[i + 1]]
        v = qv[q_off[i]:q_off[i + 1]]
        for doc in range(0, 2000, 37):
            a = orc.score_doc(ix.desc, doc, c, v, orc.ORDER_LANES16)
            b = orc.score_doc(ix.desc, doc, c, v, orc.ORDER_SEQ)
            worst = max(worst, abs(a - b) / max(1.0, abs(b)))
    assert worst <= 1e-5


def test_oracle_knn_restatement_properties():
    """Knn::new / Knn::refine restated (reference src/inverted_index.rs:448-500, 551-593); the
    reference holds no known-answer test for them, so the restatement is checked for the
    properties the source implies."""
    dim = 200
    off, comps, vals = random_dataset(91, 600, dim, nnz_lo=5, nnz_hi=50, empty_every=50)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=30))
    nknn = 5
    nb = orc.knn_build(ix.desc, nknn)
    assert len(nb) <= 600 * nknn and (nb < 600).all()
    # a document is never its own neighbour wherever the lists are full-length (no misalignment yet)
    full = len(nb) == 600 * nknn
    if full:
        assert (nb.reshape(600, nknn) != np.arange(600)[:, None]).all()
    q_off, qc, qv = random_queries(92, 25, dim, 3, 30)
    orc.knn_attach(nb, nknn)
    try:
        for i in range(25):
            c, v = qc[q_off[i]:q_off[i + 1]], qv[q_off[i]:q_off[i + 1]]
            s0, i0 = orc.search(ix.desc, c, v, 10, 2, 1.0, False)
            s1, i1 = orc.search(ix.desc, c, v, 10, 2, 1.0, False, n_knn=nknn)
            assert len(s1) >= len(s0)
            assert (s1[: len(s0)] >= s0 - 0).all()              # refinement can only improve each rank
            assert len(set(i1.tolist())) == len(i1)              # no document twice
            for d_, s_ in zip(i1, s1):                            # every score is the true inner product
                assert np.float32(orc.score_doc(ix.desc, int(d_), c, v)) == s_
    finally:
        orc.knn_attach(None, 0)


def test_index_file_keeps_the_knn_graph_and_rejects_corrupt_files(tmp_path):
    """host side of sgpu_index_save/load: the kNN graph is part of the file (the reference serialises
    InvertedIndexBase{.., knn}); a header that does not add up to the file size is an error, never a crash."""
    dim = 64
    off, comps, vals = random_dataset(81, 300, dim)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=20))
    rng = np.random.default_rng(5)
    nb = rng.integers(0, 300, 300 * 4).astype(np.uint32)
    ix.set_knn(nb, 4)                      # no device: host side only
    p = str(tmp_path / "a.idx")
    ix.save(p)
    ix2 = _native.NativeIndex.load(p)
    got, kd = ix2.get_knn()
    assert kd == 4 and np.array_equal(got, nb)
    from util import desc_equal
    desc_equal(ix.desc, ix2.desc)
    raw = bytearray(open(p, "rb").read())
    for name, edit in [("truncated", lambda b: b[:len(b) - 7]),
                       ("huge count", lambda b: b[:8 + 8 * 3] + (2 ** 62).to_bytes(8, "little") + b[8 + 8 * 4:]),
                       ("bad magic", lambda b: b"XXXXXXXX" + b[8:]),
                       ("knn id out of range", lambda b: b[:len(b) - 4] + (10 ** 6).to_bytes(4, "little"))]:
        q = str(tmp_path / "bad.idx")
        open(q, "wb").write(bytes(edit(raw)))
        with pytest.raises(_native.SeismicHipError) as e:
            _native.NativeIndex.load(q)
        assert e.value.status == 4, name   # SGPU_EIO


def test_results_tsv_and_accuracy(tmp_path):
    """perf_inverted_index's result dump (src/bin/perf_inverted_index.rs:223-235) and compute_accuracy
    of scripts/run_experiments.py:287-309."""
    from seismic_amd.index import accuracy, read_results_tsv, write_results_tsv
    sc = np.array([[3.5, 2.25, 0.0], [1.0, 0.0, 0.0]], np.float32)
    ids = np.array([[7, 3, 0], [11, 0, 0]], np.uint64)
    n = np.array([2, 1], np.uint32)
    p = str(tmp_path / "res.tsv")
    write_results_tsv(p, sc, ids, n)
    assert open(p).read() == "0\t7\t1\t3.5\n0\t3\t2\t2.25\n1\t11\t1\t1\n"
    res = read_results_tsv(p)
    assert res == {0: [7, 3], 1: [11]}
    gt = {0: [7, 9], 1: [11], 2: [5]}
    assert accuracy(res, gt) == pytest.approx(2 / 4)


def test_accuracy_against_values_the_reference_function_returned():
    """`accuracy` over `read_results_tsv` equals what the reference's own compute_accuracy
    (scripts/run_experiments.py:287-309, executed by tests/golden/make_fixtures.py in the build container)
    returned for the same files: unanswered queries, queries without ground truth, fewer than k results,
    a ground-truth row written twice."""
    import json
    from seismic_amd.index import accuracy, read_results_tsv
    exp = json.load(open(os.path.join(GOLD, "accuracy", "expected.json")))["accuracy"]
    assert len(exp) == 4
    for case, want in exp.items():
        got = accuracy(read_results_tsv(os.path.join(GOLD, "accuracy", case + "_results.tsv")),
                       read_results_tsv(os.path.join(GOLD, "accuracy", case + "_groundtruth.tsv")))
        assert got == want, (case, got, want)


def test_inner_format_against_bytes_written_by_the_reference_converter(tmp_path):
    """tests/golden/toy_inner/* was written by the reference's scripts/convert_json_to_inner_format.py
    (tests/golden/make_fixtures.py runs it where /root/reference exists). The product must read those
    bytes into exactly the CSR its own jsonl ingestion produces (same sorted token numbering, unknown
    query tokens dropped), and write the same bytes back."""
    import json
    gold = os.path.join(GOLD, "toy_inner")
    off, comps, vals = seismic_amd.read_inner_format(os.path.join(gold, "documents.bin"))
    ids, vecs, _ = seismic_amd.index.read_jsonl(os.path.join(GOLD, "toy", "documents.jsonl"))
    tm = seismic_amd.index._token_map(vecs)
    assert tm == json.load(open(os.path.join(gold, "token_to_id_mapping.json")))
    o2, c2, v2 = seismic_amd.index._to_csr(vecs, tm)
    assert np.array_equal(off, o2) and np.array_equal(comps, c2) and np.array_equal(vals.view(np.uint32), v2.view(np.uint32))
    assert len(off) - 1 == 20
    for d in range(20):   # components ascending inside every vector, as the reference's reader requires
        assert (np.diff(comps[off[d]:off[d + 1]].astype(np.int64)) > 0).all()
    qo, qc, qv = seismic_amd.read_inner_format(os.path.join(gold, "queries.bin"))
    _, qvecs, _ = seismic_amd.index.read_jsonl(os.path.join(GOLD, "toy", "queries.jsonl"))
    assert len(qo) - 1 == len(qvecs) == 5
    for i, qd in enumerate(qvecs):
        c, v = _resolve(list(qd.keys()), list(qd.values()), tm)
        assert np.array_equal(c, qc[qo[i]:qo[i + 1]]) and np.array_equal(v.view(np.uint32), qv[qo[i]:qo[i + 1]].view(np.uint32))
    for name, arrs in (("documents.bin", (off, comps, vals)), ("queries.bin", (qo, qc, qv))):
        p = str(tmp_path / name)
        seismic_amd.write_inner_format(p, *arrs)
        assert open(p, "rb").read() == open(os.path.join(gold, name), "rb").read()
    with pytest.raises(IOError):
        seismic_amd.read_inner_format(str(tmp_path / "missing.bin"))
    open(str(tmp_path / "trunc.bin"), "wb").write(open(os.path.join(gold, "documents.bin"), "rb").read()[:1000])
    with pytest.raises(IOError):
        seismic_amd.read_inner_format(str(tmp_path / "trunc.bin"))


def test_convert_to_fixed_u8_host_side(tmp_path):
    """sgpu_index_convert (the reference's convert_dataset_into, src/pylib/dotvbyte.rs:208-213): same lists,
    blocks and summaries; forward values become u8 codes with a power-of-two step; identical to the oracle's
    restatement; survives save/load; exact search over the converted index equals the oracle's."""
    dim = 150
    off, comps, vals = random_dataset(91, 2000, dim, nnz_lo=5, nnz_hi=60)
    cfg = BuildConfig.defaults(n_postings=40)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, cfg)
    u8 = ix.convert(1)
    oix = orc.OracleIndex(2, dim, off, comps, vals, cfg).convert_fixedu8()
    from util import desc_equal
    desc_equal(u8.desc, oix.desc)
    assert u8.desc.value_type == 1 and u8.desc.val_scale == oix.desc.val_scale
    import math
    assert math.frexp(u8.desc.val_scale)[0] == 0.5                       # a power of two
    a, b = orc.desc_arrays(ix.desc), orc.desc_arrays(u8.desc)
    for k_ in a:
        if k_ != "fwd_vals":
            assert np.array_equal(a[k_], b[k_]), k_
    f16 = a["fwd_vals"].view(np.float16).astype(np.float32)
    deq = b["fwd_vals"].astype(np.float32) * np.float32(u8.desc.val_scale)
    assert b["fwd_vals"].dtype == np.uint8 and np.abs(deq - f16).max() <= u8.desc.val_scale / 2 + 1e-7
    assert 255 * u8.desc.val_scale >= f16.max() > 255 * u8.desc.val_scale / 2
    p = str(tmp_path / "u8.idx")
    u8.save(p)
    loaded = _native.NativeIndex.load(p)
    desc_equal(loaded.desc, u8.desc)
    q_off, qc, qv = random_queries(92, 12, dim, 5, 40)
    sc, ids, n = u8.exact_search(q_off, qc, qv, 10)
    for i in range(12):
        es, ei = orc.exact_search(u8.desc, qc[q_off[i]:q_off[i + 1]], qv[q_off[i]:q_off[i + 1]], 10, orc.ORDER_SEQ)
        assert np.array_equal(ids[i, :n[i]], ei) and np.array_equal(sc[i, :n[i]], es)
    w = seismic_amd.SeismicIndexRaw(u8, upload=False)                     # get() of a fixed-u8 index: code * step
    gc, gv = w.get(5)
    assert gc == [int(c) for c in comps[off[5]:off[6]]]
    assert np.array_equal(np.array(gv, np.float32), deq[off[5]:off[6]])
    back = u8.convert(0)                                                  # and back to f16: values are the dequantised codes
    assert back.desc.value_type == 0
    lv = _native.NativeIndex.build(4, 70000, *random_dataset(93, 50, 70000)).convert(1)   # u32 components as well
    assert lv.desc.value_type == 1 and lv.desc.comp_width == 4                            # ("fixedu8" + u32: perf_inverted_index.rs:125-126)


def test_exact_search_validates_its_queries():
    off, comps, vals = random_dataset(95, 100, 32)
    ix = _native.NativeIndex.build(2, 32, off, comps, vals, BuildConfig.defaults(n_postings=10))
    ok = ix.exact_search(np.array([0, 2], np.uint64), np.array([1, 5], np.uint32), np.array([1, 1], np.float32), 5)
    assert ok[2][0] == 5
    for q_off, c, v in [([0, 2], [5, 1], [1, 1]),          # not ascending
                        ([0, 2], [1, 40], [1, 1]),         # component >= dim
                        ([1, 2], [1, 5], [1, 1]),          # q_off[0] != 0
                        ([0, 2], [1, 5], [1, np.nan])]:    # NaN weight
        with pytest.raises(_native.SeismicHipError) as e:
            ix.exact_search(np.array(q_off, np.uint64), np.array(c, np.uint32), np.array(v, np.float32), 5)
        assert e.value.status == 1


def test_round_1_index_files_still_load(tmp_path):
    """SGPUIDX1 files (10-word header, f16 values) written by round 1 load as what they hold; a dataset
    read with offsets but without value buffers is refused instead of written through a null pointer."""
    off, comps, vals = random_dataset(3, 400, 300, nnz_lo=2, nnz_hi=30)
    ix = _native.NativeIndex.build(2, 300, off, comps, vals, BuildConfig.defaults(n_postings=50, centroid_fraction=0.2))
    p2, p1 = str(tmp_path / "v2.idx"), str(tmp_path / "v1.idx")
    ix.save(p2)
    raw = open(p2, "rb").read()
    assert raw[:8] == b"SGPUIDX2"
    open(p1, "wb").write(b"SGPUIDX1" + raw[8:8 + 80] + raw[8 + 96:])      # drop the two words round 2 added
    old = _native.NativeIndex.load(p1)
    from util import desc_equal
    desc_equal(old.desc, ix.desc)
    assert old.desc.value_type == 0
    open(p1, "wb").write(b"SGPUIDX0" + raw[8:])
    with pytest.raises(_native.SeismicHipError):
        _native.NativeIndex.load(p1)
    dp = str(tmp_path / "d.bin")
    _native.write_inner_format(dp, off, comps, vals)
    n, nnz = ctypes.c_uint64(len(off) - 1), ctypes.c_uint64(len(comps))
    o = np.zeros(len(off), np.uint64)
    c = np.zeros(len(comps), np.uint32)
    st = _native.lib().sgpu_dataset_read(os.fsencode(dp), ctypes.byref(n), ctypes.byref(nnz), o.ctypes.data_as(ctypes.c_void_p),
                                         c.ctypes.data_as(ctypes.c_void_p), None)
    assert st == 1      # SGPU_EINVAL


def test_default_host_thread_count_override_keeps_the_index(monkeypatch):
    """num_threads == 0 takes the hardware threads capped by the container's CPU quota (common.hpp host_threads);
    SGPU_HOST_THREADS overrides it. Whatever the team size, the index and the exact search are the same bytes."""
    from util import desc_equal
    dim = 300
    off, comps, vals = random_dataset(11, 6000, dim, nnz_lo=5, nnz_hi=60)
    base = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=40))
    q_off, qc, qv = random_queries(12, 16, dim, 5, 40)
    ref = base.exact_search(q_off, qc, qv, 10)
    for nt in ("1", "3", "13"):
        monkeypatch.setenv("SGPU_HOST_THREADS", nt)
        ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=40))
        desc_equal(base.desc, ix.desc)
        got = ix.exact_search(q_off, qc, qv, 10)
        for a, b in zip(ref, got):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_chunk_plan_of_a_call_covers_every_query_once():
    """How sgpu_batch_search cuts a call into launches (abi.cpp chunk_jobs / chunk_bounds, through the debug export):
    whatever the sizes, the launches are contiguous, in order, non-empty and cover [0, nq) exactly once."""
    L = ctypes.CDLL(_native.LIB_PATH)
    L.sgpu_debug_chunk_plan.restype = ctypes.c_uint32
    L.sgpu_debug_chunk_plan.argtypes = [ctypes.c_uint32] * 6 + [ctypes.POINTER(ctypes.c_uint32)]
    bounds = (ctypes.c_uint32 * 16)()
    sizes = list(range(0, 70)) + [255, 256, 257, 511, 599, 600, 1199, 1200, 1201, 1250, 1799, 1800, 2399, 2400, 2500, 4095,
                                  4096, 4097, 9999, 10000, 65535, 1000003, 2**31 - 1, 2**32 - 1]
    for nq in sizes:
        for chunk_min, chunk_max in ((600, 4), (2048, 4), (1, 8), (0, 4), (300, 2)):
            for want_tail, coop_max in ((0, 256), (256, 256), (64, 256), (300, 256), (128, 0)):
                for lanes in (1, 2, 3, 8):
                    n = L.sgpu_debug_chunk_plan(nq, chunk_min, chunk_max, want_tail, coop_max, lanes, bounds)
                    assert 1 <= n <= min(8, lanes), (nq, chunk_min, chunk_max, want_tail, coop_max, lanes, n)
                    b = [bounds[i] for i in range(2 * n)]
                    assert b[0] == 0 and b[-1] == nq
                    for j in range(n):
                        assert b[2 * j] <= b[2 * j + 1]
                        if n > 1:
                            assert b[2 * j] < b[2 * j + 1]          # no empty launch once a call is cut
                            assert b[2 * j + 1] - b[2 * j] <= nq // 2 + 1 or b[2 * j] == 0   # rebased offsets fit their buffer
                        if j:
                            assert b[2 * j] == b[2 * j - 1]
    # the defaults: 1250 queries -> two launches, 10 000 -> four, 1000 -> one; the cooperative tail only when asked for
    assert L.sgpu_debug_chunk_plan(1250, 600, 4, 0, 256, 8, bounds) == 2 and bounds[1] == 625
    assert L.sgpu_debug_chunk_plan(10000, 600, 4, 0, 256, 8, bounds) == 4 and bounds[1] == 2500
    assert L.sgpu_debug_chunk_plan(1000, 600, 4, 0, 256, 8, bounds) == 1
    assert L.sgpu_debug_chunk_plan(1000, 600, 4, 256, 256, 8, bounds) == 2 and [bounds[i] for i in range(4)] == [0, 744, 744, 1000]


def test_cpu_quota_of_the_container_caps_the_default_host_team(tmp_path, monkeypatch):
    """cgroup v2 `cpu.max` / v1 `cfs_quota_us`: the library's default team size (num_threads == 0) and bench.py's CPU
    baseline both stay within the quota; "max" / -1 / no files mean no quota."""
    import importlib.util
    L = ctypes.CDLL(_native.LIB_PATH)
    L.sgpu_debug_host_threads.restype = ctypes.c_uint32
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv("SGPU_HOST_THREADS", raising=False)
    empty = tmp_path / "none"
    empty.mkdir()
    monkeypatch.setenv("SGPU_CGROUP_ROOT", str(empty))
    free = L.sgpu_debug_host_threads()          # no quota: what OpenMP would take
    assert free >= 1 and bench.cpu_quota(str(empty)) is None
    v2 = tmp_path / "v2"
    v2.mkdir()
    for text, want in (("max 100000\n", None), ("300000 100000\n", 3.0), ("150000 100000\n", 1.5), ("1600000 100000\n", 16.0)):
        (v2 / "cpu.max").write_text(text)
        monkeypatch.setenv("SGPU_CGROUP_ROOT", str(v2))
        assert bench.cpu_quota(str(v2)) == want
        cap = free if want is None else min(free, int(want + 0.999))
        assert L.sgpu_debug_host_threads() == cap, (text, free)
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    for text, want in (("-1\n", None), ("200000\n", 2.0)):
        (v1 / "cpu" / "cpu.cfs_quota_us").write_text(text)
        monkeypatch.setenv("SGPU_CGROUP_ROOT", str(v1))
        assert bench.cpu_quota(str(v1)) == want
        assert L.sgpu_debug_host_threads() == (free if want is None else min(free, 2))
    monkeypatch.setenv("SGPU_HOST_THREADS", "5")   # the override wins over everything
    assert L.sgpu_debug_host_threads() == 5


def test_dotvbyte_records_decode_to_their_documents(tmp_path):
    """SGPU_VAL_DOTVBYTE: every record the product packs (per 8-element slice the first component in 16 bits, three
    12-bit and four 11-bit gaps; raw fallback for a document with a gap that does not fit its field) decodes - by the
    oracle's independent restatement of the layout -
    to the document it was packed from; padding elements leave component and score alone; the conversions and the
    index file keep codes and components; what the format refuses is refused loudly."""
    rng = np.random.default_rng(11)
    dim = 60000
    lens = [0, 1, 7, 8, 9, 16, 127, 128, 129, 255, 256, 257, 300, 1000]
    vecs = []
    for d in range(600):
        n = lens[d % len(lens)]
        hi = [3000, 9000, dim][d % 3]            # all gaps small / some wide / mostly wide for short documents
        c = np.sort(rng.choice(hi, min(n, hi), replace=False)).astype(np.uint32)
        if d == 5 and n:
            c[0] = 0                              # first component 0
        if d % 50 == 3 and n > 1:
            c = np.sort(np.unique(np.concatenate([c[:-1], [dim - 1]]))).astype(np.uint32)   # a last gap that may be wide
        vecs.append((c, (rng.exponential(0.5, len(c)) + 0.01).astype(np.float32)))
    # exactly at the limits of the fields: a 12-bit gap (elements 1 .. 3 of a slice) of 4095 fits, 4096 does not; an
    # 11-bit gap (elements 4 .. 7) of 2047 fits, 2048 does not; a slice's first component is absolute (any u16 value)
    vecs.append((np.array([50000, 50000 + 4095], np.uint32), np.array([1.0, 2.0], np.float32)))
    vecs.append((np.array([10, 10 + 4096], np.uint32), np.array([1.0, 2.0], np.float32)))
    vecs.append((np.array([0, 1, 2, 3, 3 + 2047], np.uint32), np.arange(1, 6).astype(np.float32)))
    vecs.append((np.array([0, 1, 2, 3, 3 + 2048], np.uint32), np.arange(1, 6).astype(np.float32)))
    vecs.append((np.array([0, 1, 2, 3, 4, 5, 6, 7, 7 + 50000, 7 + 50001], np.uint32), np.arange(1, 11).astype(np.float32)))   # a new slice: any jump
    off, comps, vals = orc.csr(vecs)
    f16 = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=50))
    dvb = f16.convert(2)
    u8 = f16.convert(1)
    d = dvb.desc
    assert d.value_type == 2 and d.val_scale == u8.desc.val_scale and d.nnz == u8.desc.nnz
    a2, a1 = orc.desc_arrays(d), orc.desc_arrays(u8.desc)
    assert np.array_equal(a2["fwd_vals"], a1["fwd_vals"]) and np.array_equal(a2["fwd_comps"], a1["fwd_comps"])
    L = _native.lib()
    L.sgpu_debug_pack_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    need = ctypes.c_uint64(0)
    assert L.sgpu_debug_pack_forward(dvb.h, None, 0, None, ctypes.byref(need)) == 0
    fwd = np.zeros(need.value + 16, np.uint8)
    refs = np.zeros(d.n_docs, np.uint64)
    assert L.sgpu_debug_pack_forward(dvb.h, fwd.ctypes.data_as(ctypes.c_void_p), need.value, refs.ctypes.data_as(ctypes.c_void_p), ctypes.byref(need)) == 0
    need_u8 = ctypes.c_uint64(0)
    assert L.sgpu_debug_pack_forward(u8.h, None, 0, None, ctypes.byref(need_u8)) == 0
    assert need.value < need_u8.value            # the stream is smaller than the raw components
    O = orc.lib()
    O.orc_dvb_decode_record.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    fo = a2["fwd_offsets"]
    n_raw = 0
    for doc in range(int(d.n_docs)):
        ref = int(refs[doc])
        ln, raw, off16 = ref & 0x7fff, (ref >> 15) & 1, ref >> 16
        s, e = int(fo[doc]), int(fo[doc + 1])
        assert ln == e - s
        c_doc = a2["fwd_comps"][s:e].astype(np.int64)
        gaps = np.diff(c_doc)                     # gaps[i - 1] = gap of element i; elements 8s are stored absolutely
        pos = np.arange(1, ln) & 7
        wide = ln > 1 and bool(np.any((pos != 0) & (gaps >= np.where(pos <= 3, 4096, 2048))))
        assert bool(raw) == wide, (doc, raw, c_doc[:4])
        n_raw += raw
        co, vo = np.zeros(max(ln, 1), np.uint16), np.zeros(max(ln, 1), np.uint8)
        rc = O.orc_dvb_decode_record(fwd.ctypes.data + off16 * 16, ln, raw, co.ctypes.data_as(ctypes.c_void_p), vo.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, (doc, rc)
        assert np.array_equal(co[:ln], a2["fwd_comps"][s:e]) and np.array_equal(vo[:ln], a2["fwd_vals"][s:e]), doc
    assert 0 < n_raw < d.n_docs                   # both record forms occur
    # the index file keeps the value type; DotVByte <-> fixed-u8 <-> f16 conversions keep the codes
    path = str(tmp_path / "dvb.idx")
    dvb.save(path)
    back = _native.NativeIndex.load(path)
    assert back.desc.value_type == 2 and np.array_equal(orc.desc_arrays(back.desc)["fwd_vals"], a2["fwd_vals"])
    again = dvb.convert(1)   # (kept alive: desc_arrays views the index's own memory)
    assert again.desc.value_type == 1 and np.array_equal(orc.desc_arrays(again.desc)["fwd_vals"], a1["fwd_vals"])
    # the f16 index itself can take the same component stream in front of its binary16 values (r05: the sliced internal
    # layout, SGPU_FWD_STREAM=sliced in libraries built WITH_F16S=1; the host-side packing is always there): every record decodes to
    # its document with the SAME raw / packed decision per document, and the store is smaller than the plain one
    need_s, need_p = ctypes.c_uint64(0), ctypes.c_uint64(0)
    assert L.sgpu_debug_pack_forward(f16.h, None, 0, None, ctypes.byref(need_p)) == 0
    os.environ["SGPU_FWD_STREAM"] = "sliced"
    try:
        assert L.sgpu_debug_pack_forward(f16.h, None, 0, None, ctypes.byref(need_s)) == 0
        assert need_s.value < need_p.value
        fwd_s = np.zeros(need_s.value + 16, np.uint8)
        refs_s = np.zeros(d.n_docs, np.uint64)
        assert L.sgpu_debug_pack_forward(f16.h, fwd_s.ctypes.data_as(ctypes.c_void_p), need_s.value, refs_s.ctypes.data_as(ctypes.c_void_p),
                                         ctypes.byref(need_s)) == 0
    finally:
        del os.environ["SGPU_FWD_STREAM"]
    O.orc_slices_decode_record.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    af = orc.desc_arrays(f16.desc)
    for doc in range(int(d.n_docs)):
        ref, ref_d = int(refs_s[doc]), int(refs[doc])
        ln, raw, off16 = ref & 0x7fff, (ref >> 15) & 1, ref >> 16
        assert (ln, raw) == (ref_d & 0x7fff, (ref_d >> 15) & 1), doc     # same length, same raw / packed decision as the DotVByte index
        s, e = int(fo[doc]), int(fo[doc + 1])
        co, vo = np.zeros(max(ln, 1), np.uint16), np.zeros(max(ln, 1), np.uint16)
        rc = O.orc_slices_decode_record(fwd_s.ctypes.data + off16 * 16, ln, raw, 2, co.ctypes.data_as(ctypes.c_void_p), vo.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, (doc, rc)
        assert np.array_equal(co[:ln], af["fwd_comps"][s:e]) and np.array_equal(vo[:ln], af["fwd_vals"][s:e].view(np.uint16)), doc
    # u32 components have no DotVByte form (the reference's class is u16-only)
    w = _native.NativeIndex.build(4, 70000, *random_dataset(3, 50, 70000))
    with pytest.raises(_native.SeismicHipError) as ei:
        w.convert(2)
    assert ei.value.status == 1


def test_hashed_row_directory_finds_every_summary_row_and_nothing_else():
    """DevView::row_dir (r05): stage 1 finds the summary row of (posting list, query component) in the bucket
    row_dir_bucket(list << 16 | component) - or a following one - of a table of 4-slot buckets. Every row of the index must be
    found with its {first entry, entries, split point}, absent pairs must end on an empty slot, and the table is at
    most 60 % full. The lookup below restates the kernel's (search_kernel.inc: build_row_table)."""
    dim = 3000
    off, comps, vals = random_dataset(5, 20000, dim, nnz_lo=8, nnz_hi=120)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=300, centroid_fraction=0.2))
    a = orc.desc_arrays(ix.desc)
    L = _native.lib()
    L.sgpu_debug_row_dir.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    nw, bits = ctypes.c_uint64(0), ctypes.c_uint32(0)
    assert L.sgpu_debug_row_dir(ix.h, None, 0, ctypes.byref(nw), ctypes.byref(bits)) == 0
    n_rows = int(ix.desc.n_rows)
    nbk = bits.value                   # buckets of four slots, at most 60 % full
    assert nbk > 0 and nw.value == 16 * nbk and 0 < n_rows <= 0.6 * 4 * nbk + 4
    tab = np.zeros(nw.value, np.uint32)
    assert L.sgpu_debug_row_dir(ix.h, tab.ctypes.data_as(ctypes.c_void_p), nw.value, ctypes.byref(nw), ctypes.byref(bits)) == 0
    tab = tab.reshape(-1, 4, 4)     # [bucket][slot][word]

    def lookup(key):
        b = (((key * 2654435761) & 0xffffffff) * nbk) >> 32
        for _ in range(nbk + 1):
            keys = tab[b, :, 0]
            hit = np.nonzero(keys == key)[0]
            if len(hit):
                return tab[b, hit[0]]
            if np.any(keys == 0xffffffff):
                return None
            b = 0 if b + 1 == nbk else b + 1
        raise AssertionError("probe sequence does not end")

    lrs, rc, rp = a["list_row_start"], a["row_comp"], a["row_ptr"]
    sb, lbs = a["sum_bid"], a["list_block_start"]
    assert int(np.count_nonzero(tab[:, :, 0] != 0xffffffff)) == n_rows
    rng = np.random.default_rng(3)
    lists = [c for c in range(dim) if lrs[c + 1] > lrs[c]]
    for c in rng.choice(lists, 60, replace=False):
        present = set()
        nb = int(lbs[c + 1] - lbs[c])
        for r in range(int(lrs[c]), int(lrs[c + 1])):
            e = lookup((int(c) << 16) | int(rc[r]))
            assert e is not None, (c, r)
            start = int(e[1]) | ((int(e[2]) & 0xffff) << 32)
            ln, mid = int(e[2]) >> 16, int(e[3]) & 0xffff
            assert (start, ln) == (int(rp[r]), int(rp[r + 1] - rp[r])), (c, r)
            assert mid == int(np.searchsorted(sb[rp[r]:rp[r + 1]], (nb + 1) // 2)), (c, r)
            present.add(int(rc[r]))
        for x in rng.integers(0, dim, 40):
            if int(x) not in present:
                assert lookup((int(c) << 16) | int(x)) is None
    # u32 components keep the binary search
    w = _native.NativeIndex.build(4, 70000, *random_dataset(3, 50, 70000))
    assert L.sgpu_debug_row_dir(w.h, None, 0, ctypes.byref(nw), ctypes.byref(bits)) == 0 and bits.value == 0


def test_launch_plan_orders_and_sizes_a_batch_for_every_query_cut():
    """The host-side launch plan (which lists each query will walk -> LDS need; longest-expected-first order) against a
    numpy restatement of the kernel's selection rule (query_cut heaviest components by f32::total_cmp, ties by ascending
    component), for query_cut 0 (no list: the reference's k_largest_by(0)), small cuts (insertion top-k) and cuts above 16
    (the general path), on queries with tied and negative weights."""
    rng = np.random.default_rng(21)
    dim = 400
    off, comps, vals = random_dataset(22, 3000, dim, nnz_lo=4, nnz_hi=80)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=60, centroid_fraction=0.3))
    a = orc.desc_arrays(ix.desc)
    lbs, bps = a["list_block_start"].astype(np.int64), a["block_post_start"].astype(np.int64)
    nb = np.diff(lbs)
    npost = bps[lbs[1:]] - bps[lbs[:-1]]
    qs = []
    for i in range(200):
        n = int(rng.integers(0, 60))
        c = np.sort(rng.choice(dim, n, replace=False)).astype(np.uint32)
        v = rng.choice([0.5, 1.0, 2.0, -1.0, 3.25], n).astype(np.float32) if i % 2 else rng.normal(0, 1, n).astype(np.float32)
        qs.append((c, v))
    q_off, qc, qv = orc.csr(qs)
    L = _native.lib()
    L.sgpu_debug_plan.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                  ctypes.c_void_p, ctypes.c_void_p]

    def total_key(x):   # f32::total_cmp as a signed integer key
        b = np.float32(x).view(np.int32).astype(np.int64)
        return int(b ^ (((b >> 31) & 0xffffffff) >> 1)) if b >= 0 else int(np.int32(b ^ 0x7fffffff))

    for cut in (0, 1, 4, 16, 17, 40, 100):
        order = np.zeros(len(qs), np.uint32)
        out3 = np.zeros(3, np.uint32)
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        assert L.sgpu_debug_plan(ix.h, p(q_off), p(qc), p(qv), len(qs), cut, p(order), p(out3)) == 0
        cost, dots, first_nb, any_nb = [], 1, 0, 1
        for c, v in qs:
            sel = sorted(range(len(c)), key=lambda i: (-total_key(v[i]), int(c[i])))[:cut]
            cost.append(int(npost[c[sel]].sum()) if sel else 0)
            dots = max(dots, int(nb[c[sel]].sum()) if sel else 0)
            if sel:
                first_nb = max(first_nb, int(nb[c[sel[0]]]))
                any_nb = max(any_nb, int(nb[c[sel]].max()))
        assert sorted(order.tolist()) == list(range(len(qs)))                       # a permutation ...
        want = sorted(range(len(qs)), key=lambda i: (-cost[i], i))                  # ... longest expected first, ties in input order
        assert order.tolist() == want, cut
        assert (int(out3[0]), int(out3[1]), int(out3[2])) == (dots, first_nb, any_nb), (cut, out3, dots, first_nb, any_nb)
