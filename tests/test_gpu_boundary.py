"""Drop-in boundary on the GPU: concurrency, replicas, query-time edge semantics, the raw classes.
Run with `-m gpu` on an MI355X."""
import os
import threading

import numpy as np
import pytest

import orc
import seismic_amd
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import random_dataset, random_queries

pytestmark = pytest.mark.gpu


def _index(seed=61, n_docs=5000, dim=400):
    off, comps, vals = random_dataset(seed, n_docs, dim, nnz_lo=8, nnz_hi=150)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals,
                                   BuildConfig.defaults(n_postings=120, centroid_fraction=0.2, summary_energy=0.5,
                                                        max_fraction=6.0))
    return ix, dim


def _same(gpu, cpu):
    gs, gi, gn = gpu
    cs, ci, cn = cpu
    assert np.array_equal(gn, cn)
    for q in range(len(gn)):
        n = int(gn[q])
        assert np.array_equal(gi[q, :n], ci[q, :n]), q
        assert np.array_equal(gs[q, :n].view(np.uint32), cs[q, :n].view(np.uint32)), q


def test_the_library_loaded_on_this_box_is_the_build_of_this_tree():
    """(the GPU box runs a prebuilt .so: this is where a stale one would show)"""
    info = _native.build_info()
    print(info)
    if os.environ.get("SGPU_LIB"):
        pytest.skip("an experiment library was asked for: " + info)
    assert info.startswith("sources %s arch gfx950 extra [] " % _native.source_fingerprint()), info


def test_query_cut_zero_selects_no_list():
    """k_largest_by(0) selects nothing: the reference returns an empty result (src/inverted_index.rs:187-190)."""
    ix, dim = _index()
    ix.upload(0)
    q = random_queries(62, 9, dim, 3, 40)
    gs, gi, gn = ix.batch_search(*q, 10, 0, 0.9, False)
    assert (gn == 0).all()
    c = orc.batch_search(ix.desc, *q, 10, 0, 0.9, False)
    assert (c[2] == 0).all()


@pytest.mark.parametrize("hf", [-0.5, -2.0, 0.0])
@pytest.mark.parametrize("scale", [1.0, -1.0])
def test_any_heap_factor_matches_the_oracle(hf, scale):
    """heap_factor * threshold falls as the threshold rises when the factor is negative (or the
    scores are): the block filter must not use an older threshold then; the replay decides."""
    dim = 300
    off, comps, vals = random_dataset(63, 4000, dim, nnz_lo=8, nnz_hi=120, value_scale=1.0)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals * 1.0,
                                   BuildConfig.defaults(n_postings=100, centroid_fraction=0.2, summary_energy=0.5,
                                                        max_fraction=6.0)).upload(0)
    q_off, qc, qv = random_queries(64, 40, dim, 3, 50)
    qv = qv * scale   # negative query weights -> negative scores and thresholds
    for srt in (False, True):
        g = ix.batch_search(q_off, qc, qv, 10, 6, hf, srt)
        c = orc.batch_search(ix.desc, q_off, qc, qv, 10, 6, hf, srt)[:3]
        _same(g, c)


def test_four_host_threads_on_one_index():
    """sgpu_search / sgpu_batch_search are re-entrant on a shared index (the reference's search takes
    &self; S: Sync, src/index_traits.rs:106-113): 4 threads hammer one index with different
    batches, batch sizes and parameters; every answer equals the oracle's."""
    ix, dim = _index(65, 8000, 500)
    ix.upload(0)
    jobs = []
    for t in range(4):
        q = random_queries(70 + t, 48 + 16 * t, dim, 3, 60)
        k, cut, hf, srt = [(10, 4, 1.0, False), (7, 6, 0.8, True), (100, 5, 0.9, False), (1, 3, 0.7, True)][t]
        exp = orc.batch_search(ix.desc, *q, k, cut, hf, srt)[:3]
        jobs.append((q, (k, cut, hf, srt), exp))
    errors = []

    def worker(t):
        q, (k, cut, hf, srt), exp = jobs[t]
        try:
            for it in range(25):
                if it % 5 == 4:   # single-query calls in between
                    i = it % (len(q[0]) - 1)
                    s, ids = ix.search(q[1][q[0][i]:q[0][i + 1]], q[2][q[0][i]:q[0][i + 1]], k, cut, hf, srt)
                    n = int(exp[2][i])
                    assert np.array_equal(ids, exp[1][i, :n]) and np.array_equal(s.view(np.uint32), exp[0][i, :n].view(np.uint32))
                else:
                    _same(ix.batch_search(*q, k, cut, hf, srt), exp)
        except Exception as e:   # noqa: BLE001 - reported to the main thread
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_replicas_shard_a_batch_in_process():
    """sgpu_index_upload_many: replica 0 from the host, the others GPU to GPU; sgpu_batch_search cuts
    the batch into contiguous shards, one host thread per replica, rows in input order. (On a 1-GPU
    box the replicas share device 0: same code path, same peer-copy call.)"""
    ix, dim = _index(66, 6000, 450)
    ndev = _native.device_count()
    devs = list(range(ndev)) if ndev >= 2 else [0, 0, 0]
    ix.upload_many(devs)
    assert ix.replicas == len(devs)
    q = random_queries(67, 101, dim, 3, 50)   # 101: uneven shards
    for (k, cut, hf, srt) in [(10, 4, 1.0, False), (20, 8, 0.8, True)]:
        _same(ix.batch_search(*q, k, cut, hf, srt), orc.batch_search(ix.desc, *q, k, cut, hf, srt)[:3])
    # calls too small to shard (single queries of a serving loop) go to the replicas in turn
    q_off, q_comps, q_vals = q
    want = orc.batch_search(ix.desc, *q, 10, 4, 1.0, False)
    for i in range(2 * len(devs) + 1):
        a, e = int(q_off[i]), int(q_off[i + 1])
        sc, ids, n = ix.batch_search(np.array([0, e - a], np.uint64), q_comps[a:e], q_vals[a:e], 10, 4, 1.0, False)
        assert int(n[0]) == int(want[2][i])
        assert np.array_equal(ids[0, :n[0]], want[1][i, :n[0]]) and np.array_equal(sc[0, :n[0]], want[0][i, :n[0]])
    # errors name the query by its index in the caller's batch, not in a shard; offsets are checked before the cut
    bad_c = q_comps.copy()
    bad_c[int(q_off[77])] = dim + 1
    with pytest.raises(_native.SeismicHipError) as e:
        ix.batch_search(q_off, bad_c, q_vals, 10, 4, 1.0, False)
    assert "query 77" in str(e.value)
    bad_off = q_off.copy()
    bad_off[50] = bad_off[49] - 1 if bad_off[49] else bad_off[51] + 1
    with pytest.raises(_native.SeismicHipError):
        ix.batch_search(bad_off, q_comps, q_vals, 10, 4, 1.0, False)
    # device-resident batches on a chosen replica
    b = _native.DeviceBatch(ix, *q, 10, replica=len(devs) - 1)
    b.run(10, 4, 1.0, False)
    _same(b.fetch(10), orc.batch_search(ix.desc, *q, 10, 4, 1.0, False)[:3])
    ix.upload(0)   # back to one replica
    assert ix.replicas == 1
    _same(ix.batch_search(*q, 10, 4, 1.0, False), orc.batch_search(ix.desc, *q, 10, 4, 1.0, False)[:3])


def test_knn_graph_survives_save_and_load(tmp_path):
    """The reference serialises the kNN graph inside the index file (InvertedIndexBase{.., knn}):
    build(nknn) -> save -> load keeps n_knn refinement working."""
    ix, dim = _index(68, 3000, 300)
    ix.upload(0)
    ix.build_knn(5)
    q = random_queries(69, 30, dim, 3, 40)
    before = ix.batch_search(*q, 10, 3, 0.9, False, n_knn=3)
    plain = ix.batch_search(*q, 10, 3, 0.9, False)
    assert not np.array_equal(before[1], plain[1])            # the refinement does change results here
    p = str(tmp_path / "with_knn.idx")
    ix.save(p)
    ix2 = _native.NativeIndex.load(p).upload(0)
    nb, kd = ix2.get_knn()
    assert kd == 5 and np.array_equal(nb, ix.get_knn()[0])
    _same(ix2.batch_search(*q, 10, 3, 0.9, False, n_knn=3), before)


@pytest.mark.parametrize("cls,dim", [(seismic_amd.SeismicIndexRaw, 500), (seismic_amd.SeismicIndexRawLV, 90_000)])
def test_raw_classes_over_the_inner_format(cls, dim, tmp_path):
    """SeismicIndexRaw / SeismicIndexRawLV (reference src/pylib/mod.rs:663-1151): build from
    documents.bin, search with integer components, batch_search from queries.bin."""
    off, comps, vals = random_dataset(71, 3000, dim, nnz_lo=8, nnz_hi=100)
    dp, qp = str(tmp_path / "documents.bin"), str(tmp_path / "queries.bin")
    seismic_amd.write_inner_format(dp, off, comps, vals)
    q = random_queries(72, 25, dim, 3, 40)
    seismic_amd.write_inner_format(qp, *q)
    ix = cls.build(dp, n_postings=60 if dim == 500 else 2, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0)
    assert ix.len == 3000 and ix.nnz == int(off[-1])
    desc = ix._ix.desc
    exp = orc.batch_search(desc, *q, 10, 5, 0.8, True)
    got = ix.batch_search(qp, 10, 5, 0.8, 0, True)
    for i, row in enumerate(got):
        n = int(exp[2][i])
        assert [d for _, d in row] == exp[1][i, :n].tolist()
        assert [np.float32(s) for s, _ in row] == exp[0][i, :n].tolist()
    one = ix.search(q[1][q[0][3]:q[0][4]].astype(np.int32), q[2][q[0][3]:q[0][4]], 10, 5, 0.8, 0, True)
    assert one == got[3]
    # save / load round trip of the raw index
    ip = str(tmp_path / "raw.idx")
    ix.save(ip)
    assert cls.load(ip).batch_search(qp, 10, 5, 0.8, 0, True) == got
    with pytest.raises(IOError):
        cls.load(str(tmp_path / "missing.idx"))


def test_large_batches_are_pipelined_in_chunks(monkeypatch):
    """sgpu_batch_search cuts a large batch into up to four chunks on as many lanes (host preparation of
    chunk i+1 overlaps the kernel of chunk i): same rows, input order; errors name the query by its
    index in the caller's batch; two threads doing it at once share the four lanes without deadlock."""
    ix, dim = _index(75, 6000, 400)
    ix.upload(0)
    q = random_queries(76, 9001, dim, 3, 40)
    exp = orc.batch_search(ix.desc, *q, 10, 4, 0.9, False)[:3]
    _same(ix.batch_search(*q, 10, 4, 0.9, False), exp)
    monkeypatch.setenv("SGPU_CHUNK_MIN", "0")   # (read once per process: this only documents the knob)
    bad_c = q[1].copy()
    bad_c[int(q[0][7000])] = dim + 5
    with pytest.raises(_native.SeismicHipError) as e:
        ix.batch_search(q[0], bad_c, q[2], 10, 4, 0.9, False)
    assert "query 7000" in str(e.value)
    errors = []

    def worker():
        try:
            for _ in range(4):
                _same(ix.batch_search(*q, 10, 4, 0.9, False), exp)
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))
    th = [threading.Thread(target=worker) for _ in range(3)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errors, errors


def test_the_device_plans_a_chunk_as_the_host_does(monkeypatch):
    """r06: the launch plan of a staged chunk (processing order, block dots a query needs) is computed on the device
    (plan_kernel.hip) once a first chunk has told what the index needs. The device's order is the host's (make_plan:
    longest expected first, ties in input order) bit for bit, its maxima are the host's; chunks planned on the device, on
    the host (SGPU_DEVICE_PLAN=0) and with an LDS layout sized for LESS than a later chunk needs (lists then walked in
    groups) return the oracle's rows."""
    import ctypes
    ix, dim = _index(81, 9000, 700)
    ix.upload(0)
    q_off, qc, qv = random_queries(82, 3000, dim, 1, 60)
    L = _native.lib()
    sig = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    L.sgpu_debug_plan.argtypes = sig
    L.sgpu_debug_device_plan.argtypes = sig
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    for nq in (1, 2, 255, 1000, 3000):
        for cut in (1, 4, 16):
            ho, h3 = np.zeros(nq, np.uint32), np.zeros(3, np.uint32)
            do, d3 = np.zeros(nq, np.uint32), np.zeros(3, np.uint32)
            assert L.sgpu_debug_plan(ix.h, p(q_off), p(qc), p(qv), nq, cut, p(ho), p(h3)) == 0
            assert L.sgpu_debug_device_plan(ix.h, p(q_off), p(qc), p(qv), nq, cut, p(do), p(d3)) == 0
            assert np.array_equal(ho, do), (nq, cut)
            # (the host reports at least 1 for its two sizes, the device the plain maxima)
            assert (max(int(d3[0]), 1), int(d3[1]), max(int(d3[2]), 1)) == (int(h3[0]), int(h3[1]), int(h3[2])), (nq, cut, h3, d3)
    q = (q_off, qc, qv)
    exp = orc.batch_search(ix.desc, *q, 10, 6, 0.9, False)[:3]
    # first call: a short chunk of few, short queries seeds the cache low; the big call then meets queries that need more
    few = random_queries(83, 300, dim, 1, 2)
    _same(ix.batch_search(*few, 10, 6, 0.9, False), orc.batch_search(ix.desc, *few, 10, 6, 0.9, False)[:3])
    for _ in range(3):   # (device plans from the second chunk on; the cache grows to what the chunks report)
        _same(ix.batch_search(*q, 10, 6, 0.9, False), exp)
    # a device-planned chunk checks its components while its H2D copy is under way, before anything that reads them is
    # enqueued: the errors are the host-planned chunk's, and the lane serves the next call
    at = int(q_off[2500])
    for what, (c2, v2) in {"component >= dim": (np.where(np.arange(len(qc)) == at, dim + 9, qc).astype(qc.dtype), qv),
                           "NaN value": (qc, np.where(np.arange(len(qv)) == at, np.nan, qv).astype(qv.dtype))}.items():
        with pytest.raises(_native.SeismicHipError) as e:
            ix.batch_search(q_off, c2, v2, 10, 6, 0.9, False)
        assert "query 2500" in str(e.value) and what in str(e.value), str(e.value)
        _same(ix.batch_search(*q, 10, 6, 0.9, False), exp)
    # the first of a call's two chunks is not planned (input order): planned after all, and launches that leave workgroup
    # slots free (SGPU_GRID_SPARE, an experiment that was withdrawn) - the oracle's rows
    for env in ({"SGPU_PLAN_IDENTITY": "0"}, {"SGPU_GRID_SPARE": "8"}, {"SGPU_GRID_SPARE": "24", "SGPU_PLAN_IDENTITY": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        _same(ix.batch_search(*q, 10, 6, 0.9, False), exp)
        for k_ in env:
            monkeypatch.delenv(k_)
    monkeypatch.setenv("SGPU_DEVICE_PLAN", "0")
    _same(ix.batch_search(*q, 10, 6, 0.9, False), exp)
