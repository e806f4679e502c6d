"""Test-side wrapper of the CPU oracle (oracle/liborc.so). TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from seismic_amd._abi import BuildConfig, IndexDesc, SearchParams

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

ORDER_LANES16, ORDER_SEQ = 0, 1


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "algo_bytes", "blocks_total", "blocks_scored", "docs_scored",
        "postings_seen", "summary_entries", "lists_walked")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, "oracle", "liborc.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
        L = C.CDLL(path)
        L.orc_index_build.restype = C.c_void_p
        L.orc_index_build.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(BuildConfig)]
        L.orc_index_desc.argtypes = [C.c_void_p, C.POINTER(IndexDesc)]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_index_convert_fixedu8.restype = C.c_void_p
        L.orc_index_convert_fixedu8.argtypes = [C.c_void_p]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_score_doc.restype = C.c_float
        L.orc_score_doc.argtypes = [C.POINTER(IndexDesc), C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_uint32, C.c_int]
        assert L.orc_stats_size() == C.sizeof(Stats)
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def csr(vectors, dtype_c=np.uint32):
    """list of (components, values) -> (offsets u64, comps u32, vals f32)"""
    off = np.zeros(len(vectors) + 1, dtype=np.uint64)
    for i, (c, _) in enumerate(vectors):
        off[i + 1] = off[i] + len(c)
    comps = np.concatenate([np.asarray(c, dtype=np.uint32) for c, _ in vectors]) if vectors else np.zeros(0, np.uint32)
    vals = np.concatenate([np.asarray(v, dtype=np.float32) for _, v in vectors]) if vectors else np.zeros(0, np.float32)
    return off, np.ascontiguousarray(comps, dtype=np.uint32), np.ascontiguousarray(vals, dtype=np.float32)


class OracleIndex:
    """Index built by the oracle's reference-following builder."""

    def __init__(self, comp_width, dim, offsets, comps, vals, cfg=None):
        self.cfg = cfg or BuildConfig.defaults()
        self._keep = (np.ascontiguousarray(offsets, np.uint64), np.ascontiguousarray(comps, np.uint32),
                      np.ascontiguousarray(vals, np.float32))
        self.h = lib().orc_index_build(comp_width, len(offsets) - 1, dim, _p(self._keep[0]),
                                       _p(self._keep[1]), _p(self._keep[2]), C.byref(self.cfg))
        self.desc = IndexDesc()
        lib().orc_index_desc(self.h, C.byref(self.desc))

    def convert_fixedu8(self):
        """convert_dataset_into (reference src/pylib/dotvbyte.rs:208-213) restated: fixed-u8 forward values."""
        o = object.__new__(OracleIndex)
        o.cfg, o._keep = self.cfg, self._keep
        o.h = lib().orc_index_convert_fixedu8(self.h)
        o.desc = IndexDesc()
        lib().orc_index_desc(o.h, C.byref(o.desc))
        return o

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_free(self.h)
            self.h = None


def desc_arrays(desc):
    """numpy views of every array of an IndexDesc (no copy)."""
    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dt)
        addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
        return np.ctypeslib.as_array((C.c_char * (n * np.dtype(dt).itemsize)).from_address(addr)).view(dt)
    cw = np.uint16 if desc.comp_width == 2 else np.uint32
    return dict(
        fwd_offsets=arr(desc.fwd_offsets, desc.n_docs + 1, np.uint64),
        fwd_comps=arr(desc.fwd_comps, desc.nnz, cw),
        fwd_vals=arr(desc.fwd_vals, desc.nnz, np.uint16 if desc.value_type == 0 else np.uint8),
        list_block_start=arr(desc.list_block_start, desc.dim + 1, np.uint64),
        block_post_start=arr(desc.block_post_start, desc.n_blocks + 1, np.uint64),
        post_doc=arr(desc.post_doc, desc.n_postings, np.uint32),
        blk_min=arr(desc.blk_min, desc.n_blocks, np.float32),
        blk_quant=arr(desc.blk_quant, desc.n_blocks, np.float32),
        list_row_start=arr(desc.list_row_start, desc.dim + 1, np.uint64),
        row_comp=arr(desc.row_comp, desc.n_rows, cw),
        row_ptr=arr(desc.row_ptr, desc.n_rows + 1, np.uint64),
        sum_bid=arr(desc.sum_bid, desc.n_entries, np.uint16),
        sum_code=arr(desc.sum_code, desc.n_entries, np.uint8),
    )


def knn_build(desc, nknn):
    """Knn::new restated: flattened neighbour ids (documents with < nknn results contribute fewer)."""
    out = np.zeros(max(int(desc.n_docs) * nknn, 1), np.uint32)
    lib().orc_knn_build.restype = C.c_uint64
    n = lib().orc_knn_build(C.byref(desc), C.c_uint32(nknn), _p(out))
    return out[: int(n)].copy()


_KNN_KEEP = [None]


def knn_attach(neighbours, dim):
    """Graph used by search()/batch_search() when n_knn > 0 (None detaches)."""
    if neighbours is None:
        _KNN_KEEP[0] = None
        lib().orc_knn_attach(None, C.c_uint64(0), C.c_uint32(0))
        return
    a = np.ascontiguousarray(neighbours, np.uint32)
    _KNN_KEEP[0] = a
    lib().orc_knn_attach(_p(a), C.c_uint64(len(a)), C.c_uint32(dim))


def params(k, query_cut, heap_factor, first_sorted=False, n_knn=0):
    return SearchParams(k=k, query_cut=query_cut, heap_factor=heap_factor, n_knn=n_knn,
                        first_sorted=1 if first_sorted else 0)


def search(desc, comps, vals, k, query_cut, heap_factor, first_sorted=False, order=ORDER_LANES16,
           want_stats=False, n_knn=0):
    comps = np.ascontiguousarray(comps, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    sc = np.zeros(k, np.float32)
    ids = np.zeros(k, np.uint64)
    n = C.c_uint32(0)
    st = Stats()
    p = params(k, query_cut, heap_factor, first_sorted, n_knn)
    rc = lib().orc_search(C.byref(desc), _p(comps), _p(vals), len(comps), C.byref(p), order, _p(sc),
                          _p(ids), C.byref(n), C.byref(st))
    if rc:
        raise ValueError("oracle rejected the query (rc=%d)" % rc)
    out = (sc[: n.value].copy(), ids[: n.value].copy())
    return out + (st.as_dict(),) if want_stats else out


def batch_search(desc, q_off, comps, vals, k, query_cut, heap_factor, first_sorted=False,
                 order=ORDER_LANES16, num_threads=0, n_knn=0, tuned=False):
    """tuned=True: the AVX2/F16C + hash-set variant of the same algorithm (bit-identical results; what
    bench.py times as cpu_baseline)."""
    q_off = np.ascontiguousarray(q_off, np.uint64)
    comps = np.ascontiguousarray(comps, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    nq = len(q_off) - 1
    sc = np.zeros((nq, k), np.float32)
    ids = np.zeros((nq, k), np.uint64)
    n = np.zeros(nq, np.uint32)
    st = Stats()
    secs = C.c_double(0)
    used = C.c_uint32(0)
    p = params(k, query_cut, heap_factor, first_sorted, n_knn)
    fn = lib().orc_batch_search_tuned if tuned else lib().orc_batch_search
    rc = fn(C.byref(desc), _p(q_off), _p(comps), _p(vals), nq, C.byref(p), order,
            num_threads, _p(sc), _p(ids), _p(n), C.byref(st), C.byref(secs), C.byref(used))
    if rc:
        raise ValueError("oracle rejected the batch (rc=%d)" % rc)
    return sc, ids, n, st.as_dict(), secs.value, used.value


def summary_distances(desc, list_id, comps, vals):
    comps = np.ascontiguousarray(comps, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    a = desc_arrays(desc)
    nb = int(a["list_block_start"][list_id + 1] - a["list_block_start"][list_id])
    out = np.zeros(max(nb, 1), np.float32)
    n = C.c_uint32(0)
    rc = lib().orc_summary_distances(C.byref(desc), list_id, _p(comps), _p(vals), len(comps), _p(out),
                                     C.byref(n))
    assert rc == 0
    return out[: n.value].copy()


def exact_search(desc, comps, vals, k, order=ORDER_LANES16):
    comps = np.ascontiguousarray(comps, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    sc = np.zeros(k, np.float32)
    ids = np.zeros(k, np.uint64)
    n = C.c_uint32(0)
    rc = lib().orc_exact_search(C.byref(desc), _p(comps), _p(vals), len(comps), k, order, _p(sc), _p(ids),
                                C.byref(n))
    assert rc == 0
    return sc[: n.value].copy(), ids[: n.value].copy()


def quantize(values):
    v = np.ascontiguousarray(values, np.float32)
    codes = np.zeros(len(v), np.uint8)
    mn, qt = C.c_float(0), C.c_float(0)
    lib().orc_quantize(_p(v), len(v), C.byref(mn), C.byref(qt), _p(codes))
    return mn.value, qt.value, codes


def interleave_index(desc):
    """Spread the index's pages over the host's NUMA nodes; number of nodes used (0 = one node, or refused)."""
    lib().orc_interleave_index.restype = C.c_int
    lib().orc_interleave_index.argtypes = [C.POINTER(IndexDesc)]
    return int(lib().orc_interleave_index(C.byref(desc)))


def pin_plan(nt):
    """(physical cores seen, the CPU each of nt pinned threads of the tuned batch search gets)."""
    cpus = np.zeros(nt, np.int32)
    lib().orc_pin_plan.restype = C.c_int
    lib().orc_pin_plan.argtypes = [C.c_uint32, C.c_void_p]
    return int(lib().orc_pin_plan(nt, _p(cpus))), cpus


def score_doc_tuned(desc, doc, comps, vals):
    comps = np.ascontiguousarray(comps, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    lib().orc_score_doc_tuned.restype = C.c_float
    lib().orc_score_doc_tuned.argtypes = [C.POINTER(IndexDesc), C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    return float(lib().orc_score_doc_tuned(C.byref(desc), doc, _p(comps), _p(vals), len(comps)))


def score_doc(desc, doc, comps, vals, order=ORDER_LANES16):
    comps = np.ascontiguousarray(comps, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    return float(lib().orc_score_doc(C.byref(desc), doc, _p(comps), _p(vals), len(comps), order))
