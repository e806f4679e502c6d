"""ctypes binding of libseismic_hip.so (the C ABI in include/seismic_hip.h).

The library is built in-tree by seismic_amd/csrc/Makefile. There is no Python or
CPU fallback for the search path: if the shared object is missing, or no HIP
device is usable, the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from ._abi import (ABI_VERSION, BuildConfig, IndexDesc, LaunchStats, SearchParams, SynthSpec,
                   SGPU_OK)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGPU_LIB") or os.path.join(_HERE, "libseismic_hip.so")   # SGPU_LIB: experiment builds
_lib = None


class SeismicHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("seismic_hip status %d: %s" % (status, msg))
        self.status = status


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `make -C seismic_amd/csrc` (needs hipcc); "
                "there is no CPU fallback for the search path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.sgpu_last_error.restype = C.c_char_p
        L.sgpu_build_info.restype = C.c_char_p
        L.sgpu_abi_version.restype = C.c_uint32
        L.sgpu_index_device_bytes.restype = C.c_uint64
        L.sgpu_index_device_bytes.argtypes = [C.c_void_p]
        L.sgpu_index_destroy.argtypes = [C.c_void_p]
        L.sgpu_index_destroy.restype = None
        L.sgpu_batch_destroy.argtypes = [C.c_void_p]
        L.sgpu_batch_destroy.restype = None
        vp = C.c_void_p
        L.sgpu_device_count.argtypes = [C.POINTER(C.c_int32)]
        L.sgpu_index_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(vp)]
        L.sgpu_index_build.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, vp,
                                       C.POINTER(BuildConfig), C.POINTER(vp)]
        L.sgpu_index_get_desc.argtypes = [vp, C.POINTER(IndexDesc)]
        L.sgpu_index_save.argtypes = [vp, C.c_char_p]
        L.sgpu_index_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.sgpu_index_convert.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
        L.sgpu_index_upload.argtypes = [vp, C.c_int32]
        L.sgpu_index_upload_many.argtypes = [vp, vp, C.c_uint32]
        L.sgpu_index_replicas.argtypes = [vp]
        L.sgpu_index_replicas.restype = C.c_uint32
        L.sgpu_batch_create_on.argtypes = [vp, C.c_uint32, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.sgpu_index_build_knn.argtypes = [vp, C.c_uint32]
        L.sgpu_index_set_knn.argtypes = [vp, vp, C.c_uint64, C.c_uint32]
        L.sgpu_index_get_knn.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.sgpu_search.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(SearchParams), vp, vp,
                                  C.POINTER(C.c_uint32)]
        L.sgpu_batch_search.argtypes = [vp, vp, vp, vp, C.c_uint32, C.POINTER(SearchParams), vp, vp, vp]
        L.sgpu_search_sequential.argtypes = [vp, vp, vp, vp, C.c_uint32, C.POINTER(SearchParams), vp, vp, vp,
                                             C.POINTER(C.c_double), vp]
        L.sgpu_search_sequential_timed.argtypes = [vp, vp, vp, vp, C.c_uint32, C.POINTER(SearchParams), vp, vp, vp,
                                                   C.POINTER(C.c_double), vp, vp]
        L.sgpu_batch_create.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.sgpu_batch_run.argtypes = [vp, vp, C.POINTER(SearchParams), C.c_int32, C.POINTER(LaunchStats)]
        L.sgpu_batch_run_counted.argtypes = [vp, vp, C.POINTER(SearchParams), C.POINTER(LaunchStats)]
        L.sgpu_batch_sync.argtypes = [vp, C.POINTER(LaunchStats)]
        L.sgpu_batch_fetch.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
        L.sgpu_batch_fetch_stats.argtypes = [vp, vp, vp]
        L.sgpu_summary_distances.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]
        L.sgpu_exact_search.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
        L.sgpu_synth_generate.argtypes = [C.POINTER(SynthSpec), vp, vp, vp, C.c_uint64, vp, vp, vp,
                                          C.POINTER(C.c_uint64)]
        L.sgpu_dataset_read.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp, vp, vp]
        L.sgpu_dataset_write.argtypes = [C.c_char_p, C.c_uint64, vp, vp, vp]
        L.sgpu_results_write_tsv.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp, vp, vp]
        if L.sgpu_abi_version() != ABI_VERSION:
            raise ImportError("libseismic_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(status):
    if status != SGPU_OK:
        raise SeismicHipError(status, lib().sgpu_last_error().decode("utf-8", "replace"))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build_info():
    """What the loaded library says it was built from (sgpu_build_info)."""
    return lib().sgpu_build_info().decode("utf-8", "replace")


def source_fingerprint(root=None):
    """The fingerprint sgpu_build_info() carries, recomputed from a source tree (default: the one this module lives in):
    first 64 bits of the SHA-256 of seismic_amd/csrc/{*.hip,*.cpp,*.hpp,*.inc} by name, the Makefile, include/*.h by name -
    the order of SOURCES in seismic_amd/csrc/Makefile."""
    import glob
    import hashlib
    pkg = os.path.join(root, "seismic_amd") if root else os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(pkg, "csrc")
    files = sorted((p for ext in ("hip", "cpp", "hpp", "inc") for p in glob.glob(os.path.join(csrc, "*." + ext))), key=os.path.basename)
    files += [os.path.join(csrc, "Makefile")] + sorted(glob.glob(os.path.join(os.path.dirname(pkg), "include", "*.h")), key=os.path.basename)
    h = hashlib.sha256()
    for p in files:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def device_count():
    n = C.c_int32(0)
    st = lib().sgpu_device_count(C.byref(n))
    return n.value if st == SGPU_OK else 0


def params(k, query_cut, heap_factor, first_sorted, n_knn=0):
    return SearchParams(k=int(k), query_cut=int(query_cut), heap_factor=float(heap_factor),
                        n_knn=int(n_knn), first_sorted=1 if first_sorted else 0)


def _csr(q_off, comps, vals):
    return (np.ascontiguousarray(q_off, np.uint64), np.ascontiguousarray(comps, np.uint32),
            np.ascontiguousarray(vals, np.float32))


class NativeIndex:
    """Owns an sgpu_index handle."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)
        self._desc = None

    @classmethod
    def build(cls, comp_width, dim, offsets, comps, vals, cfg=None):
        cfg = cfg or BuildConfig.defaults()
        offsets = np.ascontiguousarray(offsets, np.uint64)
        cdt = np.uint16 if comp_width == 2 else np.uint32
        comps = np.ascontiguousarray(comps, cdt)
        vals = np.ascontiguousarray(vals, np.float32)
        h = C.c_void_p()
        check(lib().sgpu_index_build(comp_width, len(offsets) - 1, dim, _p(offsets), _p(comps), _p(vals),
                                     C.byref(cfg), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_desc(cls, desc):
        h = C.c_void_p()
        check(lib().sgpu_index_create(C.byref(desc), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        check(lib().sgpu_index_load(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    def save(self, path):
        check(lib().sgpu_index_save(self.h, os.fsencode(path)))

    @property
    def desc(self):
        if self._desc is None:
            d = IndexDesc()
            check(lib().sgpu_index_get_desc(self.h, C.byref(d)))
            self._desc = d
        return self._desc

    def convert(self, value_type):
        """InvertedIndexBase::convert_dataset_into: a new index whose forward index stores the document
        values as `value_type` (0 = f16, 1 = fixed-u8, 2 = DotVByte: fixed-u8 + compressed component stream);
        lists, blocks and summaries are shared."""
        h = C.c_void_p()
        check(lib().sgpu_index_convert(self.h, int(value_type), C.byref(h)))
        return NativeIndex(h.value)

    def upload(self, device=0):
        check(lib().sgpu_index_upload(self.h, int(device)))
        return self

    def upload_many(self, devices):
        """Replicate on several devices (replica 0 from the host, the others GPU to GPU);
        batch_search then shards a batch over the replicas."""
        ids = np.ascontiguousarray(devices, np.int32)
        check(lib().sgpu_index_upload_many(self.h, _p(ids), len(ids)))
        return self

    @property
    def replicas(self):
        return int(lib().sgpu_index_replicas(self.h))

    # ---- kNN graph (reference Knn, src/inverted_index.rs:430-594) ----
    def build_knn(self, nknn):
        """Knn::new on the GPU: every document searched as a query, batched through the kernel."""
        check(lib().sgpu_index_build_knn(self.h, int(nknn)))

    def set_knn(self, neighbours, knn_dim):
        a = np.ascontiguousarray(neighbours, np.uint32)
        check(lib().sgpu_index_set_knn(self.h, _p(a), len(a), int(knn_dim)))

    def get_knn(self):
        ptr, n, dim = C.c_void_p(), C.c_uint64(0), C.c_uint32(0)
        check(lib().sgpu_index_get_knn(self.h, C.byref(ptr), C.byref(n), C.byref(dim)))
        if n.value == 0:
            return np.zeros(0, np.uint32), 0
        arr = np.ctypeslib.as_array((C.c_uint32 * n.value).from_address(ptr.value)).copy()
        return arr, int(dim.value)

    def device_bytes(self):
        return int(lib().sgpu_index_device_bytes(self.h))

    def stream_stats(self):
        """(documents, elements) a DotVByte index keeps in the raw record form; (0, 0) for the other value types."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().sgpu_index_stream_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        check(lib().sgpu_index_stream_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def close(self):
        if self.h:
            lib().sgpu_index_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- search ----
    def search(self, comps, vals, k, query_cut, heap_factor, first_sorted=False, n_knn=0):
        comps = np.ascontiguousarray(comps, np.uint32)
        vals = np.ascontiguousarray(vals, np.float32)
        sc = np.zeros(max(k, 1), np.float32)
        ids = np.zeros(max(k, 1), np.uint64)
        n = C.c_uint32(0)
        p = params(k, query_cut, heap_factor, first_sorted, n_knn)
        check(lib().sgpu_search(self.h, _p(comps), _p(vals), len(comps), C.byref(p), _p(sc), _p(ids),
                                C.byref(n)))
        return sc[: n.value].copy(), ids[: n.value].copy()

    def batch_search(self, q_off, comps, vals, k, query_cut, heap_factor, first_sorted=False, n_knn=0, out=None):
        """out: optional (scores f32 [nq, k], ids u64 [nq, k], n u32 [nq]) to receive the rows (no allocation per call)."""
        q_off, comps, vals = _csr(q_off, comps, vals)
        nq = len(q_off) - 1
        if out is not None:
            sc, ids, n = out
            assert sc.shape == (nq, k) and ids.shape == (nq, k) and len(n) >= nq
            assert sc.dtype == np.float32 and ids.dtype == np.uint64 and n.dtype == np.uint32
            assert sc.flags.c_contiguous and ids.flags.c_contiguous and n.flags.c_contiguous
        else:
            sc = np.zeros((nq, max(k, 1)), np.float32)
            ids = np.zeros((nq, max(k, 1)), np.uint64)
            n = np.zeros(max(nq, 1), np.uint32)
        p = params(k, query_cut, heap_factor, first_sorted, n_knn)
        check(lib().sgpu_batch_search(self.h, _p(q_off), _p(comps), _p(vals), nq, C.byref(p), _p(sc),
                                      _p(ids), _p(n)))
        return sc, ids, n[:nq]

    def search_sequential(self, q_off, comps, vals, k, query_cut, heap_factor, first_sorted=False, n_knn=0, per_query=False):
        """The reference's AQT loop (src/bin/perf_inverted_index.rs:184-216) natively: one sgpu_search per
        query, timed around the loop. Returns (scores, ids, n, mean microseconds per query, the 8-entry
        host-side phase breakdown in microseconds per query); with per_query=True a sixth element, the wall
        time of every call in microseconds."""
        q_off, comps, vals = _csr(q_off, comps, vals)
        nq = len(q_off) - 1
        sc = np.zeros((nq, max(k, 1)), np.float32)
        ids = np.zeros((nq, max(k, 1)), np.uint64)
        n = np.zeros(max(nq, 1), np.uint32)
        mean = C.c_double(0.0)
        phases = np.zeros(8, np.float64)
        p = params(k, query_cut, heap_factor, first_sorted, n_knn)
        if per_query:
            each = np.zeros(max(nq, 1), np.float64)
            check(lib().sgpu_search_sequential_timed(self.h, _p(q_off), _p(comps), _p(vals), nq, C.byref(p), _p(sc), _p(ids),
                                                     _p(n), C.byref(mean), _p(phases), _p(each)))
            return sc, ids, n[:nq], mean.value, phases, each[:nq]
        check(lib().sgpu_search_sequential(self.h, _p(q_off), _p(comps), _p(vals), nq, C.byref(p), _p(sc), _p(ids),
                                           _p(n), C.byref(mean), _p(phases)))
        return sc, ids, n[:nq], mean.value, phases

    def summary_distances(self, list_id, comps, vals):
        comps = np.ascontiguousarray(comps, np.uint32)
        vals = np.ascontiguousarray(vals, np.float32)
        out = np.zeros(65536, np.float32)
        n = C.c_uint32(0)
        check(lib().sgpu_summary_distances(self.h, int(list_id), _p(comps), _p(vals), len(comps), _p(out),
                                           C.byref(n)))
        return out[: n.value].copy()

    def exact_search(self, q_off, comps, vals, k, num_threads=0):
        q_off, comps, vals = _csr(q_off, comps, vals)
        nq = len(q_off) - 1
        sc = np.zeros((nq, k), np.float32)
        ids = np.zeros((nq, k), np.uint64)
        n = np.zeros(max(nq, 1), np.uint32)
        check(lib().sgpu_exact_search(self.h, _p(q_off), _p(comps), _p(vals), nq, k, num_threads, _p(sc),
                                      _p(ids), _p(n)))
        return sc, ids, n[:nq]


class DeviceBatch:
    """A query batch resident in HBM (what bench.py times)."""

    def __init__(self, index, q_off, comps, vals, k_max, replica=0):
        self.index = index
        self.q_off, self.comps, self.vals = _csr(q_off, comps, vals)
        self.nq = len(self.q_off) - 1
        self.k_max = int(k_max)
        self.h = C.c_void_p()
        check(lib().sgpu_batch_create_on(index.h, int(replica), _p(self.q_off), _p(self.comps), _p(self.vals),
                                         self.nq, self.k_max, C.byref(self.h)))

    def run(self, k, query_cut, heap_factor, first_sorted=False, sync=True, n_knn=0):
        p = params(k, query_cut, heap_factor, first_sorted, n_knn)
        st = LaunchStats()
        check(lib().sgpu_batch_run(self.index.h, self.h, C.byref(p), 1 if sync else 0, C.byref(st)))
        return st

    def run_counted(self, k, query_cut, heap_factor, first_sorted=False, n_knn=0):
        """Synchronous pass with the visited bitmap: identical results, exact work counters."""
        p = params(k, query_cut, heap_factor, first_sorted, n_knn)
        st = LaunchStats()
        check(lib().sgpu_batch_run_counted(self.index.h, self.h, C.byref(p), C.byref(st)))
        return st

    def sync(self):
        st = LaunchStats()
        check(lib().sgpu_batch_sync(self.index.h, C.byref(st)))
        return st

    def fetch(self, k):
        sc = np.zeros((self.nq, k), np.float32)
        ids = np.zeros((self.nq, k), np.uint64)
        n = np.zeros(max(self.nq, 1), np.uint32)
        check(lib().sgpu_batch_fetch(self.index.h, self.h, k, _p(sc), _p(ids), _p(n)))
        return sc, ids, n[: self.nq]

    def fetch_stats(self):
        """nq x 24 counters of the last pass (see sgpu_batch_fetch_stats)."""
        st = np.zeros((max(self.nq, 1), 24), np.uint32)
        check(lib().sgpu_batch_fetch_stats(self.index.h, self.h, _p(st)))
        return st[: self.nq]

    def algorithmic_bytes(self, k, comp_width, val_bytes=2, doc_comp_bytes=None):
        """B_q of SURVEY.md 8(d), summed over the batch, from the kernel's own work counters
        (val_bytes: 2 for f16 document values, 1 for fixed-u8; doc_comp_bytes: bytes per document component as
        stored, comp_width unless the component stream is compressed - 1.5 for DotVByte's 12-byte slices of eight components)."""
        st = self.fetch_stats().astype(np.int64)
        nnz_q = np.diff(self.q_off.astype(np.int64))
        per_elem = (comp_width if doc_comp_bytes is None else doc_comp_bytes) + val_bytes
        b = (nnz_q * (comp_width + 4) + 12 * k + 8 * st[:, 0] + 8 * st[:, 1] + 3 * st[:, 2]
             + 4 * (st[:, 4] + st[:, 3]) + 8 * st[:, 5] + st[:, 6] * per_elem)
        return int(b.sum()), st

    def close(self):
        if self.h:
            lib().sgpu_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth(n_vecs, dim, seed, kind=0, docs=None, collection=0):
    """SPLADE-shaped synthetic CSR (offsets u64, comps u32, vals f32). kind 0 docs, 1 queries; collection 0 = the
    SURVEY 8(d) law, 1 = clustered (documents around latent intents; pass the same value for documents and queries)."""
    spec = SynthSpec(n_vecs=n_vecs, dim=dim, seed=seed, kind=kind, collection=collection)
    nnz = C.c_uint64(0)
    d_off = d_c = d_v = None
    nd = 0
    if kind == 1:
        d_off, d_c, d_v = _csr(*docs)
        nd = len(d_off) - 1
    check(lib().sgpu_synth_generate(C.byref(spec), _p(d_off), _p(d_c), _p(d_v), nd, None, None, None,
                                    C.byref(nnz)))
    off = np.zeros(n_vecs + 1, np.uint64)
    comps = np.zeros(max(nnz.value, 1), np.uint32)
    vals = np.zeros(max(nnz.value, 1), np.float32)
    check(lib().sgpu_synth_generate(C.byref(spec), _p(d_off), _p(d_c), _p(d_v), nd, _p(off), _p(comps),
                                    _p(vals), C.byref(nnz)))
    return off, comps[: nnz.value], vals[: nnz.value]


def read_inner_format(path):
    """documents.bin / queries.bin of the reference (scripts/convert_json_to_inner_format.py:10-27)
    -> (offsets u64, comps u32, vals f32)."""
    n, nnz = C.c_uint64(0), C.c_uint64(0)
    check(lib().sgpu_dataset_read(os.fsencode(path), C.byref(n), C.byref(nnz), None, None, None))
    off = np.zeros(n.value + 1, np.uint64)
    comps = np.zeros(max(nnz.value, 1), np.uint32)
    vals = np.zeros(max(nnz.value, 1), np.float32)
    check(lib().sgpu_dataset_read(os.fsencode(path), C.byref(n), C.byref(nnz), _p(off), _p(comps), _p(vals)))
    return off, comps[: nnz.value], vals[: nnz.value]


def write_inner_format(path, off, comps, vals):
    off, comps, vals = _csr(off, comps, vals)
    check(lib().sgpu_dataset_write(os.fsencode(path), len(off) - 1, _p(off), _p(comps), _p(vals)))


def write_results_tsv(path, scores, ids, n):
    """query_index\tdoc_id\trank\tscore (reference src/bin/perf_inverted_index.rs:223-235)."""
    scores = np.ascontiguousarray(scores, np.float32)
    ids = np.ascontiguousarray(ids, np.uint64)
    n = np.ascontiguousarray(n, np.uint32)
    nq, k = scores.shape if scores.ndim == 2 else (0, 1)
    check(lib().sgpu_results_write_tsv(os.fsencode(path), nq, k, _p(scores), _p(ids), _p(n)))
