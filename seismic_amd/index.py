"""Host-side mirror of the reference's Python API for the search path.

Same class names, argument names, defaults and result shapes as the PyO3 module
`seismic` (reference src/pylib/mod.rs), so code written against the reference
switches by changing the import:

    from seismic_amd import SeismicIndex, SeismicDataset, get_seismic_string

  SeismicIndex / SeismicIndexLV        string-keyed (u16 / u32 components)
      reference: src/pylib/mod.rs:46-661 (search 504-533, batch_search 587-655)
      wrapper semantics: src/inverted_index_wrapper.rs:58-91, 194-284
  SeismicIndexRaw / SeismicIndexRawLV  integer-keyed
      reference: src/pylib/mod.rs:663-1151 (search 1046-1076, batch_search 1111-1146)
  SeismicDataset / SeismicDatasetLV    in-memory dataset with exact search
      reference: src/pylib/dataset.rs, src/inverted_index_wrapper.rs:599-758

Search runs on the GPU through the C ABI (include/seismic_hip.h); everything
here is marshalling: token -> component id (unknown tokens dropped, sorted by
id), internal id -> document id string. There is no CPU search fallback.

Documented deviations (DESIGN.md "Boundary"):
  * token ids: tokens are numbered in SORTED order (the reference numbers them
    in HashMap iteration order, which changes from run to run;
    scripts/convert_json_to_inner_format.py:188-190 sorts too);
  * batch_search returns results in INPUT order (the reference's par_bridge
    does not guarantee any order, src/pylib/mod.rs:629-652);
  * num_threads is accepted and ignored on the GPU path (it is ineffective in
    the reference as well: the pool it builds is dropped, src/pylib/mod.rs:599-602);
  * the kNN graph (build_knn / nknn, n_knn at search time) is built ON THE GPU by searching every
    document through the same kernel; knn files are numpy .npy (the reference's *.knn.seismic
    uses vectorium's serializer).
"""
import ctypes as C
import gzip
import io
import json
import os
import tarfile

import numpy as np

from . import _native
from ._abi import BuildConfig

MAX_TOKEN_LEN = 30


def get_seismic_string():
    """numpy dtype string of token arrays (reference src/pylib/mod.rs:24-25, 41-44)."""
    return "U%d" % MAX_TOKEN_LEN


# ---------------------------------------------------------------------------
# ingestion (host side; reference src/json_utils.rs:10-78, wrapper 398-552)
# ---------------------------------------------------------------------------
def _iter_jsonl(path):
    if path.endswith(".tar.gz") or path.endswith(".tgz"):
        with tarfile.open(path, "r:gz") as tar:
            for member in tar:
                if not member.isfile():
                    continue
                f = tar.extractfile(member)
                for line in io.TextIOWrapper(f, encoding="utf-8"):
                    line = line.strip()
                    if line:
                        yield json.loads(line)
        return
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                yield json.loads(line)


def read_jsonl(path):
    """rows {id: str|int, vector: {token: weight}, content?: str} -> (ids, [dict], [content])"""
    ids, vecs, contents = [], [], []
    try:
        for row in _iter_jsonl(path):
            ids.append(str(row["id"]))
            vecs.append(row["vector"])
            contents.append(row.get("content"))
    except (OSError, KeyError, ValueError, tarfile.TarError) as e:
        raise IOError("Failed to read %s: %s" % (path, e))
    return ids, vecs, contents


def read_inner_format(path, comp_dtype=np.uint32):
    """Seismic's inner binary format (scripts/convert_json_to_inner_format.py:10-27):
    u32 n_vecs; per vector: u32 n, n x u32 components (sorted), n x f32 values; little endian.
    Parsed by the native library (sgpu_dataset_read): an 8.8M-document file reads at disk speed."""
    try:
        return _native.read_inner_format(path)
    except _native.SeismicHipError as e:
        raise IOError(str(e))


def write_inner_format(path, off, comps, vals):
    try:
        _native.write_inner_format(path, off, comps, vals)
    except _native.SeismicHipError as e:
        raise IOError(str(e))


def write_results_tsv(path, scores, ids, n):
    """`query_index \\t doc_id \\t rank \\t score` per result, the dump of the reference's
    perf_inverted_index (src/bin/perf_inverted_index.rs:223-235)."""
    try:
        _native.write_results_tsv(path, scores, ids, n)
    except _native.SeismicHipError as e:
        raise IOError(str(e))


def read_results_tsv(path):
    """-> {query_id: [doc_id, ...]} of a results / groundtruth TSV (same four columns)."""
    out = {}
    try:
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                if line.strip():
                    q, d = line.split("\t")[:2]
                    out.setdefault(int(q), []).append(int(d))
    except (OSError, ValueError) as e:
        raise IOError("Failed to read %s: %s" % (path, e))
    return out


def accuracy(results, groundtruth):
    """compute_accuracy of the reference's scripts/run_experiments.py:287-309: per query the size of
    the intersection of the two doc-id sets, summed, over the number of ground-truth rows."""
    total = sum(len(v) for v in groundtruth.values())
    hit = sum(len(set(v) & set(results.get(q, ()))) for q, v in groundtruth.items())
    return hit / total if total else 0.0


def _token_map(vecs, given=None):
    if given is not None:
        return dict(given)
    toks = set()
    for v in vecs:
        toks.update(v.keys())
    return {t: i for i, t in enumerate(sorted(toks))}


def _to_csr(vecs, token_map):
    off = np.zeros(len(vecs) + 1, np.uint64)
    cs, vs = [], []
    for i, v in enumerate(vecs):
        items = sorted((token_map[t], w) for t, w in v.items() if t in token_map)
        off[i + 1] = off[i] + len(items)
        cs.extend(c for c, _ in items)
        vs.extend(w for _, w in items)
    return off, np.asarray(cs, np.uint32), np.asarray(vs, np.float32)


def _resolve(tokens, values, token_map):
    """resolve_query_tokens (reference src/inverted_index_wrapper.rs:75-91): unknown tokens are
    dropped silently, the rest is sorted by component id."""
    if isinstance(tokens, np.ndarray):
        tokens = tokens.ravel().tolist()   # python strings in one C-level pass
    if isinstance(values, np.ndarray):
        values = values.ravel().tolist()
    pairs = sorted((token_map[t], float(v)) for t, v in zip(tokens, values) if t in token_map)
    # a token repeated in the query would give a duplicate component; keep the first, as a
    # dict-built query (the documented way to make one) cannot contain duplicates
    comps, vals, last = [], [], None
    for c, v in pairs:
        if c != last:
            comps.append(c)
            vals.append(v)
            last = c
    return np.asarray(comps, np.uint32), np.asarray(vals, np.float32)


def _cfg(n_postings, centroid_fraction, min_cluster_size, summary_energy, max_fraction, doc_cut, num_threads):
    # SGPU_BUILD_DEVICE=<n>: run the clustering step of the build on HIP device n (byte-identical index)
    dev = os.environ.get("SGPU_BUILD_DEVICE", "")
    return BuildConfig.defaults(n_postings=int(n_postings), centroid_fraction=float(centroid_fraction),
                                min_cluster_size=int(min_cluster_size), summary_energy=float(summary_energy),
                                max_fraction=float(max_fraction), doc_cut=int(doc_cut),
                                num_threads=int(num_threads), use_device=int(dev) + 1 if dev else 0)


def _no_knn_path(knn_path):
    if knn_path:
        raise ValueError("precomputed *.knn.seismic files use vectorium's serializer, which is not in the "
                         "reference tree; build the graph with nknn=... or load a .npy with load_knn()")


# ---------------------------------------------------------------------------
class _DatasetBase:
    """SeismicDataset (reference src/pylib/dataset.rs): add_document + exact search."""
    _CW = 2

    def __init__(self):
        self._ids, self._vecs, self._contents = [], [], []
        self._native = None
        self._tm = None

    def add_document(self, doc_id, tokens, values, content=None):
        self._ids.append(str(doc_id))
        self._vecs.append({str(t): float(v) for t, v in zip(tokens, values)})
        self._contents.append(content)
        self._native = None

    @property
    def len(self):
        return len(self._ids)

    def _freeze(self):
        if self._native is None:
            self._tm = _token_map(self._vecs)
            off, c, v = _to_csr(self._vecs, self._tm)
            # exact search only needs the forward index: build with the cheapest valid config
            self._native = _native.NativeIndex.build(self._CW, max(len(self._tm), 1), off, c, v,
                                                     _cfg(1, 1.0, 0, 1.0, 1.0, 1, 0))
        return self._native

    def search(self, query_id, query_components, query_values, k):
        """Exact top-k (brute force, host cores) -> [(query_id, score, doc_id)]."""
        ix = self._freeze()
        c, v = _resolve([str(t) for t in np.asarray(query_components).ravel()],
                        np.asarray(query_values, np.float32).ravel(), self._tm)
        sc, ids, n = ix.exact_search(np.array([0, len(c)], np.uint64), c, v, k)
        return [(str(query_id), float(sc[0, i]), self._ids[int(ids[0, i])]) for i in range(int(n[0]))]

    def batch_search(self, queries_ids, query_components, query_values, k, num_threads=0):
        return [self.search(q, c, v, k) for q, c, v in zip(np.asarray(queries_ids).ravel(), query_components,
                                                           query_values)]


class SeismicDataset(_DatasetBase):
    _CW = 2


class SeismicDatasetLV(_DatasetBase):
    _CW = 4


# ---------------------------------------------------------------------------
class _IndexBase:
    _CW = 2

    def __init__(self, native, token_map, doc_ids, contents=None, device=0, upload=True):
        self._ix = native
        self._tm = token_map
        self._doc_ids = doc_ids
        self._contents = contents
        self._doc_pos = None
        self._device = device
        self._uploaded = False
        if upload:
            self._ensure_device()

    def _ensure_device(self):
        if not self._uploaded:
            self._ix.upload(self._device)   # raises if no HIP device: no CPU fallback
            self._uploaded = True

    # ---- construction -------------------------------------------------
    @classmethod
    def build(cls, input_path, n_postings=3500, centroid_fraction=0.1, min_cluster_size=2, summary_energy=0.4,
              max_fraction=1.5, doc_cut=15, nknn=0, knn_path=None, batched_indexing=None,
              input_token_to_id_map=None, load_content=True, num_threads=0, device=0, upload=True):
        """Build from a .jsonl / .jsonl.gz / .tar.gz file (reference src/pylib/mod.rs:329-384)."""
        _no_knn_path(knn_path)
        ids, vecs, contents = read_jsonl(input_path)
        tm = _token_map(vecs, input_token_to_id_map)
        if cls._CW == 2 and len(tm) >= 2 ** 16:
            raise ValueError("The number of different tokens exceeds 2^16; use SeismicIndexLV")
        off, c, v = _to_csr(vecs, tm)
        ix = _native.NativeIndex.build(cls._CW, max(len(tm), 1), off, c, v,
                                       _cfg(n_postings, centroid_fraction, min_cluster_size, summary_energy,
                                            max_fraction, doc_cut, num_threads))
        return cls._finish_build(ix, nknn, device, upload, tm, ids, contents if load_content else None)

    @classmethod
    def _finish_build(cls, ix, nknn, device, upload, *meta):
        """The kNN graph is built on the index AS BUILT (u16/f16), before a class converts its forward index: the
        reference runs Index::from_file(..).knn(..) first and convert_dataset_into afterwards
        (src/pylib/dotvbyte.rs:193-209), so SeismicIndexDotVByte's graph is SeismicIndex's graph."""
        if nknn:
            ix.upload(device)      # Knn::new runs as batches through the GPU kernel
            ix.build_knn(nknn)
        self = cls(ix, *meta, device, False)
        if self._ix is ix and nknn:
            self._uploaded = True              # (not converted: the upload above is the index's)
        elif upload or nknn:
            self._ensure_device()
        return self

    @classmethod
    def build_from_dataset(cls, dataset, n_postings=3500, centroid_fraction=0.1, min_cluster_size=2,
                           summary_energy=0.4, max_fraction=1.5, doc_cut=15, nknn=0, knn_path=None,
                           batched_indexing=None, num_threads=0, device=0, upload=True):
        _no_knn_path(knn_path)
        tm = _token_map(dataset._vecs)
        off, c, v = _to_csr(dataset._vecs, tm)
        ix = _native.NativeIndex.build(cls._CW, max(len(tm), 1), off, c, v,
                                       _cfg(n_postings, centroid_fraction, min_cluster_size, summary_energy,
                                            max_fraction, doc_cut, num_threads))
        return cls._finish_build(ix, nknn, device, upload, tm, list(dataset._ids), list(dataset._contents))

    @classmethod
    def load(cls, index_path, device=0, upload=True):
        """Load `<index_path>.index.sgpu` + `.meta.json` (own format; reference files
        *.index.seismic use vectorium's serializer, which is not in the reference tree)."""
        base = index_path[:-len(".index.sgpu")] if index_path.endswith(".index.sgpu") else index_path
        try:
            ix = _native.NativeIndex.load(base + ".index.sgpu")
            with open(base + ".meta.json", "r", encoding="utf-8") as f:
                meta = json.load(f)
        except (_native.SeismicHipError, OSError, ValueError) as e:
            raise IOError("Failed to load index %s: %s" % (index_path, e))
        return cls(ix, meta["token_to_id_map"], meta["document_mapping"], meta.get("document_content"),
                   device, upload)

    def save(self, path):
        try:
            self._ix.save(path + ".index.sgpu")
            with open(path + ".meta.json", "w", encoding="utf-8") as f:
                json.dump({"token_to_id_map": self._tm, "document_mapping": self._doc_ids,
                           "document_content": self._contents}, f)
        except (_native.SeismicHipError, OSError) as e:
            raise IOError("Failed to save index %s: %s" % (path, e))

    # ---- accessors ----------------------------------------------------
    @property
    def dim(self):
        return int(self._ix.desc.dim)

    @property
    def len(self):
        return int(self._ix.desc.n_docs)

    @property
    def nnz(self):
        return int(self._ix.desc.nnz)

    @property
    def knn_len(self):
        return self._ix.get_knn()[1]

    def build_knn(self, nknn):
        """reference src/pylib/mod.rs:195-197 (build_knn). Runs on the GPU."""
        self._ensure_device()
        self._ix.build_knn(nknn)

    def save_knn(self, path):
        nb, dim = self._ix.get_knn()
        if dim == 0:
            raise ValueError("the index has no kNN graph")   # reference: PyValueError (src/pylib/mod.rs:217-221)
        np.save(path if path.endswith(".npy") else path + ".knn.npy", np.concatenate([[dim], nb]).astype(np.uint32))

    def load_knn(self, knn_path, nknn=None):
        try:
            a = np.load(knn_path if knn_path.endswith(".npy") else knn_path + ".knn.npy")
        except OSError as e:
            raise IOError(str(e))
        dim, nb = int(a[0]), a[1:].astype(np.uint32)
        if nknn is not None and nknn < dim and len(nb) == self.len * dim:   # keep the first nknn per document
            nb = nb.reshape(self.len, dim)[:, :nknn].reshape(-1)
            dim = nknn
        self._ix.set_knn(nb, dim)

    @property
    def is_empty(self):
        return self.len == 0

    def get(self, id):
        """Document `id` of the forward index -> (component ids, values as f32); reference
        src/pylib/mod.rs:157-165, 797-805 (`dataset().get(id)`, values widened with `to_f32`)."""
        d = self._ix.desc
        if not 0 <= id < d.n_docs:
            raise IndexError("document %d out of range (0..%d)" % (id, d.n_docs))   # the reference panics
        a, e = int(d.fwd_offsets[id]), int(d.fwd_offsets[id + 1])
        if e == a:
            return [], []
        cdt = np.uint16 if d.comp_width == 2 else np.uint32
        comps = np.ctypeslib.as_array((C.c_uint8 * ((e - a) * d.comp_width)).from_address(d.fwd_comps + a * d.comp_width))
        comps = comps.view(cdt)
        if d.value_type == 0:
            vals = np.ctypeslib.as_array((C.c_uint16 * (e - a)).from_address(d.fwd_vals + a * 2)).view(np.float16)
            vals = vals.astype(np.float32)
        else:
            codes = np.ctypeslib.as_array((C.c_uint8 * (e - a)).from_address(d.fwd_vals + a))
            vals = codes.astype(np.float32) * np.float32(d.val_scale)
        return [int(c) for c in comps], [float(v) for v in vals]

    def get_doc_ids_in_postings(self, list_id):
        d = self._ix.desc
        if not 0 <= list_id < d.dim:
            raise ValueError("Invalid list_id: %d" % list_id)
        b0, b1 = d.list_block_start[list_id], d.list_block_start[list_id + 1]
        p0, p1 = d.block_post_start[b0], d.block_post_start[b1]
        return [int(d.post_doc[p]) for p in range(p0, p1)]

    def print_space_usage_byte(self):
        d = self._ix.desc
        cw = d.comp_width
        fwd = d.nnz * (cw + 2) + (d.n_docs + 1) * 8
        packed = d.n_postings * 8
        boffs = (d.n_blocks + d.dim) * 8
        summ = d.n_entries * 3 + d.n_rows * (cw + 8) + d.n_blocks * 8
        print("Space Usage:")
        print("\tForward Index: %d Bytes" % fwd)
        print("\tPosting Lists: %d Bytes" % (packed + boffs + summ))
        print("\t  packed_postings: %d Bytes\n\t  block_offsets: %d Bytes\n\t  summaries: %d Bytes"
              % (packed, boffs, summ))
        knn = 4 * len(self._ix.get_knn()[0])
        print("\tKnn: %d Bytes" % knn)
        print("\tTotal: %d Bytes" % (fwd + packed + boffs + summ + knn))
        print("\tHBM resident: %d Bytes" % self._ix.device_bytes())

    def get_doc_text(self, doc_id):
        if self._contents is None:
            return None
        if self._doc_pos is None:   # built on first use: doc id -> position
            self._doc_pos = {d: i for i, d in enumerate(self._doc_ids)}
        i = self._doc_pos.get(doc_id)
        return None if i is None else self._contents[i]

    # ---- search -------------------------------------------------------
    def _remap(self, query_id, sc, ids, n):
        n, q, names = int(n), str(query_id), self._doc_ids
        return [(q, s, names[i]) for s, i in zip(sc[:n].tolist(), ids[:n].tolist())]

    def search(self, query_id, query_components, query_values, k, query_cut, heap_factor, n_knn=0, sorted=True):
        """-> [(query_id, score, doc_id)], best first (reference src/pylib/mod.rs:490-533)."""
        self._ensure_device()
        c, v = _resolve(np.asarray(query_components).astype(str), np.asarray(query_values, np.float32), self._tm)
        sc, ids = self._ix.search(c, v, k, query_cut, heap_factor, first_sorted=bool(sorted), n_knn=n_knn)
        return self._remap(query_id, sc, ids, len(ids))

    def batch_search(self, queries_ids, query_components, query_values, k, query_cut, heap_factor, n_knn=0,
                     sorted=True, num_threads=0):
        """-> [[(query_id, score, doc_id)]] in input order (reference src/pylib/mod.rs:572-655).
        One GPU pass over the whole batch."""
        self._ensure_device()
        qids = [str(x) for x in np.asarray(queries_ids).ravel()]
        off = np.zeros(len(qids) + 1, np.uint64)
        cs, vs = [], []
        for i, (qc, qv) in enumerate(zip(query_components, query_values)):
            c, v = _resolve(np.asarray(qc).astype(str), np.asarray(qv, np.float32), self._tm)
            cs.append(c)
            vs.append(v)
            off[i + 1] = off[i] + len(c)
        comps = np.concatenate(cs) if cs else np.zeros(0, np.uint32)
        vals = np.concatenate(vs) if vs else np.zeros(0, np.float32)
        sc, ids, n = self._ix.batch_search(off, comps, vals, k, query_cut, heap_factor, first_sorted=bool(sorted),
                                           n_knn=n_knn)
        return [self._remap(qids[i], sc[i], ids[i], n[i]) for i in range(len(qids))]


class SeismicIndex(_IndexBase):
    """u16 components (vocabulary < 65536)."""
    _CW = 2


class SeismicIndexLV(_IndexBase):
    """u32 components: large vocabularies."""
    _CW = 4


class SeismicIndexDotVByte(_IndexBase):
    """The reference's compressed index (src/pylib/dotvbyte.rs:20-36): the standard u16/f16 index is
    built first and its forward index is then converted (`convert_dataset_into`, :208-213) - here to
    SGPU_VAL_DOTVBYTE: fixed-u8 document values and a compressed component stream (12 bytes per
    8-element slice), 2.5 bytes per component in HBM instead of 4. Same query API and results type as
    SeismicIndex; scores differ by the 8-bit quantisation of the document values. u16 components only, as
    in the reference. (vectorium's DotVByteFixedU8Encoder is not in the reference tree: the fixed-point
    step and the layout of the lossless component stream are restated, see include/seismic_hip.h - parity
    unpinned; results are bit-identical to the fixed-u8 index, which the tests assert.)
    `component_stream=False` keeps the raw u16 components (SGPU_VAL_FIXEDU8, 3 bytes per component)."""
    _CW = 2

    def __init__(self, native, token_map, doc_ids, contents=None, device=0, upload=True, component_stream=True):
        from ._abi import SGPU_VAL_DOTVBYTE, SGPU_VAL_FIXEDU8
        want = SGPU_VAL_DOTVBYTE if component_stream else SGPU_VAL_FIXEDU8
        if native.desc.value_type != want:
            native = native.convert(want)
        super().__init__(native, token_map, doc_ids, contents, device, upload)


# ---------------------------------------------------------------------------
class _RawBase:
    """Integer-keyed index over the inner binary format (reference src/pylib/mod.rs:663-1151)."""
    _CW = 2

    def __init__(self, native, device=0, upload=True):
        self._ix = native
        self._device = device
        self._uploaded = False
        if upload:
            self._ensure_device()

    _ensure_device = _IndexBase._ensure_device
    dim = _IndexBase.dim
    len = _IndexBase.len
    nnz = _IndexBase.nnz
    knn_len = _IndexBase.knn_len
    is_empty = _IndexBase.is_empty
    get = _IndexBase.get
    get_doc_ids_in_postings = _IndexBase.get_doc_ids_in_postings
    print_space_usage_byte = _IndexBase.print_space_usage_byte

    @classmethod
    def build(cls, input_file, n_postings=3500, centroid_fraction=0.1, min_cluster_size=2, summary_energy=0.4,
              max_fraction=1.5, doc_cut=15, nknn=0, knn_path=None, batched_indexing=None, num_threads=0,
              device=0, upload=True):
        _no_knn_path(knn_path)
        off, c, v = read_inner_format(input_file)
        dim = int(c.max()) + 1 if len(c) else 1
        if cls._CW == 2 and dim > 65536:
            raise ValueError("component ids do not fit u16; use SeismicIndexRawLV")
        ix = _native.NativeIndex.build(cls._CW, dim, off, c, v,
                                       _cfg(n_postings, centroid_fraction, min_cluster_size, summary_energy,
                                            max_fraction, doc_cut, num_threads))
        self = cls(ix, device, upload or bool(nknn))
        if nknn:
            self.build_knn(nknn)
        return self

    build_knn = _IndexBase.build_knn
    save_knn = _IndexBase.save_knn
    load_knn = _IndexBase.load_knn

    @classmethod
    def load(cls, index_path, device=0, upload=True):
        try:
            return cls(_native.NativeIndex.load(index_path), device, upload)
        except _native.SeismicHipError as e:
            raise IOError("Failed to load index %s: %s" % (index_path, e))

    def save(self, path):
        try:
            self._ix.save(path)
        except _native.SeismicHipError as e:
            raise IOError(str(e))

    def search(self, query_components, query_values, k, query_cut, heap_factor, n_knn, sorted):
        """-> [(score, doc_id)] (reference src/pylib/mod.rs:1033-1076)."""
        self._ensure_device()
        sc, ids = self._ix.search(np.asarray(query_components).astype(np.uint32),
                                  np.asarray(query_values, np.float32), k, query_cut, heap_factor,
                                  first_sorted=bool(sorted), n_knn=n_knn)
        return [(float(s), int(i)) for s, i in zip(sc, ids)]

    def batch_search(self, query_path, k, query_cut, heap_factor, n_knn, sorted, num_threads=0):
        """queries.bin in the inner format -> [[(score, doc_id)]] in file order (src/pylib/mod.rs:1098-1146)."""
        self._ensure_device()
        off, c, v = read_inner_format(query_path)
        sc, ids, n = self._ix.batch_search(off, c, v, k, query_cut, heap_factor, first_sorted=bool(sorted),
                                           n_knn=n_knn)
        return [[(float(sc[q, i]), int(ids[q, i])) for i in range(int(n[q]))] for q in range(len(off) - 1)]


class SeismicIndexRaw(_RawBase):
    _CW = 2


class SeismicIndexRawLV(_RawBase):
    _CW = 4
