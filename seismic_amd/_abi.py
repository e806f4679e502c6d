"""ctypes mirror of include/seismic_hip.h (plain data layouts only)."""
import ctypes as C

SGPU_OK, SGPU_EINVAL, SGPU_EDEVICE, SGPU_ENOMEM, SGPU_EIO, SGPU_ELIMIT = range(6)
ABI_VERSION = 4
SGPU_VAL_F16, SGPU_VAL_FIXEDU8, SGPU_VAL_DOTVBYTE = 0, 1, 2

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


class IndexDesc(C.Structure):
    _fields_ = [
        ("comp_width", C.c_uint32),
        ("value_type", C.c_uint32),
        ("n_docs", C.c_uint64),
        ("dim", C.c_uint64),
        ("nnz", C.c_uint64),
        ("n_blocks", C.c_uint64),
        ("n_postings", C.c_uint64),
        ("n_rows", C.c_uint64),
        ("n_entries", C.c_uint64),
        ("fwd_offsets", u64p),
        ("fwd_comps", C.c_void_p),
        ("fwd_vals", C.c_void_p),
        ("list_block_start", u64p),
        ("block_post_start", u64p),
        ("post_doc", u32p),
        ("blk_min", f32p),
        ("blk_quant", f32p),
        ("list_row_start", u64p),
        ("row_comp", C.c_void_p),
        ("row_ptr", u64p),
        ("sum_bid", u16p),
        ("sum_code", u8p),
        ("val_scale", C.c_float),
        ("reserved", C.c_uint32),
    ]


class BuildConfig(C.Structure):
    _fields_ = [
        ("n_postings", C.c_uint64),
        ("centroid_fraction", C.c_float),
        ("min_cluster_size", C.c_uint32),
        ("summary_energy", C.c_float),
        ("max_fraction", C.c_float),
        ("doc_cut", C.c_uint32),
        ("num_threads", C.c_uint32),
        ("use_device", C.c_uint32),
        ("reserved", C.c_uint32),
    ]

    @classmethod
    def defaults(cls, **kw):
        # reference Python defaults: src/pylib/mod.rs:329
        d = dict(n_postings=3500, centroid_fraction=0.1, min_cluster_size=2,
                 summary_energy=0.4, max_fraction=1.5, doc_cut=15, num_threads=0, use_device=0)
        d.update(kw)
        return cls(**d)


class SearchParams(C.Structure):
    _fields_ = [
        ("k", C.c_uint32),
        ("query_cut", C.c_uint32),
        ("heap_factor", C.c_float),
        ("n_knn", C.c_uint32),
        ("first_sorted", C.c_int32),
    ]


class LaunchStats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_float),
        ("n_queries", C.c_uint32),
        ("grid", C.c_uint32),
        ("block", C.c_uint32),
        ("lds_bytes", C.c_uint32),
    ]


class SynthSpec(C.Structure):
    _fields_ = [
        ("n_vecs", C.c_uint64),
        ("dim", C.c_uint64),
        ("seed", C.c_uint64),
        ("kind", C.c_uint32),
        ("collection", C.c_uint32),   # 0 = SURVEY 8(d) law, 1 = clustered (synth.cpp)
    ]
