/*
 * seismic_hip.h — C ABI of the MI355X-native Seismic search hot path.
 *
 * This is the drop-in boundary. The reference (TusKANNy/seismic, Rust) has no
 * C ABI of its own; its narrowest seam for this path is the Rust method
 *
 *   InvertedIndexBase::<S>::search(&self, query: SparseVectorView<C, f32>,
 *       k, query_cut, heap_factor, n_knn, first_sorted) -> Vec<ScoredVectorDotProduct>
 *   (reference: src/inverted_index.rs:153-234)
 *
 * wrapped by SeismicIndex::search_raw/search (src/inverted_index_wrapper.rs:218-284)
 * and by the PyO3 classes (src/pylib/mod.rs:504-533, 587-655, 1046-1076, 1111-1146).
 * Every entry point below names the reference item it replaces. INTEGRATION.md
 * shows the `extern "C"` block a maintainer of the reference would add to bind it.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer it passes in
 *     or receives results in; the library keeps no pointer past the call
 *     (sgpu_index_create COPIES the descriptor's arrays).
 *   - every function returns an sgpu_status; nothing aborts or throws across
 *     the boundary (the reference panics instead: src/inverted_index.rs:172-175,
 *     src/utils.rs:23).
 *   - results are best-first; fewer than k results is not an error
 *     (src/bin/perf_inverted_index.rs:201-206); out_n[q] gives the valid count.
 *   - the search entry points run on the GPU only. There is no CPU fallback:
 *     without a usable HIP device they return SGPU_EDEVICE.
 */
#ifndef SEISMIC_HIP_H
#define SEISMIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sgpu_status {
  SGPU_OK = 0,
  SGPU_EINVAL = 1,   /* bad argument: k==0, unsorted/duplicate query components,
                        component >= dim, inconsistent descriptor */
  SGPU_EDEVICE = 2,  /* HIP failure / no device / index not uploaded */
  SGPU_ENOMEM = 3,   /* host or device allocation failed */
  SGPU_EIO = 4,      /* file could not be read / written / parsed */
  SGPU_ELIMIT = 5    /* a documented capacity limit of the kernel was exceeded */
} sgpu_status;

/* ------------------------------------------------------------------------
 * Index descriptor: the searchable state of the reference's
 * InvertedIndexBase<S> (src/inverted_index.rs:39-52) as flat host SoA arrays.
 *
 *   forward index  (reference: vectorium SparseDataset; call sites
 *                   src/posting_list.rs:203,210, src/inverted_index.rs:231)
 *     doc d = components fwd_comps[fwd_offsets[d] .. fwd_offsets[d+1]) (ascending)
 *             values     fwd_vals [same range], IEEE binary16 bit patterns
 *   posting lists  (reference: PostingList<C>, src/posting_list.rs:68-73)
 *     list c (one per component id) owns blocks
 *             [list_block_start[c], list_block_start[c+1])      (global block ids)
 *     block b owns postings [block_post_start[b], block_post_start[b+1])
 *     posting p refers to document post_doc[p]  (the reference packs
 *             (offset<<16)|len, src/posting_list.rs:32-60; offset/len are
 *             recovered here from fwd_offsets)
 *   quantized summaries (reference: QuantizedSummary<C>, src/quantized_summary.rs:15-24)
 *     blk_min[b], blk_quant[b]: dequantisation of block b's summary
 *     list c owns summary rows [list_row_start[c], list_row_start[c+1]);
 *     row r: component row_comp[r] (ascending within a list), entries
 *            [row_ptr[r], row_ptr[r+1]): sum_bid[e] = block id LOCAL to the list
 *            (ascending within a row), sum_code[e] = u8 code.
 *     (the reference stores the same CSR with Elias-Fano offsets and a packed
 *      BitField of summary ids; the codecs carry no arithmetic.)
 * ---------------------------------------------------------------------- */
/* Document value storage (the reference's value types, src/bin/build_inverted_index.rs:255-292):
 *   SGPU_VAL_F16     half::f16, the default.
 *   SGPU_VAL_FIXEDU8 unsigned 8-bit fixed point: value = code * val_scale. Replaces vectorium's
 *                    FixedU8Q / the value half of DotVByteFixedU8Encoder ("8-bit fixed-point
 *                    quantization (Q0.8)", docs/TomlInstructions.md:100). The codec is not in the
 *                    reference tree: PARITY UNPINNED. Restated as: val_scale = the smallest power of
 *                    two with 255 * val_scale >= the largest value (2^-8, i.e. Q0.8, for values
 *                    below 1); code = min(255, round_half_away(v / val_scale)), negatives -> 0.
 *                    Either component width (the reference's "fixedu8" goes with u16 and u32 components,
 *                    src/bin/perf_inverted_index.rs:110-126; its DotVByte class is u16-only).
 *   SGPU_VAL_DOTVBYTE the forward index of the reference's DotVByte index (SeismicIndexDotVByte,
 *                    src/pylib/dotvbyte.rs:15-22, 208-213; `dotvbyte` value type of the CLI,
 *                    src/bin/perf_inverted_index.rs:110-126): the FIXEDU8 values PLUS a compressed
 *                    component stream. The host-side arrays of the descriptor are those of FIXEDU8 (u16
 *                    components, u8 codes); the compression is the HBM layout: per 8-element slice three dwords
 *                    instead of four (the first component in 16 bits, three 12-bit and four 11-bit gaps), a document
 *                    with a gap that does not fit its field keeps the raw record form. The reference's codec (vectorium's DotVByte, a
 *                    variable-byte gap stream) is not in the tree: PARITY UNPINNED; it is lossless, so results
 *                    are those of the FIXEDU8 index - which is what the tests assert. u16 components only (as the
 *                    reference's class), documents of fewer than 32768 components. */
enum { SGPU_VAL_F16 = 0, SGPU_VAL_FIXEDU8 = 1, SGPU_VAL_DOTVBYTE = 2 };

typedef struct sgpu_index_desc {
  uint32_t comp_width;            /* 2 (SeismicIndex, u16) or 4 (SeismicIndexLV, u32) */
  uint32_t value_type;            /* SGPU_VAL_F16, SGPU_VAL_FIXEDU8 or SGPU_VAL_DOTVBYTE: how fwd_vals stores document values */
  uint64_t n_docs;
  uint64_t dim;                   /* number of components == number of posting lists */
  uint64_t nnz;                   /* fwd_offsets[n_docs] */
  uint64_t n_blocks;              /* list_block_start[dim] */
  uint64_t n_postings;            /* block_post_start[n_blocks] */
  uint64_t n_rows;                /* list_row_start[dim] */
  uint64_t n_entries;             /* row_ptr[n_rows] */
  const uint64_t* fwd_offsets;    /* n_docs + 1 */
  const void* fwd_comps;          /* nnz x comp_width bytes */
  const void* fwd_vals;           /* nnz binary16 bit patterns (F16) or nnz u8 codes (FIXEDU8, DOTVBYTE) */
  const uint64_t* list_block_start; /* dim + 1 */
  const uint64_t* block_post_start; /* n_blocks + 1 */
  const uint32_t* post_doc;       /* n_postings */
  const float* blk_min;           /* n_blocks */
  const float* blk_quant;         /* n_blocks */
  const uint64_t* list_row_start; /* dim + 1 */
  const void* row_comp;           /* n_rows x comp_width bytes */
  const uint64_t* row_ptr;        /* n_rows + 1 */
  const uint16_t* sum_bid;        /* n_entries */
  const uint8_t* sum_code;        /* n_entries */
  float val_scale;                /* FIXEDU8 / DOTVBYTE: value = code * val_scale (a power of two); 0 for F16 */
  uint32_t reserved;
} sgpu_index_desc;

/* Build-time configuration; mirrors Configuration (src/configurations.rs:15-129)
 * restricted to the path the Python API exposes (src/pylib/mod.rs:329-369):
 * GlobalThreshold pruning, RandomKmeansInvertedIndexApprox blocking,
 * EnergyPreserving summaries. Defaults = the reference's Python defaults. */
typedef struct sgpu_build_config {
  uint64_t n_postings;        /* 3500 */
  float centroid_fraction;    /* 0.1  */
  uint32_t min_cluster_size;  /* 2    */
  float summary_energy;       /* 0.4  */
  float max_fraction;         /* 1.5  */
  uint32_t doc_cut;           /* 15   */
  uint32_t num_threads;       /* 0 = all host cores */
  uint32_t use_device;        /* 0 = build on the host cores only; n > 0: the clustering (the k-means
                                 assignment of every posting, src/utils.rs:146-237) and the per-block
                                 summaries (src/posting_list.rs:329-368) run on HIP device n - 1. The
                                 index is byte-identical either way; a missing or failing device is an
                                 error, never a silent host build. */
  uint32_t reserved;
} sgpu_build_config;

/* Query-time knobs of InvertedIndexBase::search (src/inverted_index.rs:153-161). */
typedef struct sgpu_search_params {
  uint32_t k;            /* > 0 */
  uint32_t query_cut;    /* number of heaviest query components whose lists are walked */
  float heap_factor;     /* skip block iff heap full && dot < heap_factor * kth_best */
  uint32_t n_knn;        /* neighbours of each result to rescore (Knn::refine); needs a graph, else ignored */
  int32_t first_sorted;  /* !=0: first list visited by descending summary dot
                            (PostingList::sort_and_search, src/posting_list.rs:149-185) */
} sgpu_search_params;

typedef struct sgpu_index sgpu_index;     /* host + device state of one index */
typedef struct sgpu_batch sgpu_batch;     /* a device-resident query batch + result slab */

/* Per-launch measurements of the search kernel (HIP events on the library's stream). */
typedef struct sgpu_launch_stats {
  float kernel_ms;         /* duration of the last search kernel launch */
  uint32_t n_queries;
  uint32_t grid;           /* workgroups launched */
  uint32_t block;          /* threads per workgroup */
  uint32_t lds_bytes;      /* dynamic LDS per workgroup */
} sgpu_launch_stats;

/* ---- library ---------------------------------------------------------- */
/* Thread-local message for the last non-OK status returned on this thread. */
const char* sgpu_last_error(void);
/* ABI version of this header (bumped on any layout change). */
uint32_t sgpu_abi_version(void);
/* What this binary was built from: "sources <16 hex digits> arch gfx950 extra [<flags>] HIP version: ...". The hex
   digits are the first 64 bits of the SHA-256 of the library's source files in a fixed order (seismic_amd/csrc/Makefile,
   SOURCES); seismic_amd._native.source_fingerprint() recomputes them from a source tree. Static storage. */
const char* sgpu_build_info(void);
/* Number of visible HIP devices; SGPU_EDEVICE (and *n = 0) when none. */
sgpu_status sgpu_device_count(int32_t* n);

/* ---- index life cycle -------------------------------------------------- */
/* Replaces: holding an InvertedIndexBase<S> (src/inverted_index.rs:39-52).
 * Validates the descriptor, copies it, derives the HBM layout. */
sgpu_status sgpu_index_create(const sgpu_index_desc* desc, sgpu_index** out);
/* Replaces: InvertedIndexBase::build (src/inverted_index.rs:603-686) for the
 * Python-exposed configuration. Offline; on the host cores, or with cfg->use_device the clustering and
 * the block summaries on a HIP device (same index, byte for byte). Input = a sparse dataset in
 * CSR form, values already f32 (they are rounded to binary16 as
 * from_f32_saturating does, src/json_utils.rs:64). comps are comp_width bytes each. */
sgpu_status sgpu_index_build(uint32_t comp_width, uint64_t n_docs, uint64_t dim,
                             const uint64_t* offsets, const void* comps,
                             const float* vals, const sgpu_build_config* cfg,
                             sgpu_index** out);
/* Replaces: InvertedIndexBase::convert_dataset_into (call sites src/pylib/dotvbyte.rs:208-213,
 * src/bin/build_inverted_index.rs:294-306): a new index with the same posting lists, blocks and
 * summaries whose forward index stores the documents with another value type (F16 -> FIXEDU8 is what
 * SeismicIndexDotVByte.build does after building the standard index). *out is independent of src. */
sgpu_status sgpu_index_convert(const sgpu_index* src, uint32_t value_type, sgpu_index** out);
/* View of the index's canonical host arrays; valid until sgpu_index_destroy. */
sgpu_status sgpu_index_get_desc(const sgpu_index* idx, sgpu_index_desc* out);
/* Replaces: IndexSerializer::save_index / load_index (src/pylib/mod.rs:186-221).
 * Own flat SoA file format (the reference's wire format lives in un-vendored vectorium). */
sgpu_status sgpu_index_save(const sgpu_index* idx, const char* path);
sgpu_status sgpu_index_load(const char* path, sgpu_index** out);
/* Copies the index into the HBM of HIP device `device` (replaces any earlier upload). */
sgpu_status sgpu_index_upload(sgpu_index* idx, int32_t device);
/* Replicates the index on n devices of this process (index replicated, queries
 * sharded, no collective). Replica 0 is uploaded from the host; the others are copied from it GPU to
 * GPU (hipMemcpyPeer: xGMI on an MI355X node). With more than one replica, sgpu_batch_search shards a
 * batch contiguously over them, one host thread per device, and returns the rows in input order -
 * what the reference's rayon loop over queries does on host cores (src/pylib/mod.rs:629-652, 1129-1145).
 * Calls with fewer than two queries per replica (sgpu_search) are not cut: the replicas take them in turn.
 * A device id may be listed more than once (replicas sharing a device; used to test the path on a
 * single-GPU box). Replaces any earlier upload. */
sgpu_status sgpu_index_upload_many(sgpu_index* idx, const int32_t* device_ids, uint32_t n);
/* Number of device replicas (0 before upload). */
uint32_t sgpu_index_replicas(const sgpu_index* idx);
/* kNN graph — replaces Knn::new / Knn::refine (src/inverted_index.rs:448-500, 551-593).
 * build: every document is searched as a query (k = nknn+1, query_cut 10, heap_factor 0.7) as
 * batches through the GPU kernel; needs an uploaded index. set/get: attach or read the flattened
 * neighbour lists (document d's neighbours at [d*knn_dim, (d+1)*knn_dim)). With a graph attached,
 * sgpu_search_params.n_knn > 0 rescans the first n_knn neighbours of every top-k document after
 * the posting lists; without one n_knn is ignored, as in the reference (src/inverted_index.rs:215-216). */
sgpu_status sgpu_index_build_knn(sgpu_index* idx, uint32_t nknn);
sgpu_status sgpu_index_set_knn(sgpu_index* idx, const uint32_t* neighbours, uint64_t n_total, uint32_t knn_dim);
sgpu_status sgpu_index_get_knn(const sgpu_index* idx, const uint32_t** neighbours, uint64_t* n_total,
                               uint32_t* knn_dim);
/* Bytes resident in HBM after upload (0 before). */
uint64_t sgpu_index_device_bytes(const sgpu_index* idx);
/* A DotVByte index (SGPU_VAL_DOTVBYTE; reference src/pylib/dotvbyte.rs:15-22) keeps a document whose component gaps do
 * not fit the packed stream's fields in the raw form (2 bytes per component instead of 1.5): how many documents and
 * how many of their elements that is (0, 0 for the other value types). bench.py charges those elements their stored
 * bytes in the algorithmic-byte count. */
sgpu_status sgpu_index_stream_stats(const sgpu_index* idx, uint64_t* raw_docs, uint64_t* raw_elements);
void sgpu_index_destroy(sgpu_index* idx);

/* ---- search ------------------------------------------------------------ */
/* Replaces: InvertedIndexBase::search (src/inverted_index.rs:153-234) /
 * SeismicIndexRaw.search (src/pylib/mod.rs:1046-1076).
 * comps ascending, strictly increasing, < dim (u32 regardless of comp_width).
 * out_scores/out_doc_ids have room for params->k entries. */
sgpu_status sgpu_search(sgpu_index* idx, const uint32_t* comps, const float* vals,
                        uint32_t nnz, const sgpu_search_params* params,
                        float* out_scores, uint64_t* out_doc_ids, uint32_t* out_n);
/* Replaces: SeismicIndexRaw.batch_search (src/pylib/mod.rs:1111-1146) and the
 * per-query loop of SeismicIndex.batch_search (src/pylib/mod.rs:629-652).
 * Queries in CSR form: query q = [q_off[q], q_off[q+1]). Results in input
 * order: out_scores/out_doc_ids are nq x k (row q padded past out_n[q]).
 * Thread safety: sgpu_search / sgpu_batch_search may be called from any number of host threads on one
 * index (the reference's search takes &self and the index is Sync, src/index_traits.rs:106-113). Each
 * call borrows one of a small pool of (stream, recycled device batch) lanes of the replica it runs on,
 * so concurrent calls overlap on the device; a call allocates nothing once its lane's batch has grown
 * to the call's size. Query-time semantics that differ from a panic in the reference: query_cut == 0
 * walks no list and returns no result (k_largest_by(0), src/inverted_index.rs:187-190); any
 * heap_factor is accepted, negative ones included (the skip test is evaluated exactly as
 * src/posting_list.rs:130 does). */
sgpu_status sgpu_batch_search(sgpu_index* idx, const uint64_t* q_off,
                              const uint32_t* comps, const float* vals, uint32_t nq,
                              const sgpu_search_params* params, float* out_scores,
                              uint64_t* out_doc_ids, uint32_t* out_n);

/* Replaces: the sequential AQT loop of perf_inverted_index (src/bin/perf_inverted_index.rs:184-216), the
 * loop behind the reference's published per-query latency: the nq queries of a CSR set are searched ONE
 * AT A TIME, each through sgpu_search (host buffers in, results out, the call returns before the next
 * one starts). *mean_us (may be NULL) = wall time of the loop / nq. breakdown_us (may be NULL) receives 8
 * doubles, the mean microseconds per query the calling thread spent in each host-side phase:
 *   [0] validation + launch plan  [1] staging into the pinned arena  [2] enqueue H2D
 *   [3] launch configuration + kernel launch  [4] enqueue D2H  [5] waiting for the stream (the kernel
 *   runs here)  [6] copying the rows out  [7] reserved. */
sgpu_status sgpu_search_sequential(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps,
                                   const float* vals, uint32_t nq, const sgpu_search_params* params,
                                   float* out_scores, uint64_t* out_doc_ids, uint32_t* out_n,
                                   double* mean_us, double* breakdown_us);
/* Same loop; per_query_us (may be NULL) additionally receives the wall time of every call, nq doubles - the
 * distribution (p50 / p95 / p99 / max) behind the mean that perf_inverted_index reports
 * (src/bin/perf_inverted_index.rs:184-216 times the whole loop only). */
sgpu_status sgpu_search_sequential_timed(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps,
                                         const float* vals, uint32_t nq, const sgpu_search_params* params,
                                         float* out_scores, uint64_t* out_doc_ids, uint32_t* out_n,
                                         double* mean_us, double* breakdown_us, double* per_query_us);

/* Device-resident variant (what bench.py times: inputs already in HBM when the
 * timed region starts; results stay in HBM until fetched). */
sgpu_status sgpu_batch_create(sgpu_index* idx, const uint64_t* q_off,
                              const uint32_t* comps, const float* vals, uint32_t nq,
                              uint32_t k_max, sgpu_batch** out);
/* Same, on replica `replica` of an index uploaded with sgpu_index_upload_many (run / fetch / stats
 * find the replica from the batch). */
sgpu_status sgpu_batch_create_on(sgpu_index* idx, uint32_t replica, const uint64_t* q_off,
                                 const uint32_t* comps, const float* vals, uint32_t nq,
                                 uint32_t k_max, sgpu_batch** out);
/* Enqueues one search pass over the batch on the library's stream. With
 * sync != 0 waits for it and fills *stats (may be NULL). */
sgpu_status sgpu_batch_run(sgpu_index* idx, sgpu_batch* batch,
                           const sgpu_search_params* params, int32_t sync,
                           sgpu_launch_stats* stats);
/* Same search, synchronous, but with the reference's visited set materialised as a bitmap in HBM
 * instead of the (exactly equivalent, cheaper) heap-membership test: results are identical, and
 * work counters [5],[6] of sgpu_batch_fetch_stats then exclude re-encountered documents exactly
 * as the reference's FxHashSet does (src/inverted_index.rs:181-184). Used for accounting. */
sgpu_status sgpu_batch_run_counted(sgpu_index* idx, sgpu_batch* batch,
                                   const sgpu_search_params* params, sgpu_launch_stats* stats);
/* Blocks until all enqueued passes (of every replica) are done; *stats (may be NULL) gets the MEAN
 * kernel duration of replica 0's passes enqueued since the previous sync. */
sgpu_status sgpu_batch_sync(sgpu_index* idx, sgpu_launch_stats* stats);
sgpu_status sgpu_batch_fetch(sgpu_index* idx, sgpu_batch* batch, uint32_t k,
                             float* out_scores, uint64_t* out_doc_ids, uint32_t* out_n);
/* Work counters of the batch's LAST pass, nq x 24 uint32 per query:
 *   [0] blocks of the walked lists   [1] summary rows matched   [2] summary entries read
 *   [3] blocks that passed the skip test   [4] postings of those blocks
 *   [5] documents scored (as the reference would; exact after sgpu_batch_run_counted, otherwise
 *       re-encountered documents are included)   [6] sum of their component counts
 *   [7] documents the kernel scored speculatively (>= [5]; the surplus is overhead)
 *   [8..19] kernel phase clocks (shader cycles / 16; zero unless the library was built with
 *           -DSGPU_PROF, `make prof`), [20] workgroup slot, [21..23] reserved
 * Counters [3..6] are exact only after sgpu_batch_run_counted (the default pass skips replay windows
 * that cannot change the heap and does not track re-encountered documents).
 * Counters [0..6] are what the ALGORITHM touches and feed the roofline accounting. */
sgpu_status sgpu_batch_fetch_stats(sgpu_index* idx, sgpu_batch* batch, uint32_t* out_counters);
void sgpu_batch_destroy(sgpu_batch* batch);

/* Hot loop A in isolation — replaces QuantizedSummary::distances
 * (src/quantized_summary.rs:64-160) for posting list `list`: writes one f32
 * per block of that list (out_dots has room for n_blocks(list) entries). */
sgpu_status sgpu_summary_distances(sgpu_index* idx, uint32_t list, const uint32_t* comps,
                                   const float* vals, uint32_t nnz, float* out_dots,
                                   uint32_t* out_n_blocks);

/* ---- host-side helpers around the path (CPU, not on the hot path) ------ */
/* Replaces: SeismicDataset.search (exact, vectorium FlatIndex;
 * src/inverted_index_wrapper.rs:721-742) — ground truth for recall@k.
 * Brute force over the forward index of `idx`, multi-threaded on the host. */
sgpu_status sgpu_exact_search(const sgpu_index* idx, const uint64_t* q_off,
                              const uint32_t* comps, const float* vals, uint32_t nq,
                              uint32_t k, uint32_t num_threads, float* out_scores,
                              uint64_t* out_doc_ids, uint32_t* out_n);
/* Seismic's inner binary dataset format (documents.bin / queries.bin: written by the reference's
 * scripts/convert_json_to_inner_format.py:10-27, read by vectorium's read_seismic_format at
 * src/pylib/mod.rs:987,1127): u32 n_vecs; per vector u32 n, n x u32 components, n x f32 values.
 * read: call with offsets == NULL to size (*n_vecs, *nnz receive the counts), then with buffers of
 * n_vecs + 1 offsets and nnz components / values (*n_vecs, *nnz carry the capacities in). */
sgpu_status sgpu_dataset_read(const char* path, uint64_t* n_vecs, uint64_t* nnz, uint64_t* offsets,
                              uint32_t* comps, float* vals);
sgpu_status sgpu_dataset_write(const char* path, uint64_t n_vecs, const uint64_t* offsets,
                               const uint32_t* comps, const float* vals);
/* Replaces: the result dump of perf_inverted_index (src/bin/perf_inverted_index.rs:223-235), the file
 * scripts/run_experiments.py:287-309 compares with groundtruth.tsv:
 * query_index \t doc_id \t rank (from 1) \t score, one line per result (rows of nq x k slabs). */
sgpu_status sgpu_results_write_tsv(const char* path, uint32_t nq, uint32_t k, const float* scores,
                                   const uint64_t* doc_ids, const uint32_t* n);
/* Deterministic SPLADE-shaped synthetic data: writes a
 * CSR dataset into caller-provided buffers. Call with comps==NULL to size:
 * *out_nnz receives the number of entries. kind: 0 = documents, 1 = queries
 * (queries draw 60% of their tokens from a random source document of `docs_*`). */
typedef struct sgpu_synth_spec {
  uint64_t n_vecs;
  uint64_t dim;
  uint64_t seed;
  uint32_t kind;        /* 0 docs, 1 queries */
  uint32_t collection;  /* 0 = the SURVEY 8(d) law (the benchmark's headline collection); 1 = "clustered": documents drawn
                           around latent intents, queries carry their source document's weights (synth.cpp) */
} sgpu_synth_spec;
sgpu_status sgpu_synth_generate(const sgpu_synth_spec* spec,
                                const uint64_t* docs_offsets, const uint32_t* docs_comps,
                                const float* docs_vals, uint64_t n_docs,
                                uint64_t* out_offsets, uint32_t* out_comps, float* out_vals,
                                uint64_t* out_nnz);

#ifdef __cplusplus
}
#endif
#endif /* SEISMIC_HIP_H */
