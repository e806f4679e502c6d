// device_types.hpp — plain structs shared by the kernel and its host-side launcher.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sgpu {

#if defined(__HIPCC__)
#define SGPU_HD_EARLY __host__ __device__
#else
#define SGPU_HD_EARLY
#endif

// The index as it lies in HBM (SoA; see DESIGN.md "Data layout in HBM").
struct DevView {
  const uint8_t* fwd;                 // document records, 16-byte aligned:
                                      //   [npad components][npad binary16 values], npad = len rounded up to 8
  const uint32_t* list_block_start;   // dim + 1
  const uint32_t* block_post_start;   // n_blocks + 1
  const uint64_t* post_ref;           // n_postings: (record offset / 16) << 16 | len
                                      //   (the reference's PackedPostingBlock, src/posting_list.rs:32-60)
  const uint32_t* post_doc;           // n_postings: document id (visited set key, result id)
  const uint32_t* list_row_start;     // dim + 1
  const void* row_comp;               // n_rows, ascending within a list
  const uint64_t* row_ptr;            // n_rows + 1 (entries can exceed 2^32 on large indexes)
  const uint16_t* sum_bid;            // n_entries: list-local block id, ascending within a row
  const uint16_t* row_mid;            // n_rows: entries of the row whose block id is below half the list's blocks
                                      //   (stage 1 splits every list between two wavefronts by block id)
  const float* sum_deq;               // n_entries: code*quant + min of the entry's block, rounded as the
                                      //   reference does (src/quantized_summary.rs:102-104), precomputed at upload
  // kNN graph (reference: Knn, src/inverted_index.rs:430-594): flattened neighbour lists, and the
  // record ref of every document (the refine step scores neighbours that come without a posting)
  const uint32_t* knn;                // knn_total ids, document d's neighbours at [d*knn_dim, +knn_dim); null = no graph
  const uint64_t* doc_ref;            // n_docs: same packing as post_ref
  uint64_t knn_total;
  uint32_t knn_dim;
  uint32_t dim, n_docs, n_bitmap_words;
#if defined(SGPU_LAZY_DOCS) && SGPU_LAZY_DOCS
  uint32_t n_postings_lt_2g;          // posting indices fit 31 bits (lazy document ids keep one in it_doc next to the visited bit)
#endif
  // Hashed row directory (r05; u16 components): (posting list, component) -> the list's summary row of that component,
  // so that stage 1 finds a query component's row in ONE trip to HBM instead of the ~12 dependent probes of a binary
  // search over row_comp. Buckets of four 16-byte entries {key = list << 16 | component, first entry (low 32 bits),
  // first entry (high 16) | entries << 16, split point}; a key lives in the first bucket from row_dir_bucket(key) on
  // that had a free slot (empty key 0xffffffff); the table is at most 60 % full (8.8M-document index: 141 M rows,
  // 3.8 GB). Every u16 index has one (its kernels carry no search and row_comp / list_row_start / row_ptr / row_mid
  // stay null); u32 indexes have none: binary search over those arrays.
  const uint4* row_dir;
  uint32_t row_dir_buckets;
};
enum : uint32_t { kRowDirEmpty = 0xffffffffu };
SGPU_HD_EARLY inline uint32_t row_dir_bucket(uint32_t key, uint32_t n_buckets) {   // hash, then multiply-high onto [0, n_buckets)
  return (uint32_t)(((uint64_t)(key * 2654435761u) * n_buckets) >> 32);
}

struct BatchView {
  const uint32_t* q_off;   // nq + 1
  const uint32_t* q_comp;
  const float* q_val;
  uint32_t nq;
  uint32_t k_stride;       // row stride of the result slabs
  float* out_scores;
  uint64_t* out_ids;
  uint32_t* out_n;
  const uint32_t* q_order; // optional processing order (queue ticket -> query), longest expected first
  const uint32_t* q_seed;  // LK_HASH: per query, the seed of its collision-free hash multiplier
  uint32_t* out_stats;     // optional nq x STATS_WORDS counters (zeroed by the host before a pass)
  uint32_t* status;        // optional launch status word (zero before the launch; the cooperative variant stores a
                           //   protocol-error code here - it comes back to the host with the result rows)
  uint32_t* done;          // optional (cooperative launches whose rows go straight to the pinned host arena): the workgroup
  uint32_t done_seq;       //   that finishes the LAST query stores done_seq here, at system scope, after every row has
                           //   left its writer - the host may hand the rows out while the launch winds down
};

// LaunchArgs::value_type is SGPU_VAL_* (seismic_hip.h: 0 f16, 1 fixed-u8, 2 DotVByte) or this internal layout of an f16
// index: binary16 values behind the compressed component stream (search_kernel.inc VT_F16S; chosen at upload)
enum { kDevValF16Sliced = 3 };
enum { MODE_SEARCH = 0, MODE_DOTS = 1, MODE_COUNTED = 2 };   // COUNTED: search with the visited bitmap (exact counters)
// query lookup table in LDS: {32 bits, rank} per 32 ids | one byte per id | bits + 16-bit ranks | hashed {id, weight} entries
enum { LK_PACKED = 0, LK_DENSE = 1, LK_SPLIT = 2, LK_HASH = 3 };
// LK_HASH: a table of kHashSlots 8-byte entries {component id, weight bits} addressed by a multiplicative
// hash of the component id; an empty slot holds the id 0xffffffff. A document component costs ONE LDS read
// (ds_read_b64) and a compare. Used for u32 components (large vocabularies) and, for u16 components, where
// the dense byte table is not available (measurements: profiles/r03_lds_sensitivity.md). The host picks, per query, a multiplier of this family under which the
// query's components fall into distinct slots (make_plan); a batch with a query for which none of the
// kHashSeeds works uses another layout. Component ids below 2^24 (v_mul_u32_u24).
enum { kHashBits = 12, kHashSlots = 1 << kHashBits, kHashSeeds = 64 };
#if defined(__HIPCC__)
#define SGPU_HD __host__ __device__
#else
#define SGPU_HD
#endif
SGPU_HD inline uint32_t hash_mult(uint32_t seed) {   // odd 24-bit multipliers
  return (((seed + 1u) * 2654435761u) >> 8 | 1u) & 0xffffffu;
}
SGPU_HD inline uint32_t hash_slot(uint32_t c, uint32_t mult) {   // top bits of the low 32 bits of the 24 x 24-bit product
  return (uint32_t)(((uint64_t)(c & 0xffffffu) * mult) & 0xffffffffull) >> (32 - kHashBits);
}
enum { STATS_WORDS = 24 };
enum { kStateWords = 160 };   // per-workgroup state words in LDS (ST_* in search_kernel.inc)   // per-query stats: 8 work counters + 12 phase clocks (>>4) + slot + pad

// Cooperative mode (small launches, launch tails): workgroups that find the query queue empty do not
// exit but HELP the workgroups that still own a query. An owner whose heap is full publishes one WIDE
// ROUND - every remaining block of its list group in traversal order, as {summary dot, global block id}
// records - on a board in HBM; helpers (and the owner itself) claim chunks of positions, apply the skip
// test against the round's starting threshold, score the documents of the surviving blocks and append the
// items that beat that threshold to the owner's candidate list; the owner then replays the candidates in
// traversal order with the reference's sequential rules (live threshold, block decided at its first
// item, heap membership for re-encountered documents). Scores do not depend on who computes them and a
// staler threshold only admits more blocks, so the result is the reference's, bit for bit.
// All shared words are written and read with agent-scope (sc1) accesses; see search_kernel.inc.
struct CoopView {
  uint64_t* open;       // [8] bit s: slot s has an open round with unclaimed positions (one 64-byte line)
  uint32_t* counters;   // [0] idle workgroups [1] finished queries [2] exited workgroups [3] sticky protocol-error code
                        //   (1: an owner's wait for its chunks gave up, 2: a helper's wait for work gave up, 3: control
                        //   words of another round than the claimed one); the host reads it after every cooperative launch
  uint64_t* slots;      // per owner slot (blockIdx.x), kCoopSlotWords u64:
                        //   [0] claim = next chunk:20 | positions of the round:22 | positions per chunk:10 | round:10
                        //       (search_kernel.inc: co_claim_word; claimed with fetch_add(1) - round-independent)
                        //   [1] done:32 (positions reported)   [2] n_cand:32   [3] query:32 | thr0:32   [4] cut:32 | round:32
                        // Invariant between launches: open[] and counters[0..2] are zero (the last workgroup out
                        // re-zeroes them and the slot words). Nothing else of the board needs to be: a slot's words
                        // are rewritten in full BEFORE its open bit is set, and pos_pub / cands are only ever read
                        // below the counts those words publish - so stale regions left by a launch with another
                        // grid or max_pos are never interpreted. Keep it that way: no reader may touch a slot's
                        // words before it has seen the slot's open bit (or holds a valid claim on it).
  uint64_t* pos_pub;    // [slots x max_pos] {dot bits : 32 | global block id : 32}, traversal order
  uint64_t* cands;      // [slots x max_cand x 2] {score bits:32 | key:32 (pos << 16 | posting in block)}, {bdot bits:32 | doc:32}
  uint32_t max_pos;     // positions per slot (<= 65535)
  uint32_t max_cand;    // candidates per round the owner can sort in LDS; more = the owner falls back to local rounds
  uint32_t chunk;       // most positions per claim (sizes the helpers' LDS tables; <= the workgroup size)
  uint32_t chunk_min;   // fewest positions per claim (the owner picks the round's figure in between)
  uint32_t idle_ratio;  // an owner goes wide only while idle workgroups >= idle_ratio x workgroups still owning a query
  uint32_t min_items;   // local items replayed before a query may go wide (a useful threshold first)
  uint32_t first_reach; // positions of a query's first wide round (the following one takes the rest)
  uint32_t idle_min;    // idle workgroups needed before an owner goes wide (0 = always, used by the tests)
  uint32_t poll_sleep;  // idle workgroups look at the board every poll_sleep x ~0.45 us
  uint32_t enabled;
  uint64_t* trace;      // [slots x 16] event times (trace builds: -DSGPU_COOP_TRACE), else null
};
enum { kCoopSlotWords = 8 };

struct KParams {
  uint32_t k, query_cut;
  float heap_factor;
  int32_t first_sorted;
  uint32_t mode;
  uint32_t items_max;    // speculative documents per round (LDS item table)
  uint32_t items_init;   // first round's budget; adapts to the replay's keep ratio
  uint32_t items_min;    // lower bound of the budget
  uint32_t rblocks_max;  // blocks filtered per thread and round
  uint32_t n_knn;        // neighbours of each result to rescore after the lists (Knn::refine); 0 = off
  uint32_t target_list;  // MODE_DOTS: the posting list whose summary dots are wanted
  float val_scale;       // fixed-u8 document values: value = code * val_scale (a power of two)
  uint32_t queue_base;   // value of the work counter when the launch starts (0 when the counter is zeroed per launch;
                         // latency-bound calls let it run on: a launch of nq queries on a grid of G workgroups takes
                         // exactly nq + G tickets, so the host knows where the next launch begins - no reset to enqueue)
  uint32_t ring;         // streamed stage 2 (r06; search_kernel.inc "stage 2 as a stream"): item slots of the ring in
                         // LDS, a power of two >= 256 (stream_ring_bytes of them fit the union region); 0 = the round loop
};

// Streamed stage 2 (r06): the union region then holds a RING of item slots {record ref / score 8 B, document 4 B, dot
// index 2 B}, the two length-class queues of slot numbers (2 x 2 B per slot), one counter per 64-slot window and the
// feeder's block tables (kStreamFeedPos positions x {inclusive item count 4 B, first posting 4 B, dot index 2 B}).
#ifndef SGPU_STREAM_FEED_POS
#define SGPU_STREAM_FEED_POS 192
#endif
enum { kStreamFeedPos = SGPU_STREAM_FEED_POS };   // positions per step of the feeder (a multiple of 64)
SGPU_HD_EARLY inline uint32_t stream_ring_bytes(uint32_t ring) { return ring * 18u + ring / 16u + (uint32_t)kStreamFeedPos * 10u; }

// The scoring loop's weights (q_sc, preceded by the 0.0 slot) sit at a FIXED offset of the dynamic LDS, so that the
// scaled-byte dense lookup (search_kernel.inc: kScaledMaxNnz) can address them as `byte + constant`.
enum { kQscOffset = 0 };
struct LdsLayout {   // byte offsets into dynamic LDS, all multiples of 16
  uint32_t q_comp, q_val, q_sc, q_bits, q_rank, sel, rt_start, rt_mid, rt_pre, dots, order, uni, part, heap, st;
  uint32_t qc, qn;   // capacities: lists per query, components per query
  uint32_t qg;       // lists per GROUP (<= qc): the row tables of stage 1 (rt_*) hold this many lists
  uint32_t dots_cap; // block dots that fit the dots area: a query's lists are processed in groups of at most this many blocks
  uint32_t total;
};

struct LaunchArgs {
  DevView ix;
  BatchView qb;
  KParams p;
  LdsLayout L;
  uint32_t* queue;
  uint32_t* bitmaps;
  CoopView coop;    // enabled == 0: the plain kernel variants
  uint32_t comp_width, grid, block, lds_bytes;
  uint32_t lookup;  // LK_*: layout of the query lookup table in LDS
  uint32_t value_type; // SGPU_VAL_*: how the records store document values
  uint32_t counted; // 1: the accounting variant of the kernel (visited bitmap in HBM, exact work counters)
  uint32_t streamed; // 1: the variant whose stage 2 is a stream (feeder / scorers / replayer wavefronts; KParams::ring)
  hipStream_t stream;
};

hipError_t launch_search(const LaunchArgs& a);
// plan_kernel.hip: the launch plan of a staged chunk computed on the device (order + the maxima the LDS layout wants)
enum { kDevicePlanMaxQueries = 16384, kDevicePlanMinQueries = 256, kDevicePlanCutMax = 16, kCusPerXcd = 32 };
hipError_t launch_device_plan(const DevView& ix, const uint32_t* q_off, const uint32_t* q_comp, const float* q_val, uint32_t nq,
                              uint32_t cut, uint64_t* keys_scratch, uint32_t* maxima, uint32_t* order, hipStream_t stream);

// Registers of wavefront 0 per heap entry array (RegHeap<KR>): the kernel variant that serves k.
inline uint32_t heap_variant(uint32_t k) { return k <= 64 ? 1u : (k <= 128 ? 2u : (k <= 256 ? 4u : (k <= 512 ? 8u : 16u))); }
// Which (workgroup size, heap registers, counted, cooperative) combinations the library instantiates per kernel family:
//   plain        512 threads: every k (<= 1024);  1024 threads (launches of at most n_cu queries): k <= 128
//   counted      512 threads only (the accounting pass is not a latency path)
//   cooperative  512 threads: k <= 256;  1024 threads: k <= 128  (larger k: the plain variant)
// 17 symbols per family (r04: 30, of which the k > 256 cooperative and 1024-thread ones spilled 190 - 230 VGPRs and
// were never chosen by a benchmark configuration).
//   streamed     (r06) the plain variants once more, with stage 2 as a stream: 512 threads every k, 1024 threads k <= 128
// 24 symbols per family.
inline bool variant_built(uint32_t block, uint32_t kr, bool counted, bool coop, bool streamed = false) {
  if (block != 512 && block != 1024) return false;
  if (streamed && (counted || coop)) return false;
  if (counted) return block == 512;
  if (block == 1024) return kr <= 2;
  return coop ? kr <= 4 : true;
}

}  // namespace sgpu
