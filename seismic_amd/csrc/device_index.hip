// device_index.hip — HBM residency of an index, launch configuration, query batches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "device_types.hpp"
#include "host_index.hpp"

namespace sgpu {

// one translation unit per kernel family (sk_*.hip): occupancy query when `occupancy` is given, else launch
hipError_t run_u16_dense(const LaunchArgs& a, int* occupancy);
hipError_t run_u16_packed(const LaunchArgs& a, int* occupancy);
hipError_t run_u32_split(const LaunchArgs& a, int* occupancy);
hipError_t run_u32_packed(const LaunchArgs& a, int* occupancy);
hipError_t run_u32_hash(const LaunchArgs& a, int* occupancy);
hipError_t run_u16_dense_u8(const LaunchArgs& a, int* occupancy);
hipError_t run_u16_packed_u8(const LaunchArgs& a, int* occupancy);
hipError_t run_u32_packed_u8(const LaunchArgs& a, int* occupancy);
hipError_t run_u32_split_u8(const LaunchArgs& a, int* occupancy);
hipError_t run_u16_dense_dvb(const LaunchArgs& a, int* occupancy);
hipError_t run_u16_packed_dvb(const LaunchArgs& a, int* occupancy);
#ifdef SGPU_WITH_F16S   // (make WITH_F16S=1: the sliced layout of an f16 index - measured slower than the plain records, off by default)
hipError_t run_u16_dense_f16s(const LaunchArgs& a, int* occupancy);
hipError_t run_u16_packed_f16s(const LaunchArgs& a, int* occupancy);
#endif

static hipError_t run_any(const LaunchArgs& a, int* occ) {
  // (u16 components: dense byte table or packed {bits, rank} words. The hashed lookup is the u32 layout: for u16 it
  // measured slower than both - 6.46 against 5.81 ms per 10 000-query launch, profiles/r03_lds_sensitivity.md - and its
  // three families were dropped in r05)
  if (a.comp_width == 2 && a.lookup != LK_DENSE && a.lookup != LK_PACKED) return hipErrorInvalidConfiguration;
#ifdef SGPU_WITH_F16S
  if (a.value_type == kDevValF16Sliced) return a.lookup == LK_DENSE ? run_u16_dense_f16s(a, occ) : run_u16_packed_f16s(a, occ);
#else
  if (a.value_type == kDevValF16Sliced) return hipErrorInvalidConfiguration;
#endif
  if (a.value_type == SGPU_VAL_DOTVBYTE)   // (u16 components only)
    return a.lookup == LK_DENSE ? run_u16_dense_dvb(a, occ) : run_u16_packed_dvb(a, occ);
  if (a.value_type == SGPU_VAL_FIXEDU8 && a.comp_width == 4) return a.lookup == LK_SPLIT ? run_u32_split_u8(a, occ) : run_u32_packed_u8(a, occ);
  if (a.value_type == SGPU_VAL_FIXEDU8) return a.lookup == LK_DENSE ? run_u16_dense_u8(a, occ) : run_u16_packed_u8(a, occ);
  if (a.comp_width == 2) return a.lookup == LK_DENSE ? run_u16_dense(a, occ) : run_u16_packed(a, occ);
  if (a.lookup == LK_HASH) return run_u32_hash(a, occ);
  return a.lookup == LK_SPLIT ? run_u32_split(a, occ) : run_u32_packed(a, occ);
}
hipError_t occupancy_search(const LaunchArgs& a, int* blocks_per_cu) { return run_any(a, blocks_per_cu); }
hipError_t launch_search(const LaunchArgs& a) { return run_any(a, nullptr); }

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(SGPU_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// (a test hook read outside the per-call knob cache: only while SGPU_TEST_HOOKS=1 is set; see hooks_on below)
static const char* hook_raw(const char* name) {
  const char* t = std::getenv("SGPU_TEST_HOOKS");
  return (t && *t && *t != '0') ? std::getenv(name) : nullptr;
}

}  // namespace sgpu
struct sgpu_batch;
namespace sgpu {
void batch_free(sgpu_batch* b);

// One stream's worth of launch state. The device-resident batch API (sgpu_batch_*) runs on the
// index's `main` lane; sgpu_search / sgpu_batch_search borrow a lane from a small pool, so calls
// from several host threads on one index overlap (copies and kernels of different lanes run
// concurrently on the device) - the reference's `search(&self)` is re-entrant, S: Sync
// (src/index_traits.rs:106-113).
struct Lane {
  hipStream_t stream = nullptr;
  uint32_t* queue = nullptr;       // the launch's work counter (queries are pulled from it)
  // latency-bound calls (direct-in): a second counter (queue + 16) that is never reset - the host knows its value
  uint32_t queue_pos = 0;          //   ... after everything enqueued so far
  bool queue_dirty = true;         //   ... unless a launch failed (or none ran yet): zero it first
  uint32_t done_seq = 0;           // BatchView::done_seq of the lane's last early-done launch
  sgpu_batch* scratch = nullptr;   // pool lanes: the recycled device batch (no allocation per call)
  uint32_t* bitmaps = nullptr;     // visited bitmaps of the counted pass, one per resident workgroup
  uint32_t bitmaps_slots = 0;
  uint8_t* coop = nullptr;         // board of the cooperative kernel variant (CoopView), all zero between launches
  size_t coop_bytes = 0, coop_trace_off = 0;
  uint32_t coop_trace_n = 0;
  bool coop_last = false;          // the lane's last launch was a cooperative one (its board's error word is then checked)
  std::vector<hipEvent_t> ev0, ev1;   // timing of enqueued launches
  int ev_pending = 0;
  double sum_ms = 0;
  uint32_t n_timed = 0;
  sgpu_launch_stats last{};
  bool busy = false;
};

struct Alloc {
  void* p;
  size_t bytes;
  size_t field;   // byte offset (inside DeviceIndex) of the view pointer that refers to it
};

struct DeviceIndex {
  int device = -1;
  DevView view{};
  uint32_t comp_width = 2;
  std::vector<Alloc> allocs;
  uint64_t bytes = 0;
  uint32_t n_cu = 0;
  uint32_t max_lds = 0;
  std::vector<uint32_t> list_nb, list_np;   // blocks / postings per posting list (host copy)
  uint32_t max_nb = 0;
  uint32_t value_type = SGPU_VAL_F16;   // how the records store document values
  bool fwd_sliced = false;        // an f16 index in the sliced layout (compressed component stream: kernel VT_F16S)
  float val_scale = 0.0f;
  bool fwd_block_major = false;   // forward store holds a copy of every posting's record, block by block
  bool coop_broken = false;       // a cooperative launch reported a protocol error: the variant stays off for this replica
  static constexpr int kMainEvents = 64, kPool = 8;   // two concurrent calls of four chunks each
  Lane main;
  Lane pool[kPool];
  std::unordered_map<uint64_t, int> occupancy;   // kernel variant + LDS size -> workgroups per CU
  // Device-side launch plans (plan_kernel.hip): per query_cut, the largest number of block dots a query of an EARLIER chunk
  // needed (all its lists) - what the LDS layout of the next chunk is sized for; a query that needs more walks its lists in
  // groups. Seeded by the first chunk's host plan, raised by what every device plan reports back with its rows. (mu)
  std::unordered_map<uint32_t, uint32_t> plan_dots_seen;
  std::mutex mu;        // main lane, occupancy cache
  std::atomic<uint32_t> calls_inflight{0};               // entry-point calls on this replica right now (call_enter / call_exit)
  std::atomic<int64_t> concurrent_seen_us{-((int64_t)1 << 60)};   // when two of them were last seen together
  std::mutex pool_mu;   // pool lane hand-out
  std::condition_variable pool_cv;
};

static sgpu_status lane_init(Lane* l, int n_events) {
  if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess)
    return fail(SGPU_EDEVICE, "hipStreamCreate failed");
  if (hipMalloc((void**)&l->queue, 256) != hipSuccess) return fail(SGPU_ENOMEM, "hipMalloc(queue) failed");
  if (hipMemset(l->queue, 0, 256) != hipSuccess) return fail(SGPU_EDEVICE, "hipMemset(queue) failed");   // ([32]: the lane's sticky status word)
  for (int i = 0; i < n_events; ++i) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return fail(SGPU_EDEVICE, "hipEventCreate failed");
    l->ev0.push_back(a);
    if (hipEventCreate(&b) != hipSuccess) return fail(SGPU_EDEVICE, "hipEventCreate failed");
    l->ev1.push_back(b);
  }
  return SGPU_OK;
}

static void lane_free(Lane* l) {
  if (l->stream) (void)hipStreamSynchronize(l->stream);
  if (l->scratch) batch_free(l->scratch);
  if (l->queue) (void)hipFree(l->queue);
  if (l->bitmaps) (void)hipFree(l->bitmaps);
  if (l->coop) (void)hipFree(l->coop);
  for (hipEvent_t e : l->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : l->ev1) (void)hipEventDestroy(e);
  if (l->stream) (void)hipStreamDestroy(l->stream);
  *l = Lane{};
}

Lane* lane_acquire(DeviceIndex* d) {
  std::unique_lock<std::mutex> lk(d->pool_mu);
  for (;;) {
    for (Lane& l : d->pool)
      if (!l.busy) {
        l.busy = true;
        return &l;
      }
    d->pool_cv.wait(lk);
  }
}

// Largest launch the cooperative kernel variant is chosen for on its own (configure); 0 = switched off.
uint32_t coop_auto_max_queries(const DeviceIndex* d) {
  const char* cm = std::getenv("SGPU_COOP");
  if (cm && !std::strcmp(cm, "0")) return 0;
  const char* v = std::getenv("SGPU_COOP_MAX_NQ");
  return v && *v ? (uint32_t)std::strtoul(v, nullptr, 10) : (uint32_t)d->n_cu;
}

// A free lane, or null: a call that already holds one lane never WAITS for another (two callers each
// holding some lanes and waiting for more would deadlock).
Lane* lane_try_acquire(DeviceIndex* d) {
  std::lock_guard<std::mutex> lk(d->pool_mu);
  int n_free = 0;
  for (Lane& l : d->pool) n_free += !l.busy;
  if (n_free < 2) return nullptr;   // the last free lane is left to a caller that has none
  for (Lane& l : d->pool)
    if (!l.busy) {
      l.busy = true;
      return &l;
    }
  return nullptr;
}

// Is this replica serving several calls at a time? call_enter() = true when another entry-point call is in flight on it
// now or was within the last 50 ms (abi.cpp then cuts a large call into fewer chunks: the other calls' kernels already
// hide this call's host side). The short memory matters: request threads that run in lockstep enter at the same instant,
// and the first of them would see an idle replica every time.
static inline int64_t mono_us() {
  return (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
bool call_enter(DeviceIndex* d) {
  const uint32_t n = d->calls_inflight.fetch_add(1, std::memory_order_relaxed) + 1u;
  const int64_t now = mono_us();
  if (n > 1) d->concurrent_seen_us.store(now, std::memory_order_relaxed);
  return n > 1 || now - d->concurrent_seen_us.load(std::memory_order_relaxed) < 50000;
}
void call_exit(DeviceIndex* d) {
  if (d->calls_inflight.fetch_sub(1, std::memory_order_relaxed) > 1u) d->concurrent_seen_us.store(mono_us(), std::memory_order_relaxed);
}

void lane_release(DeviceIndex* d, Lane* l) {
  {
    std::lock_guard<std::mutex> lk(d->pool_mu);
    l->busy = false;
  }
  d->pool_cv.notify_one();
}

Lane* lane_main(DeviceIndex* d) { return &d->main; }
sgpu_batch** lane_scratch(Lane* l) { return &l->scratch; }

template <class T>
static sgpu_status dev_copy(DeviceIndex* d, const T* src, size_t n, const T** out) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  if (hipMalloc(&p, bytes) != hipSuccess) return fail(SGPU_ENOMEM, "hipMalloc of %zu bytes failed", bytes);
  d->allocs.push_back(Alloc{p, bytes, (size_t)((const char*)out - (const char*)d)});
  d->bytes += bytes;
  if (hook_raw("SGPU_DEBUG_ALLOC")) std::fprintf(stderr, "sgpu alloc: field@%zu %p..%p\n", (size_t)((const char*)out - (const char*)d), p, (void*)((char*)p + bytes));
  if (n) HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T*)p;
  return SGPU_OK;
}

void device_index_free(DeviceIndex* d) {
  if (!d) return;
  if (d->device >= 0) (void)hipSetDevice(d->device);
  lane_free(&d->main);
  for (Lane& l : d->pool) lane_free(&l);
  for (const Alloc& a : d->allocs) (void)hipFree(a.p);
  delete d;
}

uint64_t device_index_bytes(const DeviceIndex* d) { return d ? d->bytes : 0; }

int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Block-major forward store: the record of every posting is copied next to the records of its
// block-mates, so that a block's documents are ONE contiguous run of HBM. One 16-lane group per
// posting, 16 bytes per lane and step (the document-major store is the source).
__global__ __launch_bounds__(256) void replicate_records_kernel(uint8_t* fwd, const uint64_t* __restrict__ doc_ref,
                                                                const uint32_t* __restrict__ post_doc,
                                                                const uint64_t* __restrict__ post_ref, uint64_t n_postings,
                                                                uint32_t bytes_per_elem, uint32_t dvb, uint32_t val_bytes) {
  const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const uint32_t sub = threadIdx.x & 15;
  const uint64_t n_groups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
  for (uint64_t p = g; p < n_postings; p += n_groups) {
    const uint64_t dst = post_ref[p], src = doc_ref[post_doc[p]];
    uint32_t n16;   // 16-byte units of the record
    if (dvb && !(dst & 0x8000u)) {   // sliced record (pack_index.cpp: record_bytes)
      const uint32_t ns = (((uint32_t)dst & 0x7fffu) + 7u) >> 3, vb8 = 8u * val_bytes;
      n16 = val_bytes == 1 ? (ns * 20u + 15u) >> 4                                          // DotVByte: [ns x 16 B][ns x 4 B]
                           : ((((ns * 12u + vb8 - 1u) & ~(vb8 - 1u)) + ns * vb8) + 15u) >> 4;   // sliced f16
    } else {
      const uint32_t len = (uint32_t)dst & (dvb ? 0x7fffu : 0xffffu);
      n16 = (((len + 7u) & ~7u) * bytes_per_elem + 15u) >> 4;
    }
    const uint4* s4 = (const uint4*)(fwd + (src >> 16) * 16ull);
    uint4* d4 = (uint4*)(fwd + (dst >> 16) * 16ull);
    for (uint32_t i = sub; i < n16; i += 16) d4[i] = s4[i];
  }
}

// Packs the canonical arrays into the HBM layout and copies them to `device`.
sgpu_status device_index_upload(const HostIndex& h, int device, DeviceIndex** out) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(SGPU_EDEVICE, "no HIP device available (the search path has no CPU fallback)");
  if (device < 0 || device >= n_dev) return fail(SGPU_EDEVICE, "device %d out of range (0..%d)", device, n_dev - 1);
  if (h.n_blocks() >= 0xffffffffull || h.n_postings() >= 0xffffffffull || h.n_rows() >= 0xffffffffull)
    return fail(SGPU_ELIMIT, "index too large for 32-bit device offsets (blocks / postings / summary rows)");
  HIP_TRY(hipSetDevice(device));
  DeviceIndex* d = new DeviceIndex();
  d->device = device;
  d->comp_width = h.comp_width;
  sgpu_status st = SGPU_OK;
  auto bail = [&](sgpu_status s) {
    device_index_free(d);
    return s;
  };
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(fail(SGPU_EDEVICE, "hipGetDeviceProperties failed"));
  d->n_cu = (uint32_t)prop.multiProcessorCount;
  d->max_lds = (uint32_t)prop.sharedMemPerBlock;
  if ((st = lane_init(&d->main, DeviceIndex::kMainEvents)) != SGPU_OK) return bail(st);
  for (Lane& l : d->pool)
    if ((st = lane_init(&l, 1)) != SGPU_OK) return bail(st);

  try {
    const uint32_t cw = h.comp_width, vb = h.val_bytes();
    d->value_type = h.value_type;
    d->val_scale = h.val_scale;
    // ---- document records (pack_index.cpp): [npad comps][npad values (f16, or u8 codes)], npad = len rounded up
    // to 8, the record padded to 16 bytes. A record is moved to the next 128-byte line only if it would
    // otherwise touch more lines than its size needs: at 16-byte alignment a 480-byte record straddles
    // ~4.75 lines, line-fitted 4.
    std::vector<uint64_t> rec_off16;
    std::vector<uint8_t> dvb_raw;   // sliced layouts: the documents that keep the raw record form (a gap too wide for its field)
    // SGPU_FWD_STREAM=sliced (libraries built with `make WITH_F16S=1` only): an f16 index over u16 components takes the
    // SLICED layout - the DotVByte index's compressed component stream in front of the binary16 values, 28 bytes per
    // 8-element slice instead of 32; lossless, rows bit-identical (kernel VT_F16S). Measured r05 on the 8.8M-document
    // collection: 33.0 GB resident instead of 36.3 and 6.02 ms per 10 000-query launch instead of 5.94 - the decode costs
    // more than the bytes return (profiles/r05_bytes_experiments.md); a footprint option, not in the default build.
    {
      const char* fs = std::getenv("SGPU_FWD_STREAM");
      const bool want = fs && std::strcmp(fs, "sliced") == 0;
#ifndef SGPU_WITH_F16S
      if (want) return bail(fail(SGPU_EINVAL, "SGPU_FWD_STREAM=sliced needs a library built with `make WITH_F16S=1`"));
#endif
      pack_dvb_raw_flags(h, want, &dvb_raw);
      d->fwd_sliced = h.value_type == SGPU_VAL_F16 && !dvb_raw.empty();
    }
    {
      const char* env_line = hook_raw("SGPU_REC_LINE");
      pack_record_offsets(h, dvb_raw, std::max<uint64_t>(16, env_line ? std::strtoul(env_line, nullptr, 10) : 128) / 16, &rec_off16);
    }
    if (rec_off16[h.n_docs] >= (1ull << 48)) return bail(fail(SGPU_ELIMIT, "forward index exceeds 48-bit record offsets"));
    std::vector<uint8_t> fwd;
    pack_records(h, dvb_raw, rec_off16, &fwd);
    // ---- postings: (record offset / 16) << 16 | len, the reference's PackedPostingBlock
    // (src/posting_list.rs:32-60). Forward store layout:
    //   block-major (default when it fits): after the document-major records, every posting gets its
    //     own copy of its document's record, laid out block by block (a block starts on a 128-byte
    //     line, its records follow each other). A block's documents are then one contiguous run of
    //     HBM: whole lines are useful and consecutive lines share DRAM pages, where document-major
    //     records are ~4 scattered lines each with the first and last one half used. It costs
    //     n_postings / n_docs (5-6x on MS MARCO shapes: 23 GB) of the 288 GB; used up to 96 GB of copies
    //     (the 5M x 200K-vocabulary shape would take 195 GB and, not being bound by line fetches with
    //     its u32 components and packed lookup, gains nothing from them: measured 63.6% either way).
    //   document-major: one record per document, postings point into it (large indexes).
    const uint64_t doc_units = rec_off16[h.n_docs];
    std::vector<uint64_t> pref;
    uint64_t blk_units = 0;
    {
      std::vector<uint64_t> bsize;
      pack_block_sizes(h, dvb_raw, &bsize);
      blk_units = bsize[h.n_blocks()];
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const char* env_layout = std::getenv("SGPU_FWD_LAYOUT");   // "block" | "doc"; default: block when it fits
      const uint64_t need = (doc_units + blk_units) * 16 + h.n_postings() * 24 + h.n_entries() * 8;
      bool block_major = blk_units > 0 && (doc_units + 8 + blk_units) < (1ull << 48) &&
                         need < (uint64_t)(0.6 * (double)free_b) && blk_units * 16 <= (96ull << 30);
      if (env_layout && std::strcmp(env_layout, "doc") == 0) block_major = false;
      if (env_layout && std::strcmp(env_layout, "block") == 0 && blk_units > 0) block_major = true;
      const uint64_t blk_base = (doc_units + 7) & ~7ull;
      // the store is allocated before the posting refs are laid out: if the block-major region does
      // not fit after all (memory taken by another process), the index falls back to document-major
      void* fp = nullptr;
      size_t fbytes = 0;
      for (;;) {
        const uint64_t total_units = block_major ? blk_base + blk_units : doc_units;
        fbytes = std::max<uint64_t>(total_units * 16, 16) + 16;   // (+16: a 16-byte load may start 12 bytes before the end)
        if (hipMalloc(&fp, fbytes) == hipSuccess) break;
        (void)hipGetLastError();
        if (!block_major) return bail(fail(SGPU_ENOMEM, "hipMalloc of %zu bytes failed", fbytes));
        block_major = false;
      }
      d->fwd_block_major = block_major;
      pack_post_refs(h, dvb_raw, rec_off16, bsize, block_major, blk_base, &pref);
      d->allocs.push_back(Alloc{fp, fbytes, (size_t)((const char*)&d->view.fwd - (const char*)d)});
      d->bytes += fbytes;
      d->view.fwd = (const uint8_t*)fp;
      if (hook_raw("SGPU_DEBUG_ALLOC")) std::fprintf(stderr, "sgpu alloc: fwd %p..%p\n", fp, (void*)((char*)fp + fbytes));
      HIP_TRY(hipMemcpy(fp, fwd.data(), fwd.size(), hipMemcpyHostToDevice));
    }
    fwd.clear();
    fwd.shrink_to_fit();
    if ((st = dev_copy(d, pref.data(), pref.size(), &d->view.post_ref)) != SGPU_OK) return bail(st);
    {
      std::vector<uint64_t> dref;
      pack_doc_refs(h, dvb_raw, rec_off16, &dref);
      if ((st = dev_copy(d, dref.data(), dref.size(), &d->view.doc_ref)) != SGPU_OK) return bail(st);
    }
    pref.clear();
    pref.shrink_to_fit();
    if ((st = dev_copy(d, h.post_doc.data(), h.post_doc.size(), &d->view.post_doc)) != SGPU_OK) return bail(st);
    if (d->fwd_block_major && h.n_postings()) {
      (void)hipGetLastError();   // (judge this launch alone: an earlier failed call of the thread leaves its error behind)
      hipLaunchKernelGGL(replicate_records_kernel, dim3(d->n_cu * 8), dim3(256), 0, d->main.stream,
                         (uint8_t*)d->view.fwd, d->view.doc_ref, d->view.post_doc, d->view.post_ref,
                         (uint64_t)h.n_postings(), (uint32_t)(cw + vb), (uint32_t)(!dvb_raw.empty()), (uint32_t)vb);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(d->main.stream));
    }
    auto narrow = [](const std::vector<uint64_t>& v) {
      std::vector<uint32_t> o;
      pack_narrow(v, &o);
      return o;
    };
    {
      auto v = narrow(h.list_block_start);
      if ((st = dev_copy(d, v.data(), v.size(), &d->view.list_block_start)) != SGPU_OK) return bail(st);
      v = narrow(h.block_post_start);
      if ((st = dev_copy(d, v.data(), v.size(), &d->view.block_post_start)) != SGPU_OK) return bail(st);
    }
    if ((st = dev_copy(d, h.sum_bid.data(), h.sum_bid.size(), &d->view.sum_bid)) != SGPU_OK) return bail(st);
    {
      // split point of every summary row at half the list's block ids (rows are ascending in block
      // id, validate_desc): stage 1 gives each half of a list to its own wavefront
      std::vector<uint16_t> mid;
      pack_row_mid(h, &mid);
      d->view.list_row_start = nullptr;
      d->view.row_ptr = nullptr;
      d->view.row_comp = nullptr;
      d->view.row_mid = nullptr;
      d->view.row_dir = nullptr;
      d->view.row_dir_buckets = 0;
      if (cw == 2) {
        // u16 components: the hashed row directory IS the row index on the device - a query component's summary row
        // {first entry, entries, split point} in one trip (search_kernel.inc: build_row_table); the CSR arrays the
        // binary search of the u32 kernels walks (row_comp, row_ptr, row_mid, list_row_start: 1.7 GB for the 8.8M-document
        // index) are not uploaded at all.
        std::vector<uint32_t> dir;
        uint32_t n_buckets = 0;
        if (!pack_row_dir(h, mid, &dir, &n_buckets))
          return bail(fail(SGPU_ELIMIT, h.dim > 65535 ? "u16 components need dim <= 65535 for the device's row directory"
                                                       : "the index has too many summary rows for the device's row directory (2^31 buckets)"));
        static_assert(sizeof(d->view.row_dir) == sizeof(const uint32_t*), "pointer field");
        if ((st = dev_copy(d, dir.data(), dir.size(), (const uint32_t**)&d->view.row_dir)) != SGPU_OK) return bail(st);
        d->view.row_dir_buckets = n_buckets;
      } else {
        auto v = narrow(h.list_row_start);
        if ((st = dev_copy(d, v.data(), v.size(), &d->view.list_row_start)) != SGPU_OK) return bail(st);
        if ((st = dev_copy(d, h.row_ptr.data(), h.row_ptr.size(), &d->view.row_ptr)) != SGPU_OK) return bail(st);
        static_assert(sizeof(d->view.row_comp) == sizeof(const uint8_t*), "pointer field");
        if ((st = dev_copy(d, h.row_comp.data(), h.row_comp.size(), (const uint8_t**)&d->view.row_comp)) != SGPU_OK)
          return bail(st);
        if ((st = dev_copy(d, mid.data(), mid.size(), &d->view.row_mid)) != SGPU_OK) return bail(st);
      }
    }
    {
      // dequantised summary values: code*quant + min with the reference's two roundings
      // (src/quantized_summary.rs:102-104; this file is compiled with -ffp-contract=off)
      std::vector<float> deq;
      pack_sum_deq(h, &deq);
      if ((st = dev_copy(d, deq.data(), deq.size(), &d->view.sum_deq)) != SGPU_OK) return bail(st);
    }
    d->view.knn = nullptr;
    d->view.knn_total = 0;
    d->view.knn_dim = 0;
    if (!h.knn.empty() && h.knn_dim) {
      if ((st = dev_copy(d, h.knn.data(), h.knn.size(), &d->view.knn)) != SGPU_OK) return bail(st);
      d->view.knn_total = h.knn.size();
      d->view.knn_dim = h.knn_dim;
    }
    d->view.dim = (uint32_t)h.dim;
    d->view.n_docs = (uint32_t)h.n_docs;
    d->view.n_bitmap_words = (uint32_t)((h.n_docs + 31) / 32);
#if defined(SGPU_LAZY_DOCS) && SGPU_LAZY_DOCS
    d->view.n_postings_lt_2g = h.n_postings() < (1ull << 31) ? 1u : 0u;
#endif
    // ---- block-count statistics for LDS sizing
    d->list_nb.resize(h.dim);
    d->list_np.resize(h.dim);
    for (uint64_t c = 0; c < h.dim; ++c) {
      d->list_nb[c] = (uint32_t)(h.list_block_start[c + 1] - h.list_block_start[c]);
      d->list_np[c] = (uint32_t)(h.block_post_start[h.list_block_start[c + 1]] - h.block_post_start[h.list_block_start[c]]);
      d->max_nb = std::max(d->max_nb, d->list_nb[c]);
    }
  } catch (const std::bad_alloc&) {
    return bail(fail(SGPU_ENOMEM, "out of host memory packing the index for upload"));
  }
  *out = d;
  return SGPU_OK;
}

// A replica of `src` on HIP device `device`: every array is copied GPU to GPU (hipMemcpyPeer goes
// over xGMI between the MI355Xs of a node; same-device "replicas" are plain device copies), nothing
// is repacked or re-sent from the host.
sgpu_status device_index_clone(const DeviceIndex* src, int device, DeviceIndex** out) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(SGPU_EDEVICE, "no HIP device available");
  if (device < 0 || device >= n_dev) return fail(SGPU_EDEVICE, "device %d out of range (0..%d)", device, n_dev - 1);
  HIP_TRY(hipSetDevice(device));
  if (device != src->device) {
    int can = 0;
    (void)hipDeviceCanAccessPeer(&can, device, src->device);
    if (can) {
      const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    }
  }
  DeviceIndex* d = new DeviceIndex();
  d->device = device;
  d->comp_width = src->comp_width;
  d->view = src->view;
  d->n_cu = src->n_cu;
  d->max_lds = src->max_lds;
  d->list_nb = src->list_nb;
  d->list_np = src->list_np;
  d->max_nb = src->max_nb;
  d->fwd_block_major = src->fwd_block_major;
  d->value_type = src->value_type;
  d->fwd_sliced = src->fwd_sliced;
  d->val_scale = src->val_scale;
  auto bail = [&](sgpu_status s) {
    device_index_free(d);
    return s;
  };
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(fail(SGPU_EDEVICE, "hipGetDeviceProperties failed"));
  d->n_cu = (uint32_t)prop.multiProcessorCount;
  d->max_lds = (uint32_t)prop.sharedMemPerBlock;
  sgpu_status st;
  if ((st = lane_init(&d->main, DeviceIndex::kMainEvents)) != SGPU_OK) return bail(st);
  for (Lane& l : d->pool)
    if ((st = lane_init(&l, 1)) != SGPU_OK) return bail(st);
  // pointers of the copied view must not be freed if an allocation below fails half way
  for (const Alloc& a : src->allocs) *(const void**)((char*)d + a.field) = nullptr;
  for (const Alloc& a : src->allocs) {
    void* p = nullptr;
    if (hipMalloc(&p, a.bytes) != hipSuccess) return bail(fail(SGPU_ENOMEM, "hipMalloc of %zu bytes failed on device %d", a.bytes, device));
    d->allocs.push_back(Alloc{p, a.bytes, a.field});
    d->bytes += a.bytes;
    *(const void**)((char*)d + a.field) = p;
    const hipError_t e = hipMemcpyPeerAsync(p, device, a.p, src->device, a.bytes, d->main.stream);
    if (e != hipSuccess) return bail(fail(SGPU_EDEVICE, "hipMemcpyPeerAsync %d -> %d failed: %s", src->device, device, hipGetErrorString(e)));
  }
  if (hipStreamSynchronize(d->main.stream) != hipSuccess) return bail(fail(SGPU_EDEVICE, "peer copy failed"));
  *out = d;
  return SGPU_OK;
}

// ---------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------
}  // namespace sgpu

struct sgpu_batch_plan {   // per (batch, query_cut): LDS need and processing order (host side)
  uint32_t query_cut = 0;
  uint32_t dots_cap = 1;    // max over queries of the blocks of the lists it walks
  uint32_t max_nb = 0;      // largest single list walked first (sort buffer sizing)
  uint32_t max_list_nb = 1; // largest single list walked at all (smallest possible dots area)
  std::vector<uint32_t> order;   // queries, longest expected first; followed (second half) by every query's
                                 // LK_HASH seed: the hash multiplier under which its components do not collide
  bool hash_ok = true;           // every query of the batch has such a seed
};

struct sgpu_batch {
  std::vector<uint64_t> h_off;
  std::vector<uint32_t> h_off32;   // what the device holds (the copies below read the batch's own arrays,
                                   // which live as long as the batch: nothing depends on when a copy from
                                   // pageable memory lets go of its source)
  std::vector<uint32_t> h_comp;
  std::vector<float> h_val;
  std::vector<sgpu_batch_plan> plans;
  int device = -1;
  const sgpu::DeviceIndex* owner = nullptr;   // the replica the batch was created on
  uint32_t nq = 0, k_max = 0, max_nnz = 0;
  uint64_t cap_nq = 0, cap_nnz = 0, cap_slab = 0;   // allocated capacities (a scratch batch is reused)
  uint32_t* q_off = nullptr;
  uint32_t* q_comp = nullptr;
  float* q_val = nullptr;
  uint32_t* q_order = nullptr;     // cap_nq: the processing order of the plan named by order_cut
  uint32_t* h_order = nullptr;     // its pinned host staging copy (the H2D is then truly asynchronous)
  uint32_t order_cut = 0xffffffffu;
  float* out_scores = nullptr;
  uint64_t* out_ids = nullptr;
  uint32_t* out_n = nullptr;
  uint32_t* out_stats = nullptr;   // nq x STATS_WORDS work counters of the last pass
  uint32_t* status = nullptr;      // staged batches: the launch status word (BatchView::status) as the kernel addresses it
  size_t status_off = 0;           //   ... and where it lies in the arena (the last 16 bytes of the input region)
  // Staged batches (the recycled batch of a pool lane, behind sgpu_search / sgpu_batch_search): every
  // device array above is a slice of ONE device arena mirrored by ONE pinned host buffer, so a call is
  // one H2D (work counter, queries, launch order), the kernel, one D2H (counts, scores, ids).
  bool staged = false;
  uint8_t* arena_dev = nullptr;
  uint8_t* arena_host = nullptr;
  uint8_t* arena_host_dev = nullptr;   // the pinned host arena as the device sees it (small calls write their rows straight into it)
  bool direct_out = false;
  uint32_t done_seq = 0;               // direct-out calls: the value the launch's last query stores into the arena's done word (0: none)
  bool direct_in = false;              // the kernel reads the queries from the pinned host arena (no H2D copy to enqueue)
  uint32_t queue_base = 0;             // KParams::queue_base of the next launch
  size_t arena_cap = 0, in_bytes = 0, out_off = 0, out_bytes = 0;
  uint32_t* queue_dev = nullptr;
  uint32_t device_plan_cut = 0xffffffffu;   // the chunk's launch plan was computed on the device for this query_cut (its
                                            //   maxima come back in words 1 - 3 of the status block); 0xffffffff: host plan
  bool followed = false;                    // another chunk of the call comes after this one, or other calls are in flight
  bool plan_identity = false;               // a device-planned chunk that takes its queries in input order (staged_launch)
};

namespace sgpu {

// Environment knobs (SGPU_*: experiments, tests, debugging aids) are looked at on the per-launch path; a getenv
// walks the whole environment, and a launch reads some forty of them. They are therefore cached per thread
// and keyed by a fingerprint of the environment (the pointers of `environ`: setenv / unsetenv replace an
// entry's pointer, which is what Python's os.environ and the test suite's monkeypatch do), taken once per call
// (env_refresh): ~0.1 us per launch instead of ~4 (r03: configure + launch 6.6 us of a 148 us single query).
// A putenv() of a buffer edited in place is not noticed - these are not product switches.
}  // namespace sgpu
extern char** environ;
namespace sgpu {
struct EnvCache {
  uint64_t fp = 0;
  int n = 0;
  struct E {
    const char* name;
    const char* val;
  } e[96];
};
static thread_local EnvCache tl_env;
static void env_refresh() {
  uint64_t h = 1469598103934665603ull;
  for (char** p = environ; p && *p; ++p) h = (h ^ (uint64_t)(uintptr_t)*p) * 1099511628211ull;
  if (h != tl_env.fp || tl_env.n == 0) {
    tl_env.fp = h;
    tl_env.n = 0;
  }
}
static const char* env_get(const char* name) {   // `name` is a string literal: compared by address first
  EnvCache& c = tl_env;
  for (int i = 0; i < c.n; ++i)
    if (c.e[i].name == name) return c.e[i].val;
  const char* v = std::getenv(name);
  if (c.n < (int)(sizeof(c.e) / sizeof(c.e[0]))) c.e[c.n++] = EnvCache::E{name, v};
  return v;
}
static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* v = env_get(name);
  if (!v || !*v) return dflt;
  return (uint32_t)std::strtoul(v, nullptr, 10);
}
// The runtime knobs of the library are the ones INTEGRATION.md section 5 lists (read with env_get / env_u32). Every other
// SGPU_* name is a TEST HOOK - it forces a code path the launch configuration would not choose by itself (small item
// tables, a lookup layout, cooperative chunk sizes ...) - and is honoured only while SGPU_TEST_HOOKS=1 is set, which
// tests/conftest.py and the tools under tools/ do: a deployment's behaviour does not depend on them.
static bool hooks_on() {
  const char* v = env_get("SGPU_TEST_HOOKS");
  return v && *v && *v != '0';
}
// (a tool that sets a hook without the switch used to measure the default configuration in silence - the
// library now says so, once per process)
static void hook_ignored(const char* name) {
  static std::atomic<bool> said{false};
  const char* v = env_get(name);
  if (v && *v && !said.exchange(true))
    std::fprintf(stderr, "seismic_hip: %s is a test hook and is ignored unless SGPU_TEST_HOOKS=1 is set (further ignored hooks are not reported)\n", name);
}
static uint32_t hook_u32(const char* name, uint32_t dflt) {
  if (hooks_on()) return env_u32(name, dflt);
  hook_ignored(name);
  return dflt;
}
static const char* hook_get(const char* name) {
  if (hooks_on()) return env_get(name);
  hook_ignored(name);
  return nullptr;
}

// ---- host-side breakdown of a staged call (tools/latency_probe.py, sgpu_search_sequential) ----
// A thread that sets call_timing() gets the wall time of every phase of its staged calls added up:
// [0] validate + launch plan  [1] staging (pinned arena)  [2] enqueue H2D  [3] configure + kernel launch
// [4] enqueue D2H  [5] wait for the stream  [6] copy the rows out
double*& call_timing() {
  static thread_local double* t = nullptr;
  return t;
}
static inline double now_us() {
  return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3;
}
struct PhaseClock {
  double* acc;
  double t;
  PhaseClock() : acc(call_timing()), t(acc ? now_us() : 0.0) {}
  inline void lap(int i) {
    if (!acc) return;
    const double n = now_us();
    acc[i] += n - t;
    t = n;
  }
};

// How a call waits for its lane. hipStreamSynchronize parks the thread on an interrupt; waking it costs
// what the host's idle state costs (tens to hundreds of microseconds on a quiet box: VERDICT r02 #8 saw
// 192 us outside the kernel on the driver's box against 17 us on a busy one). A call small enough to be
// latency-bound polls the stream instead (hipStreamQuery, no system call once the signal is mapped)
// for at most SGPU_SPIN_US microseconds (default 2000) and only then parks. SGPU_WAIT=block|spin overrides.
// (done / done_seq: the launch stores done_seq into *done - pinned host memory - once every result row is there: the
// wait ends then, whether or not the launch itself has ended)
static hipError_t wait_lane(hipStream_t s, uint32_t nq, const volatile uint32_t* done = nullptr, uint32_t done_seq = 0) {
  // (both knobs are read through the per-call cache, like every other one: a test that flips them is not silently ignored)
  const char* wv = env_get("SGPU_WAIT");
  const int mode = (wv && !std::strcmp(wv, "block")) ? 0 : ((wv && !std::strcmp(wv, "spin")) ? 2 : 1);
  const double spin_us = (double)env_u32("SGPU_SPIN_US", 2000);
  if (mode == 2 || (mode == 1 && nq <= 256)) {
    const double t0 = now_us();
    for (;;) {
      if (done && __atomic_load_n((const uint32_t*)done, __ATOMIC_ACQUIRE) == done_seq) return hipSuccess;
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) return hipSuccess;
      if (e != hipErrorNotReady) return e;
      if (now_us() - t0 > spin_us) break;
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#elif defined(__aarch64__)
      asm volatile("yield");
#endif
    }
  }
  return hipStreamSynchronize(s);
}

void batch_free(sgpu_batch* b) {
  if (!b) return;
  if (b->device >= 0) (void)hipSetDevice(b->device);
  if (b->staged) {
    (void)hipFree(b->arena_dev);
    (void)hipHostFree(b->arena_host);
    delete b;
    return;
  }
  (void)hipFree(b->q_off);
  (void)hipFree(b->q_comp);
  (void)hipFree(b->q_val);
  (void)hipFree(b->q_order);
  (void)hipHostFree(b->h_order);
  (void)hipFree(b->out_scores);
  (void)hipFree(b->out_ids);
  (void)hipFree(b->out_n);
  (void)hipFree(b->out_stats);
  delete b;
}

// Creates (or, when *out already holds a large enough batch of the same device, refills) a device
// batch. The copies are enqueued on `lane`'s stream from the batch's own host copies of the arrays;
// the caller's buffers are not referenced after the call returns.
sgpu_status batch_create(DeviceIndex* d, Lane* lane, uint64_t dim, const uint64_t* q_off, const uint32_t* comps,
                         const float* vals, uint32_t nq, uint32_t k_max, sgpu_batch** out) {
  if (!d) return fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload)");
  if (k_max == 0) return fail(SGPU_EINVAL, "k == 0");
  if (k_max > 1024) return fail(SGPU_ELIMIT, "k = %u exceeds the heap limit of 1024", k_max);
  uint32_t max_nnz = 0;
  sgpu_status st = validate_queries(dim, q_off, comps, vals, nq, &max_nnz);
  if (st != SGPU_OK) return st;
  HIP_TRY(hipSetDevice(d->device));
  const uint64_t nnz = q_off[nq];
  const size_t slab = std::max<size_t>((size_t)nq * k_max, 1);
  sgpu_batch* b = *out;
  const bool reuse = b && b->owner == d && b->cap_nq >= nq && b->cap_nnz >= nnz && b->cap_slab >= slab;
  if (b && !reuse) {
    HIP_TRY(hipStreamSynchronize(lane->stream));
    batch_free(b);
    b = nullptr;
    *out = nullptr;
  }
  if (reuse) HIP_TRY(hipStreamSynchronize(lane->stream));   // earlier copies out of the batch's host arrays are done
  try {
    if (!b) b = new sgpu_batch();
    b->device = d->device;
    b->owner = d;
    b->nq = nq;
    b->k_max = k_max;
    b->max_nnz = max_nnz;
    b->plans.clear();
    b->order_cut = 0xffffffffu;
    b->h_off.assign(q_off, q_off + nq + 1);
    b->h_comp.assign(comps, comps + nnz);
    b->h_val.assign(vals, vals + nnz);
    b->h_off32.resize((size_t)nq + 1);
  } catch (const std::bad_alloc&) {
    if (!reuse) delete b;
    else b->nq = 0;   // the recycled batch stays valid, and empty
    return fail(SGPU_ENOMEM, "out of host memory creating a query batch");
  }
  for (uint32_t q = 0; q <= nq; ++q) b->h_off32[q] = (uint32_t)q_off[q];
  bool ok = true;
  if (!reuse) {
    // a recycled batch grows geometrically so that a stream of slightly different calls settles
    b->cap_nq = std::max<uint64_t>(nq, 1);
    b->cap_nnz = std::max<uint64_t>(nnz, 1);
    b->cap_slab = slab;
    ok = hipMalloc((void**)&b->q_off, (b->cap_nq + 1) * 4) == hipSuccess &&
         hipMalloc((void**)&b->q_comp, b->cap_nnz * 4) == hipSuccess &&
         hipMalloc((void**)&b->q_val, b->cap_nnz * 4) == hipSuccess &&
         hipMalloc((void**)&b->q_order, b->cap_nq * 8) == hipSuccess &&
         hipHostMalloc((void**)&b->h_order, b->cap_nq * 8, hipHostMallocDefault) == hipSuccess &&
         hipMalloc((void**)&b->out_scores, std::max<size_t>(slab, 65536) * 4) == hipSuccess &&
         hipMalloc((void**)&b->out_ids, slab * 8) == hipSuccess &&
         hipMalloc((void**)&b->out_n, b->cap_nq * 4) == hipSuccess &&
         hipMalloc((void**)&b->out_stats, b->cap_nq * STATS_WORDS * 4) == hipSuccess;
  }
  if (!ok) {
    batch_free(b);
    *out = nullptr;
    return fail(SGPU_ENOMEM, "hipMalloc failed creating a query batch");
  }
  ok = hipMemcpyAsync(b->q_off, b->h_off32.data(), (nq + 1) * 4, hipMemcpyHostToDevice, lane->stream) == hipSuccess &&
       (nnz == 0 ||
        (hipMemcpyAsync(b->q_comp, b->h_comp.data(), nnz * 4, hipMemcpyHostToDevice, lane->stream) == hipSuccess &&
         hipMemcpyAsync(b->q_val, b->h_val.data(), nnz * 4, hipMemcpyHostToDevice, lane->stream) == hipSuccess));
  if (!ok) {
    (void)hipStreamSynchronize(lane->stream);
    batch_free(b);
    *out = nullptr;
    return fail(SGPU_EDEVICE, "hipMemcpy of the query batch failed");
  }
  *out = b;
  return SGPU_OK;
}

const DeviceIndex* batch_replica(const sgpu_batch* b) { return b ? b->owner : nullptr; }
static inline size_t al16_(size_t x) { return (x + 15) & ~(size_t)15; }

// Which lists each query will walk (the device applies the same rule: query_cut heaviest
// components by f32::total_cmp, ties by ascending component), hence how many block dots it
// needs in LDS, and an a-priori cost (postings of those lists) to start long queries first.
static sgpu_status make_plan(const DeviceIndex* d, const uint64_t* h_off, const uint32_t* h_comp, const float* h_val,
                             uint32_t nq, uint32_t query_cut, sgpu_batch_plan* out) {
  try {
    sgpu_batch_plan& pl = *out;
    pl.query_cut = query_cut;
    std::vector<uint64_t> keys(nq);
    uint32_t max_nb = 0, dots_cap = 1, max_list_nb = 1;
    // (test hook SGPU_AFFINITY_CLASSES = n > 0, the bytes experiment of r04 / r05: inside each of n cost classes of the
    // longest-first order, queries that walk the same FIRST list are queued next to each other - they then run at the
    // same time on different workgroups and can meet each other's summary rows and records in L2 / the Infinity Cache.
    // Measured with counters in r05, profiles/r05_bytes_experiments.md: no effect on traffic or time; off.)
    const uint32_t aff_classes = nq >= 4096 ? hook_u32("SGPU_AFFINITY_CLASSES", 0) : 0;
    std::vector<uint32_t> first_list;
    if (aff_classes) first_list.resize(nq);
    {   // serial on purpose (an OpenMP team costs more to wake than this takes): ~60 ns per query for query_cut <= 16
      constexpr uint32_t kSmall = 16;
      std::vector<std::pair<int32_t, uint32_t>> kv;   // (large query_cut only)
      for (int64_t q = 0; q < (int64_t)nq; ++q) {
        const uint64_t qa = h_off[q], qe = h_off[q + 1];
        // the query_cut heaviest components by (f32::total_cmp descending, component ascending) - the order the
        // kernel's select_lists produces. Components arrive ascending, so of two equal keys the earlier one wins.
        int32_t tk[kSmall];
        uint32_t tc[kSmall];
        const uint32_t* sel = tc;
        size_t nl = 0;
        if (query_cut == 0) {
          // (no list is walked: the reference's k_largest_by(0))
        } else if (query_cut <= kSmall) {
          for (uint64_t i = qa; i < qe; ++i) {
            const int32_t key = total_key(h_val[i]);
            if (nl == query_cut && !(key > tk[nl - 1])) continue;
            size_t j = nl < query_cut ? nl++ : nl - 1;
            while (j > 0 && tk[j - 1] < key) {
              tk[j] = tk[j - 1];
              tc[j] = tc[j - 1];
              --j;
            }
            tk[j] = key;
            tc[j] = h_comp[i];
          }
        } else {
          kv.clear();
          for (uint64_t i = qa; i < qe; ++i) kv.emplace_back(total_key(h_val[i]), h_comp[i]);
          nl = std::min<size_t>(query_cut, kv.size());
          std::partial_sort(kv.begin(), kv.begin() + (long)nl, kv.end(),
                            [](const std::pair<int32_t, uint32_t>& a, const std::pair<int32_t, uint32_t>& c) {
                              if (a.first != c.first) return a.first > c.first;
                              return a.second < c.second;
                            });
        }
        uint64_t np = 0;
        uint32_t nb = 0;
        for (size_t i = 0; i < nl; ++i) {
          const uint32_t c = query_cut <= kSmall ? sel[i] : kv[i].second;
          nb += d->list_nb[c];
          np += d->list_np[c];
          max_list_nb = std::max(max_list_nb, d->list_nb[c]);
        }
        const uint32_t c0 = nl ? (query_cut <= kSmall ? sel[0] : kv[0].second) : 0xffffffffu;
        if (nl) max_nb = std::max(max_nb, d->list_nb[c0]);
        if (!first_list.empty()) first_list[(size_t)q] = c0;
        dots_cap = std::max(dots_cap, nb);
        keys[(size_t)q] = ((uint64_t)(0xffffffffu - (uint32_t)std::min<uint64_t>(np, 0xffffffffull)) << 32) | (uint32_t)q;
      }
    }
    pl.max_nb = max_nb;
    pl.dots_cap = dots_cap;
    pl.max_list_nb = max_list_nb;
    // longest expected first, ties in input order: ONE 64-bit key per query, the cost complemented in the high word and
    // the query in the low one, so a plain ascending integer sort (half the time of a sort of pairs through a
    // comparator, which was a quarter of the host side of a call). A cost saturates at 2^32 - 1 postings - four times the
    // postings of the whole 8.8M-document collection - beyond which queries simply keep their input order.
    std::sort(keys.begin(), keys.end());
    if (aff_classes) {
      const size_t per = ((size_t)nq + aff_classes - 1) / aff_classes;
      for (size_t c0 = 0; c0 < nq; c0 += per)
        std::stable_sort(keys.begin() + (long)c0, keys.begin() + (long)std::min<size_t>(nq, c0 + per),
                         [&](uint64_t x, uint64_t y) { return first_list[(uint32_t)x] < first_list[(uint32_t)y]; });
    }
    pl.order.resize(2 * (size_t)nq);
    for (uint32_t i = 0; i < nq; ++i) pl.order[i] = (uint32_t)keys[i];
    // LK_HASH seeds: the first multiplier of the family that sends the query's components to distinct slots
    // (component ids below 2^24; a query may fill at most a quarter of the table)
    // (only u32 components with f16 values have hashed kernels: configure; the seeds cost a millisecond of host time per
    // 10 000 queries and are not computed for the others)
    pl.hash_ok = d->view.dim < (1u << 24) && d->comp_width == 4 && d->value_type == SGPU_VAL_F16 && !hook_u32("SGPU_NO_HASH", 0);
    if (pl.hash_ok) {
      std::vector<uint32_t> stamp(kHashSlots, 0xffffffffu);
      uint32_t epoch = 0;
      for (uint32_t q = 0; q < nq && pl.hash_ok; ++q) {
        const uint64_t a = h_off[q], e = h_off[q + 1];
        uint32_t seed = kHashSeeds;
        if (e - a <= kHashSlots / 4)
          for (uint32_t s = 0; s < kHashSeeds && seed == kHashSeeds; ++s) {
            const uint32_t mult = hash_mult(s);
            bool clash = false;
            for (uint64_t i = a; i < e && !clash; ++i) {
              uint32_t& slot = stamp[hash_slot(h_comp[i], mult)];
              clash = slot == epoch;
              slot = epoch;
            }
            ++epoch;
            if (!clash) seed = s;
          }
        if (seed == kHashSeeds) pl.hash_ok = false;
        pl.order[(size_t)nq + q] = seed & (kHashSeeds - 1);
      }
    }
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory planning a query batch");
  }
  return SGPU_OK;
}

// (test hook, no device needed: the launch plan of a batch against a host index - the processing order, the block
// dots a query needs at most, the largest list walked first / at all. sgpu_debug_plan, tests/test_abi_and_host.py)
sgpu_status debug_plan(const HostIndex& h, const uint64_t* q_off, const uint32_t* comps, const float* vals, uint32_t nq,
                       uint32_t query_cut, uint32_t* order_out, uint32_t* out3) {
  DeviceIndex d;
  d.comp_width = h.comp_width;
  d.view.dim = (uint32_t)h.dim;
  d.list_nb.resize(h.dim);
  d.list_np.resize(h.dim);
  for (uint64_t c = 0; c < h.dim; ++c) {
    d.list_nb[c] = (uint32_t)(h.list_block_start[c + 1] - h.list_block_start[c]);
    d.list_np[c] = (uint32_t)(h.block_post_start[h.list_block_start[c + 1]] - h.block_post_start[h.list_block_start[c]]);
  }
  sgpu_batch_plan pl;
  env_refresh();
  const sgpu_status st = make_plan(&d, q_off, comps, vals, nq, query_cut, &pl);
  if (st != SGPU_OK) return st;
  for (uint32_t i = 0; i < nq; ++i) order_out[i] = pl.order[i];
  out3[0] = pl.dots_cap;
  out3[1] = pl.max_nb;
  out3[2] = pl.max_list_nb;
  return SGPU_OK;
}

// Will chunks of a call with these parameters be planned on the device (once a first chunk has seeded the cache)? abi.cpp
// then cuts a call into fewer chunks: there is no host-side planning left to hide behind the previous chunk's kernel.
bool device_plan_applies(const DeviceIndex* d, const sgpu_search_params& sp) {
  const bool hash_family = d->comp_width == 4 && d->value_type == SGPU_VAL_F16 && d->view.dim < (1u << 24);
  const char* v = std::getenv("SGPU_DEVICE_PLAN");
  return sp.query_cut >= 1 && sp.query_cut <= kDevicePlanCutMax && !sp.first_sorted && !hash_family && !(v && *v == '0');
}

// (test hook: the DEVICE's launch plan of a batch - plan_kernel.hip - in the terms of debug_plan: order, out3 = {block dots
// a query needs at most, largest list walked first, largest list walked}. sgpu_debug_device_plan, tests/test_gpu_boundary.py)
sgpu_status debug_device_plan(DeviceIndex* d, const uint64_t* q_off, const uint32_t* comps, const float* vals, uint32_t nq,
                              uint32_t query_cut, uint32_t* order_out, uint32_t* out3) {
  if (!d) return fail(SGPU_EDEVICE, "index is not uploaded to a device");
  if (nq == 0 || nq > kDevicePlanMaxQueries || query_cut == 0 || query_cut > kDevicePlanCutMax)
    return fail(SGPU_EINVAL, "the device plans 1 ... %u queries at query_cut 1 ... %u", (unsigned)kDevicePlanMaxQueries, (unsigned)kDevicePlanCutMax);
  HIP_TRY(hipSetDevice(d->device));
  const uint64_t nnz = q_off[nq];
  uint32_t n2 = 2;
  while (n2 < nq) n2 <<= 1;
  std::vector<uint32_t> off32((size_t)nq + 1);
  for (uint32_t q = 0; q <= nq; ++q) off32[q] = (uint32_t)q_off[q];
  const size_t o_comp = al16_((size_t)(nq + 1) * 4), o_val = o_comp + al16_(nnz * 4), o_keys = o_val + al16_(nnz * 4),
               o_max = o_keys + (size_t)n2 * 8, o_order = o_max + 16, total = o_order + (size_t)nq * 4;
  uint8_t* buf = nullptr;
  if (hipMalloc((void**)&buf, total) != hipSuccess) return fail(SGPU_ENOMEM, "hipMalloc of %zu bytes failed", total);
  hipError_t e = hipMemcpy(buf, off32.data(), (size_t)(nq + 1) * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && nnz) e = hipMemcpy(buf + o_comp, comps, nnz * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && nnz) e = hipMemcpy(buf + o_val, vals, nnz * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(buf + o_max, 0, 16);
  if (e == hipSuccess)
    e = launch_device_plan(d->view, (const uint32_t*)buf, (const uint32_t*)(buf + o_comp), (const float*)(buf + o_val), nq, query_cut,
                           (uint64_t*)(buf + o_keys), (uint32_t*)(buf + o_max), (uint32_t*)(buf + o_order), d->main.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(d->main.stream);
  if (e == hipSuccess) e = hipMemcpy(order_out, buf + o_order, (size_t)nq * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out3, buf + o_max, 12, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(SGPU_EDEVICE, "device plan failed: %s", hipGetErrorString(e));
  return SGPU_OK;
}

static sgpu_status plan_for(DeviceIndex* d, sgpu_batch* b, uint32_t query_cut, const sgpu_batch_plan** out) {
  for (const auto& pl : b->plans)
    if (pl.query_cut == query_cut) {
      *out = &pl;
      return SGPU_OK;
    }
  if (b->staged) return fail(SGPU_EINVAL, "a staged batch is planned when it is staged");
  sgpu_batch_plan pl;
  sgpu_status st = make_plan(d, b->h_off.data(), b->h_comp.data(), b->h_val.data(), b->nq, query_cut, &pl);
  if (st != SGPU_OK) return st;
  try {
    b->plans.push_back(std::move(pl));
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory planning a query batch");
  }
  *out = &b->plans.back();
  return SGPU_OK;
}

// Which launch sizes take the cooperative variant on their own. Until r05: up to n_cu queries. With the streamed plain
// variants the picture has two windows (r06, 8.8M documents, kernel us per launch, cooperative / plain - profiles/
// r06_small_launches.txt): 144 queries 304 / 326, 176: 327 / 329, 224: 351 / 339, 256: 365 / 340 - one 1024-thread streamed
// workgroup per CU beats the helpers once most CUs own a query; 288: 420 / 453, 352: 448 / 473, 416: 481 / 512, 448: 496 / 502,
// 480: 513 / 507 - past n_cu the plain launch falls back to 512-thread workgroups and the tail help pays again until ~1.75
// queries per CU. SGPU_COOP_MAX_NQ = n keeps the old rule "up to n".
static bool coop_by_size(const DeviceIndex* d, uint32_t nq) {
  const char* v = env_get("SGPU_COOP_MAX_NQ");
  if (v && *v) return nq <= (uint32_t)std::strtoul(v, nullptr, 10);
  return nq <= d->n_cu * 11u / 16u || (nq > d->n_cu && nq <= d->n_cu * 7u / 4u);
}

// Chooses block size, LDS layout and grid for one search pass (caller holds d->mu).
static sgpu_status configure(DeviceIndex* d, Lane* lane, sgpu_batch* b, const sgpu_search_params& sp, uint32_t mode,
                             LaunchArgs* a) {
  if (sp.k == 0) return fail(SGPU_EINVAL, "k must be > 0 (KHeap::new asserts, reference src/utils.rs:23)");
  if (sp.k > b->k_max) return fail(SGPU_EINVAL, "k = %u exceeds the batch's k_max = %u", sp.k, b->k_max);
  if (std::isnan(sp.heap_factor)) return fail(SGPU_EINVAL, "heap_factor is NaN");
  // 512 threads x 2 workgroups per CU for throughput; when the batch has no more queries than
  // CUs, one 1024-thread workgroup per query (twice the scoring lanes) is ~20% faster
  uint32_t NT = hook_u32("SGPU_BLOCK", (mode != MODE_DOTS && b->nq <= d->n_cu) ? 1024 : 512);
  if (NT != 512 && NT != 1024) return fail(SGPU_EINVAL, "SGPU_BLOCK must be 512 or 1024");
  // (1024-thread variants exist for k <= 128 and never for the counted pass: device_types.hpp, variant_built)
  const bool want_counted = mode == MODE_COUNTED || hook_u32("SGPU_VISITED_BITMAP", 0);
  if (NT == 1024 && (want_counted || heap_variant(sp.k) > 2)) NT = 512;
  const uint32_t qn = std::max<uint32_t>(4, (b->max_nnz + 3u) & ~3u);
  const bool searching = mode != MODE_DOTS;
  // lists walked per query. query_cut == 0 walks none: the reference's k_largest_by(0) selects no
  // list and the result is empty (src/inverted_index.rs:187-190); the LDS tables keep one slot.
  const uint32_t cut = mode == MODE_DOTS ? 1u : std::min<uint32_t>(sp.query_cut, qn);
  const uint32_t qc = std::max<uint32_t>(1, cut);
  const uint32_t words = d->view.dim / 32 + 1;   // + the word of the padding sentinel `dim`
  uint32_t items_max = hook_u32("SGPU_ITEMS_MAX", 1024);
  uint32_t dots_cap = 1, sort_nb = 0;
  const sgpu_batch_plan* pl = nullptr;
  if (mode == MODE_DOTS) {
    dots_cap = std::max<uint32_t>(1, d->list_nb[sp.query_cut]);
  } else {
    sgpu_status pst = plan_for(d, b, cut, &pl);
    if (pst != SGPU_OK) return pst;
    dots_cap = pl->dots_cap;
    sort_nb = pl->max_nb;
  }
  // LDS layout, computed in 64 bits (qc * qn reaches 2^32 for 64K-component queries) and checked
  // against the limit before it is narrowed
  const uint64_t lds_limit = std::min<uint32_t>(d->max_lds ? d->max_lds : 65536, 160 * 1024);
  auto up = [](uint64_t x) { return (x + 15ull) & ~15ull; };
  LdsLayout L{};
  uint64_t o = 0;
  // the weights the scoring loop reads come FIRST (LDS byte 0 is the 0.0 slot non-matching components resolve to)
  static_assert(kQscOffset == 0, "the weights come first");
  L.q_sc = (uint32_t)o; o += up(((uint64_t)qn + 1) * 4);   // == kQscOffset: the kernel addresses the weights as byte + constant
  L.q_val = L.q_sc;     // f16 documents: the query's values themselves
  if (d->value_type != SGPU_VAL_F16) {   // fixed-u8 documents: q_sc is a second copy of the weights, scaled by val_scale
    L.q_val = (uint32_t)o;
    o += up(((uint64_t)qn + 1) * 4);
  }
  L.q_comp = (uint32_t)o; o += up((uint64_t)qn * 4);
  L.sel = (uint32_t)o; o += up((6ull * qc + 1) * 4);
  const uint64_t target = hook_u32("SGPU_LDS_TARGET", 160u * 1024u / 2u);   // 2 workgroups per CU
  // what the layout needs besides the row tables and the block dots (the dense lookup table where it may be used)
  const uint64_t rest = up((sp.first_sorted && searching) ? (uint64_t)sort_nb * 2 : 0) + up(2 * (NT / 64 + 1) * 4) +
                        up((uint64_t)heap_variant(sp.k) * 64 * 8) +
                        up(kStateWords * 4) +
                        ((d->comp_width == 2 && d->view.dim <= 65535 && b->max_nnz <= 255) ? up((uint64_t)d->view.dim + 1)
                                                                                           : up((uint64_t)words * 6)) +
                        up(512 * 16 + NT * 12);
  // Row tables of stage 1 (matched rows of every (list, query component) pair): for all of a query's lists when
  // that - and all their block dots - fits next to everything else at 2 workgroups per CU (the benchmark
  // configurations); otherwise the lists are walked in GROUPS of at most qg lists and the tables hold one group
  // (r03 sized them for all lists: query_cut 16 x 96 query components = 29 KB, which cost the dense lookup table
  // - the 0.95-recall operating point ran the packed lookup at 0.55 of peak). Four lists keep all eight
  // wavefronts of stage 1 busy (two per list), so groups do not go below four unless asked (SGPU_LIST_GROUP).
  auto rt_bytes = [&](uint64_t g) { return up(g * qn * 8) + up(g * qn * 2) + up(2ull * g * (qn + 1) * 4); };
  uint32_t qg = qc;
  if (pl && qc > 4 && o + rt_bytes(qc) + rest + (uint64_t)dots_cap * 4 > target) qg = 4;
  qg = std::max<uint32_t>(1, std::min<uint32_t>(qg, hook_u32("SGPU_LIST_GROUP", qc)));
  if (o + rt_bytes(qg) > lds_limit)
    return fail(SGPU_ELIMIT, "query_cut %u x %u query components do not fit the row tables in LDS", qc, qn);
  L.rt_start = (uint32_t)o; o += up((uint64_t)qg * qn * 8);
  L.rt_mid = (uint32_t)o; o += up((uint64_t)qg * qn * 2);
  L.rt_pre = (uint32_t)o; o += up(2ull * qg * (qn + 1) * 4);   // two streams (block-id halves) per list
  // Block dots: all of a query's lists at once when that fits next to everything else at 2 workgroups
  // per CU; otherwise the kernel walks the lists in groups and the area shrinks, down to the largest
  // single list (docs/Guidelines.md:44-70: query_cut 10 over lists of 3600 blocks is 144 KB at once).
  if (pl && (uint64_t)dots_cap * 4 > 24u * 1024u) {
    const uint64_t fit = target > o + rest ? (target - o - rest) / 4 : 0;
    dots_cap = (uint32_t)std::max<uint64_t>(pl->max_list_nb, std::min<uint64_t>(dots_cap, std::max<uint64_t>(fit, 6144)));
  }
  dots_cap = std::min<uint32_t>(dots_cap, hook_u32("SGPU_DOTS_CAP", 0xffffffffu));
  if (pl) dots_cap = std::max(dots_cap, pl->max_list_nb);
  L.dots_cap = dots_cap;
  L.dots = (uint32_t)o; o += up((uint64_t)dots_cap * 4);
  L.order = (uint32_t)o; o += up((sp.first_sorted && searching) ? (uint64_t)sort_nb * 2 : 0);
  L.part = (uint32_t)o; o += up(2 * (NT / 64 + 1) * 4);   // two scan scratch areas, used alternately
  L.heap = (uint32_t)o; o += up((uint64_t)heap_variant(sp.k) * 64 * 8);   // the top-k between replays
  L.st = (uint32_t)o; o += up(kStateWords * 4);   // state words + candidate lists
  // [lookup table | union region (sort keys, item tables)]
  uint64_t sort_bytes = 0;
  if (sp.first_sorted && searching && sort_nb > 1) {
    uint64_t n2 = 1;
    while (n2 < sort_nb) n2 <<= 1;
    sort_bytes = n2 * 8;
  }
  if (o > lds_limit)
    return fail(SGPU_ELIMIT, "query needs %llu bytes of LDS before its lookup table (dots %u blocks) > %llu available",
                (unsigned long long)o, dots_cap, (unsigned long long)lds_limit);
  const uint64_t budget = hook_u32("SGPU_LDS_TARGET", 160u * 1024u / 2u);   // 2 workgroups per CU
  // query lookup table: dense u8 index (1 B per vocabulary id + the padding sentinel) when it is
  // allowed and fits at 2 workgroups per CU, else {bits, rank} per 32 vocabulary ids
  const uint64_t dense_bytes = std::max<uint64_t>(up((uint64_t)d->view.dim + 1), hook_u32("SGPU_DENSE_BYTES", 0)), bitmap_bytes = up((uint64_t)words * 8);
  const bool dense_ok = d->comp_width == 2 && d->view.dim <= 65535 && b->max_nnz <= 255 &&
                        !hook_u32("SGPU_NO_DENSE", 0) && searching;
  const uint64_t split_bits = up((uint64_t)words * 4), split_bytes = split_bits + up((uint64_t)words * 2);
  // the round's item tables shrink (down to 256 items) if that is what keeps 2 workgroups per CU
  // Streamed stage 2 (r06, search_kernel.inc "stage 2 as a stream"): plain search launches (the cooperative variant - small
  // launches - and the counted pass keep the round loop). Its union region holds a RING of item slots, a power of two: the
  // fitting loops below step `items` down by 128 as they always did; for a streamed launch 1024 items mean a ring of 1024
  // slots (20.4 KB where the round loop's tables of 1024 items take 22.5), anything from 512 on one of 512, less 256.
  // SGPU_STREAM=0 (test hook): the round loop.
  const bool coop_wanted = [&] {
    const char* cm = env_get("SGPU_COOP");
    const bool force = cm && !std::strcmp(cm, "force");
    const bool off = (cm && !std::strcmp(cm, "0")) || d->coop_broken;
    return !off && !want_counted && mode == MODE_SEARCH && (force || coop_by_size(d, b->nq)) &&
           variant_built(NT, heap_variant(sp.k), false, true);
  }();
  const bool stream_wanted = !coop_wanted && !want_counted && mode == MODE_SEARCH && hook_u32("SGPU_STREAM", 1) &&
                             variant_built(NT, heap_variant(sp.k), false, false, true);
  const uint32_t ring_cap = [&] {
    uint32_t r = std::min<uint32_t>(1024, std::max<uint32_t>(256, hook_u32("SGPU_RING_MAX", 1024)));
    while (r & (r - 1)) r &= r - 1;
    return r;
  }();
  auto ring_of = [&](uint32_t items) { return std::min<uint32_t>(ring_cap, items >= 1024 ? 1024u : (items >= 512 ? 512u : 256u)); };
  auto uni_for = [&](uint32_t items) {
    if (stream_wanted)   // (+ room for the item tables of a kNN refinement round of at least 128 items, which uses the same region)
      return up(std::max<uint64_t>(std::max<uint64_t>(stream_ring_bytes(ring_of(items)), 128 * 16 + NT * 12), sort_bytes));
    return up(std::max<uint64_t>((uint64_t)items * 16 + NT * 12, sort_bytes));
  };
  const uint64_t smallest_lookup = (d->comp_width == 4) ? split_bytes : bitmap_bytes;
  const uint32_t want_items = items_max;
  if (!hook_get("SGPU_ITEMS_MAX")) {
    const uint32_t want = items_max;
    if (dense_ok)   // the dense table is worth smaller rounds (down to 512 items)
      while (items_max > 512 && o + dense_bytes + uni_for(items_max) > budget) items_max -= 128;
    if (!dense_ok || o + dense_bytes + uni_for(items_max) > budget) {
      items_max = want;
      while (items_max > 256 && o + smallest_lookup + uni_for(items_max) > budget) items_max -= 128;
    }
  }
  const uint64_t min_uni = uni_for(items_max);
  const bool dense = dense_ok && (o + dense_bytes + min_uni <= budget || hook_u32("SGPU_FORCE_DENSE", 0));
  // large vocabularies: bits + 16-bit ranks (6 B per 32 ids) when the packed table (8 B) would
  // cost the second workgroup per CU
  const bool split = !dense && d->comp_width == 4 && b->max_nnz <= 65535 &&
                     (o + bitmap_bytes + min_uni > budget || hook_u32("SGPU_FORCE_SPLIT", 0)) &&
                     !hook_u32("SGPU_NO_SPLIT", 0);
  // hashed {component id, weight} entries (32 KB): ONE LDS read per document component where the dense byte
  // table needs two (profiles/r03_lds_sensitivity.md) - preferred whenever every query of the batch has a
  // collision-free seed, the launch order (which carries the seeds) is in use, and it fits at 2 workgroups
  // per CU (the round's item tables shrink for it as they do for the dense table)
  const uint64_t hash_bytes = (uint64_t)kHashSlots * 8;
  const bool hash_family = d->comp_width == 4 && d->value_type == SGPU_VAL_F16;   // (the hashed kernels: u32 components, f16 values)
  // (measured r03 on the 8.8M-document shape, u16 components: 6.46 ms per launch against the dense byte
  // table's 5.81 - a random 8-byte read costs two bank passes where the byte read costs one and the dense
  // layout's second read is mostly a broadcast of one address; fixed-u8 6.36 against 5.56. The hashed
  // entries therefore serve u32 components, where they replace a byte read PLUS an 8-byte read; the u16
  // hashed families were dropped in r05 - u16 components use the dense table or the packed words.)
  bool hashed = searching && hash_family && pl && pl->hash_ok && !hook_u32("SGPU_NO_LPT", 0) && !hook_u32("SGPU_NO_HASH", 0) &&
                !hook_u32("SGPU_FORCE_SPLIT", 0) && !hook_u32("SGPU_FORCE_DENSE", 0) &&
                (d->comp_width == 4 || !dense || hook_u32("SGPU_FORCE_HASH", 0));
  if (hashed && !hook_get("SGPU_ITEMS_MAX") && !hook_u32("SGPU_FORCE_HASH", 0)) {
    uint32_t im = want_items;
    while (im > 512 && o + hash_bytes + uni_for(im) > budget) im -= 128;
    if (o + hash_bytes + uni_for(im) <= budget) items_max = im;
    else hashed = false;
  } else if (hashed && !hook_u32("SGPU_FORCE_HASH", 0) && o + hash_bytes + uni_for(items_max) > budget) {
    hashed = false;
  }
  const uint32_t lookup = hashed ? LK_HASH : (dense ? LK_DENSE : (split ? LK_SPLIT : LK_PACKED));
  const uint64_t lookup_bytes =
      mode == MODE_DOTS ? 0u : (hashed ? hash_bytes : (dense ? dense_bytes : (split ? split_bytes : bitmap_bytes)));
  L.q_bits = (uint32_t)o;
  L.q_rank = (uint32_t)(o + (split && !hashed ? split_bits : lookup_bytes));
  o += lookup_bytes;
  L.uni = (uint32_t)std::min<uint64_t>(o, 0xffffffffu);
  const uint64_t uni = uni_for(items_max);
  o += uni;
  L.qc = qc;
  L.qn = qn;
  L.qg = qg;
  a->lookup = lookup;
  if (o > lds_limit)
    return fail(SGPU_ELIMIT,
                "query needs %llu bytes of LDS (dots %u blocks, %u-word bitmap, sort %llu B) > %llu available; "
                "lower query_cut / use first_sorted=0 / rebuild with a smaller centroid_fraction",
                (unsigned long long)o, dots_cap, words, (unsigned long long)sort_bytes, (unsigned long long)lds_limit);
  L.total = (uint32_t)o;
  a->L = L;
  a->p.k = sp.k;
  a->p.query_cut = cut;
  a->p.heap_factor = sp.heap_factor;
  a->p.first_sorted = sp.first_sorted != 0;
  a->p.mode = mode == MODE_COUNTED ? (uint32_t)MODE_SEARCH : mode;
  a->p.n_knn = mode == MODE_DOTS ? 0u : sp.n_knn;   // ignored when the index has no graph, as the reference does
  a->counted = want_counted ? 1u : 0u;
  // (a streamed launch: the ring the region was sized for; items_max is then what is left for the item tables of a kNN
  // refinement round in the same region)
  const uint32_t ring = stream_wanted ? ring_of(items_max) : 0u;
  if (stream_wanted) items_max = std::max<uint32_t>(128, std::min<uint32_t>(1024, (uint32_t)((uni - NT * 12) / 16) & ~127u));
  a->p.items_max = items_max;
  a->p.items_init = std::min<uint32_t>(items_max, hook_u32("SGPU_ITEMS_INIT", 128));
  a->p.items_min = std::min<uint32_t>(a->p.items_init, hook_u32("SGPU_ITEMS_MIN", 64));
  a->p.rblocks_max = std::min<uint32_t>(32, std::max<uint32_t>(1, hook_u32("SGPU_RBLOCKS", 8)));   // one mask bit per block
  a->p.target_list = mode == MODE_DOTS ? sp.query_cut : 0;
  a->p.val_scale = d->val_scale;
  a->p.queue_base = b->staged ? b->queue_base : 0u;
  a->value_type = d->fwd_sliced ? (uint32_t)kDevValF16Sliced : d->value_type;
  a->ix = d->view;
  a->comp_width = d->comp_width;
  a->block = NT;
  a->lds_bytes = (uint32_t)o;
  if (env_u32("SGPU_DEBUG", 0))
    std::fprintf(stderr, "sgpu configure: nq %u NT %u lookup %u items_max %u ring %u dots_cap %u lists/group %u of %u uni %llu lds %llu\n", b->nq, NT, lookup,
                 items_max, ring, dots_cap, qg, qc, (unsigned long long)uni, (unsigned long long)o);
  a->stream = lane->stream;
  a->qb.q_off = b->q_off;
  a->qb.q_comp = b->q_comp;
  a->qb.q_val = b->q_val;
  a->qb.nq = b->nq;
  a->qb.k_stride = b->k_max;
  a->qb.out_scores = b->out_scores;
  a->qb.out_ids = b->out_ids;
  a->qb.out_n = b->out_n;
  a->qb.q_order = nullptr;
  a->qb.q_seed = nullptr;
  if (pl && !hook_u32("SGPU_NO_LPT", 0) && b->nq && !b->plan_identity) {
    // the processing order lives in the batch's own device buffer (no allocation on the path)
    if (b->order_cut != cut) {
      if (b->order_cut != 0xffffffffu) HIP_TRY(hipStreamSynchronize(lane->stream));   // the staging copy may be in flight
      std::memcpy(b->h_order, pl->order.data(), (size_t)b->nq * 8);   // [order | hash seeds]
      HIP_TRY(hipMemcpyAsync(b->q_order, b->h_order, (size_t)b->nq * 8, hipMemcpyHostToDevice, lane->stream));
      b->order_cut = cut;
    }
    a->qb.q_order = b->q_order;
    a->qb.q_seed = b->q_order + b->nq;
  }
  a->qb.out_stats = mode != MODE_DOTS ? b->out_stats : nullptr;   // (null for staged batches: no work counters)
  // launch status word: a staged batch's travels back with its rows; a device-resident batch uses the lane's sticky word
  // (queue + 32, zero from lane_init on; read after the rows by batch_fetch / batch_sync)
  a->qb.status = b->staged ? b->status : lane->queue + 32;
  a->qb.done = nullptr;   // (set below for cooperative launches that write their rows to the host arena)
  a->qb.done_seq = 0;
  // cooperative variant wanted? (decided before the occupancy query: it is its own kernel symbol)
  a->coop = CoopView{};
  {
    const char* cm = env_get("SGPU_COOP");
    const bool force = cm && !std::strcmp(cm, "force");
    const bool off = (cm && !std::strcmp(cm, "0")) || d->coop_broken;
    // (measured r03, 8.8M documents: 1 query 133 vs 200 us, 8: 147 vs 297, 64: 316 vs 412, 256: 431 vs 486;
    // from ~1000 queries per launch on the variant's own cost - 6 % slower rounds, idle workgroups kept
    // resident - outweighs what its tail help returns: 780 vs 742 us)
    if (!off && !a->counted && mode == MODE_SEARCH && a->lds_bytes - a->L.uni >= 4096 && (force || coop_by_size(d, b->nq)) &&
        variant_built(NT, heap_variant(sp.k), false, true))   // (k > 256: the plain variant)
      a->coop.enabled = force ? 2u : 1u;
  }
  // streamed stage 2: decided with the LDS layout above (stream_wanted); the cooperative decision just made must agree
  a->streamed = 0;
  a->p.ring = 0;
  if (stream_wanted && !a->coop.enabled && dots_cap <= 65535u && stream_ring_bytes(ring) <= a->lds_bytes - a->L.uni) {
    a->streamed = 1;
    a->p.ring = ring;
    a->p.items_init = std::min<uint32_t>(std::min<uint32_t>(hook_u32("SGPU_ITEMS_INIT", 128), 1024), ring);
    a->p.items_min = std::min<uint32_t>(a->p.items_min, a->p.items_init);
  }
  // occupancy of this kernel variant at this LDS size: queried once, then remembered
  int per_cu = 0;
  {
    const uint64_t key = ((uint64_t)(a->streamed != 0) << 59) | ((uint64_t)(a->coop.enabled != 0) << 58) | ((uint64_t)a->comp_width << 56) | ((uint64_t)a->counted << 55) | ((uint64_t)a->value_type << 52) | ((uint64_t)a->block << 40) | ((uint64_t)a->lookup << 36) |
                         ((uint64_t)heap_variant(a->p.k) << 28) | (uint64_t)(a->lds_bytes >> 4);
    auto it = d->occupancy.find(key);
    if (it == d->occupancy.end()) {
      HIP_TRY(occupancy_search(*a, &per_cu));
      d->occupancy[key] = per_cu;
    } else {
      per_cu = it->second;
    }
  }
  if (per_cu < 1) return fail(SGPU_ELIMIT, "the search kernel does not fit on a CU with %llu bytes of LDS", (unsigned long long)o);
  const uint32_t cap = hook_u32("SGPU_WG_PER_CU", 0);
  if (cap && (uint32_t)per_cu > cap) per_cu = (int)cap;
  uint32_t grid = d->n_cu * (uint32_t)per_cu;
  // Cooperative variant (small launches and their tails; DESIGN.md "Cooperative mode"): every slot of the
  // chip is launched, workgroups without a query help the ones that own one. SGPU_COOP=0 disables it,
  // SGPU_COOP=force enables it for any launch and lets owners go wide without waiting for idle workgroups
  // (the tests: the whole protocol then runs inside big batches too).
  {
    const bool force = a->coop.enabled == 2;
    const uint64_t uni_bytes = a->lds_bytes - a->L.uni;
    if (a->coop.enabled && !force) {
      // How many workgroups a latency-bound launch keeps resident (r04, measured at three operating points of the 8.8M-
      // document collection, profiles/r04_coop_grid.txt): every slot of the chip is too many for one or two queries -
      // 256 helpers attach, load the query, claim from one word and have to leave again before the launch ends: a single
      // query takes 147.6 us with 256 workgroups, 129.0 with 64 - 96 and 129.6 with 48 (the 0.95 / 0.99-recall indexes:
      // 256 -> 228, 361 -> 326 us); two queries are best served by 96 - 128 (157 -> 146 us), eight by 128 - 256 (+-1 %),
      // 32 and more by all of them. Rule: 48 + 32 per query, scaled to the chip's CU count. SGPU_COOP_GRID overrides.
      const uint32_t rule = (uint32_t)(((uint64_t)48 + 32ull * b->nq) * d->n_cu / 256);
      uint32_t cg = hook_u32("SGPU_COOP_GRID", 0xffffffffu);   // unset: the rule; 0: every slot; n: at most n workgroups
      if (cg == 0xffffffffu) cg = std::max<uint32_t>(rule, 8);
      if (cg) grid = std::max<uint32_t>(std::min<uint32_t>(grid, cg), std::min<uint32_t>(b->nq, grid));
    }
    if (a->coop.enabled && grid > 512) {   // the open-round bitmap has 512 bits: the cooperative grid is capped there
      if (env_u32("SGPU_DEBUG", 0)) std::fprintf(stderr, "sgpu coop: grid %u capped at 512 (board size)\n", grid);
      grid = 512;
    }
    if (a->coop.enabled) {
      CoopView& c = a->coop;
      c.max_pos = std::min<uint32_t>(65535u, std::max<uint32_t>(a->L.dots_cap, 1u));
      c.max_cand = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(hook_u32("SGPU_COOP_MAX_CAND", 1024), NT), uni_bytes / 20);
      c.chunk = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(std::max<uint32_t>(hook_u32("SGPU_COOP_CHUNK", 128), 1u), std::min<uint32_t>(NT, 1023u)),
                                             uni_bytes / 16);
      c.chunk_min = std::min<uint32_t>(c.chunk, std::max<uint32_t>(hook_u32("SGPU_COOP_CHUNK_MIN", 4), 1u));
      c.min_items = hook_u32("SGPU_COOP_MIN_ITEMS", 64);
      c.first_reach = std::max<uint32_t>(1, hook_u32("SGPU_COOP_FIRST_REACH", 256));
      c.idle_min = hook_u32("SGPU_COOP_IDLE_MIN", force ? 0 : 8);
      // (r05: an owner goes wide once there is ONE idle workgroup per workgroup still owning a query - until r04 eight, which
      // kept the 64 owners of a 64-query launch on their own although 192 helpers were resident: 324 -> 221 us per launch,
      // 256 queries 442 -> 380 us; gpurun_out r05c, profiles/r05_coop_policy.txt)
      c.idle_ratio = hook_u32("SGPU_COOP_IDLE_RATIO", force ? 0 : 1);
      c.poll_sleep = std::max<uint32_t>(1, hook_u32("SGPU_COOP_POLL", 2));
      c.enabled = 1u;
      const size_t o_slots = 128, o_pos = o_slots + (size_t)grid * kCoopSlotWords * 8,
                   o_cand = o_pos + (size_t)grid * c.max_pos * 8, o_trace = o_cand + (size_t)grid * c.max_cand * 16,
                   total = o_trace + (size_t)grid * 16 * 8;
      if (lane->coop_bytes < total) {
        if (lane->coop) {
          HIP_TRY(hipStreamSynchronize(lane->stream));
          (void)hipFree(lane->coop);
        }
        lane->coop = nullptr;
        lane->coop_bytes = 0;
        if (hipMalloc((void**)&lane->coop, total) != hipSuccess)
          return fail(SGPU_ENOMEM, "hipMalloc of the %zu-byte cooperative board failed", total);
        lane->coop_bytes = total;
        HIP_TRY(hipMemsetAsync(lane->coop, 0, total, lane->stream));
      } else if (lane->coop_bytes && hook_u32("SGPU_COOP_RESET", 0)) {   // (debugging aid: do not trust the kernel's own clean-up)
        HIP_TRY(hipMemsetAsync(lane->coop, 0, o_pos, lane->stream));
      }
      if (env_u32("SGPU_DEBUG", 0))
        std::fprintf(stderr, "sgpu coop: board %p..%p slots@%zu pos@%zu cand@%zu grid %u max_pos %u max_cand %u chunk %u..%u min_items %u idle_min %u idle_ratio %u\n",
                     (void*)lane->coop, (void*)(lane->coop + total), o_slots, o_pos, o_cand, grid, c.max_pos, c.max_cand, c.chunk_min, c.chunk,
                     c.min_items, c.idle_min, c.idle_ratio);
      c.open = (uint64_t*)lane->coop;
      c.counters = (uint32_t*)(lane->coop + 64);
      c.slots = (uint64_t*)(lane->coop + o_slots);
      c.pos_pub = (uint64_t*)(lane->coop + o_pos);
      c.cands = (uint64_t*)(lane->coop + o_cand);
      c.trace = nullptr;
      if (hook_u32("SGPU_COOP_TRACE", 0)) {   // (trace builds) the timeline of this launch: zeroed here, dumped by coop_trace_dump
        c.trace = (uint64_t*)(lane->coop + o_trace);
        HIP_TRY(hipMemsetAsync(lane->coop + o_trace, 0, (size_t)grid * 16 * 8, lane->stream));
        lane->coop_trace_off = o_trace;
        lane->coop_trace_n = grid * 16;
      }
    }
  }
  // latency-bound launches bootstrap the threshold with ONE local round before they go wide
  if (a->coop.enabled && b->nq <= d->n_cu && !hook_get("SGPU_ITEMS_INIT"))
    a->p.items_init = std::min<uint32_t>(a->p.items_max, std::max<uint32_t>(1, hook_u32("SGPU_COOP_ITEMS_INIT", 128)));
  if (!a->coop.enabled) grid = std::max<uint32_t>(1, std::min<uint32_t>(grid, b->nq));
  // SGPU_GRID_SPARE=n (a test hook): a device-planned chunk launches n workgroups fewer than the chip holds. The idea - the
  // plan kernels of the NEXT chunk run in the free slots instead of waiting for this launch's tail - did not survive its
  // measurements (profiles/r06_entry_point_final.txt): with one free slot per XCD the next chunk's plan still completes
  // only when this launch's workgroups start to leave (events on the two streams, no profiler attached); with 32 and more
  // it completes at once, but the next search launch then shares the chip with this one from the start and the call
  // gets slower (6.15 against 5.95 ms). No slots are left free by default.
  if (!a->coop.enabled && b->staged && b->device_plan_cut != 0xffffffffu && grid == d->n_cu * (uint32_t)per_cu) {
    const uint32_t spare = hook_u32("SGPU_GRID_SPARE", 0);
    if (spare && grid > 2 * spare) grid -= spare;
  }
  a->grid = grid;
  // A cooperative launch that writes its rows straight into the pinned host arena also tells the host when the LAST
  // query's rows are there (BatchView::done): the call returns them while the launch winds down (helpers leaving, the
  // board being cleared), and the next call's launch queues up behind it (r04; SGPU_EARLY_DONE=0: wait for the launch).
  b->done_seq = 0;
  if (a->coop.enabled && b->staged && b->direct_out && b->status && env_u32("SGPU_EARLY_DONE", 1)) {
    lane->done_seq = lane->done_seq == 0xffffffffu ? 1u : lane->done_seq + 1u;
    b->done_seq = lane->done_seq;
    a->qb.done = b->status + 1;
    a->qb.done_seq = b->done_seq;
  }
  // visited bitmaps: one per resident workgroup (counted pass)
  a->bitmaps = nullptr;
  if (a->counted) {
    if (lane->bitmaps_slots < grid) {
      if (lane->bitmaps) {
        HIP_TRY(hipStreamSynchronize(lane->stream));
        (void)hipFree(lane->bitmaps);
      }
      lane->bitmaps = nullptr;
      lane->bitmaps_slots = 0;
      const uint32_t slots = std::max<uint32_t>(grid, d->n_cu * (uint32_t)per_cu);
      const size_t bytes = (size_t)slots * std::max<uint32_t>(d->view.n_bitmap_words, 1) * 4;
      if (hipMalloc((void**)&lane->bitmaps, bytes) != hipSuccess)
        return fail(SGPU_ENOMEM, "hipMalloc of %zu bytes of visited bitmaps failed", bytes);
      HIP_TRY(hipMemsetAsync(lane->bitmaps, 0, bytes, lane->stream));
      lane->bitmaps_slots = slots;
    }
    a->bitmaps = lane->bitmaps;
  }
  a->queue = b->staged ? b->queue_dev : lane->queue;
  lane->coop_last = a->coop.enabled != 0;
  return SGPU_OK;
}

static void drain_events(Lane* l) {   // stream must be idle
  for (int i = 0; i < l->ev_pending; ++i) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, l->ev0[i], l->ev1[i]) == hipSuccess) {
      l->sum_ms += ms;
      l->n_timed += 1;
    }
  }
  l->ev_pending = 0;
}

// A cooperative launch that met a protocol error (a bounded device-side wait that gave up, control words of the
// wrong round) says so in its status word / the board's sticky word. EVERY cooperative launch is checked (r03 did
// so only under SGPU_COOP_CHECK): the call fails loudly, the sticky word is cleared so that later calls are
// judged on their own, and the variant stays off for this replica from then on (plain launches carry on).
static sgpu_status coop_report(DeviceIndex* d, Lane* lane, uint32_t code) {
  if (!code) return SGPU_OK;
  if (code >= 16) {   // streamed stage 2: a bounded wait of a role gave up (16 replayer, 17 feeder, 18 scorer)
    (void)hipStreamSynchronize(lane->stream);
    return fail(SGPU_EDEVICE, "search kernel (streamed stage 2): a bounded wait gave up (code %u)", code);
  }
  if (lane->coop) (void)hipMemsetAsync(lane->coop + 64 + 12, 0, 4, lane->stream);
  (void)hipStreamSynchronize(lane->stream);
  if (!d->coop_broken) std::fprintf(stderr, "seismic_hip: cooperative search kernel reported protocol error %u on device %d; "
                                            "the cooperative variant is switched off for this index replica\n", code, d->device);
  d->coop_broken = true;
  return fail(SGPU_EDEVICE, "cooperative search kernel: protocol wait gave up (code %u)", code);
}
static sgpu_status coop_check_board(DeviceIndex* d, Lane* lane) {   // device-resident batches (not the latency path)
  uint32_t flag = 0;
  HIP_TRY(hipMemcpy(&flag, lane->queue + 32, 4, hipMemcpyDeviceToHost));   // the lane's sticky status word (streamed variants)
  if (flag) {
    (void)hipMemset(lane->queue + 32, 0, 4);
    return coop_report(d, lane, flag);
  }
  if (!lane->coop || !lane->coop_last) return SGPU_OK;
  HIP_TRY(hipMemcpy(&flag, lane->coop + 64 + 12, 4, hipMemcpyDeviceToHost));
  return coop_report(d, lane, flag);
}

// Enqueues one pass on `lane` (the index's main lane when null).
sgpu_status batch_run(DeviceIndex* d, Lane* lane, sgpu_batch* b, const sgpu_search_params& sp, uint32_t mode,
                      int sync, sgpu_launch_stats* stats) {
  if (!d) return fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload)");
  if (!b || b->owner != d) return fail(SGPU_EINVAL, "batch does not belong to this index replica");
  if (!lane) lane = &d->main;
  HIP_TRY(hipSetDevice(d->device));
  if (b->nq == 0) {
    if (stats) *stats = sgpu_launch_stats{};
    return SGPU_OK;
  }
  env_refresh();
  {
    // configuration touches state shared by the lanes (occupancy cache, bitmaps); a main-lane
    // launch also owns the main lane's event ring
    std::lock_guard<std::mutex> lock(d->mu);
    LaunchArgs a{};
    sgpu_status st = configure(d, lane, b, sp, mode, &a);
    if (st != SGPU_OK) return st;
    if (lane->ev_pending == (int)lane->ev0.size()) {
      HIP_TRY(hipStreamSynchronize(lane->stream));
      drain_events(lane);
    }
    HIP_TRY(hipMemsetAsync(lane->queue, 0, 4, lane->stream));
    if (a.qb.out_stats) HIP_TRY(hipMemsetAsync(b->out_stats, 0, (size_t)b->nq * STATS_WORDS * 4, lane->stream));
    const int e = lane->ev_pending++;
    HIP_TRY(hipEventRecord(lane->ev0[e], lane->stream));
    HIP_TRY(launch_search(a));
    HIP_TRY(hipEventRecord(lane->ev1[e], lane->stream));
    lane->last.n_queries = b->nq;
    lane->last.grid = a.grid;
    lane->last.block = a.block;
    lane->last.lds_bytes = a.lds_bytes;
  }
  if (sync) {
    HIP_TRY(hipStreamSynchronize(lane->stream));
    std::lock_guard<std::mutex> lock(d->mu);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, lane->ev0[lane->ev_pending - 1], lane->ev1[lane->ev_pending - 1]));
    drain_events(lane);
    lane->last.kernel_ms = ms;
    if (stats) *stats = lane->last;
  }
  return SGPU_OK;
}

// Waits for everything enqueued on the main lane; stats->kernel_ms = mean kernel duration since the last sync.
sgpu_status batch_sync(DeviceIndex* d, sgpu_launch_stats* stats) {
  if (!d) return fail(SGPU_EDEVICE, "index is not uploaded to a device");
  Lane* l = &d->main;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipStreamSynchronize(l->stream));
  std::lock_guard<std::mutex> lock(d->mu);
  drain_events(l);
  if (stats) {
    *stats = l->last;
    stats->kernel_ms = l->n_timed ? (float)(l->sum_ms / l->n_timed) : 0.0f;
  }
  l->sum_ms = 0;
  l->n_timed = 0;
  return SGPU_OK;
}

sgpu_status batch_fetch(DeviceIndex* d, Lane* lane, sgpu_batch* b, uint32_t k, float* out_scores, uint64_t* out_ids,
                        uint32_t* out_n) {
  if (!d || !b) return fail(SGPU_EINVAL, "null index/batch");
  if (k == 0 || k > b->k_max) return fail(SGPU_EINVAL, "k out of range for this batch");
  if (!lane) lane = &d->main;
  HIP_TRY(hipSetDevice(d->device));
  if (b->nq == 0) {
    HIP_TRY(hipStreamSynchronize(lane->stream));
    return SGPU_OK;
  }
  if (k == b->k_max) {
    HIP_TRY(hipMemcpyAsync(out_scores, b->out_scores, (size_t)b->nq * k * 4, hipMemcpyDeviceToHost, lane->stream));
    HIP_TRY(hipMemcpyAsync(out_ids, b->out_ids, (size_t)b->nq * k * 8, hipMemcpyDeviceToHost, lane->stream));
  } else {
    HIP_TRY(hipMemcpy2DAsync(out_scores, (size_t)k * 4, b->out_scores, (size_t)b->k_max * 4, (size_t)k * 4, b->nq,
                             hipMemcpyDeviceToHost, lane->stream));
    HIP_TRY(hipMemcpy2DAsync(out_ids, (size_t)k * 8, b->out_ids, (size_t)b->k_max * 8, (size_t)k * 8, b->nq,
                             hipMemcpyDeviceToHost, lane->stream));
  }
  HIP_TRY(hipMemcpyAsync(out_n, b->out_n, (size_t)b->nq * 4, hipMemcpyDeviceToHost, lane->stream));
  HIP_TRY(hipStreamSynchronize(lane->stream));
  return coop_check_board(d, lane);
}

sgpu_status batch_fetch_stats(DeviceIndex* d, sgpu_batch* b, uint32_t* out) {
  if (!d || !b || !out) return fail(SGPU_EINVAL, "null index/batch/out");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipStreamSynchronize(d->main.stream));
  if (b->nq) HIP_TRY(hipMemcpy(out, b->out_stats, (size_t)b->nq * STATS_WORDS * 4, hipMemcpyDeviceToHost));
  return SGPU_OK;
}

// (trace builds) the event times of the last traced cooperative launch of any pool lane of replica d
uint32_t coop_trace_dump(DeviceIndex* d, uint64_t* out, uint32_t cap) {
  if (!d) return 0;
  (void)hipSetDevice(d->device);
  Lane* lanes[DeviceIndex::kPool + 1];
  int nl = 0;
  lanes[nl++] = &d->main;
  for (Lane& l : d->pool) lanes[nl++] = &l;
  for (int i = 0; i < nl; ++i) {
    Lane* l = lanes[i];
    if (!l->coop || !l->coop_trace_n) continue;
    (void)hipStreamSynchronize(l->stream);
    const uint32_t n = std::min<uint32_t>(cap, l->coop_trace_n);
    if (hipMemcpy(out, l->coop + l->coop_trace_off, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
  }
  return 0;
}

// ---- staged batches: the lean path behind sgpu_search / sgpu_batch_search ----------------------
static inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

// Validates, plans, stages and launches one search of `nq` queries on `lane`: one H2D, the kernel, one
// D2H, all enqueued; staged_finish waits and hands the rows out. *slot is the lane's recycled batch.
// (q_base: the index of the first query in the caller's batch, for error messages.)
sgpu_status staged_launch(DeviceIndex* d, Lane* lane, uint64_t dim, const uint64_t* q_off, const uint32_t* comps,
                          const float* vals, uint32_t nq, uint32_t q_base, const sgpu_search_params& sp, sgpu_batch** slot,
                          uint32_t followed) {
  if (!d) return fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload)");
  if (sp.k == 0) return fail(SGPU_EINVAL, "k must be > 0 (KHeap::new asserts, reference src/utils.rs:23)");
  if (sp.k > 1024) return fail(SGPU_ELIMIT, "k = %u exceeds the heap limit of 1024", sp.k);
  uint32_t max_nnz = 0;
  PhaseClock pc;
  env_refresh();
  // The offsets first (sizes, max_nnz); the components of a chunk whose plan the host computes right away - make_plan
  // indexes the index's arrays with them - and those of a device-planned chunk AFTER its H2D copy is enqueued: the pass
  // over the components (~50 us per 5000 queries) then runs while the copy does, ahead of the plan kernels and the search.
  if (!q_off || q_off[0] != 0) return fail(SGPU_EINVAL, "q_off[0] must be 0");
  sgpu_status st = validate_query_offsets(q_off, nq, q_base, &max_nnz);
  if (st != SGPU_OK) return st;
  if (q_off[nq] && (!comps || !vals)) return fail(SGPU_EINVAL, "null query arrays");
  bool validated = false;
  HIP_TRY(hipSetDevice(d->device));
  const uint64_t nnz = q_off[nq];
  const uint32_t k = sp.k;
  // arena: [work counter 16 B | fixed status 16 B | q_off | q_comp | q_val | order | status 16 B]  ->  [status | out_n | out_scores | out_ids]
  // (the trailing status word goes down zeroed with the input and comes back with the rows in one D2H copy; a launch that
  // writes its rows straight into the pinned host arena uses the FIXED status / done words instead: their address does not
  // move with nq and nnz, so a launch that is still winding down when the lane's next call stages its input - early done -
  // can only ever touch those two words, never the next call's queries)
  const size_t o_fix = 16, o_off = 32, o_comp = o_off + al16((size_t)(nq + 1) * 4), o_val = o_comp + al16(nnz * 4),
               o_order = o_val + al16(nnz * 4), o_status = o_order + al16((size_t)nq * 8),   // [order | hash seeds]
               in_bytes = o_status + 16;
  const size_t r_n = in_bytes, r_sc = r_n + al16((size_t)nq * 4), r_id = r_sc + al16((size_t)nq * k * 4),
               total = r_id + al16((size_t)nq * k * 8);
  sgpu_batch* b = *slot;
  if (b && (!b->staged || b->owner != d || b->arena_cap < total)) {
    HIP_TRY(hipStreamSynchronize(lane->stream));
    batch_free(b);
    b = nullptr;
    *slot = nullptr;
  }
  if (!b) {
    b = new (std::nothrow) sgpu_batch();
    if (!b) return fail(SGPU_ENOMEM, "out of host memory");
    b->staged = true;
    b->device = d->device;
    b->owner = d;
    // twice the need: a lane that has served one chunk of a call cut in four also holds a chunk of a call cut in two (the
    // number of chunks follows the load of the replica, abi.cpp) - no pinned reallocation in the middle of a stream of calls
    // (2.5 x: chunks of one call differ by a few per cent in components; the slack is bounded - the arena exists twice, in
    // HBM and as pinned host memory, per lane: ADVICE r05)
    b->arena_cap = std::max<size_t>(std::min<size_t>(total * 5 / 2, total + ((size_t)64 << 20)), 1 << 16);
    if (hipMalloc((void**)&b->arena_dev, b->arena_cap) != hipSuccess ||
        hipHostMalloc((void**)&b->arena_host, b->arena_cap, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {   // (fine-grained by request, not by the runtime's default: the kernel's rows and done word must be visible as they land)
      batch_free(b);
      return fail(SGPU_ENOMEM, "allocation of a %zu-byte staging arena failed", b->arena_cap);
    }
    if (hipHostGetDevicePointer((void**)&b->arena_host_dev, b->arena_host, 0) != hipSuccess) b->arena_host_dev = nullptr;
    *slot = b;
    if (hook_raw("SGPU_DEBUG_ALLOC")) std::fprintf(stderr, "sgpu alloc: arena %p..%p\n", (void*)b->arena_dev, (void*)(b->arena_dev + b->arena_cap));
  }
  b->nq = nq;
  b->k_max = k;
  b->max_nnz = max_nnz;
  b->in_bytes = in_bytes;
  b->out_off = r_n;
  b->out_bytes = total - r_n;
  if (nq == 0) {
    b->status_off = o_status;
    return SGPU_OK;
  }
  const uint32_t direct_max = hook_u32("SGPU_DIRECT_OUT_MAX", 16);   // (read per call: the knob cache is refreshed per call)
  const uint32_t qn = std::max<uint32_t>(4, (max_nnz + 3u) & ~3u);
  const uint32_t cut = std::min<uint32_t>(sp.query_cut, qn);
  try {
    b->plans.clear();
    b->plans.emplace_back();
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory planning a query batch");
  }
  // The launch plan: on the DEVICE for chunks of 256 ... 16384 queries once a first chunk has told what this index and
  // query_cut need (r06, plan_kernel.hip: the calling thread no longer spends ~0.2 us per query on it - a third of a
  // one-thread call); on the host for small chunks (microseconds), for the first chunk, for a sorted first list (its
  // sort buffer must hold the chunk's largest first list: the host plan knows it exactly), for query_cut > 16 and for the
  // hashed lookup of u32 indexes (per-query seeds). SGPU_DEVICE_PLAN=0: always the host.
  b->device_plan_cut = 0xffffffffu;
  b->followed = followed != 0;
  b->plan_identity = false;
  {
    const bool hash_family = d->comp_width == 4 && d->value_type == SGPU_VAL_F16 && d->view.dim < (1u << 24);
    uint32_t seen = 0;
    if (nq >= kDevicePlanMinQueries && nq <= kDevicePlanMaxQueries && cut >= 1 && cut <= kDevicePlanCutMax && !sp.first_sorted &&
        !hash_family && env_u32("SGPU_DEVICE_PLAN", 1) && !hook_u32("SGPU_NO_LPT", 0) && !hook_u32("SGPU_AFFINITY_CLASSES", 0)) {
      std::lock_guard<std::mutex> lock(d->mu);
      auto it = d->plan_dots_seen.find(cut);
      if (it != d->plan_dots_seen.end()) seen = it->second;
    }
    if (seen && nq <= direct_max) seen = 0;   // (a call that small never copies its queries down: kDevicePlanMinQueries is far above, this is for the knob)
    if (!seen) {
      st = validate_queries(dim, q_off, comps, vals, nq, &max_nnz, q_base);
      if (st != SGPU_OK) return st;
      validated = true;
    }
    if (seen) {
      sgpu_batch_plan& pl = b->plans.back();
      pl.query_cut = cut;
      pl.dots_cap = std::max(seen, std::max<uint32_t>(d->max_nb, 1));
      pl.max_nb = 0;
      pl.max_list_nb = std::max<uint32_t>(d->max_nb, 1);   // (no list of the index has more blocks: a bound, not the chunk's maximum)
      pl.hash_ok = false;
      pl.order.clear();
      b->device_plan_cut = cut;
      // The chunk's own call sends another chunk right behind it: the order in which this one takes its queries decides
      // how long ITS tail is, and the next chunk's workgroups fill that tail whatever its length - the plan kernels (56 us
      // ahead of the first search workgroup of a call) are skipped, the queries are taken in input order (q_order = null),
      // and the LDS need of the index keeps coming from the chunks that are planned (SGPU_PLAN_IDENTITY=0, a test hook).
      b->plan_identity = followed == 2 && hook_u32("SGPU_PLAN_IDENTITY", 1) != 0;
    } else {
      st = make_plan(d, q_off, comps, vals, nq, cut, &b->plans.back());
      if (st != SGPU_OK) return st;
      std::lock_guard<std::mutex> lock(d->mu);
      uint32_t& v = d->plan_dots_seen[cut];
      v = std::max(v, b->plans.back().dots_cap);
    }
  }
  pc.lap(0);
  uint8_t* hs = b->arena_host;
  std::memset(hs, 0, 16);
  uint32_t* h32 = (uint32_t*)(hs + o_off);
  for (uint32_t q = 0; q <= nq; ++q) h32[q] = (uint32_t)q_off[q];
  if (nnz) {
    std::memcpy(hs + o_comp, comps, nnz * 4);
    std::memcpy(hs + o_val, vals, nnz * 4);
  }
  if (b->device_plan_cut == 0xffffffffu) std::memcpy(hs + o_order, b->plans.back().order.data(), (size_t)nq * 8);
  std::memset(hs + o_status, 0, 16);
  std::memset(hs + o_fix, 0, 16);
  // A latency-bound call (a handful of queries) has the kernel write its few result rows straight into the pinned
  // host arena (mapped, fine-grained: posted writes over PCIe, visible once the stream is done): no D2H copy to
  // enqueue, none to wait for. Larger calls keep the device-side slab and one D2H. Since r04 such a call also lets the
  // kernel READ its queries (a few hundred bytes) from that arena: no H2D copy to enqueue (5 us of host time and a
  // copy command ahead of the kernel) - which leaves the work counter: it is not zeroed per launch but runs on
  // (KParams::queue_base; Lane::queue_pos mirrors it on the host).
  const uint32_t direct_in_on = env_u32("SGPU_DIRECT_IN", 1);
  b->direct_out = b->arena_host_dev != nullptr && nq <= direct_max;
  b->direct_in = b->direct_out && direct_in_on != 0;
  uint8_t* in_base = b->direct_in ? b->arena_host_dev : b->arena_dev;
  b->queue_dev = b->direct_in ? lane->queue + 16 : (uint32_t*)b->arena_dev;
  b->queue_base = 0;
  b->q_off = (uint32_t*)(in_base + o_off);
  b->q_comp = (uint32_t*)(in_base + o_comp);
  b->q_val = (float*)(in_base + o_val);
  b->q_order = (uint32_t*)(in_base + o_order);
  b->order_cut = cut;
  uint8_t* out_base = b->direct_out ? b->arena_host_dev : b->arena_dev;
  b->out_n = (uint32_t*)(out_base + r_n);
  b->out_scores = (float*)(out_base + r_sc);
  b->out_ids = (uint64_t*)(out_base + r_id);
  b->out_stats = nullptr;
  b->status_off = b->direct_out ? o_fix : o_status;
  b->status = (uint32_t*)(out_base + b->status_off);
  pc.lap(1);
  if (b->direct_in) {
    if (lane->queue_dirty) {   // first use of the lane, or a launch on it failed: the counter's value is not known
      HIP_TRY(hipMemsetAsync(lane->queue + 16, 0, 4, lane->stream));
      lane->queue_pos = 0;
      lane->queue_dirty = false;
    }
    b->queue_base = lane->queue_pos;
  } else {
    HIP_TRY(hipMemcpyAsync(b->arena_dev, hs, in_bytes, hipMemcpyHostToDevice, lane->stream));
    if (!validated) {   // (the copy is on its way: nothing that reads the components has been enqueued yet)
      st = validate_queries(dim, q_off, comps, vals, nq, &max_nnz, q_base);
      if (st != SGPU_OK) {
        const std::string msg = last_error();
        (void)hipStreamSynchronize(lane->stream);
        last_error() = msg;
        return st;
      }
      validated = true;
    }
    if (b->device_plan_cut != 0xffffffffu && !b->plan_identity) {
      // order -> the arena's order region; the maxima -> words 1 - 3 of the status block (zeroed by the copy above, they
      // come back with the rows); the sort keys borrow the output region, which the search kernel overwrites afterwards
      // (16 bytes per query at least: room for the <= 2 nq keys)
      HIP_TRY(launch_device_plan(d->view, b->q_off, b->q_comp, b->q_val, nq, cut, (uint64_t*)(b->arena_dev + r_n),
                                 (uint32_t*)(b->arena_dev + o_status) + 1, b->q_order, lane->stream));
    }
  }
  pc.lap(2);
  // from here on a failure waits for the stream: the lane (and its pinned arena) goes back to the pool
  hipError_t he = hipSuccess;
  {
    std::lock_guard<std::mutex> lock(d->mu);   // the occupancy cache is shared by the lanes
    LaunchArgs a{};
    st = configure(d, lane, b, sp, MODE_SEARCH, &a);
    if (st == SGPU_OK) he = launch_search(a);
    if (b->direct_in) {
      if (st == SGPU_OK && he == hipSuccess) lane->queue_pos += nq + a.grid;   // the tickets this launch takes
      else lane->queue_dirty = true;
    }
  }
  pc.lap(3);
  if (st == SGPU_OK && he == hipSuccess && !b->direct_out)
    he = hipMemcpyAsync(hs + o_status, b->arena_dev + o_status, 16 + b->out_bytes, hipMemcpyDeviceToHost, lane->stream);
  pc.lap(4);
  if (st != SGPU_OK || he != hipSuccess) {
    (void)hipStreamSynchronize(lane->stream);
    if (st == SGPU_OK) st = fail(SGPU_EDEVICE, "launch of the search failed: %s", hipGetErrorString(he));
    return st;
  }
  return SGPU_OK;
}

// Waits for the lane's staged search and copies the rows out (nq x k slabs, row q padded past out_n[q]).
sgpu_status staged_finish(DeviceIndex* d, Lane* lane, sgpu_batch* b, float* out_scores, uint64_t* out_ids, uint32_t* out_n) {
  HIP_TRY(hipSetDevice(d->device));
  PhaseClock pc;
  {
    const volatile uint32_t* done = (b && b->done_seq) ? (const volatile uint32_t*)(b->arena_host + b->status_off + 4) : nullptr;
    const hipError_t we = wait_lane(lane->stream, b ? b->nq : 0, done, b ? b->done_seq : 0u);
    if (we != hipSuccess) lane->queue_dirty = true;   // (the running work counter of direct-in calls is no longer known)
    HIP_TRY(we);
  }
  pc.lap(5);
  if (!b || b->nq == 0) return SGPU_OK;
  {   // the launch's status word came back with the rows (cooperative protocol errors, bounded waits of the streamed variant)
    const sgpu_status cs = coop_report(d, lane, *(const volatile uint32_t*)(b->arena_host + b->status_off));
    if (cs != SGPU_OK) {
      lane->queue_dirty = true;
      return cs;
    }
  }
  if (b->device_plan_cut != 0xffffffffu) {   // what this chunk's queries needed: the next chunks' LDS layouts are sized for it
    const uint32_t need = ((const volatile uint32_t*)(b->arena_host + b->status_off))[1];
    std::lock_guard<std::mutex> lock(d->mu);
    uint32_t& v = d->plan_dots_seen[b->device_plan_cut];
    v = std::max(v, need);
  }
  const size_t nq = b->nq, k = b->k_max;
  const uint8_t* r = b->arena_host + b->out_off;
  std::memcpy(out_n, r, nq * 4);
  std::memcpy(out_scores, r + al16(nq * 4), nq * k * 4);
  std::memcpy(out_ids, r + al16(nq * 4) + al16(nq * k * 4), nq * k * 8);
  pc.lap(6);
  return SGPU_OK;
}

// sgpu_summary_distances: one-query batch in MODE_DOTS; dots come back through out_scores.
sgpu_status summary_distances(DeviceIndex* d, const HostIndex& h, uint32_t list, const uint32_t* comps,
                              const float* vals, uint32_t nnz, float* out_dots, uint32_t* out_nb) {
  if (!d) return fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload)");
  if (list >= h.dim) return fail(SGPU_EINVAL, "list %u >= dim", list);
  const uint32_t nb = (uint32_t)(h.list_block_start[list + 1] - h.list_block_start[list]);
  *out_nb = nb;
  if (nb == 0 || nnz == 0) {
    for (uint32_t i = 0; i < nb; ++i) out_dots[i] = 0.0f;
    return SGPU_OK;
  }
  // MODE_DOTS runs stages 0-1 of the search kernel for one query, aimed at `list`
  // (KParams::target_list, carried here in query_cut), and dumps the accumulators.
  const uint64_t q_off[2] = {0, nnz};
  sgpu_batch* b = nullptr;
  sgpu_status st = batch_create(d, &d->main, h.dim, q_off, comps, vals, 1, 1, &b);
  if (st != SGPU_OK) return st;
  sgpu_search_params sp{};
  sp.k = 1;
  sp.query_cut = list;   // MODE_DOTS: carries the target list id
  sp.heap_factor = 0;
  st = batch_run(d, nullptr, b, sp, MODE_DOTS, 1, nullptr);
  if (st == SGPU_OK) {
    if (hipMemcpy(out_dots, b->out_scores, (size_t)nb * 4, hipMemcpyDeviceToHost) != hipSuccess)
      st = fail(SGPU_EDEVICE, "hipMemcpy of summary dots failed");
  }
  batch_free(b);
  return st;
}


// (Re)attaches a kNN graph to a resident index.
sgpu_status device_index_set_knn(DeviceIndex* d, const std::vector<uint32_t>& knn, uint32_t knn_dim) {
  if (!d) return SGPU_OK;
  std::lock_guard<std::mutex> lock(d->mu);
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipDeviceSynchronize());
  if (d->view.knn) {   // drop the previous graph's device copy
    auto it = std::find_if(d->allocs.begin(), d->allocs.end(), [&](const Alloc& a) { return a.p == (void*)d->view.knn; });
    if (it != d->allocs.end()) d->allocs.erase(it);
    (void)hipFree((void*)d->view.knn);
    d->bytes -= std::min<uint64_t>(d->bytes, d->view.knn_total * 4);
  }
  d->view.knn = nullptr;
  d->view.knn_total = 0;
  d->view.knn_dim = 0;
  if (knn.empty() || !knn_dim) return SGPU_OK;
  sgpu_status st = dev_copy(d, knn.data(), knn.size(), &d->view.knn);
  if (st != SGPU_OK) return st;
  d->view.knn_total = knn.size();
  d->view.knn_dim = knn_dim;
  return SGPU_OK;
}

// Knn::new (reference src/inverted_index.rs:448-500): every document is used as a query with
// k = nknn + 1, query_cut = 10, heap_factor = 0.7, first_sorted = false; itself is removed, the
// first nknn remaining results are its neighbours; lists are flattened in document order (a
// document with fewer results contributes fewer ids, exactly as the reference's flatten does).
// The N_docs searches run as batches through the same GPU kernel.
sgpu_status build_knn_on_device(DeviceIndex* d, HostIndex& h, uint32_t nknn) {
  if (!d) return fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload)");
  if (nknn == 0 || nknn + 1 > 1024) return fail(SGPU_EINVAL, "nknn must be in 1..1023");
  const uint32_t k = nknn + 1;
  const uint64_t chunk = hook_u32("SGPU_KNN_CHUNK", 32768);
  std::vector<uint32_t> out;
  out.reserve((size_t)h.n_docs * nknn);
  sgpu_search_params sp{};
  sp.k = k;
  sp.query_cut = 10;
  sp.heap_factor = 0.7f;
  sp.n_knn = 0;
  sp.first_sorted = 0;
  // make sure no stale graph is used while building
  sgpu_status st = device_index_set_knn(d, {}, 0);
  if (st != SGPU_OK) return st;
  std::vector<uint64_t> q_off;
  std::vector<uint32_t> q_comp;
  std::vector<float> q_val, sc;
  std::vector<uint64_t> ids;
  std::vector<uint32_t> n;
  for (uint64_t d0 = 0; d0 < h.n_docs; d0 += chunk) {
    const uint64_t d1 = std::min<uint64_t>(h.n_docs, d0 + chunk);
    const uint32_t nq = (uint32_t)(d1 - d0);
    const uint64_t e0 = h.fwd_offsets[d0], e1 = h.fwd_offsets[d1];
    q_off.resize(nq + 1);
    q_comp.resize(e1 - e0);
    q_val.resize(e1 - e0);
    for (uint32_t q = 0; q <= nq; ++q) q_off[q] = h.fwd_offsets[d0 + q] - e0;
    for (uint64_t i = e0; i < e1; ++i) {
      q_comp[i - e0] = h.comp(i);
      q_val[i - e0] = h.val(i);
    }
    sgpu_batch* b = nullptr;
    st = batch_create(d, &d->main, h.dim, q_off.data(), q_comp.data(), q_val.data(), nq, k, &b);
    if (st != SGPU_OK) return st;
    st = batch_run(d, nullptr, b, sp, MODE_SEARCH, 1, nullptr);
    sc.resize((size_t)nq * k);
    ids.resize((size_t)nq * k);
    n.resize(nq);
    if (st == SGPU_OK) st = batch_fetch(d, nullptr, b, k, sc.data(), ids.data(), n.data());
    batch_free(b);
    if (st != SGPU_OK) return st;
    for (uint32_t q = 0; q < nq; ++q) {
      uint32_t taken = 0;
      for (uint32_t i = 0; i < n[q] && taken < nknn; ++i) {
        const uint64_t id = ids[(size_t)q * k + i];
        if (id == d0 + q) continue;   // remove the document itself
        out.push_back((uint32_t)id);
        ++taken;
      }
    }
  }
  h.knn.swap(out);
  h.knn_dim = nknn;
  return device_index_set_knn(d, h.knn, h.knn_dim);
}

}  // namespace sgpu
