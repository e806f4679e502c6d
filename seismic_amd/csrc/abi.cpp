// abi.cpp — the extern "C" surface declared in include/seismic_hip.h.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "host_index.hpp"

struct sgpu_batch;

namespace sgpu {
// device_index.hip
struct Lane;
sgpu_status device_index_upload(const HostIndex& h, int device, DeviceIndex** out);
sgpu_status device_index_clone(const DeviceIndex* src, int device, DeviceIndex** out);
void device_index_free(DeviceIndex* d);
uint64_t device_index_bytes(const DeviceIndex* d);
Lane* lane_acquire(DeviceIndex* d);
Lane* lane_try_acquire(DeviceIndex* d);
bool call_enter(DeviceIndex* d);
void call_exit(DeviceIndex* d);
uint32_t coop_auto_max_queries(const DeviceIndex* d);
void lane_release(DeviceIndex* d, Lane* l);
Lane* lane_main(DeviceIndex* d);
sgpu_batch** lane_scratch(Lane* l);
sgpu_status batch_create(DeviceIndex* d, Lane* lane, uint64_t dim, const uint64_t* q_off, const uint32_t* comps,
                         const float* vals, uint32_t nq, uint32_t k_max, sgpu_batch** out);
void batch_free(sgpu_batch* b);
sgpu_status batch_run(DeviceIndex* d, Lane* lane, sgpu_batch* b, const sgpu_search_params& sp, uint32_t mode,
                      int sync, sgpu_launch_stats* stats);
sgpu_status batch_sync(DeviceIndex* d, sgpu_launch_stats* stats);
sgpu_status batch_fetch(DeviceIndex* d, Lane* lane, sgpu_batch* b, uint32_t k, float* out_scores, uint64_t* out_ids,
                        uint32_t* out_n);
sgpu_status batch_fetch_stats(DeviceIndex* d, sgpu_batch* b, uint32_t* out);
sgpu_status staged_launch(DeviceIndex* d, Lane* lane, uint64_t dim, const uint64_t* q_off, const uint32_t* comps,
                          const float* vals, uint32_t nq, uint32_t q_base, const sgpu_search_params& sp, sgpu_batch** slot,
                          uint32_t followed);   // (0: nothing behind this chunk; 1: other calls; 2: the call's own next chunk)
sgpu_status staged_finish(DeviceIndex* d, Lane* lane, sgpu_batch* b, float* out_scores, uint64_t* out_ids, uint32_t* out_n);
sgpu_status summary_distances(DeviceIndex* d, const HostIndex& h, uint32_t list, const uint32_t* comps,
                              const float* vals, uint32_t nnz, float* out_dots, uint32_t* out_nb);
int device_count();
double*& call_timing();
bool device_plan_applies(const DeviceIndex* d, const sgpu_search_params& sp);
sgpu_status debug_device_plan(DeviceIndex* d, const uint64_t* q_off, const uint32_t* comps, const float* vals, uint32_t nq,
                              uint32_t query_cut, uint32_t* order_out, uint32_t* out3);
sgpu_status debug_plan(const HostIndex& h, const uint64_t* q_off, const uint32_t* comps, const float* vals, uint32_t nq,
                       uint32_t query_cut, uint32_t* order_out, uint32_t* out3);
uint32_t coop_trace_dump(DeviceIndex* d, uint64_t* out, uint32_t cap);
const DeviceIndex* batch_replica(const sgpu_batch* b);
sgpu_status device_index_set_knn(DeviceIndex* d, const std::vector<uint32_t>& knn, uint32_t knn_dim);
sgpu_status build_knn_on_device(DeviceIndex* d, HostIndex& h, uint32_t nknn);
}  // namespace sgpu

using namespace sgpu;

extern "C" {

const char* sgpu_last_error(void) { return last_error().c_str(); }
uint32_t sgpu_abi_version(void) { return 4; }

sgpu_status sgpu_device_count(int32_t* n) {
  if (!n) return fail(SGPU_EINVAL, "null argument");
  *n = device_count();
  if (*n <= 0) {
    *n = 0;
    return fail(SGPU_EDEVICE, "no HIP device visible");
  }
  return SGPU_OK;
}

sgpu_status sgpu_index_create(const sgpu_index_desc* desc, sgpu_index** out) {
  if (!desc || !out) return fail(SGPU_EINVAL, "null argument");
  sgpu_index* ix = new (std::nothrow) sgpu_index();
  if (!ix) return fail(SGPU_ENOMEM, "out of memory");
  sgpu_status st = host_index_from_desc(*desc, &ix->host);
  if (st != SGPU_OK) {
    delete ix;
    return st;
  }
  *out = ix;
  return SGPU_OK;
}

sgpu_status sgpu_index_build(uint32_t comp_width, uint64_t n_docs, uint64_t dim, const uint64_t* offsets,
                             const void* comps, const float* vals, const sgpu_build_config* cfg,
                             sgpu_index** out) {
  if (!offsets || !cfg || !out || (offsets[n_docs] && (!comps || !vals))) return fail(SGPU_EINVAL, "null argument");
  sgpu_index* ix = new (std::nothrow) sgpu_index();
  if (!ix) return fail(SGPU_ENOMEM, "out of memory");
  sgpu_status st = build_host_index(comp_width, n_docs, dim, offsets, comps, vals, *cfg, &ix->host);
  if (st != SGPU_OK) {
    delete ix;
    return st;
  }
  *out = ix;
  return SGPU_OK;
}

sgpu_status sgpu_index_convert(const sgpu_index* src, uint32_t value_type, sgpu_index** out) {
  if (!src || !out) return fail(SGPU_EINVAL, "null argument");
  sgpu_index* ix = new (std::nothrow) sgpu_index();
  if (!ix) return fail(SGPU_ENOMEM, "out of memory");
  sgpu_status st = host_index_convert(src->host, value_type, &ix->host);
  if (st != SGPU_OK) {
    delete ix;
    return st;
  }
  *out = ix;
  return SGPU_OK;
}

sgpu_status sgpu_index_get_desc(const sgpu_index* idx, sgpu_index_desc* out) {
  if (!idx || !out) return fail(SGPU_EINVAL, "null argument");
  idx->host.fill_desc(out);
  return SGPU_OK;
}

sgpu_status sgpu_index_save(const sgpu_index* idx, const char* path) {
  if (!idx || !path) return fail(SGPU_EINVAL, "null argument");
  return host_index_save(idx->host, path);
}

sgpu_status sgpu_index_load(const char* path, sgpu_index** out) {
  if (!path || !out) return fail(SGPU_EINVAL, "null argument");
  sgpu_index* ix = new (std::nothrow) sgpu_index();
  if (!ix) return fail(SGPU_ENOMEM, "out of memory");
  sgpu_status st = host_index_load(path, &ix->host);
  if (st != SGPU_OK) {
    delete ix;
    return st;
  }
  *out = ix;
  return SGPU_OK;
}

static void drop_replicas(sgpu_index* idx) {
  for (DeviceIndex* d : idx->replicas) device_index_free(d);
  idx->replicas.clear();
  idx->dev = nullptr;
}

sgpu_status sgpu_index_upload_many(sgpu_index* idx, const int32_t* device_ids, uint32_t n) {
  if (!idx || !device_ids || n == 0) return fail(SGPU_EINVAL, "null argument / no device given");
  drop_replicas(idx);
  // replica 0 is packed on the host and copied over PCIe once; the others are copied from it
  // GPU to GPU (hipMemcpyPeer: xGMI on an MI355X node) instead of n more host uploads
  DeviceIndex* first = nullptr;
  sgpu_status st = device_index_upload(idx->host, device_ids[0], &first);
  if (st != SGPU_OK) return st;
  idx->replicas.push_back(first);
  idx->dev = first;
  for (uint32_t i = 1; i < n; ++i) {
    DeviceIndex* r = nullptr;
    st = device_index_clone(first, device_ids[i], &r);
    if (st != SGPU_OK) {
      const std::string msg = last_error();   // the error of the failing replica survives the clean-up
      drop_replicas(idx);
      last_error() = msg;
      return st;
    }
    idx->replicas.push_back(r);
  }
  return SGPU_OK;
}

sgpu_status sgpu_index_upload(sgpu_index* idx, int32_t device) { return sgpu_index_upload_many(idx, &device, 1); }

uint32_t sgpu_index_replicas(const sgpu_index* idx) { return idx ? (uint32_t)idx->replicas.size() : 0; }

sgpu_status sgpu_index_set_knn(sgpu_index* idx, const uint32_t* neighbours, uint64_t n_total, uint32_t knn_dim) {
  if (!idx || (n_total && !neighbours)) return fail(SGPU_EINVAL, "null argument");
  if (n_total && knn_dim == 0) return fail(SGPU_EINVAL, "knn_dim == 0 with a non-empty neighbour array");
  for (uint64_t i = 0; i < n_total; ++i)
    if (neighbours[i] >= idx->host.n_docs) return fail(SGPU_EINVAL, "neighbour id >= n_docs");
  try {
    idx->host.knn.assign(neighbours, neighbours + n_total);
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of memory");
  }
  idx->host.knn_dim = n_total ? knn_dim : 0;
  for (DeviceIndex* d : idx->replicas) {
    sgpu_status st = device_index_set_knn(d, idx->host.knn, idx->host.knn_dim);
    if (st != SGPU_OK) return st;
  }
  return SGPU_OK;
}

sgpu_status sgpu_index_get_knn(const sgpu_index* idx, const uint32_t** neighbours, uint64_t* n_total,
                               uint32_t* knn_dim) {
  if (!idx || !neighbours || !n_total || !knn_dim) return fail(SGPU_EINVAL, "null argument");
  *neighbours = idx->host.knn.data();
  *n_total = idx->host.knn.size();
  *knn_dim = idx->host.knn_dim;
  return SGPU_OK;
}

sgpu_status sgpu_index_build_knn(sgpu_index* idx, uint32_t nknn) {
  if (!idx) return fail(SGPU_EINVAL, "null argument");
  try {
    sgpu_status st = build_knn_on_device(idx->dev, idx->host, nknn);   // searches run on replica 0
    for (size_t i = 1; st == SGPU_OK && i < idx->replicas.size(); ++i)
      st = device_index_set_knn(idx->replicas[i], idx->host.knn, idx->host.knn_dim);
    return st;
  } catch (const std::exception&) {
    return fail(SGPU_ENOMEM, "out of host memory building the kNN graph");
  }
}

uint64_t sgpu_index_device_bytes(const sgpu_index* idx) { return idx ? device_index_bytes(idx->dev) : 0; }

sgpu_status sgpu_index_stream_stats(const sgpu_index* idx, uint64_t* raw_docs, uint64_t* raw_elements) {
  if (!idx || !raw_docs || !raw_elements) return fail(SGPU_EINVAL, "null argument");
  *raw_docs = *raw_elements = 0;
  try {
    std::vector<uint8_t> raw;
    pack_dvb_raw_flags(idx->host, false, &raw);   // (empty unless the index is a DotVByte one)
    for (uint64_t d = 0; d < raw.size(); ++d)
      if (raw[d]) {
        *raw_docs += 1;
        *raw_elements += idx->host.fwd_offsets[d + 1] - idx->host.fwd_offsets[d];
      }
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory");
  }
  return SGPU_OK;
}

void sgpu_index_destroy(sgpu_index* idx) {
  if (!idx) return;
  drop_replicas(idx);
  delete idx;
}

static DeviceIndex* replica_of(sgpu_index* idx, uint32_t replica) {
  if (!idx || replica >= idx->replicas.size()) {
    fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload) / replica out of range");
    return nullptr;
  }
  return idx->replicas[replica];
}
static DeviceIndex* replica_of_batch(sgpu_index* idx, const sgpu_batch* b) {
  for (DeviceIndex* d : idx->replicas)
    if (batch_replica(b) == d) return d;
  fail(SGPU_EINVAL, "batch does not belong to this index");
  return nullptr;
}

sgpu_status sgpu_batch_create_on(sgpu_index* idx, uint32_t replica, const uint64_t* q_off, const uint32_t* comps,
                                 const float* vals, uint32_t nq, uint32_t k_max, sgpu_batch** out) {
  if (!idx || !q_off || !out || (q_off[nq] && (!comps || !vals))) return fail(SGPU_EINVAL, "null argument");
  *out = nullptr;
  DeviceIndex* d = replica_of(idx, replica);
  if (!d) return SGPU_EDEVICE;
  return batch_create(d, lane_main(d), idx->host.dim, q_off, comps, vals, nq, k_max, out);
}

sgpu_status sgpu_batch_create(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                              uint32_t nq, uint32_t k_max, sgpu_batch** out) {
  return sgpu_batch_create_on(idx, 0, q_off, comps, vals, nq, k_max, out);
}

sgpu_status sgpu_batch_run(sgpu_index* idx, sgpu_batch* batch, const sgpu_search_params* params, int32_t sync,
                           sgpu_launch_stats* stats) {
  if (!idx || !batch || !params) return fail(SGPU_EINVAL, "null argument");
  DeviceIndex* d = replica_of_batch(idx, batch);
  if (!d) return SGPU_EINVAL;
  return batch_run(d, nullptr, batch, *params, 0 /*MODE_SEARCH*/, sync, stats);
}

sgpu_status sgpu_batch_run_counted(sgpu_index* idx, sgpu_batch* batch, const sgpu_search_params* params,
                                   sgpu_launch_stats* stats) {
  if (!idx || !batch || !params) return fail(SGPU_EINVAL, "null argument");
  DeviceIndex* d = replica_of_batch(idx, batch);
  if (!d) return SGPU_EINVAL;
  return batch_run(d, nullptr, batch, *params, 2 /*MODE_COUNTED*/, 1, stats);
}

sgpu_status sgpu_batch_sync(sgpu_index* idx, sgpu_launch_stats* stats) {
  if (!idx) return fail(SGPU_EINVAL, "null argument");
  // every replica's main lane; the statistics are replica 0's
  for (size_t i = idx->replicas.size(); i-- > 1;) {
    sgpu_status st = batch_sync(idx->replicas[i], nullptr);
    if (st != SGPU_OK) return st;
  }
  return batch_sync(idx->dev, stats);
}

sgpu_status sgpu_batch_fetch(sgpu_index* idx, sgpu_batch* batch, uint32_t k, float* out_scores,
                             uint64_t* out_doc_ids, uint32_t* out_n) {
  if (!idx || !batch || !out_scores || !out_doc_ids || !out_n) return fail(SGPU_EINVAL, "null argument");
  DeviceIndex* d = replica_of_batch(idx, batch);
  if (!d) return SGPU_EINVAL;
  return batch_fetch(d, nullptr, batch, k, out_scores, out_doc_ids, out_n);
}

sgpu_status sgpu_batch_fetch_stats(sgpu_index* idx, sgpu_batch* batch, uint32_t* out_counters) {
  if (!idx || !batch || !out_counters) return fail(SGPU_EINVAL, "null argument");
  DeviceIndex* d = replica_of_batch(idx, batch);
  if (!d) return SGPU_EINVAL;
  return batch_fetch_stats(d, batch, out_counters);
}

void sgpu_batch_destroy(sgpu_batch* batch) { batch_free(batch); }

// How a shard of nq queries is cut (pure: tests/test_abi_and_host.py checks it through sgpu_debug_chunk_plan).
// chunk_jobs: the number of launches wanted before lanes are taken; *tail > 0 means "everything but the last *tail
// queries, then those" (SGPU_TAIL_COOP, two launches). chunk_bounds: the queries [q0, q1) of launch j of n_jobs.
static uint32_t chunk_jobs(uint32_t nq, uint32_t chunk_min, uint32_t chunk_max, uint32_t want_tail, uint32_t coop_max,
                           uint32_t* tail) {
  uint32_t n_jobs = 1;
  *tail = 0;
  if (chunk_min && nq >= 2 * chunk_min) n_jobs = std::min<uint32_t>(chunk_max, nq / chunk_min);
  if (n_jobs == 1 && want_tail && coop_max) {
    const uint32_t t = std::min(want_tail, coop_max);
    if (nq > coop_max + t && 2 * t < nq) {
      *tail = t;
      n_jobs = 2;
    }
  }
  return n_jobs;
}
static void chunk_bounds(uint32_t nq, uint32_t n_jobs, uint32_t tail, uint32_t j, uint32_t* q0, uint32_t* q1, uint32_t first = 0) {
  if (tail && n_jobs == 2) {
    *q0 = j == 0 ? 0 : nq - tail;
    *q1 = j == 0 ? nq - tail : nq;
    return;
  }
  if (first && n_jobs == 2 && first < nq) {   // (two chunks, the first one of `first` queries)
    *q0 = j == 0 ? 0 : first;
    *q1 = j == 0 ? first : nq;
    return;
  }
  *q0 = (uint32_t)((uint64_t)nq * j / n_jobs);
  *q1 = (uint32_t)((uint64_t)nq * (j + 1) / n_jobs);
}

// One shard of a batch on one replica: borrow a lane (its stream and recycled device batch), H2D of
// the queries, one kernel pass, D2H of the results. No allocation once the lane's batch has grown to
// the call's size; calls from different host threads take different lanes and overlap.
// A large shard is cut into up to four chunks on as many lanes (as many as are free): the host side of
// chunk i+1 (validation, launch plan, staging of the H2D) runs while the GPU searches chunk i, and
// the workgroups of chunk i+1 fill the CUs that chunk i's tail leaves idle.
static sgpu_status search_shard(DeviceIndex* d, uint64_t dim, const uint64_t* q_off, const uint32_t* comps,
                                const float* vals, uint32_t nq, uint32_t q_base, const sgpu_search_params& params,
                                float* out_scores, uint64_t* out_doc_ids, uint32_t* out_n) {
  static const uint32_t chunk_min = [] {
    const char* v = std::getenv("SGPU_CHUNK_MIN");
    // (600 since r03: a 1250-query call - one rank's shard of a 10 000-query batch on eight GPUs - takes 1115 us in two
    // chunks against 1260 in one, a 2500-query call 1.94 against 2.35 ms: the host side of a chunk, ~0.3 us per query,
    // hides behind the previous chunk's kernel; chunks of ~300 lose to their launch tails. profiles/r03_chunk_probe.txt)
    return v && *v ? (uint32_t)std::strtoul(v, nullptr, 10) : 0xffffffffu;   // (unset: by the rule below; 0: never cut a call)
  }();
  struct Job {
    Lane* lane;
    uint32_t q0, q1;
  };
  // Chunks per call: FOUR when this call is alone on the replica (its host side - validation, plan, staging: ~0.2 us per
  // query - hides behind its own earlier chunks' kernels: 1.44 M queries/s from one request thread against 1.31 / 1.11 M
  // with two / one chunk), TWO when other calls are - or were a moment ago - in flight on it (their kernels hide it, and every extra chunk is a launch
  // tail: 10 000-query calls from two / three request threads 1.670 / 1.683 M queries/s with two chunks against 1.636 /
  // 1.596 M with four - the device-resident rate is 1.685 M; r05, gpurun_out r05y). SGPU_CHUNK_MAX fixes the number.
  static const uint32_t chunk_max_env = [] {
    const char* v = std::getenv("SGPU_CHUNK_MAX");
    const uint32_t n = v && *v ? (uint32_t)std::strtoul(v, nullptr, 10) : 0u;
    return n > 8 ? 8u : n;
  }();
  struct InFlight {   // (counts this call on the replica for as long as it runs)
    DeviceIndex* d;
    bool shared;
    explicit InFlight(DeviceIndex* d_) : d(d_), shared(call_enter(d_)) {}
    ~InFlight() { call_exit(d); }
  } in_flight(d);
  // (r06: with the launch plan computed on the device - plan_kernel.hip - a chunk's host side is validation and a copy,
  // ~20 us per 1000 queries: two chunks whoever else is calling - one request thread 1.61 M queries/s against 1.59 / 1.54 M
  // with four / one, two threads 1.71 against 1.69 / 1.68; profiles/r06_entry_point_chunks.txt)
  const uint32_t chunk_max = chunk_max_env ? chunk_max_env : ((in_flight.shared || device_plan_applies(d, params)) ? 2u : 4u);
  Job jobs[8];
  // Mid-size shards (SGPU_TAIL_COOP = n, an experiment, off by default): more queries than the cooperative variant takes
  // on its own, fewer than two chunks - the last n queries go out as a second launch on another lane, small enough for
  // the cooperative variant; its workgroups become resident as the first launch's run out of queries. Measured 7 %
  // slower than one launch (profiles/r03_chunk_probe.txt).
  uint32_t tail = 0;
  uint32_t n_jobs;
  {
    const char* th = std::getenv("SGPU_TEST_HOOKS");   // (a test hook: honoured only while SGPU_TEST_HOOKS=1 is set)
    const char* tv = (th && *th && *th != '0') ? std::getenv("SGPU_TAIL_COOP") : nullptr;
    const uint32_t want = tv && *tv ? (uint32_t)std::strtoul(tv, nullptr, 10) : 0u;
    // (r06: 1300 where the chunks are planned on the device - a 1250- or 2500-query call is then ONE launch: from two request
    // threads 864 -> 781 us and 1551 -> 1475 us per call, from one thread no difference; profiles/r06_shard_probe_chunk_min.txt)
    const uint32_t cmin = chunk_min != 0xffffffffu ? chunk_min : (device_plan_applies(d, params) ? 1300u : 600u);
    n_jobs = chunk_jobs(nq, cmin, chunk_max, want, want ? coop_auto_max_queries(d) : 0u, &tail);
  }
  // (experiment, a test hook: SGPU_CHUNK_FIRST = share of the call, in per mille, that the first of two chunks takes)
  uint32_t first = 0;
  if (n_jobs == 2 && !tail) {
    const char* th = std::getenv("SGPU_TEST_HOOKS");
    const char* fv = (th && *th && *th != '0') ? std::getenv("SGPU_CHUNK_FIRST") : nullptr;
    const uint32_t pm = fv && *fv ? (uint32_t)std::strtoul(fv, nullptr, 10) : 0u;
    if (pm > 0 && pm < 1000) first = std::max<uint32_t>(1, (uint32_t)((uint64_t)nq * pm / 1000));
  }
  std::vector<uint64_t> off;   // a chunk's offsets, rebased (sized here: nothing below allocates host memory)
  if (n_jobs > 1) {
    try {
      off.resize(first ? (size_t)nq + 2 : (size_t)nq / 2 + 2);
    } catch (const std::exception&) {
      n_jobs = 1;
    }
  }
  jobs[0].lane = lane_acquire(d);
  for (uint32_t j = 1; j < n_jobs; ++j) {
    jobs[j].lane = lane_try_acquire(d);
    if (!jobs[j].lane) {
      n_jobs = j;
      break;
    }
  }
  if (n_jobs < 2) tail = 0;
  const uint32_t k = params.k;
  sgpu_status st = SGPU_OK;
  std::string msg;
  uint32_t launched = 0;
  if (n_jobs > 1) {   // the offsets of the whole shard hold before a chunk is launched (each chunk checks its own components)
    uint32_t max_nnz = 0;
    st = q_off[0] != 0 ? fail(SGPU_EINVAL, "q_off[0] must be 0") : validate_query_offsets(q_off, nq, q_base, &max_nnz);
    if (st != SGPU_OK) msg = last_error();
  }
  for (uint32_t j = 0; j < n_jobs && st == SGPU_OK; ++j) {
    Job& jb = jobs[j];
    chunk_bounds(nq, n_jobs, tail, j, &jb.q0, &jb.q1, first);
    const uint64_t* qo = q_off;
    if (n_jobs > 1 && jb.q0 != 0) {   // (a chunk that starts at query 0 uses the caller's offsets as they are)
      for (uint32_t q = jb.q0; q <= jb.q1; ++q) off[q - jb.q0] = q_off[q] - q_off[jb.q0];
      qo = off.data();
    }
    st = staged_launch(d, jb.lane, dim, qo, comps ? comps + q_off[jb.q0] : nullptr, vals ? vals + q_off[jb.q0] : nullptr,
                       jb.q1 - jb.q0, q_base + jb.q0, params, lane_scratch(jb.lane), j + 1 < n_jobs ? 2u : (in_flight.shared ? 1u : 0u));
    if (st == SGPU_OK) ++launched;
    else msg = last_error();
  }
  for (uint32_t j = 0; j < launched; ++j) {   // every launched chunk is waited for, also after an error
    Job& jb = jobs[j];
    const sgpu_status fs = staged_finish(d, jb.lane, *lane_scratch(jb.lane), out_scores + (size_t)jb.q0 * k,
                                         out_doc_ids + (size_t)jb.q0 * k, out_n + jb.q0);
    if (fs != SGPU_OK && st == SGPU_OK) {
      st = fs;
      msg = last_error();
    }
  }
  for (uint32_t j = 0; j < n_jobs; ++j) lane_release(d, jobs[j].lane);
  if (st != SGPU_OK) last_error() = msg;
  return st;
}

sgpu_status sgpu_batch_search(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                              uint32_t nq, const sgpu_search_params* params, float* out_scores,
                              uint64_t* out_doc_ids, uint32_t* out_n) {
  if (!idx || !params || !q_off || !out_scores || !out_doc_ids || !out_n) return fail(SGPU_EINVAL, "null argument");
  if (params->k == 0) return fail(SGPU_EINVAL, "k must be > 0 (KHeap::new asserts, reference src/utils.rs:23)");
  if (idx->replicas.empty()) return fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload)");
  if (q_off[nq] && (!comps || !vals)) return fail(SGPU_EINVAL, "null argument");
  const uint32_t n_rep = (uint32_t)idx->replicas.size();
  if (n_rep == 1) return search_shard(idx->dev, idx->host.dim, q_off, comps, vals, nq, 0, *params, out_scores, out_doc_ids, out_n);
  if (nq < 2 * n_rep) {   // too small to shard (single queries of a serving loop): the replicas take such calls in turn
    DeviceIndex* d = idx->replicas[idx->next_replica.fetch_add(1, std::memory_order_relaxed) % n_rep];
    return search_shard(d, idx->host.dim, q_off, comps, vals, nq, 0, *params, out_scores, out_doc_ids, out_n);
  }
  // Index replicated on several GPUs: contiguous shards of the batch, one host thread per GPU, no
  // collective; results land in input order (the reference's rayon loop over queries,
  // src/pylib/mod.rs:629-652, 1129-1145, becomes one shard per device).
  {   // the offsets of the whole batch hold before it is cut (each shard checks its own components)
    uint32_t max_nnz = 0;
    if (q_off[0] != 0) return fail(SGPU_EINVAL, "q_off[0] must be 0");
    const sgpu_status vst = validate_query_offsets(q_off, nq, 0, &max_nnz);
    if (vst != SGPU_OK) return vst;
  }
  const uint32_t k = params->k;
  try {
    std::vector<sgpu_status> sts(n_rep, SGPU_OK);
    std::vector<std::string> msgs(n_rep);
    std::vector<std::vector<uint64_t>> offs(n_rep);   // every shard's offsets, rebased; sized before a thread starts
    for (uint32_t r = 0; r < n_rep; ++r) {
      const uint32_t q0 = (uint32_t)((uint64_t)nq * r / n_rep), q1 = (uint32_t)((uint64_t)nq * (r + 1) / n_rep);
      offs[r].resize(q1 - q0 + 1);
      for (uint32_t q = q0; q <= q1; ++q) offs[r][q - q0] = q_off[q] - q_off[q0];
    }
    auto shard = [&](uint32_t r) {
      const uint32_t q0 = (uint32_t)((uint64_t)nq * r / n_rep), q1 = (uint32_t)((uint64_t)nq * (r + 1) / n_rep);
      try {
        sts[r] = search_shard(idx->replicas[r], idx->host.dim, offs[r].data(), comps ? comps + q_off[q0] : nullptr,
                              vals ? vals + q_off[q0] : nullptr, q1 - q0, q0, *params, out_scores + (size_t)q0 * k,
                              out_doc_ids + (size_t)q0 * k, out_n + q0);
        if (sts[r] != SGPU_OK) msgs[r] = last_error();
      } catch (const std::exception&) {
        sts[r] = SGPU_ENOMEM;
      }
    };
    std::vector<std::thread> threads;
    threads.reserve(n_rep);
    for (uint32_t r = 1; r < n_rep; ++r) {
      try {
        threads.emplace_back(shard, r);
      } catch (const std::exception&) {   // no thread to be had: the shard runs on this one
        shard(r);
      }
    }
    shard(0);   // the calling thread drives replica 0
    for (auto& t : threads) t.join();
    for (uint32_t r = 0; r < n_rep; ++r)
      if (sts[r] != SGPU_OK) {
        if (msgs[r].empty()) return fail(sts[r], "out of host memory searching shard %u", r);
        last_error() = msgs[r];
        return sts[r];
      }
  } catch (const std::exception&) {
    return fail(SGPU_ENOMEM, "out of host memory sharding a query batch");
  }
  return SGPU_OK;
}

sgpu_status sgpu_search(sgpu_index* idx, const uint32_t* comps, const float* vals, uint32_t nnz,
                        const sgpu_search_params* params, float* out_scores, uint64_t* out_doc_ids,
                        uint32_t* out_n) {
  const uint64_t q_off[2] = {0, nnz};
  return sgpu_batch_search(idx, q_off, comps, vals, 1, params, out_scores, out_doc_ids, out_n);
}

// The reference's AQT loop (src/bin/perf_inverted_index.rs:184-216): the queries of a set searched one
// at a time, each through sgpu_search, timed around the whole loop.
sgpu_status sgpu_search_sequential_timed(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                                         uint32_t nq, const sgpu_search_params* params, float* out_scores,
                                         uint64_t* out_doc_ids, uint32_t* out_n, double* mean_us, double* breakdown_us,
                                         double* per_query_us) {
  if (!idx || !params || !q_off || !out_scores || !out_doc_ids || !out_n) return fail(SGPU_EINVAL, "null argument");
  if (params->k == 0) return fail(SGPU_EINVAL, "k must be > 0 (KHeap::new asserts, reference src/utils.rs:23)");
  double phases[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double*& slot = call_timing();
  double* const saved = slot;
  if (breakdown_us) slot = phases;
  const uint32_t k = params->k;
  sgpu_status st = SGPU_OK;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t q = 0; q < nq && st == SGPU_OK; ++q) {
    if (q_off[q + 1] < q_off[q] || q_off[q + 1] - q_off[q] > 0xffffffffull) {
      st = fail(SGPU_EINVAL, "query offsets must be monotone (query %u)", q);
      break;
    }
    const auto tq = per_query_us ? std::chrono::steady_clock::now() : t0;
    st = sgpu_search(idx, comps ? comps + q_off[q] : nullptr, vals ? vals + q_off[q] : nullptr,
                     (uint32_t)(q_off[q + 1] - q_off[q]), params, out_scores + (size_t)q * k,
                     out_doc_ids + (size_t)q * k, out_n + q);
    if (per_query_us) per_query_us[q] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tq).count();
  }
  const double total = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  slot = saved;
  if (mean_us) *mean_us = nq ? total / nq : 0.0;
  if (breakdown_us)
    for (int i = 0; i < 8; ++i) breakdown_us[i] = nq ? phases[i] / nq : 0.0;
  return st;
}

sgpu_status sgpu_search_sequential(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                                   uint32_t nq, const sgpu_search_params* params, float* out_scores,
                                   uint64_t* out_doc_ids, uint32_t* out_n, double* mean_us, double* breakdown_us) {
  return sgpu_search_sequential_timed(idx, q_off, comps, vals, nq, params, out_scores, out_doc_ids, out_n, mean_us,
                                      breakdown_us, nullptr);
}

// ---- test hooks (include/seismic_hip_testing.h; NOT part of the boundary) ---------------------------------------
// Entry points the test suite and the tools use to look at host-side decisions (team size, chunk plan, launch plan,
// packed records, phase clocks, cooperative trace). Like the undocumented environment names they are inert unless
// SGPU_TEST_HOOKS=1 is set: status-returning ones fail with SGPU_EINVAL, the others return 0 / do nothing.
static bool test_hooks_on() {
  const char* t = std::getenv("SGPU_TEST_HOOKS");
  return t && *t && *t != '0';
}
#define SGPU_HOOK_OR(ret)                                                                            \
  if (!test_hooks_on()) {                                                                            \
    (void)fail(SGPU_EINVAL, "%s is a test hook: set SGPU_TEST_HOOKS=1 (include/seismic_hip_testing.h)", __func__); \
    return ret;                                                                                      \
  }

// (the team size a host-parallel phase would take for num_threads == 0)
uint32_t sgpu_debug_host_threads(void) {
  SGPU_HOOK_OR(0u);
  return (uint32_t)host_threads();
}

// (not part of the boundary: how search_shard would cut a call of nq queries when `lanes_free` lanes can be had -
// bounds[2 * j], bounds[2 * j + 1] = the queries [q0, q1) of launch j; returns the number of launches)
uint32_t sgpu_debug_chunk_plan(uint32_t nq, uint32_t chunk_min, uint32_t chunk_max, uint32_t want_tail, uint32_t coop_max,
                               uint32_t lanes_free, uint32_t* bounds) {
  SGPU_HOOK_OR(0u);
  uint32_t tail = 0;
  uint32_t n_jobs = chunk_jobs(nq, chunk_min, chunk_max < 1 ? 1 : (chunk_max > 8 ? 8 : chunk_max), want_tail, coop_max, &tail);
  if (lanes_free >= 1 && n_jobs > lanes_free) n_jobs = lanes_free;
  if (n_jobs < 2) tail = 0;
  for (uint32_t j = 0; j < n_jobs; ++j) chunk_bounds(nq, n_jobs, tail, j, bounds + 2 * j, bounds + 2 * j + 1);
  return n_jobs;
}

// (not part of the boundary: the forward store as sgpu_index_upload packs it - document-major records - and the ref
// of every document, (record offset / 16) << 16 | length field; out_fwd == null: *out_bytes = the size needed.
// tests/test_abi_and_host.py decodes every record with the oracle's restatement of the layout.)
sgpu_status sgpu_debug_pack_forward(const sgpu_index* idx, uint8_t* out_fwd, uint64_t cap, uint64_t* out_doc_ref, uint64_t* out_bytes) {
  SGPU_HOOK_OR(SGPU_EINVAL);
  if (!idx || !out_bytes) return fail(SGPU_EINVAL, "null argument");
  try {
    std::vector<uint8_t> raw, fwd;
    std::vector<uint64_t> off16, dref;
    {   // (an f16 index: the layout sgpu_index_upload would choose - plain unless SGPU_FWD_STREAM=sliced)
      const char* fs = std::getenv("SGPU_FWD_STREAM");
      pack_dvb_raw_flags(idx->host, fs && std::string(fs) == "sliced", &raw);
    }
    pack_record_offsets(idx->host, raw, 128 / 16, &off16);
    *out_bytes = std::max<uint64_t>(off16[idx->host.n_docs] * 16, 16);
    if (!out_fwd) return SGPU_OK;
    if (cap < *out_bytes || !out_doc_ref) return fail(SGPU_EINVAL, "buffer too small");
    pack_records(idx->host, raw, off16, &fwd);
    pack_doc_refs(idx->host, raw, off16, &dref);
    std::memcpy(out_fwd, fwd.data(), fwd.size());
    std::memcpy(out_doc_ref, dref.data(), dref.size() * 8);
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory");
  }
  return SGPU_OK;
}

// (the hashed row directory as sgpu_index_upload builds it: 4 words per slot, 4 slots per bucket; out == null: *n_words
// and *n_buckets only. *n_buckets == 0: this index gets none)
sgpu_status sgpu_debug_row_dir(const sgpu_index* idx, uint32_t* out, uint64_t cap_words, uint64_t* n_words, uint32_t* n_buckets) {
  SGPU_HOOK_OR(SGPU_EINVAL);
  if (!idx || !n_words || !n_buckets) return fail(SGPU_EINVAL, "null argument");
  try {
    std::vector<uint16_t> mid;
    std::vector<uint32_t> dir;
    pack_row_mid(idx->host, &mid);
    *n_buckets = 0;
    if (!pack_row_dir(idx->host, mid, &dir, n_buckets)) *n_buckets = 0;
    *n_words = dir.size();
    if (!out) return SGPU_OK;
    if (cap_words < dir.size()) return fail(SGPU_EINVAL, "buffer too small");
    std::memcpy(out, dir.data(), dir.size() * 4);
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory");
  }
  return SGPU_OK;
}

// (not part of the boundary: the launch plan of a batch - processing order (longest expected first), out3 = {block dots a
// query needs at most, largest list walked first, largest list walked}; needs no device)
sgpu_status sgpu_debug_plan(const sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals, uint32_t nq,
                            uint32_t query_cut, uint32_t* order_out, uint32_t* out3) {
  SGPU_HOOK_OR(SGPU_EINVAL);
  if (!idx || !q_off || !order_out || !out3) return fail(SGPU_EINVAL, "null argument");
  return debug_plan(idx->host, q_off, comps, vals, nq, query_cut, order_out, out3);
}

// (not part of the boundary: the same plan as the DEVICE computes it for staged chunks - plan_kernel.hip; replica 0)
sgpu_status sgpu_debug_device_plan(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals, uint32_t nq,
                                   uint32_t query_cut, uint32_t* order_out, uint32_t* out3) {
  SGPU_HOOK_OR(SGPU_EINVAL);
  if (!idx || !q_off || !order_out || !out3) return fail(SGPU_EINVAL, "null argument");
  return debug_device_plan(idx->dev, q_off, comps, vals, nq, query_cut, order_out, out3);
}

// (not part of the boundary: the calling thread's staged calls add the wall time of their host-side phases to
// buf[0..7] from now on - see device_index.hip: call_timing; null switches it off. tools/shard_probe.py)
void sgpu_debug_call_timing(double* buf8) {
  SGPU_HOOK_OR();
  call_timing() = buf8;
}

// (not part of the boundary: the timeline of a cooperative launch, for tools/coop_trace.py on a trace build)
uint32_t sgpu_debug_coop_trace(sgpu_index* idx, uint64_t* out, uint32_t cap) {
  SGPU_HOOK_OR(0u);
  return idx ? coop_trace_dump(idx->dev, out, cap) : 0;
}

sgpu_status sgpu_summary_distances(sgpu_index* idx, uint32_t list, const uint32_t* comps, const float* vals,
                                   uint32_t nnz, float* out_dots, uint32_t* out_n_blocks) {
  if (!idx || !out_dots || !out_n_blocks || (nnz && (!comps || !vals))) return fail(SGPU_EINVAL, "null argument");
  return summary_distances(idx->dev, idx->host, list, comps, vals, nnz, out_dots, out_n_blocks);
}

sgpu_status sgpu_exact_search(const sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps,
                              const float* vals, uint32_t nq, uint32_t k, uint32_t num_threads,
                              float* out_scores, uint64_t* out_doc_ids, uint32_t* out_n) {
  if (!idx || !q_off || !out_scores || !out_doc_ids || !out_n) return fail(SGPU_EINVAL, "null argument");
  return exact_search_host(idx->host, q_off, comps, vals, nq, k, num_threads, out_scores, out_doc_ids, out_n);
}

}  // extern "C"
