// abi.cpp — the extern "C" surface declared in include/seismic_hip.h.
#include <mutex>
#include <new>

#include "host_index.hpp"

struct sgpu_batch;

namespace sgpu {
// device_index.hip
sgpu_status device_index_upload(const HostIndex& h, int device, DeviceIndex** out);
void device_index_free(DeviceIndex* d);
uint64_t device_index_bytes(const DeviceIndex* d);
sgpu_status batch_create(DeviceIndex* d, uint64_t dim, const uint64_t* q_off, const uint32_t* comps,
                         const float* vals, uint32_t nq, uint32_t k_max, sgpu_batch** out);
void batch_free(sgpu_batch* b);
sgpu_status batch_run(DeviceIndex* d, sgpu_batch* b, const sgpu_search_params& sp, uint32_t mode, int sync,
                      sgpu_launch_stats* stats);
sgpu_status batch_sync(DeviceIndex* d, sgpu_launch_stats* stats);
sgpu_status batch_fetch(DeviceIndex* d, sgpu_batch* b, uint32_t k, float* out_scores, uint64_t* out_ids,
                        uint32_t* out_n);
sgpu_status batch_fetch_stats(DeviceIndex* d, sgpu_batch* b, uint32_t* out);
sgpu_status summary_distances(DeviceIndex* d, const HostIndex& h, uint32_t list, const uint32_t* comps,
                              const float* vals, uint32_t nnz, float* out_dots, uint32_t* out_nb);
int device_count();
sgpu_batch** device_index_scratch_batch(DeviceIndex* d);
sgpu_status device_index_set_knn(DeviceIndex* d, const std::vector<uint32_t>& knn, uint32_t knn_dim);
sgpu_status build_knn_on_device(DeviceIndex* d, HostIndex& h, uint32_t nknn);
}  // namespace sgpu

using namespace sgpu;

extern "C" {

const char* sgpu_last_error(void) { return last_error().c_str(); }
uint32_t sgpu_abi_version(void) { return 1; }

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

sgpu_status sgpu_index_upload(sgpu_index* idx, int32_t device) {
  if (!idx) return fail(SGPU_EINVAL, "null argument");
  if (idx->dev) {
    device_index_free(idx->dev);
    idx->dev = nullptr;
  }
  return device_index_upload(idx->host, device, &idx->dev);
}

sgpu_status sgpu_index_set_knn(sgpu_index* idx, const uint32_t* neighbours, uint64_t n_total, uint32_t knn_dim) {
  if (!idx || (n_total && !neighbours)) return fail(SGPU_EINVAL, "null argument");
  for (uint64_t i = 0; i < n_total; ++i)
    if (neighbours[i] >= idx->host.n_docs) return fail(SGPU_EINVAL, "neighbour id >= n_docs");
  try {
    idx->host.knn.assign(neighbours, neighbours + n_total);
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of memory");
  }
  idx->host.knn_dim = n_total ? knn_dim : 0;
  return device_index_set_knn(idx->dev, idx->host.knn, idx->host.knn_dim);
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
  return build_knn_on_device(idx->dev, idx->host, nknn);
}

uint64_t sgpu_index_device_bytes(const sgpu_index* idx) { return idx ? device_index_bytes(idx->dev) : 0; }

void sgpu_index_destroy(sgpu_index* idx) {
  if (!idx) return;
  if (idx->dev) device_index_free(idx->dev);
  delete idx;
}

sgpu_status sgpu_batch_create(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                              uint32_t nq, uint32_t k_max, sgpu_batch** out) {
  if (!idx || !q_off || !out || (q_off[nq] && (!comps || !vals))) return fail(SGPU_EINVAL, "null argument");
  *out = nullptr;
  return batch_create(idx->dev, idx->host.dim, q_off, comps, vals, nq, k_max, out);
}

sgpu_status sgpu_batch_run(sgpu_index* idx, sgpu_batch* batch, const sgpu_search_params* params, int32_t sync,
                           sgpu_launch_stats* stats) {
  if (!idx || !batch || !params) return fail(SGPU_EINVAL, "null argument");
  return batch_run(idx->dev, batch, *params, 0 /*MODE_SEARCH*/, sync, stats);
}

sgpu_status sgpu_batch_run_counted(sgpu_index* idx, sgpu_batch* batch, const sgpu_search_params* params,
                                   sgpu_launch_stats* stats) {
  if (!idx || !batch || !params) return fail(SGPU_EINVAL, "null argument");
  return batch_run(idx->dev, batch, *params, 2 /*MODE_COUNTED*/, 1, stats);
}

sgpu_status sgpu_batch_sync(sgpu_index* idx, sgpu_launch_stats* stats) {
  if (!idx) return fail(SGPU_EINVAL, "null argument");
  return batch_sync(idx->dev, stats);
}

sgpu_status sgpu_batch_fetch(sgpu_index* idx, sgpu_batch* batch, uint32_t k, float* out_scores,
                             uint64_t* out_doc_ids, uint32_t* out_n) {
  if (!idx || !batch || !out_scores || !out_doc_ids || !out_n) return fail(SGPU_EINVAL, "null argument");
  return batch_fetch(idx->dev, batch, k, out_scores, out_doc_ids, out_n);
}

sgpu_status sgpu_batch_fetch_stats(sgpu_index* idx, sgpu_batch* batch, uint32_t* out_counters) {
  if (!idx || !batch || !out_counters) return fail(SGPU_EINVAL, "null argument");
  return batch_fetch_stats(idx->dev, batch, out_counters);
}

void sgpu_batch_destroy(sgpu_batch* batch) { batch_free(batch); }

sgpu_status sgpu_batch_search(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                              uint32_t nq, const sgpu_search_params* params, float* out_scores,
                              uint64_t* out_doc_ids, uint32_t* out_n) {
  if (!idx || !params || !q_off || !out_scores || !out_doc_ids || !out_n) return fail(SGPU_EINVAL, "null argument");
  if (params->k == 0) return fail(SGPU_EINVAL, "k must be > 0 (KHeap::new asserts, reference src/utils.rs:23)");
  if (!idx->dev) return fail(SGPU_EDEVICE, "index is not uploaded to a device (call sgpu_index_upload)");
  if (q_off[nq] && (!comps || !vals)) return fail(SGPU_EINVAL, "null argument");
  // one recycled device batch per index: a call costs the H2D of the queries, one kernel pass and
  // the D2H of the results, no allocation (calls on one index are serialised by this mutex)
  static std::mutex scratch_mu;
  std::lock_guard<std::mutex> lock(scratch_mu);
  sgpu_batch** slot = device_index_scratch_batch(idx->dev);
  sgpu_status st = batch_create(idx->dev, idx->host.dim, q_off, comps, vals, nq, params->k, slot);
  if (st != SGPU_OK) return st;
  st = batch_run(idx->dev, *slot, *params, 0, 1, nullptr);
  if (st == SGPU_OK) st = batch_fetch(idx->dev, *slot, params->k, out_scores, out_doc_ids, out_n);
  return st;
}

sgpu_status sgpu_search(sgpu_index* idx, const uint32_t* comps, const float* vals, uint32_t nnz,
                        const sgpu_search_params* params, float* out_scores, uint64_t* out_doc_ids,
                        uint32_t* out_n) {
  const uint64_t q_off[2] = {0, nnz};
  return sgpu_batch_search(idx, q_off, comps, vals, 1, params, out_scores, out_doc_ids, out_n);
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
