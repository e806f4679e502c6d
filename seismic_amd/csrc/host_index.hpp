// host_index.hpp — canonical host-side index (the arrays of sgpu_index_desc, owned).
#pragma once
#include <cstdint>
#include <atomic>
#include <vector>

#include "common.hpp"

namespace sgpu {

struct DeviceIndex;  // search.hip

struct HostIndex {
  uint32_t comp_width = 2;
  uint64_t n_docs = 0, dim = 0;
  std::vector<uint64_t> fwd_offsets;
  std::vector<uint8_t> fwd_comps;  // nnz * comp_width bytes
  std::vector<uint16_t> fwd_vals;   // value_type F16: binary16 bits
  std::vector<uint8_t> fwd_codes;   // value_type FIXEDU8: value = code * val_scale
  uint32_t value_type = SGPU_VAL_F16;
  float val_scale = 0.0f;
  std::vector<uint64_t> list_block_start, block_post_start;
  std::vector<uint32_t> post_doc;
  std::vector<float> blk_min, blk_quant;
  std::vector<uint64_t> list_row_start;
  std::vector<uint8_t> row_comp;  // n_rows * comp_width bytes
  std::vector<uint64_t> row_ptr;
  std::vector<uint16_t> sum_bid;
  std::vector<uint8_t> sum_code;
  // optional kNN graph (reference Knn, src/inverted_index.rs:430-435): neighbour ids flattened in
  // document order, knn_dim per document (the reference flattens the same way, :487-493)
  std::vector<uint32_t> knn;
  uint32_t knn_dim = 0;

  uint64_t nnz() const { return fwd_offsets.empty() ? 0 : fwd_offsets.back(); }
  uint64_t n_blocks() const { return list_block_start.empty() ? 0 : list_block_start.back(); }
  uint64_t n_postings() const { return block_post_start.empty() ? 0 : block_post_start.back(); }
  uint64_t n_rows() const { return list_row_start.empty() ? 0 : list_row_start.back(); }
  uint64_t n_entries() const { return row_ptr.empty() ? 0 : row_ptr.back(); }
  inline uint32_t comp(uint64_t i) const {
    return comp_width == 2 ? (uint32_t)((const uint16_t*)fwd_comps.data())[i]
                           : ((const uint32_t*)fwd_comps.data())[i];
  }
  inline float val(uint64_t i) const {   // document value i as f32 (exact for both value types)
    return value_type == SGPU_VAL_F16 ? f16_to_f32(fwd_vals[i]) : (float)fwd_codes[i] * val_scale;
  }
  inline uint32_t val_bytes() const { return value_type == SGPU_VAL_F16 ? 2u : 1u; }   // (fixed-u8 and DotVByte: u8 codes)
  inline uint32_t rcomp(uint64_t i) const {
    return comp_width == 2 ? (uint32_t)((const uint16_t*)row_comp.data())[i]
                           : ((const uint32_t*)row_comp.data())[i];
  }
  void fill_desc(sgpu_index_desc* d) const;
};

// structural validation of a descriptor (monotone offsets, ids in range, sorted rows ...)
sgpu_status validate_desc(const sgpu_index_desc& d);
sgpu_status host_index_from_desc(const sgpu_index_desc& d, HostIndex* out);
sgpu_status host_index_save(const HostIndex& ix, const char* path);
sgpu_status host_index_load(const char* path, HostIndex* out);
sgpu_status host_index_convert(const HostIndex& src, uint32_t value_type, HostIndex* out);

// CSR query batch: q_off[0] == 0 and monotone, components strictly ascending and < dim, no NaN,
// at most 65535 components per query; *max_nnz = the longest query. Error messages name query q_base + q
// (a chunk or shard of a larger batch reports the caller's numbering).
sgpu_status validate_queries(uint64_t dim, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                             uint32_t nq, uint32_t* max_nnz, uint32_t q_base = 0);
// the offsets alone (monotone, query and batch size limits): what has to hold before a batch is cut
sgpu_status validate_query_offsets(const uint64_t* q_off, uint32_t nq, uint32_t q_base, uint32_t* max_nnz);
// pack_index.cpp: the host half of the upload (HBM layout of DESIGN.md section 2), on all host cores
// (raw: per document, 1 = a DotVByte index keeps the document in the raw record form; empty for the other value types)
// (f16_slices: also give an f16 index over u16 components the compressed component stream - search_kernel.inc VT_F16S)
bool pack_f16_slices_possible(const HostIndex& h);
void pack_dvb_raw_flags(const HostIndex& h, bool f16_slices, std::vector<uint8_t>* raw);
void pack_record_offsets(const HostIndex& h, const std::vector<uint8_t>& raw, uint64_t line16, std::vector<uint64_t>* rec_off16);
void pack_records(const HostIndex& h, const std::vector<uint8_t>& raw, const std::vector<uint64_t>& rec_off16, std::vector<uint8_t>* fwd);
void pack_block_sizes(const HostIndex& h, const std::vector<uint8_t>& raw, std::vector<uint64_t>* bsize);
void pack_post_refs(const HostIndex& h, const std::vector<uint8_t>& raw, const std::vector<uint64_t>& rec_off16,
                    const std::vector<uint64_t>& bsize, bool block_major, uint64_t blk_base, std::vector<uint64_t>* pref);
void pack_doc_refs(const HostIndex& h, const std::vector<uint8_t>& raw, const std::vector<uint64_t>& rec_off16, std::vector<uint64_t>* dref);
void pack_narrow(const std::vector<uint64_t>& v, std::vector<uint32_t>* out);
void pack_row_mid(const HostIndex& h, std::vector<uint16_t>* mid);
bool pack_row_dir(const HostIndex& h, const std::vector<uint16_t>& mid, std::vector<uint32_t>* out, uint32_t* n_buckets);
void pack_sum_deq(const HostIndex& h, std::vector<float>* deq);

// builder.cpp
sgpu_status build_host_index(uint32_t comp_width, uint64_t n_docs, uint64_t dim, const uint64_t* offsets,
                             const void* comps, const float* vals, const sgpu_build_config& cfg,
                             HostIndex* out);
// exact.cpp
sgpu_status exact_search_host(const HostIndex& ix, const uint64_t* q_off, const uint32_t* comps,
                              const float* vals, uint32_t nq, uint32_t k, uint32_t num_threads,
                              float* out_scores, uint64_t* out_ids, uint32_t* out_n);

}  // namespace sgpu

// the opaque handle of the C ABI
struct sgpu_index {
  sgpu::HostIndex host;
  std::vector<sgpu::DeviceIndex*> replicas;   // one per device the index was uploaded to
  sgpu::DeviceIndex* dev = nullptr;           // replicas[0] (null before upload)
  std::atomic<uint32_t> next_replica{0};      // calls too small to shard go to the replicas in turn
};
