// build_device.hpp — the steps of the index build that run on a HIP device (build_assign.hip,
// build_summaries.hip), as seen from the host builder (builder.cpp). Plain types only.
#pragma once
#include <cstdint>
#include <vector>

#include "common.hpp"

namespace sgpu {

// ---- clustering (build_assign.hip) -------------------------------------------------------------
// Largest number of centroids a list may have to be clustered on `device` (others stay on the host): bounded by the
// LDS a workgroup may have there. 0 = the device cannot be queried.
uint32_t device_assign_max_centroids(int device);
// cid_out[lp_off[c] + t] = index (within list c's centroids) of the centroid posting t of list c belongs
// to, for every list with eligible[c] != 0. `top` holds doc_cut {component | ~0, f32 bits} pairs per document.
sgpu_status device_assign_clusters(int device, uint32_t comp_width, uint64_t n_docs, uint64_t dim, uint64_t nnz,
                                   const uint64_t* doc_off, const void* doc_comp, const uint16_t* doc_val,
                                   const void* top, uint32_t doc_cut, uint32_t min_cluster_size,
                                   const uint64_t* lp_off, const uint32_t* post, const uint64_t* lc_off,
                                   const uint32_t* cent, const uint8_t* eligible, uint64_t inv_cap,
                                   uint32_t* cid_out);

// ---- per-block summaries (build_summaries.hip) ---------------------------------------------------
struct DeviceSummaries {
  std::vector<uint8_t> done;     // per block: 1 = summarised on the device (others: the host does them)
  std::vector<uint32_t> keep;    // kept components of the block
  std::vector<float> mn, qt;     // its quantisation (minimum, step)
  std::vector<uint64_t> start;   // first kept entry of the block in comp / code
  std::vector<uint32_t> comp;    // kept components, ascending inside a block
  std::vector<uint8_t> code;     // their u8 codes
};
// Largest number of document entries (sum of the block's document lengths) a block may have.
uint32_t device_summary_max_entries();
// Block b holds the documents post[blk_post[b] .. blk_post[b+1]) with blk_entries[b] entries in all.
sgpu_status device_block_summaries(int device, uint32_t comp_width, uint64_t n_docs, uint64_t nnz, const uint64_t* doc_off,
                                   const void* doc_comp, const uint16_t* doc_val, float summary_energy, uint64_t n_blocks,
                                   const uint64_t* blk_post, const uint32_t* blk_entries, const uint32_t* post,
                                   DeviceSummaries* out);

}  // namespace sgpu
