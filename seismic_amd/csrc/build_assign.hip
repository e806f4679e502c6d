// build_assign.hip — index build, the clustering step on the GPU.
//
// blocking_with_random_kmeans (reference src/posting_list.rs:227-300) spends its time in
// do_random_kmeans_on_docids_ii_approx_dot_product (src/utils.rs:146-237): every posting of a list is
// assigned to the sampled centroid with the largest approximate dot product (the document's doc_cut
// heaviest components against an inverted file of the centroids), clusters of at most
// min_cluster_size documents are dissolved and their documents assigned again among the remaining
// centroids. That is hot-loop-A-shaped work (stream a few rows of an inverted file, scatter-add into a
// small array of accumulators), and it runs here with the same structure as the search kernel's
// stage 1: one posting list per workgroup (persistent workgroups pull lists from a queue, longest
// first), one document per wavefront, the accumulators of the wavefront in LDS. A wavefront walks its
// document's components in the reference's order and the lanes of a step touch distinct centroids, so
// every accumulator receives the reference's additions in the reference's order with the reference's
// roundings (product, then sum; no FMA): the assignment is IDENTICAL to the host builder's, and the
// index built with it is byte-identical (tests/test_builder_parity.py, tests/test_gpu_build.py).
//
// Sampling the centroids, sorting postings, forming blocks and the per-block summaries stay on the
// host (builder.cpp); this file only answers "which centroid does each posting of each list go to".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "build_device.hpp"

namespace sgpu {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(SGPU_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

constexpr int kAssignThreads = 256;            // 4 wavefronts per workgroup, a document each
constexpr uint32_t kAssignMaxCentroids = 8192;  // accumulators of a wavefront: up to 32 KB of LDS (r04: 8192; lists of n_postings 6000 x max_fraction 4 x centroid_fraction 0.2 = 4800 centroids stayed on the host, 25 s)

struct AssignView {
  // documents (the forward index as the builder holds it)
  const uint64_t* doc_off;
  const void* doc_comp;       // comp_width bytes per component
  const uint16_t* doc_val;    // binary16
  const uint2* top;           // n_docs x doc_cut: {component or ~0, value bits}, heaviest first
  uint32_t comp_width, doc_cut, dim, min_cluster_size;
  // lists
  const uint64_t* lp_off;     // postings of list c: post[lp_off[c] .. lp_off[c+1])
  const uint32_t* post;
  const uint64_t* lc_off;     // centroid documents of list c
  const uint32_t* cent;
  const uint32_t* order;      // lists to process, longest first
  uint32_t n_lists;
  // output: index (within the list's centroids) of the centroid each posting is assigned to
  uint32_t* cid_out;
  // per-workgroup scratch
  uint32_t* comp_cnt;         // [grid][dim]   entries of the centroid inverted file per component (0 outside a list)
  uint32_t* comp_pos;         // [grid][dim]
  uint32_t* comp_cur;         // [grid][dim]
  uint32_t* touched;          // [grid][touched_cap]
  uint32_t* inv_cid;          // [grid][inv_cap]
  float* inv_val;             // [grid][inv_cap]
  uint32_t* csize;            // [grid][kAssignMaxCentroids] cluster sizes
  uint32_t* counters;         // [grid][4]: touched count, inverted-file size
  uint64_t touched_cap, inv_cap;
  uint32_t* queue;
};

__device__ __forceinline__ uint32_t comp_of(const AssignView& v, uint64_t i) {
  return v.comp_width == 2 ? (uint32_t)((const uint16_t*)v.doc_comp)[i] : ((const uint32_t*)v.doc_comp)[i];
}
__device__ __forceinline__ float half_to_float(uint16_t h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return (float)x;
}
// f32::total_cmp order as an unsigned key
__device__ __forceinline__ uint32_t order_key(float f) {
  int32_t b = __float_as_int(f);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return (uint32_t)b ^ 0x80000000u;
}

// The centroid with the largest (score by total_cmp, index) among the non-avoided ones - Rust's
// max_by keeps the LAST maximum (src/utils.rs:135-141) - or centroid 0 when every one is avoided.
__device__ __forceinline__ uint32_t best_centroid(const float* scores, const uint32_t* avoided_bits, uint32_t nc) {
  const uint32_t lane = threadIdx.x & 63;
  unsigned long long best = 0ull;   // (key + 1) << 32 | cid; 0 = none
  for (uint32_t c = lane; c < nc; c += 64) {
    if (avoided_bits && ((avoided_bits[c >> 5] >> (c & 31)) & 1u)) continue;
    const unsigned long long k = (((unsigned long long)order_key(scores[c]) + 1ull) << 32) | c;
    best = k > best ? k : best;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d);
    best = o > best ? o : best;
  }
  return best ? (uint32_t)best : 0u;
}

__global__ __launch_bounds__(kAssignThreads) void assign_clusters_kernel(AssignView v) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ uint32_t s_list;
  __shared__ uint32_t s_avoid[kAssignMaxCentroids / 32];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr uint32_t NW = kAssignThreads / 64;
  uint32_t* cnt = v.comp_cnt + (size_t)blockIdx.x * v.dim;
  uint32_t* pos = v.comp_pos + (size_t)blockIdx.x * v.dim;
  uint32_t* cur = v.comp_cur + (size_t)blockIdx.x * v.dim;
  uint32_t* touched = v.touched + (size_t)blockIdx.x * v.touched_cap;
  uint32_t* inv_cid = v.inv_cid + (size_t)blockIdx.x * v.inv_cap;
  float* inv_val = v.inv_val + (size_t)blockIdx.x * v.inv_cap;
  uint32_t* csize = v.csize + (size_t)blockIdx.x * kAssignMaxCentroids;
  uint32_t* ctr = v.counters + (size_t)blockIdx.x * 4;

  for (;;) {
    if (threadIdx.x == 0) {
      const uint32_t t = atomicAdd(v.queue, 1u);
      s_list = t < v.n_lists ? v.order[t] : 0xffffffffu;
    }
    __syncthreads();
    const uint32_t c = s_list;
    if (c == 0xffffffffu) break;
    const uint64_t p0 = v.lp_off[c];
    const uint32_t len = (uint32_t)(v.lp_off[c + 1] - p0);
    const uint32_t* cent = v.cent + v.lc_off[c];
    const uint32_t nc = (uint32_t)(v.lc_off[c + 1] - v.lc_off[c]);
    float* scores = (float*)smem + (size_t)wave * nc;
    for (uint32_t i = lane; i < nc; i += 64) scores[i] = 0.0f;
    for (uint32_t i = threadIdx.x; i < nc; i += kAssignThreads) csize[i] = 0;
    for (uint32_t i = threadIdx.x; i < kAssignMaxCentroids / 32; i += kAssignThreads) s_avoid[i] = 0;

    // ---- the centroids' inverted file: component -> [(centroid, value)] (order inside a component's
    // entries is irrelevant: a centroid occurs once per component)
    for (uint32_t cid = wave; cid < nc; cid += NW) {
      const uint32_t cd = cent[cid];
      for (uint64_t i = v.doc_off[cd] + lane; i < v.doc_off[cd + 1]; i += 64) {
        const uint32_t comp = comp_of(v, i);
        if (atomicAdd(&cnt[comp], 1u) == 0u) touched[atomicAdd(&ctr[0], 1u)] = comp;
      }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t n_touched = __hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t t = threadIdx.x; t < n_touched; t += kAssignThreads) {
      // (values produced by atomics are read past the L1; plain stores of this workgroup are visible to
      // its own plain loads after the barrier: one workgroup, one CU, one L1)
      const uint32_t comp = touched[t];
      const uint32_t n = __hip_atomic_load(&cnt[comp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pos[comp] = atomicAdd(&ctr[1], n);
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t cid = wave; cid < nc; cid += NW) {
      const uint32_t cd = cent[cid];
      for (uint64_t i = v.doc_off[cd] + lane; i < v.doc_off[cd + 1]; i += 64) {
        const uint32_t comp = comp_of(v, i);
        const uint32_t p = pos[comp] + atomicAdd(&cur[comp], 1u);
        inv_cid[p] = cid;
        inv_val[p] = half_to_float(v.doc_val[i]);
      }
    }
    __threadfence_block();
    __syncthreads();

    // ---- assignment: pass 0 every posting; pass 1 the postings of dissolved clusters, among the rest
    for (int pass = 0; pass < 2; ++pass) {
      for (uint32_t t = wave; t < len; t += NW) {
        if (pass == 1) {
          const uint32_t old = v.cid_out[p0 + t];
          if (!((s_avoid[old >> 5] >> (old & 31)) & 1u)) continue;
        }
        const uint32_t doc = v.post[p0 + t];
        const uint2* top = v.top + (size_t)doc * v.doc_cut;
        for (uint32_t i = 0; i < v.doc_cut; ++i) {
          const uint2 tc = top[i];
          if (tc.x == 0xffffffffu) break;
          const uint32_t n = __hip_atomic_load(&cnt[tc.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!n) continue;
          const uint32_t q0 = pos[tc.x];
          const float dv = __uint_as_float(tc.y);
          for (uint32_t e = lane; e < n; e += 64) {
            const uint32_t cid = inv_cid[q0 + e];
            const float cv = inv_val[q0 + e];
            // src/utils.rs:130: scores[centroid_id] += score * value  (product rounded, then the sum)
            scores[cid] = __fadd_rn(scores[cid], __fmul_rn(cv, dv));
          }
        }
        const uint32_t best = best_centroid(scores, pass ? s_avoid : nullptr, nc);
        for (uint32_t i = lane; i < nc; i += 64) scores[i] = 0.0f;
        if (lane == 0) {
          v.cid_out[p0 + t] = best;
          if (pass == 0) atomicAdd(&csize[best], 1u);
        }
      }
      __threadfence_block();
      __syncthreads();
      if (pass == 0) {
        // clusters of 1 .. min_cluster_size documents are dissolved (src/utils.rs:196-209); a centroid
        // nobody chose is not a group of the reference's chunk_by and stays available
        bool any = false;
        for (uint32_t i = threadIdx.x; i < nc; i += kAssignThreads) {
          const uint32_t n = __hip_atomic_load(&csize[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (n >= 1 && n <= v.min_cluster_size) {
            atomicOr(&s_avoid[i >> 5], 1u << (i & 31));
            any = true;
          }
        }
        if (!__syncthreads_or(any)) break;
      }
    }
    // ---- leave the scratch as it was found
    for (uint32_t t = threadIdx.x; t < n_touched; t += kAssignThreads) {
      const uint32_t comp = touched[t];
      cnt[comp] = 0;
      cur[comp] = 0;
    }
    if (threadIdx.x == 0) {
      ctr[0] = 0;
      ctr[1] = 0;
    }
    __threadfence_block();
    __syncthreads();
  }
}

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  template <class T>
  sgpu_status put(const T* src, size_t n) {
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(SGPU_ENOMEM, "hipMalloc of %zu bytes failed (index build)", bytes);
    if (src && n) HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return SGPU_OK;
  }
  sgpu_status zero(size_t bytes) {
    bytes = std::max<size_t>(bytes, 4);
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(SGPU_ENOMEM, "hipMalloc of %zu bytes failed (index build)", bytes);
    HIP_TRY(hipMemset(p, 0, bytes));
    return SGPU_OK;
  }
};
}  // namespace

// Largest number of centroids a list may have to be clustered on the device (others stay on the host): the kernel keeps
// one f32 accumulator per centroid and wavefront in dynamic LDS next to its static words, so the figure follows from the
// LDS a workgroup may have on THIS device (gfx950: 160 KB -> the full 8192; a 64 KB part: 3968). 0 = no usable device -
// device_assign_clusters then reports it.
uint32_t device_assign_max_centroids(int device) {
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || lds <= 0) {
    (void)hipGetLastError();   // (the runtime's sticky error must not surface at a later, unrelated hipGetLastError)
    return 0;
  }
  const size_t fixed = kAssignMaxCentroids / 32 * 4 + 256;   // s_avoid, s_list, alignment
  if ((size_t)lds <= fixed) return 0;
  const size_t fit = ((size_t)lds - fixed) / ((size_t)(kAssignThreads / 64) * 4);
  return (uint32_t)std::min<size_t>(kAssignMaxCentroids, fit);
}

// cid_out[lp_off[c] + t] = index (within list c's centroids) of the centroid posting t of list c belongs to,
// for every list with eligible[c] != 0. `top` holds doc_cut {component | ~0, f32 bits} pairs per document.
sgpu_status device_assign_clusters(int device, uint32_t comp_width, uint64_t n_docs, uint64_t dim, uint64_t nnz,
                                   const uint64_t* doc_off, const void* doc_comp, const uint16_t* doc_val,
                                   const void* top, uint32_t doc_cut, uint32_t min_cluster_size,
                                   const uint64_t* lp_off, const uint32_t* post, const uint64_t* lc_off,
                                   const uint32_t* cent, const uint8_t* eligible, uint64_t inv_cap,
                                   uint32_t* cid_out) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(SGPU_EDEVICE, "no HIP device available for the device-assisted index build");
  if (device < 0 || device >= n_dev) return fail(SGPU_EDEVICE, "device %d out of range (0..%d)", device, n_dev - 1);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  std::vector<uint32_t> order;
  uint32_t max_nc = 1;
  for (uint64_t c = 0; c < dim; ++c)
    if (eligible[c] && lp_off[c + 1] > lp_off[c]) {
      order.push_back((uint32_t)c);
      max_nc = std::max<uint32_t>(max_nc, (uint32_t)(lc_off[c + 1] - lc_off[c]));
    }
  if (order.empty()) return SGPU_OK;
  if (max_nc > device_assign_max_centroids(device))
    return fail(SGPU_EINVAL, "a list with %u centroids was marked for the device (it takes %u)", max_nc, device_assign_max_centroids(device));
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lp_off[a + 1] - lp_off[a] > lp_off[b + 1] - lp_off[b]; });
  const size_t lds = (size_t)(kAssignThreads / 64) * max_nc * 4;
  HIP_TRY(hipFuncSetAttribute((const void*)assign_clusters_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, assign_clusters_kernel, kAssignThreads, lds));
  if (per_cu < 1) return fail(SGPU_ELIMIT, "the cluster assignment kernel does not fit on a CU");
  const uint32_t grid = (uint32_t)std::min<size_t>(order.size(), (size_t)prop.multiProcessorCount * (size_t)std::min(per_cu, 8));
  const uint64_t touched_cap = std::min<uint64_t>(dim, std::max<uint64_t>(inv_cap, 1));

  DevBuf d_off, d_comp, d_val, d_top, d_lp, d_post, d_lc, d_cent, d_order, d_cid, d_cnt, d_pos, d_cur, d_touched, d_icid, d_ival,
      d_csize, d_ctr, d_queue;
  sgpu_status st;
  const uint64_t n_post = lp_off[dim], n_cent = lc_off[dim];
  if ((st = d_off.put(doc_off, n_docs + 1)) != SGPU_OK || (st = d_comp.put((const uint8_t*)doc_comp, nnz * comp_width)) != SGPU_OK ||
      (st = d_val.put(doc_val, nnz)) != SGPU_OK || (st = d_top.put((const uint2*)top, n_docs * doc_cut)) != SGPU_OK ||
      (st = d_lp.put(lp_off, dim + 1)) != SGPU_OK || (st = d_post.put(post, n_post)) != SGPU_OK ||
      (st = d_lc.put(lc_off, dim + 1)) != SGPU_OK || (st = d_cent.put(cent, n_cent)) != SGPU_OK ||
      (st = d_order.put(order.data(), order.size())) != SGPU_OK || (st = d_cid.zero(n_post * 4)) != SGPU_OK ||
      (st = d_cnt.zero((size_t)grid * dim * 4)) != SGPU_OK || (st = d_pos.zero((size_t)grid * dim * 4)) != SGPU_OK ||
      (st = d_cur.zero((size_t)grid * dim * 4)) != SGPU_OK || (st = d_touched.zero((size_t)grid * touched_cap * 4)) != SGPU_OK ||
      (st = d_icid.zero((size_t)grid * std::max<uint64_t>(inv_cap, 1) * 4)) != SGPU_OK ||
      (st = d_ival.zero((size_t)grid * std::max<uint64_t>(inv_cap, 1) * 4)) != SGPU_OK ||
      (st = d_csize.zero((size_t)grid * kAssignMaxCentroids * 4)) != SGPU_OK || (st = d_ctr.zero((size_t)grid * 16)) != SGPU_OK ||
      (st = d_queue.zero(256)) != SGPU_OK)
    return st;
  AssignView v{};
  v.doc_off = (const uint64_t*)d_off.p;
  v.doc_comp = d_comp.p;
  v.doc_val = (const uint16_t*)d_val.p;
  v.top = (const uint2*)d_top.p;
  v.comp_width = comp_width;
  v.doc_cut = doc_cut;
  v.dim = (uint32_t)dim;
  v.min_cluster_size = min_cluster_size;
  v.lp_off = (const uint64_t*)d_lp.p;
  v.post = (const uint32_t*)d_post.p;
  v.lc_off = (const uint64_t*)d_lc.p;
  v.cent = (const uint32_t*)d_cent.p;
  v.order = (const uint32_t*)d_order.p;
  v.n_lists = (uint32_t)order.size();
  v.cid_out = (uint32_t*)d_cid.p;
  v.comp_cnt = (uint32_t*)d_cnt.p;
  v.comp_pos = (uint32_t*)d_pos.p;
  v.comp_cur = (uint32_t*)d_cur.p;
  v.touched = (uint32_t*)d_touched.p;
  v.inv_cid = (uint32_t*)d_icid.p;
  v.inv_val = (float*)d_ival.p;
  v.csize = (uint32_t*)d_csize.p;
  v.counters = (uint32_t*)d_ctr.p;
  v.touched_cap = touched_cap;
  v.inv_cap = std::max<uint64_t>(inv_cap, 1);
  v.queue = (uint32_t*)d_queue.p;
  (void)hipGetLastError();   // (this launch is judged alone: an earlier failed call of the thread leaves its error behind)
  hipLaunchKernelGGL(assign_clusters_kernel, dim3(grid), dim3(kAssignThreads), lds, 0, v);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(cid_out, d_cid.p, n_post * 4, hipMemcpyDeviceToHost));
  return SGPU_OK;
}

}  // namespace sgpu
