// plan_kernel.hip - the launch plan of a staged query chunk, computed ON THE DEVICE (r06).
//
// The search kernel wants two things per launch that depend on the queries: the processing order (longest expected
// query first: the a-priori cost is the postings of the lists a query will walk) and, for the LDS layout, the largest
// number of block dots a query needs. Until r05 the host computed both (make_plan, device_index.hip: ~0.2 us per query,
// serial on the calling thread: 2.2 ms of a 10 000-query sgpu_batch_search call next to a 5.7 ms kernel; VERDICT r05
// "one call, one thread, full rate"). The reference's batch_search is ONE call whose per-query cost does not depend on
// who calls it (src/pylib/mod.rs:1111-1146); here the plan moves behind the H2D copy of the queries:
//   plan_cost_kernel   one thread per query: the query_cut heaviest components by (f32::total_cmp descending,
//                      component ascending) - the rule of the search kernel's select_lists and of make_plan - their
//                      lists' postings (cost) and blocks; key = (~cost) << 32 | query; running maxima of the blocks a
//                      query needs (all its lists / its first list / any one list)
//   plan_sort_kernel   one workgroup: bitonic sort of the keys in LDS, ascending = longest expected first, ties in
//                      input order - the order make_plan's integer sort produces, bit for bit (tested)
// The LDS layout itself is sized on the host from the maxima of EARLIER chunks on the same index and query_cut (they
// travel back with the rows); a query that needs more block dots than that walks its lists in groups - slower for that
// query, identical results (search_kernel.inc: plan_list_group).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_types.hpp"

namespace sgpu {

__device__ __forceinline__ int32_t total_key_plan(float f) {   // Rust f32::total_cmp order (common.hpp: total_key)
  int32_t b = __float_as_int(f);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return b;
}

constexpr uint32_t kPlanCutMax = 16;   // (larger query_cut: the host plans)

__global__ __launch_bounds__(256) void plan_cost_kernel(const uint32_t* __restrict__ q_off, const uint32_t* __restrict__ q_comp,
                                                        const float* __restrict__ q_val, uint32_t nq, uint32_t cut,
                                                        const uint32_t* __restrict__ list_block_start,
                                                        const uint32_t* __restrict__ block_post_start, uint64_t* __restrict__ keys,
                                                        uint32_t n_keys, uint32_t* __restrict__ maxima) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_keys) return;
  if (q >= nq) {   // padding of the sort: after every real key
    keys[q] = ~0ull;
    return;
  }
  int32_t tk[kPlanCutMax];
  uint32_t tc[kPlanCutMax];
  uint32_t nl = 0;
  const uint32_t a = q_off[q], e = q_off[q + 1];
  for (uint32_t i = a; i < e; ++i) {   // components arrive ascending: of two equal keys the earlier one wins
    const int32_t key = total_key_plan(q_val[i]);
    if (nl == cut && !(key > tk[nl - 1])) continue;
    uint32_t j = nl < cut ? nl++ : nl - 1;
    while (j > 0 && tk[j - 1] < key) {
      tk[j] = tk[j - 1];
      tc[j] = tc[j - 1];
      --j;
    }
    tk[j] = key;
    tc[j] = q_comp[i];
  }
  uint64_t np = 0;
  uint32_t nb = 0, first_nb = 0, list_nb_max = 0;
  for (uint32_t i = 0; i < nl; ++i) {
    const uint32_t b0 = list_block_start[tc[i]], b1 = list_block_start[tc[i] + 1];
    nb += b1 - b0;
    np += (uint64_t)(block_post_start[b1] - block_post_start[b0]);
    list_nb_max = list_nb_max > b1 - b0 ? list_nb_max : b1 - b0;
    if (i == 0) first_nb = b1 - b0;
  }
  const uint32_t cost = np > 0xffffffffull ? 0xffffffffu : (uint32_t)np;
  keys[q] = ((uint64_t)(0xffffffffu - cost) << 32) | q;
  // (one atomic per wavefront and maximum would do; the kernel is a few microseconds either way)
  atomicMax(&maxima[0], nb);
  atomicMax(&maxima[1], first_nb);
  atomicMax(&maxima[2], list_nb_max);
}

__global__ __launch_bounds__(1024) void plan_sort_kernel(const uint64_t* __restrict__ keys, uint32_t n2, uint32_t nq,
                                                         uint32_t* __restrict__ order) {
  extern __shared__ uint64_t sk[];
  for (uint32_t i = threadIdx.x; i < n2; i += 1024) sk[i] = keys[i];
  __syncthreads();
  for (uint32_t size = 2; size <= n2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = threadIdx.x; t < n2 / 2; t += 1024) {
        const uint32_t i = 2 * t - (t & (stride - 1));
        const uint32_t j = i + stride;
        const bool up = (i & size) == 0;
        const uint64_t x = sk[i], y = sk[j];
        if ((x > y) == up) {
          sk[i] = y;
          sk[j] = x;
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = threadIdx.x; i < nq; i += 1024) order[i] = (uint32_t)sk[i];
}

// Enqueues the plan of a chunk on `stream`: keys_scratch holds n2 (a power of two >= nq, <= kDevicePlanMaxQueries)
// 64-bit keys; maxima = three zeroed words; order = nq words.
hipError_t launch_device_plan(const DevView& ix, const uint32_t* q_off, const uint32_t* q_comp, const float* q_val, uint32_t nq,
                              uint32_t cut, uint64_t* keys_scratch, uint32_t* maxima, uint32_t* order, hipStream_t stream) {
  uint32_t n2 = 2;
  while (n2 < nq) n2 <<= 1;
  if ((size_t)n2 * 8 > 48 * 1024) {   // (per device and cheap: no flag to keep)
    hipError_t e = hipFuncSetAttribute((const void*)plan_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kDevicePlanMaxQueries * 8));
    if (e != hipSuccess) return e;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(plan_cost_kernel, dim3((n2 + 255) / 256), dim3(256), 0, stream, q_off, q_comp, q_val, nq, cut,
                     ix.list_block_start, ix.block_post_start, keys_scratch, n2, maxima);
  hipLaunchKernelGGL(plan_sort_kernel, dim3(1), dim3(1024), (size_t)n2 * 8, stream, (const uint64_t*)keys_scratch, n2, nq, order);
  return hipGetLastError();
}

}  // namespace sgpu
