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
//                      query needs (all its lists / its first list / any one list), one atomic per wavefront
//   plan_rank_kernel   order[rank of a key among all keys] = query: ascending keys = longest expected first, ties in
//                      input order - the order make_plan's integer sort produces, bit for bit (tested). Keys are
//                      distinct (the query is their low half), so a rank is a count of smaller keys: 64 queries per
//                      workgroup, its four wavefronts count over a quarter of the keys each (a tile of 64 keys sits
//                      in one register pair across the lanes and is read back lane by lane).
// Both are grids of 256-thread workgroups without LDS to speak of. The first version (one 1024-thread workgroup sorting
// bitonically in 64 KB of LDS, three same-address atomics per query) took 205 us per 5000-query chunk on an empty chip
// (profiles/r06_entry_timeline.txt); these take 56. Beside a resident search launch - whose persistent workgroups hold
// every register of every CU - neither version runs before that launch's first workgroups leave: the next chunk's search
// starts in this one's tail (profiles/r06_entry_point_final.txt).
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

constexpr int kPlanCutMax = 16;   // (larger query_cut: the host plans)

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
    v = v > o ? v : o;
  }
  return v;
}

template <int CAP>   // (the kept components live in registers: 4, 8 or kPlanCutMax of them)
__global__ __launch_bounds__(256) void plan_cost_kernel(const uint32_t* __restrict__ q_off, const uint32_t* __restrict__ q_comp,
                                                        const float* __restrict__ q_val, uint32_t nq, uint32_t cut,
                                                        const uint32_t* __restrict__ list_block_start,
                                                        const uint32_t* __restrict__ block_post_start, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ maxima) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t nb = 0, first_nb = 0, list_nb_max = 0;
  if (q < nq) {
    // the heaviest components, descending: a new one goes behind every kept one that is not lighter (components arrive
    // ascending: of two equal keys the earlier one wins). Keys are widened so that no f32 key equals the filler.
    int64_t tk[CAP];
    uint32_t tc[CAP];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      tk[j] = INT64_MIN;
      tc[j] = 0;
    }
    const uint32_t a = q_off[q], e = q_off[q + 1];
    for (uint32_t i0 = a; i0 < e; i0 += 8) {   // eight components' loads in flight (a thread's walk is a chain of latencies)
      float v8[8];
      uint32_t c8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t i = i0 + u < e ? i0 + u : e - 1;
        v8[u] = q_val[i];
        c8[u] = q_comp[i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (i0 + u < e) {
          const int64_t key = (int64_t)total_key_plan(v8[u]);
          const uint32_t c = c8[u];
#pragma unroll
          for (int j = CAP - 1; j >= 0; --j) {
            const bool stays = tk[j] >= key;
            const bool here = j == 0 ? true : tk[j - 1] >= key;
            const int64_t up_k = j == 0 ? key : tk[j - 1];
            const uint32_t up_c = j == 0 ? c : tc[j - 1];
            tc[j] = stays ? tc[j] : (here ? c : up_c);
            tk[j] = stays ? tk[j] : (here ? key : up_k);
          }
        }
      }
    }
    const uint32_t nl = e - a < cut ? e - a : cut;
    uint64_t np = 0;
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
      if ((uint32_t)i < nl) {
        const uint32_t b0 = list_block_start[tc[i]], b1 = list_block_start[tc[i] + 1];
        nb += b1 - b0;
        np += (uint64_t)(block_post_start[b1] - block_post_start[b0]);
        list_nb_max = list_nb_max > b1 - b0 ? list_nb_max : b1 - b0;
        if (i == 0) first_nb = b1 - b0;
      }
    }
    const uint32_t cost = np > 0xffffffffull ? 0xffffffffu : (uint32_t)np;
    keys[q] = ((uint64_t)(0xffffffffu - cost) << 32) | q;
  }
  nb = wave_max_u32(nb);
  first_nb = wave_max_u32(first_nb);
  list_nb_max = wave_max_u32(list_nb_max);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&maxima[0], nb);
    atomicMax(&maxima[1], first_nb);
    atomicMax(&maxima[2], list_nb_max);
  }
}

__global__ __launch_bounds__(256) void plan_rank_kernel(const uint64_t* __restrict__ keys, uint32_t nq, uint32_t* __restrict__ order) {
  __shared__ uint32_t part[4][64];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 64 + lane;
  const uint64_t mine = q < nq ? keys[q] : 0ull;
  const uint32_t n_tiles = (nq + 63) >> 6;
  uint32_t cnt = 0;
  uint32_t t = wave;
  uint64_t cur = t < n_tiles && t * 64 + lane < nq ? keys[t * 64 + lane] : ~0ull;
  while (t < n_tiles) {
    const uint32_t tn = t + 4;
    const uint64_t nxt = tn < n_tiles && tn * 64 + lane < nq ? keys[tn * 64 + lane] : ~0ull;   // (in flight under the counting)
    const uint32_t lo = (uint32_t)cur, hi = (uint32_t)(cur >> 32);
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const uint64_t kj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, j);
      cnt += kj < mine ? 1u : 0u;
    }
    cur = nxt;
    t = tn;
  }
  part[wave][lane] = cnt;
  __syncthreads();
  if (wave == 0 && q < nq) order[part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]] = (uint32_t)mine;
}

// Enqueues the plan of a chunk on `stream`: keys_scratch holds nq 64-bit keys; maxima = three zeroed words; order = nq words.
hipError_t launch_device_plan(const DevView& ix, const uint32_t* q_off, const uint32_t* q_comp, const float* q_val, uint32_t nq,
                              uint32_t cut, uint64_t* keys_scratch, uint32_t* maxima, uint32_t* order, hipStream_t stream) {
  (void)hipGetLastError();
  const dim3 grid((nq + 255) / 256), block(256);
  if (cut <= 4)
    hipLaunchKernelGGL(plan_cost_kernel<4>, grid, block, 0, stream, q_off, q_comp, q_val, nq, cut, ix.list_block_start, ix.block_post_start, keys_scratch, maxima);
  else if (cut <= 8)
    hipLaunchKernelGGL(plan_cost_kernel<8>, grid, block, 0, stream, q_off, q_comp, q_val, nq, cut, ix.list_block_start, ix.block_post_start, keys_scratch, maxima);
  else
    hipLaunchKernelGGL(plan_cost_kernel<kPlanCutMax>, grid, block, 0, stream, q_off, q_comp, q_val, nq, cut, ix.list_block_start, ix.block_post_start, keys_scratch, maxima);
  hipLaunchKernelGGL(plan_rank_kernel, dim3((nq + 63) / 64), dim3(256), 0, stream, (const uint64_t*)keys_scratch, nq, order);
  return hipGetLastError();
}

}  // namespace sgpu
