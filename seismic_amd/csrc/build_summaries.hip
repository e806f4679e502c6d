// build_summaries.hip — index build, the per-block summaries on the GPU.
//
// energy_preserving_summary (reference src/posting_list.rs:329-368) + quantize (src/utils.rs:68-90) for
// every block of every posting list: the component-wise maximum over the block's documents, the
// components taken in descending value until their running sum reaches summary_energy of the total
// (the crossing one included), re-sorted by component, quantised to u8 with the block's minimum and
// step. After the clustering moved to the device (build_assign.hip) this was 15 of the remaining 23
// seconds of an 8.8M-document build on the host cores.
//
// One block per workgroup (persistent workgroups pull blocks from a queue): the block's document
// entries are gathered into LDS as 64-bit keys and sorted three times with a bitonic network
// (by component to find the maxima, by value for the energy cut, by component for the output). The two
// running sums are SEQUENTIAL f32 additions in the reference and stay sequential here (one thread),
// so the kept set, the minimum, the step and every code are identical to the host builder's; the
// divisions and roundf are IEEE-correct on the device. Blocks with more entries than the LDS buffers
// hold (a few per cent) are left to the host.
//
// Known difference (documented, not reachable with non-negative weights): the maximum of a component
// is taken by f32::total_cmp here and by `<` on the host; they differ only when a block holds both
// +0.0 and -0.0 for one component.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "build_device.hpp"

namespace sgpu {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(SGPU_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

constexpr int kSumThreads = 256;
constexpr uint32_t kSumCap = 4096;   // entries of a block: two key buffers of 32 KB in LDS

struct SumView {
  const uint64_t* doc_off;
  const void* doc_comp;
  const uint16_t* doc_val;
  uint32_t comp_width;
  float energy;
  const uint64_t* blk_post;     // n_blocks + 1
  const uint32_t* blk_entries;
  const uint32_t* post;
  const uint32_t* work;         // block ids of this launch
  uint32_t n_work;
  // per block (indexed by block id)
  unsigned long long* out_start;
  uint32_t* out_keep;
  float* out_mn;
  float* out_qt;
  // entries of this launch
  uint32_t* out_comp;
  uint8_t* out_code;
  unsigned long long* cursor;
  uint32_t* queue;
};

__device__ __forceinline__ uint32_t okey(float f) {   // f32::total_cmp order as an unsigned key
  int32_t b = __float_as_int(f);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return (uint32_t)b ^ 0x80000000u;
}
__device__ __forceinline__ float okey_inv(uint32_t k) {
  int32_t b = (int32_t)(k ^ 0x80000000u);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return __int_as_float(b);
}
__device__ __forceinline__ float h2f(uint16_t h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return (float)x;
}

// ascending bitonic sort of keys[0 .. p2), p2 a power of two, by the whole workgroup
__device__ void bitonic(unsigned long long* keys, uint32_t p2) {
  for (uint32_t size = 2; size <= p2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = threadIdx.x; t < p2 / 2; t += kSumThreads) {
        const uint32_t i = 2 * t - (t & (stride - 1));
        const uint32_t j = i + stride;
        const bool up = (i & size) == 0;
        const unsigned long long a = keys[i], b = keys[j];
        if ((a > b) == up) {
          keys[i] = b;
          keys[j] = a;
        }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(kSumThreads) void block_summaries_kernel(SumView v) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long smem_keys[];   // two buffers of kSumCap keys
  unsigned long long* A = smem_keys;
  unsigned long long* B = smem_keys + kSumCap;
  __shared__ uint32_t s_blk, s_cnt, s_part[kSumThreads / 64 + 1], s_keep, s_kmin, s_kmax;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    if (threadIdx.x == 0) {
      const uint32_t t = atomicAdd(v.queue, 1u);
      s_blk = t < v.n_work ? v.work[t] : 0xffffffffu;
      s_cnt = 0;
      s_kmin = 0xffffffffu;
      s_kmax = 0;
    }
    __syncthreads();
    const uint32_t b = s_blk;
    if (b == 0xffffffffu) break;
    const uint64_t p0 = v.blk_post[b], p1 = v.blk_post[b + 1];
    const uint32_t n = v.blk_entries[b];
    uint32_t p2 = 1;
    while (p2 < n) p2 <<= 1;
    // ---- gather: (component << 32) | ~key(value): ascending order = component, then value descending
    for (uint64_t t = p0 + threadIdx.x; t < p1; t += kSumThreads) {
      const uint32_t doc = v.post[t];
      const uint64_t e0 = v.doc_off[doc], e1 = v.doc_off[doc + 1];
      uint32_t pos = atomicAdd(&s_cnt, (uint32_t)(e1 - e0));
      for (uint64_t i = e0; i < e1; ++i, ++pos) {
        const uint32_t c = v.comp_width == 2 ? (uint32_t)((const uint16_t*)v.doc_comp)[i] : ((const uint32_t*)v.doc_comp)[i];
        A[pos] = ((unsigned long long)c << 32) | (unsigned long long)(~okey(h2f(v.doc_val[i])));
      }
    }
    for (uint32_t i = n + threadIdx.x; i < p2; i += kSumThreads) A[i] = ~0ull;
    __syncthreads();
    bitonic(A, p2);
    // ---- the maximum of every component (head of its run) -> B as (~key(value) << 32) | component
    constexpr uint32_t PER = kSumCap / kSumThreads;   // consecutive elements per thread
    const uint32_t i0 = threadIdx.x * PER;
    uint32_t heads = 0;
    for (uint32_t i = i0; i < i0 + PER && i < n; ++i) heads += (i == 0 || (uint32_t)(A[i] >> 32) != (uint32_t)(A[i - 1] >> 32));
    uint32_t incl = heads;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if ((int)lane >= d) incl += o;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t base = 0, m = 0;
    for (uint32_t w = 0; w < kSumThreads / 64; ++w) {
      if (w < wave) base += s_part[w];
      m += s_part[w];
    }
    {
      uint32_t o = base + incl - heads;
      for (uint32_t i = i0; i < i0 + PER && i < n; ++i)
        if (i == 0 || (uint32_t)(A[i] >> 32) != (uint32_t)(A[i - 1] >> 32))
          B[o++] = ((A[i] & 0xffffffffull) << 32) | (A[i] >> 32);
    }
    uint32_t q2 = 1;
    while (q2 < m) q2 <<= 1;
    __syncthreads();
    for (uint32_t i = m + threadIdx.x; i < q2; i += kSumThreads) B[i] = ~0ull;
    __syncthreads();
    bitonic(B, q2);   // value descending (by total_cmp), component ascending: the reference's sort (src/posting_list.rs:347-351)
    // ---- the energy cut: two SEQUENTIAL running sums, as the reference computes them
    if (threadIdx.x == 0) {
      float tot = 0.0f;
      for (uint32_t i = 0; i < m; ++i) tot = __fadd_rn(tot, okey_inv(~(uint32_t)(B[i] >> 32)));
      const float until = __fmul_rn(tot, v.energy);
      float acc = 0.0f;
      uint32_t keep = 0;
      while (keep < m) {   // take_while_inclusive
        acc = __fadd_rn(acc, okey_inv(~(uint32_t)(B[keep] >> 32)));
        ++keep;
        if (!(acc < until)) break;
      }
      s_keep = keep;
    }
    __syncthreads();
    const uint32_t keep = s_keep;
    // ---- kept entries by component: (component << 32) | value bits -> A; minimum / maximum by total_cmp
    uint32_t kmin = 0xffffffffu, kmax = 0;
    for (uint32_t i = threadIdx.x; i < keep; i += kSumThreads) {
      const uint32_t k = ~(uint32_t)(B[i] >> 32);
      A[i] = ((B[i] & 0xffffffffull) << 32) | (unsigned long long)__float_as_uint(okey_inv(k));
      kmin = k < kmin ? k : kmin;
      kmax = k > kmax ? k : kmax;
    }
    uint32_t r2 = 1;
    while (r2 < keep) r2 <<= 1;
    for (uint32_t i = keep + threadIdx.x; i < r2; i += kSumThreads) A[i] = ~0ull;
    atomicMin(&s_kmin, kmin);
    atomicMax(&s_kmax, kmax);
    __syncthreads();
    bitonic(A, r2);
    const float mn = okey_inv(s_kmin), mx = okey_inv(s_kmax);
    const float quant = __fdiv_rn(__fsub_rn(mx, mn), 255.0f);   // src/utils.rs:75
    __shared__ unsigned long long s_pos;
    if (threadIdx.x == 0) {
      s_pos = atomicAdd(v.cursor, (unsigned long long)keep);
      v.out_start[b] = s_pos;
      v.out_keep[b] = keep;
      v.out_mn[b] = mn;
      v.out_qt[b] = quant;
    }
    __syncthreads();
    const unsigned long long pos = s_pos;
    for (uint32_t i = threadIdx.x; i < keep; i += kSumThreads) {
      const float val = __uint_as_float((uint32_t)A[i]);
      // ((v - min) / quant).round() as u8: half away from zero, saturating, NaN -> 0 (src/utils.rs:80-86)
      const float r = __builtin_roundf(__fdiv_rn(__fsub_rn(val, mn), quant));
      const uint8_t code = (r != r) ? 0 : (r <= 0.0f ? 0 : (r >= 255.0f ? 255 : (uint8_t)r));
      v.out_comp[pos + i] = (uint32_t)(A[i] >> 32);
      v.out_code[pos + i] = code;
    }
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
  sgpu_status raw(size_t bytes) {
    bytes = std::max<size_t>(bytes, 16);
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(SGPU_ENOMEM, "hipMalloc of %zu bytes failed (index build)", bytes);
    return SGPU_OK;
  }
};
}  // namespace

uint32_t device_summary_max_entries() { return kSumCap; }

sgpu_status device_block_summaries(int device, uint32_t comp_width, uint64_t n_docs, uint64_t nnz, const uint64_t* doc_off,
                                   const void* doc_comp, const uint16_t* doc_val, float summary_energy, uint64_t n_blocks,
                                   const uint64_t* blk_post, const uint32_t* blk_entries, const uint32_t* post,
                                   DeviceSummaries* out) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(SGPU_EDEVICE, "no HIP device available for the device-assisted index build");
  if (device < 0 || device >= n_dev) return fail(SGPU_EDEVICE, "device %d out of range (0..%d)", device, n_dev - 1);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  try {
    out->done.assign(n_blocks, 0);
    out->keep.assign(n_blocks, 0);
    out->mn.assign(n_blocks, 0.0f);
    out->qt.assign(n_blocks, 0.0f);
    out->start.assign(n_blocks, 0);
    // launches: runs of blocks whose entries add up to at most kChunk (bounds the output buffers)
    constexpr uint64_t kChunk = 256ull << 20;
    std::vector<uint32_t> work;
    std::vector<std::pair<size_t, size_t>> launches;   // [first, last) into work
    std::vector<uint64_t> launch_entries;
    {
      uint64_t run = 0;
      size_t first = 0;
      for (uint64_t b = 0; b < n_blocks; ++b) {
        if (blk_entries[b] == 0 || blk_entries[b] > kSumCap) continue;
        if (run + blk_entries[b] > kChunk) {
          launches.emplace_back(first, work.size());
          launch_entries.push_back(run);
          first = work.size();
          run = 0;
        }
        work.push_back((uint32_t)b);
        run += blk_entries[b];
      }
      if (work.size() > first) {
        launches.emplace_back(first, work.size());
        launch_entries.push_back(run);
      }
    }
    if (work.empty()) return SGPU_OK;
    const uint64_t max_entries = *std::max_element(launch_entries.begin(), launch_entries.end());
    const uint64_t n_post = blk_post[n_blocks];
    DevBuf d_off, d_comp, d_val, d_bp, d_be, d_post, d_work, d_start, d_keep, d_mn, d_qt, d_oc, d_oq, d_cur, d_queue;
    sgpu_status st;
    if ((st = d_off.put(doc_off, n_docs + 1)) != SGPU_OK || (st = d_comp.put((const uint8_t*)doc_comp, nnz * comp_width)) != SGPU_OK ||
        (st = d_val.put(doc_val, nnz)) != SGPU_OK || (st = d_bp.put(blk_post, n_blocks + 1)) != SGPU_OK ||
        (st = d_be.put(blk_entries, n_blocks)) != SGPU_OK || (st = d_post.put(post, n_post)) != SGPU_OK ||
        (st = d_work.put(work.data(), work.size())) != SGPU_OK || (st = d_start.raw(n_blocks * 8)) != SGPU_OK ||
        (st = d_keep.raw(n_blocks * 4)) != SGPU_OK || (st = d_mn.raw(n_blocks * 4)) != SGPU_OK ||
        (st = d_qt.raw(n_blocks * 4)) != SGPU_OK || (st = d_oc.raw(max_entries * 4)) != SGPU_OK ||
        (st = d_oq.raw(max_entries)) != SGPU_OK || (st = d_cur.raw(16)) != SGPU_OK || (st = d_queue.raw(16)) != SGPU_OK)
      return st;
    int per_cu = 0;
    constexpr size_t kLds = 2 * (size_t)kSumCap * 8;
    HIP_TRY(hipFuncSetAttribute((const void*)block_summaries_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, block_summaries_kernel, kSumThreads, kLds));
    if (per_cu < 1) return fail(SGPU_ELIMIT, "the block summary kernel does not fit on a CU");
    SumView v{};
    v.doc_off = (const uint64_t*)d_off.p;
    v.doc_comp = d_comp.p;
    v.doc_val = (const uint16_t*)d_val.p;
    v.comp_width = comp_width;
    v.energy = summary_energy;
    v.blk_post = (const uint64_t*)d_bp.p;
    v.blk_entries = (const uint32_t*)d_be.p;
    v.post = (const uint32_t*)d_post.p;
    v.out_start = (unsigned long long*)d_start.p;
    v.out_keep = (uint32_t*)d_keep.p;
    v.out_mn = (float*)d_mn.p;
    v.out_qt = (float*)d_qt.p;
    v.out_comp = (uint32_t*)d_oc.p;
    v.out_code = (uint8_t*)d_oq.p;
    v.cursor = (unsigned long long*)d_cur.p;
    v.queue = (uint32_t*)d_queue.p;
    uint64_t total = 0;   // kept entries so far: launch l's entries land at [total, total + used_l)
    std::vector<uint64_t> launch_base;
    for (size_t l = 0; l < launches.size(); ++l) {
      const size_t first = launches[l].first, last = launches[l].second;
      v.work = (const uint32_t*)d_work.p + first;
      v.n_work = (uint32_t)(last - first);
      HIP_TRY(hipMemset(d_cur.p, 0, 16));
      HIP_TRY(hipMemset(d_queue.p, 0, 16));
      const uint32_t grid = (uint32_t)std::min<size_t>(last - first, (size_t)prop.multiProcessorCount * (size_t)per_cu);
      (void)hipGetLastError();   // (this launch is judged alone: an earlier failed call of the thread leaves its error behind)
      hipLaunchKernelGGL(block_summaries_kernel, dim3(grid), dim3(kSumThreads), kLds, 0, v);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      unsigned long long used = 0;
      HIP_TRY(hipMemcpy(&used, d_cur.p, 8, hipMemcpyDeviceToHost));
      if (used > launch_entries[l]) return fail(SGPU_EDEVICE, "block summary kernel wrote past its buffer");
      out->comp.resize(total + used);
      out->code.resize(total + used);
      if (used) {
        HIP_TRY(hipMemcpy(out->comp.data() + total, d_oc.p, used * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out->code.data() + total, d_oq.p, used, hipMemcpyDeviceToHost));
      }
      launch_base.push_back(total);
      total += used;
    }
    HIP_TRY(hipMemcpy(out->start.data(), d_start.p, n_blocks * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->keep.data(), d_keep.p, n_blocks * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->mn.data(), d_mn.p, n_blocks * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->qt.data(), d_qt.p, n_blocks * 4, hipMemcpyDeviceToHost));
    for (size_t l = 0; l < launches.size(); ++l)
      for (size_t w = launches[l].first; w < launches[l].second; ++w) {
        const uint32_t b = work[w];
        out->start[b] += launch_base[l];
        out->done[b] = 1;
      }
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory in the device-assisted index build");
  }
  return SGPU_OK;
}

}  // namespace sgpu
