// Micro-benchmark: ceiling for the search kernel's dominant access pattern — random reads of
// ~480-byte records (16 lanes x 16 B, comps then values) from a buffer larger than the caches.
// Build: hipcc --offload-arch=gfx950 -O3 -o random_record_read random_record_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(512) void rr(const uint8_t* buf, const uint32_t* idx, uint32_t n_items,
                                          uint32_t rec_bytes, uint32_t du, uint32_t* out) {
  const uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const uint32_t n_grp = (gridDim.x * blockDim.x) >> 4;
  uint32_t acc = 0;
  for (uint32_t i = grp; i < n_items; i += 4 * n_grp) {
    uint4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t iu = i + u * n_grp;
      if (iu < n_items && (uint32_t)u < du) {
        const uint8_t* rec = buf + (size_t)idx[iu] * 16;
        a[u] = *(const uint4*)(rec + sub * 16);
        b[u] = *(const uint4*)(rec + rec_bytes / 2 + sub * 16);
      } else {
        a[u] = b[u] = make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += a[u].x ^ a[u].w ^ b[u].y ^ b[u].z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t buf_bytes = (argc > 1 ? atol(argv[1]) : 480) * (1ull << 20);
  const uint32_t rec_bytes = argc > 2 ? atoi(argv[2]) : 480;
  const uint32_t align16 = argc > 3 ? atoi(argv[3]) : 1;   // alignment in 16-byte units (1 = 16 B, 8 = 128 B)
  const uint32_t n_items = 8u << 20;
  uint8_t* buf;
  uint32_t *idx, *out;
  hipMalloc(&buf, buf_bytes + 4096);
  hipMemset(buf, 1, buf_bytes + 4096);
  hipMalloc(&idx, n_items * 4);
  hipMalloc(&out, 4);
  std::vector<uint32_t> h(n_items);
  uint64_t s = 88172645463325252ull;
  const uint64_t slots = (buf_bytes - rec_bytes) / 16 / align16;
  for (auto& x : h) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    x = (uint32_t)((s % slots) * align16);
  }
  hipMemcpy(idx, h.data(), n_items * 4, hipMemcpyHostToDevice);
  for (uint32_t wg_per_cu : {2u, 4u}) {
    for (uint32_t du : {1u, 2u, 4u}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      rr<<<256 * wg_per_cu, 512>>>(buf, idx, n_items, rec_bytes, du, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      rr<<<256 * wg_per_cu, 512>>>(buf, idx, n_items, rec_bytes, du, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // with du < 4 only du/4 of the items are read
      const double items = (double)n_items * du / 4.0;
      printf("buf %zu MB rec %u B align %u B wg/cu %u docs-in-flight/group %u: %.3f ms  %.2f TB/s useful (%.1f M records/s)\n",
             buf_bytes >> 20, rec_bytes, align16 * 16, wg_per_cu, du, ms, items * rec_bytes / ms / 1e9,
             items / ms / 1e3);
    }
  }
  return 0;
}
