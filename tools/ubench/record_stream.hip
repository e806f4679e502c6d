// Micro-benchmark: do boundary cache lines shared by ADJACENT records get fetched once or twice?
// Records of `rec_bytes` (multiple of 32: comps then values, as the search kernel's document records)
// laid out back to back; every record is read exactly once by a 16-lane group, two 16-byte loads per
// lane (component half, value half), four records in flight per group.
//   mode 0: group-consecutive - a group takes 4 consecutive records (the four groups of a wavefront
//           read records base+0..3, base+4..7, ...; adjacent records meet in DIFFERENT instructions)
//   mode 1: wave-interleaved  - slot u of group g reads record base + 4u + g (one instruction covers
//           four adjacent records)
//   mode 2: random            - records in random order (the document-major pattern)
// Build: hipcc --offload-arch=gfx950 -O3 -o record_stream record_stream.hip ; run under
// rocprofv3 --pmc FETCH_SIZE to see the fabric bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(512) void rs(const uint8_t* buf, const uint32_t* perm, uint32_t n_rec, uint32_t rec_bytes,
                                          int mode, uint32_t* out) {
  const uint32_t sub = threadIdx.x & 15, grp = (threadIdx.x & 63) >> 4;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (uint32_t base = wave * 16; base + 16 <= n_rec; base += n_waves * 16) {
    uint4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint32_t r = mode == 0 ? base + grp * 4 + u : base + u * 4 + grp;
      if (mode == 2) r = perm[r];
      const uint8_t* rec = buf + (size_t)r * rec_bytes;
      const uint32_t half = rec_bytes / 2;
      const uint32_t o = sub * 16 < half ? sub * 16 : 0;
      a[u] = *(const uint4*)(rec + o);
      b[u] = *(const uint4*)(rec + half + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += a[u].x ^ a[u].w ^ b[u].y ^ b[u].z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const uint32_t rec_bytes = argc > 1 ? atoi(argv[1]) : 480;
  const size_t buf_bytes = (argc > 2 ? atol(argv[2]) : 4096) * (1ull << 20);
  const uint32_t n_rec = (uint32_t)(buf_bytes / rec_bytes) & ~15u;
  uint8_t* buf;
  uint32_t *perm, *out;
  hipMalloc(&buf, buf_bytes + 4096);
  hipMemset(buf, 1, buf_bytes + 4096);
  hipMalloc(&perm, (size_t)n_rec * 4);
  hipMalloc(&out, 4);
  std::vector<uint32_t> h(n_rec);
  for (uint32_t i = 0; i < n_rec; ++i) h[i] = i;
  uint64_t s = 88172645463325252ull;
  for (uint32_t i = n_rec - 1; i > 0; --i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    std::swap(h[i], h[(uint32_t)(s % (i + 1))]);
  }
  hipMemcpy(perm, h.data(), (size_t)n_rec * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rs<<<512, 512>>>(buf, perm, n_rec, rec_bytes, mode, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rs<<<512, 512>>>(buf, perm, n_rec, rec_bytes, mode, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("rec %u B, %u records (%.2f GB), mode %d: %.3f ms  %.2f TB/s useful\n", rec_bytes, n_rec,
           (double)n_rec * rec_bytes / 1e9, mode, ms, (double)n_rec * rec_bytes / ms / 1e9);
  }
  return 0;
}
