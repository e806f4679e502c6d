// hash_check.hip — the device's hash_slot_dev (v_mul_u32_u24 + shift) against the host's hash_slot, every
// component id below 2^17 under every multiplier of the family.  hipcc --offload-arch=gfx950 -O2 -I../../seismic_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "device_types.hpp"
using namespace sgpu;
__global__ void k(uint32_t* out, uint32_t n, uint32_t mult) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) out[c] = (uint32_t)__umul24(c, mult) >> (32 - kHashBits);   // (HIP declares __umul24 as returning int)
}
int main() {
  const uint32_t n = 1u << 17;
  uint32_t* d;
  if (hipMalloc(&d, n * 4) != hipSuccess) return 2;
  std::vector<uint32_t> h(n);
  uint64_t bad = 0;
  for (uint32_t s = 0; s < kHashSeeds; ++s) {
    const uint32_t m = hash_mult(s);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n, m);
    if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (uint32_t c = 0; c < n; ++c) bad += h[c] != hash_slot(c, m);
    if (s < 3) printf("seed %u mult %#x slot(12345) dev %u host %u\n", s, m, h[12345], hash_slot(12345, m));
  }
  printf("mismatches: %llu\n", (unsigned long long)bad);
  return bad != 0;
}
