// Checks that v_fma_mix_f32(q, half, -0.0) equals v_mul_f32(q, v_cvt_f32_f16(half)) bit for bit on gfx950
// (denormal weights, denormal products, f16 denormals, signed zeros).
// Measured on MI355X: 0 mismatches of 2097152.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__device__ inline float mix_lo(float q, uint32_t packed) {
  float r;
  asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(q), "v"(packed), "s"(0x80000000u));
  return r;
}
__device__ inline float mix_hi(float q, uint32_t packed) {
  float r;
  asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(q), "v"(packed), "s"(0x80000000u));
  return r;
}
__global__ void k(const float* q, const uint32_t* h, float* o_ref, float* o_mix, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = h[i];
  _Float16 lo, hi;
  uint16_t l16 = p & 0xffff, h16 = p >> 16;
  __builtin_memcpy(&lo, &l16, 2); __builtin_memcpy(&hi, &h16, 2);
  o_ref[2 * i] = __fmul_rn(q[i], (float)lo);
  o_ref[2 * i + 1] = __fmul_rn(q[i], (float)hi);
  o_mix[2 * i] = mix_lo(q[i], p);
  o_mix[2 * i + 1] = mix_hi(q[i], p);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> q(n); std::vector<uint32_t> h(n);
  uint64_t s = 999;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); };
  for (int i = 0; i < n; ++i) {
    uint32_t x = rnd(); h[i] = rnd();
    int kind = i & 7;
    if (kind == 0) x &= 0x807fffffu;                       // denormal weight
    if (kind == 1) x = (x & 0x80ffffffu) | 0x00800000u;    // tiny weight: denormal products
    if (kind == 2) h[i] &= 0x83ff83ffu;                    // f16 denormals
    if (kind == 3) x = (x & 0x8fffffffu) | 0x30000000u;
    if (kind == 4) { x = 0; }
    if (kind == 5) { x = 0x80000000u; }
    if (kind == 6) { h[i] &= 0x80008000u; }                // +-0 halves
    // no inf/nan halves
    if ((h[i] & 0x7c00u) == 0x7c00u) h[i] &= ~0x0400u;
    if ((h[i] & 0x7c000000u) == 0x7c000000u) h[i] &= ~0x04000000u;
    if ((x & 0x7f800000u) == 0x7f800000u) x &= ~0x00800000u;
    memcpy(&q[i], &x, 4);
  }
  float *dq, *d1, *d2; uint32_t* dh;
  hipMalloc(&dq, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dq, q.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dq, dh, d1, d2, n);
  std::vector<float> r1(2 * n), r2(2 * n);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 2 * n; ++i) {
    uint32_t u, v; memcpy(&u, &r1[i], 4); memcpy(&v, &r2[i], 4);
    if (u != v) { if (bad < 10) printf("mismatch kind %d: q %08x h %08x ref %08x mix %08x\n", (i / 2) & 7, *(uint32_t*)&q[i / 2], h[i / 2], u, v); bad++; }
  }
  printf("fma_mix_check: %d mismatches of %d\n", bad, 2 * n);
  return bad != 0;
}
