// common.hpp — host-side utilities shared by the product's translation units.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/seismic_hip.h"

namespace sgpu {

// thread-local error text behind sgpu_last_error()
std::string& last_error();
sgpu_status fail(sgpu_status st, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Threads a host-parallel phase uses when the caller passes num_threads == 0 ("all cores", the reference's rayon
// default): the hardware threads this process may run on, capped by the container's CPU quota (cgroup cpu.max /
// cfs_quota_us) - a team larger than the quota is descheduled for most of every accounting period. SGPU_HOST_THREADS
// overrides.
int default_host_threads(int omp_max_threads);
}  // namespace sgpu
#ifdef _OPENMP
#include <omp.h>
#endif
namespace sgpu {
inline int host_threads() {
#ifdef _OPENMP
  return default_host_threads(omp_get_max_threads());
#else
  return 1;
#endif
}

// ---- binary16 (document values are half::f16 in the reference:
// src/index_traits.rs:57-142) ---------------------------------------------
inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t em = h & 0x7fffu;
  uint32_t bits;
  if (em >= 0x7c00u) {
    bits = sign | 0x7f800000u | ((em & 0x3ffu) << 13);
  } else if (em >= 0x0400u) {
    bits = sign | ((em << 13) + 0x38000000u);
  } else if (em == 0) {
    bits = sign;
  } else {  // subnormal half = em * 2^-24, exactly representable
    float f = (float)em * 5.9604644775390625e-08f;
    uint32_t fb;
    std::memcpy(&fb, &f, 4);
    bits = sign | fb;
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// round-to-nearest-even, finite overflow saturates to +-65504 (from_f32_saturating
// call site: reference src/json_utils.rs:64); NaN propagates.
inline uint16_t f32_to_f16_sat(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  x &= 0x7fffffffu;
  if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);
  if (x < 0x38800000u) {  // result is subnormal or zero: scale by 2^24 and round to integer
    float a;
    std::memcpy(&a, &x, 4);
    // a * 2^24 is exact in f32 (power-of-two scaling); nearbyint rounds half to even
    float r = __builtin_nearbyintf(a * 16777216.0f);
    return (uint16_t)(sign | (uint16_t)r);
  }
  // normal: add rounding bias then truncate (ties to even)
  const uint32_t lsb = (x >> 13) & 1u;
  x += 0xfffu + lsb;
  return (uint16_t)(sign | (uint16_t)((x - 0x38000000u) >> 13));
}

// Rust f32::total_cmp order as an integer key
inline int32_t total_key(float f) {
  int32_t b;
  std::memcpy(&b, &f, 4);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return b;
}

// descending-numeric sort key of a half value (bigger value -> smaller key);
// +0 and -0 compare equal, as partial_cmp does.
inline uint32_t f16_desc_key(uint16_t h) {
  if ((h & 0x7fffu) == 0) h = 0;
  uint32_t k = (h & 0x8000u) ? (uint32_t)(0x8000u - (h & 0x7fffu)) : (uint32_t)(0x8000u + h);
  return 0x10000u - k;  // ascending key order == descending value order
}

// Stand-in for rand::StdRng (not reproducible without the crate).
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  inline uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  inline uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
  inline double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
};

}  // namespace sgpu
