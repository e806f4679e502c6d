// hits_only.hip - would it pay to read the query weight only for the document components that hit the query?
//
// The product's dense-table scoring loop reads, per document component, one byte of the query index (random address) and
// one weight (q_sc[rank - 1], 0.0 for a miss): two DS instructions per component, the wall of profiles/r03_lds_sensitivity.md.
// The alternative measured here keeps the eight byte reads of a slice and then walks only the non-zero bytes of the lane
// (lowest element first, so the canonical accumulation order is kept and skipping the +0.0 terms is exact): one weight
// read per HIT, at the price of the walk (find-first-set, variable shifts to fetch rank and value, a loop the whole
// wavefront stays in until its busiest lane is done). Same isolated loop as gap12.hip (16 lanes per document, four
// documents in flight, 2M documents gathered at random); `hits` query components are planted into every document on
// average so that the hit density can be set (0.15 per document by chance; the product's scored documents have several).
// Scores must be identical between the two loops (checked).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o hits_only hits_only.hip && ./hits_only [n_docs] [n_visits] [hits per document]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr uint32_t kDim = 30522, kTable = 32768, kNT = 512;

struct Visit { uint32_t off4; uint32_t len_first; };   // record offset in 4-byte units; len | first component << 16

static inline uint16_t f32_to_f16(float f) {   // positive normal values only (the generator's range)
  uint32_t u; memcpy(&u, &f, 4);
  uint32_t e = ((u >> 23) & 0xff) - 127 + 15, m = (u >> 13) & 0x3ff;
  return (uint16_t)((e << 10) | m);
}

__device__ inline int dpp_shr(int x, int n) {
  switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    case 4: return __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    default: return __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
  }
}
__device__ inline float dpp_ror(float v, int n) {
  int x = __float_as_int(v), r;
  switch (n) {
    case 8: r = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x122, 0xf, 0xf, false); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x121, 0xf, 0xf, false); break;
  }
  return __int_as_float(r);
}

template <bool G12, bool U8>
struct Slice {
  uint32_t c[G12 ? 3 : 4];
  uint32_t v[U8 ? 2 : 4];
};

template <bool G12, bool U8>
__device__ inline void load_slice(Slice<G12, U8>& s, const uint32_t* rec, uint32_t j, bool valid) {
  constexpr uint32_t CW = G12 ? 3 : 4, VW = U8 ? 2 : 4, SW = CW + VW;
#pragma unroll
  for (uint32_t i = 0; i < CW; ++i) s.c[i] = 0;
#pragma unroll
  for (uint32_t i = 0; i < VW; ++i) s.v[i] = 0;
  if (!G12) {
#pragma unroll
    for (uint32_t i = 0; i < CW; ++i) s.c[i] = kDim | (kDim << 16);   // sentinel: its byte in the table is always 0
  }
  if (valid) {
    const uint32_t* p = rec + (size_t)j * SW;
#pragma unroll
    for (uint32_t i = 0; i < CW; ++i) s.c[i] = p[i];
#pragma unroll
    for (uint32_t i = 0; i < VW; ++i) s.v[i] = p[CW + i];
  }
}

template <bool U8>
__device__ inline float val_at(const uint32_t* v, int i) {
  if (U8) return (float)((v[i >> 2] >> (8 * (i & 3))) & 0xffu);
  const uint32_t h = (v[i >> 1] >> (16 * (i & 1))) & 0xffffu;
  _Float16 x;
  const unsigned short hs = (unsigned short)h;
  __builtin_memcpy(&x, &hs, 2);
  return (float)x;
}

// scores one pass (up to 16 slices) of one document; `base` = the component before this pass's first element
template <bool G12, bool U8, bool HITS>
__device__ inline float score_slice(const Slice<G12, U8>& s, const uint8_t* q_idx, const float* q_sc, uint32_t& base, float acc) {
  uint32_t c[8];
  if constexpr (G12) {
    const uint32_t w0 = s.c[0], w1 = s.c[1], w2 = s.c[2];
    uint32_t g[8];
    g[0] = w0 & 0xfffu; g[1] = (w0 >> 12) & 0xfffu; g[2] = ((w0 >> 24) | (w1 << 8)) & 0xfffu; g[3] = (w1 >> 4) & 0xfffu;
    g[4] = (w1 >> 16) & 0xfffu; g[5] = ((w1 >> 28) | (w2 << 4)) & 0xfffu; g[6] = (w2 >> 8) & 0xfffu; g[7] = w2 >> 20;
    const uint32_t t = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
    int x = (int)t;
    x += dpp_shr(x, 1); x += dpp_shr(x, 2); x += dpp_shr(x, 4); x += dpp_shr(x, 8);
    uint32_t run = base + (uint32_t)x - t;
#pragma unroll
    for (int i = 0; i < 8; ++i) { run += g[i]; c[i] = run; }
    base += (uint32_t)__shfl(x, 15, 16);   // (only multi-pass documents use it)
  } else {
    c[0] = s.c[0] & 0xffffu; c[1] = s.c[0] >> 16; c[2] = s.c[1] & 0xffffu; c[3] = s.c[1] >> 16;
    c[4] = s.c[2] & 0xffffu; c[5] = s.c[2] >> 16; c[6] = s.c[3] & 0xffffu; c[7] = s.c[3] >> 16;
  }
  uint32_t r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = q_idx[c[i]];
  if constexpr (HITS) {
    // the eight ranks as bytes of two dwords; bit 7 of every non-zero byte marks a hit
    const uint32_t rlo = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24), rhi = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
    const uint32_t mlo = (rlo | ((rlo & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    const uint32_t mhi = (rhi | ((rhi & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    uint64_t hits = ((uint64_t)mhi << 32) | mlo;
    const uint64_t ranks = ((uint64_t)rhi << 32) | rlo;
    while (hits) {
      const int sh = __ffsll((unsigned long long)hits) - 8;   // 8 * element
      hits &= hits - 1;
      const uint32_t ri = (uint32_t)(ranks >> sh) & 0xffu;
      const float qw = q_sc[(int)ri - 1];
      float v;
      if (U8) {
        const uint64_t codes = ((uint64_t)s.v[1] << 32) | s.v[0];
        v = (float)((uint32_t)(codes >> sh) & 0xffu);
      } else {
        const uint64_t v01 = ((uint64_t)s.v[1] << 32) | s.v[0], v23 = ((uint64_t)s.v[U8 ? 1 : 3] << 32) | s.v[U8 ? 0 : 2];
        const uint64_t w = (sh & 32) ? v23 : v01;                      // element >= 4
        const unsigned short hs = (unsigned short)(w >> ((sh & 24) * 2));   // 16 * (element & 3)
        _Float16 x;
        __builtin_memcpy(&x, &hs, 2);
        v = (float)x;
      }
      acc = __fadd_rn(acc, __fmul_rn(qw, v));
    }
    return acc;
  }
  float q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = q_sc[(int)r[i] - 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = __fadd_rn(acc, __fmul_rn(q[i], val_at<U8>(s.v, i)));
  return acc;
}

template <bool G12, bool U8, bool HITS>
__global__ __launch_bounds__(kNT) void score_kernel(const uint32_t* __restrict__ recs, const Visit* __restrict__ visits, uint32_t n_visits,
                                                    const uint32_t* __restrict__ q_comp, const float* __restrict__ q_val, uint32_t nnz,
                                                    float* __restrict__ out) {
  __shared__ uint8_t q_idx[kTable];
  __shared__ float q_sc_[257];
  float* q_sc = q_sc_ + 1;
  for (uint32_t i = threadIdx.x; i < kTable / 4; i += kNT) ((uint32_t*)q_idx)[i] = 0;
  if (threadIdx.x == 0) q_sc_[0] = 0.0f;
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < nnz; j += kNT) { q_idx[q_comp[j]] = (uint8_t)(j + 1); q_sc[j] = q_val[j]; }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 15u, grp = (blockIdx.x * kNT + threadIdx.x) >> 4, n_grp = (gridDim.x * kNT) >> 4;
  for (uint32_t v0 = grp * 4; v0 < n_visits; v0 += n_grp * 4) {
    Visit vis[4];
    Slice<G12, U8> sl[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      vis[d] = v0 + d < n_visits ? visits[v0 + d] : Visit{0, 0};
      const uint32_t nsl = ((vis[d].len_first & 0xffffu) + 7) >> 3;
      load_slice<G12, U8>(sl[d], recs + vis[d].off4, lane, lane < nsl);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t nsl = ((vis[d].len_first & 0xffffu) + 7) >> 3;
      uint32_t base = vis[d].len_first >> 16;
      float acc = score_slice<G12, U8, HITS>(sl[d], q_idx, q_sc, base, 0.0f);
      for (uint32_t s0 = 16; s0 < nsl; s0 += 16) {   // long documents: further passes
        Slice<G12, U8> more;
        load_slice<G12, U8>(more, recs + vis[d].off4, s0 + lane, s0 + lane < nsl);
        acc = score_slice<G12, U8, HITS>(more, q_idx, q_sc, base, acc);
      }
      acc = __fadd_rn(acc, dpp_ror(acc, 8)); acc = __fadd_rn(acc, dpp_ror(acc, 4));
      acc = __fadd_rn(acc, dpp_ror(acc, 2)); acc = __fadd_rn(acc, dpp_ror(acc, 1));
      if (lane == 0 && v0 + d < n_visits) out[v0 + d] = acc;
    }
  }
}

struct Layout { std::vector<uint32_t> recs; std::vector<Visit> doc; };

template <bool G12, bool U8>
static Layout encode(const std::vector<uint32_t>& off, const std::vector<uint16_t>& comps, const std::vector<float>& vals) {
  constexpr uint32_t CW = G12 ? 3 : 4, VW = U8 ? 2 : 4, SW = CW + VW;
  Layout L;
  const size_t n = off.size() - 1;
  L.doc.resize(n);
  for (size_t d = 0; d < n; ++d) {
    const uint32_t len = off[d + 1] - off[d], nsl = (len + 7) / 8;
    L.doc[d] = Visit{(uint32_t)L.recs.size(), len | ((uint32_t)comps[off[d]] << 16)};
    uint32_t prev = comps[off[d]];
    for (uint32_t s = 0; s < nsl; ++s) {
      uint32_t w[8] = {0};
      uint64_t lo = 0, hi = 0;   // 96 bits of gaps
      for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t e = s * 8 + i;
        const uint32_t c = e < len ? comps[off[d] + e] : (G12 ? prev : kDim);
        const float v = e < len ? vals[off[d] + e] : 0.0f;
        if (G12) {
          const uint32_t g = c - prev;
          if (g >= 4096) { fprintf(stderr, "gap %u needs the raw form\n", g); exit(3); }
          const uint32_t bit = 12 * i;
          if (bit < 64) { lo |= (uint64_t)g << bit; if (bit + 12 > 64) hi |= (uint64_t)g >> (64 - bit); }
          else hi |= (uint64_t)g << (bit - 64);
          prev = c;
        } else {
          w[i / 2] |= c << (16 * (i & 1));
        }
        if (U8) w[CW + i / 4] |= (uint32_t)(v * 64.0f + 0.5f) << (8 * (i & 3));
        else w[CW + i / 2] |= (uint32_t)f32_to_f16(e < len ? v : 1.0f) * (e < len ? 1u : 0u) << (16 * (i & 1));
      }
      if (G12) { w[0] = (uint32_t)lo; w[1] = (uint32_t)(lo >> 32); w[2] = (uint32_t)hi; }
      L.recs.insert(L.recs.end(), w, w + SW);
    }
  }
  return L;
}

template <bool G12, bool U8, bool HITS>
static double run(const char* name, const Layout& L, const std::vector<uint32_t>& order, const uint32_t* d_qc, const float* d_qv, uint32_t nnz,
                  std::vector<float>& scores) {
  std::vector<Visit> vis(order.size());
  uint64_t bytes = 0;
  constexpr uint32_t SW = (G12 ? 3 : 4) + (U8 ? 2 : 4);
  for (size_t i = 0; i < order.size(); ++i) {
    vis[i] = L.doc[order[i]];
    bytes += (uint64_t)(((vis[i].len_first & 0xffffu) + 7) / 8) * SW * 4 + sizeof(Visit);
  }
  uint32_t* d_recs; Visit* d_vis; float* d_out;
  CK(hipMalloc(&d_recs, L.recs.size() * 4 + 256));
  CK(hipMalloc(&d_vis, vis.size() * sizeof(Visit)));
  CK(hipMalloc(&d_out, vis.size() * 4));
  CK(hipMemcpy(d_recs, L.recs.data(), L.recs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_vis, vis.data(), vis.size() * sizeof(Visit), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (uint32_t grid : {512u, 768u, 1024u}) {   // 2, 3, 4 workgroups per CU (the f16 variants hold 65 VGPRs: 3 fit)
    best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((score_kernel<G12, U8, HITS>), dim3(grid), dim3(kNT), 0, 0, d_recs, d_vis, (uint32_t)vis.size(), d_qc, d_qv, nnz, d_out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) best = std::min(best, ms);
    }
    printf("%-8s grid %4u  %5.1f B/slice  records %6.2f GB  %7.3f ms  %7.1f M documents/s  %7.1f GB/s\n", name, grid, SW * 4.0, bytes / 1e9, best,
           vis.size() / best / 1e3, bytes / best / 1e6);
    fflush(stdout);
  }
  scores.resize(vis.size());
  CK(hipMemcpy(scores.data(), d_out, vis.size() * 4, hipMemcpyDeviceToHost));
  CK(hipFree(d_recs)); CK(hipFree(d_vis)); CK(hipFree(d_out));
  return best;
}

int main(int argc, char** argv) {
  const size_t n_docs = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2000000;
  const size_t n_visits = argc > 2 ? strtoull(argv[2], nullptr, 10) : 16000000;
  const double planted = argc > 3 ? atof(argv[3]) : 3.0;
  uint64_t rng = 0x9e3779b97f4a7c15ull;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  const uint32_t nnz = 47;
  std::vector<uint32_t> qc(nnz);
  std::vector<float> qv(nnz);
  for (uint32_t j = 0; j < nnz; ++j) { qc[j] = (uint32_t)(next() % 4000) * 7 % kDim; qv[j] = (float)(1 + next() % 100) / 32.0f; }
  std::sort(qc.begin(), qc.end());
  qc.erase(std::unique(qc.begin(), qc.end()), qc.end());
  std::vector<uint32_t> off(1, 0);
  std::vector<uint16_t> comps;
  std::vector<float> vals;
  std::vector<uint32_t> cur;
  uint64_t hits_total = 0;
  std::vector<uint8_t> in_query(kDim, 0);
  for (uint32_t c : qc) in_query[c] = 1;
  const uint64_t plant_thr = (uint64_t)(planted / (double)qc.size() * 4294967296.0);
  for (size_t d = 0; d < n_docs; ++d) {
    const uint32_t len = 40 + (uint32_t)(next() % 155);              // mean 117
    const double mean_gap = (double)(kDim - 600) / len;              // spread over the vocabulary like a SPLADE document
    uint32_t c = (uint32_t)(next() % 600);
    cur.clear();
    for (uint32_t i = 0; i < len && c < kDim; ++i) {
      cur.push_back(c);
      const double u = (double)((next() >> 11) + 1) / 9007199254740993.0;
      c += 1 + (uint32_t)(-mean_gap * 0.97 * __builtin_log(u));
    }
    for (uint32_t qcomp : qc)
      if ((next() & 0xffffffffull) < plant_thr) cur.push_back(qcomp);   // planted hits
    std::sort(cur.begin(), cur.end());
    cur.erase(std::unique(cur.begin(), cur.end()), cur.end());
    for (uint32_t x : cur) {
      comps.push_back((uint16_t)x);
      vals.push_back((float)(1 + next() % 200) / 64.0f);             // exact in f16 and as a u8 code with step 2^-6
      hits_total += in_query[x];
    }
    off.push_back((uint32_t)comps.size());
  }
  printf("%zu documents, %zu components (mean %.1f), query components per document: %.2f (planted %.1f)\n", n_docs, comps.size(),
         (double)comps.size() / n_docs, (double)hits_total / n_docs, planted);
  std::vector<uint32_t> order(n_visits);
  for (auto& o : order) o = (uint32_t)(next() % n_docs);
  uint32_t* d_qc; float* d_qv;
  CK(hipMalloc(&d_qc, 256 * 4)); CK(hipMalloc(&d_qv, 256 * 4));
  CK(hipMemcpy(d_qc, qc.data(), qc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_qv, qv.data(), qc.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> a16, b16, a8, b8;
  {
    Layout L = encode<false, false>(off, comps, vals);
    run<false, false, false>("f16 all", L, order, d_qc, d_qv, (uint32_t)qc.size(), a16);
    run<false, false, true>("f16 hits", L, order, d_qc, d_qv, (uint32_t)qc.size(), b16);
  }
  {
    Layout L = encode<false, true>(off, comps, vals);
    run<false, true, false>("u8 all", L, order, d_qc, d_qv, (uint32_t)qc.size(), a8);
    run<false, true, true>("u8 hits", L, order, d_qc, d_qv, (uint32_t)qc.size(), b8);
  }
  size_t bad16 = 0, bad8 = 0, nonzero = 0;
  for (size_t i = 0; i < n_visits; ++i) {
    bad16 += memcmp(&a16[i], &b16[i], 4) != 0;
    bad8 += memcmp(&a8[i], &b8[i], 4) != 0;
    nonzero += a16[i] != 0.0f;
  }
  printf("scores differing between the two loops: f16 %zu, u8 %zu of %zu (%zu non-zero)\n", bad16, bad8, n_visits, nonzero);
  return bad16 || bad8 ? 1 : 0;
}
