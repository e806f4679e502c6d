// seismic_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's (TusKANNy/seismic, Rust) search hot path
// and of the minimal index build needed to obtain something to search. Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product (seismic_amd/csrc) never links, loads or calls it.
//
// PARITY PINNING STATUS
//   Pinned by the reference's own known-answer tests (tests/test_oracle_kat.py):
//     - test_empty_vectors          (reference src/inverted_index.rs:716-772)
//     - test_distances_iter property (reference src/quantized_summary.rs:519-598)
//     - docs/RustUsage.md:138-157 example
//   PARITY UNPINNED for what lives in the un-vendored, un-pinned `vectorium`
//   crate (Cargo.toml:40, no Cargo.lock): the accumulation order of
//   compute_distance (f32 query x f16 doc), the f32->f16 rounding of
//   from_f32_saturating, DotProduct/ScoredRange tie ordering, and the
//   rand::StdRng stream used for centroid sampling (src/utils.rs:163-168).
//   The reference cannot be compiled or imported here (no rustc/cargo, no
//   network), so those choices are restated below and marked [CHOICE].
//
// Every function cites the reference file:line it follows (paths relative to
// the reference repository root).
//
// Build: see oracle/Makefile (g++ -O3 -march=native -ffp-contract=off -fopenmp;
// never -ffast-math: Rust does not contract a*b+c and neither may we).

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/seismic_hip.h"  // only for the sgpu_index_desc layout (plain data)

namespace {

// ---------------------------------------------------------------------------
// binary16 <-> binary32. The reference stores document values as half::f16
// (src/index_traits.rs:57-142) and decodes them with to_f32() (exact).
// ---------------------------------------------------------------------------
inline float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: normalise
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// [CHOICE] vectorium FromF32::from_f32_saturating (call site src/json_utils.rs:64):
// IEEE round-to-nearest-even, finite overflow saturates to +-65504, NaN stays NaN.
inline uint16_t f32_to_f16_sat(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);   // NaN
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);  // >= 65520 (rounds past max) or inf
  if (ax < 0x33000001u) return (uint16_t)sign;               // < 2^-25 (+ tie) -> 0
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  uint32_t shift;
  int32_t he;
  if (e < -14) {  // subnormal half
    shift = (uint32_t)(13 + (-14 - e));
    he = 0;
  } else {
    shift = 13;
    he = e + 15;
  }
  uint32_t q = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  uint32_t out;
  if (he == 0) {
    out = q;  // may carry into exponent 1: still correct encoding
  } else {
    out = ((uint32_t)he << 10) + (q - 0x400u);  // q has implicit bit at 0x400; carry propagates
  }
  return (uint16_t)(sign | out);
}

// f32::total_cmp key (Rust core): monotone map of the bit pattern to i32.
inline int32_t total_key(float f) {
  int32_t b;
  std::memcpy(&b, &f, 4);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return b;
}

// [CHOICE] rand::StdRng (ChaCha12) cannot be reproduced without the crate;
// SplitMix64 stands in. Consequence: centroid sampling differs from any real
// Seismic-built index; the index built here is self-consistent.
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  // multiply-shift bounded draw in [0, n)
  uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};

struct OracleIndex {
  uint32_t comp_width = 2;
  uint64_t n_docs = 0, dim = 0;
  std::vector<uint64_t> fwd_offsets;
  std::vector<uint32_t> fwd_comps;  // kept as u32 internally; exported per comp_width
  std::vector<uint16_t> fwd_vals;
  std::vector<uint8_t> fwd_codes;   // value_type FIXEDU8 (after orc_index_convert_fixedu8)
  uint32_t value_type = SGPU_VAL_F16;
  float val_scale = 0.0f;
  std::vector<uint64_t> list_block_start, block_post_start;
  std::vector<uint32_t> post_doc;
  std::vector<float> blk_min, blk_quant;
  std::vector<uint64_t> list_row_start, row_ptr;
  std::vector<uint32_t> row_comp;  // u32 internally
  std::vector<uint16_t> sum_bid;
  std::vector<uint8_t> sum_code;
  // exported narrow copies
  std::vector<uint16_t> fwd_comps16, row_comp16;
};

// ---------------------------------------------------------------------------
// quantize — reference src/utils.rs:68-90.
//   quant = (max-min)/255; code = ((v-min)/quant).round() as u8
//   Rust `round` = half away from zero (roundf); `as u8` saturates and maps NaN
//   to 0 (all-equal values: quant==0 -> 0/0 = NaN -> code 0 -> dequant == min).
// ---------------------------------------------------------------------------
void quantize_ref(const float* v, size_t n, float* out_min, float* out_quant, uint8_t* codes) {
  assert(n > 0);
  float mn = v[0], mx = v[0];
  for (size_t i = 1; i < n; ++i) {  // minmax_by(total_cmp)
    if (total_key(v[i]) < total_key(mn)) mn = v[i];
    if (total_key(v[i]) >= total_key(mx)) mx = v[i];
  }
  const float quant = (mx - mn) / 255.0f;
  for (size_t i = 0; i < n; ++i) {
    float r = std::round((v[i] - mn) / quant);
    uint8_t c;
    if (std::isnan(r)) c = 0;
    else if (r <= 0.0f) c = 0;
    else if (r >= 255.0f) c = 255;
    else c = (uint8_t)r;
    codes[i] = c;
  }
  *out_min = mn;
  *out_quant = quant;
}

// ---------------------------------------------------------------------------
// Dataset view used by the builder (documents already rounded to f16).
// ---------------------------------------------------------------------------
struct DocsView {
  uint64_t n_docs, dim;
  const uint64_t* off;
  const uint32_t* comps;
  const uint16_t* vals;
};

struct CV {
  uint32_t c;
  float v;
};

// top-`cut` (component,value) of a document by value, descending
// (itertools k_largest_by(total_cmp), src/utils.rs:125-127). [CHOICE] ties:
// ascending component id.
std::vector<CV> top_components(const DocsView& d, uint32_t doc, size_t cut) {
  std::vector<CV> cv;
  for (uint64_t i = d.off[doc]; i < d.off[doc + 1]; ++i) cv.push_back({d.comps[i], f16_to_f32(d.vals[i])});
  std::stable_sort(cv.begin(), cv.end(), [](const CV& a, const CV& b) {
    int32_t ka = total_key(a.v), kb = total_key(b.v);
    if (ka != kb) return ka > kb;
    return a.c < b.c;
  });
  if (cv.size() > cut) cv.resize(cut);
  return cv;
}

// compute_centroid_assignments_approx_dot_product — reference src/utils.rs:106-144.
// Returns (centroid_doc_id, doc_id) pairs.
std::vector<std::pair<uint32_t, uint32_t>> assign_approx(
    const std::vector<uint32_t>& doc_ids,
    const std::unordered_map<uint32_t, std::vector<std::pair<uint32_t, float>>>& inv,
    const DocsView& d, const std::vector<uint32_t>& centroid_docs,
    const std::unordered_set<uint32_t>& to_avoid, size_t doc_cut) {
  std::vector<float> scores(centroid_docs.size());
  std::vector<std::pair<uint32_t, uint32_t>> out;
  out.reserve(doc_ids.size());
  for (uint32_t doc : doc_ids) {
    std::fill(scores.begin(), scores.end(), 0.0f);
    for (const CV& cv : top_components(d, doc, doc_cut)) {
      auto it = inv.find(cv.c);
      if (it == inv.end()) continue;
      for (const auto& cs : it->second) {
        float p = cs.second * cv.v;  // score.to_f32() * value.to_f32()
        scores[cs.first] = scores[cs.first] + p;
      }
    }
    // max_by(total_cmp) over non-avoided centroids: Rust's max_by returns the
    // LAST maximal element; unwrap_or((centroids[0], 0.0)).
    int64_t best = -1;
    for (size_t i = 0; i < centroid_docs.size(); ++i) {
      if (to_avoid.count(centroid_docs[i])) continue;
      if (best < 0 || total_key(scores[i]) >= total_key(scores[best])) best = (int64_t)i;
    }
    uint32_t cdoc = best < 0 ? centroid_docs[0] : centroid_docs[(size_t)best];
    out.emplace_back(cdoc, doc);
  }
  return out;
}

// do_random_kmeans_on_docids_ii_approx_dot_product — reference src/utils.rs:153-237.
std::vector<std::pair<uint32_t, uint32_t>> random_kmeans_approx(const std::vector<uint32_t>& doc_ids,
                                                                size_t n_clusters, const DocsView& d,
                                                                size_t min_cluster_size,
                                                                size_t doc_cut) {
  // centroid sampling: doc_ids.choose_multiple(StdRng::seed_from_u64(1142), n).
  // [CHOICE] partial Fisher-Yates driven by SplitMix64(1142).
  SplitMix64 rng(1142);
  std::vector<uint32_t> pool(doc_ids);
  std::vector<uint32_t> centroid_docs;
  size_t n = std::min(n_clusters, pool.size());
  for (size_t i = 0; i < n; ++i) {
    size_t j = i + (size_t)rng.below(pool.size() - i);
    std::swap(pool[i], pool[j]);
    centroid_docs.push_back(pool[i]);
  }
  // inverted index of the centroids: component -> [(centroid_id, value)]
  std::unordered_map<uint32_t, std::vector<std::pair<uint32_t, float>>> inv;
  for (size_t cid = 0; cid < centroid_docs.size(); ++cid) {
    uint32_t cd = centroid_docs[cid];
    for (uint64_t i = d.off[cd]; i < d.off[cd + 1]; ++i)
      inv[d.comps[i]].emplace_back((uint32_t)cid, f16_to_f32(d.vals[i]));
  }
  auto assign = assign_approx(doc_ids, inv, d, centroid_docs, {}, doc_cut);
  std::sort(assign.begin(), assign.end());
  std::vector<uint32_t> to_reassign;
  std::vector<std::pair<uint32_t, uint32_t>> fin;
  std::unordered_set<uint32_t> removed;
  for (size_t i = 0; i < assign.size();) {
    size_t j = i;
    while (j < assign.size() && assign[j].first == assign[i].first) ++j;
    if (j - i <= min_cluster_size) {  // src/utils.rs:203
      for (size_t t = i; t < j; ++t) to_reassign.push_back(assign[t].second);
      removed.insert(assign[i].first);
    } else {
      for (size_t t = i; t < j; ++t) fin.push_back(assign[t]);
    }
    i = j;
  }
  auto re = assign_approx(to_reassign, inv, d, centroid_docs, removed, doc_cut);
  fin.insert(fin.end(), re.begin(), re.end());
  std::sort(fin.begin(), fin.end());
  return fin;
}

// energy_preserving_summary — reference src/posting_list.rs:329-368.
std::vector<CV> energy_summary(const DocsView& d, const uint32_t* block, size_t n, float fraction) {
  std::map<uint32_t, float> mx;
  for (size_t t = 0; t < n; ++t) {
    uint32_t doc = block[t];
    for (uint64_t i = d.off[doc]; i < d.off[doc + 1]; ++i) {
      float v = f16_to_f32(d.vals[i]);
      auto it = mx.find(d.comps[i]);
      if (it == mx.end()) mx.emplace(d.comps[i], v);
      else if (it->second < v) it->second = v;
    }
  }
  std::vector<CV> cv;
  for (auto& kv : mx) cv.push_back({kv.first, kv.second});
  // sort_unstable_by(|a,b| b.1.total_cmp(&a.1)); [CHOICE] ties: ascending component
  std::stable_sort(cv.begin(), cv.end(), [](const CV& a, const CV& b) {
    int32_t ka = total_key(a.v), kb = total_key(b.v);
    if (ka != kb) return ka > kb;
    return a.c < b.c;
  });
  float total = 0.0f;
  for (auto& x : cv) total = total + x.v;
  const float until = total * fraction;
  float acc = 0.0f;
  std::vector<CV> keep;
  for (auto& x : cv) {  // take_while_inclusive(acc += v; acc < until)
    acc = acc + x.v;
    keep.push_back(x);
    if (!(acc < until)) break;
  }
  std::sort(keep.begin(), keep.end(), [](const CV& a, const CV& b) { return a.c < b.c; });
  return keep;
}

// InvertedIndexBase::build with GlobalThreshold + RandomKmeansInvertedIndexApprox +
// EnergyPreserving — reference src/inverted_index.rs:354-389, 603-686;
// src/posting_list.rs:227-300, 375-450; src/quantized_summary.rs:297-405.
OracleIndex* build_index(uint32_t comp_width, uint64_t n_docs, uint64_t dim, const uint64_t* off,
                         const uint32_t* comps, const float* vals, const sgpu_build_config& cfg) {
  auto* ix = new OracleIndex();
  ix->comp_width = comp_width;
  ix->n_docs = n_docs;
  ix->dim = dim;
  ix->fwd_offsets.assign(off, off + n_docs + 1);
  uint64_t nnz = off[n_docs];
  ix->fwd_comps.assign(comps, comps + nnz);
  ix->fwd_vals.resize(nnz);
  for (uint64_t i = 0; i < nnz; ++i) ix->fwd_vals[i] = f32_to_f16_sat(vals[i]);
  DocsView d{n_docs, dim, ix->fwd_offsets.data(), ix->fwd_comps.data(), ix->fwd_vals.data()};

  // ---- global_threshold_pruning (src/inverted_index.rs:354-389) ----
  struct E {
    uint32_t doc, comp;
    float v;
  };
  std::vector<E> ent;
  ent.reserve(nnz);
  for (uint64_t doc = 0; doc < n_docs; ++doc)
    for (uint64_t i = off[doc]; i < off[doc + 1]; ++i)
      ent.push_back({(uint32_t)doc, comps[i], f16_to_f32(ix->fwd_vals[i])});
  // k_largest_by(tot, partial_cmp on value): descending. [CHOICE] ties: doc asc, comp asc.
  std::stable_sort(ent.begin(), ent.end(), [](const E& a, const E& b) {
    if (a.v != b.v) return a.v > b.v;
    if (a.doc != b.doc) return a.doc < b.doc;
    return a.comp < b.comp;
  });
  const uint64_t tot = dim * cfg.n_postings;
  if (ent.size() > tot) ent.resize(tot);
  const size_t cap = (size_t)((float)cfg.n_postings * cfg.max_fraction);
  std::vector<std::vector<uint32_t>> lists(dim);
  for (const E& e : ent)
    if (lists[e.comp].size() < cap) lists[e.comp].push_back(e.doc);

  // ---- per list: blocking, summaries, quantisation ----
  ix->list_block_start.assign(1, 0);
  ix->block_post_start.assign(1, 0);
  ix->list_row_start.assign(1, 0);
  ix->row_ptr.assign(1, 0);
  for (uint64_t c = 0; c < dim; ++c) {
    std::vector<uint32_t>& pl = lists[c];
    std::vector<size_t> block_offsets;
    if (!pl.empty()) {
      // blocking_with_random_kmeans (src/posting_list.rs:227-300)
      size_t n_centroids = std::max<size_t>(1, (size_t)(cfg.centroid_fraction * (float)pl.size()));
      assert(n_centroids <= 65535);
      auto clusters = random_kmeans_approx(pl, n_centroids, d, cfg.min_cluster_size, cfg.doc_cut);
      std::vector<uint32_t> reordered;
      block_offsets.push_back(0);
      for (size_t i = 0; i < clusters.size();) {
        size_t j = i;
        while (j < clusters.size() && clusters[j].first == clusters[i].first) ++j;
        for (size_t t = i; t < j; ++t) reordered.push_back(clusters[t].second);
        block_offsets.push_back(reordered.size());
        i = j;
      }
      pl = reordered;
    }
    size_t nb = block_offsets.empty() ? 0 : block_offsets.size() - 1;
    // summaries (src/posting_list.rs:409-438) + QuantizedSummary::from
    std::map<uint32_t, std::vector<std::pair<uint8_t, uint16_t>>> inv;  // comp -> (code, summary id)
    for (size_t b = 0; b < nb; ++b) {
      auto s = energy_summary(d, pl.data() + block_offsets[b], block_offsets[b + 1] - block_offsets[b],
                              cfg.summary_energy);
      std::vector<float> sv(s.size());
      for (size_t i = 0; i < s.size(); ++i) sv[i] = s[i].v;
      std::vector<uint8_t> codes(s.size());
      float mn, qt;
      quantize_ref(sv.data(), sv.size(), &mn, &qt, codes.data());
      ix->blk_min.push_back(mn);
      ix->blk_quant.push_back(qt);
      for (size_t i = 0; i < s.size(); ++i) inv[s[i].c].emplace_back(codes[i], (uint16_t)b);
      ix->block_post_start.push_back(ix->block_post_start[ix->list_block_start.back()] +
                                     block_offsets[b + 1]);
    }
    for (uint32_t doc : pl) ix->post_doc.push_back(doc);
    ix->list_block_start.push_back(ix->list_block_start.back() + nb);
    for (auto& kv : inv) {
      ix->row_comp.push_back(kv.first);
      for (auto& cs : kv.second) {
        ix->sum_code.push_back(cs.first);
        ix->sum_bid.push_back(cs.second);
      }
      ix->row_ptr.push_back(ix->sum_bid.size());
    }
    ix->list_row_start.push_back(ix->row_comp.size());
  }
  return ix;
}

void fill_desc(OracleIndex* ix, sgpu_index_desc* out) {
  std::memset(out, 0, sizeof(*out));
  out->comp_width = ix->comp_width;
  out->n_docs = ix->n_docs;
  out->dim = ix->dim;
  out->nnz = ix->fwd_offsets.back();
  out->n_blocks = ix->list_block_start.back();
  out->n_postings = ix->block_post_start.back();
  out->n_rows = ix->list_row_start.back();
  out->n_entries = ix->row_ptr.back();
  out->fwd_offsets = ix->fwd_offsets.data();
  if (ix->comp_width == 2) {
    ix->fwd_comps16.assign(ix->fwd_comps.begin(), ix->fwd_comps.end());
    ix->row_comp16.assign(ix->row_comp.begin(), ix->row_comp.end());
    out->fwd_comps = ix->fwd_comps16.data();
    out->row_comp = ix->row_comp16.data();
  } else {
    out->fwd_comps = ix->fwd_comps.data();
    out->row_comp = ix->row_comp.data();
  }
  out->value_type = ix->value_type;
  out->val_scale = ix->val_scale;
  out->fwd_vals = ix->value_type == SGPU_VAL_FIXEDU8 ? (const void*)ix->fwd_codes.data() : (const void*)ix->fwd_vals.data();
  out->list_block_start = ix->list_block_start.data();
  out->block_post_start = ix->block_post_start.data();
  out->post_doc = ix->post_doc.data();
  out->blk_min = ix->blk_min.data();
  out->blk_quant = ix->blk_quant.data();
  out->list_row_start = ix->list_row_start.data();
  out->row_ptr = ix->row_ptr.data();
  out->sum_bid = ix->sum_bid.data();
  out->sum_code = ix->sum_code.data();
}

// ---------------------------------------------------------------------------
// Search restatement.
// ---------------------------------------------------------------------------
inline uint32_t comp_at(const void* p, uint32_t w, uint64_t i) {
  return w == 2 ? (uint32_t)((const uint16_t*)p)[i] : ((const uint32_t*)p)[i];
}

// KHeap<ScoredRange<DotProduct>> — reference src/utils.rs:12-66.
// A max-heap under the reversed DotProduct order, i.e. its root is the WORST
// retained score. push: insert while len<k, else replace the root iff the new
// item is strictly better (`item < *max`, src/utils.rs:36-39).
// [CHOICE] vectorium's Ord for ScoredRange is not in the tree; restated as the
// total order (score descending, then doc id ascending) for picking the root
// and for into_sorted_vec, while replacement compares SCORES ONLY (an equal
// score never evicts, matching an Ord that looks at the distance alone).
struct KHeap {
  struct Item {
    float score;
    uint32_t doc;
  };
  std::vector<Item> h;  // binary max-heap on "worse"
  size_t k;
  explicit KHeap(size_t k_) : k(k_) { h.reserve(k_); }
  static bool worse(const Item& a, const Item& b) {  // a strictly worse than b
    if (a.score != b.score) return a.score < b.score;
    return a.doc > b.doc;
  }
  static bool less_for_heap(const Item& a, const Item& b) { return worse(b, a); }  // max-heap of worst
  size_t len() const { return h.size(); }
  const Item& peek() const { return h.front(); }
  void push(Item it) {
    if (h.size() < k) {
      h.push_back(it);
      std::push_heap(h.begin(), h.end(), less_for_heap);
    } else if (it.score > h.front().score) {
      std::pop_heap(h.begin(), h.end(), less_for_heap);
      h.back() = it;
      std::push_heap(h.begin(), h.end(), less_for_heap);
    }
  }
  std::vector<Item> into_sorted_vec() {  // best first
    std::vector<Item> v = h;
    std::sort(v.begin(), v.end(), [](const Item& a, const Item& b) { return worse(b, a); });
    return v;
  }
};

struct QueryCtx {            // per-thread scratch (the reference allocates per query)
  std::vector<float> dense;  // dense f32 query over the vocabulary
  std::vector<uint32_t> visited_epoch;
  uint32_t epoch = 0;
  std::vector<float> dots;
  std::vector<uint32_t> order;
};

// QuantizedSummary::distances — reference src/quantized_summary.rs:64-160
// (sparse/merge-join branch 73-118; the dense branch 119-157 performs the same
// arithmetic in the same per-accumulator order).
//   acc[s] += (code as f32 * quants[s] + minimums[s]) * qv     -- no FMA
void summary_distances(const sgpu_index_desc& ix, uint32_t list, const uint32_t* qc, const float* qv,
                       uint32_t nnz, std::vector<float>& acc, uint64_t* touched_entries,
                       uint64_t* matched_rows) {
  const uint64_t b0 = ix.list_block_start[list];
  const uint64_t nb = ix.list_block_start[list + 1] - b0;
  acc.assign(nb, 0.0f);
  const float* mins = ix.blk_min + b0;
  const float* quants = ix.blk_quant + b0;
  uint64_t i = ix.list_row_start[list], iend = ix.list_row_start[list + 1];
  uint32_t j = 0;
  while (i < iend && j < nnz) {
    uint32_t comp = comp_at(ix.row_comp, ix.comp_width, i);
    if (comp == qc[j]) {
      const float q = qv[j];
      for (uint64_t pos = ix.row_ptr[i]; pos < ix.row_ptr[i + 1]; ++pos) {
        const uint32_t s = ix.sum_bid[pos];
        const float deq = (float)ix.sum_code[pos] * quants[s] + mins[s];
        acc[s] = acc[s] + deq * q;
      }
      if (touched_entries) *touched_entries += ix.row_ptr[i + 1] - ix.row_ptr[i];
      if (matched_rows) *matched_rows += 1;
      ++i;
      ++j;
    } else if (comp < qc[j]) {
      ++i;
    } else {
      ++j;
    }
  }
}

// QueryEvaluator::compute_distance (vectorium; call site src/posting_list.rs:210-211).
// [CHOICE] the accumulation order is not in the tree. Two orders are offered:
//   ORDER_SEQ     : one accumulator, left to right in stored (ascending
//                   component) order;
//   ORDER_LANES16 : 16 accumulators; element e of the document goes to
//                   accumulator (e / 8) % 16 (so each accumulator takes runs of
//                   8 consecutive elements, 128 elements per round), added in
//                   increasing e; the 16 partials are then combined by the
//                   butterfly t[j] += t[j ^ s], s = 8,4,2,1. This is the order
//                   the HIP kernel uses (16 lanes x 16-byte loads) and is the
//                   CANONICAL order of this oracle. Multi-accumulator dot
//                   products are what SIMD CPU code produces as well.
// Both skip non-matching components; no FMA.
enum { ORDER_LANES16 = 0, ORDER_SEQ = 1 };

// Document value i of the forward index as f32. F16: exact widening. FIXEDU8 (vectorium's FixedU8Q /
// DotVByteFixedU8Encoder, NOT in the reference tree - parity unpinned; docs/TomlInstructions.md:100
// calls it "8-bit fixed-point quantization (Q0.8)"): [CHOICE] value = code * step with step a power
// of two (2^-8 = Q0.8 when every value is below 1), so the product below is exact.
inline float doc_val(const sgpu_index_desc& ix, uint64_t i) {
  if (ix.value_type == SGPU_VAL_FIXEDU8) return (float)((const uint8_t*)ix.fwd_vals)[i] * ix.val_scale;
  return f16_to_f32(((const uint16_t*)ix.fwd_vals)[i]);
}

inline float score_doc(const sgpu_index_desc& ix, uint32_t doc, const float* dense, int order) {
  const uint64_t s = ix.fwd_offsets[doc], e = ix.fwd_offsets[doc + 1];
  if (order == ORDER_SEQ) {
    float acc = 0.0f;
    for (uint64_t i = s; i < e; ++i) {
      float q = dense[comp_at(ix.fwd_comps, ix.comp_width, i)];
      if (q != 0.0f) acc = acc + q * doc_val(ix, i);
    }
    return acc;
  }
  float t[16];
  for (int j = 0; j < 16; ++j) t[j] = 0.0f;
  for (uint64_t i = s; i < e; ++i) {
    float q = dense[comp_at(ix.fwd_comps, ix.comp_width, i)];
    if (q != 0.0f) {
      int lane = (int)(((i - s) >> 3) & 15);
      t[lane] = t[lane] + q * doc_val(ix, i);
    }
  }
  for (int st = 8; st >= 1; st >>= 1) {
    float u[16];
    for (int j = 0; j < 16; ++j) u[j] = t[j] + t[j ^ st];
    for (int j = 0; j < 16; ++j) t[j] = u[j];
  }
  return t[0];
}

// Knn (reference src/inverted_index.rs:430-435): neighbour ids flattened in document order.
struct KnnView {
  const uint32_t* neighbours = nullptr;
  uint64_t n_total = 0;
  uint32_t dim = 0;
};

struct orc_stats_t {
  uint64_t algo_bytes;      // B_q of SURVEY.md section 8(d)
  uint64_t blocks_total;    // blocks of the walked lists
  uint64_t blocks_scored;   // blocks that passed the skip test
  uint64_t docs_scored;     // compute_distance calls
  uint64_t postings_seen;   // postings of scored blocks (incl. already-visited)
  uint64_t summary_entries; // (code,id) pairs touched by distances()
  uint64_t lists_walked;
};

// InvertedIndexBase::search — reference src/inverted_index.rs:153-234, with
// PostingList::search / sort_and_search / evaluate_posting_block
// (src/posting_list.rs:115-215).
int search_one(const sgpu_index_desc& ix, QueryCtx& ctx, const uint32_t* qc, const float* qv,
               uint32_t nnz, uint32_t k, uint32_t query_cut, float heap_factor, int first_sorted,
               int order, float* out_scores, uint64_t* out_ids, uint32_t* out_n, orc_stats_t* st,
               const KnnView* knn = nullptr, uint32_t in_n_knn = 0) {
  if (k == 0) return 1;                                      // KHeap::new assert (src/utils.rs:23)
  for (uint32_t i = 0; i < nnz; ++i) {
    if (qc[i] >= ix.dim) return 1;                           // Rust bounds panic (src/inverted_index.rs:193)
    if (i && qc[i] <= qc[i - 1]) return 1;                   // sorted assert (src/inverted_index.rs:172-175)
  }
  if (ctx.dense.size() != ix.dim) ctx.dense.assign(ix.dim, 0.0f);
  if (ctx.visited_epoch.size() != ix.n_docs) {
    ctx.visited_epoch.assign(ix.n_docs, 0);
    ctx.epoch = 0;
  }
  if (++ctx.epoch == 0) {
    std::fill(ctx.visited_epoch.begin(), ctx.visited_epoch.end(), 0);
    ctx.epoch = 1;
  }
  for (uint32_t i = 0; i < nnz; ++i) ctx.dense[qc[i]] = qv[i];  // query_evaluator (src/inverted_index.rs:177-178)

  KHeap heap(k);
  // k_largest_by(query_cut, total_cmp on values) — descending (src/inverted_index.rs:187-190).
  // [CHOICE] ties: ascending component id.
  std::vector<uint32_t> sel(nnz);
  std::iota(sel.begin(), sel.end(), 0u);
  std::stable_sort(sel.begin(), sel.end(), [&](uint32_t a, uint32_t b) {
    int32_t ka = total_key(qv[a]), kb = total_key(qv[b]);
    if (ka != kb) return ka > kb;
    return qc[a] < qc[b];
  });
  if (sel.size() > query_cut) sel.resize(query_cut);

  const uint32_t cw = ix.comp_width;
  uint64_t bytes = (uint64_t)nnz * (cw + 4) + 12ull * k;
  bool first = true;
  for (uint32_t si : sel) {
    const uint32_t list = qc[si];
    const uint64_t b0 = ix.list_block_start[list];
    const uint64_t nb = ix.list_block_start[list + 1] - b0;
    uint64_t touched = 0, rows = 0;
    summary_distances(ix, list, qc, qv, nnz, ctx.dots, &touched, &rows);
    bytes += 8 * nb + 3 * touched + 8 * rows;
    if (st) {
      st->blocks_total += nb;
      st->summary_entries += touched;
      st->lists_walked += 1;
    }
    ctx.order.resize(nb);
    std::iota(ctx.order.begin(), ctx.order.end(), 0u);
    if (first && first_sorted) {
      // sorted_unstable_by(|a,b| b.total_cmp(a)) (src/posting_list.rs:162-166). [CHOICE] ties: block asc.
      std::stable_sort(ctx.order.begin(), ctx.order.end(), [&](uint32_t a, uint32_t b) {
        return total_key(ctx.dots[a]) > total_key(ctx.dots[b]);
      });
    }
    first = false;
    for (uint32_t bi = 0; bi < nb; ++bi) {
      const uint32_t b = ctx.order[bi];
      const float dot = ctx.dots[b];
      if (heap.len() == k && dot < heap_factor * heap.peek().score) continue;  // src/posting_list.rs:130
      const uint64_t p0 = ix.block_post_start[b0 + b], p1 = ix.block_post_start[b0 + b + 1];
      bytes += 4 * (p1 - p0 + 1);
      if (st) {
        st->blocks_scored += 1;
        st->postings_seen += p1 - p0;
      }
      // evaluate_posting_block pass 1: prefetch (src/posting_list.rs:198-204)
      for (uint64_t p = p0; p < p1; ++p) {
        uint32_t doc = ix.post_doc[p];
        if (ctx.visited_epoch[doc] == ctx.epoch) continue;
        const uint64_t o = ix.fwd_offsets[doc];
        __builtin_prefetch((const char*)ix.fwd_comps + o * cw);
        __builtin_prefetch((const char*)ix.fwd_vals + o * (ix.value_type == SGPU_VAL_FIXEDU8 ? 1 : 2));
      }
      // pass 2 (src/posting_list.rs:206-214)
      for (uint64_t p = p0; p < p1; ++p) {
        uint32_t doc = ix.post_doc[p];
        if (ctx.visited_epoch[doc] == ctx.epoch) continue;
        ctx.visited_epoch[doc] = ctx.epoch;
        float d = score_doc(ix, doc, ctx.dense.data(), order);
        heap.push({d, doc});
        bytes += 8 + (ix.fwd_offsets[doc + 1] - ix.fwd_offsets[doc]) * (cw + (ix.value_type == SGPU_VAL_FIXEDU8 ? 1 : 2));
        if (st) st->docs_scored += 1;
      }
    }
  }
  // Knn::refine — reference src/inverted_index.rs:215-225, 551-593: snapshot of the heap (best
  // first); for each of its documents the first n_knn neighbours; unvisited ones are scored + pushed.
  if (in_n_knn > 0 && knn && knn->neighbours) {
    const uint32_t n_knn = std::min(knn->dim, in_n_knn);
    auto snap = heap.into_sorted_vec();
    for (const auto& it : snap) {
      const uint64_t base = (uint64_t)it.doc * knn->dim;
      for (uint32_t i = 0; i < n_knn; ++i) {
        if (base + i >= knn->n_total) break;   // the reference reads unchecked here (debug_assert only)
        const uint32_t nb = knn->neighbours[base + i];
        if (nb >= ix.n_docs) continue;
        if (ctx.visited_epoch[nb] == ctx.epoch) continue;
        ctx.visited_epoch[nb] = ctx.epoch;
        heap.push({score_doc(ix, nb, ctx.dense.data(), order), nb});
        if (st) st->docs_scored += 1;
      }
    }
  }
  for (uint32_t i = 0; i < nnz; ++i) ctx.dense[qc[i]] = 0.0f;
  auto res = heap.into_sorted_vec();  // src/inverted_index.rs:227-233
  *out_n = (uint32_t)res.size();
  for (size_t i = 0; i < res.size(); ++i) {
    out_scores[i] = res[i].score;
    out_ids[i] = res[i].doc;
  }
  if (st) st->algo_bytes += bytes;
  return 0;
}

}  // namespace

// ===========================================================================
// C interface for ctypes (tests / smoke / bench cpu_baseline only).
// ===========================================================================
extern "C" {

struct orc_index;  // opaque = OracleIndex

uint16_t orc_f32_to_f16(float f) { return f32_to_f16_sat(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

void orc_quantize(const float* v, uint64_t n, float* out_min, float* out_quant, uint8_t* codes) {
  quantize_ref(v, n, out_min, out_quant, codes);
}

orc_index* orc_index_build(uint32_t comp_width, uint64_t n_docs, uint64_t dim, const uint64_t* offsets,
                           const uint32_t* comps, const float* vals, const sgpu_build_config* cfg) {
  return (orc_index*)build_index(comp_width, n_docs, dim, offsets, comps, vals, *cfg);
}
void orc_index_desc(orc_index* ix, sgpu_index_desc* out) { fill_desc((OracleIndex*)ix, out); }
void orc_index_free(orc_index* ix) { delete (OracleIndex*)ix; }

// InvertedIndexBase::convert_dataset_into::<PackedSparseDataset<DotVByteFixedU8Encoder>> (call sites
// src/pylib/dotvbyte.rs:208-213, src/bin/build_inverted_index.rs:294-306): the standard f16 index is
// built first, then the forward index is re-encoded; posting lists, blocks and summaries stay.
// [CHOICE, parity unpinned] step = the smallest power of two with 255 * step >= the largest value
// (2^-8 when all values are below 1: "Q0.8"); code = min(255, round(v / step)) with Rust's `round`
// (half away from zero); the component stream of DotVByte is a lossless storage codec and carries
// no arithmetic.
orc_index* orc_index_convert_fixedu8(orc_index* src_) {
  const OracleIndex* src = (const OracleIndex*)src_;
  OracleIndex* ix = new OracleIndex(*src);
  const uint64_t nnz = src->fwd_vals.size();
  float vmax = 0.0f;
  for (uint64_t i = 0; i < nnz; ++i) vmax = std::max(vmax, f16_to_f32(src->fwd_vals[i]));
  float step = 1.0f / 256.0f;
  while (255.0f * step < vmax) step *= 2.0f;
  ix->value_type = SGPU_VAL_FIXEDU8;
  ix->val_scale = step;
  ix->fwd_codes.resize(nnz);
  for (uint64_t i = 0; i < nnz; ++i) {
    const float r = std::round(f16_to_f32(src->fwd_vals[i]) / step);
    ix->fwd_codes[i] = r >= 255.0f ? 255 : (r > 0.0f ? (uint8_t)r : 0);
  }
  ix->fwd_vals.clear();
  return (orc_index*)ix;
}

// kNN graph used by orc_search / orc_batch_search when params->n_knn > 0 (test infrastructure:
// one process-wide attachment; pass NULL to detach).
static KnnView g_knn;
void orc_knn_attach(const uint32_t* neighbours, uint64_t n_total, uint32_t dim) {
  g_knn.neighbours = neighbours;
  g_knn.n_total = n_total;
  g_knn.dim = dim;
}

// Knn::new — reference src/inverted_index.rs:448-500. out must hold n_docs * nknn ids; returns the
// number written (documents with fewer than nknn results contribute fewer, as the reference's
// flatten does).
uint64_t orc_knn_build(const sgpu_index_desc* ix, uint32_t nknn, uint32_t* out) {
  QueryCtx ctx;
  const uint32_t k = nknn + 1;
  std::vector<float> sc(k);
  std::vector<uint64_t> ids(k);
  std::vector<uint32_t> qc;
  std::vector<float> qv;
  uint64_t w = 0;
  for (uint64_t d = 0; d < ix->n_docs; ++d) {
    qc.clear();
    qv.clear();
    for (uint64_t i = ix->fwd_offsets[d]; i < ix->fwd_offsets[d + 1]; ++i) {
      qc.push_back(comp_at(ix->fwd_comps, ix->comp_width, i));
      qv.push_back(doc_val(*ix, i));
    }
    uint32_t n = 0;
    search_one(*ix, ctx, qc.data(), qv.data(), (uint32_t)qc.size(), k, 10, 0.7f, 0, ORDER_LANES16, sc.data(),
               ids.data(), &n, nullptr);
    uint32_t taken = 0;
    for (uint32_t i = 0; i < n && taken < nknn; ++i) {
      if (ids[i] == d) continue;
      out[w++] = (uint32_t)ids[i];
      ++taken;
    }
  }
  return w;
}

// hot loop A alone
int orc_summary_distances(const sgpu_index_desc* ix, uint32_t list, const uint32_t* qc, const float* qv,
                          uint32_t nnz, float* out_dots, uint32_t* out_nb) {
  if (list >= ix->dim) return 1;
  std::vector<float> acc;
  summary_distances(*ix, list, qc, qv, nnz, acc, nullptr, nullptr);
  *out_nb = (uint32_t)acc.size();
  std::memcpy(out_dots, acc.data(), acc.size() * 4);
  return 0;
}

// single query; stats may be NULL. order: 0 = LANES16 (canonical), 1 = SEQ.
int orc_search(const sgpu_index_desc* ix, const uint32_t* qc, const float* qv, uint32_t nnz,
               const sgpu_search_params* p, int order, float* out_scores, uint64_t* out_ids,
               uint32_t* out_n, orc_stats_t* st) {
  static thread_local QueryCtx ctx;
  return search_one(*ix, ctx, qc, qv, nnz, p->k, p->query_cut, p->heap_factor, p->first_sorted, order,
                    out_scores, out_ids, out_n, st, &g_knn, p->n_knn);
}

// batch; num_threads 1 = the sequential loop of perf_inverted_index
// (src/bin/perf_inverted_index.rs:184-211), 0 = all cores (rayon global pool,
// src/pylib/mod.rs:629-652). Returns elapsed seconds of the search loop in *secs.
int orc_batch_search(const sgpu_index_desc* ix, const uint64_t* q_off, const uint32_t* qc,
                     const float* qv, uint32_t nq, const sgpu_search_params* p, int order,
                     uint32_t num_threads, float* out_scores, uint64_t* out_ids, uint32_t* out_n,
                     orc_stats_t* st_total, double* secs, uint32_t* threads_used) {
  if (p->k == 0) return 1;
  int nt = 1;
#ifdef _OPENMP
  nt = num_threads ? (int)num_threads : omp_get_max_threads();
#endif
  if (threads_used) *threads_used = (uint32_t)nt;
  int err = 0;
  std::vector<orc_stats_t> sts((size_t)nt);
  std::memset(sts.data(), 0, sts.size() * sizeof(orc_stats_t));
  // warm the per-thread contexts outside the timed region (the reference's
  // allocations are per query; ours are hoisted — a conservative, faster baseline)
  std::vector<QueryCtx> ctxs((size_t)nt);
  for (auto& c : ctxs) {
    c.dense.assign(ix->dim, 0.0f);
    c.visited_epoch.assign(ix->n_docs, 0);
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
  for (int64_t q = 0; q < (int64_t)nq; ++q) {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    int e = search_one(*ix, ctxs[(size_t)tid], qc + q_off[q], qv + q_off[q],
                       (uint32_t)(q_off[q + 1] - q_off[q]), p->k, p->query_cut, p->heap_factor,
                       p->first_sorted, order, out_scores + (size_t)q * p->k,
                       out_ids + (size_t)q * p->k, out_n + q, &sts[(size_t)tid], &g_knn, p->n_knn);
    if (e) {
#pragma omp atomic write
      err = e;
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (secs) *secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (st_total) {
    std::memset(st_total, 0, sizeof(*st_total));
    for (auto& s : sts) {
      st_total->algo_bytes += s.algo_bytes;
      st_total->blocks_total += s.blocks_total;
      st_total->blocks_scored += s.blocks_scored;
      st_total->docs_scored += s.docs_scored;
      st_total->postings_seen += s.postings_seen;
      st_total->summary_entries += s.summary_entries;
      st_total->lists_walked += s.lists_walked;
    }
  }
  return err;
}

// Exact top-k by brute force over every document — semantics of
// SeismicDataset.search / vectorium FlatIndex (src/inverted_index_wrapper.rs:721-742).
// Scores use `order`; results best-first, ties by ascending doc id; docs with
// no overlap score 0 and ARE eligible (a flat scan returns them), so callers
// comparing against the inverted index should use queries with >= k matches.
int orc_exact_search(const sgpu_index_desc* ix, const uint32_t* qc, const float* qv, uint32_t nnz,
                     uint32_t k, int order, float* out_scores, uint64_t* out_ids, uint32_t* out_n) {
  if (k == 0) return 1;
  std::vector<float> dense(ix->dim, 0.0f);
  for (uint32_t i = 0; i < nnz; ++i) dense[qc[i]] = qv[i];
  KHeap heap(k);
  for (uint64_t d = 0; d < ix->n_docs; ++d) heap.push({score_doc(*ix, (uint32_t)d, dense.data(), order), (uint32_t)d});
  auto res = heap.into_sorted_vec();
  *out_n = (uint32_t)res.size();
  for (size_t i = 0; i < res.size(); ++i) {
    out_scores[i] = res[i].score;
    out_ids[i] = res[i].doc;
  }
  return 0;
}

// score one document against a query (for kernel-level parity tests)
float orc_score_doc(const sgpu_index_desc* ix, uint32_t doc, const uint32_t* qc, const float* qv,
                    uint32_t nnz, int order) {
  std::vector<float> dense(ix->dim, 0.0f);
  for (uint32_t i = 0; i < nnz; ++i) dense[qc[i]] = qv[i];
  return score_doc(*ix, doc, dense.data(), order);
}

uint32_t orc_stats_size(void) { return (uint32_t)sizeof(orc_stats_t); }

}  // extern "C"
