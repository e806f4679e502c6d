// builder.cpp — CPU index construction (offline, host cores).
//
// What it computes follows the reference's build for the configuration its
// Python API exposes (src/pylib/mod.rs:329-369):
//   GlobalThreshold pruning            src/inverted_index.rs:354-389
//   RandomKmeansInvertedIndexApprox    src/posting_list.rs:227-300, src/utils.rs:106-237
//   EnergyPreserving summaries         src/posting_list.rs:329-368
//   u8 scalar quantisation             src/utils.rs:68-90
//   per-list summary CSR               src/quantized_summary.rs:297-405
// How it computes it is built for 10^6..10^7 documents: a histogram over the
// 65536 binary16 patterns finds the global threshold in one pass (no sort of
// all entries), posting lists are filled by a counting pass, and every list is
// clustered / summarised independently on all host cores with dense per-thread
// scratch (no hash maps). Tie-breaking rules where the reference is
// unspecified are documented in DESIGN.md ("Restated choices") and are the same
// as the oracle's, so both builders produce the identical index.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <numeric>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "build_device.hpp"
#include "host_index.hpp"

namespace sgpu {
namespace {

struct Docs {
  uint64_t n_docs, dim;
  const uint64_t* off;
  const uint32_t* comps;  // widened
  const uint16_t* vals;
};

struct TopC {  // a document's doc_cut heaviest components, heaviest first
  uint32_t c;
  float v;
};

// quantize (src/utils.rs:68-90): round half away from zero, saturating u8 cast, NaN -> 0.
inline void quantize_block(const float* v, size_t n, float* mn_out, float* qt_out, uint8_t* codes) {
  float mn = v[0], mx = v[0];
  for (size_t i = 1; i < n; ++i) {
    if (total_key(v[i]) < total_key(mn)) mn = v[i];
    if (total_key(v[i]) >= total_key(mx)) mx = v[i];
  }
  const float quant = (mx - mn) / 255.0f;
  for (size_t i = 0; i < n; ++i) {
    const float r = std::round((v[i] - mn) / quant);
    codes[i] = std::isnan(r) ? 0 : (r <= 0.0f ? 0 : (r >= 255.0f ? 255 : (uint8_t)r));
  }
  *mn_out = mn;
  *qt_out = quant;
}

// Per-thread scratch, sized once.
struct Scratch {
  // centroid inverted file (component -> [(centroid id, value)]) as a touched-reset CSR
  std::vector<uint32_t> comp_cnt, comp_pos;  // dim
  std::vector<uint32_t> touched_comps;
  std::vector<uint32_t> inv_cid;
  std::vector<float> inv_val;
  std::vector<float> scores;  // n_centroids
  std::vector<uint32_t> touched_c;
  std::vector<uint8_t> avoided, is_touched;
  // summaries
  std::vector<float> maxv;  // dim, NaN-free sentinel via has[]
  std::vector<uint8_t> has;
  std::vector<uint32_t> scomps;
  explicit Scratch(uint64_t dim) : comp_cnt(dim, 0), comp_pos(dim, 0), maxv(dim, 0.0f), has(dim, 0) {}
};

struct ListOut {
  std::vector<uint32_t> post;          // reordered doc ids
  std::vector<uint32_t> block_off;     // nb + 1 (local)
  std::vector<float> mins, quants;     // nb
  std::vector<uint32_t> row_comp;      // rows
  std::vector<uint32_t> row_ptr;       // rows + 1 (local)
  std::vector<uint16_t> bid;
  std::vector<uint8_t> code;
};

// compute_centroid_assignments_approx_dot_product (src/utils.rs:106-144) without
// the O(n_centroids) reset/scan per document: only centroids reached through the
// document's top components can differ from +0.0.
void assign_docs(const std::vector<uint32_t>& docs, const std::vector<uint32_t>& centroid_docs,
                 const std::vector<TopC>& top, uint32_t doc_cut, Scratch& s,
                 std::vector<std::pair<uint32_t, uint32_t>>& out) {
  const size_t nc = centroid_docs.size();
  // largest non-avoided centroid index and, lazily, the largest untouched one
  for (uint32_t doc : docs) {
    s.touched_c.clear();
    const TopC* t = &top[(size_t)doc * doc_cut];
    for (uint32_t i = 0; i < doc_cut && t[i].c != 0xffffffffu; ++i) {
      const uint32_t c = t[i].c;
      const uint32_t n = s.comp_cnt[c];
      if (!n) continue;
      const uint32_t p0 = s.comp_pos[c];
      const float dv = t[i].v;
      for (uint32_t e = 0; e < n; ++e) {
        const uint32_t cid = s.inv_cid[p0 + e];
        if (!s.is_touched[cid]) {
          s.is_touched[cid] = 1;
          s.touched_c.push_back(cid);
        }
        s.scores[cid] = s.scores[cid] + s.inv_val[p0 + e] * dv;
      }
    }
    // argmax by (total_cmp key, index): Rust max_by keeps the LAST maximum.
    int64_t best = -1;
    int32_t best_key = 0;
    for (uint32_t cid : s.touched_c) {
      if (s.avoided[cid]) continue;
      const int32_t k = total_key(s.scores[cid]);
      if (best < 0 || k > best_key || (k == best_key && (int64_t)cid > best)) {
        best = cid;
        best_key = k;
      }
    }
    // best untouched non-avoided centroid: score +0.0, largest index
    int64_t u = (int64_t)nc - 1;
    while (u >= 0 && (s.avoided[(size_t)u] || s.is_touched[(size_t)u])) --u;
    if (u >= 0) {
      const int32_t k0 = total_key(0.0f);
      if (best < 0 || k0 > best_key || (k0 == best_key && u > best)) best = u;
    }
    const uint32_t cdoc = best < 0 ? centroid_docs[0] : centroid_docs[(size_t)best];
    out.emplace_back(cdoc, doc);
    for (uint32_t cid : s.touched_c) {
      s.scores[cid] = 0.0f;
      s.is_touched[cid] = 0;
    }
  }
}

// n_centroids of a list of `len` postings and the sampled centroid documents (src/posting_list.rs:241-246,
// src/utils.rs:163-168; rand's choose_multiple restated as a partial Fisher-Yates over SplitMix64, seed 1142).
void sample_centroids(const std::vector<uint32_t>& postings_by_value, const sgpu_build_config& cfg,
                      std::vector<uint32_t>& centroid_docs) {
  const size_t len = postings_by_value.size();
  const size_t n_centroids = std::max<size_t>(1, (size_t)(cfg.centroid_fraction * (float)len));
  SplitMix64 rng(1142);  // seed of src/utils.rs:163
  std::vector<uint32_t> pool(postings_by_value);
  const size_t nc = std::min(n_centroids, len);
  centroid_docs.resize(nc);
  for (size_t i = 0; i < nc; ++i) {
    const size_t j = i + (size_t)rng.below(len - i);
    std::swap(pool[i], pool[j]);
    centroid_docs[i] = pool[i];
  }
}

// do_random_kmeans_on_docids_ii_approx_dot_product (src/utils.rs:146-237) on the host: (centroid doc, doc)
// pairs of the final assignment, unsorted.
void cluster_list_cpu(const Docs& d, const std::vector<uint32_t>& postings_by_value, const std::vector<uint32_t>& centroid_docs,
                      const sgpu_build_config& cfg, const std::vector<TopC>& top, Scratch& s,
                      std::vector<std::pair<uint32_t, uint32_t>>& fin) {
  const size_t len = postings_by_value.size();
  const size_t nc = centroid_docs.size();
  // centroid inverted file
  s.touched_comps.clear();
  size_t total = 0;
  for (uint32_t cd : centroid_docs)
    for (uint64_t i = d.off[cd]; i < d.off[cd + 1]; ++i) {
      if (s.comp_cnt[d.comps[i]]++ == 0) s.touched_comps.push_back(d.comps[i]);
      ++total;
    }
  s.inv_cid.resize(total);
  s.inv_val.resize(total);
  {
    uint32_t run = 0;
    for (uint32_t c : s.touched_comps) {
      s.comp_pos[c] = run;
      run += s.comp_cnt[c];
      s.comp_cnt[c] = 0;  // reused as fill cursor
    }
    for (size_t cid = 0; cid < nc; ++cid) {
      const uint32_t cd = centroid_docs[cid];
      for (uint64_t i = d.off[cd]; i < d.off[cd + 1]; ++i) {
        const uint32_t c = d.comps[i];
        const uint32_t p = s.comp_pos[c] + s.comp_cnt[c]++;
        s.inv_cid[p] = (uint32_t)cid;
        s.inv_val[p] = f16_to_f32(d.vals[i]);
      }
    }
  }
  s.scores.assign(nc, 0.0f);
  s.avoided.assign(nc, 0);
  s.is_touched.assign(nc, 0);

  std::vector<std::pair<uint32_t, uint32_t>> assign;
  assign.reserve(len);
  assign_docs(postings_by_value, centroid_docs, top, cfg.doc_cut, s, assign);
  std::sort(assign.begin(), assign.end());
  // dissolve clusters with <= min_cluster_size members (src/utils.rs:196-209)
  std::vector<uint32_t> redo;
  fin.clear();
  fin.reserve(len);
  std::vector<std::pair<uint32_t, uint32_t>> cdoc_to_cid(nc);
  for (size_t i = 0; i < nc; ++i) cdoc_to_cid[i] = {centroid_docs[i], (uint32_t)i};
  std::sort(cdoc_to_cid.begin(), cdoc_to_cid.end());
  for (size_t i = 0; i < assign.size();) {
    size_t j = i;
    while (j < assign.size() && assign[j].first == assign[i].first) ++j;
    if (j - i <= cfg.min_cluster_size) {
      for (size_t t = i; t < j; ++t) redo.push_back(assign[t].second);
      auto it = std::lower_bound(cdoc_to_cid.begin(), cdoc_to_cid.end(), std::make_pair(assign[i].first, 0u));
      s.avoided[it->second] = 1;
    } else {
      fin.insert(fin.end(), assign.begin() + (long)i, assign.begin() + (long)j);
    }
    i = j;
  }
  if (!redo.empty()) {
    std::vector<std::pair<uint32_t, uint32_t>> re;
    re.reserve(redo.size());
    assign_docs(redo, centroid_docs, top, cfg.doc_cut, s, re);
    fin.insert(fin.end(), re.begin(), re.end());
  }
  for (uint32_t c : s.touched_comps) s.comp_cnt[c] = 0;
}

// Blocks from the final (centroid doc, doc) pairs (src/posting_list.rs:262-300): postings reordered
// cluster by cluster, clusters in ascending centroid document id.
void form_blocks(std::vector<std::pair<uint32_t, uint32_t>>& fin, ListOut& o) {
  const size_t len = fin.size();
  std::sort(fin.begin(), fin.end());
  o.post.resize(len);
  o.block_off.assign(1, 0);
  for (size_t i = 0; i < fin.size();) {
    size_t j = i;
    while (j < fin.size() && fin[j].first == fin[i].first) ++j;
    for (size_t t = i; t < j; ++t) o.post[t] = fin[t].second;
    o.block_off.push_back((uint32_t)j);
    i = j;
  }
}

struct SummaryTmp {
  std::vector<std::pair<float, uint32_t>> cv;  // (value, comp)
  std::vector<float> kv;
};

// energy_preserving_summary of one block + its u8 quantisation (src/posting_list.rs:329-368,
// src/utils.rs:68-90): kept components ascending in `comps`, their codes in `codes`.
void summarize_block_cpu(const Docs& d, const uint32_t* docs, uint32_t n_docs_b, float summary_energy, Scratch& s,
                         SummaryTmp& t, std::vector<uint32_t>& comps, std::vector<uint8_t>& codes, float* mn, float* qt) {
  s.scomps.clear();
  for (uint32_t x = 0; x < n_docs_b; ++x) {
    const uint32_t doc = docs[x];
    for (uint64_t i = d.off[doc]; i < d.off[doc + 1]; ++i) {
      const uint32_t c = d.comps[i];
      const float v = f16_to_f32(d.vals[i]);
      if (!s.has[c]) {
        s.has[c] = 1;
        s.maxv[c] = v;
        s.scomps.push_back(c);
      } else if (s.maxv[c] < v) {
        s.maxv[c] = v;
      }
    }
  }
  auto& cv = t.cv;
  cv.clear();
  for (uint32_t c : s.scomps) {
    cv.emplace_back(s.maxv[c], c);
    s.has[c] = 0;
  }
  std::sort(cv.begin(), cv.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) {
    const int32_t ka = total_key(a.first), kb = total_key(b.first);
    if (ka != kb) return ka > kb;
    return a.second < b.second;
  });
  float tot = 0.0f;
  for (auto& x : cv) tot = tot + x.first;
  const float until = tot * summary_energy;
  float acc = 0.0f;
  size_t keep = 0;
  for (; keep < cv.size();) {  // take_while_inclusive
    acc = acc + cv[keep].first;
    ++keep;
    if (!(acc < until)) break;
  }
  cv.resize(keep);
  std::sort(cv.begin(), cv.end(),
            [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.second < b.second; });
  t.kv.resize(keep);
  comps.resize(keep);
  codes.resize(keep);
  for (size_t i = 0; i < keep; ++i) {
    t.kv[i] = cv[i].first;
    comps[i] = cv[i].second;
  }
  quantize_block(t.kv.data(), keep, mn, qt, codes.data());
}

struct Ent {
  uint32_t comp;
  uint16_t bid;
  uint8_t code;
};

// per-list summary CSR by component (src/quantized_summary.rs:303-357) from the blocks' kept entries
// (block after block, components ascending inside a block)
void assemble_csr(std::vector<Ent>& ents, ListOut& o) {
  std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.comp < b.comp; });
  o.row_comp.clear();
  o.row_ptr.assign(1, 0);
  o.bid.resize(ents.size());
  o.code.resize(ents.size());
  for (size_t i = 0; i < ents.size(); ++i) {
    if (i == 0 || ents[i].comp != ents[i - 1].comp) {
      if (i) o.row_ptr.push_back((uint32_t)i);
      o.row_comp.push_back(ents[i].comp);
    }
    o.bid[i] = ents[i].bid;
    o.code[i] = ents[i].code;
  }
  if (!ents.empty()) o.row_ptr.push_back((uint32_t)ents.size());
}

}  // namespace

sgpu_status build_host_index(uint32_t comp_width, uint64_t n_docs, uint64_t dim, const uint64_t* offsets,
                             const void* comps_in, const float* vals, const sgpu_build_config& cfg,
                             HostIndex* out) {
  if (comp_width != 2 && comp_width != 4) return fail(SGPU_EINVAL, "comp_width must be 2 or 4");
  if (dim == 0 || (comp_width == 2 && dim > 65535)) return fail(SGPU_EINVAL, "dim out of range for comp_width (u16 components: at most 65535; use comp_width 4)");
  if (n_docs > 0x7fffffffull) return fail(SGPU_EINVAL, "too many documents");
  if (cfg.n_postings == 0 || cfg.doc_cut == 0) return fail(SGPU_EINVAL, "n_postings and doc_cut must be > 0");
  if (!offsets || offsets[0] != 0) return fail(SGPU_EINVAL, "offsets[0] != 0");
  const uint64_t nnz = offsets[n_docs];
#ifdef _OPENMP
  const int nt = cfg.num_threads ? (int)cfg.num_threads : default_host_threads(omp_get_max_threads());
#else
  const int nt = 1;
#endif
  const bool debug = std::getenv("SGPU_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = debug ? now() : 0.0;
  auto lap = [&](const char* what) {
    if (!debug) return;
    const double t = now();
    std::fprintf(stderr, "sgpu build: %-28s %.2f s\n", what, t - t_last);
    t_last = t;
  };
  try {
    HostIndex& h = *out;
    h.comp_width = comp_width;
    h.n_docs = n_docs;
    h.dim = dim;
    h.fwd_offsets.assign(offsets, offsets + n_docs + 1);
    h.fwd_comps.assign((const uint8_t*)comps_in, (const uint8_t*)comps_in + nnz * comp_width);
    h.fwd_vals.resize(nnz);
    std::vector<uint32_t> wide(nnz);
    std::atomic<int> bad{0};
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t doc = 0; doc < (int64_t)n_docs; ++doc) {
      const uint64_t s = offsets[doc], e = offsets[doc + 1];
      if (e < s || e - s > 65535) bad = 1;
      for (uint64_t i = s; i < e && i < nnz; ++i) {
        const uint32_t c = comp_width == 2 ? (uint32_t)((const uint16_t*)comps_in)[i] : ((const uint32_t*)comps_in)[i];
        wide[i] = c;
        if (c >= dim || (i > s && c <= wide[i - 1])) bad = 2;
        if (std::isnan(vals[i])) bad = 3;  // partial_cmp().unwrap() would panic (src/inverted_index.rs:377)
        h.fwd_vals[i] = f32_to_f16_sat(vals[i]);
      }
    }
    if (bad == 1) return fail(SGPU_EINVAL, "offsets not monotone or a document has > 65535 components");
    if (bad == 2) return fail(SGPU_EINVAL, "document components must be strictly ascending and < dim");
    if (bad == 3) return fail(SGPU_EINVAL, "NaN document value");
    Docs d{n_docs, dim, h.fwd_offsets.data(), wide.data(), h.fwd_vals.data()};
    lap("ingest (f16 rounding)");

    // ---- global_threshold_pruning (src/inverted_index.rs:354-389) ----
    // top dim*n_postings entries by value; ties by (doc asc, comp asc) = scan order.
    const uint64_t tot = dim * cfg.n_postings;
    std::vector<uint64_t> hist(65537, 0);  // indexed by f16_desc_key
    {
      std::vector<std::vector<uint64_t>> th((size_t)nt, std::vector<uint64_t>(65537, 0));
#pragma omp parallel num_threads(nt)
      {
#ifdef _OPENMP
        auto& my = th[(size_t)omp_get_thread_num()];
#else
        auto& my = th[0];
#endif
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)nnz; ++i) my[f16_desc_key(h.fwd_vals[(size_t)i])]++;
      }
      for (auto& t : th)
        for (size_t k = 0; k < 65537; ++k) hist[k] += t[k];
    }
    uint32_t thr_key = 65537;  // entries with key < thr_key are all kept
    uint64_t need_eq = 0;      // plus the first need_eq entries (scan order) with key == thr_key
    if (nnz > tot) {
      uint64_t run = 0;
      for (uint32_t k = 0; k < 65537; ++k) {
        if (run + hist[k] >= tot) {
          thr_key = k;
          need_eq = tot - run;
          break;
        }
        run += hist[k];
      }
    }
    // count per list, then fill (value key, doc) pairs. The tie rule is defined on scan order (the first
    // need_eq entries equal to the threshold are kept), so the documents are cut into contiguous chunks:
    // a counting pass gives every chunk the number of threshold-equal entries before it, the selection
    // pass then runs chunk-parallel, and so does the fill (a list's entries of chunk t precede those of t+1).
    std::vector<uint64_t> list_cnt(dim + 1, 0);
    std::vector<uint8_t> sel(nnz, 0);
    const int n_chunks = (int)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)nt, 64ull, (1ull << 26) / std::max<uint64_t>(dim, 1), n_docs}));
    std::vector<uint64_t> chunk_doc((size_t)n_chunks + 1, n_docs);
    chunk_doc[0] = 0;
    for (int t = 1; t < n_chunks; ++t) {   // chunk boundaries at documents, balanced by entries
      const uint64_t want = nnz / (uint64_t)n_chunks * (uint64_t)t;
      chunk_doc[(size_t)t] = (uint64_t)(std::lower_bound(offsets, offsets + n_docs + 1, want) - offsets);
      if (chunk_doc[(size_t)t] > n_docs) chunk_doc[(size_t)t] = n_docs;
      if (chunk_doc[(size_t)t] < chunk_doc[(size_t)t - 1]) chunk_doc[(size_t)t] = chunk_doc[(size_t)t - 1];
    }
    std::vector<uint64_t> eq_before((size_t)n_chunks + 1, 0);
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < n_chunks; ++t) {
      uint64_t eq = 0;
      for (uint64_t i = offsets[chunk_doc[(size_t)t]]; i < offsets[chunk_doc[(size_t)t + 1]]; ++i)
        eq += f16_desc_key(h.fwd_vals[i]) == thr_key;
      eq_before[(size_t)t + 1] = eq;
    }
    for (int t = 0; t < n_chunks; ++t) eq_before[(size_t)t + 1] += eq_before[(size_t)t];
    std::vector<uint32_t> chunk_cnt((size_t)n_chunks * dim, 0);   // entries kept per (chunk, list)
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < n_chunks; ++t) {
      uint64_t eq_seen = eq_before[(size_t)t];
      uint32_t* my = chunk_cnt.data() + (size_t)t * dim;
      for (uint64_t i = offsets[chunk_doc[(size_t)t]]; i < offsets[chunk_doc[(size_t)t + 1]]; ++i) {
        const uint32_t k = f16_desc_key(h.fwd_vals[i]);
        bool take = k < thr_key;
        if (!take && k == thr_key && eq_seen < need_eq) take = true;
        eq_seen += k == thr_key;
        if (take) {
          sel[i] = 1;
          my[wide[i]]++;
        }
      }
    }
    for (uint64_t c = 0; c < dim; ++c) {
      uint64_t n = 0;
      for (int t = 0; t < n_chunks; ++t) n += chunk_cnt[(size_t)t * dim + c];
      list_cnt[c + 1] = list_cnt[c] + n;
    }
    const uint64_t n_sel = list_cnt[dim];
    std::vector<uint64_t> pairs(n_sel);  // (desc key << 32) | doc  -> ascending sort == (value desc, doc asc)
    {
      // the cursor of chunk t in list c: the list's start + what the chunks before t keep there
      std::vector<uint64_t> cur((size_t)n_chunks * dim);
#pragma omp parallel for schedule(static) num_threads(nt)
      for (int64_t c = 0; c < (int64_t)dim; ++c) {
        uint64_t at = list_cnt[(size_t)c];
        for (int t = 0; t < n_chunks; ++t) {
          cur[(size_t)t * dim + (size_t)c] = at;
          at += chunk_cnt[(size_t)t * dim + (size_t)c];
        }
      }
#pragma omp parallel for schedule(static, 1) num_threads(nt)
      for (int t = 0; t < n_chunks; ++t) {
        uint64_t* my = cur.data() + (size_t)t * dim;
        for (uint64_t doc = chunk_doc[(size_t)t]; doc < chunk_doc[(size_t)t + 1]; ++doc)
          for (uint64_t i = offsets[doc]; i < offsets[doc + 1]; ++i)
            if (sel[i]) pairs[my[wide[i]]++] = ((uint64_t)f16_desc_key(h.fwd_vals[i]) << 32) | doc;
      }
    }
    sel.clear();
    sel.shrink_to_fit();
    lap("global threshold pruning");
    const size_t cap = (size_t)((float)cfg.n_postings * cfg.max_fraction);

    // every document's doc_cut heaviest components (k_largest_by, src/utils.rs:125-127),
    // computed once instead of once per posting
    const uint32_t dc = cfg.doc_cut;
    std::vector<TopC> top((size_t)n_docs * dc);
    // allocations inside the parallel regions below cannot reach the enclosing try: the per-thread
    // scratch is created before the regions, and every loop body catches and flags
    std::atomic<int> oom{0};
    std::vector<std::unique_ptr<Scratch>> scratch((size_t)nt);
    for (auto& sp : scratch) sp.reset(new Scratch(dim));
#pragma omp parallel num_threads(nt)
    {
      std::vector<std::pair<int32_t, uint32_t>> tmp;  // (-key, comp)
#pragma omp for schedule(dynamic, 1024)
      for (int64_t doc = 0; doc < (int64_t)n_docs; ++doc) try {
        tmp.clear();
        for (uint64_t i = offsets[doc]; i < offsets[doc + 1]; ++i)
          tmp.emplace_back(total_key(f16_to_f32(h.fwd_vals[i])), wide[i]);
        auto cmp = [](const std::pair<int32_t, uint32_t>& a, const std::pair<int32_t, uint32_t>& b) {
          if (a.first != b.first) return a.first > b.first;
          return a.second < b.second;
        };
        const size_t kk = std::min<size_t>(dc, tmp.size());
        std::partial_sort(tmp.begin(), tmp.begin() + (long)kk, tmp.end(), cmp);
        TopC* t = &top[(size_t)doc * dc];
        for (size_t i = 0; i < dc; ++i) {
          if (i < kk) {
            t[i].c = tmp[i].second;
            int32_t kb = tmp[i].first;  // invert total_key
            kb ^= (int32_t)(((uint32_t)(kb >> 31)) >> 1);
            std::memcpy(&t[i].v, &kb, 4);
          } else {
            t[i].c = 0xffffffffu;
            t[i].v = 0.0f;
          }
        }
      } catch (const std::bad_alloc&) {
        oom = 1;
      }
    }
    if (oom) return fail(SGPU_ENOMEM, "out of host memory building the index");

    lap("top components per doc");
    // ---- per list, phase 1: the postings kept (heaviest first) and the sampled centroids ----
    std::vector<uint64_t> lp_off(dim + 1, 0), lc_off(dim + 1, 0);
    for (uint64_t c = 0; c < dim; ++c) {
      const size_t len = std::min<size_t>(list_cnt[c + 1] - list_cnt[c], cap);
      if ((size_t)(cfg.centroid_fraction * (float)len) > 65535)   // src/posting_list.rs:243-246
        return fail(SGPU_ELIMIT, "a posting list needs more than 65535 centroids; decrease centroid_fraction");
      const size_t nc = len ? std::min(std::max<size_t>(1, (size_t)(cfg.centroid_fraction * (float)len)), len) : 0;
      lp_off[c + 1] = lp_off[c] + len;
      lc_off[c + 1] = lc_off[c] + nc;
    }
    std::vector<uint32_t> post_flat(lp_off[dim]), cent_flat(lc_off[dim]);
    const bool on_device = cfg.use_device != 0;
    std::vector<uint8_t> eligible(dim, 0);
    const uint32_t dev_max_centroids = on_device ? device_assign_max_centroids((int)cfg.use_device - 1) : 0u;   // (by the device's LDS)
    uint64_t inv_cap = 0;
#pragma omp parallel num_threads(nt)
    {
      std::vector<uint32_t> pl, cd;
      uint64_t my_cap = 0;
#pragma omp for schedule(dynamic, 8)
      for (int64_t c = 0; c < (int64_t)dim; ++c) try {
        const uint64_t a = list_cnt[(size_t)c], b = list_cnt[(size_t)c + 1];
        const size_t len = (size_t)(lp_off[(size_t)c + 1] - lp_off[(size_t)c]);
        if (a == b || len == 0) continue;
        std::sort(pairs.begin() + (long)a, pairs.begin() + (long)b);
        pl.resize(len);
        for (size_t i = 0; i < len; ++i) pl[i] = (uint32_t)(pairs[a + i] & 0xffffffffu);
        std::copy(pl.begin(), pl.end(), post_flat.begin() + (long)lp_off[(size_t)c]);
        sample_centroids(pl, cfg, cd);
        std::copy(cd.begin(), cd.end(), cent_flat.begin() + (long)lc_off[(size_t)c]);
        if (on_device && cd.size() <= dev_max_centroids) {
          eligible[(size_t)c] = 1;
          uint64_t entries = 0;
          for (uint32_t x : cd) entries += d.off[x + 1] - d.off[x];
          my_cap = std::max(my_cap, entries);
        }
      } catch (const std::bad_alloc&) {
        oom = 1;
      }
#pragma omp critical
      inv_cap = std::max(inv_cap, my_cap);
    }
    if (oom) return fail(SGPU_ENOMEM, "out of host memory building the index");
    pairs.clear();
    pairs.shrink_to_fit();

    lap("sort postings + centroids");
    // ---- phase 2: cluster assignment on the device (build_assign.hip) for the lists it can take ----
    std::vector<uint32_t> cid_flat;
    if (on_device) {
      cid_flat.resize(post_flat.size());
      sgpu_status dst = device_assign_clusters((int)cfg.use_device - 1, comp_width, n_docs, dim, nnz, h.fwd_offsets.data(),
                                               h.fwd_comps.data(), h.fwd_vals.data(), top.data(), dc, cfg.min_cluster_size,
                                               lp_off.data(), post_flat.data(), lc_off.data(), cent_flat.data(),
                                               eligible.data(), inv_cap, cid_flat.data());
      if (dst != SGPU_OK) return dst;   // no silent fall back to the host path
    }

    lap("device clustering");
    // ---- phase 3a: blocks (and the clustering of the lists that stayed on the host) ----
    std::vector<ListOut> outs(dim);
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      Scratch& s = *scratch[(size_t)omp_get_thread_num()];
#else
      Scratch& s = *scratch[0];
#endif
      std::vector<uint32_t> pl, cd;
      std::vector<std::pair<uint32_t, uint32_t>> fin;
#pragma omp for schedule(dynamic, 8)
      for (int64_t c = 0; c < (int64_t)dim; ++c) try {
        const size_t len = (size_t)(lp_off[(size_t)c + 1] - lp_off[(size_t)c]);
        if (len == 0) continue;
        const uint32_t* pp = post_flat.data() + lp_off[(size_t)c];
        const uint32_t* cp = cent_flat.data() + lc_off[(size_t)c];
        if (eligible[(size_t)c]) {
          const uint32_t* ci = cid_flat.data() + lp_off[(size_t)c];
          fin.resize(len);
          for (size_t i = 0; i < len; ++i) fin[i] = {cp[ci[i]], pp[i]};
        } else {
          pl.assign(pp, pp + len);
          cd.assign(cp, cp + (lc_off[(size_t)c + 1] - lc_off[(size_t)c]));
          cluster_list_cpu(d, pl, cd, cfg, top, s, fin);
        }
        form_blocks(fin, outs[(size_t)c]);
      } catch (const std::bad_alloc&) {
        oom = 1;
      }
    }
    if (oom) return fail(SGPU_ENOMEM, "out of host memory building the index");
    lap("clustering (host) + blocks");

    // ---- phase 3b: per-block summaries on the device (build_summaries.hip) for the blocks it can take ----
    // global block numbering and the reordered postings, list after list
    std::vector<uint64_t> gb_start(dim + 1, 0);
    for (uint64_t c = 0; c < dim; ++c) gb_start[c + 1] = gb_start[c] + (outs[c].block_off.empty() ? 0 : outs[c].block_off.size() - 1);
    const uint64_t n_blocks_all = gb_start[dim];
    std::vector<uint64_t> sb_post(n_blocks_all + 1, 0);   // postings of block b: post_flat2[sb_post[b] .. sb_post[b+1])
    std::vector<uint32_t> sb_entries(n_blocks_all, 0);    // document entries of the block
    std::vector<uint32_t>& post2 = post_flat;             // reused: the postings in block order
    DeviceSummaries ds;
    if (on_device) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
      for (int64_t c = 0; c < (int64_t)dim; ++c) {
        const ListOut& o = outs[(size_t)c];
        const uint64_t nb = o.block_off.empty() ? 0 : o.block_off.size() - 1;
        std::copy(o.post.begin(), o.post.end(), post2.begin() + (long)lp_off[(size_t)c]);
        for (uint64_t b = 0; b < nb; ++b) {
          uint64_t e = 0;
          for (uint32_t t = o.block_off[b]; t < o.block_off[b + 1]; ++t) e += d.off[o.post[t] + 1] - d.off[o.post[t]];
          sb_post[gb_start[(size_t)c] + b + 1] = lp_off[(size_t)c] + o.block_off[b + 1];
          sb_entries[gb_start[(size_t)c] + b] = (uint32_t)std::min<uint64_t>(e, 0xffffffffu);
        }
      }
      for (uint64_t c = 0; c < dim; ++c)   // first block of a list starts where the list's postings start
        if (gb_start[c + 1] > gb_start[c]) sb_post[gb_start[c]] = lp_off[c];
      sgpu_status dst = device_block_summaries((int)cfg.use_device - 1, comp_width, n_docs, nnz, h.fwd_offsets.data(),
                                               h.fwd_comps.data(), h.fwd_vals.data(), cfg.summary_energy, n_blocks_all,
                                               sb_post.data(), sb_entries.data(), post2.data(), &ds);
      if (dst != SGPU_OK) return dst;   // no silent fall back to the host path
      lap("device summaries");
    }

    // ---- phase 3c: the remaining summaries on the host, and every list's summary CSR ----
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      Scratch& s = *scratch[(size_t)omp_get_thread_num()];
#else
      Scratch& s = *scratch[0];
#endif
      SummaryTmp tmp;
      std::vector<Ent> ents;
      std::vector<uint32_t> kc;
      std::vector<uint8_t> kq;
#pragma omp for schedule(dynamic, 8)
      for (int64_t c = 0; c < (int64_t)dim; ++c) try {
        ListOut& o = outs[(size_t)c];
        const size_t nb = o.block_off.empty() ? 0 : o.block_off.size() - 1;
        if (nb == 0) continue;
        o.mins.resize(nb);
        o.quants.resize(nb);
        ents.clear();
        for (size_t b = 0; b < nb; ++b) {
          const uint64_t gb = gb_start[(size_t)c] + b;
          if (on_device && ds.done[gb]) {
            o.mins[b] = ds.mn[gb];
            o.quants[b] = ds.qt[gb];
            const uint64_t p0 = ds.start[gb];
            for (uint32_t i = 0; i < ds.keep[gb]; ++i) ents.push_back({ds.comp[p0 + i], (uint16_t)b, ds.code[p0 + i]});
          } else {
            summarize_block_cpu(d, o.post.data() + o.block_off[b], o.block_off[b + 1] - o.block_off[b], cfg.summary_energy, s,
                                tmp, kc, kq, &o.mins[b], &o.quants[b]);
            for (size_t i = 0; i < kc.size(); ++i) ents.push_back({kc[i], (uint16_t)b, kq[i]});
          }
        }
        assemble_csr(ents, o);
      } catch (const std::bad_alloc&) {
        oom = 1;
      }
    }
    if (oom) return fail(SGPU_ENOMEM, "out of host memory building the index");
    top.clear();
    top.shrink_to_fit();
    lap("summaries (host) + CSR");

    // ---- concatenate ----
    h.list_block_start.assign(dim + 1, 0);
    h.list_row_start.assign(dim + 1, 0);
    std::vector<uint64_t> post_base(dim + 1, 0), ent_base(dim + 1, 0);
    for (uint64_t c = 0; c < dim; ++c) {
      const ListOut& o = outs[c];
      const uint64_t nb = o.block_off.empty() ? 0 : o.block_off.size() - 1;
      h.list_block_start[c + 1] = h.list_block_start[c] + nb;
      h.list_row_start[c + 1] = h.list_row_start[c] + o.row_comp.size();
      post_base[c + 1] = post_base[c] + o.post.size();
      ent_base[c + 1] = ent_base[c] + o.bid.size();
    }
    lap("concat: offsets");
    const uint64_t NB = h.list_block_start[dim], NR = h.list_row_start[dim];
    h.block_post_start.assign(NB + 1, 0);
    h.post_doc.resize(post_base[dim]);
    h.blk_min.resize(NB);
    h.blk_quant.resize(NB);
    h.row_comp.resize(NR * comp_width);
    h.row_ptr.assign(NR + 1, 0);
    h.sum_bid.resize(ent_base[dim]);
    h.sum_code.resize(ent_base[dim]);
    lap("concat: alloc");
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (int64_t c = 0; c < (int64_t)dim; ++c) {
      const ListOut& o = outs[(size_t)c];
      const uint64_t b0 = h.list_block_start[(size_t)c], r0 = h.list_row_start[(size_t)c];
      const uint64_t nb = o.block_off.empty() ? 0 : o.block_off.size() - 1;
      for (uint64_t b = 0; b < nb; ++b) {
        h.block_post_start[b0 + b + 1] = post_base[(size_t)c] + o.block_off[b + 1];
        h.blk_min[b0 + b] = o.mins[b];
        h.blk_quant[b0 + b] = o.quants[b];
      }
      std::copy(o.post.begin(), o.post.end(), h.post_doc.begin() + (long)post_base[(size_t)c]);
      for (size_t r = 0; r < o.row_comp.size(); ++r) {
        if (comp_width == 2) ((uint16_t*)h.row_comp.data())[r0 + r] = (uint16_t)o.row_comp[r];
        else ((uint32_t*)h.row_comp.data())[r0 + r] = o.row_comp[r];
        h.row_ptr[r0 + r + 1] = ent_base[(size_t)c] + o.row_ptr[r + 1];
      }
      std::copy(o.bid.begin(), o.bid.end(), h.sum_bid.begin() + (long)ent_base[(size_t)c]);
      std::copy(o.code.begin(), o.code.end(), h.sum_code.begin() + (long)ent_base[(size_t)c]);
    }
    lap("concatenate");
    // block_post_start[b0] of an empty-list boundary: fill forward so the array is monotone
    // (entries written above are exact; index 0 stays 0 and every list's first block starts
    // at post_base[c], which equals the previous list's last value).
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory building the index");
  }
  return SGPU_OK;
}

}  // namespace sgpu
