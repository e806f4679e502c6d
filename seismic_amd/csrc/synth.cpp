// synth.cpp — deterministic SPLADE-shaped synthetic sparse vectors.
//
// There is no network for MS MARCO / SPLADE-v3 embeddings, so bench.py and the
// full-size parity tests draw documents and queries of that SHAPE:
//   vocabulary V, token popularity Zipf(s=1) over a seeded permutation of ids;
//   document nnz ~ round(lognormal(ln 110, 0.35)) clipped to [16, 400] (mean ~120);
//   values v = min(3.5, 0.02 + Exp(mean 0.45)) (toy_dataset range 0.001..2.6);
//   topical structure: every document belongs to one of T topics and draws 70%
//   of its tokens from that topic's 256-token set (so k-means blocks and their
//   summaries are as discriminative as on real learned-sparse data), 30% from
//   the global Zipf law;
//   queries: nnz ~ clipped normal(43, 10) in [8, 96]; 60% of the tokens are the
//   heaviest components of a random source document, 40% Zipf; values from the
//   same law in f32, all distinct within a query (no tie ambiguity).
// PRNG: SplitMix64 only; no std:: distributions (stable across libstdc++ builds).
//
// Second collection (sgpu_synth_spec::collection == 1, "clustered"; bench.py --collection clustered): same sizes, same
// PRNG discipline, but documents are drawn around LATENT INTENTS the way passages about one subject are: the
// collection is cut into groups of ~96 documents; a group belongs to a topic and owns a core of kCoreTokens
// of the topic's tokens with a weight each; a document takes most of its group's core (80 % of the tokens, weight x
// lognormal noise), a fifth of its tokens from the rest of its topic and the remainder from the Zipf law. A query
// repeats the heaviest components of its source document WITH that document's weights (x noise), so its exact
// neighbours are the source's group - well separated from the rest of the topic, as the answers to an MS MARCO
// query are. The SURVEY 8(d) collection above makes every document of a topic about equally close to a query
// (recall 0.83 at the reference's recall_95 parameters, 3.3 MB touched per query); this one tells whether the
// kernel's figures survive on work per query of the published size.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace sgpu {
namespace {

struct Zipf {
  std::vector<double> cdf;
  std::vector<uint32_t> perm;
  Zipf(uint64_t dim, uint64_t seed) : cdf(dim), perm(dim) {
    double h = 0;
    for (uint64_t r = 0; r < dim; ++r) {
      h += 1.0 / (double)(r + 1);
      cdf[r] = h;
    }
    for (auto& x : cdf) x /= h;
    for (uint64_t i = 0; i < dim; ++i) perm[i] = (uint32_t)i;
    SplitMix64 rng(seed ^ 0x5eed5eedull);
    for (uint64_t i = dim - 1; i > 0; --i) std::swap(perm[i], perm[rng.below(i + 1)]);
  }
  uint32_t draw(SplitMix64& rng) const {
    const double u = rng.unit();
    size_t r = (size_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
    if (r >= cdf.size()) r = cdf.size() - 1;
    return perm[r];
  }
};

inline double normal(SplitMix64& rng) {  // Box-Muller, one value
  double u1 = rng.unit(), u2 = rng.unit();
  if (u1 < 1e-300) u1 = 1e-300;
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}
inline float value_law(SplitMix64& rng, double scale) {
  double u = rng.unit();
  if (u < 1e-300) u = 1e-300;
  double v = 0.02 + (-std::log(u)) * 0.45 * scale;
  return (float)std::min(3.5, v);
}

constexpr uint32_t kTopicTokens = 256;
constexpr uint32_t kCoreTokens = 48;    // clustered collection: tokens of a group's core
// ... and its law's parameters: documents per latent intent (mean), share of the core a document takes, sigma of the
// lognormal noise on a document's core weights and on the weights a query repeats from its source document.
// (SGPU_SYNTH_CLUSTER="group,take,doc_sigma,query_sigma" overrides them while SGPU_TEST_HOOKS=1 is set: tools/clustered_tune.py)
struct ClusterLaw {
  uint32_t group_docs = 96;
  double take = 0.8, doc_sigma = 0.25, query_sigma = 0.3;
};
static ClusterLaw cluster_law() {
  ClusterLaw l;
  const char* t = std::getenv("SGPU_TEST_HOOKS");
  const char* v = (t && *t && *t != '0') ? std::getenv("SGPU_SYNTH_CLUSTER") : nullptr;
  if (v) {
    unsigned g = 0;
    double a = 0, b = 0, c = 0;
    if (std::sscanf(v, "%u,%lf,%lf,%lf", &g, &a, &b, &c) == 4 && g >= 1) {
      l.group_docs = g;
      l.take = a;
      l.doc_sigma = b;
      l.query_sigma = c;
    }
  }
  return l;
}

}  // namespace

// Topic tables depend only on (dim, topic count), never on the stream seed, so
// documents (seed 42) and queries (seed 43) share the same topics.
static void make_topics(uint64_t dim, uint32_t n_topics, const Zipf& z, std::vector<uint32_t>& tok,
                        std::vector<float>& scale) {
  tok.resize((size_t)n_topics * kTopicTokens);
  scale.resize(tok.size());
  SplitMix64 rng(0x70b1c5ull ^ dim);
  std::vector<uint8_t> used(dim, 0);
  for (uint32_t t = 0; t < n_topics; ++t) {
    uint32_t* tt = &tok[(size_t)t * kTopicTokens];
    const uint32_t want = (uint32_t)std::min<uint64_t>(kTopicTokens, dim);
    uint32_t n = 0;
    while (n < want) {
      // half Zipf (shared, popular), half uniform (topic-specific, rare)
      uint32_t c = (n & 1) ? (uint32_t)rng.below(dim) : z.draw(rng);
      if (used[c]) continue;
      used[c] = 1;
      tt[n] = c;
      scale[(size_t)t * kTopicTokens + n] = (float)(0.5 + 1.5 * rng.unit());
      ++n;
    }
    for (uint32_t i = 0; i < n; ++i) used[tt[i]] = 0;
    for (uint32_t i = n; i < kTopicTokens; ++i) tt[i] = tt[i % std::max(1u, n)];
  }
}

static inline uint64_t mix_seed(uint64_t seed, uint64_t i) {
  SplitMix64 m(seed * 0x9e3779b97f4a7c15ull + i);
  m.next();
  return m.next();
}

// number of components of vector i: the first draws of its private stream
static inline uint32_t draw_nnz(SplitMix64& rng, uint32_t kind, uint64_t dim) {
  if (kind == 0) {
    const double ln = std::log(110.0) + 0.35 * normal(rng);
    uint32_t n = (uint32_t)std::llround(std::exp(ln));
    n = std::max(16u, std::min(400u, n));
    return (uint32_t)std::min<uint64_t>(n, dim / 2);
  }
  const double g = 43.0 + 10.0 * normal(rng);
  return (uint32_t)std::llround(std::max(8.0, std::min(96.0, g)));
}

// Clustered collection: group g's topic and core - kCoreTokens distinct slots of the topic's token table, each with a
// weight. A function of (dim, g) only, so documents and queries agree on it.
struct GroupCore {
  uint32_t topic;
  uint32_t slot[kCoreTokens];
  float weight[kCoreTokens];
};
static GroupCore group_core(uint64_t dim, uint64_t g, uint32_t n_topics, const std::vector<float>& tscale) {
  GroupCore gc;
  SplitMix64 rng(mix_seed(0xc1057e2ull ^ dim, g));
  gc.topic = (uint32_t)rng.below(n_topics);
  uint8_t used[kTopicTokens] = {0};
  for (uint32_t i = 0; i < kCoreTokens;) {
    const uint32_t j = (uint32_t)rng.below(kTopicTokens);
    if (used[j]) continue;
    used[j] = 1;
    gc.slot[i] = j;
    gc.weight[i] = value_law(rng, tscale[(size_t)gc.topic * kTopicTokens + j]);
    ++i;
  }
  return gc;
}
static inline uint64_t group_of(uint64_t doc, uint64_t n_docs, uint32_t group_docs) {   // documents are dealt to groups by a hash: no id locality
  const uint64_t n_groups = std::max<uint64_t>(1, n_docs / group_docs);
  return mix_seed(0x6209ull, doc) % n_groups;
}

extern "C" sgpu_status sgpu_synth_generate(const sgpu_synth_spec* spec, const uint64_t* docs_offsets,
                                           const uint32_t* docs_comps, const float* docs_vals,
                                           uint64_t n_docs, uint64_t* out_offsets, uint32_t* out_comps,
                                           float* out_vals, uint64_t* out_nnz) {
  if (!spec || !out_nnz) return fail(SGPU_EINVAL, "null spec / out_nnz");
  if (spec->dim < 1024) return fail(SGPU_EINVAL, "dim must be >= 1024");
  if (spec->collection > 1) return fail(SGPU_EINVAL, "unknown synthetic collection %u (0 = SURVEY 8d law, 1 = clustered)", spec->collection);
  if (spec->kind == 1 && (n_docs == 0 || !docs_offsets || !docs_comps || !docs_vals))
    return fail(SGPU_EINVAL, "queries need the source documents");
  const uint64_t dim = spec->dim;
  const bool write = out_comps != nullptr;
  if (write && (!out_offsets || !out_vals)) return fail(SGPU_EINVAL, "null output arrays");
  // pass 1: sizes (every vector has its own stream, so this is cheap and the fill is parallel)
  std::vector<uint64_t> off(spec->n_vecs + 1, 0);
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
  for (int64_t i = 0; i < (int64_t)spec->n_vecs; ++i) {
    SplitMix64 rng(mix_seed(spec->seed, (uint64_t)i));
    off[(size_t)i + 1] = draw_nnz(rng, spec->kind, dim);
  }
  for (uint64_t i = 0; i < spec->n_vecs; ++i) off[i + 1] += off[i];
  *out_nnz = off[spec->n_vecs];
  if (!write) return SGPU_OK;
  std::copy(off.begin(), off.end(), out_offsets);

  const ClusterLaw law = cluster_law();
  Zipf z(dim, 0xd1ce);
  const uint32_t n_topics = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(4096, dim / 64));
  std::vector<uint32_t> ttok;
  std::vector<float> tscale;
  make_topics(dim, n_topics, z, ttok, tscale);

#pragma omp parallel num_threads(sgpu::host_threads())
  {
    std::vector<uint8_t> used(dim, 0);
    std::vector<std::pair<uint32_t, float>> cur;
    std::vector<std::pair<float, uint32_t>> sv;
    std::vector<float> seen;
#pragma omp for schedule(dynamic, 256)
    for (int64_t ii = 0; ii < (int64_t)spec->n_vecs; ++ii) {
      const uint64_t i = (uint64_t)ii;
      SplitMix64 rng(mix_seed(spec->seed, i));
      const uint32_t n = draw_nnz(rng, spec->kind, dim);
      cur.clear();
      if (spec->kind == 0 && spec->collection == 1) {
        // clustered: most of the group's core (weights x lognormal noise), a fifth from the rest of the topic, then Zipf
        const GroupCore gc = group_core(dim, group_of(i, spec->n_vecs, law.group_docs), n_topics, tscale);
        for (uint32_t j = 0; j < kCoreTokens && cur.size() < n; ++j) {
          const bool take = rng.unit() < law.take;
          const double noise = std::exp(law.doc_sigma * normal(rng));
          const uint32_t c = ttok[(size_t)gc.topic * kTopicTokens + gc.slot[j]];
          if (!take || used[c]) continue;
          used[c] = 1;
          cur.emplace_back(c, (float)std::min(3.5, std::max(0.02, (double)gc.weight[j] * noise)));
        }
        const uint32_t n_topic = std::min<uint32_t>((uint32_t)cur.size() + (uint32_t)(0.2 * n), n);
        uint32_t guard = 0;
        while (cur.size() < n_topic && guard++ < 8 * kTopicTokens) {
          const uint32_t j = (uint32_t)rng.below(kTopicTokens);
          const uint32_t c = ttok[(size_t)gc.topic * kTopicTokens + j];
          if (used[c]) continue;
          used[c] = 1;
          cur.emplace_back(c, value_law(rng, 0.6 * tscale[(size_t)gc.topic * kTopicTokens + j]));
        }
        while (cur.size() < n) {
          const uint32_t c = z.draw(rng);
          if (used[c]) continue;
          used[c] = 1;
          cur.emplace_back(c, value_law(rng, 0.6));
        }
      } else if (spec->kind == 0) {
        const uint32_t topic = (uint32_t)rng.below(n_topics);
        const uint32_t n_topic = std::min<uint32_t>((uint32_t)(0.7 * n), kTopicTokens / 2);
        uint32_t guard = 0;
        while (cur.size() < n_topic && guard++ < 8 * kTopicTokens) {
          const uint32_t j = (uint32_t)rng.below(kTopicTokens);
          const uint32_t c = ttok[(size_t)topic * kTopicTokens + j];
          if (used[c]) continue;
          used[c] = 1;
          cur.emplace_back(c, value_law(rng, tscale[(size_t)topic * kTopicTokens + j]));
        }
        while (cur.size() < n) {
          const uint32_t c = z.draw(rng);
          if (used[c]) continue;
          used[c] = 1;
          cur.emplace_back(c, value_law(rng, 1.0));
        }
      } else {
        const uint64_t src = rng.below(n_docs);
        const uint64_t s = docs_offsets[src], e = docs_offsets[src + 1];
        sv.clear();
        for (uint64_t p = s; p < e; ++p) sv.emplace_back(-docs_vals[p], docs_comps[p]);
        std::sort(sv.begin(), sv.end());
        const uint32_t n_src = (uint32_t)std::min<size_t>((size_t)std::llround(0.6 * n), sv.size());
        for (uint32_t j = 0; j < n_src; ++j) {
          used[sv[j].second] = 1;
          // (clustered: the source document's own weight x noise - the query is ABOUT that document)
          const float v = spec->collection == 1 ? (float)std::min(3.5, std::max(0.02, (double)(-sv[j].first) * std::exp(law.query_sigma * normal(rng))))
                                                : value_law(rng, 1.0);
          cur.emplace_back(sv[j].second, v);
        }
        while (cur.size() < n) {
          const uint32_t c = z.draw(rng);
          if (used[c]) continue;
          used[c] = 1;
          cur.emplace_back(c, value_law(rng, spec->collection == 1 ? 0.4 : 1.0));
        }
        seen.clear();  // make the values of one query pairwise distinct
        for (auto& cv : cur) {
          while (std::find(seen.begin(), seen.end(), cv.second) != seen.end())
            cv.second = std::nextafterf(cv.second, 0.0f);
          seen.push_back(cv.second);
        }
      }
      std::sort(cur.begin(), cur.end());
      for (auto& cv : cur) used[cv.first] = 0;
      const uint64_t base = off[i];
      for (size_t j = 0; j < cur.size(); ++j) {
        out_comps[base + j] = cur[j].first;
        out_vals[base + j] = cur[j].second;
      }
    }
  }
  return SGPU_OK;
}

}  // namespace sgpu
