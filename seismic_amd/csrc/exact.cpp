// exact.cpp — exact top-k over ALL documents (ground truth for recall@k).
//
// Semantics of SeismicDataset.search / vectorium FlatIndex (reference
// src/inverted_index_wrapper.rs:721-742): score every document by the inner
// product, return the k best. Implemented as term-at-a-time accumulation over a
// full (unpruned) inverted file of the forward index, queries in parallel on
// the host cores. Each document's partial sums are added in ascending component
// order (f32, no FMA), i.e. the sequential left-to-right dot product. Host code;
// not on the hot path (used by bench.py and tests to measure recall).
#include <algorithm>
#include <atomic>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "host_index.hpp"

namespace sgpu {

sgpu_status exact_search_host(const HostIndex& ix, const uint64_t* q_off, const uint32_t* comps,
                              const float* vals, uint32_t nq, uint32_t k, uint32_t num_threads,
                              float* out_scores, uint64_t* out_ids, uint32_t* out_n) {
  if (k == 0) return fail(SGPU_EINVAL, "k == 0");
  const uint64_t nnz = ix.nnz();
  {
    uint32_t max_nnz = 0;
    const sgpu_status vst = validate_queries(ix.dim, q_off, comps, vals, nq, &max_nnz);
    if (vst != SGPU_OK) return vst;
  }
#ifdef _OPENMP
  const int nt = num_threads ? (int)num_threads : default_host_threads(omp_get_max_threads());
#else
  const int nt = 1;
#endif
  try {
    // full inverted file (component -> (doc, value)), docs ascending within a list: a counting sort, the documents cut
    // into one contiguous range per thread (a thread's entries of a component follow those of the threads before it)
    std::vector<uint64_t> ptr(ix.dim + 1, 0);
    std::vector<uint32_t> idoc(nnz);
    std::vector<float> ival(nnz);   // document values widened once (exact for f16 and fixed-u8)
    {
      // (the per-thread counters are threads x vocabulary 64-bit words: the team is cut down so that they stay under 256 MB -
      // a 1M-id vocabulary on 128 threads asked for 1 GB of scratch where the serial pass needs 8 MB)
      const int bt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(nt < 1 ? 1 : nt), ((uint64_t)256 << 20) / (8 * std::max<uint64_t>(ix.dim, 1))));
      std::vector<uint64_t> cur((size_t)bt * ix.dim, 0);   // [thread][component]: count, then first slot
      auto range = [&](int t, uint64_t* d0, uint64_t* d1) {
        *d0 = ix.n_docs * (uint64_t)t / (uint64_t)bt;
        *d1 = ix.n_docs * (uint64_t)(t + 1) / (uint64_t)bt;
      };
#pragma omp parallel for num_threads(bt) schedule(static, 1)
      for (int t = 0; t < bt; ++t) {
        uint64_t d0, d1;
        range(t, &d0, &d1);
        uint64_t* c = cur.data() + (size_t)t * ix.dim;
        for (uint64_t i = ix.fwd_offsets[d0]; i < ix.fwd_offsets[d1]; ++i) c[ix.comp(i)]++;
      }
      uint64_t run = 0;
      for (uint64_t c = 0; c < ix.dim; ++c) {
        ptr[c] = run;
        for (int t = 0; t < bt; ++t) {
          const uint64_t n = cur[(size_t)t * ix.dim + c];
          cur[(size_t)t * ix.dim + c] = run;
          run += n;
        }
      }
      ptr[ix.dim] = run;
#pragma omp parallel for num_threads(bt) schedule(static, 1)
      for (int t = 0; t < bt; ++t) {
        uint64_t d0, d1;
        range(t, &d0, &d1);
        uint64_t* c = cur.data() + (size_t)t * ix.dim;
        for (uint64_t d = d0; d < d1; ++d)
          for (uint64_t i = ix.fwd_offsets[d]; i < ix.fwd_offsets[d + 1]; ++i) {
            const uint64_t p = c[ix.comp(i)]++;
            idoc[p] = (uint32_t)d;
            ival[p] = ix.val(i);
          }
      }
    }
    // per-thread scratch is sized here, where an allocation failure reaches the enclosing try; inside
    // the parallel region a failure (growth of the small vectors) is caught per query and flagged
    struct Thread {
      std::vector<float> acc;
      std::vector<uint8_t> seen;
      std::vector<uint32_t> touched;
      std::vector<std::pair<float, uint32_t>> cand;
    };
    std::vector<Thread> threads((size_t)nt);
    std::atomic<int> oom{0};
    // (sized by the threads themselves - 44 MB each at 8.8M documents - in a region without a
    // worksharing construct, so a failure is caught where it happens)
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      Thread& me = threads[(size_t)omp_get_thread_num()];
#else
      Thread& me = threads[0];
#endif
      try {
        me.acc.assign(ix.n_docs, 0.0f);
        me.seen.assign(ix.n_docs, 0);
      } catch (const std::bad_alloc&) {
        oom = 1;
      }
    }
    if (oom) return fail(SGPU_ENOMEM, "out of host memory in exact search");
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      Thread& me = threads[(size_t)omp_get_thread_num()];
#else
      Thread& me = threads[0];
#endif
      std::vector<float>& acc = me.acc;
      std::vector<uint8_t>& seen = me.seen;
      std::vector<uint32_t>& touched = me.touched;
      std::vector<std::pair<float, uint32_t>>& cand = me.cand;
      const bool ready = acc.size() == ix.n_docs && seen.size() == ix.n_docs;   // (a smaller team than asked for leaves slots unused)
#pragma omp for schedule(dynamic, 1)
      for (int64_t q = 0; q < (int64_t)nq; ++q) try {
        if (!ready) throw std::bad_alloc();
        touched.clear();
        for (uint64_t j = q_off[q]; j < q_off[q + 1]; ++j) {  // ascending component
          const float qv = vals[j];
          for (uint64_t p = ptr[comps[j]]; p < ptr[comps[j] + 1]; ++p) {
            const uint32_t d = idoc[p];
            if (!seen[d]) {
              seen[d] = 1;
              touched.push_back(d);
            }
            acc[d] = acc[d] + qv * ival[p];
          }
        }
        cand.clear();
        for (uint32_t d : touched) cand.emplace_back(acc[d], d);
        auto better = [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) {
          if (a.first != b.first) return a.first > b.first;
          return a.second < b.second;
        };
        size_t kk = std::min<size_t>(k, cand.size());
        std::partial_sort(cand.begin(), cand.begin() + (long)kk, cand.end(), better);
        // documents sharing no component score 0 and are eligible in a flat scan
        if (kk < k || !(cand[kk - 1].first > 0.0f)) {
          size_t added = 0;
          for (uint32_t d = 0; d < ix.n_docs && added < k; ++d)
            if (!seen[d]) {
              cand.emplace_back(0.0f, d);
              ++added;
            }
          kk = std::min<size_t>(k, cand.size());
          std::partial_sort(cand.begin(), cand.begin() + (long)kk, cand.end(), better);
        }
        out_n[q] = (uint32_t)kk;
        for (size_t i = 0; i < kk; ++i) {
          out_scores[(size_t)q * k + i] = cand[i].first;
          out_ids[(size_t)q * k + i] = cand[i].second;
        }
        for (uint32_t d : touched) {
          acc[d] = 0.0f;
          seen[d] = 0;
        }
      } catch (const std::bad_alloc&) {
        oom = 1;
        out_n[q] = 0;
      }
    }
    if (oom) return fail(SGPU_ENOMEM, "out of host memory in exact search");
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory in exact search");
  }
  return SGPU_OK;
}

}  // namespace sgpu
