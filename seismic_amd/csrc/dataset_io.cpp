// dataset_io.cpp — Seismic's inner binary dataset format and the result TSV of perf_inverted_index.
//
// Inner format (written by the reference's scripts/convert_json_to_inner_format.py:10-27 and read by
// vectorium's read_seismic_format, call sites src/pylib/mod.rs:987,1127): little endian
//     u32 n_vecs; then per vector: u32 n, n x u32 components (ascending), n x f32 values.
// Result TSV (src/bin/perf_inverted_index.rs:223-235): one line per result,
//     query_index \t doc_id \t rank (1-based) \t score
// which scripts/run_experiments.py:287-309 compares with groundtruth.tsv of the same layout.
#include <cerrno>
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace sgpu {

struct File {
  FILE* f;
  explicit File(FILE* f_) : f(f_) {}
  ~File() {
    if (f) fclose(f);
  }
};

}  // namespace sgpu

using namespace sgpu;

extern "C" {

sgpu_status sgpu_dataset_read(const char* path, uint64_t* n_vecs, uint64_t* nnz, uint64_t* offsets,
                              uint32_t* comps, float* vals) {
  if (!path || !n_vecs || !nnz) return fail(SGPU_EINVAL, "null argument");
  File in(fopen(path, "rb"));
  if (!in.f) return fail(SGPU_EIO, "cannot open %s: %s", path, strerror(errno));
  static_assert(sizeof(float) == 4, "f32");
  uint32_t n = 0;
  if (fread(&n, 4, 1, in.f) != 1) return fail(SGPU_EIO, "%s: missing vector count", path);
  if (offsets && (!comps || !vals)) return fail(SGPU_EINVAL, "offsets given without components / values buffers");
  const bool fill = offsets != nullptr;
  const uint64_t cap_vecs = *n_vecs, cap_nnz = *nnz;
  if (fill && cap_vecs < n) return fail(SGPU_EINVAL, "%s holds %u vectors, room for %llu", path, n, (unsigned long long)cap_vecs);
  uint64_t total = 0;
  if (fill) offsets[0] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t len = 0;
    if (fread(&len, 4, 1, in.f) != 1) return fail(SGPU_EIO, "%s: truncated at vector %u", path, i);
    if (fill) {
      if (total + len > cap_nnz) return fail(SGPU_EINVAL, "%s: more entries than the caller sized for", path);
      if (len && (fread(comps + total, 4, len, in.f) != len || fread(vals + total, 4, len, in.f) != len))
        return fail(SGPU_EIO, "%s: truncated inside vector %u", path, i);
      offsets[i + 1] = total + len;
    } else if (len && fseek(in.f, (long)len * 8, SEEK_CUR) != 0) {
      return fail(SGPU_EIO, "%s: truncated inside vector %u", path, i);
    }
    total += len;
  }
  if (!fill) {   // a seek past the end succeeds: check the size once
    const long pos = ftell(in.f);
    if (pos < 0 || fseek(in.f, 0, SEEK_END) != 0 || ftell(in.f) < pos) return fail(SGPU_EIO, "%s: truncated", path);
  }
  *n_vecs = n;
  *nnz = total;
  return SGPU_OK;
}

sgpu_status sgpu_dataset_write(const char* path, uint64_t n_vecs, const uint64_t* offsets, const uint32_t* comps,
                               const float* vals) {
  if (!path || !offsets || (offsets[n_vecs] && (!comps || !vals))) return fail(SGPU_EINVAL, "null argument");
  if (n_vecs > 0xffffffffull) return fail(SGPU_ELIMIT, "the inner format counts vectors in 32 bits");
  File out(fopen(path, "wb"));
  if (!out.f) return fail(SGPU_EIO, "cannot open %s for writing: %s", path, strerror(errno));
  const uint32_t n = (uint32_t)n_vecs;
  bool ok = fwrite(&n, 4, 1, out.f) == 1;
  for (uint64_t i = 0; ok && i < n_vecs; ++i) {
    if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0xffffffffull) return fail(SGPU_EINVAL, "offsets not monotone");
    const uint32_t len = (uint32_t)(offsets[i + 1] - offsets[i]);
    ok = fwrite(&len, 4, 1, out.f) == 1 &&
         (len == 0 || (fwrite(comps + offsets[i], 4, len, out.f) == len && fwrite(vals + offsets[i], 4, len, out.f) == len));
  }
  FILE* f = out.f;
  out.f = nullptr;
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail(SGPU_EIO, "short write to %s", path);
  return SGPU_OK;
}

sgpu_status sgpu_results_write_tsv(const char* path, uint32_t nq, uint32_t k, const float* scores,
                                   const uint64_t* doc_ids, const uint32_t* n) {
  if (!path || (nq && (!scores || !doc_ids || !n))) return fail(SGPU_EINVAL, "null argument");
  File out(fopen(path, "w"));
  if (!out.f) return fail(SGPU_EIO, "cannot open %s for writing: %s", path, strerror(errno));
  bool ok = true;
  for (uint32_t q = 0; ok && q < nq; ++q)
    for (uint32_t i = 0; ok && i < n[q] && i < k; ++i)
      // Rust's `{}` prints the shortest decimal that round-trips the f32: %.9g round-trips too
      ok = fprintf(out.f, "%u\t%llu\t%u\t%.9g\n", q, (unsigned long long)doc_ids[(size_t)q * k + i], i + 1,
                   (double)scores[(size_t)q * k + i]) > 0;
  FILE* f = out.f;
  out.f = nullptr;
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail(SGPU_EIO, "short write to %s", path);
  return SGPU_OK;
}

}  // extern "C"
