// host_index.cpp — descriptor validation, ownership, own flat SoA file format.
//
// The reference persists indexes with vectorium's IndexSerializer
// (src/pylib/mod.rs:186-221); that wire format is not in the reference tree, so
// real *.index.seismic files cannot be read. This file defines a versioned flat
// format of the canonical arrays instead.
#include "host_index.hpp"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <exception>

namespace sgpu {

static double cgroup_cpu_quota() {   // CPUs' worth of time per period, 0 = none
  // (SGPU_CGROUP_ROOT: where the cgroup files are looked for - a test hook, tests/test_abi_and_host.py)
  const char* root_env = std::getenv("SGPU_CGROUP_ROOT");
  const std::string root = root_env && *root_env ? root_env : "/sys/fs/cgroup";
  double q = 0, per = 0;
  char word[32] = {0};
  if (FILE* f = std::fopen((root + "/cpu.max").c_str(), "r")) {   // cgroup v2: "max 100000" or "1600000 100000"
    const int n = std::fscanf(f, "%31s %lf", word, &per);
    std::fclose(f);
    if (n == 2 && per > 0 && std::strcmp(word, "max") != 0) return std::atof(word) / per;
    if (n == 2) return 0;
  }
  if (FILE* f = std::fopen((root + "/cpu/cpu.cfs_quota_us").c_str(), "r")) {   // v1
    const int n = std::fscanf(f, "%lf", &q);
    std::fclose(f);
    if (n == 1 && q > 0) {
      if (FILE* g = std::fopen((root + "/cpu/cpu.cfs_period_us").c_str(), "r")) {
        const int m = std::fscanf(g, "%lf", &per);
        std::fclose(g);
        if (m == 1 && per > 0) return q / per;
      }
    }
  }
  return 0;
}
int default_host_threads(int omp_max_threads) {
  if (const char* e = std::getenv("SGPU_HOST_THREADS")) {
    const int v = std::atoi(e);
    if (v > 0) return v;
  }
  int nt = omp_max_threads > 0 ? omp_max_threads : 1;
  const double quota = cgroup_cpu_quota();   // (two small files per host-parallel phase: read every time, a quota may change)
  if (quota > 0) {
    const int cap = (int)(quota + 0.999);
    if (cap >= 1 && cap < nt) nt = cap;
  }
  return nt;
}
std::string& last_error() {
  static thread_local std::string e;
  return e;
}

sgpu_status fail(sgpu_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return st;
}

void HostIndex::fill_desc(sgpu_index_desc* d) const {
  std::memset(d, 0, sizeof *d);
  d->comp_width = comp_width;
  d->value_type = value_type;
  d->val_scale = value_type == SGPU_VAL_F16 ? 0.0f : val_scale;
  d->n_docs = n_docs;
  d->dim = dim;
  d->nnz = nnz();
  d->n_blocks = n_blocks();
  d->n_postings = n_postings();
  d->n_rows = n_rows();
  d->n_entries = n_entries();
  d->fwd_offsets = fwd_offsets.data();
  d->fwd_comps = fwd_comps.data();
  d->fwd_vals = value_type == SGPU_VAL_F16 ? (const void*)fwd_vals.data() : (const void*)fwd_codes.data();
  d->list_block_start = list_block_start.data();
  d->block_post_start = block_post_start.data();
  d->post_doc = post_doc.data();
  d->blk_min = blk_min.data();
  d->blk_quant = blk_quant.data();
  d->list_row_start = list_row_start.data();
  d->row_comp = row_comp.data();
  d->row_ptr = row_ptr.data();
  d->sum_bid = sum_bid.data();
  d->sum_code = sum_code.data();
}

static bool monotone(const uint64_t* a, uint64_t n_plus_1, uint64_t last) {
  if (a[0] != 0) return false;
  for (uint64_t i = 1; i < n_plus_1; ++i)
    if (a[i] < a[i - 1]) return false;
  return a[n_plus_1 - 1] == last;
}

sgpu_status validate_desc(const sgpu_index_desc& d) {
  if (d.comp_width != 2 && d.comp_width != 4) return fail(SGPU_EINVAL, "comp_width must be 2 or 4");
  if (d.dim == 0) return fail(SGPU_EINVAL, "dim == 0");
  if (d.value_type > SGPU_VAL_DOTVBYTE) return fail(SGPU_EINVAL, "unknown value_type %u", d.value_type);
  if (d.value_type == SGPU_VAL_DOTVBYTE && d.comp_width != 2)
    return fail(SGPU_EINVAL, "a DotVByte index has u16 components (reference src/pylib/dotvbyte.rs:20-27)");
  if (d.value_type != SGPU_VAL_F16) {
    int e = 0;
    if (!(d.val_scale > 0.0f) || std::frexp(d.val_scale, &e) != 0.5f) return fail(SGPU_EINVAL, "val_scale must be a positive power of two");
  }
  // (65535, not 65536: the records pad with the sentinel id `dim` and the device's row directory keys a row by
  // list << 16 | component with 0xffffffff as its empty marker - an index of 65536 u16 ids could be built and saved but not
  // uploaded; ADVICE r05. Use u32 components for such a vocabulary.)
  if (d.comp_width == 2 && d.dim > 65535) return fail(SGPU_EINVAL, "dim %llu does not fit u16 components (at most 65535; use comp_width 4)", (unsigned long long)d.dim);
  if (d.dim > 0xffffffffull || d.n_docs > 0x7fffffffull) return fail(SGPU_EINVAL, "dim/n_docs out of range");
  if (!d.fwd_offsets || !d.list_block_start || !d.block_post_start || !d.list_row_start || !d.row_ptr)
    return fail(SGPU_EINVAL, "null offset array in descriptor");
  if (!monotone(d.fwd_offsets, d.n_docs + 1, d.nnz)) return fail(SGPU_EINVAL, "fwd_offsets not monotone / nnz mismatch");
  if (!monotone(d.list_block_start, d.dim + 1, d.n_blocks)) return fail(SGPU_EINVAL, "list_block_start not monotone / n_blocks mismatch");
  if (!monotone(d.block_post_start, d.n_blocks + 1, d.n_postings)) return fail(SGPU_EINVAL, "block_post_start not monotone / n_postings mismatch");
  if (!monotone(d.list_row_start, d.dim + 1, d.n_rows)) return fail(SGPU_EINVAL, "list_row_start not monotone / n_rows mismatch");
  if (!monotone(d.row_ptr, d.n_rows + 1, d.n_entries)) return fail(SGPU_EINVAL, "row_ptr not monotone / n_entries mismatch");
  if ((d.nnz && (!d.fwd_comps || !d.fwd_vals)) || (d.n_postings && !d.post_doc) ||
      (d.n_blocks && (!d.blk_min || !d.blk_quant)) || (d.n_rows && !d.row_comp) ||
      (d.n_entries && (!d.sum_bid || !d.sum_code)))
    return fail(SGPU_EINVAL, "null data array in descriptor");
  auto compv = [&](const void* p, uint64_t i) -> uint32_t {
    return d.comp_width == 2 ? (uint32_t)((const uint16_t*)p)[i] : ((const uint32_t*)p)[i];
  };
  for (uint64_t doc = 0; doc < d.n_docs; ++doc) {
    const uint64_t s = d.fwd_offsets[doc], e = d.fwd_offsets[doc + 1];
    if (e - s > 65535) return fail(SGPU_EINVAL, "document %llu has more than 65535 components (16-bit length, reference src/posting_list.rs:45-48)", (unsigned long long)doc);
    if (d.value_type == SGPU_VAL_DOTVBYTE && e - s > 32767) return fail(SGPU_ELIMIT, "document %llu has more than 32767 components (DotVByte forward index)", (unsigned long long)doc);
    for (uint64_t i = s; i < e; ++i) {
      const uint32_t c = compv(d.fwd_comps, i);
      if (c >= d.dim) return fail(SGPU_EINVAL, "document component >= dim");
      if (i > s && c <= compv(d.fwd_comps, i - 1)) return fail(SGPU_EINVAL, "document %llu components not strictly ascending", (unsigned long long)doc);
    }
  }
  for (uint64_t p = 0; p < d.n_postings; ++p)
    if (d.post_doc[p] >= d.n_docs) return fail(SGPU_EINVAL, "posting refers to doc >= n_docs");
  for (uint64_t c = 0; c < d.dim; ++c) {
    const uint64_t nb = d.list_block_start[c + 1] - d.list_block_start[c];
    if (nb > 65535) return fail(SGPU_EINVAL, "list %llu has more than 65535 blocks (reference src/posting_list.rs:243-246)", (unsigned long long)c);
    for (uint64_t r = d.list_row_start[c]; r < d.list_row_start[c + 1]; ++r) {
      const uint32_t rc = compv(d.row_comp, r);
      if (rc >= d.dim) return fail(SGPU_EINVAL, "summary row component >= dim");
      if (r > d.list_row_start[c] && rc <= compv(d.row_comp, r - 1)) return fail(SGPU_EINVAL, "summary rows of list %llu not strictly ascending", (unsigned long long)c);
      for (uint64_t e = d.row_ptr[r]; e < d.row_ptr[r + 1]; ++e) {
        if (d.sum_bid[e] >= nb) return fail(SGPU_EINVAL, "summary entry block id out of range");
        if (e > d.row_ptr[r] && d.sum_bid[e] <= d.sum_bid[e - 1]) return fail(SGPU_EINVAL, "summary row block ids not strictly ascending");
      }
    }
  }
  return SGPU_OK;
}

sgpu_status host_index_from_desc(const sgpu_index_desc& d, HostIndex* out) {
  sgpu_status st = validate_desc(d);
  if (st != SGPU_OK) return st;
  try {
    HostIndex& h = *out;
    h.comp_width = d.comp_width;
    h.n_docs = d.n_docs;
    h.dim = d.dim;
    h.fwd_offsets.assign(d.fwd_offsets, d.fwd_offsets + d.n_docs + 1);
    h.fwd_comps.assign((const uint8_t*)d.fwd_comps, (const uint8_t*)d.fwd_comps + d.nnz * d.comp_width);
    h.value_type = d.value_type;
    h.val_scale = d.value_type == SGPU_VAL_F16 ? 0.0f : d.val_scale;
    if (d.value_type == SGPU_VAL_F16) h.fwd_vals.assign((const uint16_t*)d.fwd_vals, (const uint16_t*)d.fwd_vals + d.nnz);
    else h.fwd_codes.assign((const uint8_t*)d.fwd_vals, (const uint8_t*)d.fwd_vals + d.nnz);
    h.list_block_start.assign(d.list_block_start, d.list_block_start + d.dim + 1);
    h.block_post_start.assign(d.block_post_start, d.block_post_start + d.n_blocks + 1);
    h.post_doc.assign(d.post_doc, d.post_doc + d.n_postings);
    h.blk_min.assign(d.blk_min, d.blk_min + d.n_blocks);
    h.blk_quant.assign(d.blk_quant, d.blk_quant + d.n_blocks);
    h.list_row_start.assign(d.list_row_start, d.list_row_start + d.dim + 1);
    h.row_comp.assign((const uint8_t*)d.row_comp, (const uint8_t*)d.row_comp + d.n_rows * d.comp_width);
    h.row_ptr.assign(d.row_ptr, d.row_ptr + d.n_rows + 1);
    h.sum_bid.assign(d.sum_bid, d.sum_bid + d.n_entries);
    h.sum_code.assign(d.sum_code, d.sum_code + d.n_entries);
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory copying the index descriptor");
  }
  return SGPU_OK;
}

// Query batch validation shared by the search entry points and the exact search.
sgpu_status validate_query_offsets(const uint64_t* q_off, uint32_t nq, uint32_t q_base, uint32_t* max_nnz) {
  if (!q_off) return fail(SGPU_EINVAL, "null q_off");
  uint32_t mx = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    if (q_off[q + 1] < q_off[q]) return fail(SGPU_EINVAL, "q_off not monotone");
    const uint64_t n = q_off[q + 1] - q_off[q];
    if (n > 0xffffu)
      return fail(SGPU_ELIMIT, "query %u has %llu components (limit 65535)", q_base + q, (unsigned long long)n);
    mx = std::max<uint32_t>(mx, (uint32_t)n);
  }
  if (q_off[nq] - q_off[0] >= 0xffffffffull) return fail(SGPU_ELIMIT, "batch too large");
  *max_nnz = mx;
  return SGPU_OK;
}

sgpu_status validate_queries(uint64_t dim, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                             uint32_t nq, uint32_t* max_nnz, uint32_t q_base) {
  if (!q_off || q_off[0] != 0) return fail(SGPU_EINVAL, "q_off[0] must be 0");
  uint32_t mx = 0;
  const sgpu_status ost = validate_query_offsets(q_off, nq, q_base, &mx);
  if (ost != SGPU_OK) return ost;
  if (q_off[nq] && (!comps || !vals)) return fail(SGPU_EINVAL, "null query arrays");
  // InvertedIndexBase::search asserts sorted components (reference src/inverted_index.rs:172-175)
  // and indexes posting_lists[component] (bounds panic, :193); duplicates are rejected too.
  // (serial on purpose: half a millisecond per 10 000 queries, where waking an OpenMP team on a
  // 256-thread host costs tens of milliseconds - this runs inside every search call)
  int bad_kind = 0;
  uint32_t bad_q = 0xffffffffu;
  const uint32_t dim32 = dim > 0xffffffffull ? 0xffffffffu : (uint32_t)dim;
  for (uint32_t q = 0; q < nq && !bad_kind; ++q) {
    // the common case - a valid query - is ONE branch-free pass the compiler vectorises (r05: this loop was a third of the
    // host side of a call); only a query that fails it is walked again to name what is wrong with it
    const uint64_t a = q_off[q], e = q_off[q + 1];
    uint32_t bad = 0;
    for (uint64_t i = a; i < e; ++i) bad |= (uint32_t)(comps[i] >= dim32) | (uint32_t)(vals[i] != vals[i]);
    for (uint64_t i = a + 1; i < e; ++i) bad |= (uint32_t)(comps[i] <= comps[i - 1]);
    if (!bad) continue;
    for (uint64_t i = a; i < e && !bad_kind; ++i) {
      if (comps[i] >= dim) bad_kind = 1;
      else if (i > a && comps[i] <= comps[i - 1]) bad_kind = 2;
      else if (std::isnan(vals[i])) bad_kind = 3;
    }
    bad_q = q_base + q;
  }
  if (bad_kind == 1) return fail(SGPU_EINVAL, "query %u: component >= dim", bad_q);
  if (bad_kind == 2) return fail(SGPU_EINVAL, "query %u: components must be strictly ascending", bad_q);
  if (bad_kind == 3) return fail(SGPU_EINVAL, "query %u: NaN value", bad_q);
  *max_nnz = mx;
  return SGPU_OK;
}

// ---- file format: "SGPUIDX2", header of 12 u64, then the arrays in desc order (+ the kNN graph) ----
static const char kMagic[8] = {'S', 'G', 'P', 'U', 'I', 'D', 'X', '2'};
static constexpr int kHdr = 12;

template <class T>
static bool wr(FILE* f, const std::vector<T>& v) {
  return v.empty() || fwrite(v.data(), sizeof(T), v.size(), f) == v.size();
}
template <class T>
static bool rd(FILE* f, std::vector<T>& v, uint64_t n) {
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

sgpu_status host_index_save(const HostIndex& ix, const char* path) {
  FILE* f = fopen(path, "wb");
  if (!f) return fail(SGPU_EIO, "cannot open %s for writing: %s", path, strerror(errno));
  // hdr[8] = neighbours per document, hdr[9] = total neighbour ids of the kNN graph (0, 0 = no graph);
  // the reference serialises InvertedIndexBase{.., knn} in one file too (src/inverted_index.rs:39-52)
  uint32_t scale_bits = 0;
  std::memcpy(&scale_bits, &ix.val_scale, 4);
  // (a graph without a single neighbour - no document had one - is written as "no graph": the loader takes
  // knn_dim == 0 and an empty neighbour array together)
  uint64_t hdr[kHdr] = {ix.comp_width,   ix.n_docs,   ix.dim,         ix.nnz(),   ix.n_blocks(),
                        ix.n_postings(), ix.n_rows(), ix.n_entries(), ix.knn.empty() ? 0u : ix.knn_dim, ix.knn.size(),
                        ix.value_type,   scale_bits};
  bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, 8, kHdr, f) == kHdr && wr(f, ix.fwd_offsets) &&
            wr(f, ix.fwd_comps) && wr(f, ix.fwd_vals) && wr(f, ix.fwd_codes) && wr(f, ix.list_block_start) &&
            wr(f, ix.block_post_start) && wr(f, ix.post_doc) && wr(f, ix.blk_min) && wr(f, ix.blk_quant) &&
            wr(f, ix.list_row_start) && wr(f, ix.row_comp) && wr(f, ix.row_ptr) && wr(f, ix.sum_bid) &&
            wr(f, ix.sum_code) && wr(f, ix.knn);
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail(SGPU_EIO, "short write to %s", path);
  return SGPU_OK;
}

sgpu_status host_index_load(const char* path, HostIndex* out) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail(SGPU_EIO, "cannot open %s: %s", path, strerror(errno));
  char magic[8];
  uint64_t hdr[kHdr] = {0};
  // SGPUIDX1 (round 1): 10-word header (no value type / scale words), f16 values - the same arrays otherwise
  static const char kMagic1[8] = {'S', 'G', 'P', 'U', 'I', 'D', 'X', '1'};
  bool got = fread(magic, 1, 8, f) == 8;
  const bool v1 = got && memcmp(magic, kMagic1, 8) == 0;
  got = got && (v1 || memcmp(magic, kMagic, 8) == 0);
  if (got && v1) {
    uint64_t h1[10];
    got = fread(h1, 8, 10, f) == 10;
    if (got) {   // the v2 header's first ten words; f16 values, no value scale
      for (int i = 0; i < 10; ++i) hdr[i] = h1[i];
      hdr[10] = SGPU_VAL_F16;
      hdr[11] = 0;
    }
  } else if (got) {
    got = fread(hdr, 8, kHdr, f) == kHdr;
  }
  if (!got) {
    fclose(f);
    return fail(SGPU_EIO, "%s is not an SGPUIDX1 / SGPUIDX2 index file", path);
  }
  HostIndex& h = *out;
  h.comp_width = (uint32_t)hdr[0];
  h.n_docs = hdr[1];
  h.dim = hdr[2];
  const uint64_t nnz = hdr[3], nb = hdr[4], np = hdr[5], nr = hdr[6], ne = hdr[7], knn_dim = hdr[8], nk = hdr[9];
  h.value_type = (uint32_t)hdr[10];
  {
    const uint32_t scale_bits = (uint32_t)hdr[11];
    std::memcpy(&h.val_scale, &scale_bits, 4);
  }
  bool ok = (h.comp_width == 2 || h.comp_width == 4) && h.value_type <= SGPU_VAL_DOTVBYTE &&
            (h.value_type != SGPU_VAL_DOTVBYTE || h.comp_width == 2);
  const uint64_t vb = h.value_type == SGPU_VAL_F16 ? 2 : 1;
  // the header is untrusted: the counts must add up to the file's size before anything is resized
  if (ok) {
    const uint64_t lim = 1ull << 48;
    ok = h.n_docs < lim && h.dim < lim && nnz < lim && nb < lim && np < lim && nr < lim && ne < lim && nk < lim &&
         knn_dim <= 0xffffffffull && ((knn_dim == 0) == (nk == 0));
    long pos = ftell(f);
    ok = ok && pos >= 0 && fseek(f, 0, SEEK_END) == 0;
    const long end = ok ? ftell(f) : -1;
    ok = ok && end >= 0 && fseek(f, pos, SEEK_SET) == 0;
    if (ok) {
      const uint64_t cw = h.comp_width;
      const uint64_t need = 8 * (h.n_docs + 1) + nnz * cw + nnz * vb + 8 * (h.dim + 1) + 8 * (nb + 1) + 4 * np + 8 * nb +
                            8 * (h.dim + 1) + nr * cw + 8 * (nr + 1) + 3 * ne + 4 * nk;
      ok = (uint64_t)(end - pos) == need;
    }
  }
  if (!ok) {
    fclose(f);
    return fail(SGPU_EIO, "corrupt index file %s (header does not match the file size)", path);
  }
  try {
    ok = rd(f, h.fwd_offsets, h.n_docs + 1) && rd(f, h.fwd_comps, nnz * h.comp_width) &&
         rd(f, h.fwd_vals, vb == 2 ? nnz : 0) && rd(f, h.fwd_codes, vb == 1 ? nnz : 0) &&
         rd(f, h.list_block_start, h.dim + 1) && rd(f, h.block_post_start, nb + 1) &&
         rd(f, h.post_doc, np) && rd(f, h.blk_min, nb) && rd(f, h.blk_quant, nb) &&
         rd(f, h.list_row_start, h.dim + 1) && rd(f, h.row_comp, nr * h.comp_width) && rd(f, h.row_ptr, nr + 1) &&
         rd(f, h.sum_bid, ne) && rd(f, h.sum_code, ne) && rd(f, h.knn, nk);
  } catch (const std::bad_alloc&) {
    fclose(f);
    return fail(SGPU_ENOMEM, "out of host memory loading %s", path);
  } catch (const std::exception& e) {
    fclose(f);
    return fail(SGPU_EIO, "cannot load %s: %s", path, e.what());
  }
  fclose(f);
  if (!ok) return fail(SGPU_EIO, "truncated or corrupt index file %s", path);
  h.knn_dim = (uint32_t)knn_dim;
  for (uint32_t id : h.knn)
    if (id >= h.n_docs) return fail(SGPU_EIO, "corrupt index file %s: kNN neighbour id >= n_docs", path);
  sgpu_index_desc d;
  h.fill_desc(&d);
  return validate_desc(d);
}

// InvertedIndexBase::convert_dataset_into: same lists / blocks / summaries, the forward index re-encoded.
sgpu_status host_index_convert(const HostIndex& src, uint32_t value_type, HostIndex* out) {
  if (value_type > SGPU_VAL_DOTVBYTE) return fail(SGPU_EINVAL, "unknown value_type %u", value_type);
  // (either component width for fixed-u8: the reference's "fixedu8" value type goes with u16 and with u32 components,
  // src/bin/perf_inverted_index.rs:110-126; its DotVByte class is u16-only, src/pylib/dotvbyte.rs:20-27)
  if (value_type == SGPU_VAL_DOTVBYTE) {
    if (src.comp_width != 2) return fail(SGPU_EINVAL, "a DotVByte index has u16 components (reference src/pylib/dotvbyte.rs:20-27)");
    for (uint64_t doc = 0; doc < src.n_docs; ++doc)
      if (src.fwd_offsets[doc + 1] - src.fwd_offsets[doc] > 32767)
        return fail(SGPU_ELIMIT, "document %llu has more than 32767 components (DotVByte forward index)", (unsigned long long)doc);
  }
  try {
    *out = src;
    HostIndex& h = *out;
    const uint64_t nnz = src.nnz();
    if (value_type == src.value_type) return SGPU_OK;
    // DotVByte <-> fixed-u8: the same codes, the component stream is (de)compressed at upload
    if (value_type != SGPU_VAL_F16 && src.value_type != SGPU_VAL_F16) {
      h.value_type = value_type;
      return SGPU_OK;
    }
    if (value_type != SGPU_VAL_F16) {
      // [restated, parity unpinned] step = smallest power of two with 255 * step >= max value
      float vmax = 0.0f;
      for (uint64_t i = 0; i < nnz; ++i) {
        const float v = src.val(i);
        if (v > vmax) vmax = v;
      }
      float step = 0.00390625f;   // 2^-8: Q0.8
      while (255.0f * step < vmax) step *= 2.0f;
      h.value_type = value_type;
      h.val_scale = step;
      h.fwd_codes.resize(nnz);
      h.fwd_vals.clear();
      h.fwd_vals.shrink_to_fit();
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
      for (int64_t i = 0; i < (int64_t)nnz; ++i) {
        const float r = std::round(src.val((uint64_t)i) / step);   // exact division (power of two); half away from zero
        h.fwd_codes[(size_t)i] = r >= 255.0f ? 255 : (r > 0.0f ? (uint8_t)r : 0);
      }
    } else {
      h.value_type = SGPU_VAL_F16;
      h.val_scale = 0.0f;
      h.fwd_vals.resize(nnz);
      for (uint64_t i = 0; i < nnz; ++i) h.fwd_vals[i] = f32_to_f16_sat(src.val(i));
      h.fwd_codes.clear();
      h.fwd_codes.shrink_to_fit();
    }
  } catch (const std::bad_alloc&) {
    return fail(SGPU_ENOMEM, "out of host memory converting the index");
  }
  return SGPU_OK;
}

}  // namespace sgpu
