// pack_index.cpp — the host half of sgpu_index_upload: the canonical arrays of a HostIndex packed into
// the HBM layout of DESIGN.md section 2 (document records, posting refs, summary split points and
// dequantised summary values). Host code with OpenMP (device_index.hip is compiled by hipcc without it);
// every output is sized before its parallel loop, so nothing is allocated inside a parallel region.
#include <algorithm>
#include <cstring>

#include "host_index.hpp"

namespace sgpu {

static inline uint32_t row_dir_bucket_host(uint32_t key, uint32_t n_buckets) {   // == device_types.hpp: row_dir_bucket
  return (uint32_t)(((uint64_t)(uint32_t)(key * 2654435761u) * n_buckets) >> 32);
}

// ---- compressed component stream (SGPU_VAL_DOTVBYTE: search_kernel.inc VT_DVB; and, since r05, the sliced layout of an
// f16 index over u16 components: VT_F16S) ---------------------------------------------------------------------------
// A document is stored as 12-byte slices of eight elements - the slice's first component in 16 bits, the gaps of
// elements 1 .. 3 in 12 bits each, those of elements 4 .. 7 in 11 bits each - when every gap fits its field; otherwise
// it keeps the raw record form and its refs carry kDvbRawBit in the length field. `raw` is EMPTY for an index that is
// not sliced at all (every pack_* function below keys on that), else one flag per document.
static constexpr uint64_t kDvbRawBit = 0x8000;
static inline uint32_t dvb_gap_limit(uint64_t i) { return (i & 7) <= 3 ? 4096u : 2048u; }   // element i of its slice (i & 7 != 0)
void pack_dvb_raw_flags(const HostIndex& h, bool f16_slices, std::vector<uint8_t>* out) {
  std::vector<uint8_t>& raw = *out;
  const bool sliced = h.value_type == SGPU_VAL_DOTVBYTE || (f16_slices && pack_f16_slices_possible(h));
  raw.assign(sliced ? h.n_docs : 0, 0);
  if (raw.empty()) return;
  const uint16_t* comps = (const uint16_t*)h.fwd_comps.data();
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
  for (int64_t doc = 0; doc < (int64_t)h.n_docs; ++doc) {
    const uint64_t s0 = h.fwd_offsets[(size_t)doc], s1 = h.fwd_offsets[(size_t)doc + 1];
    uint8_t r = s1 - s0 > 32767 ? 1 : 0;   // (the raw flag takes bit 15 of the length field; an f16 document that long is stored raw - and cannot be: see pack_f16_slices_possible)
    for (uint64_t i = s0 + 1; i < s1; ++i)
      if (((i - s0) & 7) != 0 && (uint32_t)comps[i] - (uint32_t)comps[i - 1] >= dvb_gap_limit(i - s0)) r = 1;
    raw[(size_t)doc] = r;
  }
}
// An f16 index can take the sliced layout when its components are u16 and no document has more than 32767 of them (the
// raw flag lives in bit 15 of a ref's length field).
bool pack_f16_slices_possible(const HostIndex& h) {
  if (h.value_type != SGPU_VAL_F16 || h.comp_width != 2) return false;
  for (uint64_t d = 0; d < h.n_docs; ++d)
    if (h.fwd_offsets[d + 1] - h.fwd_offsets[d] > 32767) return false;
  return true;
}
// bytes of a document's record (before the padding to 16)
static inline uint64_t record_bytes(const HostIndex& h, const std::vector<uint8_t>& raw, uint64_t doc, uint64_t len) {
  const uint64_t npad = (len + 7) & ~7ull;
  if (!raw.empty() && !raw[doc]) {
    const uint64_t ns = npad / 8, vb8 = 8ull * h.val_bytes();
    if (h.val_bytes() == 1) return ns * 20;   // DotVByte (r05): [ns x 16 B: w0 w1 w2 | codes 0-3][ns x 4 B: codes 4-7]
    return ((ns * 12 + vb8 - 1) & ~(vb8 - 1)) + ns * vb8;   // sliced f16: [ns x 12 B gaps][pad to 16 B][ns x 8 binary16 values]
  }
  return npad * (h.comp_width + h.val_bytes());
}
static inline uint64_t ref_len_field(const HostIndex& h, const std::vector<uint8_t>& raw, uint64_t doc, uint64_t len) {
  return len | ((!raw.empty() && raw[doc]) ? kDvbRawBit : 0ull);
}

// Record offsets in 16-byte units. A record is moved to the next `line16`-unit line only if it would
// otherwise touch more lines than its size needs (a sequential prefix: cheap, one pass).
void pack_record_offsets(const HostIndex& h, const std::vector<uint8_t>& raw, uint64_t line16, std::vector<uint64_t>* out) {
  std::vector<uint64_t>& rec_off16 = *out;
  rec_off16.assign(h.n_docs + 1, 0);
  uint64_t cur = 0;
  for (uint64_t doc = 0; doc < h.n_docs; ++doc) {
    const uint64_t len = h.fwd_offsets[doc + 1] - h.fwd_offsets[doc];
    const uint64_t size16 = (record_bytes(h, raw, doc, len) + 15) / 16;
    const uint64_t in_line = cur % line16;
    if (size16 && (in_line + size16 + line16 - 1) / line16 > (size16 + line16 - 1) / line16) cur += line16 - in_line;
    rec_off16[doc] = cur;
    cur += size16;
  }
  rec_off16[h.n_docs] = cur;
}

// Document records: [npad components][npad values (f16, or u8 codes)], npad = len rounded up to 8, the
// record padded to 16 bytes. Padding components carry the sentinel id `dim` (never a query component)
// when it is representable; their values are 0.
void pack_records(const HostIndex& h, const std::vector<uint8_t>& raw, const std::vector<uint64_t>& rec_off16, std::vector<uint8_t>* out) {
  const uint32_t cw = h.comp_width, vb = h.val_bytes();
  std::vector<uint8_t>& fwd = *out;
  fwd.assign(std::max<uint64_t>(rec_off16[h.n_docs] * 16, 16), 0);
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
  for (int64_t doc = 0; doc < (int64_t)h.n_docs; ++doc) {
    const uint64_t s0 = h.fwd_offsets[(size_t)doc], len = h.fwd_offsets[(size_t)doc + 1] - s0;
    const uint64_t npad = (len + 7) & ~7ull;
    uint8_t* rec = fwd.data() + rec_off16[(size_t)doc] * 16;
    if (!raw.empty() && !raw[(size_t)doc]) {
      // per slice three dwords = 96 bits: [0,16) first component | [16,28) [28,40) [40,52) gaps of elements 1 .. 3 |
      // [52,63) [63,74) [74,85) [85,96) gaps of elements 4 .. 7; padding elements: gap 0, code 0
      const uint16_t* comps = (const uint16_t*)h.fwd_comps.data() + s0;
      const uint64_t ns = npad / 8;
      uint32_t* gw = (uint32_t*)rec;
      const uint64_t vb8 = 8ull * vb;
      uint8_t* codes = rec + ((ns * 12 + vb8 - 1) & ~(vb8 - 1));   // (sliced f16: the binary16 values behind the 12-byte slices)
      if (vb == 1) {
        // DotVByte (r05): a slice's three gap words and its first four codes share ONE aligned 16-byte unit, the other
        // four codes follow in a dword array: [ns x 16 B][ns x 4 B] (until r04: [ns x 12 B][pad][ns x 8 B])
        const uint8_t* cd = h.fwd_codes.data() + s0;
        uint32_t* hi = (uint32_t*)(rec + ns * 16);
        for (uint64_t sl = 0; sl < ns; ++sl) {
          uint32_t g[8], cw[2] = {0, 0};
          g[0] = comps[sl * 8];
          for (uint64_t i = 1; i < 8; ++i) {
            const uint64_t e = sl * 8 + i;
            g[i] = e < len ? (uint32_t)comps[e] - (uint32_t)comps[e - 1] : 0u;
          }
          for (uint64_t i = 0; i < 8; ++i) {
            const uint64_t e = sl * 8 + i;
            if (e < len) cw[i >> 2] |= (uint32_t)cd[e] << (8 * (i & 3));
          }
          gw[4 * sl + 0] = g[0] | (g[1] << 16) | (g[2] << 28);
          gw[4 * sl + 1] = (g[2] >> 4) | (g[3] << 8) | (g[4] << 20) | (g[5] << 31);
          gw[4 * sl + 2] = (g[5] >> 1) | (g[6] << 10) | (g[7] << 21);
          gw[4 * sl + 3] = cw[0];
          hi[sl] = cw[1];
        }
        continue;
      }
      for (uint64_t sl = 0; sl < ns; ++sl) {
        uint32_t g[8];
        g[0] = comps[sl * 8];
        for (uint64_t i = 1; i < 8; ++i) {
          const uint64_t e = sl * 8 + i;
          g[i] = e < len ? (uint32_t)comps[e] - (uint32_t)comps[e - 1] : 0u;
        }
        gw[3 * sl + 0] = g[0] | (g[1] << 16) | (g[2] << 28);
        gw[3 * sl + 1] = (g[2] >> 4) | (g[3] << 8) | (g[4] << 20) | (g[5] << 31);
        gw[3 * sl + 2] = (g[5] >> 1) | (g[6] << 10) | (g[7] << 21);
      }
      std::memcpy(codes, h.fwd_vals.data() + s0, len * 2);
      continue;
    }
    std::memcpy(rec, h.fwd_comps.data() + s0 * cw, len * cw);
    if (vb == 2) std::memcpy(rec + npad * cw, h.fwd_vals.data() + s0, len * 2);
    else std::memcpy(rec + npad * cw, h.fwd_codes.data() + s0, len);
    if (cw == 2 && h.dim <= 65535) {
      for (uint64_t e = len; e < npad; ++e) ((uint16_t*)rec)[e] = (uint16_t)h.dim;
    } else if (cw == 4) {
      for (uint64_t e = len; e < npad; ++e) ((uint32_t*)rec)[e] = (uint32_t)h.dim;
    }
  }
}

// Block-major store: bsize[b] = 16-byte units before block b (every block starts on a 128-byte line).
void pack_block_sizes(const HostIndex& h, const std::vector<uint8_t>& raw, std::vector<uint64_t>* out) {
  const uint64_t nb = h.n_blocks();
  std::vector<uint64_t>& bsize = *out;
  bsize.assign(nb + 1, 0);
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
  for (int64_t b = 0; b < (int64_t)nb; ++b) {
    uint64_t u = 0;
    for (uint64_t p = h.block_post_start[(size_t)b]; p < h.block_post_start[(size_t)b + 1]; ++p) {
      const uint32_t doc = h.post_doc[p];
      const uint64_t len = h.fwd_offsets[doc + 1] - h.fwd_offsets[doc];
      u += (record_bytes(h, raw, doc, len) + 15) / 16;
    }
    bsize[(size_t)b + 1] = (u + 7) & ~7ull;
  }
  for (uint64_t b = 0; b < nb; ++b) bsize[b + 1] += bsize[b];
}

// Posting refs (record offset / 16) << 16 | len, the reference's PackedPostingBlock (src/posting_list.rs:32-60).
// Block-major: inside a block the records are grouped by the scoring loop's length class (<= 128 elements
// first, longer ones after; posting order within a class), so consecutive items of a class are adjacent records.
void pack_post_refs(const HostIndex& h, const std::vector<uint8_t>& raw, const std::vector<uint64_t>& rec_off16,
                    const std::vector<uint64_t>& bsize, bool block_major, uint64_t blk_base, std::vector<uint64_t>* out) {
  const uint64_t nb = h.n_blocks();
  std::vector<uint64_t>& pref = *out;
  pref.assign(h.n_postings(), 0);
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
  for (int64_t b = 0; b < (int64_t)nb; ++b) {
    uint64_t cur = blk_base + bsize[(size_t)b];
    for (int cls = 0; cls < 2; ++cls)
      for (uint64_t p = h.block_post_start[(size_t)b]; p < h.block_post_start[(size_t)b + 1]; ++p) {
        const uint32_t doc = h.post_doc[p];
        const uint64_t len = h.fwd_offsets[doc + 1] - h.fwd_offsets[doc];
        if ((len > 128) != (cls == 1)) continue;
        pref[p] = ((block_major ? cur : rec_off16[doc]) << 16) | ref_len_field(h, raw, doc, len);
        cur += (record_bytes(h, raw, doc, len) + 15) / 16;
      }
  }
}

void pack_doc_refs(const HostIndex& h, const std::vector<uint8_t>& raw, const std::vector<uint64_t>& rec_off16, std::vector<uint64_t>* out) {
  std::vector<uint64_t>& dref = *out;
  dref.assign(h.n_docs, 0);
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
  for (int64_t doc = 0; doc < (int64_t)h.n_docs; ++doc)
    dref[(size_t)doc] = (rec_off16[(size_t)doc] << 16) |
                        ref_len_field(h, raw, (uint64_t)doc, h.fwd_offsets[(size_t)doc + 1] - h.fwd_offsets[(size_t)doc]);
}

void pack_narrow(const std::vector<uint64_t>& v, std::vector<uint32_t>* out) {
  std::vector<uint32_t>& o = *out;
  o.assign(v.size(), 0);
#pragma omp parallel for schedule(static) num_threads(sgpu::host_threads())
  for (int64_t i = 0; i < (int64_t)v.size(); ++i) o[(size_t)i] = (uint32_t)v[(size_t)i];
}

// Split point of every summary row at half the list's block ids (rows are ascending in block id,
// validate_desc): stage 1 gives each half of a list to its own wavefront.
void pack_row_mid(const HostIndex& h, std::vector<uint16_t>* out) {
  std::vector<uint16_t>& mid = *out;
  mid.assign(h.n_rows(), 0);
#pragma omp parallel for schedule(dynamic, 64) num_threads(sgpu::host_threads())
  for (int64_t c = 0; c < (int64_t)h.dim; ++c) {
    const uint64_t nb = h.list_block_start[(size_t)c + 1] - h.list_block_start[(size_t)c];
    const uint16_t half = (uint16_t)((nb + 1) / 2);
    for (uint64_t r = h.list_row_start[(size_t)c]; r < h.list_row_start[(size_t)c + 1]; ++r) {
      const uint16_t* b = h.sum_bid.data() + h.row_ptr[r];
      const uint16_t* e = h.sum_bid.data() + h.row_ptr[r + 1];
      mid[r] = (uint16_t)(std::lower_bound(b, e, half) - b);
    }
  }
}

// Hashed row directory (DevView::row_dir): for every summary row its {key, first entry, entries, split point} in the
// first bucket from row_dir_bucket(key) on with a free slot; as many 4-slot buckets as leave the table at most 60 % full.
// Built on all host threads (a slot is claimed with a compare-and-swap on its key word). Returns false when the index
// cannot have one (u32 components, dim 65536).
bool pack_row_dir(const HostIndex& h, const std::vector<uint16_t>& mid, std::vector<uint32_t>* out, uint32_t* n_buckets_out) {
  out->clear();
  const uint64_t n_rows = h.n_rows();
  if (h.comp_width != 2 || h.dim > 65535) return false;   // (an index without rows gets one empty bucket)
  const uint64_t n_buckets = std::max<uint64_t>(1, (n_rows * 10 / 6 + 3) / 4);   // 4-slot buckets, at most 60 % full
  if (n_buckets >= (1ull << 31)) return false;
  out->assign(n_buckets * 16, 0xffffffffu);   // four words per slot, every key empty
  uint32_t* tab = out->data();
  const uint16_t* rc = (const uint16_t*)h.row_comp.data();
#pragma omp parallel for schedule(dynamic, 64) num_threads(sgpu::host_threads())
  for (int64_t c = 0; c < (int64_t)h.dim; ++c)
    for (uint64_t r = h.list_row_start[(size_t)c]; r < h.list_row_start[(size_t)c + 1]; ++r) {
      const uint32_t key = ((uint32_t)c << 16) | rc[r];
      const uint64_t start = h.row_ptr[r], len = h.row_ptr[r + 1] - start;
      uint64_t b = row_dir_bucket_host(key, (uint32_t)n_buckets);
      for (bool placed = false; !placed; b = b + 1 == n_buckets ? 0 : b + 1)
        for (uint32_t s = 0; s < 4 && !placed; ++s) {
          uint32_t* e = tab + (b * 4 + s) * 4;
          uint32_t expect = 0xffffffffu;
          if (__atomic_compare_exchange_n(&e[0], &expect, key, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
            e[1] = (uint32_t)start;
            e[2] = (uint32_t)(start >> 32) | ((uint32_t)len << 16);
            e[3] = mid[r];
            placed = true;
          }
        }
    }
  *n_buckets_out = (uint32_t)n_buckets;
  return true;
}

// Dequantised summary values: code * quant + min with the reference's two roundings
// (src/quantized_summary.rs:102-104; this file is compiled with -ffp-contract=off).
void pack_sum_deq(const HostIndex& h, std::vector<float>* out) {
  std::vector<float>& deq = *out;
  deq.assign(h.n_entries(), 0.0f);
#pragma omp parallel for schedule(dynamic, 64) num_threads(sgpu::host_threads())
  for (int64_t c = 0; c < (int64_t)h.dim; ++c) {
    const uint64_t b0 = h.list_block_start[(size_t)c];
    for (uint64_t r = h.list_row_start[(size_t)c]; r < h.list_row_start[(size_t)c + 1]; ++r)
      for (uint64_t e = h.row_ptr[r]; e < h.row_ptr[r + 1]; ++e) {
        const uint64_t blk = b0 + h.sum_bid[e];
        const volatile float t = (float)h.sum_code[e] * h.blk_quant[blk];
        deq[e] = t + h.blk_min[blk];
      }
  }
}

}  // namespace sgpu
