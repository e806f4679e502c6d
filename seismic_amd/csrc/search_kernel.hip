// search_kernel.hip — the Seismic search hot path as one persistent gfx950 kernel.
//
// One query per workgroup (persistent workgroups pull queries from an atomic
// queue, longest-expected first). Per query, following InvertedIndexBase::search
// (reference src/inverted_index.rs:153-234):
//
//   stage 0  query -> LDS: sorted (component, value) pairs; the query_cut heaviest
//            components are ranked (k_largest_by, src/inverted_index.rs:187-190)
//            -> the posting lists to walk, in order.
//   stage 1  hot loop A, QuantizedSummary::distances (src/quantized_summary.rs:64-160)
//            for ALL selected lists at once (a pure function of the query):
//            matching summary rows are located by binary search; one wavefront per
//            list then streams the list's matched rows in ascending query component
//            order, 64 (block id, dequantised value) entries per step, the loads of
//            the next 16 steps in flight, and adds value * weight to f32
//            accumulators in LDS. A wavefront issues its DS operations in order and
//            block ids are distinct within a row, so every accumulator receives its
//            additions in ascending query component order with the reference's
//            roundings ((code*quant + min) * qv, then +=; no FMA; code*quant + min
//            is precomputed once at upload): BIT-EXACT.
//   lookup   the query lookup table (cleared during stage 1 by the wavefronts that
//            have no list): one byte per vocabulary id (1 + rank in the query,
//            0 = absent) when it fits, else {32 bits, rank} per 32 ids.
//   stage 2  hot loop B, PostingList::search / sort_and_search /
//            evaluate_posting_block (src/posting_list.rs:115-215), list by list.
//            The reference's skip test reads the LIVE k-th best score, so the set of
//            scored documents depends on the traversal order. The threshold only
//            rises, hence testing a block against an OLDER threshold can only admit
//            more blocks. Each round therefore
//              (a) filters the remaining blocks against the current threshold and
//                  compacts the survivors in traversal order, under an item budget;
//              (b) fetches their postings and scores every document SPECULATIVELY:
//                  16 lanes per document, 16-byte loads of the record (components |
//                  f16 values); per lane group four documents of <= 128 elements
//                  (or two longer ones) in flight in rotating register slots,
//                  documents pulled from a shared counter;
//              (c) REPLAYS the reference's sequential decisions on one wavefront over
//                  the (block dot, doc score) table in LDS: same skip tests with the
//                  live threshold, same heap pushes in the same order. Once the heap
//                  is full only items scoring above the round's starting threshold
//                  can matter; phase (b) collects them and the replay walks just
//                  those, in order, in registers.
//            The result is the reference's exact candidate set and top-k.
//   visited  the reference's FxHashSet (src/inverted_index.rs:181-184) is replaced by
//            an exactly equivalent heap-membership test at push time (proof at
//            replay_chunk); a bitmap in HBM is kept only for the "counted" pass that
//            reports exact work counters.
//   top-k    KHeap (src/utils.rs:12-66) lives in the VGPRs of wavefront 0 as a sorted
//            array (entry e in lane e%64, register e/64): insertion is a ballot + one
//            DPP lane shift, the threshold is a readlane.
//
// No MFMA: this is gather / scatter-add / reduce work bounded by HBM latency and
// bandwidth. Floating point follows the reference (Rust never contracts): compiled
// with -ffp-contract=off and written with __fmul_rn/__fadd_rn.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_types.hpp"

namespace sgpu {

#define SGPU_DEV __device__ __forceinline__
#ifndef SGPU_WAVES_PER_EU
#define SGPU_WAVES_PER_EU 4   // 2 workgroups of 512 threads per CU (<= 128 VGPRs)
#endif

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
SGPU_DEV int32_t total_key_dev(float f) {  // Rust f32::total_cmp order
  int32_t b = __float_as_int(f);
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return b;
}

// q * (float)half(packed, hi) with ONE rounding, the product the reference computes after widening
// the stored f16: v_fma_mix_f32 widens the selected half on the fly and adds -0.0, which leaves the
// rounded product untouched, signed zeros and denormals included (tools/ubench/fma_mix_check.hip:
// bit-identical to v_cvt_f32_f16 + v_mul_f32). One instruction instead of convert (+ shift) + multiply.
SGPU_DEV float mul_f32_f16(float q, uint32_t packed, int hi) {
  float r;
  if (hi)
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(q), "v"(packed), "s"(0x80000000u));
  else
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(q), "v"(packed), "s"(0x80000000u));
  return r;
}

SGPU_DEV float half_bits_to_float(uint32_t h) {   // exact binary16 -> binary32 (v_cvt_f32_f16)
  const unsigned short b = (unsigned short)h;
  _Float16 x;
  __builtin_memcpy(&x, &b, 2);
  return (float)x;
}

SGPU_DEV uint32_t lane_id() { return __lane_id(); }

// value of `v` in lane `l`, l wave-uniform (v_readlane_b32; no LDS round trip)
SGPU_DEV uint32_t readlane_u(uint32_t v, uint32_t l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)l));
}
SGPU_DEV float readlane_f(float v, uint32_t l) { return __uint_as_float(readlane_u(__float_as_uint(v), l)); }
// lane i receives lane i-1's value, lane 0 keeps its own (DPP wave_shr:1)
SGPU_DEV uint32_t shift_up1_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}
SGPU_DEV float shift_up1_f(float v) { return __uint_as_float(shift_up1_u(__float_as_uint(v))); }

// inclusive scan of one u32 per thread over the whole workgroup.
// `part` points to NT/64 + 1 LDS words. Returns inclusive prefix; *total = sum.
// Inclusive prefix sum over the 64 lanes: four row shifts inside each row of 16, then the row
// totals carried with row_bcast:15 / row_bcast:31 (lanes without a source add 0).
#define SGPU_DPP0(x, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, false))
SGPU_DEV uint32_t wave_inclusive_scan(uint32_t x) {
  x += SGPU_DPP0(x, 0x111, 0xf);   // row_shr:1
  x += SGPU_DPP0(x, 0x112, 0xf);   // row_shr:2
  x += SGPU_DPP0(x, 0x114, 0xf);   // row_shr:4
  x += SGPU_DPP0(x, 0x118, 0xf);   // row_shr:8
  x += SGPU_DPP0(x, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  x += SGPU_DPP0(x, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3
  return x;
}

// One barrier: the caller guarantees that every thread is past its reads of `part` from the previous
// scan that used it (another barrier lies in between).
template <int NT>
SGPU_DEV uint32_t wg_inclusive_scan(uint32_t v, uint32_t* part, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t x = wave_inclusive_scan(v);
  if (lane == 63) part[wave] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t pw = part[w];
    if ((uint32_t)w < wave) base += pw;
    tot += pw;
  }
  *total = tot;
  return x + base;
}

// ---------------------------------------------------------------------------
// LDS view
// ---------------------------------------------------------------------------
struct Lds {
  uint32_t* q_comp;
  float* q_val;
  uint2* q_word;        // bitmap mode: [ceil(dim/32)] {32 vocabulary bits, rank of the word's first query component}
  uint8_t* q_idx;       // dense mode (same LDS region): [dim] 1 + rank of the component in the query, 0 = absent
  uint32_t* q_bits32;   // split mode (same region): [ceil(dim/32)] bits, followed by
  uint16_t* q_rank16;   //   [ceil(dim/32)] rank of the word's first query component
  uint32_t* sel_comp;   // [QC] list (component) ids in traversal order
  uint32_t* sel_nb;     // [QC] blocks in the list
  uint32_t* sel_b0;     // [QC] first global block id
  uint32_t* sel_doff;   // [QC+1] offset of the list's dots in `dots`
  uint32_t* sel_r0;     // [QC] first global summary row
  uint32_t* sel_nr;     // [QC] number of summary rows
  uint64_t* rt_start;   // [QC*QN] global entry index of matched row (l, j) (64-bit: > 4 G summary entries)
  uint16_t* rt_mid;     // [QC*QN] split of the row between the list's two block-id halves
  uint32_t* rt_pre;     // [2*QC*(QN+1)] per (list, half): prefix of the rows' 64-entry chunk counts
  float* dots;
  uint16_t* order;
  uint8_t* lookup;      // start of the query lookup table (whichever layout)
  uint8_t* uni;         // union region
  uint32_t* part;       // scan partials [NT/64 + 1]
  uint32_t* st;         // state words
};
enum { ST_Q = 0, ST_NLISTS = 1, ST_THR = 2, ST_HLEN = 3, ST_TMP0 = 4, ST_TMP1 = 5, ST_TMP2 = 6, ST_NCAND = 7,
       ST_CAND = 8 /* 64 candidate item indices */, ST_CAND_SORTED = 72 /* 64 */, ST_NSHORT = 136, ST_NLONG = 137,
       ST_PULL_L = 138, ST_NBLK = 139, ST_ENTRIES = 140, ST_ROWS = 141 };
static_assert(ST_ROWS < kStateWords, "state words");
constexpr uint32_t kMaxCand = 64;

SGPU_DEV Lds carve(uint8_t* smem, const LdsLayout& L) {
  Lds l;
  l.q_comp = (uint32_t*)(smem + L.q_comp);
  l.q_val = (float*)(smem + L.q_val) + 1;   // q_val[-1] is the 0.0 every non-matching component resolves to
  l.q_word = (uint2*)(smem + L.q_bits);
  l.q_idx = smem + L.q_bits;
  l.q_bits32 = (uint32_t*)(smem + L.q_bits);
  l.q_rank16 = (uint16_t*)(smem + L.q_rank);
  l.sel_comp = (uint32_t*)(smem + L.sel);
  l.sel_nb = l.sel_comp + L.qc;
  l.sel_b0 = l.sel_nb + L.qc;
  l.sel_doff = l.sel_b0 + L.qc;          // qc + 1
  l.sel_r0 = l.sel_doff + L.qc + 1;
  l.sel_nr = l.sel_r0 + L.qc;
  l.rt_start = (uint64_t*)(smem + L.rt_start);
  l.rt_mid = (uint16_t*)(smem + L.rt_mid);
  l.rt_pre = (uint32_t*)(smem + L.rt_pre);
  l.dots = (float*)(smem + L.dots);
  l.order = (uint16_t*)(smem + L.order);
  l.lookup = smem + L.q_bits;
  l.uni = smem + L.uni;
  l.part = (uint32_t*)(smem + L.part);
  l.st = (uint32_t*)(smem + L.st);
  return l;
}

// ---------------------------------------------------------------------------
// stage 0: query into LDS, choose the lists
// ---------------------------------------------------------------------------
template <int NT>
SGPU_DEV void load_query(const Lds& s, const BatchView& qb, uint32_t q, uint32_t* nnz_out) {
  const uint32_t o0 = qb.q_off[q], o1 = qb.q_off[q + 1];
  const uint32_t nnz = o1 - o0;
  for (uint32_t j = threadIdx.x; j < nnz; j += NT) {
    s.q_comp[j] = qb.q_comp[o0 + j];
    s.q_val[j] = qb.q_val[o0 + j];
  }
  *nnz_out = nnz;
}

// Query lookup table layouts (device_types.hpp: LK_*).
// Fills the query lookup table used by the scoring loop (cleared during stage 1, lookup_clear).
//   dense : one byte per vocabulary id: 1 + rank of the id in the query, 0 = absent
//   bitmap: {32 vocabulary bits, rank of the word's first query component} per 32 ids
template <int NT, int LK>
SGPU_DEV void build_lookup(const Lds& s, uint32_t dim, uint32_t nnz) {
  for (uint32_t j = threadIdx.x; j < nnz; j += NT) {
    const uint32_t c = s.q_comp[j];
    if (LK == LK_DENSE) {
      s.q_idx[c] = (uint8_t)(j + 1);
    } else if (LK == LK_PACKED) {
      atomicOr(&s.q_word[c >> 5].x, 1u << (c & 31));
      if (j == 0 || (s.q_comp[j - 1] >> 5) != (c >> 5)) s.q_word[c >> 5].y = j;
    } else {   // split: 32 bits + a 16-bit rank per 32 ids (large vocabularies: 6 B instead of 8 B)
      atomicOr(&s.q_bits32[c >> 5], 1u << (c & 31));
      if (j == 0 || (s.q_comp[j - 1] >> 5) != (c >> 5)) s.q_rank16[c >> 5] = (uint16_t)j;
    }
  }
  __syncthreads();
}

// k_largest_by(query_cut, total_cmp) in descending order; ties: ascending component.
template <int NT>
SGPU_DEV void select_lists(const Lds& s, const DevView& ix, uint32_t nnz, uint32_t query_cut) {
  const uint32_t nl = nnz < query_cut ? nnz : query_cut;
  for (uint32_t j = threadIdx.x; j < nnz; j += NT) {
    const int32_t kj = total_key_dev(s.q_val[j]);
    uint32_t rank = 0;
    for (uint32_t i = 0; i < nnz; ++i) {
      const int32_t ki = total_key_dev(s.q_val[i]);
      rank += (ki > kj) || (ki == kj && i < j);   // i < j  <=>  smaller component id
    }
    if (rank < nl) {
      const uint32_t c = s.q_comp[j];
      const uint32_t b0 = ix.list_block_start[c], b1 = ix.list_block_start[c + 1];
      const uint32_t r0 = ix.list_row_start[c], r1 = ix.list_row_start[c + 1];
      s.sel_comp[rank] = c;
      s.sel_b0[rank] = b0;
      s.sel_nb[rank] = b1 - b0;
      s.sel_r0[rank] = r0;
      s.sel_nr[rank] = r1 - r0;
    }
  }
  if (threadIdx.x == 0) s.st[ST_NLISTS] = nl;
}

// ---------------------------------------------------------------------------
// stage 1: summary dots
// ---------------------------------------------------------------------------
template <typename CT, int NT>
SGPU_DEV void build_row_table(const Lds& s, const DevView& ix, uint32_t nnz, uint32_t nl, uint32_t qn) {
  const CT* row_comp = (const CT*)ix.row_comp;
  // one (list, query component) pair per thread: binary search the list's sorted row components
  for (uint32_t t = threadIdx.x; t < nl * nnz; t += NT) {
    const uint32_t l = t / nnz, j = t - l * nnz;
    const uint32_t target = s.q_comp[j];
    uint32_t lo = s.sel_r0[l], hi = lo + s.sel_nr[l];
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      const uint32_t c = (uint32_t)row_comp[mid];
      if (c < target) lo = mid + 1; else hi = mid;
    }
    uint64_t start = 0;
    uint32_t len = 0, mid = 0;
    if (lo < s.sel_r0[l] + s.sel_nr[l] && (uint32_t)row_comp[lo] == target) {
      start = ix.row_ptr[lo];
      len = (uint32_t)(ix.row_ptr[lo + 1] - start);
      mid = ix.row_mid[lo];
    }
    if (len) {   // work counters (entries, matched rows)
      atomicAdd(&s.st[ST_ENTRIES], len);
      atomicAdd(&s.st[ST_ROWS], 1u);
    }
    s.rt_start[l * qn + j] = (start << 16) | (uint64_t)len;   // a row has at most one entry per block: len <= 65535
    s.rt_mid[l * qn + j] = (uint16_t)mid;
    // 64-entry chunks of the two halves [0, mid) and [mid, len); turned into prefixes below
    s.rt_pre[(2 * l) * (qn + 1) + j + 1] = (mid + 63u) >> 6;
    s.rt_pre[(2 * l + 1) * (qn + 1) + j + 1] = (len - mid + 63u) >> 6;
  }
  __syncthreads();
  // per-stream prefix over the query components (wave w handles streams w, w+NW, ...)
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t l = wave; l < 2 * nl; l += NT / 64) {
    uint32_t* pre = s.rt_pre + l * (qn + 1);
    uint32_t carry = 0;
    for (uint32_t j0 = 0; j0 < nnz; j0 += 64) {
      const uint32_t j = j0 + lane;
      const uint32_t x = wave_inclusive_scan(j < nnz ? pre[j + 1] : 0);
      if (j < nnz) pre[j + 1] = x + carry;
      carry += readlane_u(x, 63);
    }
    if (lane == 0) pre[0] = 0;
  }
  // dots offsets (serial, tiny)
  if (threadIdx.x == 0) {
    uint32_t o = 0;
    for (uint32_t l = 0; l < nl; ++l) {
      s.sel_doff[l] = o;
      o += s.sel_nb[l];
    }
    s.sel_doff[nl] = o;
  }
  __syncthreads();
}

// QuantizedSummary::distances for lists [0, nl): one wavefront per list streams the list's matched
// rows in ascending query-component order, 64 entries (one chunk of one row) per step:
//     acc[block] += (code * quant + min) * query_weight          (src/quantized_summary.rs:96-108)
// `code * quant + min` is computed once at upload with the reference's roundings (sum_deq); the
// product and the addition are separately rounded (no FMA). Block ids are distinct within a row,
// so the lanes of a step never collide, and a wavefront's LDS operations execute in issue order,
// so every accumulator sees its additions in ascending component order - the reference's order,
// bit for bit. The (block id, value) loads of the next kDepth steps are in flight while a step
// accumulates, so the HBM latency is paid once per list rather than once per row.
// Wavefronts without a list clear the query lookup table meanwhile (lookup_clear).
template <int LK>
SGPU_DEV void lookup_clear(const Lds& s, uint32_t dim, uint32_t tid, uint32_t nthreads) {
  uint32_t* z = (uint32_t*)s.lookup;
  const uint32_t words = (dim + 31) / 32;
  const uint32_t nz = LK == LK_DENSE ? (dim + 1 + 3) / 4 : (LK == LK_PACKED ? 2 * words : words);   // split: bits only
  for (uint32_t i = tid; i < nz; i += nthreads) z[i] = 0;
}

#ifndef SGPU_STREAM_DEPTH
#define SGPU_STREAM_DEPTH 16
#endif
SGPU_DEV uint32_t uniform_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

SGPU_DEV void stream_list_dots(const Lds& s, const DevView& ix, uint32_t nnz_, uint32_t qn, uint32_t sl_) {
  constexpr int kDepth = SGPU_STREAM_DEPTH;
  const uint32_t lane = lane_id();
  // wave-uniform values, pinned to scalar registers so that the step bookkeeping runs on the SALU
  // stream sl = (list l, half h): the entries of l's rows whose block id lies in l's lower (h = 0) or
  // upper (h = 1) half. The two streams of a list touch disjoint accumulators.
  const uint32_t nnz = uniform_u(nnz_), sl = uniform_u(sl_), l = sl >> 1, h = sl & 1u;
  float* acc = s.dots + s.sel_doff[l];
  const uint64_t* rts = s.rt_start + l * qn;
  const uint16_t* mids = s.rt_mid + l * qn;
  const uint32_t* cpre = s.rt_pre + sl * (qn + 1);
  for (uint32_t jb = 0; jb < nnz; jb += 64) {   // rows in blocks of 64: lane i keeps row jb + i
    const uint32_t nj = nnz - jb < 64u ? nnz - jb : 64u;
    const bool has = lane < nj;
    uint64_t r = 0ull;
    if (has) {   // this half of the row: (first entry) << 16 | entries
      const uint64_t full = rts[jb + lane];
      const uint32_t len = (uint32_t)full & 0xffffu, mid = mids[jb + lane];
      r = h ? ((((full >> 16) + mid) << 16) | (uint64_t)(len - mid)) : (((full >> 16) << 16) | (uint64_t)mid);
    }
    const uint32_t r_lo = (uint32_t)r, r_hi = (uint32_t)(r >> 32);
    const uint32_t c0 = cpre[jb + (has ? lane : nj)];   // first chunk of the row; lanes >= nj hold the block's end
    const float w = has ? s.q_val[jb + lane] : 0.0f;
    const uint32_t t_begin = readlane_u(c0, 0);
    const uint32_t t_end = uniform_u(cpre[jb + nj]);
    for (uint32_t tb = t_begin; tb < t_end; tb += 64) {   // chunks in blocks of 64: lane i describes chunk tb + i
      const uint32_t cnt = t_end - tb < 64u ? t_end - tb : 64u;
      uint32_t d_lo, d_hn;
      float d_q;
      {
        const uint32_t x = tb + lane;
        uint32_t lo = 0, hi = 64;   // the row owning chunk x: the last j with c0[j] <= x
#pragma unroll
        for (int it = 0; it < 6; ++it) {
          const uint32_t mid = (lo + hi) >> 1;
          if ((uint32_t)__shfl((int)c0, (int)mid) <= x) lo = mid; else hi = mid;
        }
        const uint32_t rl = (uint32_t)__shfl((int)r_lo, (int)lo), rh = (uint32_t)__shfl((int)r_hi, (int)lo);
        const uint32_t u = x - (uint32_t)__shfl((int)c0, (int)lo);
        const uint32_t len = rl & 0xffffu;
        const uint64_t g = ((((uint64_t)rh << 32) | rl) >> 16) + (uint64_t)u * 64u;
        uint32_t n = len - u * 64u;
        n = n < 64u ? n : 64u;
        if (lane >= cnt) n = 0;
        d_lo = (uint32_t)g;
        d_hn = (uint32_t)(g >> 32) | (n << 16);   // entry offsets are below 2^48
        d_q = __shfl(w, (int)lo);
      }
      uint32_t bid[kDepth];
      float prod[kDepth];
      bool ok[kDepth];
      auto fetch = [&](uint32_t i, int d) {
        ok[d] = false;
        bid[d] = 0;
        prod[d] = 0.0f;
        if (i < cnt) {
          const uint32_t hn = readlane_u(d_hn, i);
          const uint64_t g = ((((uint64_t)(hn & 0xffffu)) << 32) | readlane_u(d_lo, i)) + lane;
          ok[d] = lane < (hn >> 16);
          if (ok[d]) {
            bid[d] = (uint32_t)ix.sum_bid[g];
            prod[d] = ix.sum_deq[g];   // multiplied by the row's query weight when consumed
          }
        }
      };
#pragma unroll
      for (int d = 0; d < kDepth; ++d) fetch((uint32_t)d, d);
      for (uint32_t i = 0; i < cnt; i += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
          // (a return-less LDS float add, ds_add_f32, rounds identically - tools/ubench/lds_fadd_check.hip -
          // but costs ~350 cycles per wavefront instruction against ~100 for this read-add-write;
          // sending the idle lanes to a spare accumulator instead of branching is slower still)
          if (ok[d]) acc[bid[d]] = __fadd_rn(acc[bid[d]], __fmul_rn(prod[d], readlane_f(d_q, i + (uint32_t)d)));
          fetch(i + (uint32_t)d + kDepth, d);
        }
      }
    }
  }
}

template <int NT, int LK>
SGPU_DEV void summary_dots(const Lds& s, const DevView& ix, uint32_t nnz, uint32_t nl, uint32_t qn, uint32_t dim,
                           bool want_lookup) {
  const uint32_t wave = threadIdx.x >> 6;
  constexpr uint32_t NW = NT / 64;
  const uint32_t total_blocks = s.sel_doff[nl];
  for (uint32_t i = threadIdx.x; i < total_blocks; i += NT) s.dots[i] = 0.0f;
  __syncthreads();
  const uint32_t ns = 2 * nl;                // streams: two block-id halves per list
  const uint32_t busy = ns < NW ? ns : NW;   // wavefronts that own a stream
  if (wave < busy) {
    for (uint32_t sl = wave; sl < ns; sl += NW) stream_list_dots(s, ix, nnz, qn, sl);
  } else if (want_lookup) {
    lookup_clear<LK>(s, dim, threadIdx.x - busy * 64u, NT - busy * 64u);
  }
  __syncthreads();
  if (want_lookup && busy == NW) {
    lookup_clear<LK>(s, dim, threadIdx.x, NT);
    __syncthreads();
  }
}

// descending sort of list 0's blocks by (total_cmp(dot) desc, block id asc) -> s.order
template <int NT>
SGPU_DEV void sort_first_list(const Lds& s, uint32_t nb) {
  uint64_t* keys = (uint64_t*)s.uni;
  uint32_t n2 = 1;
  while (n2 < nb) n2 <<= 1;
  for (uint32_t i = threadIdx.x; i < n2; i += NT) {
    uint64_t k = ~0ull;
    if (i < nb) {
      // ascending u64 order == (dot descending, block ascending)
      const uint32_t uk = (uint32_t)total_key_dev(s.dots[i]) ^ 0x80000000u;   // monotone unsigned
      k = ((uint64_t)(~uk) << 32) | i;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (uint32_t size = 2; size <= n2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = threadIdx.x; t < n2 / 2; t += NT) {
        const uint32_t i = 2 * t - (t & (stride - 1));
        const uint32_t j = i + stride;
        const bool up = (i & size) == 0;
        const uint64_t a = keys[i], b = keys[j];
        if ((a > b) == up) {
          keys[i] = b;
          keys[j] = a;
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = threadIdx.x; i < nb; i += NT) s.order[i] = (uint16_t)(keys[i] & 0xffffu);
  __syncthreads();
}

// ---------------------------------------------------------------------------
// top-k heap in the registers of wavefront 0 (KHeap, src/utils.rs:12-66)
// ---------------------------------------------------------------------------
template <int KR>
struct RegHeap {
  float sc[KR];
  uint32_t doc[KR];
  uint32_t len;   // wave-uniform
  float thr;      // wave-uniform: score of entry k-1 once len == k
  SGPU_DEV void reset() {
    thr = 0.0f;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      sc[r] = -__builtin_inff();
      doc[r] = 0xffffffffu;
    }
    len = 0;
  }
  // x (wave-uniform) is inserted at its rank; entries at index >= k are dropped.
  SGPU_DEV void insert(float xs, uint32_t xd, uint32_t k) {
    const uint32_t lane = lane_id();
    uint32_t pos = 0;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const bool better = (sc[r] > xs) || (sc[r] == xs && doc[r] < xd);
      pos += (uint32_t)__popcll(__ballot(better));
    }
#pragma unroll
    for (int r = KR - 1; r >= 0; --r) {
      float ps = shift_up1_f(sc[r]);
      uint32_t pd = shift_up1_u(doc[r]);
      if (r > 0) {
        const float cs = readlane_f(sc[r - 1], 63);
        const uint32_t cd = readlane_u(doc[r - 1], 63);
        if (lane == 0) {
          ps = cs;
          pd = cd;
        }
      }
      const uint32_t e = (uint32_t)r * 64u + lane;
      if (e == pos) {
        sc[r] = xs;
        doc[r] = xd;
      } else if (e > pos) {
        sc[r] = ps;
        doc[r] = pd;
      }
      if (e >= k) {
        sc[r] = -__builtin_inff();
        doc[r] = 0xffffffffu;
      }
    }
    if (len < k) ++len;
    if (len == k) thr = kth(k);
  }
  SGPU_DEV float kth(uint32_t k) const {   // score of entry k-1 (valid when len == k)
    float v = 0.0f;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const float x = readlane_f(sc[r], (k - 1) & 63);
      if ((uint32_t)r == ((k - 1) >> 6)) v = x;
    }
    return v;
  }
};

// ---------------------------------------------------------------------------
// stage 2 pieces
// ---------------------------------------------------------------------------
struct ChunkBufs {   // carved from the union region
  uint16_t* live_pos;   // [NT] positions (in traversal order) of live blocks
  uint32_t* cb_incl;    // [NT] inclusive item prefix
  uint32_t* cb_p0;      // [NT] first posting of the block
  uint16_t* cb_blk;     // [NT] block id (list-local)
  uint64_t* it_ref;     // [ITEMS] packed doc record ref; high 32 bits reused for the score
  uint32_t* it_doc;     // [ITEMS] doc id | (already visited) << 31
  uint16_t* it_blk;     // [ITEMS] list-local block id (its summary dot is dots[it_blk])
  uint16_t* it_ord;     // [ITEMS] item indices by length class: <= 128 elements from the front, longer from the back
};

template <int NT>
SGPU_DEV ChunkBufs carve_chunk(uint8_t* uni, uint32_t items_max) {
  ChunkBufs c;
  uint8_t* p = uni;
  c.it_ref = (uint64_t*)p;  p += (size_t)items_max * 8;
  c.it_doc = (uint32_t*)p;  p += (size_t)items_max * 4;
  c.cb_incl = (uint32_t*)p; p += NT * 4;
  c.cb_p0 = (uint32_t*)p;   p += NT * 4;
  c.it_blk = (uint16_t*)p;  p += (size_t)items_max * 2;
  c.it_ord = (uint16_t*)p;  p += (size_t)items_max * 2;
  c.live_pos = (uint16_t*)p; p += NT * 2;
  c.cb_blk = (uint16_t*)p;
  return c;
}

SGPU_DEV bool visited_test(const uint32_t* bitmap, uint32_t doc) {
  // written by other waves of this workgroup through L2 atomics: read past the (per-CU,
  // not atomically updated) L1 with an agent-scope load
  const uint32_t w = __hip_atomic_load(bitmap + (doc >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (w >> (doc & 31)) & 1u;
}
SGPU_DEV void visited_mark(uint32_t* bitmap, uint32_t doc) {
  __hip_atomic_fetch_or(bitmap + (doc >> 5), 1u << (doc & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// QueryEvaluator::compute_distance for one document by one 16-lane group.
// Canonical order: lane j accumulates elements [128 s + 8 j, +8) in
// increasing index, then t[j] += t[j ^ d] for d = 8, 4, 2, 1.
// Non-matching components resolve to the weight 0.0 (slot qn of q_val) and are
// added as +-0.0, which leaves an accumulator that started at +0.0 bit-identical
// to skipping them (x + (+-0) == x, and the accumulator can never be -0.0).
template <typename CT>
struct DocChunk {   // 8 consecutive elements of one document, as loaded
  uint4 c0, c1, v;
};

template <typename CT>
SGPU_DEV void load_chunk(DocChunk<CT>& d, const uint8_t* rec, const uint8_t* vals, uint32_t e0) {
  if (sizeof(CT) == 2) {
    d.c0 = *(const uint4*)(rec + (size_t)e0 * 2);
  } else {
    d.c0 = *(const uint4*)(rec + (size_t)e0 * 4);
    d.c1 = *(const uint4*)(rec + (size_t)e0 * 4 + 16);
  }
  d.v = *(const uint4*)(vals + (size_t)e0 * 2);
}

template <typename CT, int LK>
SGPU_DEV float accumulate_chunk(const Lds& s, const DocChunk<CT>& d, uint32_t e0, uint32_t len, float acc) {
  uint32_t c[8];
  if (sizeof(CT) == 2) {
    c[0] = d.c0.x & 0xffffu; c[1] = d.c0.x >> 16; c[2] = d.c0.y & 0xffffu; c[3] = d.c0.y >> 16;
    c[4] = d.c0.z & 0xffffu; c[5] = d.c0.z >> 16; c[6] = d.c0.w & 0xffffu; c[7] = d.c0.w >> 16;
  } else {
    c[0] = d.c0.x; c[1] = d.c0.y; c[2] = d.c0.z; c[3] = d.c0.w;
    c[4] = d.c1.x; c[5] = d.c1.y; c[6] = d.c1.z; c[7] = d.c1.w;
  }
  const uint32_t v[4] = {d.v.x, d.v.y, d.v.z, d.v.w};
  float qv[8];
  if (LK == LK_DENSE) {
    // one byte per vocabulary id: 1 + rank in the query, 0 = absent (-> q_val[-1] == 0.0).
    // Padding components carry the sentinel id `dim`, whose byte is always 0: no length test.
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = s.q_idx[c[i]];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = s.q_val[(int)r[i] - 1];
  } else {
    uint2 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (LK == LK_PACKED) w[i] = s.q_word[c[i] >> 5];
      else w[i] = make_uint2(s.q_bits32[c[i] >> 5], (uint32_t)s.q_rank16[c[i] >> 5]);
    }
    int r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t bit = c[i] & 31u;
      const bool hit = ((w[i].x >> bit) & 1u) && (e0 + (uint32_t)i < len);
      r[i] = hit ? (int)(w[i].y + (uint32_t)__popc(w[i].x & ((1u << bit) - 1u))) : -1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = s.q_val[r[i]];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = __fadd_rn(acc, mul_f32_f16(qv[i], v[i >> 1], i & 1));
  return acc;
}

// Sum over the 16 lanes of a group, valid in lane 0 of the group. DPP row rotations (no LDS
// round trip): lane i adds lane (i + d) % 16 for d = 8, 4, 2, 1. For lane 0 every partner is the
// same as in the butterfly t[j] += t[j ^ d] of the canonical order, so the value is bit-identical.
SGPU_DEV float dpp_row_ror(float v, int ctrl8421) {
  const int x = (int)__float_as_uint(v);
  int r;
  switch (ctrl8421) {
    case 8: r = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x122, 0xf, 0xf, false); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x121, 0xf, 0xf, false); break;
  }
  return __uint_as_float((uint32_t)r);
}
SGPU_DEV float reduce16(float acc) {
  acc = __fadd_rn(acc, dpp_row_ror(acc, 8));
  acc = __fadd_rn(acc, dpp_row_ror(acc, 4));
  acc = __fadd_rn(acc, dpp_row_ror(acc, 2));
  acc = __fadd_rn(acc, dpp_row_ror(acc, 1));
  return acc;
}
// lane 0 of each 16-lane row, broadcast to the row (DPP row_share:0)
SGPU_DEV uint32_t row_bcast0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150, 0xf, 0xf, false);
}

// Work the reference algorithm performs for this query (lane-local partial counts, wave 0).
struct WorkCount {
  uint32_t blocks, posts, docs, len;
};

// Is document `d` (wave-uniform) currently in the heap?
template <int KR>
SGPU_DEV bool heap_contains(const RegHeap<KR>& heap, uint32_t d) {
  bool m = false;
#pragma unroll
  for (int r = 0; r < KR; ++r) m |= __ballot(heap.doc[r] == d) != 0ull;
  return m;
}

// Sequential replay of the reference's decisions over the chunk's items (wavefront 0 only).
//
// Visited set (reference: FxHashSet keyed by the document's forward-index offset,
// src/inverted_index.rs:181-184, src/posting_list.rs:200,209). A document can recur only in a
// LATER list of the same query (a list holds a document once). USE_BITMAP = true keeps a
// per-workgroup bitmap in HBM (exact work counters; ~0.8 MB of extra traffic per query).
// USE_BITMAP = false needs no visited set at all and is exactly equivalent: a recurring document
// d has the same score s as the first time, and
//   * while the heap is not full every visited document is in the heap, so membership == visited;
//   * once full, a push is attempted only if s > kth. The first visit pushed d (then-kth <= kth < s)
//     and d cannot have been evicted while s > kth, so d is still in the heap: membership == visited
//     for exactly the items whose push would change anything. Recurring documents with s <= kth
//     are rejected by the reference's visited test and by the push rule alike.
// The price is re-scoring the recurring documents (1-2% of the postings).
template <int KR, bool USE_BITMAP>
SGPU_DEV void replay_chunk(RegHeap<KR>& heap, const ChunkBufs& cb, uint32_t n_items, uint32_t k,
                           float heap_factor, uint32_t* bitmap, uint32_t& decided_blk,
                           bool block_starts_at_0, WorkCount& wc, uint32_t& live_items, bool dups,
                           const float* dots) {
  // dots: summary dots of the current list (LDS); nullptr = no block test (kNN refinement)
  // dups: the round may hold the same document more than once (kNN refinement: two results can
  // share a neighbour; a posting list never repeats a document). The reference inserts into its
  // visited set item by item, so a later copy must see the earlier one.
  const uint32_t lane = lane_id();
  const uint32_t* it_words = (const uint32_t*)cb.it_ref;   // [2i] = low ref word (len), [2i+1] = score
  uint32_t i = 0;
  while (i < n_items) {
    const uint32_t idx = i + lane;
    const bool valid = idx < n_items;
    if (!USE_BITMAP && heap.len == k) {
      // Fast skip: the k-th best score only rises, so a window in which no score exceeds it
      // cannot change the heap; without a visited set to maintain there is nothing else to do.
      // (Work counters are exact only in the counted pass, which never takes this shortcut.)
      const float sc0 = valid ? __uint_as_float(it_words[2 * idx + 1]) : 0.0f;
      const float bd0 = valid ? (dots ? dots[cb.it_blk[idx]] : __builtin_inff()) : 0.0f;
      if (__ballot(valid && sc0 > heap.thr) == 0ull) {
        live_items += (uint32_t)__popcll(__ballot(valid && !(bd0 < __fmul_rn(heap_factor, heap.thr))));
        i += 64;
        continue;
      }
    }
    float sc = 0.0f;
    uint32_t doc = 0, blk = 0, len = 0;
    float bdot = 0.0f;
    bool vis = true, first = false;
    if (valid) {
      sc = __uint_as_float(it_words[2 * idx + 1]);
      len = it_words[2 * idx] & 0xffffu;
      const uint32_t d = cb.it_doc[idx];
      doc = d & 0x7fffffffu;
      vis = (d >> 31) != 0;
      blk = cb.it_blk[idx];
      bdot = dots ? dots[blk] : __builtin_inff();
      first = idx == 0 ? block_starts_at_0 : (cb.it_blk[idx - 1] != blk);
    }
    if (dups && USE_BITMAP) {   // marks made earlier in this round (by this wavefront) must be seen
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
      if (valid && !vis) vis = visited_test(bitmap, doc);
    }
    if (heap.len < k) {
      // heap not full: every block that starts now is evaluated, every new doc is pushed
      if (!USE_BITMAP) {   // visited == already in the heap
#pragma unroll
        for (int r = 0; r < KR; ++r) {
          const uint32_t lim = heap.len > (uint32_t)r * 64u ? heap.len - (uint32_t)r * 64u : 0u;
          for (uint32_t l = 0; l < (lim < 64u ? lim : 64u); ++l) vis |= doc == readlane_u(heap.doc[r], l);
        }
      }
      if (dups) {   // a later copy of a document inside this window is "visited" by the earlier one
        const uint64_t nv0 = __ballot(valid && !vis);
        for (uint32_t l = 0; l < 63; ++l)
          if (((nv0 >> l) & 1ull) && lane > l && doc == readlane_u(doc, l)) vis = true;
      }
      const uint64_t nvm = __ballot(valid && !vis);
      const uint32_t need = k - heap.len;
      const uint32_t before = (uint32_t)__popcll(nvm & ((1ull << lane) - 1ull));
      const bool take = valid && !vis && before < need;
      const uint64_t tm = __ballot(take);
      uint32_t last;   // window position of the last item consumed in this step
      if ((uint32_t)__popcll(nvm) >= need) last = 63u - (uint32_t)__clzll(tm);
      else last = (n_items - i < 64u ? n_items - i : 64u) - 1u;
      if (take) {
        if (USE_BITMAP) visited_mark(bitmap, doc);
        wc.docs += 1;
        wc.len += len;
      }
      if (valid && lane <= last) {
        wc.posts += 1;
        wc.blocks += first;
      }
      live_items += last + 1;
      uint64_t m = tm;
      while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        heap.insert(readlane_f(sc, (uint32_t)l), readlane_u(doc, (uint32_t)l), k);
      }
      decided_blk = readlane_u(blk, last);
      i += last + 1;
      continue;
    }
    // Heap full: the window's 64 items stay in registers; every heap change re-evaluates the
    // remaining lanes against the new threshold (the reference's per-item test, 64 at a time).
    uint32_t start = 0, advance = 64;
    for (;;) {
      const float thr = heap.thr;
      const float cut = __fmul_rn(heap_factor, thr);
      const bool live = valid && lane >= start && ((blk == decided_blk) || !(bdot < cut));
      const bool changing = live && !vis && (sc > thr);
      uint64_t cm = __ballot(changing);
      if (!USE_BITMAP) {   // drop recurring documents: they are in the heap already
        while (cm) {
          const uint32_t c0 = (uint32_t)(__ffsll((long long)cm) - 1);
          if (!heap_contains<KR>(heap, readlane_u(doc, c0))) break;
          cm &= cm - 1;
        }
      }
      const uint32_t f = cm ? (uint32_t)(__ffsll((long long)cm) - 1) : 63u;
      live_items += (uint32_t)__popcll(__ballot(live && lane <= f));
      if (USE_BITMAP && live && lane <= f) {   // exact work counters: the counted pass only
        wc.posts += 1;
        wc.blocks += first;
        if (!vis) {
          visited_mark(bitmap, doc);
          wc.docs += 1;
          wc.len += len;
        }
      }
      if (cm == 0) break;
      heap.insert(readlane_f(sc, f), readlane_u(doc, f), k);
      decided_blk = readlane_u(blk, f);
      if (dups) {   // later copies of a document must see this step's visited marks: reload the window
        advance = f + 1;
        break;
      }
      start = f + 1;
      if (start >= 64) break;
    }
    i += advance;
  }
}

// Replay when the heap was already full at the start of the round and no visited bitmap is kept:
// only items whose score exceeds the round's starting threshold can change the heap (the
// threshold never decreases). Phase B collected their indices (<= kMaxCand, unordered); they are
// ranked by item index and walked in order with the same liveness / membership / push rules as
// replay_chunk, everything in registers.
template <int KR>
SGPU_DEV void replay_candidates(RegHeap<KR>& heap, const ChunkBufs& cb, const uint32_t* st, uint32_t nc,
                                uint32_t k, float heap_factor, uint32_t& decided_blk, const float* dots) {
  const uint32_t lane = lane_id();
  const uint32_t* it_words = (const uint32_t*)cb.it_ref;
  uint32_t idx = lane < nc ? st[ST_CAND + lane] : 0xffffffffu;
  uint32_t rank = 0;
  for (uint32_t l = 0; l < nc; ++l) rank += readlane_u(idx, l) < idx;
  uint32_t* sorted = (uint32_t*)st + ST_CAND_SORTED;
  if (lane < nc) sorted[rank] = idx;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  idx = lane < nc ? sorted[lane] : 0u;
  float sc = 0.0f, bdot = 0.0f;
  uint32_t doc = 0, blk = 0;
  if (lane < nc) {
    sc = __uint_as_float(it_words[2 * idx + 1]);
    doc = cb.it_doc[idx] & 0x7fffffffu;
    blk = cb.it_blk[idx];
    bdot = dots ? dots[blk] : __builtin_inff();
  }
  // lane c holds the c-th candidate in item order; each step jumps to the next one that still beats
  // the (rising) threshold, so the loop runs once per heap change rather than once per candidate
  uint64_t todo = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
  for (;;) {
    const float thr = heap.thr;
    todo &= __ballot(sc > thr);
    if (todo == 0) break;
    const uint32_t c = (uint32_t)(__ffsll((long long)todo) - 1);
    todo &= todo - 1;
    const uint32_t blk_c = readlane_u(blk, c);
    const bool live = (blk_c == decided_blk) || !(readlane_f(bdot, c) < __fmul_rn(heap_factor, thr));
    if (!live) continue;
    const uint32_t doc_c = readlane_u(doc, c);
    if (heap_contains<KR>(heap, doc_c)) continue;   // re-encountered document: already in the heap
    heap.insert(readlane_f(sc, c), doc_c, k);
    decided_blk = blk_c;
  }
}

// Phase B: speculative scoring of the round's items, 16 lanes per document. Phase A sorted the
// items into two classes by length (ChunkBufs::it_ord): documents of at most 128 elements (one
// slice of 8 elements per lane; two thirds of a SPLADE-shaped collection) occupy one slot of 8
// registers each and a lane group keeps FOUR of them in flight; longer ones take two slices and a
// group keeps two in flight (both ways 128 B per lane, what the register budget of 2 x 512 threads
// per CU allows). The slots ROTATE: as soon as a slot's document is scored the next document's
// loads are issued into it, so the memory pipe stays busy while the other slots are being scored.
// The loads are unconditional (idle lanes re-read the record's first bytes): nothing but
// straight-line code lies between a load and its use, and the compiler's counter waits name
// exactly the slot they need (vmcnt(6), not vmcnt(0)).
// Groups pull documents from a shared counter, a batch ahead. When the heap is already full (and
// no visited bitmap is kept) the items scoring above the round's starting threshold are collected
// for replay_candidates.
template <typename CT, int LK, int ND, int NS>
SGPU_DEV void score_class(const Lds& s, const DevView& ix, const ChunkBufs& cb, const uint16_t* list, int dir,
                              uint32_t n, uint32_t* pull, bool collect, float thr0, uint32_t& spec_docs) {
  if (n == 0) return;
  const uint32_t sub = threadIdx.x & 15;
  float* it_score = (float*)cb.it_ref;
  const uint32_t e0 = sub * 8u;
  uint32_t len[ND], item[ND];
  const uint8_t* rec[ND];
  DocChunk<CT> d[ND][NS];
  auto issue = [&](int u, uint32_t iu) {
    const bool has = iu < n;
    const uint32_t it = (uint32_t)list[(int)(has ? iu : 0u) * dir];
    const uint64_t ref = cb.it_ref[it];
    item[u] = has ? it : 0xffffffffu;
    len[u] = has ? (uint32_t)(ref & 0xffffu) : 0u;
    rec[u] = ix.fwd + (has ? (ref >> 16) : 0ull) * 16ull;
    const uint8_t* val = rec[u] + (size_t)((len[u] + 7u) & ~7u) * sizeof(CT);
#pragma unroll
    for (int h = 0; h < NS; ++h) {
      const uint32_t e = e0 + 128u * h;
      const bool act = e < len[u];
      const uint8_t* pc = act ? rec[u] + (size_t)e * sizeof(CT) : rec[u];
      const uint8_t* pv = act ? val + (size_t)e * 2 : rec[u];
      d[u][h].c0 = *(const uint4*)pc;
      if (sizeof(CT) == 4) d[u][h].c1 = *(const uint4*)(pc + 16);
      d[u][h].v = *(const uint4*)pv;
    }
  };
  auto consume = [&](int u) {
    float a = 0.0f;
#pragma unroll
    for (int h = 0; h < NS; ++h)
      if (e0 + 128u * h < len[u]) a = accumulate_chunk<CT, LK>(s, d[u][h], e0 + 128u * h, len[u], a);
    if (NS > 1) {
      const uint8_t* val = rec[u] + (size_t)((len[u] + 7u) & ~7u) * sizeof(CT);
      for (uint32_t e = e0 + 128u * NS; e < len[u]; e += 128u) {   // documents longer than 256
        DocChunk<CT> t;
        load_chunk<CT>(t, rec[u], val, e);
        a = accumulate_chunk<CT, LK>(s, t, e, len[u], a);
      }
    }
    a = reduce16(a);
    spec_docs += (sub == 0 && len[u] != 0);
    if (sub == 0 && item[u] != 0xffffffffu) {
      it_score[2 * item[u] + 1] = a;
      if (collect && len[u] != 0 && a > thr0) {
        const uint32_t slot = atomicAdd(&s.st[ST_NCAND], 1u);
        if (slot < kMaxCand) s.st[ST_CAND + slot] = item[u];
      }
    }
  };
  uint32_t pulled = 0;
  if (sub == 0) pulled = atomicAdd(pull, (uint32_t)ND);
  uint32_t base = row_bcast0(pulled);
  if (base >= n) return;
#pragma unroll
  for (int u = 0; u < ND; ++u) issue(u, base + (uint32_t)u);
  for (;;) {
    if (sub == 0) pulled = atomicAdd(pull, (uint32_t)ND);   // the batch that refills the slots
    consume(0);
    base = row_bcast0(pulled);
    issue(0, base);
#pragma unroll
    for (int u = 1; u < ND; ++u) {
      consume(u);
      issue(u, base + (uint32_t)u);
    }
    if (base >= n) break;   // every slot was refilled with nothing
  }
}

#ifndef SGPU_ND_SHORT
#define SGPU_ND_SHORT 4
#endif
#ifndef SGPU_ND_LONG
#define SGPU_ND_LONG 2
#endif
template <typename CT, int NT, int LK>
SGPU_DEV void score_items(const Lds& s, const DevView& ix, const ChunkBufs& cb, const KParams& p,
                          uint32_t& spec_docs) {
  const bool collect = !p.use_bitmap && s.st[ST_HLEN] == p.k;   // heap full: only scores above the
  const float thr0 = __uint_as_float(s.st[ST_THR]);            // current k-th best can matter
  const uint32_t n_short = s.st[ST_NSHORT] & 0xffffu, n_long = s.st[ST_NSHORT] >> 16;
  score_class<CT, LK, SGPU_ND_SHORT, 1>(s, ix, cb, cb.it_ord, 1, n_short, &s.st[ST_TMP2], collect, thr0, spec_docs);
  score_class<CT, LK, SGPU_ND_LONG, 2>(s, ix, cb, cb.it_ord + (p.items_max - 1), -1, n_long, &s.st[ST_PULL_L], collect, thr0,
                            spec_docs);
}

// Phase A's last step: file item i under its length class (wave-aggregated list appends). Items the
// counted pass already knows as visited are not scored at all.
SGPU_DEV void classify_item(uint32_t* st, const ChunkBufs& cb, uint32_t items_max, uint32_t i, uint32_t len,
                            bool take) {
  const bool sh = take && len <= 128u, lg = take && len > 128u;
  const uint64_t ms = __ballot(sh), ml = __ballot(lg);
  const uint32_t lane = lane_id();
  uint32_t both = 0;   // both list lengths live in one word: short | long << 16
  if (lane == 0 && (ms | ml)) both = atomicAdd(&st[ST_NSHORT], (uint32_t)__popcll(ms) | ((uint32_t)__popcll(ml) << 16));
  both = (uint32_t)__builtin_amdgcn_readfirstlane((int)both);
  const uint32_t bs = both & 0xffffu, bl = both >> 16;
  const uint64_t below = (1ull << lane) - 1ull;
  if (sh) cb.it_ord[bs + (uint32_t)__popcll(ms & below)] = (uint16_t)i;
  if (lg) cb.it_ord[items_max - 1u - (bl + (uint32_t)__popcll(ml & below))] = (uint16_t)i;
}

// Replay of one round on wavefront 0 + publication of the new heap state for the next filter.
template <int KR>
SGPU_DEV void replay_round(RegHeap<KR>& heap, const ChunkBufs& cb, const Lds& s, const KParams& p,
                           uint32_t n_items, uint32_t* bitmap, uint32_t& decided_blk, bool block_starts_at_0,
                           WorkCount& wc, const float* dots, bool dups = false) {
  uint32_t live_items = 0;
  const uint32_t nc = s.st[ST_NCAND];
  if (!p.use_bitmap && heap.len == p.k && nc <= kMaxCand) {
    // (the heap was full when the round started: phase B collected every item that can matter)
    replay_candidates<KR>(heap, cb, s.st, nc, p.k, p.heap_factor, decided_blk, dots);
    live_items = n_items;
  } else if (p.use_bitmap) {
    replay_chunk<KR, true>(heap, cb, n_items, p.k, p.heap_factor, bitmap, decided_blk, block_starts_at_0, wc,
                           live_items, dups, dots);
  } else {
    replay_chunk<KR, false>(heap, cb, n_items, p.k, p.heap_factor, bitmap, decided_blk, block_starts_at_0, wc,
                            live_items, dups, dots);
  }
  if (lane_id() == 0) {
    s.st[ST_HLEN] = heap.len;
    s.st[ST_THR] = __float_as_uint(heap.len == p.k ? heap.thr : 0.0f);
    s.st[ST_TMP1] = live_items;
    s.st[ST_NSHORT] = 0;   // the next round's class lists start empty
  }
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
template <typename CT, int NT, int KR, int LK>
__global__ __launch_bounds__(NT, SGPU_WAVES_PER_EU) void seismic_search_kernel(DevView ix, BatchView qb, KParams p,
                                                           LdsLayout L, uint32_t* queue,
                                                           uint32_t* bitmaps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const Lds s = carve(smem, L);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* bitmap = bitmaps + (size_t)blockIdx.x * ix.n_bitmap_words;
  const ChunkBufs cb = carve_chunk<NT>(s.uni, p.items_max);
  RegHeap<KR> heap;   // meaningful in wavefront 0
  WorkCount wc;       // lane-local partial counts, wavefront 0

  // one-time LDS init
  if (threadIdx.x == 0) s.q_val[-1] = 0.0f;   // the weight every non-matching component resolves to
  __syncthreads();

  for (;;) {
    if (threadIdx.x == 0) {
      const uint32_t ticket = atomicAdd(queue, 1u);
      // longest-expected-first order prepared by the host (tail balance); identity if absent
      s.st[ST_Q] = ticket < qb.nq ? (qb.q_order ? qb.q_order[ticket] : ticket) : 0xffffffffu;
    }
    __syncthreads();
    const uint32_t q = s.st[ST_Q];
    if (q == 0xffffffffu) break;

    // phase clocks (s_memtime): time since the previous TICK goes to bucket i
    uint32_t prof[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) prof[i] = 0;
    uint64_t tprev = clock64();
    (void)tprev;
#ifndef SGPU_PROF   // phase clocks only in the profiling build (make prof): they cost ~2% and 14 registers
#define TICK(i) {}
#else
#define TICK(i)                                   \
  {                                               \
    const uint64_t t_ = clock64();                \
    prof[i] += (uint32_t)((t_ - tprev) >> 4);     \
    tprev = t_;                                   \
  }
#endif
    // ---- stage 0 ----
    uint32_t nnz;
    load_query<NT>(s, qb, q, &nnz);
    heap.reset();
    wc = WorkCount{0, 0, 0, 0};
    uint32_t spec_docs = 0, st_entries = 0, st_rows = 0;
    if (threadIdx.x == 0) {
      s.st[ST_HLEN] = 0;
      s.st[ST_THR] = 0;
      s.st[ST_NSHORT] = 0;
      s.st[ST_ENTRIES] = 0;
      s.st[ST_ROWS] = 0;
    }
    __syncthreads();
    if (p.mode == MODE_DOTS) {   // sgpu_summary_distances: aim stage 1 at one given list
      if (threadIdx.x == 0) {
        const uint32_t c = p.target_list;
        s.sel_comp[0] = c;
        s.sel_b0[0] = ix.list_block_start[c];
        s.sel_nb[0] = ix.list_block_start[c + 1] - ix.list_block_start[c];
        s.sel_r0[0] = ix.list_row_start[c];
        s.sel_nr[0] = ix.list_row_start[c + 1] - ix.list_row_start[c];
        s.st[ST_NLISTS] = nnz ? 1u : 0u;
      }
    } else {
      select_lists<NT>(s, ix, nnz, p.query_cut);
    }
    __syncthreads();
    const uint32_t nl = s.st[ST_NLISTS];
    TICK(0);

    if (nl > 0) {
      // ---- stage 1 ----
      build_row_table<CT, NT>(s, ix, nnz, nl, L.qn);
      TICK(1);
      if (threadIdx.x == 0) {
        st_entries = s.st[ST_ENTRIES];
        st_rows = s.st[ST_ROWS];
      }
      summary_dots<NT, LK>(s, ix, nnz, nl, L.qn, ix.dim, p.mode != MODE_DOTS);
      TICK(2);

      if (p.mode == MODE_DOTS) {   // sgpu_summary_distances: dump the dots of list 0
        for (uint32_t i = threadIdx.x; i < s.sel_nb[0]; i += NT) qb.out_scores[i] = s.dots[i];
        if (threadIdx.x == 0) qb.out_n[q] = s.sel_nb[0];
        __syncthreads();
        continue;
      }

      // ---- stage 2 ----
      build_lookup<NT, LK>(s, ix.dim, nnz);
      uint32_t budget = p.items_init;
      for (uint32_t l = 0; l < nl; ++l) {
        const uint32_t nb = s.sel_nb[l];
        const uint32_t b0 = s.sel_b0[l];
        const float* dots = s.dots + s.sel_doff[l];
        const bool sorted = (l == 0) && p.first_sorted && nb > 1;
        if (sorted) sort_first_list<NT>(s, nb);
        TICK(3);
        uint32_t decided_blk = 0xffffffffu;   // wavefront 0 state
        uint32_t pos = 0;
        while (pos < nb) {
          // (a) filter the remaining blocks against the current threshold
          const uint32_t hlen = s.st[ST_HLEN];
          // A block may be dropped here against a threshold that is older than the one the replay
          // will use only because heap_factor * threshold never decreases - true for
          // heap_factor >= 0. A negative factor prunes nothing here; the replay decides alone.
          const bool full = hlen == p.k && p.heap_factor >= 0.0f;
          const float cut = __fmul_rn(p.heap_factor, __uint_as_float(s.st[ST_THR]));
          const uint32_t remaining = nb - pos;
          uint32_t R = (remaining + NT - 1) / NT;
          if (R > p.rblocks_max) R = p.rblocks_max;
          const uint32_t scan_end = (pos + R * NT < nb) ? pos + R * NT : nb;
          uint32_t live_mask = 0;   // bit r: block my0 + r passes the skip test (R <= 32)
          const uint32_t my0 = pos + threadIdx.x * R;
          for (uint32_t r = 0; r < R; ++r) {
            const uint32_t idx = my0 + r;
            if (idx < scan_end) {
              const uint32_t b = sorted ? (uint32_t)s.order[idx] : idx;
              live_mask |= (uint32_t)!(full && dots[b] < cut) << r;
            }
          }
          const uint32_t my_live = (uint32_t)__popc(live_mask);
          uint32_t n_live_total;
          const uint32_t live_incl = wg_inclusive_scan<NT>(my_live, s.part, &n_live_total);
          if (n_live_total == 0) {
            pos = scan_end;
            __syncthreads();   // everyone is done with s.part before the next scan writes it
            TICK(4);
            continue;
          }
          {
            uint32_t o = live_incl - my_live;
            for (uint32_t m = live_mask; m && o < NT; m &= m - 1)
              cb.live_pos[o++] = (uint16_t)(my0 + (uint32_t)__ffs((int)m) - 1u);
          }
          const uint32_t n_live = n_live_total < NT ? n_live_total : NT;
          __syncthreads();
          TICK(4);
          // (b) postings of the live blocks, budget cut
          uint32_t cnt = 0, p0 = 0, myb = 0;
          if (threadIdx.x < n_live) {
            const uint32_t idx = cb.live_pos[threadIdx.x];
            myb = sorted ? (uint32_t)s.order[idx] : idx;
            p0 = ix.block_post_start[b0 + myb];
            cnt = ix.block_post_start[b0 + myb + 1] - p0;
          }
          uint32_t items_total;
          if (threadIdx.x == 0) {
            s.st[ST_TMP0] = cnt;   // size of the first live block
            s.st[ST_NBLK] = 0;
          }
          const uint32_t incl = wg_inclusive_scan<NT>(cnt, s.part + (NT / 64 + 1), &items_total);
          const uint32_t first_cnt = s.st[ST_TMP0];
          uint32_t B = budget;
          if (first_cnt > B) B = first_cnt;             // always make progress
          const bool oversize = first_cnt > p.items_max;
          if (B > p.items_max) B = p.items_max;
          {   // blocks taken this round: the live blocks whose inclusive item count fits the budget
            const uint64_t tk = __ballot(threadIdx.x < n_live && incl <= B);
            if (lane == 0 && tk) atomicAdd(&s.st[ST_NBLK], (uint32_t)__popcll(tk));
          }
          if (threadIdx.x < n_live) {
            cb.cb_incl[threadIdx.x] = incl;
            cb.cb_p0[threadIdx.x] = p0;
            cb.cb_blk[threadIdx.x] = (uint16_t)myb;
          }
          __syncthreads();
          uint32_t nblk = s.st[ST_NBLK];
          uint32_t next_pos;
          uint32_t n_pieces = 1, piece_items = 0;
          if (oversize) {   // one block larger than the item buffer: evaluate it in pieces
            nblk = 1;
            n_pieces = (first_cnt + p.items_max - 1) / p.items_max;
          }
          if (nblk == n_live && n_live == n_live_total) next_pos = scan_end;
          else next_pos = (uint32_t)cb.live_pos[nblk - 1] + 1;
          TICK(5);
#ifdef SGPU_PROF
          prof[10] += 1;
#endif

          for (uint32_t piece = 0; piece < n_pieces; ++piece) {
            uint32_t n_items, item0 = 0;
            if (oversize) {
              item0 = piece * p.items_max;
              n_items = first_cnt - item0 < p.items_max ? first_cnt - item0 : p.items_max;
            } else {
              n_items = cb.cb_incl[nblk - 1];
            }
            piece_items = n_items;
            // (c) phase A: posting refs (thread per item; + visited bits in the counted pass)
            if (threadIdx.x == 0) {
              s.st[ST_TMP2] = 0;    // phase B's item counters
              s.st[ST_PULL_L] = 0;
              s.st[ST_NCAND] = 0;   // phase B's candidate counter
            }
            for (uint32_t ib = 0; ib < n_items; ib += NT) {
              const uint32_t i = ib + threadIdx.x;
              if (i >= n_items) {
                classify_item(s.st, cb, p.items_max, 0, 0, false);
                continue;
              }
              const uint32_t gi = i + item0;
              uint32_t lo = 0, hi = nblk;   // first block with cb_incl > gi
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cb.cb_incl[mid] <= gi) lo = mid + 1; else hi = mid;
              }
              const uint32_t excl = lo ? cb.cb_incl[lo - 1] : 0;
              const uint32_t pidx = cb.cb_p0[lo] + (gi - excl);
              const uint32_t doc = ix.post_doc[pidx];
              const uint64_t ref = ix.post_ref[pidx];
              const uint32_t vis = (p.use_bitmap && visited_test(bitmap, doc)) ? 0x80000000u : 0u;
              cb.it_ref[i] = ref;
              cb.it_doc[i] = doc | vis;
              cb.it_blk[i] = (uint16_t)cb.cb_blk[lo];
              classify_item(s.st, cb, p.items_max, i, (uint32_t)(ref & 0xffffu), vis == 0);
            }
            __syncthreads();
            TICK(6);
            // (d) phase B: speculative scoring (score_items)
            score_items<CT, NT, LK>(s, ix, cb, p, spec_docs);
            __syncthreads();
            TICK(7);
            // (e) exact replay on wavefront 0
            if (wave == 0) replay_round<KR>(heap, cb, s, p, n_items, bitmap, decided_blk, piece == 0, wc, dots);
            __syncthreads();
            TICK(8);
          }
          pos = next_pos;
          // adapt the speculation budget to how much of the last round the replay kept
          {
            const uint32_t kept = s.st[ST_TMP1];
            if (kept * 4 >= piece_items * 3) budget = budget * 2 < p.items_max ? budget * 2 : p.items_max;
            else if (kept * 2 < piece_items) budget = budget / 2 > p.items_min ? budget / 2 : p.items_min;
          }
        }
      }

      // ---- kNN refinement (Knn::refine, reference src/inverted_index.rs:551-593): for every document
      // of the current top-k (snapshot, best first) its first n_knn graph neighbours are scored and
      // pushed unless already visited. One more round through the same scoring + replay machinery;
      // "visited" is again heap membership (or the bitmap in the counted pass).
      if (p.n_knn && ix.knn) {
        uint32_t* snap = cb.cb_incl;   // cb_incl + cb_p0 are contiguous: 2*NT words >= k
        if (wave == 0) {
#pragma unroll
          for (int r = 0; r < KR; ++r) {
            const uint32_t e = (uint32_t)r * 64u + lane;
            if (e < heap.len) snap[e] = heap.doc[r];
          }
          if (lane == 0) s.st[ST_TMP0] = heap.len;
        }
        __syncthreads();
        const uint32_t hl = s.st[ST_TMP0];
        const uint32_t nk = p.n_knn < ix.knn_dim ? p.n_knn : ix.knn_dim;
        const uint32_t total = hl * nk;
        uint32_t decided_blk = 0xffffffffu;
        for (uint32_t base = 0; base < total; base += p.items_max) {
          const uint32_t n_items = total - base < p.items_max ? total - base : p.items_max;
          if (threadIdx.x == 0) {
            s.st[ST_TMP2] = 0;
            s.st[ST_PULL_L] = 0;
            s.st[ST_NCAND] = 0;
          }
          for (uint32_t ib = 0; ib < n_items; ib += NT) {
            const uint32_t i = ib + threadIdx.x;
            if (i >= n_items) {
              classify_item(s.st, cb, p.items_max, 0, 0, false);
              continue;
            }
            const uint32_t j = base + i;
            const uint32_t r = j / nk, t = j - r * nk;
            const uint64_t pos = (uint64_t)snap[r] * ix.knn_dim + t;
            bool ok = pos < ix.knn_total;
            const uint32_t nbr = ok ? ix.knn[pos] : 0u;
            ok = ok && nbr < ix.n_docs;
            const uint64_t ref = ok ? ix.doc_ref[nbr] : 0ull;
            const bool vis = !ok || (p.use_bitmap && visited_test(bitmap, nbr));
            cb.it_ref[i] = ref;
            cb.it_doc[i] = nbr | (vis ? 0x80000000u : 0u);
            cb.it_blk[i] = 0;   // no block test in the refinement (replay gets dots == nullptr)
            classify_item(s.st, cb, p.items_max, i, (uint32_t)(ref & 0xffffu), !vis);
          }
          __syncthreads();
          score_items<CT, NT, LK>(s, ix, cb, p, spec_docs);
          __syncthreads();
          if (wave == 0) replay_round<KR>(heap, cb, s, p, n_items, bitmap, decided_blk, true, wc, nullptr, true);
          __syncthreads();
        }
      }
    }

    // ---- results: best first (into_sorted_vec, src/inverted_index.rs:227-233) ----
    if (wave == 0) {
      float* os = qb.out_scores + (size_t)q * qb.k_stride;
      uint64_t* oi = qb.out_ids + (size_t)q * qb.k_stride;
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const uint32_t e = (uint32_t)r * 64u + lane;
        if (e < p.k) {
          os[e] = e < heap.len ? heap.sc[r] : 0.0f;
          oi[e] = e < heap.len ? (uint64_t)heap.doc[r] : ~0ull;
        }
      }
      if (lane == 0) qb.out_n[q] = heap.len;
    }
    if (qb.out_stats) {   // work counters (see DESIGN.md "Algorithmic bytes")
      uint32_t sd = spec_docs;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        sd += __shfl_xor(sd, d);
        wc.blocks += __shfl_xor(wc.blocks, d);
        wc.posts += __shfl_xor(wc.posts, d);
        wc.docs += __shfl_xor(wc.docs, d);
        wc.len += __shfl_xor(wc.len, d);
      }
      uint32_t* os = qb.out_stats + (size_t)q * STATS_WORDS;
      if (lane == 0) atomicAdd(&os[7], sd);
      if (threadIdx.x == 0) {
        uint32_t nbt = 0;
        for (uint32_t l = 0; l < nl; ++l) nbt += s.sel_nb[l];
        os[0] = nbt;
        os[1] = st_rows;
        os[2] = st_entries;
        os[3] = wc.blocks;
        os[4] = wc.posts;
        os[5] = wc.docs;
        os[6] = wc.len;
      }
    }
    TICK(9);
    // ---- per-query cleanup: visited bitmap, query bits ----
    if (p.use_bitmap)
      for (uint32_t i = threadIdx.x; i < ix.n_bitmap_words; i += NT) bitmap[i] = 0;
    __threadfence_block();
    __syncthreads();
    TICK(11);
    if (qb.out_stats && threadIdx.x == 0) {
      uint32_t* os = qb.out_stats + (size_t)q * STATS_WORDS + 8;
#pragma unroll
      for (int i = 0; i < 12; ++i) os[i] = prof[i];
      os[12] = blockIdx.x;
    }
#undef TICK
  }
}

// ---------------------------------------------------------------------------
// host-callable launcher table
// ---------------------------------------------------------------------------
template <typename CT, int NT, int KR, int LK>
static hipError_t run_one(const LaunchArgs& a, int* occupancy) {
  auto kern = seismic_search_kernel<CT, NT, KR, LK>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)a.lds_bytes);
  if (e != hipSuccess) return e;
  if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, kern, NT, a.lds_bytes);
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(NT), a.lds_bytes, a.stream, a.ix, a.qb, a.p, a.L, a.queue,
                     a.bitmaps);
  return hipGetLastError();
}

template <typename CT, int NT, int LK>
static hipError_t run_kr(const LaunchArgs& a, int* occ) {
  const uint32_t k = a.p.k;
  switch (heap_variant(k)) {
    case 1: return run_one<CT, NT, 1, LK>(a, occ);
    case 2: return run_one<CT, NT, 2, LK>(a, occ);
    case 4: return run_one<CT, NT, 4, LK>(a, occ);
    case 8: return run_one<CT, NT, 8, LK>(a, occ);
    default: return run_one<CT, NT, 16, LK>(a, occ);
  }
}

static hipError_t run_any(const LaunchArgs& a, int* occ) {
  if (a.comp_width == 2) {
    if (a.lookup == LK_DENSE) return a.block == 1024 ? run_kr<uint16_t, 1024, LK_DENSE>(a, occ) : run_kr<uint16_t, 512, LK_DENSE>(a, occ);
    return a.block == 1024 ? run_kr<uint16_t, 1024, LK_PACKED>(a, occ) : run_kr<uint16_t, 512, LK_PACKED>(a, occ);
  }
  if (a.lookup == LK_SPLIT) return a.block == 1024 ? run_kr<uint32_t, 1024, LK_SPLIT>(a, occ) : run_kr<uint32_t, 512, LK_SPLIT>(a, occ);
  return a.block == 1024 ? run_kr<uint32_t, 1024, LK_PACKED>(a, occ) : run_kr<uint32_t, 512, LK_PACKED>(a, occ);
}

hipError_t occupancy_search(const LaunchArgs& a, int* n) { return run_any(a, n); }
hipError_t launch_search(const LaunchArgs& a) { return run_any(a, nullptr); }

}  // namespace sgpu
