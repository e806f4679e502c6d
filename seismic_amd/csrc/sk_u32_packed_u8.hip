// sk_u32_packed_u8.hip — the search kernel family for uint32_t components with the LK_PACKED query lookup table,
// fixed-u8 document values (the reference's "fixedu8" value type on large vocabularies, src/bin/perf_inverted_index.rs:125-126).
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u32_packed_u8(const LaunchArgs& a, int* occupancy) { return run_family<uint32_t, LK_PACKED, VT_U8>(a, occupancy); }
}  // namespace sgpu
