// sk_u16_packed_u8.hip — the search kernel family for uint16_t components with the LK_PACKED query lookup table,
// fixed-u8 document values (the forward index of the reference's DotVByte / fixedu8 indexes).
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u16_packed_u8(const LaunchArgs& a, int* occupancy) { return run_family<uint16_t, LK_PACKED, VT_U8>(a, occupancy); }
}  // namespace sgpu
