// sk_u16_packed_f16s.hip — uint16_t components, LK_PACKED lookup, f16 values behind the compressed component stream (VT_F16S).
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u16_packed_f16s(const LaunchArgs& a, int* occupancy) { return run_family<uint16_t, LK_PACKED, VT_F16S>(a, occupancy); }
}  // namespace sgpu
