// sk_u32_packed.hip — the search kernel family for uint32_t components with the LK_PACKED query lookup table, f16 values.
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u32_packed(const LaunchArgs& a, int* occupancy) { return run_family<uint32_t, LK_PACKED, VT_F16>(a, occupancy); }
}  // namespace sgpu
