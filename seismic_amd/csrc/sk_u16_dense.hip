// sk_u16_dense.hip — the search kernel family for uint16_t components with the LK_DENSE query lookup table, f16 values.
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u16_dense(const LaunchArgs& a, int* occupancy) { return run_family<uint16_t, LK_DENSE, VT_F16>(a, occupancy); }
}  // namespace sgpu
