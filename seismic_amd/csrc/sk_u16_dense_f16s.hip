// sk_u16_dense_f16s.hip — uint16_t components, LK_DENSE lookup, f16 values behind the compressed component stream (VT_F16S:
// the sliced internal layout of an f16 index, chosen at upload).
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u16_dense_f16s(const LaunchArgs& a, int* occupancy) { return run_family<uint16_t, LK_DENSE, VT_F16S>(a, occupancy); }
}  // namespace sgpu
