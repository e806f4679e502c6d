// sk_u16_hash.hip — the search kernel family for uint16_t components with the hashed {id, weight} query lookup
// table (one LDS read per document component), f16 values.
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u16_hash(const LaunchArgs& a, int* occupancy) { return run_family<uint16_t, LK_HASH, VT_F16>(a, occupancy); }
}  // namespace sgpu
