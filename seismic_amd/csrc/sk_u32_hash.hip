// sk_u32_hash.hip — the search kernel family for uint32_t components with the hashed query lookup
// table (large vocabularies: one random LDS read per document component instead of two), f16 values.
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u32_hash(const LaunchArgs& a, int* occupancy) { return run_family<uint32_t, LK_HASH, VT_F16>(a, occupancy); }
}  // namespace sgpu
