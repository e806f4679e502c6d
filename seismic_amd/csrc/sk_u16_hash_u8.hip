// sk_u16_hash_u8.hip — the search kernel family for uint16_t components with the hashed {id, weight} query lookup
// table (one LDS read per document component), fixed-u8 document values.
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u16_hash_u8(const LaunchArgs& a, int* occupancy) { return run_family<uint16_t, LK_HASH, VT_U8>(a, occupancy); }
}  // namespace sgpu
