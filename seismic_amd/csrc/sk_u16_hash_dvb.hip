// sk_u16_hash_dvb.hip — the search kernel family for uint16_t components with the LK_HASH query lookup table,
// DotVByte forward index: fixed-u8 codes + eight 12-bit component gaps per slice (search_kernel.inc: VT_DVB).
#include "search_kernel.inc"

namespace sgpu {
hipError_t run_u16_hash_dvb(const LaunchArgs& a, int* occupancy) { return run_family<uint16_t, LK_HASH, VT_DVB>(a, occupancy); }
}  // namespace sgpu
