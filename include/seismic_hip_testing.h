/* seismic_hip_testing.h - TEST HOOKS of libseismic_hip.so. Not part of the drop-in boundary (that is seismic_hip.h):
 * these entry points let tests/ and tools/ look at host-side decisions of the library. They are inert unless the
 * environment has SGPU_TEST_HOOKS=1 (as tests/conftest.py sets it): the status-returning ones then fail with
 * SGPU_EINVAL, the others return 0. The undocumented SGPU_* environment names obey the
 * same switch. */
#ifndef SEISMIC_HIP_TESTING_H
#define SEISMIC_HIP_TESTING_H
#include "seismic_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* team size a host-parallel phase takes for num_threads == 0 (hardware threads capped by the cgroup CPU quota) */
uint32_t sgpu_debug_host_threads(void);
/* how sgpu_batch_search cuts a call of nq queries into launches: bounds[2j], bounds[2j+1] = queries [q0, q1) of launch j */
uint32_t sgpu_debug_chunk_plan(uint32_t nq, uint32_t chunk_min, uint32_t chunk_max, uint32_t want_tail, uint32_t coop_max,
                               uint32_t lanes_free, uint32_t* bounds);
/* the forward store as sgpu_index_upload packs it (document-major records) and every document's ref */
sgpu_status sgpu_debug_pack_forward(const sgpu_index* idx, uint8_t* out_fwd, uint64_t cap, uint64_t* out_doc_ref,
                                    uint64_t* out_bytes);
/* the hashed row directory as sgpu_index_upload builds it (4 words per slot, 4 slots per bucket); *n_buckets == 0: none */
sgpu_status sgpu_debug_row_dir(const sgpu_index* idx, uint32_t* out, uint64_t cap_words, uint64_t* n_words, uint32_t* n_buckets);
/* the launch plan of a batch: processing order, out3 = {block dots needed at most, largest first list, largest list} */
sgpu_status sgpu_debug_plan(const sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                            uint32_t nq, uint32_t query_cut, uint32_t* order_out, uint32_t* out3);
/* the same plan as the DEVICE computes it for staged chunks (needs an uploaded index; 1 ... 16384 queries, query_cut 1 ... 16) */
sgpu_status sgpu_debug_device_plan(sgpu_index* idx, const uint64_t* q_off, const uint32_t* comps, const float* vals,
                                   uint32_t nq, uint32_t query_cut, uint32_t* order_out, uint32_t* out3);
/* the calling thread's staged calls add their host-side phase times to buf8[0..7] from now on (NULL: off) */
void sgpu_debug_call_timing(double* buf8);
/* timeline of the last cooperative launch (trace builds) */
uint32_t sgpu_debug_coop_trace(sgpu_index* idx, uint64_t* out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SEISMIC_HIP_TESTING_H */
