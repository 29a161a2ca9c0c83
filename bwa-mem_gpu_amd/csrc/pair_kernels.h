// Batch interface of the mate-rescue alignments (pair_kernels.hip) for pair_post.cpp.
#pragma once
#include <cstdint>
#include "../../include/bwamem_hip.h"

// one ksw_align2 call of mem_matesw: the mate `read` of the batch (its reverse complement when is_rev) against text [rb, re)
struct bmh_msw_job_t { int64_t rb, re; uint32_t read; int32_t l_ms, is_rev, xtra; uint32_t bl_off, pad; };

#ifdef __cplusplus
extern "C" {
#endif
int bmh_matesw_device_takes(int l_ms, int64_t tlen, int xtra);
// d_reads / d_offs: the batch's ASCII reads on the device; jobs / out: host arrays (out[n][7] = kswr_t: score, te, qe, score2, te2, tb, qb)
int bmh_matesw_batch_device(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const bmh_ext_params_t *ep,
                            bmh_msw_job_t *jobs, uint64_t n_jobs, int32_t *out, void *stream);
#ifdef __cplusplus
}
#endif
