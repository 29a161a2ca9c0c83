// Batch interface of the mate-rescue alignments (pair_kernels.hip) for pair_post.cpp.
#pragma once
#include <cstdint>
#include "../../include/bwamem_hip.h"

// one ksw_align2 call of mem_matesw: the mate `read` of the batch (its reverse complement when is_rev) against text [rb, re)
struct bmh_msw_job_t { int64_t rb, re; uint32_t read; int32_t l_ms, is_rev, xtra; uint32_t bl_off, pad; };

// which call of the pair's walk a job belongs to: hit j of end i (mem_sam_pe's b[i][j]), orientation r
struct bmh_msw_key_t { uint32_t pair; uint16_t j; uint8_t i, r; };
// the regions of a batch behind mem_sort_dedup_patch as they lie on the device (d_dedup [..][16], d_opr / d_roff: their number and first record per read),
// the reads' lengths, the contigs' offsets
struct bmh_rescue_in_t { const int32_t *d_dedup; const uint32_t *d_opr, *d_roff, *d_lens; const int64_t *d_ctg_off; int n_contigs; };

#ifdef __cplusplus
extern "C" {
#endif
int bmh_matesw_device_takes(int l_ms, int64_t tlen, int xtra);
// d_reads / d_offs: the batch's ASCII reads on the device; jobs / out: host arrays (out[n][7] = kswr_t: score, te, qe, score2, te2, tb, qb)
int bmh_matesw_batch_device(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const bmh_ext_params_t *ep,
                            bmh_msw_job_t *jobs, uint64_t n_jobs, int32_t *out, void *stream);
// the same with the jobs found on the device (pair_kernels.hip: rescue_jobs_kernel): see there
int64_t bmh_rescue_count_device(const bmh_index_t *idx, const bmh_rescue_in_t *in, const bmh_ext_params_t *ep, int min_seed_len, const bmh_pe_opt_t *pe,
                                const double *pes, uint32_t n_reads, uint32_t *pair_off, uint8_t *active, void *stream);
int bmh_rescue_run_device(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const bmh_ext_params_t *ep, bmh_msw_key_t *keys, int32_t *out, void *stream);
#ifdef __cplusplus
}
#endif
