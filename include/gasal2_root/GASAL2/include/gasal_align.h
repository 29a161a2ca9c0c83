/* GASAL2/include/gasal_align.h -- see gasal.h in this directory. */
#ifndef __GASAL_ALIGN_H__
#define __GASAL_ALIGN_H__
#include "gasal.h"
#include "args_parser.h"

/* H2D + extension kernel + D2H, asynchronous on the storage's stream (src/bwamem.c:2127) */
void gasal_aln_async(gasal_gpu_storage_t *gpu_storage, const uint32_t actual_query_batch_bytes, const uint32_t actual_target_batch_bytes,
                     const uint32_t actual_n_alns, Parameters *params);
/* 0 = finished, results are in host_res and is_free is set; -1 = still running; -2 = nothing launched
 * (src/bwamem.c:2181) */
int gasal_is_aln_async_done(gasal_gpu_storage_t *gpu_storage);
#endif
