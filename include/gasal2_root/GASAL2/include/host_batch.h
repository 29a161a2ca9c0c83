/* GASAL2/include/host_batch.h -- see gasal.h in this directory. */
#ifndef __HOST_BATCH__
#define __HOST_BATCH__
#include "gasal.h"

/* copy `size` bases (codes 0..4) at byte position idx of the unpacked host batch, pad to a multiple of 8
 * with N_CODE, return the new fill position (src/bwamem.c:1128-1137) */
uint32_t gasal_host_batch_fill(gasal_gpu_storage_t *gpu_storage_t, uint32_t idx, const uint8_t *data, uint32_t size, data_source SRC);
uint32_t gasal_host_batch_fill(gasal_gpu_storage_t *gpu_storage_t, uint32_t idx, const char *data, uint32_t size, data_source SRC);
/* one base (src/bwamem.c:1033-1036) */
uint32_t gasal_host_batch_addbase(gasal_gpu_storage_t *gpu_storage_t, uint32_t idx, const char base, data_source SRC);
host_batch_t *gasal_host_batch_new(uint32_t batch_bytes, uint32_t offset);
void gasal_host_batch_destroy(host_batch_t *res);
#endif
