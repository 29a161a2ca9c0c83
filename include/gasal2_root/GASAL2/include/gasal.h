/*
 * GASAL2/include/gasal.h -- source-compatible replacement for the subset of the GASAL2 API that
 * sflorescu/BWA-MEM_GPU calls (the GASAL2 submodule is un-vendored; symbols reconstructed from the
 * call sites: /root/reference/src/fastmap.c:417-534, src/bwamem.c:1021-1167,1791-1908,2004-2032,
 * 2106-2211, src/kthread.c:158-161, src/bntseq.h:35-40,74-83).
 * Backed by the MI355X extension kernel behind bmh_extend_batch (include/bwamem_hip.h) in
 * libbwamem_hip.so.  Only algo == KSW / start_pos == WITHOUT_START (what the reference uses) exist.
 *
 * Place this directory where the reference expects the submodule (<reference>/GASAL2/include/) or put
 * <this repo>/include/gasal2_root/src on the include path: "../GASAL2/include/gasal.h" then resolves here.
 */
#ifndef __GASAL_H__
#define __GASAL_H__

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifndef N_CODE
#define N_CODE 4        /* README.md:38 build options of the reference */
#endif
#ifndef N_PENALTY
#define N_PENALTY 1
#endif

enum comp_start { WITH_START, WITHOUT_START };
enum data_source { NONE, QUERY, TARGET, BOTH };
enum algo_type { UNKNOWN, GLOBAL, SEMI_GLOBAL, LOCAL, BANDED, KSW };
enum operation_on_seq { FORWARD_NATURAL, REVERSE_NATURAL, FORWARD_COMPLEMENT, REVERSE_COMPLEMENT };

/* one page of unpacked host sequence data (decoy_cpu_align walks data/offset/next, src/bwamem.c:1813-1868) */
struct host_batch {
	uint8_t *data;
	uint32_t page_size;
	uint32_t data_size;
	uint32_t offset;
	int is_locked;
	struct host_batch *next;
};
typedef struct host_batch host_batch_t;

/* results, one entry per alignment (src/bwamem.c:2207-2211) */
struct gasal_res {
	int32_t *aln_score;
	int32_t *query_batch_end;
	int32_t *target_batch_end;
	int32_t *query_batch_start;
	int32_t *target_batch_start;
};
typedef struct gasal_res gasal_res_t;

typedef struct {
	/* fields the reference's host code reads or writes */
	host_batch_t *extensible_host_unpacked_query_batch;
	host_batch_t *extensible_host_unpacked_target_batch;
	uint32_t *host_query_batch_offsets;
	uint32_t *host_target_batch_offsets;
	uint32_t *host_query_batch_lens;
	uint32_t *host_target_batch_lens;
	uint32_t *host_seed_scores;
	gasal_res_t *host_res;
	uint32_t host_max_query_batch_bytes;
	uint32_t host_max_target_batch_bytes;
	uint32_t host_max_n_alns;
	uint32_t current_n_alns;
	int is_free;
	/* implementation state (MI355X back-end) */
	uint32_t gpu_max_query_batch_bytes, gpu_max_target_batch_bytes, gpu_max_n_alns;
	void *impl;
} gasal_gpu_storage_t;

typedef struct {
	int n;
	gasal_gpu_storage_t *a;
} gasal_gpu_storage_v;

typedef struct {
	int32_t match;
	int32_t mismatch;
	int32_t gap_open;
	int32_t gap_extend;
} gasal_subst_scores;

void gasal_copy_subst_scores(gasal_subst_scores *subst);

/* extras of this back-end (not in GASAL2): ksw_extend2's end bonus (pen_clip5) and z-drop used by the
 * CPU stand-in decoy_cpu_align (src/bwamem.c:1887-1890); defaults 5 and 0 = the reference's mem_opt_init */
void gasal_set_ksw_extras(int end_bonus, int zdrop);

#endif
