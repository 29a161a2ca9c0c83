/*
 * include/seed_gen.h -- drop-in replacement for the reference's seeding-library header
 * /root/reference/src/GPUSeed/seed_gen.h (C ABI at :92-106), backed by
 * libbwamem_hip.so (hand-written HIP for MI355X / gfx950).
 *
 * Same type names, field names, field order and function signatures as the
 * reference header, so src/fastmap.c:436-465 and src/bwamem.c:404-433 compile and
 * link against it unchanged.  Differences a maintainer should know:
 *   - int2 / uint2 are defined here for host C/C++ translation units (the
 *     reference gets them from the CUDA headers); layout {x, y}, identical.
 *   - bwt_t_gpu returned by gpu_cpy_wrapper() carries device pointers in
 *     .bwt/.sa/.sa_upper_bits exactly like the reference, and a HOST copy of L2 in
 *     .L2 (the reference leaves .L2 dangling and keeps L2 in __constant__ memory).
 *   - score[] is n_occ at the first slot of each SMEM group and 0 elsewhere (the
 *     reference leaves the other slots uninitialised, seed_gen.cu:540,2044).
 *   - errors: like the reference, fatal errors print to stderr and exit(EXIT_FAILURE)
 *     (seed_gen.cu:11-14); there is no CPU fallback.
 */
#ifndef __SEED_GEN_H__
#define __SEED_GEN_H__

#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>

#if !defined(__HIPCC__) && !defined(__CUDACC__) && !defined(BMH_HAVE_VECTOR_TYPES)
typedef struct { int x, y; } int2;
typedef struct { unsigned int x, y; } uint2;
#endif

typedef uint64_t bwtint_t_gpu;

typedef struct {
	bwtint_t_gpu primary;     /* S^{-1}(0), or the primary index of BWT */
	bwtint_t_gpu *L2;
	bwtint_t_gpu seq_len;     /* sequence length */
	bwtint_t_gpu bwt_size;    /* size of bwt in 32-bit words */
	uint32_t *bwt;            /* BWT + Occ blocks (32 bytes per 64 symbols) */
	int sa_intv;
	bwtint_t_gpu n_sa;
	uint32_t *sa;             /* low 32 bits of each sample */
	uint32_t *sa_upper_bits;  /* packed upper bit(s) */
	uint8_t pack_size;
} bwt_t_gpu;

typedef struct {
	int64_t offset;
	int32_t len;
	int32_t n_ambs;
	uint32_t gi;
	int32_t is_alt;
	char *name, *anno;
} bntann2_t;

typedef struct {
	int64_t offset;
	int32_t len;
	char amb;
} bntamb2_t;

typedef struct {
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	bntann2_t *anns;
	int32_t n_holes;
	bntamb2_t *ambs;
	FILE *fp_pac;
} bntseq2_t;

typedef struct {
	int64_t rbeg;
	int32_t qbeg, len;
	int score;
} mem_seed_t;

typedef struct { size_t n, m; mem_seed_t *a; int seed_counter; } mem_seed_v;

/* seeds of the whole read file, flat SoA (reference seed_gen.h:68-75) */
typedef struct {
	bwtint_t_gpu *rbeg;                        /* [n_seeds] position in the fwd+revcomp text */
	int2 *qbeg;                                /* [n_seeds] {x = begin, y = end} in the read */
	uint32_t *score;                           /* [n_seeds] #occurrences at each group head */
	uint32_t *n_ref_pos_fow_rev_results;       /* [n_reads] occurrences per read */
	uint32_t *n_ref_pos_fow_rev_prefix_sums;   /* [n_reads] exclusive scan over the file */
	uint64_t file_bytes_skip;
} mem_seed_v_gpu;

typedef struct {
	char *read_file;
	char *query_file;          /* (sic) the reference prefix, src/fastmap.c:438 */
	bwt_t_gpu *bwt;
	bwt_t_gpu bwt_gpu;
	uint2 *pre_calc_seed_intervals;
	int pre_calc_seed_intervals_flag;
	int pre_calc_seed_len;
	int min_seed_size;
	int is_smem;
	uint64_t file_bytes_skip;
} gpuseed_storage_vector;

#ifdef __cplusplus
extern "C" {
#endif

/* reference seed_gen.cu:1338 */ void bwt_destroy_gpu(bwt_t_gpu *bwt);
/* reference seed_gen.cu:1386 */ void bwt_restore_sa_gpu(const char *fn, bwt_t_gpu *bwt);
/* reference seed_gen.cu:1438 */ bwt_t_gpu *bwt_restore_bwt_gpu(const char *fn);
/* reference seed_gen.cu:1524 */ bwt_t_gpu gpu_cpy_wrapper(bwt_t_gpu *bwt);
/* reference seed_gen.cu:1558 (never called by the host, src/fastmap.c:455) */
void pre_calc_seed_intervals_wrapper(uint2 *pre_calc_seed_intervals, int pre_calc_seed_len, bwt_t_gpu bwt_gpu);
/* reference seed_gen.cu:1348 */ void free_gpuseed_data(gpuseed_storage_vector *gpuseed_data);
/* reference seed_gen.cu:1625 */ mem_seed_v_gpu *seed_gpu(gpuseed_storage_vector *gpuseed_data);

/* number of reads seeded by the last seed_gpu() call (the reference returns no count;
 * its caller learns it from bseq_read).  Extension, not in the reference header. */
uint64_t seed_gpu_last_n_reads(void);

#ifdef __cplusplus
}
#endif

#endif
