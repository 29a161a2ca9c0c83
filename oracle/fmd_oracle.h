/*
 * oracle/fmd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's seed-and-extend arithmetic.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this; the product path (bwa-mem_gpu_amd/csrc, include/) never does.
 *
 * Parity pin: this restatement is checked against the reference's own
 * compiled C (oracle/_ref: src/ksw.c ksw_extend2, src/bwt.c bwt_smem1/bwt_sa)
 * by tests/test_oracle_vs_ref.py and against the committed golden vectors in
 * tests/golden/ (generated from that compiled reference by
 * tests/golden/make_golden.py).
 */
#ifndef FMD_ORACLE_H
#define FMD_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GPU-layout FMD index (reference: src/GPUSeed/seed_gen.h:21-33, layout
 * src/GPUSeed/seed_gen.cu:28-48, files bwa_index/bwtindex.c:174-197). */
typedef struct {
	uint64_t primary;
	uint64_t L2[5];
	uint64_t seq_len;
	uint64_t n_words;        /* words in bwt[] (interleaved occ/bwt blocks) */
	const uint32_t *bwt;
	int sa_intv;
	uint64_t n_sa;
	const uint32_t *sa;      /* sa[0] = 0xFFFFFFFF */
	const uint32_t *sa_bits; /* one upper bit per sample (pack_size == 1) */
} fmd_t;

/* work counters used for the roofline's algorithmic bytes */
typedef struct {
	uint64_t n_blk;      /* 32-byte index blocks the algorithm must touch */
	uint64_t n_sa;       /* located occurrences (one SA sample each) */
	uint64_t n_fwd_steps, n_back_steps, n_lf_steps;
	uint64_t n_blk_fwd, n_blk_back, n_blk_lf;   /* n_blk split by phase */
} fmd_work_t;

/* Occ(k,c) on the full (seq_len+1)-row matrix; reference src/bwt.c:235-261
 * restated for 64-symbol blocks with 32-bit counts (seed_gen.cu:100-120). */
uint64_t fmd_occ(const fmd_t *f, uint64_t k, int c);
void fmd_occ4(const fmd_t *f, uint64_t k, uint64_t cnt[4]);
/* inverse Psi / LF step, CPU form (src/bwt.c:64-70) */
uint64_t fmd_inv_psi(const fmd_t *f, uint64_t k);
/* SA value of row k (src/bwt.c:105-115 with the packed 33rd bit) */
uint64_t fmd_sa(const fmd_t *f, uint64_t k, fmd_work_t *w);

/* seeds of a read set in the reference's mem_seed_v_gpu layout
 * (src/GPUSeed/seed_gen.h:68-75): flat SoA, per read SMEMs by end ascending,
 * occurrences by SA row ascending; score = #occ at the group head, 0 elsewhere
 * (the reference leaves the non-head slots uninitialised, seed_gen.cu:540). */
typedef struct {
	uint64_t n_seeds;
	uint64_t *rbeg;
	int32_t *qbeg;      /* pairs {x = begin, y = end} */
	uint32_t *score;
	uint32_t *n_ref_pos; /* per read */
	uint32_t *prefix;    /* per read, exclusive scan */
	/* the SMEM list itself (before locate), for kernel-level tests */
	uint64_t n_smems;
	uint64_t *smem_k;    /* SA interval start */
	uint32_t *smem_s;    /* SA interval size */
	int32_t *smem_qb, *smem_qe;
	uint32_t *smem_read;
	fmd_work_t work;
} oracle_seeds_t;

/* reads: nt4 codes (0..3, >3 = ambiguous), concatenated; offs/lens per read.
 * Restates bwt_smem1 (src/bwt.c:483-566) driven as the seeding loop of
 * bwa_index/bwamem.c:114-131 (first pass only: the GPU pipeline has no
 * re-seeding, README.md:93) with the length filter src/bwamem.c:260-263,
 * then bwt_sa per occurrence. max_occ_locate == 0 locates every occurrence. */
oracle_seeds_t *oracle_seed_reads(const fmd_t *f, const uint8_t *reads, const uint64_t *offs,
                                  const uint32_t *lens, uint32_t n_reads, int min_seed_len,
                                  int n_threads);
void oracle_seeds_free(oracle_seeds_t *s);

/* ksw_extend2 (src/ksw.c:864-986) restated; opt_ext == 0 path only (what the
 * GPU pipeline uses, src/bwamem.c:1887-1890). Returns score. */
typedef struct {
	int a, b;               /* match score, mismatch penalty (positive) */
	int o_del, e_del, o_ins, e_ins;
	int zdrop, end_bonus;   /* end_bonus == pen_clip5 */
	int n_penalty;          /* score against code 4: -n_penalty (bwa.c:99-108: 1) */
} ksw_params_t;

int oracle_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                       const ksw_params_t *p, int h0, int *qle, int *tle, int *gtle,
                       int *gscore, int *max_off, uint64_t *cells);

/* batch of extensions + the local-vs-to-end rule (src/bwamem.c:1893-1901).
 * out3 = {aln_score, query_end, target_end} per alignment; raw6 (optional) =
 * {score,qle,tle,gtle,gscore,max_off}. Returns total DP cells executed. */
uint64_t oracle_extend_batch(uint32_t n, const uint8_t *q, const uint32_t *qoff, const uint32_t *qlen,
                             const uint8_t *t, const uint32_t *toff, const uint32_t *tlen,
                             const uint32_t *h0, const ksw_params_t *p, int32_t *out3, int32_t *raw6,
                             int n_threads);

uint64_t oracle_extend_trace(uint32_t n, const uint8_t *q, const uint32_t *qoff, const uint32_t *qlen,
                             const uint8_t *t, const uint32_t *toff, const uint32_t *tlen,
                             const uint32_t *h0, const ksw_params_t *p, uint32_t *rows, int16_t *trace, uint64_t cap);

/* ---- region -> CIGAR / NM / MD (cigar_oracle.c; ksw_global2, bwa_gen_cigar2, mem_reg2aln) */
int oracle_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const ksw_params_t *p, int w,
                       int *n_cigar, uint32_t *cigar, int cap);
int oracle_gen_cigar2(const ksw_params_t *p, int w_, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query,
                      int64_t rb, int64_t re, int *score, int *n_cigar, uint32_t *cigar, int cap, int *NM, char *md, int md_cap);
int oracle_reg2aln(const ksw_params_t *p, int opt_w, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *read,
                   int qb, int qe, int64_t rb, int64_t re, int truesc, int reg_w,
                   int64_t *pos, int *is_rev, int *n_cigar, uint32_t *cigar, int cap, int *NM, char *md, int md_cap, int *score);

#ifdef __cplusplus
}
#endif
#endif
