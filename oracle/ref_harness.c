/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thin driver around the REFERENCE's own compiled C for the hot path.  It is
 * linked (by oracle/Makefile, target _ref) with objects compiled directly
 * from /root/reference/src/{ksw,bwt,utils,malloc_wrap,kstring}.c -- no
 * reference source is copied into this repository -- and produces
 * oracle/_ref/libref.so.  It exists only to pin the oracle C files and the golden
 * vectors in tests/golden/ to the reference:
 *   ref_extend_batch : reference ksw_extend2 (src/ksw.c:864) + the rule of
 *                      decoy_cpu_align (src/bwamem.c:1893-1901)
 *   ref_seed_reads   : reference bwt_smem1 (src/bwt.c:563) + bwt_sa (:105)
 *                      driven like bwa_index/bwamem.c:114-131 (first pass)
 * The reference's CPU FM-index code needs the vanilla layout (128-symbol
 * blocks, 64-bit counts, src/bwt.h:35,91-92); ref_bwt_from_symbols builds that
 * in memory from the plain BWT symbol string and lets the reference compute
 * its own SA samples (bwt_cal_sa, src/bwt.c:81-103).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "bwt.h"
#include "ksw.h"
#include "kvec.h"

/* build a vanilla-layout bwt_t from BWT symbols (codes 0..3, $ row removed) */
bwt_t *ref_bwt_from_symbols(const uint8_t *sym, uint64_t seq_len, uint64_t primary, const uint64_t L2[5], int sa_intv)
{
	bwt_t *b = (bwt_t *)calloc(1, sizeof(bwt_t));
	uint64_t n_occ = (seq_len + OCC_INTERVAL - 1) / OCC_INTERVAL + 1;
	uint64_t raw_words = (seq_len + 15) >> 4, i, k;
	uint64_t c[4] = {0, 0, 0, 0};
	b->primary = primary;
	memcpy(b->L2, L2, 5 * sizeof(uint64_t));
	b->seq_len = seq_len;
	b->bwt_size = raw_words + n_occ * sizeof(bwtint_t);
	b->bwt = (uint32_t *)calloc(b->bwt_size, 4);
	for (i = k = 0; i < seq_len; ++i) {
		if (i % OCC_INTERVAL == 0) { memcpy(b->bwt + k, c, 4 * sizeof(bwtint_t)); k += sizeof(bwtint_t); }
		if (i % 16 == 0) ++k;
		b->bwt[k - 1] |= (uint32_t)sym[i] << ((~i & 15) << 1);
		++c[sym[i]];
	}
	memcpy(b->bwt + k, c, 4 * sizeof(bwtint_t));
	bwt_gen_cnt_table(b);
	bwt_cal_sa(b, sa_intv);
	return b;
}

/* The same vanilla-layout bwt_t straight from the GPU-layout words, for hg38-scale texts where the symbol string (6.2 GB) and
 * the reference's own sequential bwt_cal_sa (6.2e9 dependent LF steps) are too slow to be setup of a benchmark: the 16-symbol
 * words are identical in both layouts, the 32-bit Occ quadruple of every second 64-symbol block is widened to 64 bits
 * (src/bwt.h:33-36,91-92), and the suffix-array samples are copied from the 32+1-bit samples of the GPU files (same rows:
 * multiples of sa_intv, sa[0] = -1; src/bwt.c:105-115).  What runs on it afterwards is the reference's compiled code. */
bwt_t *ref_bwt_from_gpu_layout(const uint32_t *gw, uint64_t seq_len, uint64_t primary, const uint64_t L2[5], int sa_intv,
                               const uint32_t *sa32, const uint32_t *sa_bits)
{
	bwt_t *b = (bwt_t *)calloc(1, sizeof(bwt_t));
	uint64_t n_occ = (seq_len + OCC_INTERVAL - 1) / OCC_INTERVAL + 1;
	uint64_t raw_words = (seq_len + 15) >> 4, nb128 = (seq_len + 127) >> 7, n64 = (seq_len + 63) >> 6, i;
	int c;
	b->primary = primary;
	memcpy(b->L2, L2, 5 * sizeof(uint64_t));
	b->seq_len = seq_len;
	b->bwt_size = raw_words + n_occ * sizeof(bwtint_t);
	b->bwt = (uint32_t *)calloc((nb128 + 1) * 16, 4);
	for (i = 0; i < nb128; ++i) {
		const uint32_t *g0 = gw + (2 * i) * 8;
		uint32_t *o = b->bwt + i * 16;
		uint64_t occ[4];
		for (c = 0; c < 4; ++c) occ[c] = g0[c];
		memcpy(o, occ, 32);
		memcpy(o + 8, g0 + 4, 16);
		if (2 * i + 1 < n64) memcpy(o + 12, g0 + 12, 16);
	}
	{	/* trailing Occ quadruple = totals */
		uint64_t tot[4];
		for (c = 0; c < 4; ++c) tot[c] = L2[c + 1] - L2[c];
		memcpy(b->bwt + nb128 * 16, tot, 32);
	}
	bwt_gen_cnt_table(b);
	b->sa_intv = sa_intv;
	b->n_sa = (seq_len + sa_intv) / sa_intv;
	b->sa = (bwtint_t *)malloc(b->n_sa * sizeof(bwtint_t));
	b->sa[0] = (bwtint_t)-1;
	for (i = 1; i < b->n_sa; ++i) b->sa[i] = (uint64_t)sa32[i] | ((uint64_t)((sa_bits[i >> 5] >> (i & 31)) & 1u) << 32);
	return b;
}

void ref_bwt_free(bwt_t *b) { if (b) { free(b->sa); free(b->bwt); free(b); } }

uint64_t ref_occ(const bwt_t *b, uint64_t k, int c) { return bwt_occ(b, k, c); }
uint64_t ref_sa(const bwt_t *b, uint64_t k) { return bwt_sa(b, k); }

/* seeds in mem_seed_v_gpu order; two-call protocol: first with rbeg == NULL to count */
typedef struct { size_t n, m; uint64_t *k; uint32_t *s; int32_t *qb, *qe; uint32_t *rd; } ref_smems_t;

ref_smems_t *ref_collect_smems(const bwt_t *b, const uint8_t *reads, const uint64_t *offs, const uint32_t *lens,
                               uint32_t n_reads, int min_seed_len)
{
	ref_smems_t *o = (ref_smems_t *)calloc(1, sizeof(ref_smems_t));
	bwtintv_v mem = {0, 0, 0}, t0 = {0, 0, 0}, t1 = {0, 0, 0};
	bwtintv_v *tmpv[2] = {&t0, &t1};
	for (uint32_t r = 0; r < n_reads; ++r) {
		const uint8_t *q = reads + offs[r];
		int len = (int)lens[r], x = 0;
		while (x < len) {
			if (q[x] < 4) {
				x = bwt_smem1(b, len, q, x, 1, &mem, tmpv);
				for (size_t i = 0; i < mem.n; ++i) {
					int beg = (int)(mem.a[i].info >> 32), end = (int)(uint32_t)mem.a[i].info;
					if (end - beg < min_seed_len) continue;
					if (o->n == o->m) {
						o->m = o->m ? o->m * 2 : 1024;
						o->k = (uint64_t *)realloc(o->k, o->m * 8); o->s = (uint32_t *)realloc(o->s, o->m * 4);
						o->qb = (int32_t *)realloc(o->qb, o->m * 4); o->qe = (int32_t *)realloc(o->qe, o->m * 4);
						o->rd = (uint32_t *)realloc(o->rd, o->m * 4);
					}
					o->k[o->n] = mem.a[i].x[0]; o->s[o->n] = (uint32_t)mem.a[i].x[2];
					o->qb[o->n] = beg; o->qe[o->n] = end; o->rd[o->n] = r; o->n++;
				}
			} else ++x;
		}
	}
	free(mem.a); free(t0.a); free(t1.a);
	return o;
}

uint64_t ref_smems_n(const ref_smems_t *o) { return o->n; }
void ref_smems_get(const ref_smems_t *o, uint64_t *k, uint32_t *s, int32_t *qb, int32_t *qe, uint32_t *rd)
{
	memcpy(k, o->k, o->n * 8); memcpy(s, o->s, o->n * 4); memcpy(qb, o->qb, o->n * 4);
	memcpy(qe, o->qe, o->n * 4); memcpy(rd, o->rd, o->n * 4);
}
void ref_smems_free(ref_smems_t *o) { free(o->k); free(o->s); free(o->qb); free(o->qe); free(o->rd); free(o); }

/* locate rows [k, k+s) with the reference's bwt_sa */
void ref_locate(const bwt_t *b, uint64_t k, uint32_t s, uint64_t *out)
{
	for (uint32_t t = 0; t < s; ++t) out[t] = bwt_sa(b, k + t);
}

/* reference ksw_extend2, opt_ext = 0, then the local/to-end rule */
void ref_extend_batch(uint32_t n, const uint8_t *q, const uint32_t *qoff, const uint32_t *qlen,
                      const uint8_t *t, const uint32_t *toff, const uint32_t *tlen, const uint32_t *h0,
                      int a, int bmis, int o_del, int e_del, int o_ins, int e_ins, int w, int zdrop, int pen_clip5,
                      int32_t *out3, int32_t *raw6)
{
	int8_t mat[25];
	int i, j, k;
	for (i = k = 0; i < 4; ++i) {               /* same values as bwa_fill_scmat, src/bwa.c:99-108 */
		for (j = 0; j < 4; ++j) mat[k++] = i == j ? a : -bmis;
		mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) mat[k++] = -1;
	for (uint32_t x = 0; x < n; ++x) {
		int qle, tle, gtle, gscore, max_off;
		int score = ksw_extend2((int)qlen[x], q + qoff[x], (int)tlen[x], t + toff[x], 5, mat, o_del, e_del, o_ins, e_ins,
		                        w, pen_clip5, zdrop, (int)h0[x], &qle, &tle, &gtle, &gscore, &max_off, 0);
		if (raw6) { int32_t *r = raw6 + 6 * (uint64_t)x; r[0] = score; r[1] = qle; r[2] = tle; r[3] = gtle; r[4] = gscore; r[5] = max_off; }
		int32_t *o = out3 + 3 * (uint64_t)x;
		if (gscore <= 0 || gscore <= score - pen_clip5) { o[0] = score; o[1] = qle; o[2] = tle; }
		else { o[0] = gscore; o[1] = (int32_t)qlen[x]; o[2] = gtle; }
	}
}

/* reference ksw_global2 (src/ksw.c:1120): score + CIGAR; returns the number of ops (copies at most cap) */
int ref_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int a, int bmis, int o_del, int e_del, int o_ins, int e_ins,
                int w, int *score, uint32_t *cigar, int cap)
{
	int8_t mat[25];
	int i, j, k, n_cigar = 0;
	uint32_t *cg = 0;
	for (i = k = 0; i < 4; ++i) {
		for (j = 0; j < 4; ++j) mat[k++] = i == j ? a : -bmis;
		mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) mat[k++] = -1;
	*score = ksw_global2(qlen, query, tlen, target, 5, mat, o_del, e_del, o_ins, e_ins, w, &n_cigar, &cg);
	for (i = 0; i < n_cigar && i < cap; ++i) cigar[i] = cg[i];
	free(cg);
	return n_cigar;
}

/* reference ksw_align2 (src/ksw.c:698), scalar SSE2 kernels (avx2 = 0), as mem_matesw calls it */
void ref_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int a, int bmis, int o_del, int e_del, int o_ins, int e_ins, int xtra, int32_t out[7])
{
	int8_t mat[25];
	int i, j, k;
	for (i = k = 0; i < 4; ++i) {
		for (j = 0; j < 4; ++j) mat[k++] = i == j ? a : -bmis;
		mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) mat[k++] = -1;
	kswr_t r = ksw_align2(qlen, query, tlen, target, 5, mat, o_del, e_del, o_ins, e_ins, xtra, 0, 0);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}
