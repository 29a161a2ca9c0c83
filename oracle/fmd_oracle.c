/*
 * oracle/fmd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see fmd_oracle.h).
 *
 * CPU restatement of the seeding half of the hot path on the reference's
 * GPU-layout FMD index: Occ / bi-interval extension / SMEM collection / SA
 * locate.  Follows (restates, does not copy):
 *   Occ, 2occ4          /root/reference/src/bwt.c:227-261, 363-405
 *                       (block layout of src/GPUSeed/seed_gen.cu:28-120)
 *   bwt_extend          src/bwt.c:455-470
 *   bwt_smem1a          src/bwt.c:483-561  (max_intv = 0, min_intv = 1)
 *   seeding loop        bwa_index/bwamem.c:114-131 (first pass only)
 *   length filter       src/bwamem.c:260-263 / seed_gen.cu:1070
 *   bwt_invPsi, bwt_sa  src/bwt.c:64-70, 105-115 (+33rd bit, seed_gen.cu:653-657)
 *   output layout       src/GPUSeed/seed_gen.h:68-75, seed_gen.cu:520-545,2073-2155
 */
#include "fmd_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- rank ---------------------------------------------------------------- */

/* number of symbols equal to c among the first nsym (0..16) symbols of w
 * (symbol t sits at bits 31-2t..30-2t) */
static inline int cnt16(uint32_t w, int c, int nsym)
{
	uint32_t x = ~(w ^ (0x55555555u * (uint32_t)c));
	x = x & (x >> 1) & 0x55555555u;
	if (nsym < 16) x = nsym ? (x & (0xFFFFFFFFu << (2 * (16 - nsym)))) : 0;
	return __builtin_popcount(x);
}

static inline int sym_at(const fmd_t *f, uint64_t x)
{
	const uint32_t *p = f->bwt + (x >> 6) * 8 + 4;
	uint32_t w = p[(x & 63) >> 4];
	return (w >> (30 - 2 * (x & 15))) & 3;
}

uint64_t fmd_occ(const fmd_t *f, uint64_t k, int c)
{
	if (k == f->seq_len) return f->L2[c + 1] - f->L2[c];
	if (k == (uint64_t)-1) return 0;
	k -= (k >= f->primary);
	const uint32_t *p = f->bwt + (k >> 6) * 8;
	uint64_t n = p[c];
	int off = (int)(k & 63), w;
	for (w = 0; w < (off >> 4); ++w) n += cnt16(p[4 + w], c, 16);
	n += cnt16(p[4 + w], c, (off & 15) + 1);
	return n;
}

void fmd_occ4(const fmd_t *f, uint64_t k, uint64_t cnt[4])
{
	for (int c = 0; c < 4; ++c) cnt[c] = fmd_occ(f, k, c);
}

/* blocks a (k,l) rank pair must touch: 0 for the closed-form cases, 1 when both
 * land in one 64-symbol block, else 2 (seed_gen.cu:191-195,230-234) */
static inline int pair_blocks(const fmd_t *f, uint64_t k, uint64_t l)
{
	int sk = (k == f->seq_len || k == (uint64_t)-1), sl = (l == f->seq_len || l == (uint64_t)-1);
	if (sk && sl) return 0;
	if (sk || sl) return 1;
	uint64_t bk = (k - (k >= f->primary)) >> 6, bl = (l - (l >= f->primary)) >> 6;
	return bk == bl ? 1 : 2;
}

/* ---- bi-interval extension ------------------------------------------------ */

typedef struct { uint64_t x[3]; int beg, end; } bi_t; /* x[0]=fwd start, x[1]=rev-comp start, x[2]=size */

static void bi_extend(const fmd_t *f, const bi_t *ik, bi_t ok[4], int is_back, fmd_work_t *w)
{
	uint64_t tk[4], tl[4];
	int a = !is_back, b = is_back, c;
	uint64_t lo = ik->x[a] - 1, hi = ik->x[a] - 1 + ik->x[2];
	fmd_occ4(f, lo, tk);
	fmd_occ4(f, hi, tl);
	if (w) { int nb = pair_blocks(f, lo, hi); w->n_blk += nb; if (is_back) w->n_blk_back += nb; else w->n_blk_fwd += nb; }
	for (c = 0; c < 4; ++c) {
		ok[c].x[a] = f->L2[c] + 1 + tk[c];
		ok[c].x[2] = tl[c] - tk[c];
	}
	/* the interval of the other strand: T < G < C < A order of complements,
	 * shifted by one when the $ row falls inside [x[a], x[a]+s) */
	ok[3].x[b] = ik->x[b] + (ik->x[a] <= f->primary && ik->x[a] + ik->x[2] - 1 >= f->primary);
	ok[2].x[b] = ok[3].x[b] + ok[3].x[2];
	ok[1].x[b] = ok[2].x[b] + ok[2].x[2];
	ok[0].x[b] = ok[1].x[b] + ok[1].x[2];
}

typedef struct { bi_t *a; int n, m; } bi_v;
static inline void bi_push(bi_v *v, const bi_t *p)
{
	if (v->n == v->m) { v->m = v->m ? v->m * 2 : 64; v->a = (bi_t *)realloc(v->a, v->m * sizeof(bi_t)); }
	v->a[v->n++] = *p;
}
static inline void bi_reverse(bi_v *v)
{
	for (int i = 0; i < v->n >> 1; ++i) { bi_t t = v->a[i]; v->a[i] = v->a[v->n - 1 - i]; v->a[v->n - 1 - i] = t; }
}

/* all SMEMs covering position x; returns the start of the next pass */
static int smem_pass(const fmd_t *f, int len, const uint8_t *q, int x, bi_v *mem, bi_v *va, bi_v *vb, fmd_work_t *w)
{
	bi_t ik, ok[4];
	bi_v *prev = va, *curr = vb, *sw;
	int i, j, c, ret;
	mem->n = 0;
	if (q[x] > 3) return x + 1;
	ik.x[0] = f->L2[q[x]] + 1;
	ik.x[2] = f->L2[q[x] + 1] - f->L2[q[x]];
	ik.x[1] = f->L2[3 - q[x]] + 1;
	ik.beg = x; ik.end = x + 1;
	curr->n = 0;
	for (i = x + 1; i < len; ++i) {           /* forward */
		if (q[i] < 4) {
			c = 3 - q[i];
			bi_extend(f, &ik, ok, 0, w);
			if (w) w->n_fwd_steps++;
			if (ok[c].x[2] != ik.x[2]) {
				bi_push(curr, &ik);
				if (ok[c].x[2] < 1) break;
			}
			ok[c].beg = x; ok[c].end = i + 1;
			ik = ok[c];
		} else {
			bi_push(curr, &ik);
			break;
		}
	}
	if (i == len) bi_push(curr, &ik);
	bi_reverse(curr);                         /* longest first */
	ret = curr->a[0].end;
	sw = curr; curr = prev; prev = sw;
	for (i = x - 1; i >= -1; --i) {           /* backward */
		c = i < 0 ? -1 : (q[i] < 4 ? q[i] : -1);
		curr->n = 0;
		for (j = 0; j < prev->n; ++j) {
			bi_t *p = &prev->a[j];
			if (c >= 0) { bi_extend(f, p, ok, 1, w); if (w) w->n_back_steps++; }
			if (c < 0 || ok[c].x[2] < 1) {
				if (curr->n == 0 && (mem->n == 0 || i + 1 < mem->a[mem->n - 1].beg)) {
					bi_t m = *p;
					m.beg = i + 1;
					bi_push(mem, &m);
				}
			} else if (curr->n == 0 || ok[c].x[2] != curr->a[curr->n - 1].x[2]) {
				ok[c].beg = i; ok[c].end = p->end;
				bi_push(curr, &ok[c]);
			}
		}
		if (curr->n == 0) break;
		sw = curr; curr = prev; prev = sw;
	}
	bi_reverse(mem);                          /* by begin ascending */
	return ret;
}

/* ---- locate ---------------------------------------------------------------- */

uint64_t fmd_inv_psi(const fmd_t *f, uint64_t k)
{
	if (k == f->primary) return 0;
	uint64_t x = k - (k > f->primary);
	int c = sym_at(f, x);
	return f->L2[c] + fmd_occ(f, k, c);
}

uint64_t fmd_sa(const fmd_t *f, uint64_t k, fmd_work_t *w)
{
	uint64_t steps = 0, mask = (uint64_t)f->sa_intv - 1;
	while (k & mask) {
		++steps;
		if (w && k != f->primary) { w->n_blk++; w->n_blk_lf++; }
		k = fmd_inv_psi(f, k);
	}
	if (w) { w->n_lf_steps += steps; w->n_sa++; }
	uint64_t idx = k / (uint64_t)f->sa_intv;
	if (idx == 0) return steps - 1;           /* sa[0] == (bwtint_t)-1, src/bwt.c:103 */
	uint64_t hi = (f->sa_bits[idx >> 5] >> (idx & 31)) & 1u;
	return ((uint64_t)f->sa[idx] | (hi << 32)) + steps;
}

/* ---- driver ---------------------------------------------------------------- */

typedef struct {
	uint64_t *k; uint32_t *s; int32_t *qb, *qe; uint32_t *rd;
	uint64_t n, m;
} smem_buf_t;

static void smem_buf_push(smem_buf_t *b, uint64_t k, uint32_t s, int qb, int qe, uint32_t rd)
{
	if (b->n == b->m) {
		b->m = b->m ? b->m * 2 : 1024;
		b->k = (uint64_t *)realloc(b->k, b->m * 8);
		b->s = (uint32_t *)realloc(b->s, b->m * 4);
		b->qb = (int32_t *)realloc(b->qb, b->m * 4);
		b->qe = (int32_t *)realloc(b->qe, b->m * 4);
		b->rd = (uint32_t *)realloc(b->rd, b->m * 4);
	}
	b->k[b->n] = k; b->s[b->n] = s; b->qb[b->n] = qb; b->qe[b->n] = qe; b->rd[b->n] = rd;
	b->n++;
}

typedef struct {
	const fmd_t *f; const uint8_t *reads; const uint64_t *offs; const uint32_t *lens;
	uint32_t r0, r1; int min_seed_len;
	smem_buf_t sm;
	uint64_t *rbeg; uint64_t n_occ;
	fmd_work_t work;
} seed_job_t;

static void *seed_worker(void *arg)
{
	seed_job_t *j = (seed_job_t *)arg;
	bi_v mem = {0, 0, 0}, va = {0, 0, 0}, vb = {0, 0, 0};
	for (uint32_t r = j->r0; r < j->r1; ++r) {
		const uint8_t *q = j->reads + j->offs[r];
		int len = (int)j->lens[r], x = 0;
		while (x < len) {
			if (q[x] < 4) {
				x = smem_pass(j->f, len, q, x, &mem, &va, &vb, &j->work);
				for (int i = 0; i < mem.n; ++i)
					if (mem.a[i].end - mem.a[i].beg >= j->min_seed_len)
						smem_buf_push(&j->sm, mem.a[i].x[0], (uint32_t)mem.a[i].x[2], mem.a[i].beg, mem.a[i].end, r);
			} else ++x;
		}
	}
	uint64_t tot = 0;
	for (uint64_t i = 0; i < j->sm.n; ++i) tot += j->sm.s[i];
	j->n_occ = tot;
	j->rbeg = (uint64_t *)malloc((tot ? tot : 1) * 8);
	uint64_t o = 0;
	for (uint64_t i = 0; i < j->sm.n; ++i)
		for (uint32_t t = 0; t < j->sm.s[i]; ++t)
			j->rbeg[o++] = fmd_sa(j->f, j->sm.k[i] + t, &j->work);
	free(mem.a); free(va.a); free(vb.a);
	return 0;
}

oracle_seeds_t *oracle_seed_reads(const fmd_t *f, const uint8_t *reads, const uint64_t *offs,
                                  const uint32_t *lens, uint32_t n_reads, int min_seed_len, int n_threads)
{
	if (n_threads < 1) n_threads = 1;
	if ((uint32_t)n_threads > n_reads && n_reads) n_threads = (int)n_reads;
	seed_job_t *jobs = (seed_job_t *)calloc(n_threads, sizeof(seed_job_t));
	pthread_t *tid = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
	for (int t = 0; t < n_threads; ++t) {
		jobs[t].f = f; jobs[t].reads = reads; jobs[t].offs = offs; jobs[t].lens = lens;
		jobs[t].r0 = (uint32_t)((uint64_t)n_reads * t / n_threads);
		jobs[t].r1 = (uint32_t)((uint64_t)n_reads * (t + 1) / n_threads);
		jobs[t].min_seed_len = min_seed_len;
		if (n_threads > 1) pthread_create(&tid[t], 0, seed_worker, &jobs[t]);
		else seed_worker(&jobs[t]);
	}
	if (n_threads > 1) for (int t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
	oracle_seeds_t *o = (oracle_seeds_t *)calloc(1, sizeof(oracle_seeds_t));
	uint64_t ns = 0, no = 0;
	for (int t = 0; t < n_threads; ++t) { ns += jobs[t].sm.n; no += jobs[t].n_occ; }
	o->n_smems = ns; o->n_seeds = no;
	o->smem_k = (uint64_t *)malloc((ns + 1) * 8); o->smem_s = (uint32_t *)malloc((ns + 1) * 4);
	o->smem_qb = (int32_t *)malloc((ns + 1) * 4); o->smem_qe = (int32_t *)malloc((ns + 1) * 4);
	o->smem_read = (uint32_t *)malloc((ns + 1) * 4);
	o->rbeg = (uint64_t *)malloc((no + 1) * 8); o->qbeg = (int32_t *)malloc((no + 1) * 8);
	o->score = (uint32_t *)calloc(no + 1, 4);
	o->n_ref_pos = (uint32_t *)calloc(n_reads + 1, 4); o->prefix = (uint32_t *)calloc(n_reads + 1, 4);
	uint64_t is = 0, io = 0;
	for (int t = 0; t < n_threads; ++t) {
		seed_job_t *j = &jobs[t];
		uint64_t lo = 0;
		for (uint64_t i = 0; i < j->sm.n; ++i, ++is) {
			o->smem_k[is] = j->sm.k[i]; o->smem_s[is] = j->sm.s[i];
			o->smem_qb[is] = j->sm.qb[i]; o->smem_qe[is] = j->sm.qe[i]; o->smem_read[is] = j->sm.rd[i];
			o->n_ref_pos[j->sm.rd[i]] += j->sm.s[i];
			for (uint32_t u = 0; u < j->sm.s[i]; ++u, ++io, ++lo) {
				o->rbeg[io] = j->rbeg[lo];
				o->qbeg[2 * io] = j->sm.qb[i]; o->qbeg[2 * io + 1] = j->sm.qe[i];
				o->score[io] = u == 0 ? j->sm.s[i] : 0;
			}
		}
		o->work.n_blk += j->work.n_blk; o->work.n_sa += j->work.n_sa;
		o->work.n_fwd_steps += j->work.n_fwd_steps; o->work.n_back_steps += j->work.n_back_steps;
		o->work.n_lf_steps += j->work.n_lf_steps;
		o->work.n_blk_fwd += j->work.n_blk_fwd; o->work.n_blk_back += j->work.n_blk_back; o->work.n_blk_lf += j->work.n_blk_lf;
		free(j->sm.k); free(j->sm.s); free(j->sm.qb); free(j->sm.qe); free(j->sm.rd); free(j->rbeg);
	}
	uint32_t acc = 0;
	for (uint32_t r = 0; r < n_reads; ++r) { o->prefix[r] = acc; acc += o->n_ref_pos[r]; }
	free(jobs); free(tid);
	return o;
}

void oracle_seeds_free(oracle_seeds_t *s)
{
	if (!s) return;
	free(s->rbeg); free(s->qbeg); free(s->score); free(s->n_ref_pos); free(s->prefix);
	free(s->smem_k); free(s->smem_s); free(s->smem_qb); free(s->smem_qe); free(s->smem_read);
	free(s);
}
