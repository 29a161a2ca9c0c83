/*
 * oracle/ksw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see fmd_oracle.h).
 *
 * CPU restatement of the extension half of the hot path:
 *   ksw_extend2            /root/reference/src/ksw.c:864-986 (opt_ext == 0 path)
 *   scoring matrix         src/bwa.c:99-108 (bwa_fill_scmat)
 *   local vs to-end rule   src/bwamem.c:1893-1901 (decoy_cpu_align)
 * The row state is kept as two int arrays H (= H(i-1,j-1) slot) and E.
 */
#include "fmd_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* optional per-row trace of the trimmed range [beg,end) (analysis of the band geometry, scripts/band_probe.py) */
static __thread int16_t *g_trace; static __thread uint64_t g_ntrace, g_trace_cap;

static inline int sc(const ksw_params_t *p, int t, int q)
{
	if (t > 3 || q > 3) return -p->n_penalty;
	return t == q ? p->a : -p->b;
}

int oracle_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                       const ksw_params_t *p, int h0, int *qle, int *tle, int *gtle,
                       int *gscore_, int *max_off_, uint64_t *cells)
{
	int oe_del = p->o_del + p->e_del, oe_ins = p->o_ins + p->e_ins;
	int *H = (int *)calloc(qlen + 2, sizeof(int)), *E = (int *)calloc(qlen + 2, sizeof(int));
	int i, j, beg = 0, end = qlen, max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	uint64_t nc = 0;
	/* row -1: H(-1,j-1) slots */
	H[0] = h0;
	if (qlen >= 1) H[1] = h0 > oe_ins ? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && H[j - 1] > p->e_ins; ++j) H[j] = H[j - 1] - p->e_ins;
	for (i = 0; i < tlen; ++i) {
		int f = 0, h1, m = 0, mj = -1, ti = target[i];
		if (beg == 0) { h1 = h0 - (p->o_del + p->e_del * (i + 1)); if (h1 < 0) h1 = 0; }
		else h1 = 0;
		for (j = beg; j < end; ++j) {
			int M = H[j], e = E[j], h, t;
			H[j] = h1;
			M = M ? M + sc(p, ti, query[j]) : 0;
			h = M > e ? M : e;
			h = h > f ? h : f;
			h1 = h;
			if (!(m > h)) mj = j;          /* last column holding the row maximum */
			m = m > h ? m : h;
			t = M - oe_del; if (t < 0) t = 0;
			e -= p->e_del; if (e < t) e = t;
			E[j] = e;
			t = M - oe_ins; if (t < 0) t = 0;
			f -= p->e_ins; if (f < t) f = t;
		}
		nc += (uint64_t)(end > beg ? end - beg : 0);
		int16_t *tr = 0;
		if (g_trace && g_ntrace + 5 <= g_trace_cap) { tr = g_trace + g_ntrace; tr[0] = (int16_t)beg; tr[1] = (int16_t)end; tr[2] = 0; tr[3] = (int16_t)max; tr[4] = (int16_t)gscore; g_ntrace += 5; }
		H[end] = h1; E[end] = 0;
		if (j == qlen) {
			if (!(gscore > h1)) max_ie = i;
			if (gscore < h1) gscore = h1;
		}
		if (m == 0) break;
		if (m > max) {
			int d = mj - i; if (d < 0) d = -d;
			max = m; max_i = i; max_j = mj;
			if (d > max_off) max_off = d;
		} else if (p->zdrop > 0) {
			if (i - max_i > mj - max_j) {
				if (max - m - ((i - max_i) - (mj - max_j)) * p->e_del > p->zdrop) break;
			} else {
				if (max - m - ((mj - max_j) - (i - max_i)) * p->e_ins > p->zdrop) break;
			}
		}
		for (j = beg; j < end && H[j] == 0 && E[j] == 0; ++j) ;
		beg = j;
		for (j = end; j >= beg && H[j] == 0 && E[j] == 0; --j) ;
		end = j + 2 < qlen ? j + 2 : qlen;
		if (tr) {   /* the potential bound of the HIP kernels after this row: U = max over the NON-ZERO frontier cells of
		             * H + a * min(columns left, rows left) (a zero cell starts nothing: M == 0 stays 0), plus the first-column value */
			int phi = 0, hn = beg == 0 ? h0 - (p->o_del + p->e_del * (i + 1)) : 0, rl = tlen - 1 - i;
			if (hn > 0) { int g = qlen < rl ? qlen : rl; phi = hn + p->a * g; }
			for (j = 1; j <= qlen; ++j) if (H[j]) { int cl = qlen - 1 - (j - 1), g = cl < rl ? cl : rl, v = H[j] + p->a * g; if (v > phi) phi = v; }
			tr[2] = (int16_t)phi; tr[3] = (int16_t)max; tr[4] = (int16_t)gscore;
		}
	}
	free(H); free(E);
	if (qle) *qle = max_j + 1;
	if (tle) *tle = max_i + 1;
	if (gtle) *gtle = max_ie + 1;
	if (gscore_) *gscore_ = gscore;
	if (max_off_) *max_off_ = max_off;
	if (cells) *cells += nc;
	return max;
}

typedef struct {
	uint32_t i0, i1;
	const uint8_t *q, *t; const uint32_t *qoff, *qlen, *toff, *tlen, *h0;
	const ksw_params_t *p; int32_t *out3, *raw6; uint64_t cells;
} ext_job_t;

static void *ext_worker(void *arg)
{
	ext_job_t *j = (ext_job_t *)arg;
	for (uint32_t i = j->i0; i < j->i1; ++i) {
		int qle, tle, gtle, gscore, max_off;
		int score = oracle_ksw_extend2((int)j->qlen[i], j->q + j->qoff[i], (int)j->tlen[i], j->t + j->toff[i],
		                               j->p, (int)j->h0[i], &qle, &tle, &gtle, &gscore, &max_off, &j->cells);
		if (j->raw6) {
			int32_t *r = j->raw6 + 6 * (uint64_t)i;
			r[0] = score; r[1] = qle; r[2] = tle; r[3] = gtle; r[4] = gscore; r[5] = max_off;
		}
		int32_t *o = j->out3 + 3 * (uint64_t)i;
		if (gscore <= 0 || gscore <= score - j->p->end_bonus) { o[0] = score; o[1] = qle; o[2] = tle; }
		else { o[0] = gscore; o[1] = (int32_t)j->qlen[i]; o[2] = gtle; }
	}
	return 0;
}

uint64_t oracle_extend_batch(uint32_t n, const uint8_t *q, const uint32_t *qoff, const uint32_t *qlen,
                             const uint8_t *t, const uint32_t *toff, const uint32_t *tlen,
                             const uint32_t *h0, const ksw_params_t *p, int32_t *out3, int32_t *raw6,
                             int n_threads)
{
	if (n_threads < 1) n_threads = 1;
	if ((uint32_t)n_threads > n && n) n_threads = (int)n;
	ext_job_t *jobs = (ext_job_t *)calloc(n_threads, sizeof(ext_job_t));
	pthread_t *tid = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
	for (int k = 0; k < n_threads; ++k) {
		ext_job_t *j = &jobs[k];
		j->i0 = (uint32_t)((uint64_t)n * k / n_threads); j->i1 = (uint32_t)((uint64_t)n * (k + 1) / n_threads);
		j->q = q; j->t = t; j->qoff = qoff; j->qlen = qlen; j->toff = toff; j->tlen = tlen; j->h0 = h0;
		j->p = p; j->out3 = out3; j->raw6 = raw6;
		if (n_threads > 1) pthread_create(&tid[k], 0, ext_worker, j);
		else ext_worker(j);
	}
	uint64_t cells = 0;
	for (int k = 0; k < n_threads; ++k) {
		if (n_threads > 1) pthread_join(tid[k], 0);
		cells += jobs[k].cells;
	}
	free(jobs); free(tid);
	return cells;
}

/* per-row [beg,end) of every job, rows[i] = rows executed (incl. the row that ends with m == 0); trace = (beg, end, potential bound U after the row, max, gscore) per row, in job order.
 * Returns the number of int16 entries written (stops recording, not computing, at cap). */
uint64_t oracle_extend_trace(uint32_t n, const uint8_t *q, const uint32_t *qoff, const uint32_t *qlen,
                             const uint8_t *t, const uint32_t *toff, const uint32_t *tlen,
                             const uint32_t *h0, const ksw_params_t *p, uint32_t *rows, int16_t *trace, uint64_t cap)
{
	g_trace = trace; g_ntrace = 0; g_trace_cap = cap;
	for (uint32_t i = 0; i < n; ++i) {
		uint64_t before = g_ntrace, cells = 0;
		int qle, tle, gtle, gscore, max_off;
		oracle_ksw_extend2((int)qlen[i], q + qoff[i], (int)tlen[i], t + toff[i], p, (int)h0[i], &qle, &tle, &gtle, &gscore, &max_off, &cells);
		rows[i] = (uint32_t)((g_ntrace - before) / 5);
	}
	g_trace = 0;
	return g_ntrace;
}
