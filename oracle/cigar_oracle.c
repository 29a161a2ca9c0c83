/*
 * oracle/cigar_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see fmd_oracle.h).
 *
 * CPU restatement of the step right after the hot path: region -> CIGAR / NM / MD.
 *   ksw_global2      /root/reference/src/ksw.c:1120-1241   banded global alignment with a 6-bit direction matrix
 *   bwa_gen_cigar2   src/bwa.c:111-216                     band width, reverse-strand flip, NM and MD
 *   mem_reg2aln      src/bwamem.c:2344-2440 (+ infer_bw :1486-1494)  retry with doubled band, squeeze of a leading /
 *                                                          trailing deletion, soft clips, position
 * Row state as in ksw_oracle.c: Hd[j] = H(i-1,j-1), E[j] = E(i,j).
 */
#include "fmd_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-0x40000000)

static inline int gsc(const ksw_params_t *p, int t, int q)
{
	if (t > 3 || q > 3) return -p->n_penalty;
	return t == q ? p->a : -p->b;
}

/* returns the global score; cigar (capacity cap) receives n_cigar ops, len << 4 | op with op 0 = M, 1 = I, 2 = D;
 * *n_cigar = number of ops needed (may exceed cap: then only the first cap are stored) */
int oracle_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const ksw_params_t *p, int w,
                       int *n_cigar, uint32_t *cigar, int cap)
{
	const int oe_del = p->o_del + p->e_del, oe_ins = p->o_ins + p->e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	int *Hd = (int *)malloc(sizeof(int) * (qlen + 2)), *E = (int *)malloc(sizeof(int) * (qlen + 2));
	uint8_t *dir = (uint8_t *)malloc((size_t)(n_col > 0 ? n_col : 1) * (tlen > 0 ? tlen : 1));
	int i, j;
	Hd[0] = 0; E[0] = NEG_INF;
	for (j = 1; j <= qlen; ++j) { Hd[j] = j <= w ? -(p->o_ins + p->e_ins * j) : NEG_INF; E[j] = NEG_INF; }
	for (i = 0; i < tlen; ++i) {
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int f = NEG_INF, left = beg == 0 ? -(p->o_del + p->e_del * (i + 1)) : NEG_INF;
		uint8_t *row = dir + (size_t)i * n_col;
		for (j = beg; j < end; ++j) {
			int m = Hd[j] + gsc(p, target[i], query[j]), e = E[j], h, t;
			uint8_t d;
			Hd[j] = left;
			if (m >= e) { d = 0; h = m; } else { d = 1; h = e; }
			if (!(h >= f)) { d = 2; h = f; }
			left = h;
			t = m - oe_del; e -= p->e_del;
			if (e > t) d |= 1 << 2; else e = t;
			E[j] = e;
			t = m - oe_ins; f -= p->e_ins;
			if (f > t) d |= 2 << 4; else f = t;
			row[j - beg] = d;
		}
		Hd[end] = left; E[end] = NEG_INF;
	}
	const int score = Hd[qlen];
	if (n_cigar) {
		/* walk back from the last cell; state 0 = in H (read bits 0-1), 1 = in E (bits 2-3), 2 = in F (bits 4-5) */
		int n = 0, cap_rev = qlen + tlen + 2, state = 0, k;
		uint32_t *rev = (uint32_t *)malloc(sizeof(uint32_t) * cap_rev);
		i = tlen - 1;
		k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
#define PUSH(op, len) do { if (n && (rev[n - 1] & 0xf) == (uint32_t)(op)) rev[n - 1] += (uint32_t)(len) << 4; else rev[n++] = (uint32_t)(len) << 4 | (op); } while (0)
		while (i >= 0 && k >= 0) {
			const int beg = i > w ? i - w : 0;
			state = dir[(size_t)i * n_col + (k - beg)] >> (state << 1) & 3;
			if (state == 0) { PUSH(0, 1); --i; --k; }
			else if (state == 1) { PUSH(2, 1); --i; }
			else { PUSH(1, 1); --k; }
		}
		if (i >= 0) PUSH(2, i + 1);
		if (k >= 0) PUSH(1, k + 1);
#undef PUSH
		*n_cigar = n;
		for (j = 0; j < n && j < cap; ++j) cigar[j] = rev[n - 1 - j];
		free(rev);
	}
	free(Hd); free(E); free(dir);
	return score;
}

static int text_base(const uint8_t *pac, int64_t l_pac, int64_t i)
{
	const int rev = i >= l_pac;
	const int64_t p = rev ? (l_pac << 1) - 1 - i : i;
	const int c = (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}

/* bwa_gen_cigar2: query = the aligned part of the read (nt4 codes), [rb, re) in the fwd.revcomp text.
 * md (capacity md_cap) receives the MD string (NUL terminated).  Returns 0, or -1 for a rejected interval
 * (empty, or bridging the two strands): then *n_cigar = 0 and *NM = -1. */
int oracle_gen_cigar2(const ksw_params_t *p, int w_, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query_,
                      int64_t rb, int64_t re, int *score, int *n_cigar, uint32_t *cigar, int cap, int *NM, char *md, int md_cap)
{
	*n_cigar = 0; *NM = -1; *score = 0; if (md_cap) md[0] = 0;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return -1;
	const int rlen = (int)(re - rb);
	uint8_t *rseq = (uint8_t *)malloc(rlen), *query = (uint8_t *)malloc(l_query);
	int i;
	const int flip = rb >= l_pac;                 /* reverse strand: align the reversed sequences so that gaps go leftmost */
	for (i = 0; i < rlen; ++i) rseq[i] = (uint8_t)text_base(pac, l_pac, flip ? re - 1 - i : rb + i);
	for (i = 0; i < l_query; ++i) query[i] = query_[flip ? l_query - 1 - i : i];
	if (l_query == rlen && w_ == 0) {
		cigar[0] = (uint32_t)l_query << 4; *n_cigar = 1;
		for (i = 0; i < l_query; ++i) *score += gsc(p, rseq[i], query[i]);
	} else {
		int max_ins = (int)((double)(((l_query + 1) >> 1) * p->a - p->o_ins) / p->e_ins + 1.);
		int max_del = (int)((double)(((l_query + 1) >> 1) * p->a - p->o_del) / p->e_del + 1.);
		int max_gap = max_ins > max_del ? max_ins : max_del, w, min_w, diff = rlen > l_query ? rlen - l_query : l_query - rlen;
		if (max_gap < 1) max_gap = 1;
		w = (max_gap + diff + 1) >> 1;
		if (w > w_) w = w_;
		min_w = diff + 3;
		if (w < min_w) w = min_w;
		*score = oracle_ksw_global2(l_query, query, rlen, rseq, p, w, n_cigar, cigar, cap);
	}
	{	/* NM and MD along the CIGAR */
		const char *b2c = rb < l_pac ? "ACGTN" : "TGCAN";
		int k, x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, l = 0;
#define PUTC(c) do { if (l + 1 < md_cap) md[l] = (c); ++l; } while (0)
#define PUTW(v) do { char tmp_[16]; int n_ = snprintf(tmp_, sizeof(tmp_), "%d", (v)); for (int z_ = 0; z_ < n_; ++z_) PUTC(tmp_[z_]); } while (0)
		for (k = 0; k < *n_cigar && k < cap; ++k) {
			const int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
			if (op == 0) {
				for (i = 0; i < len; ++i) {
					if (query[x + i] != rseq[y + i]) { PUTW(u); PUTC(b2c[rseq[y + i]]); ++n_mm; u = 0; }
					else ++u;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < *n_cigar - 1) {
					PUTW(u); PUTC('^');
					for (i = 0; i < len; ++i) PUTC(b2c[rseq[y + i]]);
					u = 0; n_gap += len;
				}
				y += len;
			} else { x += len; n_gap += len; }
		}
		PUTW(u);
		if (md_cap) md[l < md_cap ? l : md_cap - 1] = 0;
#undef PUTC
#undef PUTW
		*NM = n_mm + n_gap;
	}
	free(rseq); free(query);
	return 0;
}

static int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	int w, d = l1 > l2 ? l1 - l2 : l2 - l1;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	return w < d ? d : w;
}

/* mem_reg2aln for a mapped region: read = whole read (nt4 codes); region {qb, qe, rb, re, truesc, w = ar->w};
 * opt_w = opt->w.  Out: pos = 0-based position on the forward strand of the whole concatenated reference (before the
 * contig offset is subtracted), is_rev, the final CIGAR incl. soft clips (op 3), NM, MD, global score. */
int oracle_reg2aln(const ksw_params_t *p, int opt_w, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *read,
                   int qb, int qe, int64_t rb, int64_t re, int truesc, int reg_w,
                   int64_t *pos, int *is_rev, int *n_cigar, uint32_t *cigar, int cap, int *NM, char *md, int md_cap, int *score)
{
	int tmp = infer_bw(qe - qb, (int)(re - rb), truesc, p->a, p->o_del, p->e_del);
	int w2 = infer_bw(qe - qb, (int)(re - rb), truesc, p->a, p->o_ins, p->e_ins);
	int i = 0, last_sc = -(1 << 30), n = 0;
	if (w2 < tmp) w2 = tmp;
	if (w2 > opt_w) w2 = w2 < reg_w ? w2 : reg_w;
	do {
		if (w2 > opt_w << 2) w2 = opt_w << 2;
		oracle_gen_cigar2(p, w2, l_pac, pac, qe - qb, read + qb, rb, re, score, &n, cigar, cap - 2, NM, md, md_cap);
		if (*score == last_sc || w2 == opt_w << 2) break;
		last_sc = *score;
		w2 <<= 1;
	} while (++i < 3 && *score < truesc - p->a);
	{
		const int64_t x = rb < l_pac ? rb : re - 1;
		*is_rev = x >= l_pac;
		*pos = *is_rev ? (l_pac << 1) - 1 - x : x;
	}
	if (n > cap - 2) n = cap - 2;
	if (n > 0) {          /* squeeze out a leading or trailing deletion (the MD string never holds them) */
		if ((cigar[0] & 0xf) == 2) { *pos += cigar[0] >> 4; --n; memmove(cigar, cigar + 1, 4 * (size_t)n); }
		else if ((cigar[n - 1] & 0xf) == 2) --n;
	}
	if (qb != 0 || qe != l_query) {
		const int clip5 = *is_rev ? l_query - qe : qb, clip3 = *is_rev ? qb : l_query - qe;
		if (clip5) { memmove(cigar + 1, cigar, 4 * (size_t)n); cigar[0] = (uint32_t)clip5 << 4 | 3; ++n; }
		if (clip3) cigar[n++] = (uint32_t)clip3 << 4 | 3;
	}
	*n_cigar = n;
	return 0;
}
