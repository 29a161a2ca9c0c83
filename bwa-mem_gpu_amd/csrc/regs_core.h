// The region tail of one read -- mem_sort_dedup_patch, mem_patch_reg, mem_mark_primary_se, mem_approx_mapq_se and the selection of
// mem_reg2sam (/root/reference/src/bwamem.c:581-680, 685-760, 1690-1717, 1721-1770) -- as ONE piece of code for the device kernels
// of csrc/regs_kernels.hip (lane per read / wave per read) and for a plain C++ build (tests/regs_core_host.cpp: the same core on
// the CPU against bmh_finalize_regs).  regs_post.cpp stays the host form (std::vector, threads); the two are checked against
// each other record for record.
//
// A region is worked on IN the 16-int record it leaves in:
//   [0] read  [1] score  [2] qb  [3] qe  [4,5] rb  [6,7] re  [8] truesc  [9] w  [10] sub  [11] sub_n  [12] secondary
//   [13] sequence id while the tail runs -> MAPQ   [14,15] hash of mem_mark_primary_se while it sorts -> flag, reported
// csub and seedcov are zero on this path (regs_post.cpp: reg_from_record) and n_comp is never read, so they have no slot.
// Sorting is klib's introsort (klib_sort.h: ties fall as in the reference) moving whole records; on the device a wave sorts
// records that sit in LDS.
#pragma once
#include <stdint.h>
#include "../../include/bwamem_hip.h"

#if defined(__HIPCC__)
#define RC_HD __host__ __device__
#else
#define RC_HD
#endif

namespace regs_core {

struct rec_t { int32_t v[16]; };

struct ctx_t {
	bmh_chain_opt_t co; bmh_ext_params_t ep; bmh_post_opt_t po;
	int64_t l_pac; const uint8_t *pac;
	int n_contigs; const int64_t *ctg_off;
	// log(k) for k = 0 .. n_log-1, computed by the HOST's libm and handed to the device: the MAPQ formula rounds a product of
	// logarithms to an integer, and the device's log() is not bit-identical to glibc's
	const double *logtab; int n_log;
	// scratch of the patch test's global alignment (qcap + 2 ints each); null: the caller cannot run it (the lane kernel) and the
	// core reports NEED_DP instead
	int32_t *dp_h, *dp_e; int dp_cap;
	// 1: the tail stops behind mem_sort_dedup_patch (interleaved pairs: mem_sam_pe takes the reads' regions from there): the records keep
	// [1..9] and the sequence id in [13], [0] = the read
	int dedup_only;
	// ALT contigs (src/bwamem.c:571-574,702,714-760,1742,1755,2323): ctg_alt[n_contigs] != 0 for a sequence that is one, NULL = no table.  With a table a region
	// carries is_alt in bit 30 of its sequence id ([13]) while the tail runs, the marking has its second round and the records come out as ALT-mode records
	// ([11] = secondary_all, [15] = reported | is_alt << 1 | alt_sc << 2: see bmh_post_opt_t)
	const uint8_t *ctg_alt;
	// 1 (with a table): [11] keeps sub_n and secondary_all + 1 rides in [13] above the MAPQ's eight bits -- the form the pairing kernel reads (csrc/pair_dev.hip:
	// it needs sub_n for the MAPQ of a hit it promotes and of the ALT hit it adds; it writes ALT-mode records itself when it is done)
	int alt_keep_sub_n;
};

enum { OK = 0, NEED_DP = 1, E_LOG = 2, E_DPCAP = 3 };

RC_HD inline int64_t r_rb(const rec_t &r) { return (int64_t)(uint32_t)r.v[4] | (int64_t)r.v[5] << 32; }
RC_HD inline int64_t r_re(const rec_t &r) { return (int64_t)(uint32_t)r.v[6] | (int64_t)r.v[7] << 32; }
RC_HD inline void r_set_rb(rec_t &r, int64_t x) { r.v[4] = (int32_t)(uint32_t)x; r.v[5] = (int32_t)(x >> 32); }
RC_HD inline int r_seq(const rec_t &r) { return r.v[13] >= 0 ? r.v[13] & 0x3FFFFFFF : r.v[13]; }      // the sequence id without the ALT bit (-1 stays -1)
RC_HD inline uint64_t r_hash(const rec_t &r) { return (uint64_t)(uint32_t)r.v[14] | (uint64_t)(uint32_t)r.v[15] << 32; }
RC_HD inline int r_alt(const rec_t &r) { return (int)((uint32_t)r.v[13] >> 30) == 1; }   // (bit 30 of a non-negative id)            // (while the tail runs; the dedup stage compares [13] as a whole: same sequence, same bit)

RC_HD inline int text_base(const uint8_t *pac, int64_t l_pac, int64_t i)
{
	const bool rev = i >= l_pac;
	const int64_t p = rev ? (l_pac << 1) - 1 - i : i;
	const int c = (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}
RC_HD inline int sc(const bmh_ext_params_t &p, int t, int q) { return (t > 3 || q > 3) ? -1 : (t == q ? p.a : -p.b); }
RC_HD inline int nt4(uint8_t c) { c &= 0xDF; return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
// query base i of the read: `ascii` reads are the device batch's letters, otherwise nt4 codes (the host arrays)
template <bool ASCII> RC_HD inline int qbase(const uint8_t *q, int i) { return ASCII ? nt4(q[i]) : (int)q[i]; }

RC_HD inline int pos2rid(const ctx_t &x, int64_t pos_f)          // bns_pos2rid, src/bntseq.c:349-363
{
	if (pos_f >= x.l_pac) return -1;
	if (x.n_contigs <= 1) return 0;
	int left = 0, mid = 0, right = x.n_contigs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= x.ctg_off[mid]) {
			if (mid == x.n_contigs - 1) break;
			if (pos_f < x.ctg_off[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

// ---- klib's introsort on records (klib_sort.h restated for host + device: fixed stack, no std::)
RC_HD inline void rswap(rec_t *a, rec_t *b)
{
	for (int k = 0; k < 16; ++k) { const int32_t t = a->v[k]; a->v[k] = b->v[k]; b->v[k] = t; }
}
template <class LT> RC_HD inline void r_insertion(rec_t *s, rec_t *t, LT lt)
{
	for (rec_t *i = s + 1; i < t; ++i)
		for (rec_t *j = i; j > s && lt(*j, *(j - 1)); --j) rswap(j, j - 1);
}
template <class LT> RC_HD inline void r_comb(int n, rec_t *a, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	bool swapped; int gap = n;
	do {
		if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = false;
		for (rec_t *i = a; i < a + n - gap; ++i) { rec_t *j = i + gap; if (lt(*j, *i)) { rswap(i, j); swapped = true; } }
	} while (swapped || gap > 2);
	if (gap != 1) r_insertion(a, a + n, lt);
}
template <int NSTK = 64, class LT> RC_HD inline void r_introsort(int n, rec_t *a, LT lt)
{
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) rswap(a, a + 1); return; }
	int d;
	for (d = 2; (1l << d) < n; ++d) ;
	rec_t *st_l[NSTK], *st_r[NSTK]; int st_d[NSTK], sp = 0;       // (ks_introsort pushes the larger side: the depth stays below log2 n)
	rec_t *s = a, *t = a + (n - 1);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { r_comb((int)(t - s + 1), s, lt); t = s; continue; }
			rec_t *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
			if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
			else k = lt(*j, *i) ? i : j;
			const rec_t rp = *k;
			if (k != t) rswap(k, t);
			for (;;) {
				do ++i; while (lt(*i, rp));
				do --j; while (i <= j && lt(rp, *j));
				if (j <= i) break;
				rswap(i, j);
			}
			rswap(i, t);
			if (i - s > t - i) {
				if (i - s > 16 && sp < NSTK) { st_l[sp] = s; st_r[sp] = i - 1; st_d[sp] = d; ++sp; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16 && sp < NSTK) { st_l[sp] = i + 1; st_r[sp] = t; st_d[sp] = d; ++sp; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (sp == 0) { r_insertion(a, a + n, lt); return; }
			--sp; s = st_l[sp]; t = st_r[sp]; d = st_d[sp];
		}
	}
}

struct lt_re { RC_HD bool operator()(const rec_t &p, const rec_t &q) const { return r_re(p) < r_re(q); } };
struct lt_score_rb_qb {
	RC_HD bool operator()(const rec_t &p, const rec_t &q) const
	{
		const int64_t prb = r_rb(p), qrb = r_rb(q);
		return p.v[1] > q.v[1] || (p.v[1] == q.v[1] && (prb < qrb || (prb == qrb && p.v[2] < q.v[2])));
	}
};
struct lt_score_hash {       // alnreg_hlt (src/bwamem.c:141): score, the primary assembly before ALT contigs, the hash
	RC_HD bool operator()(const rec_t &p, const rec_t &q) const
	{
		const int pa = r_alt(p), qa = r_alt(q);
		return p.v[1] > q.v[1] || (p.v[1] == q.v[1] && (pa < qa || (pa == qa && r_hash(p) < r_hash(q))));
	}
};
struct lt_alt_score_hash {   // alnreg_hlt2: the primary assembly first
	RC_HD bool operator()(const rec_t &p, const rec_t &q) const
	{
		const int pa = r_alt(p), qa = r_alt(q);
		return pa < qa || (pa == qa && (p.v[1] > q.v[1] || (p.v[1] == q.v[1] && r_hash(p) < r_hash(q))));
	}
};

// ---- the patch test's global alignment: ksw_global2's score under bwa_gen_cigar2's band (src/bwa.c:111-216, src/ksw.c:1120-1241),
// bases read where they are (2-bit reference, the read)
// what bwa_gen_cigar2 does with read[.., l_query) against text [rb, re) under band w_ (src/bwa.c:111-150): 0 = nothing (score 0),
// 1 = the ungapped sum (equal lengths, w_ == 0), 2 = ksw_global2 with band *w; *flip: both sequences are walked backwards (a window
// on the reverse strand)
RC_HD inline int gen_plan(const ctx_t &x, int w_, int l_query, int64_t rb, int64_t re, int *rlen, bool *flip, int *w_out)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const bmh_ext_params_t &p = x.ep;
	if (l_query <= 0 || rb >= re || (rb < x.l_pac && re > x.l_pac)) return 0;
	*rlen = (int)(re - rb);
	*flip = rb >= x.l_pac;
	if (l_query == *rlen && w_ == 0) return 1;
	int max_ins = (int)((double)(((l_query + 1) >> 1) * p.a - p.o_ins) / p.e_ins + 1.);
	int max_del = (int)((double)(((l_query + 1) >> 1) * p.a - p.o_del) / p.e_del + 1.);
	int max_gap = max_ins > max_del ? max_ins : max_del;
	max_gap = max_gap > 1 ? max_gap : 1;
	const int diff = *rlen > l_query ? *rlen - l_query : l_query - *rlen;
	int w = (max_gap + diff + 1) >> 1;
	w = w < w_ ? w : w_;
	w = w > diff + 3 ? w : diff + 3;
	*w_out = w;
	return 2;
}
template <bool ASCII>
RC_HD inline int gen_score(const ctx_t &x, int w_, int l_query, const uint8_t *query, int64_t rb, int64_t re, int *err)
{
	const bmh_ext_params_t &p = x.ep;
	const int64_t l_pac = x.l_pac;
	int rlen = 0, w = 0; bool flip = false;
	const int plan = gen_plan(x, w_, l_query, rb, re, &rlen, &flip, &w);
	if (plan == 0) return 0;
	auto tb = [&](int i) { return text_base(x.pac, l_pac, flip ? re - 1 - i : rb + i); };
	auto qb = [&](int i) { return qbase<ASCII>(query, flip ? l_query - 1 - i : i); };
	if (plan == 1) { int s = 0; for (int i = 0; i < l_query; ++i) s += sc(p, tb(i), qb(i)); return s; }
	if (!x.dp_h) { *err = NEED_DP; return 0; }
	if (l_query + 2 > x.dp_cap) { *err = E_DPCAP; return 0; }
	const int NEG = -0x40000000, oe_del = p.o_del + p.e_del, oe_ins = p.o_ins + p.e_ins, qlen = l_query;
	int32_t *Hd = x.dp_h, *E = x.dp_e;
	Hd[0] = 0; E[0] = NEG;
	for (int j = 1; j <= qlen; ++j) { Hd[j] = j <= w ? -(p.o_ins + p.e_ins * j) : NEG; E[j] = NEG; }
	for (int i = 0; i < rlen; ++i) {
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int f = NEG, left = beg == 0 ? -(p.o_del + p.e_del * (i + 1)) : NEG;
		const int ti = tb(i);
		for (int j = beg; j < end; ++j) {
			const int m = Hd[j] + sc(p, ti, qb(j));
			int e = E[j], h = m >= e ? m : e;
			Hd[j] = left;
			h = h >= f ? h : f;
			left = h;
			int y = m - oe_del; e -= p.e_del; E[j] = e > y ? e : y;
			y = m - oe_ins; f -= p.e_ins; f = f > y ? f : y;
		}
		Hd[end] = left; E[end] = NEG;
	}
	return Hd[qlen];
}

// the tests of mem_patch_reg that come before its global alignment (src/bwamem.c:586-600): false = it returns 0 without one.
// *w_out: the band the alignment would run with
RC_HD inline bool patch_pre(const ctx_t &x, const rec_t &a, const rec_t &b, int *w_out)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const int64_t arb = r_rb(a), are = r_re(a), brb = r_rb(b), bre = r_re(b);
	const int aqb = a.v[2], aqe = a.v[3], bqb = b.v[2], bqe = b.v[3];
	if (arb < x.l_pac && brb >= x.l_pac) return false;
	if (aqb >= bqb || aqe >= bqe || are >= bre) return false;
	int w = (int)((are - brb) - (aqe - bqb));
	w = w > 0 ? w : -w;
	double r = (double)(are - brb) / (bre - arb) - (double)(aqe - bqb) / (bqe - aqb);
	r = r > 0. ? r : -r;
	if (are < brb || aqe < bqb) { if (w > x.co.w << 1 || r >= 0.05f) return false; }
	else if (w > x.co.w << 2 || r >= 0.05f * 2) return false;
	w += a.v[9] + b.v[9];
	w = w < x.co.w << 2 ? w : x.co.w << 2;
	*w_out = w;
	return true;
}
// mem_patch_reg's verdict on the alignment across both regions (src/bwamem.c:606-612): its score, or 0
RC_HD inline int patch_accept(const rec_t &a, const rec_t &b, int score)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const int64_t arb = r_rb(a), are = r_re(a), brb = r_rb(b), bre = r_re(b);
	const int aqb = a.v[2], aqe = a.v[3], bqb = b.v[2], bqe = b.v[3];
	const int q_s = (int)((double)(bqe - aqb) / ((bqe - bqb) + (aqe - aqb)) * (b.v[1] + a.v[1]) + .499);
	const int r_s = (int)((double)(bre - arb) / ((bre - brb) + (are - arb)) * (b.v[1] + a.v[1]) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
	return score;
}
template <bool ASCII>
RC_HD inline int patch_reg(const ctx_t &x, const uint8_t *query, const rec_t &a, const rec_t &b, int *w_out, int *err)        // mem_patch_reg
{
	int w;
	if (!patch_pre(x, a, b, &w)) return 0;
	const int score = gen_score<ASCII>(x, w, b.v[3] - a.v[2], query + a.v[2], r_rb(a), r_re(b), err);
	if (*err) return 0;
	*w_out = w;
	return patch_accept(a, b, score);
}

// does region i of the lt_re-sorted list have a predecessor close enough to be looked at at all?  (the test reads only fields the
// loop never changes before it reaches i, so it can be evaluated for all i beforehand)
RC_HD inline bool dedup_near(const ctx_t &x, const rec_t *a, int i)
{
	return i >= 1 && a[i].v[13] == a[i - 1].v[13] && r_rb(a[i]) < r_re(a[i - 1]) + x.co.max_chain_gap;
}
// iteration i of mem_sort_dedup_patch's scan (src/bwamem.c:631-662) looks at the regions j = i-1, i-2, ... while dedup_in_window
// holds; dedup_redundant: the pair overlaps by mask_level_redun on both the reference and the query (then the lower-scoring one goes)
RC_HD inline bool dedup_in_window(const ctx_t &x, const rec_t &p, const rec_t &q) { return p.v[13] == q.v[13] && r_rb(p) < r_re(q) + x.co.max_chain_gap; }
RC_HD inline bool dedup_redundant(const ctx_t &x, const rec_t &p, const rec_t &q)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const int64_t prb = r_rb(p), pre = r_re(p), qrb = r_rb(q), qre = r_re(q);
	const int64_t pr = qre - prb;
	const int64_t pq = q.v[2] < p.v[2] ? q.v[3] - p.v[2] : p.v[3] - q.v[2];
	const int64_t mr = qre - qrb < pre - prb ? qre - qrb : pre - prb;
	const int64_t mq = q.v[3] - q.v[2] < p.v[3] - p.v[2] ? q.v[3] - q.v[2] : p.v[3] - p.v[2];
	return pr > x.po.mask_level_redun * mr && pq > x.po.mask_level_redun * mq;
}
// region q is merged into region p (src/bwamem.c:650-658)
RC_HD inline void dedup_patch_apply(rec_t *p, rec_t *q, int score, int w)
{
	p->v[10] = p->v[10] > q->v[10] ? p->v[10] : q->v[10];
	p->v[2] = q->v[2]; r_set_rb(*p, r_rb(*q));
	p->v[8] = p->v[1] = score;
	p->v[9] = w;
	q->v[2] = q->v[3];
}
// one pair (i, j) of the scan; *stop: region i has been emptied, its scan ends.  Returns 0 or an error
template <bool ASCII>
RC_HD inline int dedup_pair(const ctx_t &x, const uint8_t *query, rec_t *p, rec_t *q, bool *stop)
{
	*stop = false;
	if (q->v[3] == q->v[2]) return 0;
	int score, w, err = 0;
	if (dedup_redundant(x, *p, *q)) {
		if (p->v[1] < q->v[1]) { p->v[3] = p->v[2]; *stop = true; }
		else q->v[3] = q->v[2];
	} else if (r_rb(*q) < r_rb(*p) && (score = patch_reg<ASCII>(x, query, *q, *p, &w, &err)) > 0) dedup_patch_apply(p, q, score, w);
	return err;
}
template <bool ASCII>
RC_HD inline int dedup_one(const ctx_t &x, const uint8_t *query, int i, rec_t *a)
{
	rec_t *p = &a[i];
	for (int j = i - 1; j >= 0 && dedup_in_window(x, *p, a[j]); --j) {
		bool stop;
		const int err = dedup_pair<ASCII>(x, query, p, &a[j], &stop);
		if (err) return err;
		if (stop) break;
	}
	return 0;
}
template <bool ASCII>
RC_HD inline int dedup_loop(const ctx_t &x, const uint8_t *query, int n, rec_t *a)
{
	for (int i = 1; i < n; ++i) {
		if (!dedup_near(x, a, i)) continue;
		const int err = dedup_one<ASCII>(x, query, i, a);
		if (err) return -err;
	}
	int m = 0;
	for (int i = 0; i < n; ++i) if (a[i].v[3] > a[i].v[2]) { if (m != i) a[m] = a[i]; ++m; }
	return m;
}
RC_HD inline bool dedup_same(const rec_t *a, int i) { return a[i].v[1] == a[i - 1].v[1] && r_rb(a[i]) == r_rb(a[i - 1]) && a[i].v[2] == a[i - 1].v[2]; }
RC_HD inline int dedup_equal(int n, rec_t *a)
{
	for (int i = 1; i < n; ++i)
		if (dedup_same(a, i)) a[i].v[3] = a[i].v[2];
	int m = n ? 1 : 0;
	for (int i = 1; i < n; ++i) if (a[i].v[3] > a[i].v[2]) { if (m != i) a[m] = a[i]; ++m; }
	return m;
}
// mem_sort_dedup_patch on a[0..n): returns the number of regions left (at the front of a), < 0: -error
template <bool ASCII, int NSTK = 64>
RC_HD inline int sort_dedup_patch(const ctx_t &x, const uint8_t *query, int n, rec_t *a)
{
	if (n <= 1) return n;
	r_introsort<NSTK>(n, a, lt_re());
	n = dedup_loop<ASCII>(x, query, n, a);
	if (n < 0) return n;
	r_introsort<NSTK>(n, a, lt_score_rb_qb());
	return dedup_equal(n, a);
}

RC_HD inline uint64_t hash64(uint64_t key)
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

// mem_mark_primary_se in its steps: mark_init (sub, secondary, the tie-break hash of every region) -> sorted by lt_score_hash ->
// mark_loop (z: n ints of scratch, the primaries found so far)
RC_HD inline void mark_init_one(rec_t &r, int64_t id, int i)
{
	const uint64_t h = hash64((uint64_t)(id + i));
	r.v[10] = 0; r.v[12] = -1; r.v[14] = (int32_t)(uint32_t)h; r.v[15] = (int32_t)(uint32_t)(h >> 32);
}
RC_HD inline int mark_tmp(const ctx_t &x)
{
	int tmp = x.ep.a + x.ep.b;
	tmp = x.ep.o_del + x.ep.e_del > tmp ? x.ep.o_del + x.ep.e_del : tmp;
	tmp = x.ep.o_ins + x.ep.e_ins > tmp ? x.ep.o_ins + x.ep.e_ins : tmp;
	return tmp;
}
// do regions i and j overlap on the query by mask_level of the shorter one? (src/bwamem.c:731-736)
RC_HD inline bool mark_overlap(const ctx_t &x, const rec_t &aj, const rec_t &ai)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const int b_max = aj.v[2] > ai.v[2] ? aj.v[2] : ai.v[2];
	const int e_min = aj.v[3] < ai.v[3] ? aj.v[3] : ai.v[3];
	if (e_min > b_max) {
		const int li = ai.v[3] - ai.v[2], lj = aj.v[3] - aj.v[2];
		const int min_l = li < lj ? li : lj;
		if (e_min - b_max >= min_l * x.co.mask_level) return true;
	}
	return false;
}
RC_HD inline void mark_loop(const ctx_t &x, int n, rec_t *a, int32_t *z)
{
	if (n == 0) return;
	const int tmp = mark_tmp(x);
	int nz = 0;
	z[nz++] = 0;
	for (int i = 1; i < n; ++i) {
		int k;
		for (k = 0; k < nz; ++k) {
			const int j = z[k];
			if (mark_overlap(x, a[j], a[i])) {
				if (a[j].v[10] == 0) a[j].v[10] = a[i].v[1];
				if (a[j].v[1] - a[i].v[1] <= tmp && (r_alt(a[j]) || !r_alt(a[i]))) ++a[j].v[11];
				break;
			}
		}
		if (k == nz) z[nz++] = i;
		else a[i].v[12] = z[k];
	}
}
// the second round of mem_mark_primary_se (src/bwamem.c:727-757) once the first one has run on a[0..n): the rank of the first round is kept ([0]: secondary_all),
// the hits of the primary assembly come first and are marked again among themselves (sub_n adds up over both rounds, as there).  Without ALT hits in the
// read [0] = [12].  z: n ints.
template <int NSTK = 64>
RC_HD inline void mark_second_round(const ctx_t &x, int n, rec_t *a, int32_t *z)
{
	int n_pri = 0;
	for (int i = 0; i < n; ++i) n_pri += !r_alt(a[i]);
	if (n_pri == n) { for (int i = 0; i < n; ++i) a[i].v[0] = a[i].v[12]; return; }
	for (int i = 0; i < n; ++i) a[i].v[0] = i;
	if (n_pri > 0) r_introsort<NSTK>(n, a, lt_alt_score_hash());
	for (int i = 0; i < n; ++i) z[a[i].v[0]] = i;
	for (int i = 0; i < n; ++i) {
		if (a[i].v[12] >= 0) { a[i].v[0] = z[a[i].v[12]]; if (r_alt(a[i])) a[i].v[12] = 0x7FFFFFFF; }
		else a[i].v[0] = -1;
	}
	if (n_pri > 0) {
		for (int i = 0; i < n_pri; ++i) { a[i].v[10] = 0; a[i].v[12] = -1; }
		mark_loop(x, n_pri, a, z);
	}
}
template <int NSTK = 64>
RC_HD inline void mark_primary(const ctx_t &x, int n, rec_t *a, int64_t id, int32_t *z)
{
	if (n == 0) return;
	for (int i = 0; i < n; ++i) mark_init_one(a[i], id, i);
	r_introsort<NSTK>(n, a, lt_score_hash());
	mark_loop(x, n, a, z);
	if (x.ctg_alt) mark_second_round<NSTK>(x, n, a, z);
}

// mem_approx_mapq_se, mapQ_coef_len > 0 form; logarithms of integers from the host's table
RC_HD inline int approx_mapq(const ctx_t &x, const rec_t &a, float frac_rep, int *err)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	int mapq, sub = a.v[10] ? a.v[10] : x.co.min_seed_len * x.ep.a;
	const int score = a.v[1];
	if (sub >= score) return 0;
	const int lq = a.v[3] - a.v[2], lr = (int)(r_re(a) - r_rb(a));
	const int l = lq > lr ? lq : lr;
	const double identity = 1. - (double)(l * x.ep.a - score) / (x.ep.a + x.ep.b) / l;
	if (score == 0) mapq = 0;
	else {
		if (l >= x.n_log || l < 1) { *err = E_LOG; return 0; }
		double tmp = l < x.po.mapQ_coef_len ? 1. : x.po.mapQ_coef_fac / x.logtab[l];
		tmp *= identity * identity;
		mapq = (int)(6.02 * (score - sub) / x.ep.a * tmp * tmp + .499);
	}
	if (a.v[11] > 0) {
		if (a.v[11] + 1 >= x.n_log) { *err = E_LOG; return 0; }
		mapq -= (int)(4.343 * x.logtab[a.v[11] + 1] + .499);
	}
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - frac_rep) + .499);
	return mapq;
}

// first and last step of a read's tail: init_one fills [8..15] of a record that arrives with [0..7] (the merge kernel's layout);
// emit_all turns the marked regions into the output records of bmh_finalize_regs (MAPQ, flags, which are reported)
RC_HD inline void init_one(const ctx_t &x, rec_t &p)
{
	const int64_t rb = r_rb(p), re = r_re(p);
	p.v[8] = p.v[1]; p.v[9] = x.co.w; p.v[10] = 0; p.v[11] = 0; p.v[12] = -1;
	p.v[13] = pos2rid(x, rb < x.l_pac ? rb : (x.l_pac << 1) - 1 - (re - 1));
	if (x.ctg_alt && p.v[13] >= 0 && x.ctg_alt[p.v[13]]) p.v[13] |= 1 << 30;
	p.v[14] = p.v[15] = 0;
}
// what a region's record says by itself (MAPQ before the cap of the supplementary records, secondary flag, reported or not)
RC_HD inline void emit_one(const ctx_t &x, float frac_rep, const rec_t *a, int k, int *mapq, int *flag, int *rep, int *err)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const rec_t &p = a[k];
	*mapq = p.v[12] < 0 ? approx_mapq(x, p, frac_rep, err) : 0; *flag = p.v[12] >= 0 ? 0x100 : 0; *rep = 1;
	if (p.v[1] < x.po.T) *rep = 0;
	else if (p.v[12] >= 0 && (r_alt(p) || !x.po.flag_all)) *rep = 0;                                  // src/bwamem.c:1742
	else if (p.v[12] >= 0 && p.v[12] < 0x7FFFFFFF && p.v[1] < a[p.v[12]].v[1] * x.co.drop_ratio) *rep = 0;
}
// ALT mode: the score of the ALT hit that shadows region k in the first marking round (src/bwamem.c:724), from [0] = secondary_all; before any record is emitted
RC_HD inline int alt_score_of(const rec_t *a, int k)
{
	const rec_t &p = a[k];
	return (!r_alt(p) && p.v[0] >= 0 && r_alt(a[p.v[0]])) ? a[p.v[0]].v[1] : 0;
}
// z: n ints of scratch (ALT mode)
RC_HD inline int emit_all(const ctx_t &x, uint32_t read, float frac_rep, int n, rec_t *a, int32_t *z = nullptr)
{
	int l = 0, mapq0 = 0, err = 0;
	const bool altm = x.ctg_alt != nullptr;
	if (altm) for (int k = 0; k < n; ++k) z[k] = alt_score_of(a, k);              // (is_alt of another record is gone once that record is emitted)
	for (int k = 0; k < n; ++k) {
		rec_t &p = a[k];
		int mapq, flag, rep;
		emit_one(x, frac_rep, a, k, &mapq, &flag, &rep, &err);
		if (err) return -err;
		const int alt = r_alt(p);
		if (rep) {
			if (l && p.v[12] < 0) flag |= x.po.no_multi ? 0x10000 : 0x800;     // src/bwamem.c:1754
			if (l && !alt && mapq > mapq0) mapq = mapq0;                        // :1755
			if (l == 0) mapq0 = mapq;
			++l;
		}
		if (altm) { if (!x.alt_keep_sub_n) p.v[11] = p.v[0]; else mapq |= (p.v[0] + 1) << 8; rep |= alt << 1 | (z[k] > 0 ? z[k] << 2 : 0); }
		p.v[0] = (int32_t)read; p.v[13] = mapq; p.v[14] = flag; p.v[15] = rep;
	}
	return n;
}
// The tail of one read, serially: a[0..n_in) = records with [0..7] filled, [8..15] anything.  On return a[0..n) are the output
// records of bmh_finalize_regs; returns n, or -error.  z: n_in ints of scratch.  NSTK: depth of the sort's stack (1 is enough for
// up to 17 records: ks_introsort pushes only sub-ranges of more than 16)
template <bool ASCII, int NSTK = 64>
RC_HD inline int finalize_read(const ctx_t &x, const uint8_t *query, uint32_t read, int64_t id, float frac_rep, int n_in, rec_t *a, int32_t *z)
{
	for (int i = 0; i < n_in; ++i) init_one(x, a[i]);
	int n = sort_dedup_patch<ASCII, NSTK>(x, query, n_in, a);
	if (n < 0) return n;
	if (x.dedup_only) { for (int i = 0; i < n; ++i) { a[i].v[0] = (int32_t)read; a[i].v[13] = r_seq(a[i]); } return n; }
	mark_primary<NSTK>(x, n, a, id, z);
	return emit_all(x, read, frac_rep, n, a, z);
}

} // namespace regs_core
