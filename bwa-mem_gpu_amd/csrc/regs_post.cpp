// Host restatement of what the reference does with the regions of a read after the extension results are merged
// (the tail of rank 1 and the alignment-selection part of rank 4):
//   mem_sort_dedup_patch   /root/reference/src/bwamem.c:620-680  (+ mem_patch_reg :581-618: merge of two colinear regions
//                          when a global alignment across both scores well enough)
//   mem_mark_primary_se    :685-760   primary / secondary marking, sub-optimal score and count (hash_64 tie-break, utils.h:126)
//   mem_approx_mapq_se     :1690-1717
//   mem_reg2sam            :1721-1770 which regions become SAM records, supplementary flag, MAPQ cap
// Global alignment score for the patch test: ksw_global2 without traceback under bwa_gen_cigar2's band (src/bwa.c:111-216).
// Sorting goes through klib_sort.h so that ties fall as in the reference.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "bmh_internal.h"
#include "klib_sort.h"
#include "regs_post.h"

namespace rp {

// BMH_POST_STATS: how often the patch test reaches its global alignment, and how large those are
static std::atomic<unsigned long long> g_st_patch_calls{0}, g_st_dp_calls{0}, g_st_dp_cells{0};
static const bool g_post_stats = getenv("BMH_POST_STATS") != nullptr;      // the counters are only read (and only bumped) with it set

int text_base(const uint8_t *pac, int64_t l_pac, int64_t i)
{
	const bool rev = i >= l_pac;
	const int64_t p = rev ? (l_pac << 1) - 1 - i : i;
	const int c = (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}
static inline int sc(const bmh_ext_params_t &p, int t, int q) { return (t > 3 || q > 3) ? -1 : (t == q ? p.a : -p.b); }

// score of ksw_global2 (no traceback)
static int global_score(const bmh_ext_params_t &p, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w)
{
	const int NEG = -0x40000000, oe_del = p.o_del + p.e_del, oe_ins = p.o_ins + p.e_ins;
	std::vector<int> Hd(qlen + 2), E(qlen + 2);
	Hd[0] = 0; E[0] = NEG;
	for (int j = 1; j <= qlen; ++j) { Hd[j] = j <= w ? -(p.o_ins + p.e_ins * j) : NEG; E[j] = NEG; }
	for (int i = 0; i < tlen; ++i) {
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int f = NEG, left = beg == 0 ? -(p.o_del + p.e_del * (i + 1)) : NEG;
		for (int j = beg; j < end; ++j) {
			const int m = Hd[j] + sc(p, t[i], q[j]);
			int e = E[j], h = m >= e ? m : e;
			Hd[j] = left;
			h = h >= f ? h : f;
			left = h;
			int x = m - oe_del; e -= p.e_del; E[j] = e > x ? e : x;
			x = m - oe_ins; f -= p.e_ins; f = f > x ? f : x;
		}
		Hd[end] = left; E[end] = NEG;
	}
	return Hd[qlen];
}

// the score bwa_gen_cigar2 returns for read[qb, qb+l_query) against text [rb, re)
static int gen_score(const bmh_ext_params_t &p, int w_, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query, int64_t rb, int64_t re)
{
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	const int rlen = (int)(re - rb);
	const bool flip = rb >= l_pac;
	std::vector<uint8_t> rs(rlen), qs(l_query);
	for (int i = 0; i < rlen; ++i) rs[i] = (uint8_t)text_base(pac, l_pac, flip ? re - 1 - i : rb + i);
	for (int i = 0; i < l_query; ++i) qs[i] = query[flip ? l_query - 1 - i : i];
	if (l_query == rlen && w_ == 0) { int s = 0; for (int i = 0; i < l_query; ++i) s += sc(p, rs[i], qs[i]); return s; }
	int max_ins = (int)((double)(((l_query + 1) >> 1) * p.a - p.o_ins) / p.e_ins + 1.);
	int max_del = (int)((double)(((l_query + 1) >> 1) * p.a - p.o_del) / p.e_del + 1.);
	int max_gap = max_ins > max_del ? max_ins : max_del;
	max_gap = max_gap > 1 ? max_gap : 1;
	const int diff = std::abs(rlen - l_query);
	int w = (max_gap + diff + 1) >> 1;
	w = w < w_ ? w : w_;
	w = w > diff + 3 ? w : diff + 3;
	if (g_post_stats) g_st_dp_calls++, g_st_dp_cells += (unsigned long long)rlen * (unsigned long long)((2 * w + 1) < l_query ? 2 * w + 1 : l_query);
	return global_score(p, l_query, qs.data(), rlen, rs.data(), w);
}

int pos2rid(const Ctx &x, int64_t pos_f)          // bns_pos2rid, src/bntseq.c:349-363
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

static int patch_reg(const Ctx &x, const uint8_t *query, const Reg &a, const Reg &b, int *w_out)        // mem_patch_reg
{
	if (!query) return 0;                                   // mem_patch_reg without bns/pac/query (mem_matesw's call) never merges
	if (a.rb < x.l_pac && b.rb >= x.l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	int w = (int)((a.re - b.rb) - (a.qe - b.qb));
	w = w > 0 ? w : -w;
	double r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb);
	r = r > 0. ? r : -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > x.co->w << 1 || r >= 0.05f) return 0; }
	else if (w > x.co->w << 2 || r >= 0.05f * 2) return 0;
	w += a.w + b.w;
	w = w < x.co->w << 2 ? w : x.co->w << 2;
	if (g_post_stats) g_st_patch_calls++;
	const int score = gen_score(*x.ep, w, x.l_pac, x.pac, b.qe - a.qb, query + a.qb, a.rb, b.re);
	const int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
	const int r_s = (int)((double)(b.re - a.rb) / ((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
	*w_out = w;
	return score;
}

// klib's introsort of regions by a key, for lists long enough to care: the exchanges the sort makes depend on the outcomes of its comparisons only, so
// sorting (key, place) pairs of 24 bytes with the same code gives the permutation sorting the 104-byte regions would have given, ties included; the
// regions are then moved once.  (mem_matesw re-sorts the mate's list after every window: on repeat-rich pairs this was 60 % of the second walk.)
template <class K, class KeyOf, class LT> static void sort_regs_by_key(int n, Reg *a, KeyOf key_of, LT lt)
{
	if (n <= 12) { klib::klib_introsort((size_t)n, a, [&](const Reg &p, const Reg &q) { return lt(key_of(p), key_of(q)); }); return; }
	struct E { K k; int at; };
	static thread_local std::vector<E> es;
	static thread_local std::vector<Reg> tmp;
	es.resize((size_t)n);
	for (int i = 0; i < n; ++i) { es[(size_t)i].k = key_of(a[i]); es[(size_t)i].at = i; }
	klib::klib_introsort((size_t)n, es.data(), [&](const E &p, const E &q) { return lt(p.k, q.k); });
	tmp.assign(a, a + n);
	for (int i = 0; i < n; ++i) a[i] = tmp[(size_t)es[(size_t)i].at];
}
struct KeySRQ { int score; int64_t rb; int qb; };

int sort_dedup_patch(const Ctx &x, const uint8_t *query, int n, Reg *a)        // mem_sort_dedup_patch
{
	if (n <= 1) return n;
	sort_regs_by_key<int64_t>(n, a, [](const Reg &p) { return p.re; }, [](int64_t p, int64_t q) { return p < q; });
	for (int i = 0; i < n; ++i) a[i].n_comp = 1;
	for (int i = 1; i < n; ++i) {
		Reg *p = &a[i];
		if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + x.co->max_chain_gap) continue;
		for (int j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + x.co->max_chain_gap; --j) {
			Reg *q = &a[j];
			if (q->qe == q->qb) continue;
			const int64_t pr = q->re - p->rb;
			const int64_t pq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			const int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			int score, w;
			if (pr > x.po->mask_level_redun * mr && pq > x.po->mask_level_redun * mq) {
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(x, query, *q, *p, &w)) > 0) {
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	int m = 0;
	for (int i = 0; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	n = m;
	sort_regs_by_key<KeySRQ>(n, a, [](const Reg &p) { return KeySRQ{p.score, p.rb, p.qb}; }, [](const KeySRQ &p, const KeySRQ &q) {
		return p.score > q.score || (p.score == q.score && (p.rb < q.rb || (p.rb == q.rb && p.qb < q.qb)));
	});
	for (int i = 1; i < n; ++i)
		if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
	m = n ? 1 : 0;
	for (int i = 1; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	return m;
}

uint64_t hash64(uint64_t key)
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

void set_is_alt(const Ctx &x, int n, Reg *a)
{
	const uint8_t *alt = x.po->contig_is_alt;
	for (int i = 0; i < n; ++i) a[i].is_alt = (alt && a[i].rid >= 0 && alt[a[i].rid]) ? 1 : 0;
}

// mem_mark_primary_se_core, src/bwamem.c:685-712 (sub_n is NOT reset between the two rounds of mem_mark_primary_se: it adds up, as there)
static void mark_primary_core(const Ctx &x, int n, Reg *a, std::vector<int> &z)
{
	int tmp = x.ep->a + x.ep->b;
	tmp = x.ep->o_del + x.ep->e_del > tmp ? x.ep->o_del + x.ep->e_del : tmp;
	tmp = x.ep->o_ins + x.ep->e_ins > tmp ? x.ep->o_ins + x.ep->e_ins : tmp;
	z.clear();
	z.push_back(0);
	for (int i = 1; i < n; ++i) {
		size_t k;
		for (k = 0; k < z.size(); ++k) {
			const int j = z[k];
			const int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
			const int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				const int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if (e_min - b_max >= min_l * x.co->mask_level) {
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == z.size()) z.push_back(i);
		else a[i].secondary = z[k];
	}
}

int mark_primary(const Ctx &x, int n, Reg *a, int64_t id)        // mem_mark_primary_se, src/bwamem.c:714-760
{
	if (n == 0) return 0;
	int n_pri = 0;
	for (int i = 0; i < n; ++i) {
		a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash64((uint64_t)(id + i));
		if (!a[i].is_alt) ++n_pri;
	}
	klib::klib_introsort((size_t)n, a, [](const Reg &p, const Reg &q) {           // alnreg_hlt
		return p.score > q.score || (p.score == q.score && (p.is_alt < q.is_alt || (p.is_alt == q.is_alt && p.hash < q.hash)));
	});
	std::vector<int> z;
	mark_primary_core(x, n, a, z);
	for (int i = 0; i < n; ++i) {
		Reg *p = &a[i];
		p->secondary_all = i;                                                      // keep the rank of the first round
		if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt) p->alt_sc = a[p->secondary].score;
	}
	if (n_pri >= 0 && n_pri < n) {                                                // there are ALT hits
		z.assign((size_t)n, 0);
		if (n_pri > 0)
			klib::klib_introsort((size_t)n, a, [](const Reg &p, const Reg &q) {     // alnreg_hlt2: the primary assembly first
				return p.is_alt < q.is_alt || (p.is_alt == q.is_alt && (p.score > q.score || (p.score == q.score && p.hash < q.hash)));
			});
		for (int i = 0; i < n; ++i) z[(size_t)a[i].secondary_all] = i;
		for (int i = 0; i < n; ++i) {
			if (a[i].secondary >= 0) {
				a[i].secondary_all = z[(size_t)a[i].secondary];
				if (a[i].is_alt) a[i].secondary = INT32_MAX;
			} else a[i].secondary_all = -1;
		}
		if (n_pri > 0) {                                                            // mark primary for hits to the primary assembly only
			for (int i = 0; i < n_pri; ++i) { a[i].sub = 0; a[i].secondary = -1; }
			mark_primary_core(x, n_pri, a, z);
		}
	} else {
		for (int i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	}
	return n_pri;
}

int approx_mapq(const Ctx &x, const Reg &a)        // mem_approx_mapq_se, mapQ_coef_len > 0 form
{
	int mapq, sub = a.sub ? a.sub : x.co->min_seed_len * x.ep->a;
	sub = a.csub > sub ? a.csub : sub;
	if (sub >= a.score) return 0;
	const int l = a.qe - a.qb > a.re - a.rb ? a.qe - a.qb : (int)(a.re - a.rb);
	const double identity = 1. - (double)(l * x.ep->a - a.score) / (x.ep->a + x.ep->b) / l;
	if (a.score == 0) mapq = 0;
	else {
		double tmp = l < x.po->mapQ_coef_len ? 1. : x.po->mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a.score - sub) / x.ep->a * tmp * tmp + .499);
	}
	if (a.sub_n > 0) mapq -= (int)(4.343 * log(a.sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a.frac_rep) + .499);
	return mapq;
}

void reg_from_record(const Ctx &x, const int32_t *g, float frac_rep, Reg &p)
{
	memset(&p, 0, sizeof(p));
	p.score = p.truesc = g[1]; p.qb = g[2]; p.qe = g[3];
	p.rb = (int64_t)(uint32_t)g[4] | (int64_t)g[5] << 32; p.re = (int64_t)(uint32_t)g[6] | (int64_t)g[7] << 32;
	// the sequence of the region = that of its chain's seeds: the extension windows never leave it (bns_fetch_seq)
	p.rid = pos2rid(x, p.rb < x.l_pac ? p.rb : (x.l_pac << 1) - 1 - (p.re - 1)); p.w = x.co->w; p.secondary = -1; p.frac_rep = frac_rep;
}

} // namespace rp

using namespace rp;

extern "C" void bmh_post_opt_default(bmh_post_opt_t *o)        // mem_opt_init, src/bwamem.c:101-146
{
	memset(o, 0, sizeof(*o));
	o->T = 30; o->mask_level_redun = 0.95f; o->mapQ_coef_len = 50.f; o->mapQ_coef_fac = (int)log(50.f); o->flag_all = 0; o->id0 = 0;
	o->XA_drop_ratio = 0.80f; o->max_XA_hits = 5; o->max_XA_hits_alt = 200; o->contig_is_alt = nullptr; o->rg_id = nullptr;
}

// regs_in[n][8] = {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi} grouped by read (regs_per_read); frac_rep per read.
// out[..][16] = {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi, truesc, w, sub, sub_n, secondary, mapq, flag, reported};
// out_per_read[n_reads]; returns the number of output regions, < 0 on error.
// read_ids (optional): the identity of read r in its batch -- what seeds the tie-break hash (id0 + id) and goes into the records' [0] -- when the
// reads handed over are a SUBSET of a batch (csrc/align_pipeline.hip: the reads with hits on ALT contigs, redone on the host)
int64_t bmh_finalize_regs_ids(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, int64_t l_pac,
                              const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                              const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                              int n_contigs, const int64_t *contig_offset,
                              int32_t *out, uint32_t *out_per_read, int n_threads, const uint32_t *read_ids)
{
	if (!copt || !ep || !popt || !pac || (n_reads && (!reads || !read_offs || !regs_in || !regs_per_read || !out || !out_per_read))) {
		bmh_set_error("bmh_finalize_regs: null argument"); return BMH_EINVAL;
	}
	if (!(popt->mapQ_coef_len > 0)) { bmh_set_error("bmh_finalize_regs: mapQ_coef_len <= 0 (the seed-coverage form of MAPQ) is not restated"); return BMH_EINVAL; }
	if (n_contigs > 1 && !contig_offset) { bmh_set_error("bmh_finalize_regs: null contig table"); return BMH_EINVAL; }
	Ctx x = {copt, ep, popt, l_pac, pac, n_contigs, contig_offset};
	std::vector<uint64_t> in_off((size_t)n_reads + 1, 0);
	for (uint32_t r = 0; r < n_reads; ++r) in_off[r + 1] = in_off[r] + regs_per_read[r];
	if (n_threads < 1) n_threads = 1;
	auto work = [&](uint32_t r0, uint32_t r1) {
		std::vector<Reg> a;
		for (uint32_t r = r0; r < r1; ++r) {
			const int n_in = (int)regs_per_read[r];
			a.resize(n_in);
			for (int i = 0; i < n_in; ++i) {
				reg_from_record(x, regs_in + 8 * (in_off[r] + i), frac_rep ? frac_rep[r] : 0.f, a[i]);
			}
			int n = sort_dedup_patch(x, reads + read_offs[r], n_in, a.data());
			set_is_alt(x, n, a.data());
			const uint32_t rid_ = read_ids ? read_ids[r] : r;
			mark_primary(x, n, a.data(), popt->id0 + rid_);
			const bool altm = alt_mode(x);
			// mem_reg2sam: which regions are reported, supplementary flag, MAPQ cap
			int32_t *o = out + 16 * in_off[r];
			int l = 0, mapq0 = 0;
			for (int k = 0; k < n; ++k) {
				const Reg &p = a[k];
				int32_t *q = o + 16 * k;
				q[0] = (int32_t)rid_; q[1] = p.score; q[2] = p.qb; q[3] = p.qe;
				q[4] = (int32_t)(uint32_t)p.rb; q[5] = (int32_t)(p.rb >> 32); q[6] = (int32_t)(uint32_t)p.re; q[7] = (int32_t)(p.re >> 32);
				q[8] = p.truesc; q[9] = p.w; q[10] = p.sub > p.csub ? p.sub : p.csub; q[11] = altm ? p.secondary_all : p.sub_n; q[12] = p.secondary;
				int mapq = p.secondary < 0 ? approx_mapq(x, p) : 0, flag = p.secondary >= 0 ? 0x100 : 0, rep = 1;
				if (p.score < popt->T) rep = 0;
				else if (p.secondary >= 0 && (p.is_alt || !popt->flag_all)) rep = 0;                        // src/bwamem.c:1742
				else if (p.secondary >= 0 && p.secondary < INT32_MAX && p.score < a[p.secondary].score * copt->drop_ratio) rep = 0;
				if (rep) {
					if (l && p.secondary < 0) flag |= popt->no_multi ? 0x10000 : 0x800;     // src/bwamem.c:1754
					if (l && !p.is_alt && mapq > mapq0) mapq = mapq0;                        // :1755
					if (l == 0) mapq0 = mapq;
					++l;
				}
				q[13] = mapq; q[14] = flag; q[15] = rep | (p.is_alt ? 2 : 0) | (p.alt_sc > 0 ? p.alt_sc << 2 : 0);
			}
			out_per_read[r] = (uint32_t)n;
		}
	};
	if (n_threads == 1 || n_reads < 2) work(0, n_reads);
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t) th.emplace_back(work, (uint32_t)((uint64_t)n_reads * t / n_threads), (uint32_t)((uint64_t)n_reads * (t + 1) / n_threads));
		for (auto &t : th) t.join();
	}
	if (getenv("BMH_POST_STATS"))
		fprintf(stderr, "[finalize] %u reads: patch tests that reach the alignment %llu, of them with a DP %llu (%.1f cells each)\n", n_reads,
		        (unsigned long long)g_st_patch_calls.exchange(0), (unsigned long long)g_st_dp_calls.load(), (double)g_st_dp_cells.exchange(0) / (double)(g_st_dp_calls.load() ? g_st_dp_calls.load() : 1)), g_st_dp_calls = 0;
	// compact to the front (every read wrote at its input offset)
	uint64_t w = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		const uint64_t src = in_off[r];
		if (w != src) memmove(out + 16 * w, out + 16 * src, sizeof(int32_t) * 16 * out_per_read[r]);
		w += out_per_read[r];
	}
	return (int64_t)w;
}

extern "C" int64_t bmh_finalize_regs(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, int64_t l_pac,
                                     const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                     const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                                     int n_contigs, const int64_t *contig_offset,
                                     int32_t *out, uint32_t *out_per_read, int n_threads)
{
	return bmh_finalize_regs_ids(copt, ep, popt, l_pac, pac, n_reads, reads, read_offs, regs_in, regs_per_read, frac_rep, n_contigs, contig_offset, out, out_per_read, n_threads, nullptr);
}
