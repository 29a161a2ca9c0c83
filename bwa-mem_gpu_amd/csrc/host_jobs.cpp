// Host-side restatement of what sits between the two kernels of the hot path in the reference:
//   seeds -> chains            mem_chain          /root/reference/src/bwamem.c:404-477 (+ test_and_merge :337-364)
//   chain weights + filter     mem_chain_flt      :366-392, 487-559
//   chains -> extension jobs   mem_chain2aln      :1170-1479 (which seeds get extended, left/right job construction)
//   job results -> regions     mem_align1_core    :2297-2303 (score = L + R - seedlen, qb/qe/rb/re from the part ends)
// (SURVEY.md section 8f rank 1).  The reference runs this per read on its host threads; so do we (std::thread
// over contiguous read ranges), producing one flat batch of extension jobs for bmh_extend_batch.
//   seed filter               mem_flt_chained_seeds :970-991 + mem_seed_sw :774-807: a no-op while min_l > 0.05 * l_query (reads
//                             up to ~730 bp at the default -W 0); beyond that, or with a small explicit -W, every seed of the
//                             kept chains is re-scored by a local alignment of its neighbourhood (bmh_local_sw = ksw_align2)
//
// Job order: per read, per region in creation order, LEFT job then RIGHT job -- the reference's SHORT/LONG split
// (:1396-1426) only decides which GASAL batch a job rides in, results are matched back by position.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>
#include "bmh_internal.h"
#include "klib_sort.h"
#include "local_sw.h"

namespace {

struct Seed { int64_t rbeg; int32_t qbeg, len, score; };
struct Chain {
	int n_first_qbeg() const { return seeds.front().qbeg; }
	int64_t pos; int rid; int w = 0, kept = 0, first = -1; float frac_rep = 0.f; bool is_alt = false;
	std::vector<Seed> seeds;
};
struct Reg {        // the fields of mem_alnreg_t this stage fills (bwamem.h:82-112)
	int64_t rb_est, re_est; int qb_est, qe_est;
	int64_t seed_rbeg; int seed_qbeg, seedlen0, align_sides, w, rid, seedcov; float frac_rep;
	int score, truesc; int qb, qe; int64_t rb, re;
	int64_t job[2];     // index of the LEFT / RIGHT job in the batch, -1 if none
};

struct Contigs { int64_t l_pac; int n; const int64_t *offset; const int32_t *len; };

inline int pos2rid(const Contigs &c, int64_t pos_f)          // bns_pos2rid, src/bntseq.c:349-363
{
	if (pos_f >= c.l_pac) return -1;
	if (c.n <= 1) return 0;
	int left = 0, mid = 0, right = c.n;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= c.offset[mid]) {
			if (mid == c.n - 1) break;
			if (pos_f < c.offset[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
inline int64_t depos(const Contigs &c, int64_t pos, int *is_rev)   // bns_depos, src/bntseq.h:139-142
{
	return (*is_rev = (pos >= c.l_pac)) ? (c.l_pac << 1) - 1 - pos : pos;
}
inline int intv2rid(const Contigs &c, int64_t rb, int64_t re)      // bns_intv2rid, src/bntseq.c:365-373
{
	int is_rev;
	if (rb < c.l_pac && re > c.l_pac) return -2;
	int rid_b = pos2rid(c, depos(c, rb, &is_rev));
	int rid_e = rb < re ? pos2rid(c, depos(c, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}
inline int text_base(const uint8_t *pac, int64_t l_pac, int64_t i)  // symbol i of fwd . revcomp(fwd); bns_get_seq :558-580
{
	const bool rev = i >= l_pac;
	const int64_t p = rev ? (l_pac << 1) - 1 - i : i;
	const int c = (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}

inline int cal_max_gap(const bmh_chain_opt_t &o, int qlen)          // src/bwamem.c:996-1002
{
	int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
	int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < o.w << 1 ? l : o.w << 1;
}

bool test_and_merge(const bmh_chain_opt_t &o, int64_t l_pac, Chain &c, const Seed &p, int seed_rid)   // :337-364
{
	const Seed &last = c.seeds.back();
	const int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len;
	if (seed_rid != c.rid) return false;
	if (p.qbeg >= c.seeds[0].qbeg && p.qbeg + p.len <= qend && p.rbeg >= c.seeds[0].rbeg && p.rbeg + p.len <= rend) return true;
	if ((last.rbeg < l_pac || c.seeds[0].rbeg < l_pac) && p.rbeg >= l_pac) return false;
	const int64_t x = p.qbeg - last.qbeg, y = p.rbeg - last.rbeg;
	if (y >= 0 && x - y <= o.w && y - x <= o.w && x - last.len < o.max_chain_gap && y - last.len < o.max_chain_gap) {
		c.seeds.push_back(p);
		return true;
	}
	return false;
}

int chain_weight(const Chain &c)                                    // mem_chain_weight :366-392
{
	int64_t end; int w = 0, tmp;
	end = 0;
	for (const Seed &s : c.seeds) {
		if (s.qbeg >= end) w += s.len;
		else if (s.qbeg + s.len > end) w += s.qbeg + s.len - end;
		end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
	}
	tmp = w; w = 0; end = 0;
	for (const Seed &s : c.seeds) {
		if (s.rbeg >= end) w += s.len;
		else if (s.rbeg + s.len > end) w += s.rbeg + s.len - end;
		end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

inline int chn_beg(const Chain &c) { return c.seeds.front().qbeg; }
inline int chn_end(const Chain &c) { return c.seeds.back().qbeg + c.seeds.back().len; }

void chain_flt(const bmh_chain_opt_t &o, std::vector<Chain> &a)     // mem_chain_flt :487-559
{
	if (a.empty()) return;
	{
		size_t k = 0;
		for (size_t i = 0; i < a.size(); ++i) {
			Chain &c = a[i];
			c.first = -1; c.kept = 0; c.w = chain_weight(c);
			if (c.w >= o.min_chain_weight) { if (k != i) a[k] = std::move(c); ++k; }
		}
		a.resize(k);
	}
	if (a.empty()) return;
	klib::klib_introsort(a.size(), a.data(), [](const Chain &x, const Chain &y) { return x.w > y.w; });
	const int n = (int)a.size();
	std::vector<int> chains;
	a[0].kept = 3;
	chains.push_back(0);
	for (int i = 1; i < n; ++i) {
		bool large_ovlp = false;
		size_t k;
		for (k = 0; k < chains.size(); ++k) {
			const int j = chains[k];
			const int b_max = chn_beg(a[j]) > chn_beg(a[i]) ? chn_beg(a[j]) : chn_beg(a[i]);
			const int e_min = chn_end(a[j]) < chn_end(a[i]) ? chn_end(a[j]) : chn_end(a[i]);
			if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
				const int li = chn_end(a[i]) - chn_beg(a[i]), lj = chn_end(a[j]) - chn_beg(a[j]);
				const int min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {
					large_ovlp = true;
					if (a[j].first < 0) a[j].first = i;
					if (a[i].w < a[j].w * o.drop_ratio && a[j].w - a[i].w >= o.min_seed_len << 1) break;
				}
			}
		}
		if (k == chains.size()) { chains.push_back(i); a[i].kept = large_ovlp ? 2 : 3; }
	}
	for (int ci : chains) if (a[ci].first >= 0) a[a[ci].first].kept = 1;
	int i, k;
	for (i = k = 0; i < n; ++i) {
		if (a[i].kept == 0 || a[i].kept == 3) continue;
		if (++k >= o.max_chain_extend) break;
	}
	for (; i < n; ++i) if (a[i].kept < 3) a[i].kept = 0;
	size_t kk = 0;
	for (size_t q = 0; q < a.size(); ++q) if (a[q].kept != 0) { if (kk != q) a[kk] = std::move(a[q]); ++kk; }
	a.resize(kk);
}

// does the reference's seed filter run for a read of this length?  (src/bwamem.c:972-977; MEM_HSP_COEF 1.1f and MEM_SEEDSW_COEF
// 0.05f are float constants there: the products are formed in float, the comparison in double)
inline bool seed_filter_applies(const bmh_chain_opt_t &o, int l_query, int *min_HSP_score)
{
	const double min_l = o.min_chain_weight ? (double)(1.1f * (float)o.min_chain_weight) : (double)5.5f * log((double)l_query);
	if (min_HSP_score) *min_HSP_score = (int)(o.a * min_l + .499);
	return !(min_l > (double)(0.05f * (float)l_query));
}

struct Out {
	std::vector<uint8_t> q, t;
	std::vector<uint32_t> qoff, qlen, toff, tlen, h0, job_read, job_reg, job_side;
	std::vector<Reg> regs;
	std::vector<uint32_t> reg_read;
	std::vector<uint32_t> regs_per_read;
	std::vector<float> frac_rep;        // per read: fraction of the read covered by over-represented SMEMs (mem_chain :415-459)
};

struct Ctx {
	const bmh_chain_opt_t *o; Contigs ctg; const uint8_t *pac;
	const uint8_t *reads; const uint64_t *roffs; const uint32_t *rlens;
	const uint64_t *rbeg; const int32_t *qbeg; const uint32_t *score, *n_ref, *prefix;
};

void make_chains(const Ctx &x, uint32_t r, int len, std::vector<Chain> &out)     // mem_chain :404-477
{
	const bmh_chain_opt_t &o = *x.o;
	out.clear();
	if (len < o.min_seed_len) return;
	const uint64_t base = x.prefix[r];
	const uint32_t n = x.n_ref[r];
	int b = 0, e = 0, l_rep = 0;
	for (uint32_t i = 0; i < n; i += x.score[base + i]) {
		const int sb = x.qbeg[2 * (base + i)], se = x.qbeg[2 * (base + i) + 1];
		if (x.score[base + i] <= (uint32_t)o.max_occ) continue;
		if (sb > e) { l_rep += e - b; b = sb; e = se; }
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	std::multimap<int64_t, size_t> tree;       // chains keyed by pos (kbtree chn, :333)
	std::vector<Chain> pool;
	for (uint32_t i = 0; i < n; i += x.score[base + i]) {
		const uint32_t cnt = x.score[base + i];
		const int slen = x.qbeg[2 * (base + i) + 1] - x.qbeg[2 * (base + i)];
		const int step = cnt > (uint32_t)o.max_occ ? (int)(cnt / o.max_occ) : 1;
		int count = 0;
		for (int64_t k = 0; k < (int64_t)cnt && count < o.max_occ; k += step, ++count) {
			Seed s;
			s.rbeg = (int64_t)x.rbeg[base + i + k];
			s.qbeg = x.qbeg[2 * (base + i)];
			s.score = s.len = slen;
			const int rid = intv2rid(x.ctg, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue;
			bool to_add = false;
			if (!tree.empty()) {
				auto it = tree.upper_bound(s.rbeg);     // closest chain at or below the seed (kb_intervalp lower)
				if (it == tree.begin()) to_add = true;
				else { --it; if (!test_and_merge(o, x.ctg.l_pac, pool[it->second], s, rid)) to_add = true; }
			} else to_add = true;
			if (to_add) {
				Chain c; c.pos = s.rbeg; c.rid = rid; c.is_alt = o.contig_is_alt && o.contig_is_alt[rid]; c.seeds.push_back(s);      // tmp.is_alt = !!bns->anns[rid].is_alt (src/bwamem.c:446)
				pool.push_back(std::move(c));
				tree.emplace(s.rbeg, pool.size() - 1);
			}
		}
	}
	out.reserve(pool.size());
	for (auto &kv : tree) { pool[kv.second].frac_rep = (float)l_rep / len; out.push_back(std::move(pool[kv.second])); }
}

void push_job(Out &O, uint32_t read, uint32_t reg, int side, const uint8_t *q, int qn, bool qrev, const Ctx &x, int64_t t0, int tn, bool trev, int h0)
{
	O.qoff.push_back((uint32_t)O.q.size()); O.toff.push_back((uint32_t)O.t.size());
	O.qlen.push_back((uint32_t)qn); O.tlen.push_back((uint32_t)tn); O.h0.push_back((uint32_t)h0);
	O.job_read.push_back(read); O.job_reg.push_back(reg); O.job_side.push_back((uint32_t)side);
	const size_t q0 = O.q.size(); O.q.resize(q0 + qn);
	for (int i = 0; i < qn; ++i) O.q[q0 + i] = qrev ? q[qn - 1 - i] : q[i];
	const size_t t1 = O.t.size(); O.t.resize(t1 + tn);
	for (int i = 0; i < tn; ++i) O.t[t1 + i] = (uint8_t)text_base(x.pac, x.ctg.l_pac, trev ? t0 + tn - 1 - i : t0 + i);
}

void chain2aln(const Ctx &x, uint32_t r, int l_query, const uint8_t *query, const Chain &c, size_t reg0, Out &O)   // mem_chain2aln :1170-1479
{
	const bmh_chain_opt_t &o = *x.o;
	const int64_t l_pac = x.ctg.l_pac;
	if (c.seeds.empty()) return;
	int64_t rmax[2] = {l_pac << 1, 0};
	for (const Seed &t : c.seeds) {
		const int64_t b = t.rbeg - (t.qbeg + cal_max_gap(o, t.qbeg));
		const int64_t e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + cal_max_gap(o, l_query - t.qbeg - t.len));
		rmax[0] = rmax[0] < b ? rmax[0] : b;
		rmax[1] = rmax[1] > e ? rmax[1] : e;
	}
	rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
	rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
	if (rmax[0] < l_pac && l_pac < rmax[1]) { if (c.seeds[0].rbeg < l_pac) rmax[1] = l_pac; else rmax[0] = l_pac; }
	{   // bns_fetch_seq clips the window to the contig of the first seed (src/bntseq.c:531-556)
		int is_rev;
		const int rid = pos2rid(x.ctg, depos(x.ctg, c.seeds[0].rbeg, &is_rev));
		int64_t far_beg = x.ctg.n > 1 ? x.ctg.offset[rid] : 0, far_end = far_beg + (x.ctg.n > 1 ? x.ctg.len[rid] : l_pac);
		if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
		rmax[0] = rmax[0] > far_beg ? rmax[0] : far_beg;
		rmax[1] = rmax[1] < far_end ? rmax[1] : far_end;
	}
	const int n = (int)c.seeds.size();
	std::vector<uint64_t> srt(n);
	for (int i = 0; i < n; ++i) srt[i] = (uint64_t)(uint32_t)c.seeds[i].score << 32 | (uint32_t)i;
	std::sort(srt.begin(), srt.end());            // keys are distinct: any sort gives ks_introsort_64's order
	for (int k = n - 1; k >= 0; --k) {
		const Seed &s = c.seeds[(uint32_t)srt[k]];
		size_t i;
		const size_t n_regs = O.regs.size() - reg0;
		for (i = 0; i < n_regs; ++i) {             // extension (estimated) made before? :1235-1256
			const Reg &p = O.regs[reg0 + i];
			if (s.rbeg < p.rb_est || s.rbeg + s.len > p.re_est || s.qbeg < p.qb_est || s.qbeg + s.len > p.qe_est) continue;
			if (s.len - p.seedlen0 > .1 * l_query) continue;
			int qd = s.qbeg - p.qb_est; int64_t rd = s.rbeg - p.rb_est;
			int max_gap = cal_max_gap(o, qd < rd ? qd : (int)rd);
			int w = max_gap < p.w ? max_gap : p.w;
			if (qd - rd < w && rd - qd < w) break;
			qd = p.qe_est - (s.qbeg + s.len); rd = p.re_est - (s.rbeg + s.len);
			max_gap = cal_max_gap(o, qd < rd ? qd : (int)rd);
			w = max_gap < p.w ? max_gap : p.w;
			if (qd - rd < w && rd - qd < w) break;
		}
		if (i < n_regs) {                          // :1258-1276
			int j;
			for (j = k + 1; j < n; ++j) {
				if (srt[j] == 0) continue;
				const Seed &t = c.seeds[(uint32_t)srt[j]];
				if (t.len < s.len * .95) continue;
				if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
				if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
			}
			if (j == n) { srt[k] = 0; continue; }
		}
		Reg a; memset(&a, 0, sizeof(a));
		a.w = o.w; a.score = a.truesc = -1; a.rid = c.rid;
		const int fwd = (int)(0.85 * (l_query - (s.qbeg + s.len)));       // FILTER_COEF, :52, :1285-1298
		a.qe_est = (s.qbeg + s.len) + fwd < l_query ? (s.qbeg + s.len) + fwd : l_query;
		a.re_est = (s.rbeg + s.len) + fwd < l_pac << 1 ? (s.rbeg + s.len) + fwd : l_pac << 1;
		const int back = (int)(0.85 * (s.qbeg + 1));
		a.qb_est = (s.qbeg - back) > 0 ? (s.qbeg - back) : 0;
		a.rb_est = (s.rbeg - back) > 0 ? (s.rbeg - back) : 0;
		if (a.rb_est < l_pac && l_pac < a.qe_est) { if (s.rbeg < l_pac) a.re_est = l_pac; else a.rb_est = l_pac; }   // (sic) qe_est, :1292
		const int lq = s.qbeg, lr = (int)(s.rbeg - rmax[0]);
		const int rq = l_query - (lq + s.len), rr = (int)(rmax[1] - rmax[0]) - (lr + s.len);
		a.score = a.truesc = s.len;
		a.seed_qbeg = s.qbeg; a.seed_rbeg = s.rbeg;
		a.job[0] = a.job[1] = -1;
		const uint32_t reg_idx = (uint32_t)O.regs.size();
		if (lq > 0) { a.job[0] = (int64_t)O.qlen.size(); push_job(O, r, reg_idx, 0, query, lq, true, x, rmax[0], lr, true, s.len); }
		if (rq > 0) { a.job[1] = (int64_t)O.qlen.size(); push_job(O, r, reg_idx, 1, query + lq + s.len, rq, false, x, s.rbeg + s.len, rr, false, s.len); }
		a.align_sides = (lq > 0) + (rq > 0);
		if (a.align_sides == 0) { a.score = a.truesc = s.score; a.qb = 0; a.rb = s.rbeg; a.qe = l_query; a.re = s.rbeg + s.len; }
		a.seedcov = 0;       // computed at :1462-1467 from qb/qe/rb/re that are still zero at that point
		for (const Seed &t : c.seeds)
			if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) a.seedcov += t.len;
		a.seedlen0 = s.len;
		a.frac_rep = c.frac_rep;
		O.regs.push_back(a);
		O.reg_read.push_back(r);
	}
}

// mem_seed_sw :774-807: local alignment score of the seed's neighbourhood (50 bases either side), -1 when the seed or its window
// is long enough to be trusted as it is
int seed_sw(const Ctx &x, int l_query, const uint8_t *query, const Seed &s, std::vector<uint8_t> &qbuf, std::vector<uint8_t> &tbuf)
{
	const int SHORT_EXT = 50, SHORT_LEN = 200;                      // MEM_SHORT_EXT, MEM_SHORT_LEN
	const int64_t l_pac = x.ctg.l_pac;
	if (s.len >= SHORT_LEN) return -1;
	int qb = s.qbeg, qe = s.qbeg + s.len;
	int64_t rb = s.rbeg, re = s.rbeg + s.len;
	const int64_t mid = (rb + re) >> 1;
	qb -= SHORT_EXT; qb = qb > 0 ? qb : 0;
	qe += SHORT_EXT; qe = qe < l_query ? qe : l_query;
	rb -= SHORT_EXT; rb = rb > 0 ? rb : 0;
	re += SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
	if (qe - qb >= SHORT_LEN || re - rb >= SHORT_LEN) return -1;
	{   // bns_fetch_seq(bns, pac, &rb, mid, &re, &rid): the window is clipped to the sequence that holds mid (src/bntseq.c:531-556)
		int is_rev;
		const int rid = pos2rid(x.ctg, depos(x.ctg, mid, &is_rev));
		int64_t far_beg = x.ctg.n > 1 ? x.ctg.offset[rid] : 0, far_end = far_beg + (x.ctg.n > 1 ? x.ctg.len[rid] : l_pac);
		if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
		rb = rb > far_beg ? rb : far_beg;
		re = re < far_end ? re : far_end;
	}
	qbuf.assign(query + qb, query + qe);
	tbuf.resize((size_t)(re - rb));
	for (int64_t i = 0; i < re - rb; ++i) tbuf[(size_t)i] = (uint8_t)text_base(x.pac, l_pac, rb + i);
	bmh_ext_params_t p; memset(&p, 0, sizeof(p));
	p.a = x.o->a; p.b = x.o->b; p.o_del = x.o->o_del; p.e_del = x.o->e_del; p.o_ins = x.o->o_ins; p.e_ins = x.o->e_ins;
	return bmh_local_sw(qe - qb, qbuf.data(), (int)(re - rb), tbuf.data(), p, BMH_SW_XSTART).score;     // ksw_align2(..., KSW_XSTART, 0)
}

// mem_flt_chained_seeds :970-991: seeds of the kept chains re-scored, the weak ones dropped (chains may end up empty: mem_chain2aln
// returns at once for those, :1187)
void flt_chained_seeds(const Ctx &x, int l_query, const uint8_t *query, std::vector<Chain> &a)
{
	int min_HSP_score;
	if (!seed_filter_applies(*x.o, l_query, &min_HSP_score)) return;
	std::vector<uint8_t> qbuf, tbuf;
	for (Chain &c : a) {
		size_t k = 0;
		for (size_t j = 0; j < c.seeds.size(); ++j) {
			Seed s = c.seeds[j];
			s.score = seed_sw(x, l_query, query, s, qbuf, tbuf);
			if (s.score < 0 || s.score >= min_HSP_score) {
				s.score = s.score < 0 ? s.len * x.o->a : s.score;
				c.seeds[k++] = s;
			}
		}
		c.seeds.resize(k);
	}
}

void worker(const Ctx &x, uint32_t r0, uint32_t r1, Out &O)
{
	std::vector<Chain> chains;
	O.regs_per_read.assign(r1 - r0, 0);
	O.frac_rep.assign(r1 - r0, 0.f);
	for (uint32_t r = r0; r < r1; ++r) {
		const int len = (int)x.rlens[r];
		const uint8_t *query = x.reads + x.roffs[r];
		make_chains(x, r, len, chains);
		if (!chains.empty()) O.frac_rep[r - r0] = chains[0].frac_rep;
		chain_flt(*x.o, chains);
		flt_chained_seeds(x, len, query, chains);
		const size_t reg0 = O.regs.size();
		for (const Chain &c : chains) chain2aln(x, r, len, query, c, reg0, O);
		O.regs_per_read[r - r0] = (uint32_t)(O.regs.size() - reg0);
	}
}

} // namespace

struct bmh_jobs {
	Out o;
	uint32_t n_reads;
};

extern "C" void bmh_chain_opt_default(bmh_chain_opt_t *o)       // mem_opt_init, src/bwamem.c:101-146
{
	memset(o, 0, sizeof(*o));
	o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1; o->w = 300;
	o->min_seed_len = 19; o->max_occ = 500; o->max_chain_gap = 10000; o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
	o->mask_level = 0.50f; o->drop_ratio = 0.50f;
}

extern "C" bmh_jobs_t *bmh_build_jobs(const bmh_chain_opt_t *opt, int64_t l_pac, const uint8_t *pac, int n_contigs,
                                      const int64_t *contig_offset, const int32_t *contig_len, uint32_t n_reads,
                                      const uint8_t *reads, const uint64_t *read_offs, const uint32_t *read_lens,
                                      const uint64_t *rbeg, const int32_t *qbeg, const uint32_t *score,
                                      const uint32_t *n_ref_pos, const uint32_t *prefix, int n_threads)
{
	if (!opt || !pac || (n_reads && (!reads || !read_offs || !read_lens || !n_ref_pos || !prefix))) { bmh_set_error("bmh_build_jobs: null argument"); return nullptr; }
	Ctx x; x.o = opt; x.ctg = {l_pac, n_contigs, contig_offset, contig_len}; x.pac = pac;
	x.reads = reads; x.roffs = read_offs; x.rlens = read_lens; x.rbeg = rbeg; x.qbeg = qbeg; x.score = score; x.n_ref = n_ref_pos; x.prefix = prefix;
	if (n_threads < 1) n_threads = 1;
	if ((uint32_t)n_threads > n_reads) n_threads = n_reads ? (int)n_reads : 1;
	std::vector<Out> parts(n_threads);
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t) {
		const uint32_t r0 = (uint32_t)((uint64_t)n_reads * t / n_threads), r1 = (uint32_t)((uint64_t)n_reads * (t + 1) / n_threads);
		if (n_threads == 1) worker(x, r0, r1, parts[t]);
		else th.emplace_back(worker, std::cref(x), r0, r1, std::ref(parts[t]));
	}
	for (auto &t : th) t.join();
	bmh_jobs *J = new bmh_jobs();
	J->n_reads = n_reads;
	Out &O = J->o;
	for (Out &P : parts) {
		const uint32_t qb = (uint32_t)O.q.size(), tb = (uint32_t)O.t.size(), rg = (uint32_t)O.regs.size();
		const int64_t jb = (int64_t)O.qlen.size();
		if ((uint64_t)O.q.size() + P.q.size() >= (1ull << 32) || (uint64_t)O.t.size() + P.t.size() >= (1ull << 32)) {
			bmh_set_error("bmh_build_jobs: batch exceeds 4 GiB of bases; split the read set"); delete J; return nullptr;
		}
		O.q.insert(O.q.end(), P.q.begin(), P.q.end()); O.t.insert(O.t.end(), P.t.begin(), P.t.end());
		for (uint32_t v : P.qoff) O.qoff.push_back(v + qb);
		for (uint32_t v : P.toff) O.toff.push_back(v + tb);
		O.qlen.insert(O.qlen.end(), P.qlen.begin(), P.qlen.end()); O.tlen.insert(O.tlen.end(), P.tlen.begin(), P.tlen.end());
		O.h0.insert(O.h0.end(), P.h0.begin(), P.h0.end());
		O.job_read.insert(O.job_read.end(), P.job_read.begin(), P.job_read.end());
		for (uint32_t v : P.job_reg) O.job_reg.push_back(v + rg);
		O.job_side.insert(O.job_side.end(), P.job_side.begin(), P.job_side.end());
		for (Reg a : P.regs) { if (a.job[0] >= 0) a.job[0] += jb; if (a.job[1] >= 0) a.job[1] += jb; O.regs.push_back(a); }
		O.reg_read.insert(O.reg_read.end(), P.reg_read.begin(), P.reg_read.end());
		O.regs_per_read.insert(O.regs_per_read.end(), P.regs_per_read.begin(), P.regs_per_read.end());
		O.frac_rep.insert(O.frac_rep.end(), P.frac_rep.begin(), P.frac_rep.end());
	}
	return J;
}

extern "C" void bmh_jobs_free(bmh_jobs_t *j) { delete j; }

extern "C" void bmh_jobs_sizes(const bmh_jobs_t *j, uint64_t *n_jobs, uint64_t *n_regs, uint64_t *q_bytes, uint64_t *t_bytes)
{
	if (n_jobs) *n_jobs = j->o.qlen.size();
	if (n_regs) *n_regs = j->o.regs.size();
	if (q_bytes) *q_bytes = j->o.q.size();
	if (t_bytes) *t_bytes = j->o.t.size();
}

extern "C" void bmh_jobs_arrays(const bmh_jobs_t *j, const uint8_t **q, const uint32_t **qoff, const uint32_t **qlen, const uint8_t **t,
                                const uint32_t **toff, const uint32_t **tlen, const uint32_t **h0, const uint32_t **job_read,
                                const uint32_t **job_reg, const uint32_t **job_side, const uint32_t **regs_per_read)
{
	const Out &O = j->o;
	if (q) *q = O.q.data(); if (qoff) *qoff = O.qoff.data(); if (qlen) *qlen = O.qlen.data();
	if (t) *t = O.t.data(); if (toff) *toff = O.toff.data(); if (tlen) *tlen = O.tlen.data(); if (h0) *h0 = O.h0.data();
	if (job_read) *job_read = O.job_read.data(); if (job_reg) *job_reg = O.job_reg.data(); if (job_side) *job_side = O.job_side.data();
	if (regs_per_read) *regs_per_read = O.regs_per_read.data();
}

extern "C" const float *bmh_jobs_frac_rep(const bmh_jobs_t *j) { return j->o.frac_rep.data(); }

// job results -> alignment regions (src/bwamem.c:2297-2303); out3 = {aln_score, query_end, target_end} per job.
// regs_out[n_regs][8] = {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi}; rb/re split into two 32-bit halves.
extern "C" int bmh_merge_regs(const bmh_jobs_t *j, const int32_t *out3, int32_t *regs_out)
{
	const Out &O = j->o;
	for (size_t i = 0; i < O.regs.size(); ++i) {
		Reg a = O.regs[i];
		if (a.align_sides > 0) {
			int ls = 0, lq = 0, lt = 0, rs = 0, rq = 0, rt = 0;
			if (a.job[0] >= 0) { ls = out3[3 * a.job[0]]; lq = out3[3 * a.job[0] + 1]; lt = out3[3 * a.job[0] + 2]; }
			if (a.job[1] >= 0) { rs = out3[3 * a.job[1]]; rq = out3[3 * a.job[1] + 1]; rt = out3[3 * a.job[1] + 2]; }
			a.score = ls + rs - (a.align_sides == 2 ? a.seedlen0 : 0);
			a.qb = a.seed_qbeg - lq; a.qe = a.seed_qbeg + a.seedlen0 + rq;
			a.rb = a.seed_rbeg - lt; a.re = a.seed_rbeg + a.seedlen0 + rt;
			a.truesc = a.score;
		}
		int32_t *o = regs_out + 8 * i;
		o[0] = (int32_t)O.reg_read[i]; o[1] = a.score; o[2] = a.qb; o[3] = a.qe;
		o[4] = (int32_t)(uint32_t)a.rb; o[5] = (int32_t)(a.rb >> 32); o[6] = (int32_t)(uint32_t)a.re; o[7] = (int32_t)(a.re >> 32);
	}
	return BMH_OK;
}
