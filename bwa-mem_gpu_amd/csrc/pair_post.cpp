// Host restatement of the reference's paired-end tail (BASELINE configs[3]):
//   mem_pestat    /root/reference/src/bwamem_pair.c:46-117    insert-size distribution of the batch per orientation
//   mem_matesw    :119-188   mate rescue: local alignment of the mate in the window the distribution predicts
//                            (ksw_align2 -> local_sw.cpp), new regions merged into the mate's list
//   mem_pair      :190-251   best pair of regions under the distribution (score + log-likelihood of the insert size)
//   mem_sam_pe    :257-397   which regions are written for the two reads, their MAPQ and pair flags
// Runs on host threads like the reference; regions come from bmh_merge_regs / bmh_chain_merge, CIGARs are added
// afterwards by bmh_cigar_batch and the text by bmh_format_sam_pe (sam_format.cpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
#include "bmh_internal.h"
#include "klib_sort.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "local_sw.h"
#include "pair_kernels.h"
#include "regs_post.h"

using namespace rp;

namespace {

struct Pes { int low, high, failed; double avg, std; };

// The local alignments of mem_matesw can be computed as one batch on the device (pair_kernels.hip): the pairs are then walked twice.
// sw_mode 1: an alignment that is about to be computed is written down instead (and answered "nothing found"); sw_mode 2: it is taken
// from the batch's results (or computed here, if the first walk did not foresee it -- the regions a rescue adds can change which
// orientations later calls of the same pair skip); sw_mode 0: computed on the spot (the host-only form).
// With the batch's regions on the device (bmh_pairs_split_t::rescue_in) the first walk is a kernel too (pair_kernels.hip: rescue_jobs_kernel).
typedef bmh_msw_key_t SwKey;
struct PCtx {
	Ctx x; const bmh_pe_opt_t *pe; Pes pes[4];
	const int64_t *ctg_off; const int32_t *ctg_len;
	const uint8_t *reads; const uint64_t *offs; const uint32_t *lens;
	int sw_mode = 0;
	std::vector<SwKey> *col_keys = nullptr; std::vector<bmh_msw_job_t> *col_jobs = nullptr;      // sw_mode 1 (per thread)
	const SwKey *keys = nullptr; const uint64_t *pair_off = nullptr; const int32_t *res = nullptr;   // sw_mode 2
	// sw_mode 1 writes, sw_mode 2 reads: 1 = a mem_matesw call of the pair got as far as a window (only then can the rescue change the pair's
	// regions: for the other pairs -- nearly all -- the second walk skips the rescue, which is most of a walk on repeat-rich reads)
	uint8_t *pair_active = nullptr;
};

inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)      // mem_infer_dir
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

struct Span { const Reg *p; size_t n; bool empty() const { return n == 0; } size_t size() const { return n; } const Reg &operator[](size_t i) const { return p[i]; } };

int cal_sub(const PCtx &c, const Span &r)
{
	size_t j;
	for (j = 1; j < r.size(); ++j) {
		const int b_max = r[j].qb > r[0].qb ? r[j].qb : r[0].qb;
		const int e_min = r[j].qe < r[0].qe ? r[j].qe : r[0].qe;
		if (e_min > b_max) {
			const int min_l = r[j].qe - r[j].qb < r[0].qe - r[0].qb ? r[j].qe - r[j].qb : r[0].qe - r[0].qb;
			if (e_min - b_max >= min_l * c.x.co->mask_level) break;
		}
	}
	return j < r.size() ? r[j].score : c.x.co->min_seed_len * c.x.ep->a;
}

template <class RegsOf> void pestat(PCtx &c, size_t n, RegsOf regs_of, int n_threads)            // mem_pestat; regs_of(r) -> (pointer, count)
{
	std::vector<uint64_t> isize[4];
	memset(c.pes, 0, sizeof(c.pes));
	// candidate insert sizes of the pairs: ranges of pairs on threads, concatenated in pair order (the values are sorted next, so
	// only the multiset matters)
	const size_t n_pairs = n >> 1;
	int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
	if (n_pairs < 4096) nt = 1;
	std::vector<std::vector<uint64_t>> loc((size_t)nt * 4);
	auto collect = [&](int t) {
		const size_t p0 = n_pairs * (size_t)t / (size_t)nt, p1 = n_pairs * (size_t)(t + 1) / (size_t)nt;
		for (size_t i = p0; i < p1; ++i) {
			const Span r0 = regs_of(i << 1), r1 = regs_of(i << 1 | 1);
			if (r0.empty() || r1.empty()) continue;
			if (cal_sub(c, r0) > 0.8 * r0[0].score) continue;
			if (cal_sub(c, r1) > 0.8 * r1[0].score) continue;
			if (r0[0].rid != r1[0].rid) continue;
			int64_t is;
			const int dir = infer_dir(c.x.l_pac, r0[0].rb, r1[0].rb, &is);
			if (is && is <= c.pe->max_ins) loc[(size_t)t * 4 + (size_t)dir].push_back((uint64_t)is);
		}
	};
	if (nt == 1) collect(0);
	else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(collect, t); for (auto &x : th) x.join(); }
	for (int d = 0; d < 4; ++d) for (int t = 0; t < nt; ++t) isize[d].insert(isize[d].end(), loc[(size_t)t * 4 + d].begin(), loc[(size_t)t * 4 + d].end());
	for (int d = 0; d < 4; ++d) {
		Pes *r = &c.pes[d];
		std::vector<uint64_t> &q = isize[d];
		if (q.size() < 10) { r->failed = 1; continue; }
		if (c.pe->max_ins > 0 && c.pe->max_ins <= (1 << 22)) {          // values are 1..max_ins: counting sort
			std::vector<uint32_t> cntv((size_t)c.pe->max_ins + 1, 0);
			for (uint64_t v : q) ++cntv[(size_t)v];
			size_t k = 0;
			for (size_t v = 0; v < cntv.size(); ++v) for (uint32_t m = 0; m < cntv[v]; ++m) q[k++] = (uint64_t)v;
		} else std::sort(q.begin(), q.end());
		const int p25 = (int)q[(int)(.25 * q.size() + .499)], p50 = (int)q[(int)(.50 * q.size() + .499)], p75 = (int)q[(int)(.75 * q.size() + .499)];
		(void)p50;
		r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
		size_t x = 0;
		r->avg = 0;
		for (uint64_t v : q) if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) { r->avg += v; ++x; }
		r->avg /= x;
		r->std = 0;
		for (uint64_t v : q) if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) r->std += (v - r->avg) * (v - r->avg);
		r->std = sqrt(r->std / x);
		r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);
		r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
		if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);
		if (r->high < r->avg - 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
		if (r->low < 1) r->low = 1;
	}
	size_t mx = 0;
	for (int d = 0; d < 4; ++d) mx = mx > isize[d].size() ? mx : isize[d].size();
	for (int d = 0; d < 4; ++d) if (c.pes[d].failed == 0 && isize[d].size() < mx * 0.05) c.pes[d].failed = 1;
}

int pos2rid_c(const PCtx &c, int64_t pos_f) { return pos2rid(c.x, pos_f); }

// bns_fetch_seq (src/bntseq.c:531-556): [beg, end) clipped to the sequence (and strand) that holds mid
bool fetch_window(const PCtx &c, int64_t *beg, int64_t mid, int64_t *end, int *rid, std::vector<uint8_t> &seq, bool want_bases = true)
{
	if (*end < *beg) std::swap(*beg, *end);
	const int64_t l_pac = c.x.l_pac;
	const bool is_rev = mid >= l_pac;
	*rid = pos2rid_c(c, is_rev ? (l_pac << 1) - 1 - mid : mid);
	int64_t far_beg = c.x.n_contigs > 1 ? c.ctg_off[*rid] : 0, far_end = far_beg + (c.x.n_contigs > 1 ? c.ctg_len[*rid] : l_pac);
	if (is_rev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	if (*beg >= *end) { seq.clear(); return false; }
	if (!want_bases) return true;
	seq.resize((size_t)(*end - *beg));
	for (int64_t i = *beg; i < *end; ++i) seq[(size_t)(i - *beg)] = (uint8_t)text_base(c.x.pac, l_pac, i);
	return true;
}

// BMH_POST_STATS: how much local alignment the mate rescue does
static std::atomic<unsigned long long> g_ms_calls{0}, g_ms_sw{0}, g_ms_cells{0}, g_ms_hits{0};
static const bool g_pair_stats = getenv("BMH_POST_STATS") != nullptr;      // the counters are only read (and only bumped) with it set

// knob RESCUE_CHECK (BMH_RESCUE_CHECK=1 / bmh_tune_set): the device's search for the rescue's windows against the host's first walk (bmh_rescue_check_counts)
static std::atomic<unsigned long long> g_chk_batches{0}, g_chk_pairs{0}, g_chk_jobs{0}, g_chk_active{0}, g_chk_lists{0};

// BMH_PAIR_PROFILE: where the second walk spends its time (nanoseconds summed over the threads)
static const bool g_pair_prof = getenv("BMH_PAIR_PROFILE") != nullptr;
static std::atomic<unsigned long long> g_ns_msw{0}, g_ns_msw_dedup{0}, g_ns_mark{0}, g_ns_pair{0};
static inline unsigned long long now_ns() { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ma_state (optional, one byte per mate list, 0 at first): bit 0 = the list has been through a mem_sort_dedup_patch call of THIS function (the one without
// patching) and nothing was inserted since -- another such call leaves it as it is (every surviving pair of hits has been tested in its final form and the
// closing sort orders by a key no two survivors share), so it is skipped.  The list's first call is always made: the state the patching call left it in
// is not a fixed point (a hit that grew by a patch is not tested again against the hits it had been compared with).
int matesw(const PCtx &c, const Reg &a, int l_ms, const uint8_t *ms, std::vector<Reg> &ma, SwKey key = SwKey{0, 0, 0, 0}, uint32_t mate_read = 0, uint8_t *ma_state = nullptr)        // mem_matesw
{
	if (g_pair_stats) g_ms_calls++;
	const int64_t l_pac = c.x.l_pac;
	int skip[4], n = 0;
	for (int r = 0; r < 4; ++r) skip[r] = c.pes[r].failed ? 1 : 0;
	int n_skip = skip[0] + skip[1] + skip[2] + skip[3];
	for (const Reg &m : ma) {
		if (n_skip == 4) break;                                     // (nothing left to rule out: the rest of the mate's hits change nothing)
		int64_t dist;
		const int r = infer_dir(l_pac, a.rb, m.rb, &dist);
		if (dist >= c.pes[r].low && dist <= c.pes[r].high && !skip[r]) { skip[r] = 1; ++n_skip; }
	}
	if (n_skip == 4) return 0;
	static thread_local std::vector<uint8_t> rev, ref, seqbuf;
	for (int r = 0; r < 4; ++r) {
		if (skip[r]) continue;
		const int is_rev = (r >> 1 != (r & 1)), is_larger = !(r >> 1);
		const uint8_t *seq = ms;
		if (is_rev) {
			rev.resize(l_ms);
			for (int i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
			seq = rev.data();
		}
		int64_t rb, re;
		if (!is_rev) {
			rb = is_larger ? a.rb + c.pes[r].low : a.rb - c.pes[r].high;
			re = (is_larger ? a.rb + c.pes[r].high : a.rb - c.pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a.rb + c.pes[r].low : a.rb - c.pes[r].high) - l_ms;
			re = is_larger ? a.rb + c.pes[r].high : a.rb - c.pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		int rid = -1;
		bool have = false;
		if (rb < re) have = fetch_window(c, &rb, (rb + re) >> 1, &re, &rid, ref, false);
		if (have && a.rid == rid && re - rb >= c.x.co->min_seed_len) {
			if (c.sw_mode == 1 && c.pair_active) c.pair_active[key.pair] = 1;
			const int xtra = BMH_SW_XSUBO | BMH_SW_XSTART | (l_ms * c.x.ep->a < 250 ? BMH_SW_XBYTE : 0) | (c.x.co->min_seed_len * c.x.ep->a);
			bmh_sw_result_t aln = {0, -1, -1, -1, -1, -1, -1};
			bool done = false;
			key.r = (uint8_t)r;
			if (c.sw_mode == 1 && bmh_matesw_device_takes(l_ms, re - rb, xtra)) {       // written down for the batch; this walk goes on as if nothing were found
				bmh_msw_job_t jb; memset(&jb, 0, sizeof(jb));
				jb.rb = rb; jb.re = re; jb.read = mate_read; jb.l_ms = l_ms; jb.is_rev = is_rev; jb.xtra = xtra;
				c.col_keys->push_back(key); c.col_jobs->push_back(jb);
				done = true;
			} else if (c.sw_mode == 2) {
				for (uint64_t t = c.pair_off[key.pair]; t < c.pair_off[key.pair + 1]; ++t)
					if (c.keys[t].i == key.i && c.keys[t].j == key.j && c.keys[t].r == key.r) {
						const int32_t *o = c.res + 7 * t;
						aln.score = o[0]; aln.te = o[1]; aln.qe = o[2]; aln.score2 = o[3]; aln.te2 = o[4]; aln.tb = o[5]; aln.qb = o[6];
						done = true;
						break;
					}
			}
			if (!done && c.sw_mode != 1) {
				ref.resize((size_t)(re - rb));
				for (int64_t t = rb; t < re; ++t) ref[(size_t)(t - rb)] = (uint8_t)text_base(c.x.pac, l_pac, t);
				seqbuf.assign(seq, seq + l_ms);
				aln = bmh_local_sw(l_ms, seqbuf.data(), (int)(re - rb), ref.data(), *c.x.ep, xtra);
				if (g_pair_stats) g_ms_sw++, g_ms_cells += (unsigned long long)l_ms * (unsigned long long)(re - rb);
			}
			if (aln.score >= c.x.co->min_seed_len && aln.qb >= 0) {
				Reg b; memset(&b, 0, sizeof(b));
				b.rid = a.rid; b.is_alt = a.is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score; b.csub = aln.score2; b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				size_t i;
				if (g_pair_stats) g_ms_hits++;
				for (i = 0; i < ma.size(); ++i) if (ma[i].score < b.score) break;     // keep ma sorted by score
				ma.insert(ma.begin() + (long)i, b);
				if (ma_state) *ma_state = 0;
			}
			++n;
		}
		if (n && !(ma_state && (*ma_state & 1))) {
			const unsigned long long t0 = g_pair_prof ? now_ns() : 0;
			const int m = sort_dedup_patch(c.x, nullptr, (int)ma.size(), ma.data()); ma.resize((size_t)m);
			if (ma_state) *ma_state = 1;
			if (g_pair_prof) g_ns_msw_dedup += now_ns() - t0;
		}
	}
	return n;
}

struct P64 { uint64_t x, y; };
inline bool p64_lt(const P64 &a, const P64 &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

int pair_regs(const PCtx &c, std::vector<Reg> *const a[2], int id, int *sub, int *n_sub, int z[2], const int n_pri[2])     // mem_pair
{
	const int64_t l_pac = c.x.l_pac;
	static thread_local std::vector<P64> v, u;                     // (kept by the thread: two allocations a pair were a tenth of the walk)
	v.clear(); u.clear();
	for (int r = 0; r < 2; ++r)
		for (int i = 0; i < n_pri[r]; ++i) {
			const Reg &e = (*a[r])[i];
			P64 key;
			key.x = (uint64_t)(e.rb < l_pac ? e.rb : (l_pac << 1) - 1 - e.rb);
			key.x = (uint64_t)e.rid << 32 | (key.x - (uint64_t)(c.x.n_contigs > 1 ? c.ctg_off[e.rid] : 0));
			key.y = (uint64_t)e.score << 32 | (uint64_t)(i << 2) | (uint64_t)((e.rb >= l_pac) << 1) | (uint64_t)r;
			v.push_back(key);
		}
	std::sort(v.begin(), v.end(), p64_lt);
	int y[4] = {-1, -1, -1, -1};
	for (size_t i = 0; i < v.size(); ++i) {
		for (int r = 0; r < 2; ++r) {
			const int dir = r << 1 | (int)(v[i].y >> 1 & 1);
			if (c.pes[dir].failed) continue;
			const int which = r << 1 | (int)((v[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (int k = y[which]; k >= 0; --k) {
				if ((int)(v[k].y & 3) != which) continue;
				const int64_t dist = (int64_t)v[i].x - (int64_t)v[k].x;
				if (dist > c.pes[dir].high) break;
				if (dist < c.pes[dir].low) continue;
				const double ns = (dist - c.pes[dir].avg) / c.pes[dir].std;
				int q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * c.x.ep->a + .499);
				if (q < 0) q = 0;
				P64 p;
				p.y = (uint64_t)k << 32 | (uint64_t)i;
				p.x = (uint64_t)q << 32 | (hash64(p.y ^ (uint64_t)(id << 8)) & 0xffffffffU);
				u.push_back(p);
			}
		}
		y[v[i].y & 3] = (int)i;
	}
	int ret;
	if (!u.empty()) {
		int tmp = c.x.ep->a + c.x.ep->b;
		tmp = tmp > c.x.ep->o_del + c.x.ep->e_del ? tmp : c.x.ep->o_del + c.x.ep->e_del;
		tmp = tmp > c.x.ep->o_ins + c.x.ep->e_ins ? tmp : c.x.ep->o_ins + c.x.ep->e_ins;
		std::sort(u.begin(), u.end(), p64_lt);
		const size_t i = (size_t)(u.back().y >> 32), k = (size_t)(u.back().y << 32 >> 32);
		z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
		z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
		ret = (int)(u.back().x >> 32);
		*sub = u.size() > 1 ? (int)(u[u.size() - 2].x >> 32) : 0;
		*n_sub = 0;
		for (long j = (long)u.size() - 2; j >= 0; --j) if (*sub - (int)(u[(size_t)j].x >> 32) <= tmp) ++*n_sub;
	} else { ret = 0; *sub = 0; *n_sub = 0; }
	return ret;
}

inline int raw_mapq(int diff, int a) { return (int)(6.02 * diff / a + .499); }

struct ReadOut { std::vector<Reg> regs; std::vector<int> mapq, flag, rep, sec_all; int h; };     // h: record of the read's own alignment, -1 unmapped

// mem_reg2sam's selection for one read (flags without strand; extra = pair flags)
void select_se(const PCtx &c, ReadOut &o, int extra)
{
	const int n = (int)o.regs.size();
	o.mapq.assign(n, 0); o.flag.assign(n, 0); o.rep.assign(n, 0);
	int l = 0, mapq0 = 0;
	for (int k = 0; k < n; ++k) {
		const Reg &p = o.regs[k];
		int mapq = p.secondary < 0 ? approx_mapq(c.x, p) : 0, flag = p.secondary >= 0 ? 0x100 : 0, rep = 1;
		if (p.score < c.x.po->T) rep = 0;
		else if (p.secondary >= 0 && (p.is_alt || !c.x.po->flag_all)) rep = 0;                                    // src/bwamem.c:1742
		else if (p.secondary >= 0 && p.secondary < INT32_MAX && p.score < o.regs[p.secondary].score * c.x.co->drop_ratio) rep = 0;
		if (rep) {
			if (l && p.secondary < 0) flag |= c.x.po->no_multi ? 0x10000 : 0x800;
			if (l && !p.is_alt && mapq > mapq0) mapq = mapq0;                                                        // :1755
			if (l == 0) mapq0 = mapq;
			++l;
			flag |= extra;
		}
		o.mapq[k] = mapq; o.flag[k] = flag; o.rep[k] = rep;
	}
}

int sam_pe(const PCtx &c, uint64_t id, uint32_t r0, ReadOut out[2])          // mem_sam_pe, decisions only; returns extra_flag
{
	std::vector<Reg> *a[2] = {&out[0].regs, &out[1].regs};
	const uint8_t *seq[2] = {c.reads + c.offs[r0], c.reads + c.offs[r0 + 1]};
	const int l_seq[2] = {(int)c.lens[r0], (int)c.lens[r0 + 1]};
	int z[2] = {0, 0}, o, subo = 0, n_sub = 0, extra_flag = 1, n_pri[2];
	if (!c.pe->no_rescue && !(c.sw_mode == 2 && c.pair_active && !c.pair_active[r0 >> 1])) {   // mate rescue for the best regions of each end (src/bwamem_pair.c:273)
		static thread_local std::vector<Reg> b[2];
		for (int i = 0; i < 2; ++i) {
			b[i].clear();
			for (const Reg &r : *a[i]) if (r.score >= (*a[i])[0].score - c.pe->pen_unpaired) b[i].push_back(r);
		}
		const unsigned long long t0 = g_pair_prof && c.sw_mode == 2 ? now_ns() : 0;
		uint8_t ma_state[2] = {0, 0};
		for (int i = 0; i < 2; ++i)
			for (size_t j = 0; j < b[i].size() && (int)j < c.pe->max_matesw; ++j)
				matesw(c, b[i][j], l_seq[!i], seq[!i], *a[!i], SwKey{(uint32_t)(r0 >> 1), (uint16_t)j, (uint8_t)i, 0}, r0 + (uint32_t)!i, &ma_state[!i]);
		if (t0) g_ns_msw += now_ns() - t0;
	}
	if (c.sw_mode == 1) return 0;
	const unsigned long long t_m0 = g_pair_prof ? now_ns() : 0;
	for (int i = 0; i < 2; ++i) n_pri[i] = mark_primary(c.x, (int)a[i]->size(), a[i]->data(), (int64_t)(id << 1 | (uint64_t)i));
	if (g_pair_prof) g_ns_mark += now_ns() - t_m0;
	for (int i = 0; i < 2; ++i) { out[i].sec_all.resize(a[i]->size()); for (size_t j = 0; j < a[i]->size(); ++j) out[i].sec_all[j] = (*a[i])[j].secondary_all; }
	bool paired = false;
	if (!c.pe->no_pairing && n_pri[0] && n_pri[1]) {        // src/bwamem_pair.c:287
		const unsigned long long t_p0 = g_pair_prof ? now_ns() : 0;
		o = pair_regs(c, a, (int)id, &subo, &n_sub, z, n_pri);
		if (g_pair_prof) g_ns_pair += now_ns() - t_p0;
		if (o > 0) {
			int is_multi[2];
			for (int i = 0; i < 2; ++i) {
				int j;
				for (j = 1; j < n_pri[i]; ++j) if ((*a[i])[j].secondary < 0 && (*a[i])[j].score >= c.x.po->T) break;
				is_multi[i] = j < n_pri[i] ? 1 : 0;
			}
			if (!is_multi[0] && !is_multi[1]) {
				paired = true;
				int q_pe, q_se[2];
				const int score_un = (*a[0])[0].score + (*a[1])[0].score - c.pe->pen_unpaired;
				subo = subo > score_un ? subo : score_un;
				q_pe = raw_mapq(o - subo, c.x.ep->a);
				if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
				if (q_pe < 0) q_pe = 0;
				if (q_pe > 60) q_pe = 60;
				q_pe = (int)(q_pe * (1. - .5 * ((*a[0])[0].frac_rep + (*a[1])[0].frac_rep)) + .499);
				if (o > score_un) {
					Reg *cc[2] = {&(*a[0])[z[0]], &(*a[1])[z[1]]};
					for (int i = 0; i < 2; ++i) {
						if (cc[i]->secondary >= 0) { cc[i]->sub = (*a[i])[cc[i]->secondary].score; cc[i]->secondary = -2; }
						q_se[i] = approx_mapq(c.x, *cc[i]);
					}
					q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
					q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
					extra_flag |= 2;
					q_se[0] = q_se[0] < raw_mapq(cc[0]->score - cc[0]->csub, c.x.ep->a) ? q_se[0] : raw_mapq(cc[0]->score - cc[0]->csub, c.x.ep->a);
					q_se[1] = q_se[1] < raw_mapq(cc[1]->score - cc[1]->csub, c.x.ep->a) ? q_se[1] : raw_mapq(cc[1]->score - cc[1]->csub, c.x.ep->a);
				} else {
					z[0] = z[1] = 0;
					q_se[0] = approx_mapq(c.x, (*a[0])[0]);
					q_se[1] = approx_mapq(c.x, (*a[1])[0]);
				}
				for (int i = 0; i < 2; ++i) {             // the chosen region becomes the primary of its group (for the XA tag)
					const int k = out[i].sec_all[z[i]];
					if (k >= 0 && k < n_pri[i]) {
						for (size_t j = 0; j < a[i]->size(); ++j) if (out[i].sec_all[j] == k || (int)j == k) out[i].sec_all[j] = z[i];
						out[i].sec_all[z[i]] = -1;
					}
				}
				for (int i = 0; i < 2; ++i) {
					const int n = (int)a[i]->size();
					out[i].mapq.assign(n, 0); out[i].flag.assign(n, 0); out[i].rep.assign(n, 0);
					out[i].rep[z[i]] = 1; out[i].mapq[z[i]] = q_se[i];
					out[i].flag[z[i]] = ((*a[i])[z[i]].secondary >= 0 ? 0x100 : 0) | 0x40 << i | extra_flag;
					out[i].h = z[i];
					if (n_pri[i] < n) {                          // the read has ALT hits: the best of them is written as a supplementary record (src/bwamem_pair.c:349-356)
						const Reg &p = (*a[i])[n_pri[i]];
						if (!(p.score < c.x.po->T || p.secondary >= 0 || !p.is_alt)) {
							out[i].rep[n_pri[i]] = 1; out[i].mapq[n_pri[i]] = approx_mapq(c.x, p);
							out[i].flag[n_pri[i]] = 0x800 | 0x40 << i | extra_flag;
						}
					}
				}
			}
		}
	}
	if (!paired) {
		// the alignment a read shows its mate (src/bwamem_pair.c:376-385): its best hit, or -- when that one (the best of the primary assembly, which the second
		// marking round has put first) is below the threshold -- its best ALT hit; the orientation test below still takes the FIRST hits' positions (:389)
		int hh[2];
		for (int i = 0; i < 2; ++i) {
			hh[i] = -1;
			if (!a[i]->empty()) {
				if ((*a[i])[0].score >= c.x.po->T) hh[i] = 0;
				else if (n_pri[i] < (int)a[i]->size() && (*a[i])[(size_t)n_pri[i]].score >= c.x.po->T) hh[i] = n_pri[i];
			}
		}
		if (!c.pe->no_pairing && hh[0] >= 0 && hh[1] >= 0 && (*a[0])[(size_t)hh[0]].rid == (*a[1])[(size_t)hh[1]].rid) {      // src/bwamem_pair.c:386
			int64_t dist;
			const int d = infer_dir(c.x.l_pac, (*a[0])[0].rb, (*a[1])[0].rb, &dist);
			if (!c.pes[d].failed && dist >= c.pes[d].low && dist <= c.pes[d].high) extra_flag |= 2;
		}
		for (int i = 0; i < 2; ++i) { select_se(c, out[i], (i ? 0x81 : 0x41) | extra_flag); out[i].h = hh[i]; }
	}
	return extra_flag;
}

} // namespace

extern "C" void bmh_pe_opt_default(bmh_pe_opt_t *o) { o->pen_unpaired = 17; o->max_ins = 10000; o->max_matesw = 50; o->no_rescue = 0; o->no_pairing = 0; }

// Interleaved pairs (read 2i, 2i+1).  out[..][16] as bmh_finalize_regs (flag carries the pair bits 0x1 0x2 0x40 0x80 too,
// [12] = the record's primary for the XA tag or -1); out_h[r] = record of read r's own alignment within its list (what the
// mate's RNEXT / PNEXT / TLEN are taken from) or -1 if unmapped; out_unflag[r] = flag bits of the unmapped record of a read
// without reported alignment; pes_out[4][5] = {low, high, failed, avg, std} as doubles.  Returns the record count (<= cap).
// What a call needs in large arrays, kept by the calling THREAD between calls: the regions of a million reads are 400 MB, the records another 250 --
// allocated, faulted in and unmapped per call they cost a third of the call.  Freed when the thread ends.
namespace {
struct PairPart { std::vector<int32_t> rec; std::vector<uint32_t> n, pairs; std::vector<int32_t> h, uf; };
struct PairScratch {
	Reg *flat = nullptr; size_t flat_cap = 0;
	std::vector<PairPart> parts;
	std::vector<std::vector<SwKey>> tk; std::vector<std::vector<bmh_msw_job_t>> tj;
	std::vector<SwKey> all_keys; std::vector<bmh_msw_job_t> all_jobs; std::vector<uint64_t> pair_off, in_off; std::vector<int32_t> sw_res; std::vector<uint8_t> pair_active;
	std::vector<uint32_t> cnt, off32;
	~PairScratch() { free(flat); }
};
thread_local PairScratch g_pair_scratch;
}

static int64_t finalize_pairs_impl(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                                   int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                   const uint32_t *read_lens, const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                                   int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                                   int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, double *pes_out,
                                   int n_threads, const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, void *stream, bool deduped = false,
                                   bmh_pairs_split_t *split = nullptr)
{
	if (!copt || !ep || !popt || !pe || !pac || !reads || !read_offs || !read_lens || !regs_per_read || !out || !out_per_read || !out_h || !out_unflag ||
	    (n_contigs > 1 && (!contig_offset || !contig_len))) { bmh_set_error("bmh_finalize_pairs: null argument"); return BMH_EINVAL; }
	if (n_reads & 1) { bmh_set_error("bmh_finalize_pairs: odd number of reads (pairs are interleaved)"); return BMH_EINVAL; }
	if (!(popt->mapQ_coef_len > 0)) { bmh_set_error("bmh_finalize_pairs: mapQ_coef_len <= 0 is not restated"); return BMH_EINVAL; }
	const double t_in = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	PCtx c;
	c.x = {copt, ep, popt, l_pac, pac, n_contigs, contig_offset};
	c.pe = pe; c.ctg_off = contig_offset; c.ctg_len = contig_len; c.reads = reads; c.offs = read_offs; c.lens = read_lens;
	// (a caller that outlives its threads -- the lanes of bmh_aligner_run -- keeps the scratch itself: split->scratch_slot)
	if (split && split->scratch_slot && !*split->scratch_slot) *split->scratch_slot = new PairScratch();
	PairScratch &S = (split && split->scratch_slot) ? *(PairScratch *)*split->scratch_slot : g_pair_scratch;
	std::vector<uint64_t> &in_off = S.in_off;
	in_off.assign((size_t)n_reads + 1, 0);
	for (uint32_t r = 0; r < n_reads; ++r) in_off[r + 1] = in_off[r] + regs_per_read[r];
	// regions of all reads in one array (no per-read heap traffic: that, not the arithmetic, dominated on many threads)
	if ((size_t)in_off[n_reads] + 1 > S.flat_cap) {                  // not zeroed: reg_from_record sets every field
		free(S.flat);
		S.flat_cap = (size_t)in_off[n_reads] + (size_t)in_off[n_reads] / 4 + 1024;
		S.flat = (Reg *)malloc(sizeof(Reg) * S.flat_cap);
		if (!S.flat) { S.flat_cap = 0; bmh_set_error("bmh_finalize_pairs: out of memory"); return BMH_ENOMEM; }
	}
	Reg *const flat = S.flat;
	std::vector<uint32_t> &cnt = S.cnt;
	cnt.assign(n_reads, 0);
	if (n_threads < 1) n_threads = 1;
	if ((uint32_t)n_threads > n_reads / 2 + 1) n_threads = (int)(n_reads / 2 + 1);
	auto par = [&](auto fn, uint32_t n_units) {
		if (n_threads == 1 || n_units < 2) { fn(0, 0u, n_units); return; }
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t) th.emplace_back(fn, t, (uint32_t)((uint64_t)n_units * t / n_threads), (uint32_t)((uint64_t)n_units * (t + 1) / n_threads));
		for (auto &t : th) t.join();
	};
	const bool prof = getenv("BMH_PAIR_PROFILE") != nullptr;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t_a = now();
	par([&](int, uint32_t r0, uint32_t r1) {                    // per read: mem_sort_dedup_patch, in place
		for (uint32_t r = r0; r < r1; ++r) {
			const int n_in = (int)regs_per_read[r];
			Reg *a = flat + in_off[r];
			if (deduped) {
				// records of bmh_dedup_regs_device: mem_sort_dedup_patch has run on the device; [1..9] as a region leaves it, [13] its sequence
				for (int i = 0; i < n_in; ++i) {
					const int32_t *g = regs_in + 16 * (in_off[r] + i);
					Reg &p = a[i];
					memset(&p, 0, sizeof(p));
					p.score = g[1]; p.qb = g[2]; p.qe = g[3]; p.rb = (int64_t)(uint32_t)g[4] | (int64_t)g[5] << 32; p.re = (int64_t)(uint32_t)g[6] | (int64_t)g[7] << 32;
					p.truesc = g[8]; p.w = g[9]; p.rid = g[13]; p.secondary = -1; p.frac_rep = frac_rep ? frac_rep[r] : 0.f;
				}
				cnt[r] = (uint32_t)n_in;
				set_is_alt(c.x, n_in, a);
				continue;
			}
			for (int i = 0; i < n_in; ++i) reg_from_record(c.x, regs_in + 8 * (in_off[r] + i), frac_rep ? frac_rep[r] : 0.f, a[i]);
			cnt[r] = (uint32_t)sort_dedup_patch(c.x, reads + read_offs[r], n_in, a);
			set_is_alt(c.x, (int)cnt[r], a);                            // src/bwamem.c:2321-2325
		}
	}, n_reads);
	const double t_b = now();
	pestat(c, n_reads, [&](size_t r) { return Span{flat + in_off[r], cnt[r]}; }, n_threads);
	const double t_c = now();
	if (split && split->after_pestat) {
		double pv[20];
		for (int d = 0; d < 4; ++d) { pv[5 * d] = c.pes[d].low; pv[5 * d + 1] = c.pes[d].high; pv[5 * d + 2] = c.pes[d].failed; pv[5 * d + 3] = c.pes[d].avg; pv[5 * d + 4] = c.pes[d].std; }
		const int rc = split->after_pestat(split->user, pv);
		if (rc != BMH_OK) return rc;
	}
	if (pes_out) for (int d = 0; d < 4; ++d) { pes_out[5 * d] = c.pes[d].low; pes_out[5 * d + 1] = c.pes[d].high; pes_out[5 * d + 2] = c.pes[d].failed; pes_out[5 * d + 3] = c.pes[d].avg; pes_out[5 * d + 4] = c.pes[d].std; }
	// With a device: the local alignments of the mate rescue as one batch (pair_kernels.hip).  First which alignments mem_matesw asks for -- a walk of
	// all pairs here, or a kernel where the regions are on the device too --, then the kernel that computes them; the walk below takes the results.
	std::vector<SwKey> &all_keys = S.all_keys; std::vector<bmh_msw_job_t> &all_jobs = S.all_jobs; std::vector<uint64_t> &pair_off = S.pair_off; std::vector<int32_t> &sw_res = S.sw_res;
	std::vector<uint8_t> &pair_active = S.pair_active;
	all_keys.clear(); all_jobs.clear();
	double t_sw0 = now(), t_sw1 = t_sw0, t_sw2 = t_sw0;
	// the host's first walk: the alignments every pair's mem_matesw calls ask for (ranges of pairs on threads; their lists concatenate in pair order)
	auto first_walk = [&](std::vector<SwKey> &keys_out, std::vector<bmh_msw_job_t> &jobs_out, std::vector<uint8_t> &active_out) {
		std::vector<std::vector<SwKey>> &tk = S.tk; std::vector<std::vector<bmh_msw_job_t>> &tj = S.tj;
		if (tk.size() < (size_t)n_threads) { tk.resize((size_t)n_threads); tj.resize((size_t)n_threads); }
		for (auto &v : tk) v.clear();
		for (auto &v : tj) v.clear();
		active_out.assign((size_t)n_reads / 2 + 1, 0);
		par([&](int t, uint32_t p0, uint32_t p1) {
			PCtx cl = c;
			cl.sw_mode = 1; cl.col_keys = &tk[(size_t)t]; cl.col_jobs = &tj[(size_t)t];
			cl.pair_active = active_out.data();                     // (a pair belongs to one thread: plain bytes)
			ReadOut o2[2];
			for (uint32_t p = p0; p < p1; ++p) {
				for (int i = 0; i < 2; ++i) { const uint32_t r = 2 * p + (uint32_t)i; o2[i].regs.assign(flat + in_off[r], flat + in_off[r] + cnt[r]); }
				sam_pe(cl, (uint64_t)(popt->id0 / 2) + p, 2 * p, o2);
			}
		}, n_reads / 2);
		keys_out.clear(); jobs_out.clear();
		for (int t = 0; t < n_threads; ++t) { keys_out.insert(keys_out.end(), tk[(size_t)t].begin(), tk[(size_t)t].end()); jobs_out.insert(jobs_out.end(), tj[(size_t)t].begin(), tj[(size_t)t].end()); }
	};
	if (idx && !pe->no_rescue && n_reads && split && split->rescue_in) {
		// the windows of every pair's mem_matesw calls found by a kernel on the regions the device kept, aligned there, keys and results copied back
		double pv[20];
		for (int d = 0; d < 4; ++d) { pv[5 * d] = c.pes[d].low; pv[5 * d + 1] = c.pes[d].high; pv[5 * d + 2] = c.pes[d].failed; pv[5 * d + 3] = c.pes[d].avg; pv[5 * d + 4] = c.pes[d].std; }
		pair_active.assign((size_t)n_reads / 2 + 1, 0);
		S.off32.assign((size_t)n_reads / 2 + 1, 0);
		const int64_t nj = bmh_rescue_count_device(idx, split->rescue_in, ep, copt->min_seed_len, pe, pv, n_reads, S.off32.data(), pair_active.data(), stream);
		if (nj < 0) return nj;
		t_sw1 = now();
		all_keys.resize((size_t)nj); all_jobs.resize((size_t)nj); sw_res.resize(7 * (size_t)nj + 7);       // (all_jobs: its size is what the profile line prints)
		const int rc = bmh_rescue_run_device(idx, d_reads, d_offs, ep, all_keys.data(), sw_res.data(), stream);
		if (rc != BMH_OK) return rc;
		pair_off.resize((size_t)n_reads / 2 + 1);
		for (size_t p = 0; p <= (size_t)n_reads / 2; ++p) pair_off[p] = S.off32[p];
		t_sw2 = now();
		if (bmh_tune("RESCUE_CHECK", 0) != 0) {
			// the host's walk beside it -- the same pairs active (always), the same calls asked for in the same order (unless a
			// mem_sort_dedup_patch call of the walk changed a list: counted apart)
			std::vector<SwKey> hk; std::vector<bmh_msw_job_t> hj; std::vector<uint8_t> ha;
			first_walk(hk, hj, ha);
			unsigned long long bad_active = 0, bad_lists = 0;
			std::vector<uint64_t> ho((size_t)n_reads / 2 + 1, 0);
			for (const SwKey &k : hk) ++ho[(size_t)k.pair + 1];
			for (size_t p = 0; p < (size_t)n_reads / 2; ++p) ho[p + 1] += ho[p];
			for (size_t p = 0; p < (size_t)n_reads / 2; ++p) {
				bad_active += ha[p] != pair_active[p];
				bool same = ho[p + 1] - ho[p] == pair_off[p + 1] - pair_off[p];
				for (uint64_t t = 0; same && t < ho[p + 1] - ho[p]; ++t) {
					const SwKey &x = hk[ho[p] + t], &y = all_keys[pair_off[p] + t];
					same = x.pair == y.pair && x.j == y.j && x.i == y.i && x.r == y.r;
				}
				bad_lists += !same;
			}
			g_chk_batches++; g_chk_pairs += n_reads / 2; g_chk_jobs += (unsigned long long)nj; g_chk_active += bad_active; g_chk_lists += bad_lists;
		}
		c.pair_active = pair_active.data();
		c.sw_mode = 2; c.keys = all_keys.data(); c.pair_off = pair_off.data(); c.res = sw_res.data();
	} else if (idx && !pe->no_rescue && n_reads) {
		first_walk(all_keys, all_jobs, pair_active);
		c.pair_active = pair_active.data();
		pair_off.assign((size_t)n_reads / 2 + 1, 0);
		for (const SwKey &k : all_keys) ++pair_off[(size_t)k.pair + 1];
		for (size_t p = 0; p < (size_t)n_reads / 2; ++p) pair_off[p + 1] += pair_off[p];
		t_sw1 = now();
		sw_res.resize(7 * all_jobs.size() + 7);
		const int rc = bmh_matesw_batch_device(idx, d_reads, d_offs, ep, all_jobs.data(), all_jobs.size(), sw_res.data(), stream);
		if (rc != BMH_OK) return rc;
		t_sw2 = now();
		c.sw_mode = 2; c.keys = all_keys.data(); c.pair_off = pair_off.data(); c.res = sw_res.data();
	}
	// split mode: the device has made the records of the pairs the rescue does not touch (csrc/pair_dev.hip); the walk below takes the others -- the pairs
	// a mem_matesw call got a window for, and the ones the device hands back (`extra`: a lane's worth of hits exceeded, a score too close to an
	// integer to trust the device's erfc / log) -- and writes their records, and only theirs, in pair order
	const uint8_t *todo_extra = nullptr;
	if (split) {
		if (split->before_final) { const int rc = split->before_final(split->user, &todo_extra); if (rc != BMH_OK) return rc; }
		if (!c.pair_active) { pair_active.assign((size_t)n_reads / 2 + 1, 0); c.pair_active = pair_active.data(); }     // (no rescue: no pair is active, and the walk does not ask)
	}
	auto is_todo = [&](uint32_t p) { return !split || pair_active[p] || (todo_extra && todo_extra[p]); };
	// per pair: mem_sam_pe's decisions; every thread appends the records of its (contiguous) pairs to its own buffer
	typedef PairPart Part;
	std::vector<Part> &parts = S.parts;
	parts.resize((size_t)n_threads);
	for (Part &P : parts) { P.rec.clear(); P.n.clear(); P.h.clear(); P.uf.clear(); P.pairs.clear(); }
	std::vector<double> th_ms((size_t)n_threads, 0.0); const double t_w0 = now();
	par([&](int t, uint32_t p0, uint32_t p1) {
		struct Stamp { double *o; double t0; ~Stamp() { *o = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; } } stamp{&th_ms[(size_t)t], t_w0};
		Part &P = parts[(size_t)t];
		if (!split) P.rec.reserve((size_t)(in_off[2 * p1] - in_off[2 * p0]) * 16 + 64);
		ReadOut o2[2];
		for (uint32_t p = p0; p < p1; ++p) {
			if (!is_todo(p)) continue;
			if (split) P.pairs.push_back(p);
			for (int i = 0; i < 2; ++i) {
				const uint32_t r = 2 * p + (uint32_t)i;
				o2[i].regs.assign(flat + in_off[r], flat + in_off[r] + cnt[r]);
				o2[i].mapq.clear(); o2[i].flag.clear(); o2[i].rep.clear(); o2[i].sec_all.clear(); o2[i].h = -1;
			}
			const int extra = sam_pe(c, (uint64_t)(popt->id0 / 2) + p, 2 * p, o2);
			for (int i = 0; i < 2; ++i) {
				const ReadOut &o = o2[i];
				const uint32_t r = 2 * p + (uint32_t)i;
				bool any = false;
				for (size_t k = 0; k < o.regs.size(); ++k) {
					const Reg &g = o.regs[k];
					const size_t at = P.rec.size();
					P.rec.resize(at + 16);
					int32_t *q = P.rec.data() + at;
					q[0] = (int32_t)r; q[1] = g.score; q[2] = g.qb; q[3] = g.qe;
					q[4] = (int32_t)(uint32_t)g.rb; q[5] = (int32_t)(g.rb >> 32); q[6] = (int32_t)(uint32_t)g.re; q[7] = (int32_t)(g.re >> 32);
					q[8] = g.truesc; q[9] = g.w;            /* 0 / 0 for a rescued region (src/bwamem_pair.c:161: memset) */
					q[10] = g.sub > g.csub ? g.sub : g.csub;
					q[12] = o.sec_all.empty() ? g.secondary : o.sec_all[k];
					q[11] = alt_mode(c.x) ? q[12] : g.sub_n;
					q[13] = o.mapq[k]; q[14] = o.flag[k]; q[15] = o.rep[k] | (g.is_alt ? 2 : 0) | (g.alt_sc > 0 ? g.alt_sc << 2 : 0);
					any = any || o.rep[k];
				}
				P.n.push_back((uint32_t)o.regs.size()); P.h.push_back(o.h);
				// the unmapped record of a read without reported alignment carries the flags of its mem_reg2sam call
				P.uf.push_back(any ? 0 : ((i ? 0x81 : 0x41) | extra));
			}
		}
	}, n_reads / 2);
	const double t_d = now();
	if (prof) { double mn = 1e30, mx = 0, sm = 0; for (double v : th_ms) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; sm += v; }
	            fprintf(stderr, "[pairs] second walk: threads done after min %.1f / mean %.1f / max %.1f ms of its start (before it: %.1f ms waiting for the device's pairs)\n", mn, sm / th_ms.size(), mx, t_w0 - t_sw2); }
	if (prof) fprintf(stderr, "[pairs] setup %.1f ms, dedup %.1f ms, pestat %.1f ms, rescue jobs collected %.1f ms (%zu), on the device %.1f ms, mem_sam_pe %.1f ms (%d threads)\n", t_a - t_in, t_b - t_a, t_c - t_b, t_sw1 - t_sw0, all_jobs.size(), t_sw2 - t_sw1, t_d - t_sw2, n_threads);
	// the parts go out side by side: offsets first, then every thread copies its own part
	std::vector<uint64_t> w_off(parts.size() + 1, 0), r_off(parts.size() + 1, 0);
	for (size_t t = 0; t < parts.size(); ++t) { w_off[t + 1] = w_off[t] + parts[t].rec.size() / 16; r_off[t + 1] = r_off[t] + parts[t].n.size(); }
	if (w_off[parts.size()] > cap) { bmh_set_error("bmh_finalize_pairs: more than %llu output regions", (unsigned long long)cap); return BMH_ECAPACITY; }
	par([&](int, uint32_t t0, uint32_t t1) {
		for (uint32_t t = t0; t < t1; ++t) {
			const Part &P = parts[t];
			if (!P.rec.empty()) memcpy(out + 16 * w_off[t], P.rec.data(), sizeof(int32_t) * P.rec.size());
			uint64_t r = r_off[t];
			for (size_t k = 0; k < P.n.size(); ++k, ++r) { out_per_read[r] = P.n[k]; out_h[r] = P.h[k]; out_unflag[r] = P.uf[k]; }
			if (split && split->todo_pairs) for (size_t k = 0; k < P.pairs.size(); ++k) split->todo_pairs[r_off[t] / 2 + k] = P.pairs[k];
		}
	}, (uint32_t)parts.size());
	if (split) split->n_todo = r_off[parts.size()] / 2;
	if (prof) fprintf(stderr, "[pairs] second walk, summed over the threads: mem_matesw %.1f ms (of which mem_sort_dedup_patch %.1f), mem_mark_primary_se %.1f, mem_pair %.1f\n",
	                  g_ns_msw.exchange(0) / 1e6, g_ns_msw_dedup.exchange(0) / 1e6, g_ns_mark.exchange(0) / 1e6, g_ns_pair.exchange(0) / 1e6);
	if (prof) fprintf(stderr, "[pairs] records put in place %.1f ms; the whole call %.1f ms\n", now() - t_d, now() - t_in);
	if (getenv("BMH_POST_STATS"))
		fprintf(stderr, "[finalize_pairs] %u reads: mem_matesw calls %llu, local alignments on the host %llu (%.0f cells each), rescued regions %llu\n", n_reads,
		        (unsigned long long)g_ms_calls.exchange(0), (unsigned long long)g_ms_sw.load(), (double)g_ms_cells.exchange(0) / (double)(g_ms_sw.load() ? g_ms_sw.load() : 1),
		        (unsigned long long)g_ms_hits.exchange(0)), g_ms_sw = 0;
	return (int64_t)w_off[parts.size()];
}

extern "C" int64_t bmh_finalize_pairs(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                                      int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                      const uint32_t *read_lens, const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                                      int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                                      int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, double *pes_out,
                                      int n_threads)
{
	return finalize_pairs_impl(copt, ep, popt, pe, l_pac, pac, n_reads, reads, read_offs, read_lens, regs_in, regs_per_read, frac_rep, n_contigs, contig_offset, contig_len,
	                           out, cap, out_per_read, out_h, out_unflag, pes_out, n_threads, nullptr, nullptr, nullptr, nullptr);
}

// The same with the local alignments of the mate rescue computed on the device (idx: the index with its 2-bit reference in HBM;
// d_reads / d_offs: the batch's ASCII reads in HBM, the same reads as `reads`; stream: a HIP stream, waited for).  Same results.
extern "C" int64_t bmh_finalize_pairs_dev(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, void *stream,
                                          const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                                          int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                          const uint32_t *read_lens, const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                                          int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                                          int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, double *pes_out,
                                          int n_threads)
{
	if (!idx || !d_reads || !d_offs) { bmh_set_error("bmh_finalize_pairs_dev: null argument"); return BMH_EINVAL; }
	return finalize_pairs_impl(copt, ep, popt, pe, l_pac, pac, n_reads, reads, read_offs, read_lens, regs_in, regs_per_read, frac_rep, n_contigs, contig_offset, contig_len,
	                           out, cap, out_per_read, out_h, out_unflag, pes_out, n_threads, idx, d_reads, d_offs, stream);
}

// The same from regions mem_sort_dedup_patch has been through already ON THE DEVICE: dedup_recs[..][16] / dedup_per_read as bmh_dedup_regs_device left them
// (copied to the host).  Same results as bmh_finalize_pairs_dev on the regions they were made from.
extern "C" int64_t bmh_finalize_pairs_deduped(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, void *stream,
                                              const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                                              int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                              const uint32_t *read_lens, const int32_t *dedup_recs, const uint32_t *dedup_per_read, const float *frac_rep,
                                              int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                                              int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, double *pes_out,
                                              int n_threads)
{
	if (!idx || !d_reads || !d_offs) { bmh_set_error("bmh_finalize_pairs_deduped: null argument"); return BMH_EINVAL; }
	return finalize_pairs_impl(copt, ep, popt, pe, l_pac, pac, n_reads, reads, read_offs, read_lens, dedup_recs, dedup_per_read, frac_rep, n_contigs, contig_offset, contig_len,
	                           out, cap, out_per_read, out_h, out_unflag, pes_out, n_threads, idx, d_reads, d_offs, stream, true);
}

// bmh_finalize_pairs_deduped for the pairs the device does not finish itself (csrc/pair_dev.hip, csrc/align_pipeline.hip): see bmh_pairs_split_t.
// out / out_per_read / out_h / out_unflag are COMPACT: the reads of the pairs listed in split->todo_pairs, in that order.
int64_t bmh_finalize_pairs_split(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, void *stream,
                                 const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                                 int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                 const uint32_t *read_lens, const int32_t *dedup_recs, const uint32_t *dedup_per_read, const float *frac_rep,
                                 int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                                 int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, int n_threads, bmh_pairs_split_t *split)
{
	if (!idx || !d_reads || !d_offs || !split) { bmh_set_error("bmh_finalize_pairs_split: null argument"); return BMH_EINVAL; }
	return finalize_pairs_impl(copt, ep, popt, pe, l_pac, pac, n_reads, reads, read_offs, read_lens, dedup_recs, dedup_per_read, frac_rep, n_contigs, contig_offset, contig_len,
	                           out, cap, out_per_read, out_h, out_unflag, nullptr, n_threads, idx, d_reads, d_offs, stream, true, split);
}

void bmh_pairs_scratch_free(void *p) { delete (PairScratch *)p; }

// With the knob RESCUE_CHECK set every batch whose rescue windows the device found was also walked by the host: out[5] = batches, pairs, jobs of the
// device, pairs whose `active` flag differs (must be 0), pairs whose list of calls differs (possible where a mem_sort_dedup_patch of the walk changed a list).
extern "C" void bmh_rescue_check_counts(uint64_t *out)
{
	out[0] = g_chk_batches.load(); out[1] = g_chk_pairs.load(); out[2] = g_chk_jobs.load(); out[3] = g_chk_active.load(); out[4] = g_chk_lists.load();
}
