// Per-read chaining core of the device job builder: seeds -> chains -> filtered chains -> alignment regions with
// their LEFT/RIGHT extension-job geometry.  Restates, for one read and with flat scratch arrays instead of
// kbtree/kvec, the reference's host stage between its two kernels:
//   mem_chain          /root/reference/src/bwamem.c:404-477  (+ test_and_merge :337-364)
//   mem_chain_weight   :366-392      mem_chain_flt  :487-559  (ks_introsort(mem_flt), src/ksort.h:146-226)
//   mem_chain2aln      :1170-1479    (which seeds become regions; job geometry :1300-1434)
//   cal_max_gap        :996-1002     bns_pos2rid / bns_intv2rid  src/bntseq.c:349-373
// Same decisions in the same order as csrc/host_jobs.cpp (which is pinned to the reference's own host code), so the
// two produce identical batches.  Floating point appears exactly where the reference has it (double in cal_max_gap
// and the 0.85/0.95/0.1 factors, float in mask_level/drop_ratio); there is no multiply-add for the compiler to fuse.
//
// The function is templated on COOP: false = one lane (or one host thread) does everything serially; true = all 64
// lanes of a wave execute it redundantly on one read (same addresses, same values) and the loops that are
// quadratic in the number of seeds of a read -- the sorted insert, the kept-chain scan, the "already covered by a
// region" scan -- are split across the lanes.  Reads with few seeds (almost all) take the lane form, the rare read
// inside a high-copy repeat takes the wave form.
//
// Compiles as plain C++ too (tests/chain_core_host.cpp) so the logic is testable without a GPU.
#pragma once
#include <math.h>
#include <stdint.h>
#include "../../include/bwamem_hip.h"

#if defined(__HIPCC__)
#define CH_HD __host__ __device__
// (the per-read core is inlined into every kernel call site: only then does each scratch pointer have ONE address space the compiler can see -- an
// outlined copy shared by call sites with LDS and with global scratch makes every access a flat one, several times the latency of ds_read on the LDS side;
// round 6 found the wave kernel's hybrid and global-scratch forms compiled that way: 279 flat loads)
#define CH_INLINE __attribute__((always_inline))
#else
#define CH_HD
#define CH_INLINE
#endif

struct ch_seed_t { int64_t rbeg; int32_t qbeg, len; uint32_t next, pad; };                       // 24 B
struct ch_chain_t { uint32_t head, tail, n; int32_t rid, w, first, beg, end; uint32_t kept, pad; };   // 40 B
struct ch_reg_t { int64_t seed_rbeg, rmax0; int32_t seed_qbeg, seedlen0, lr, rr, rq, pad; };        // 40 B: region + job geometry
struct ch_est_t { int64_t rb_est, re_est; int32_t qb_est, qe_est, seedlen0, pad; };                // 32 B: what the "covered already" test reads

// The same records in COMPACT form for scratch that lives in LDS (round 6): the cooperative kernels' classes hold at most 1860 entries of a read of at most
// CH_MAX_READ_LEN bases, so list links, chain fields and read coordinates fit 16 bits -- 82 instead of 124 bytes per entry (54 instead of 84 in the hybrid
// classes), and LDS bytes x residency is what bounds the stage.  The core is written against a TYPE POLICY (ch_wide_ty / ch_compact_ty): same field names,
// same code, only the widths and the end-of-list mark differ.  The forms with the reference's seed filter (FLT: a seed's re-scored weight rides in `pad`),
// the lane forms and the global-scratch class (more than 65 535 entries are possible there) keep the wide records.
struct ch_seed_c { int64_t rbeg; int16_t qbeg, len; uint16_t next; int16_t pad; };                                        // 16 B
struct ch_chain_c { uint16_t head, tail, n; int16_t first; int32_t rid; uint16_t w; int16_t beg, end; uint8_t kept, pad; };   // 20 B
struct ch_est_c { int64_t rb_est, re_est; int16_t qb_est, qe_est, seedlen0, pad; };                                      // 24 B
struct ch_wide_ty { typedef ch_seed_t seed_t; typedef ch_chain_t chain_t; typedef ch_est_t est_t; typedef uint32_t idx_t; static constexpr uint32_t NIL = 0xFFFFFFFFu; };
struct ch_compact_ty { typedef ch_seed_c seed_t; typedef ch_chain_c chain_t; typedef ch_est_c est_t; typedef uint16_t idx_t; static constexpr uint32_t NIL = 0xFFFFu; };

// per-read scratch (a read never needs more entries than it has seeds): global memory, or LDS for a wave-form read
template <class TY> struct ch_scr {
	typedef TY ty;
	typename TY::seed_t *S; typename TY::chain_t *CH; typename TY::idx_t *order; int64_t *opos; typename TY::idx_t *klist; uint64_t *srt; typename TY::idx_t *cidx; typename TY::est_t *E;
};
typedef ch_scr<ch_wide_ty> ch_scr_t;

struct ch_ctx_t {
	bmh_chain_opt_t o;
	int64_t l_pac; int n_contigs; const int64_t *ctg_off; const int32_t *ctg_len;
	const uint8_t *ctg_alt;       // per sequence: is it an ALT contig (bns->anns[rid].is_alt, read from the .alt file, src/bntseq.c:179-200)?  null: none is
	const uint64_t *rbeg; const int32_t *qbeg; const uint32_t *score, *n_ref, *prefix;   // mem_seed_v_gpu arrays
	const uint32_t *read_lens;
	// the seed filter (FLT forms of chain_read only): the reads (letters or nt4 codes, both are understood), their offsets, the 2-bit reference
	const uint8_t *reads; const uint32_t *read_offs; const uint8_t *pac;
	// global scratch, every array indexed by prefix[read] + local index
	ch_scr_t g;
	ch_reg_t *regs;               // output slots, prefix[read] + i in creation order
	uint32_t *regs_per_read, *jobs_per_read;
	float *frac_rep;              // per read: part of the read covered by SMEMs with more than max_occ occurrences (mem_chain :415-459)
	int *err;                     // 1: a read longer than CH_MAX_READ_LEN, or one the reference's seed filter applies to in a form compiled without it
	long long *prof; uint32_t prof_read;
};

#define CH_MAX_READ_LEN 700
// tmp.is_alt of mem_chain (src/bwamem.c:446): a function of the chain's sequence
CH_HD inline uint32_t chain_is_alt(const ch_ctx_t &x, int rid) { return (x.ctg_alt && rid >= 0 && x.ctg_alt[rid]) ? 1u : 0u; }

// optional phase stamps (cycles) of one read, for tuning: compile with -DCH_PROFILE
#if defined(CH_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define CH_STAMP(i) do { if (x.prof && r == x.prof_read) x.prof[i] = (long long)wall_clock64(); } while (0)
// (accumulators of the occurrence batches of mem_chain: ticks into slot t, a count into the low / high half of slot c)
#define CH_ACC_BEGIN() const long long acc_t0_ = (long long)wall_clock64()
#define CH_ACC_END(t, c) do { if (x.prof && r == x.prof_read && ch_lane() == 0) { x.prof[t] += (long long)wall_clock64() - acc_t0_; x.prof[c] += 1; } } while (0)
#else
#define CH_STAMP(i) do { } while (0)
#define CH_ACC_BEGIN() do { } while (0)
#define CH_ACC_END(t, c) do { } while (0)
#endif

namespace chain_core {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int ch_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// Lanes of a wave exchange data through the read's scratch.  With the scratch in global memory the writes must have completed
// before other lanes read them (vmcnt = 0); with ALL of it in LDS (template flag LDSX) it is enough to order the LDS operations
// (lgkmcnt = 0) -- and the wave no longer waits for its output stores to global memory (one region per chain: ~1 us each).
template <bool LDSX = false> __device__ __forceinline__ void ch_wave_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_s_waitcnt(LDSX ? 0xC07F : 0);            // gfx9 encoding: vmcnt 63 / expcnt 7 left alone, lgkmcnt 0
}
// A GROUP of W lanes works on one read in the cooperative form: the whole wave (W = 64), or a quarter of it -- one 16-lane row (W = 16):
// four reads per wave, each with its own scratch in LDS (round 6: the reads of 9 .. 64 sampled seeds, the bulk of the seed-rich reads,
// whose quadratic loops are 16 wide at most and who paid a whole wave, or a lane with its scratch in global memory, each).  Every
// cross-lane operation of the core goes through this interface and stays inside the group, so the groups of a wave may diverge freely:
//   lane()      index inside the group          ballot(p)   the group's lanes with p, bit 0 = the group's first lane
//   bcast(v, u) v of the group's lane u (u uniform over the GROUP: a v_readlane for the wave, a ds_bpermute for a row)
template <int W> struct ch_grp;
template <> struct ch_grp<64> {
	static __device__ __forceinline__ int lane() { return ch_lane(); }
	static __device__ __forceinline__ unsigned long long ballot(const bool p) { return __ballot(p); }
	static __device__ __forceinline__ bool any(const bool p) { return __ballot(p) != 0ull; }
	static __device__ __forceinline__ int bcast(const int v, const int u) { return __builtin_amdgcn_readlane(v, u); }
	static __device__ __forceinline__ long long bcast64(const long long v, const int u)
	{
		return (long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFll), u) | ((long long)__builtin_amdgcn_readlane((int)(v >> 32), u) << 32);
	}
	static __device__ __forceinline__ long long shfl_up64(const long long v, const int d) { return __shfl_up(v, d); }
};
template <> struct ch_grp<16> {
	static __device__ __forceinline__ int lane() { return ch_lane() & 15; }
	static __device__ __forceinline__ unsigned long long ballot(const bool p) { return (__ballot(p) >> (ch_lane() & 48)) & 0xFFFFull; }
	static __device__ __forceinline__ bool any(const bool p) { return ballot(p) != 0ull; }
	static __device__ __forceinline__ int bcast(const int v, const int u) { return __shfl(v, u, 16); }
	static __device__ __forceinline__ long long bcast64(const long long v, const int u) { return __shfl(v, u, 16); }
	static __device__ __forceinline__ long long shfl_up64(const long long v, const int d) { return __shfl_up(v, d, 16); }
};
#endif

template <bool COOP = false, int W = 64> CH_HD inline int pos2rid(const ch_ctx_t &x, int64_t pos_f)
{
	if (pos_f >= x.l_pac) return -1;
	if (x.n_contigs <= 1) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP && x.n_contigs <= 64) {             // cooperative form: one contig start per lane, the answer is a ballot (the table sits in LDS)
		const int lane = ch_grp<W>::lane();
		int cnt = 0;
		for (int c0 = 0; c0 < x.n_contigs; c0 += W) {
			const bool le = c0 + lane < x.n_contigs && x.ctg_off[c0 + lane] <= pos_f;
			cnt += __builtin_popcountll(ch_grp<W>::ballot(le));
		}
		return cnt - 1;
	}
#endif
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
CH_HD inline int64_t depos(const ch_ctx_t &x, int64_t pos, int *is_rev) { return (*is_rev = (pos >= x.l_pac)) ? (x.l_pac << 1) - 1 - pos : pos; }
template <bool COOP = false, int W = 64> CH_HD inline int intv2rid(const ch_ctx_t &x, int64_t rb, int64_t re)
{
	int is_rev;
	if (rb < x.l_pac && re > x.l_pac) return -2;
	const int rid_b = pos2rid<COOP, W>(x, depos(x, rb, &is_rev));
	if (rb >= re || rid_b < 0 || x.n_contigs <= 1) return rid_b;
	// (the sequence of the interval's last base is only compared with rid_b, bns_intv2rid src/bntseq.c:362-373: no second search -- it is rid_b exactly when
	// that base lies in [ctg_off[rid_b], ctg_off[rid_b + 1]), the slice pos2rid's search assigns to rid_b)
	const int64_t pe = depos(x, re - 1, &is_rev);
	const int64_t hi = rid_b + 1 < x.n_contigs ? x.ctg_off[rid_b + 1] : x.l_pac;
	return pe >= x.ctg_off[rid_b] && pe < hi ? rid_b : -1;
}
// (an integer form, (n + e) / e, gives the same values but measured slower on the device than the double division)
// (gap extension penalties of 1 -- the default -- need no division: (double)n / 1 + 1. is n + 1 exactly, for either sign of n; the branch is uniform
// over a launch, and a double division is some forty instructions of which the chaining core made four to eight per seed)
CH_HD inline int ch_div_plus1(int n, int e) { return e == 1 ? n + 1 : (int)((double)n / e + 1.); }
CH_HD inline int cal_max_gap(const bmh_chain_opt_t &o, int qlen)
{
	const int l_del = ch_div_plus1(qlen * o.a - o.o_del, o.e_del);
	const int l_ins = ch_div_plus1(qlen * o.a - o.o_ins, o.e_ins);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < o.w << 1 ? l : o.w << 1;
}

// ---- order-sensitive sort of the chains by weight: klib introsort restated on 64-bit keys (weight << 32 | chain),
// compared on the weight only, descending -- ties must fall as they do in the reference.
CH_HD inline bool wlt(uint64_t a, uint64_t b) { return (uint32_t)(a >> 32) > (uint32_t)(b >> 32); }
CH_HD inline void wswap(uint64_t *a, int i, int j) { const uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
CH_HD inline void w_insertion(uint64_t *a, int s, int t)          // [s, t)
{
	for (int i = s + 1; i < t; ++i)
		for (int j = i; j > s && wlt(a[j], a[j - 1]); --j) wswap(a, j, j - 1);
}
CH_HD inline void w_comb(uint64_t *a, int n)
{
	const double shrink = 1.2473309501039786540366528676643;
	bool swapped; int gap = n;
	do {
		if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = false;
		for (int i = 0; i < n - gap; ++i) { const int j = i + gap; if (wlt(a[j], a[i])) { wswap(a, i, j); swapped = true; } }
	} while (swapped || gap > 2);
	if (gap != 1) w_insertion(a, 0, n);
}
// FINAL = false leaves out the closing insertion sort over the whole array (the caller does it: w_place_coop)
template <bool FINAL = true> CH_HD inline bool w_introsort(uint64_t *a, int n)
{
	if (n < 1) return true;
	if (n == 2) { if (wlt(a[1], a[0])) wswap(a, 0, 1); return true; }
	int d;
	for (d = 2; (1l << d) < n; ++d) ;
	int st_l[40], st_r[40], st_d[40], sp = 0;
	int s = 0, t = n - 1;
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { w_comb(a + s, t - s + 1); t = s; continue; }
			int i = s, j = t, k = i + ((j - i) >> 1) + 1;
			if (wlt(a[k], a[i])) { if (wlt(a[k], a[j])) k = j; }
			else k = wlt(a[j], a[i]) ? i : j;
			const uint64_t rp = a[k];
			if (k != t) wswap(a, k, t);
			for (;;) {
				do ++i; while (wlt(a[i], rp));
				do --j; while (i <= j && wlt(rp, a[j]));
				if (j <= i) break;
				wswap(a, i, j);
			}
			wswap(a, i, t);
			if (i - s > t - i) {
				if (i - s > 16) { if (sp >= 40) return false; st_l[sp] = s; st_r[sp] = i - 1; st_d[sp] = d; ++sp; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { if (sp >= 40) return false; st_l[sp] = i + 1; st_r[sp] = t; st_d[sp] = d; ++sp; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (sp == 0) { if (FINAL) w_insertion(a, 0, n); return true; }
			--sp; s = st_l[sp]; t = st_r[sp]; d = st_d[sp];
		}
	}
}

#if defined(__HIP_DEVICE_COMPILE__)
// The closing insertion sort of the introsort, by the whole wave.  Insertion sort is stable, so what it produces is the stable
// order by weight of what the quicksort phase left behind (which may be far from sorted: ks_introsort never looks at the first
// entry of a partition, the closing pass is what puts it right).  Entry x therefore ends at the number of entries that sort ahead
// of it: heavier ones, and equally heavy ones on its left -- one comparison on (weight << 16 | 0xFFFF - index).  NU entries per
// lane against all others, 64 of those at a time through v_readlane; the chain ids go straight to order[] (nothing reads the
// sorted keys).  Sequentially this pass was about half of the sort, and the sort a third of the chaining of a 500-seed read.
// (weights are at most the read length, indices below 2^16: mem_chain samples at most max_occ occurrences per SMEM)
template <int NU, int W = 64, class IDX> __device__ __forceinline__ void w_place_part(const uint64_t *a, const uint32_t *hi, IDX *order, int n, int xb, int lane)
{
	uint32_t cx[NU]; int pos[NU];
#pragma unroll
	for (int u = 0; u < NU; ++u) {
		const int x = xb + W * u + lane, xc = x < n ? x : n - 1;
		cx[u] = (hi[2 * xc] << 16) | (0xFFFFu - (uint32_t)xc); pos[u] = 0;
	}
	for (int yb = 0; yb < n; yb += W) {
		const int y = yb + lane;
		const uint32_t cy = y < n ? (hi[2 * y] << 16) | (0xFFFFu - (uint32_t)y) : 0u;       // 0 sorts behind every entry
		const int ke = n - yb < W ? n - yb : W;
		for (int k = 0; k < ke; ++k) {
			const uint32_t c = (uint32_t)ch_grp<W>::bcast((int)cy, k);
#pragma unroll
			for (int u = 0; u < NU; ++u) pos[u] += c > cx[u] ? 1 : 0;
		}
	}
#pragma unroll
	for (int u = 0; u < NU; ++u) { const int x = xb + W * u + lane; if (x < n) order[pos[u]] = (IDX)a[x]; }
}
// The quicksort phase of the same introsort with its partition loop spread over the wave.  ks_introsort's Hoare partition looks at
// every entry at most once from either side before the pointers meet, so which entries it exchanges follows from the array as it is
// when the partition starts: the k-th entry from the left that does not sort ahead of the pivot (weight <= pivot's, positions
// s+1..t) goes where the k-th entry from the right that does not sort behind it (weight >= pivot's, positions t-1..s) is, as long
// as the former lies left of the latter.  Pass A lists the right-hand candidates (tmp[s + k]), pass B ranks the left-hand ones,
// exchanges the pairs and finds where the left pointer stops: at the first candidate without a partner, or on the entry the last
// exchange brought to the right -- whichever comes first.  64 entries per step instead of one; pivot choice, recursion stack, depth
// limit and the order of the partitions are the sequential code's.  tmp: n words (order[], not in use before the sort ends).
template <bool LDSX, int W = 64, class IDX> __device__ inline int w_partition_coop(uint64_t *a, IDX *tmp, const int s, const int t, const uint32_t wp)
{
	const int lane = ch_grp<W>::lane();
	const uint32_t *hi = (const uint32_t *)a + 1;
	const unsigned long long below = (1ull << lane) - 1ull;
	int nle = 0;
	for (int top = t - 1; top >= s; top -= W) {                        // pass A: lane l looks at position top - l
		const int x = top - lane;
		const bool le = x >= s && hi[2 * (x >= s ? x : s)] >= wp;
		const unsigned long long m = ch_grp<W>::ballot(le);
		if (le) tmp[s + nle + (int)__builtin_popcountll(m & below)] = (IDX)x;
		nle += (int)__builtin_popcountll(m);
	}
	ch_wave_fence<LDSX>();
	int k0 = 0, ipos = -1;
	for (int lo = s + 1; ipos < 0; lo += W) {                          // pass B (position t holds the pivot: the loop ends there at the latest)
		const int x = lo + lane;
		const bool ge = x <= t && hi[2 * (x <= t ? x : t)] <= wp;
		const unsigned long long m = ch_grp<W>::ballot(ge);
		const int k = k0 + (int)__builtin_popcountll(m & below);
		int r = -1;
		if (ge && k < nle) r = (int)tmp[s + k];
		const bool sw = ge && r > x;
		const unsigned long long ms = ch_grp<W>::ballot(sw);
		if (sw) { const uint64_t va = a[x], vb = a[r]; a[x] = vb; a[r] = va; }
		const unsigned long long stop = m & ~ms;
		int cand = stop ? lo + (int)__builtin_ctzll(stop) : 0x7FFFFFFF;
		if (ms) {
			const int rl = ch_grp<W>::bcast(r, 63 - (int)__builtin_clzll(ms));               // where the last exchange put a left-hand entry
			if (rl <= lo + W - 1 && rl < cand) cand = rl;
		}
		if (cand != 0x7FFFFFFF) ipos = cand;
		k0 += (int)__builtin_popcountll(m);
		ch_wave_fence<LDSX>();
	}
	return ipos;
}
template <bool LDSX, int W = 64, class IDX> __device__ inline bool w_introsort_coop(uint64_t *a, IDX *tmp, int n)
{
	if (n < 1) return true;
	if (n == 2) { if (wlt(a[1], a[0])) wswap(a, 0, 1); ch_wave_fence<LDSX>(); return true; }
	int d;
	for (d = 2; (1l << d) < n; ++d) ;
	int st_l[40], st_r[40], st_d[40], sp = 0;
	int s = 0, t = n - 1;
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { w_comb(a + s, t - s + 1); ch_wave_fence<LDSX>(); t = s; continue; }
			int i = s, j = t, k = i + ((j - i) >> 1) + 1;
			{
				const uint64_t ak = a[k], ai = a[i], aj = a[j];
				if (wlt(ak, ai)) { if (wlt(ak, aj)) k = j; }
				else k = wlt(aj, ai) ? i : j;
			}
			const uint64_t rp = a[k];
			if (k != t) { wswap(a, k, t); ch_wave_fence<LDSX>(); }
			i = w_partition_coop<LDSX, W>(a, tmp, s, t, (uint32_t)(rp >> 32));
			if (i != t) { wswap(a, i, t); ch_wave_fence<LDSX>(); }
			if (i - s > t - i) {
				if (i - s > 16) { if (sp >= 40) return false; st_l[sp] = s; st_r[sp] = i - 1; st_d[sp] = d; ++sp; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { if (sp >= 40) return false; st_l[sp] = i + 1; st_r[sp] = t; st_d[sp] = d; ++sp; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (sp == 0) return true;
			--sp; s = st_l[sp]; t = st_r[sp]; d = st_d[sp];
		}
	}
}
template <bool LDSX, int W = 64, class IDX> __device__ inline void w_place_coop(const uint64_t *a, IDX *order, int n)
{
	const int lane = ch_grp<W>::lane();
	const uint32_t *hi = (const uint32_t *)a + 1;                     // the weights: high words of the keys
	for (int xb = 0; xb < n; xb += 8 * W) {
		const int rem = n - xb;
		if (rem <= W) w_place_part<1, W>(a, hi, order, n, xb, lane);
		else if (rem <= 2 * W) w_place_part<2, W>(a, hi, order, n, xb, lane);
		else if (rem <= 4 * W) w_place_part<4, W>(a, hi, order, n, xb, lane);
		else w_place_part<8, W>(a, hi, order, n, xb, lane);
	}
	ch_wave_fence<LDSX>();
}
#endif

// ---- helpers that are split across the wave when COOP
// insert (cv, pv) at position at of order/opos[0..nc)
template <bool COOP, bool LDSX = false, int W = 64, class IDX> CH_HD inline void sorted_insert(IDX *order, int64_t *opos, int nc, int at, uint32_t cv, int64_t pv)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) {
		// each round moves the top 4 W entries of [at, hi) up by one: four independent loads per lane, then the four stores
		// (a round of W costs the same two LDS latencies)
		// (U: entries per lane and round -- four for the wave, whose lists run to thousands; ONE for a 16-lane row, whose lists hold 64 entries at most and mostly
		// a dozen: the quarters beyond the end were three quarters of a row's instructions in these loops)
		const int lane = ch_grp<W>::lane();
		constexpr int U = W == 64 ? 4 : 1;
		for (int hi = nc; hi > at; hi -= U * W) {
			uint32_t v[U]; int64_t p[U];
#pragma unroll
			for (int u = 0; u < U; ++u) { const int j = hi - 1 - lane - W * u, jc = j >= at ? j : at; v[u] = order[jc]; p[u] = opos[jc]; }   // no branch: the loads overlap
			ch_wave_fence<LDSX>();
#pragma unroll
			for (int u = 0; u < U; ++u) { const int j = hi - 1 - lane - W * u; if (j >= at) { order[j + 1] = (IDX)v[u]; opos[j + 1] = p[u]; } }
			ch_wave_fence<LDSX>();
		}
		order[at] = (IDX)cv; opos[at] = pv;
		ch_wave_fence<LDSX>();
		return;
	}
#endif
	for (int j = nc; j > at; --j) { order[j] = order[j - 1]; opos[j] = opos[j - 1]; }
	order[at] = (IDX)cv; opos[at] = pv;
}

// upper bound of rb in the ascending opos[0..nc): number of entries <= rb
template <bool COOP, int W = 64> CH_HD inline int upper_bound_pos(const int64_t *opos, int nc, int64_t rb)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP && nc > 8 && nc <= W * W) {         // two W-way steps instead of log2(nc) dependent ones
		const int lane = ch_grp<W>::lane();
		const int stride = (nc + W - 1) / W;
		const int last = ((lane + 1) * stride < nc ? (lane + 1) * stride : nc) - 1;
		const bool le = lane * stride < nc && opos[last] <= rb;
		const int b = __builtin_popcountll(ch_grp<W>::ballot(le));
		if (stride == 1) return b;
		const int base = b * stride;
		if (base >= nc) return nc;
		const int idx = base + lane;
		const bool le2 = lane < stride && idx < nc && opos[idx] <= rb;
		return base + __builtin_popcountll(ch_grp<W>::ballot(le2));
	}
#endif
	int lo = 0, hi = nc;
	while (lo < hi) { const int mid = (lo + hi) >> 1; if (opos[mid] <= rb) lo = mid + 1; else hi = mid; }
	return lo;
}

// first index in [lo, hi) for which f is true, hi if none; f must be free of side effects
template <bool COOP, int W = 64, class F> CH_HD inline int first_true(int lo, int hi, F f)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) {
		// 4 W candidates per round: the four predicates are independent, so their LDS loads overlap (f is free of side effects)
		const int lane = ch_grp<W>::lane();
		constexpr int U = W == 64 ? 4 : 1;
		for (int b = lo; b < hi; b += U * W) {
			unsigned long long m[U];
#pragma unroll
			for (int u = 0; u < U; ++u) { const int i = b + W * u + lane; m[u] = ch_grp<W>::ballot(i < hi && f(i)); }
#pragma unroll
			for (int u = 0; u < U; ++u) if (m[u]) return b + W * u + (int)__builtin_ctzll(m[u]);
		}
		return hi;
	}
#endif
	for (int i = lo; i < hi; ++i) if (f(i)) return i;
	return hi;
}

// sort n distinct 64-bit keys ascending (any algorithm gives the same result)
template <bool COOP, bool LDSX = false, int W = 64> CH_HD inline void sort_distinct(uint64_t *a, uint64_t *tmp, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP && n > (W == 64 ? 24 : 6)) {          // rank sort: element i goes to the number of keys below it
		const int lane = ch_grp<W>::lane();
		for (int b = 0; b < n; b += W) {
			const int i = b + lane;
			if (i < n) {
				const uint64_t v = a[i]; int rank = 0;
				for (int j = 0; j < n; ++j) rank += a[j] < v;
				tmp[rank] = v;
			}
		}
		ch_wave_fence<LDSX>();
		for (int b = 0; b < n; b += W) { const int i = b + lane; if (i < n) a[i] = tmp[i]; }
		ch_wave_fence<LDSX>();
		return;
	}
#endif
	(void)tmp;
	for (int i = 1; i < n; ++i) {
		const uint64_t v = a[i]; int j = i;
		for (; j > 0 && a[j - 1] > v; --j) a[j] = a[j - 1];
		a[j] = v;
	}
}


#if defined(__HIP_DEVICE_COMPILE__)
// minimum over the 64 lanes, on DPP (row shifts inside the 16-lane rows, then row_bcast 15 / 31: lane 63 ends with the total) --
// six register-to-register steps instead of six ds_bpermute round trips of a shuffle butterfly
__device__ __forceinline__ uint32_t ch_wave_min_u32(uint32_t v)
{
	int x = (int)v;                                               // values are below 2^31: signed min is the same
	int t;
	t = __builtin_amdgcn_update_dpp(0x7FFFFFFF, x, 0x111, 0xf, 0xf, false); x = t < x ? t : x;
	t = __builtin_amdgcn_update_dpp(0x7FFFFFFF, x, 0x112, 0xf, 0xf, false); x = t < x ? t : x;
	t = __builtin_amdgcn_update_dpp(0x7FFFFFFF, x, 0x114, 0xf, 0xf, false); x = t < x ? t : x;
	t = __builtin_amdgcn_update_dpp(0x7FFFFFFF, x, 0x118, 0xf, 0xf, false); x = t < x ? t : x;
	t = __builtin_amdgcn_update_dpp(0x7FFFFFFF, x, 0x142, 0xa, 0xf, false); x = t < x ? t : x;
	t = __builtin_amdgcn_update_dpp(0x7FFFFFFF, x, 0x143, 0xc, 0xf, false); x = t < x ? t : x;
	return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// The kept-chain loop of mem_chain_flt (src/bwamem.c:506-535) for one read on a wave, without scanning the kept list per chain.
// Three facts make that exact:
//  * whether chain i "overlaps significantly" with kept chain j depends only on their query spans (beg, end).  The chains of a
//    seed-rich read have very few distinct spans (one per SMEM): every distinct span is a CLASS, owned by one lane (at most 64
//    classes, else the caller falls back to the scan);
//  * chains are visited in descending weight, so the kept list is in descending weight too, and the kept chains heavy enough to
//    drop chain i (ai.w < aj.w * drop_ratio && aj.w - ai.w >= 2 min_seed_len) are a PREFIX [0, bp) of it that only grows with i.
//    Chain i is dropped iff a class that overlaps it has its first member inside that prefix; the scan of the reference stops at
//    the earliest such member (`reach`);
//  * a kept chain's `first` is set once, by the first later chain whose scan reaches it while overlapping; a scan marks every
//    unmarked overlapping member up to `reach`, so the unmarked members of a class are always a suffix of its member list: one
//    pointer per class, advanced as members get marked.
// kept_w[k] / mem_next[k]: weight of kept chain k and the next kept chain of its class; cls[i]: class of sorted chain i.
// Returns false (nothing modified but scratch) when the read has more than 64 distinct spans.
template <bool LDSX, class CHT, class IDX> __device__ inline bool ch_kept_by_classes(const bmh_chain_opt_t &o, CHT *CH, const IDX *order, IDX *klist, int32_t *kept_w,
                                          uint32_t *mem_next, IDX *cls, int na, int &nk_out)
{
	const int lane = ch_lane();
	const uint32_t NIL = 0xFFFFFFFFu, INF = 0x7FFFFFFFu;
	uint32_t classkey = 0xFFFFFFFFu;                          // lane t: (beg << 16 | end) of class t
	int ncls = 0;
	for (int b = 0; b < na; b += 64) {
		const int i = b + lane;
		uint32_t key = 0xFFFFFFFEu;
		if (i < na) { const CHT c = CH[order[i]]; key = (uint32_t)c.beg << 16 | (uint32_t)c.end; }
		int id = -1;
		for (int t = 0; t < ncls; ++t) { const uint32_t kt = (uint32_t)__builtin_amdgcn_readlane((int)classkey, t); if (key == kt) id = t; }
		unsigned long long un = __ballot(i < na && id < 0);
		while (un) {
			if (ncls >= 64) return false;
			const uint32_t kL = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(un));
			if (lane == ncls) classkey = kL;
			if (i < na && id < 0 && key == kL) id = ncls;
			++ncls;
			un = __ballot(i < na && id < 0);
		}
		if (i < na) cls[i] = (IDX)id;
	}
	ch_wave_fence<LDSX>();
	const int cbeg = (int)(classkey >> 16), cend = (int)(classkey & 0xFFFFu);
	int c_n = 0; uint32_t c_first = INF, c_tail = NIL, c_um = NIL;
	int nk = 0, bp = 0, wbp = 0;                               // wbp = kept_w[bp] while bp < nk (kept in a register: the test runs for every chain)
	for (int b = 0; b < na; b += 64) {
		int vbeg = 0, vend = 0, vw = 0, vcls = 0; uint32_t vci = 0;
		if (b + lane < na) { vci = order[b + lane]; const CHT c = CH[vci]; vbeg = c.beg; vend = c.end; vw = c.w; vcls = (int)cls[b + lane]; }
		const int m = na - b < 64 ? na - b : 64;
		for (int u = 0; u < m; ++u) {
			const int i = b + u;
			// (u is wave-uniform: v_readlane, not a shuffle through the LDS crossbar)
			const int ibeg = __builtin_amdgcn_readlane(vbeg, u), iend = __builtin_amdgcn_readlane(vend, u), iw = __builtin_amdgcn_readlane(vw, u),
			          sc = __builtin_amdgcn_readlane(vcls, u);
			const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)vci, u);
			bool broke = false, large = false;
			if (i > 0) {
				bool ovl = false;
				if (lane < ncls && c_n > 0) {
					const int b_max = cbeg > ibeg ? cbeg : ibeg, e_min = cend < iend ? cend : iend;
					if (e_min > b_max) {
						const int li = iend - ibeg, lj = cend - cbeg;
						const int min_l = li < lj ? li : lj;
						ovl = e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap;
					}
				}
				while (bp < nk) { if (iw < wbp * o.drop_ratio && wbp - iw >= o.min_seed_len << 1) { ++bp; if (bp < nk) wbp = kept_w[bp]; } else break; }
				const uint32_t cand = ch_wave_min_u32((ovl && c_first < (uint32_t)bp) ? c_first : INF);
				broke = cand != INF;
				const uint32_t reach = broke ? cand : INF - 1;
				large = __ballot(ovl) != 0ull;
				if (ovl) while (c_um != NIL && c_um <= reach) { CH[cls[c_um]].first = i; c_um = mem_next[c_um]; }      // cls[k] = chain of kept entry k by now
			}
			if (!broke) {
				const uint32_t k = (uint32_t)nk;
				if (lane == 0) { klist[k] = (IDX)i; kept_w[k] = iw; mem_next[k] = NIL; cls[k] = (IDX)ci; CH[ci].kept = large ? 2u : 3u; }
				if ((int)k == bp) wbp = iw;                               // the prefix had caught up with the list: its next entry is this one
				if (lane == sc) { if (c_n == 0) c_first = k; else mem_next[c_tail] = k; c_tail = k; if (c_um == NIL) c_um = k; ++c_n; }
				++nk;
			}
			if (LDSX) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // LDS operations of a wave execute in order: no wait needed
			else ch_wave_fence<LDSX>();
			// A RUN of chains of the same class and weight behind chain i (the occurrences of one repetitive SMEM: hundreds of chains
			// of one seed each) goes the way chain i went, all at once.  Neither the heavy prefix nor the first members of the
			// classes move inside the run (the entries it appends weigh what its chains weigh), so: chain i dropped -- the run is
			// dropped, and its scans find nothing left to mark below `reach`; chain i kept -- every chain of the run is kept, in
			// order, each one's scan marking just the member before it (`first` = the next sorted index), the last one left
			// unmarked.  One chain per lane instead of one chain per iteration.
			{
				const unsigned long long same = __ballot(b + lane < na && vcls == sc && vw == iw);
				const unsigned long long ahead = u < 63 ? same >> (u + 1) : 0ull;
				const int run = (int)__builtin_ctzll(~ahead);                      // chains u+1 .. u+run of this block
				if (run > 0 && i > 0) {
					if (!broke) {
						const int len = iend - ibeg;
						const bool self = len > 0 && len >= len * o.mask_level && len < o.max_chain_gap;      // the class overlaps itself
						const bool large2 = large || self;
						const int j = lane - u;                                        // 1 .. run: this lane's place in the run
						const uint32_t k0 = (uint32_t)nk;                             // kept index of the run's first chain (j = 1); chain i has k0 - 1
						if (j >= 1 && j <= run) {
							const uint32_t k = k0 + (uint32_t)(j - 1);
							klist[k] = (IDX)(i + j); kept_w[k] = iw; cls[k] = (IDX)vci; mem_next[k] = j < run ? k + 1 : NIL;
							CH[vci].kept = large2 ? 2u : 3u;
							if (self && j < run) CH[vci].first = i + j + 1;            // marked by the next chain of the run
						}
						if (self && lane == u) CH[vci].first = i + 1;                  // chain i, marked by the run's first chain
						if (lane == sc) {
							mem_next[c_tail] = k0; c_tail = k0 + (uint32_t)run - 1; c_n += run;
							c_um = self ? c_tail : c_um;                              // (without self-overlap nothing of the class gets marked)
						}
						nk += run;
					}
					u += run;
					if (LDSX) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
					else ch_wave_fence<LDSX>();
				}
			}
		}
	}
	nk_out = nk;
	return true;
}
#endif

// ---------------------------------------------------------------- mem_flt_chained_seeds (src/bwamem.c:970-991) + mem_seed_sw (:774-807)
// Does the filter run for a read of this length?  (:972-977; MEM_HSP_COEF 1.1f and MEM_SEEDSW_COEF 0.05f are float constants there:
// the products are formed in float, the comparison in double.)  The logarithm only matters without -W, i.e. beyond ~730 bp.
CH_HD inline bool seed_filter_applies(const bmh_chain_opt_t &o, int l_query, int *min_HSP_score)
{
	const double min_l = o.min_chain_weight ? (double)(1.1f * (float)o.min_chain_weight) : (double)5.5f * log((double)l_query);
	if (min_HSP_score) *min_HSP_score = (int)(o.a * min_l + .499);
	return !(min_l > (double)(0.05f * (float)l_query));
}
CH_HD inline int ch_nt4(uint8_t c) { if (c <= 4) return c; c &= 0xDF; return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
CH_HD inline int ch_text_base(const uint8_t *pac, int64_t l_pac, int64_t i)    // symbol i of fwd . revcomp(fwd); bns_get_seq src/bntseq.c:558-580
{
	const bool rev = i >= l_pac;
	const int64_t p = rev ? (l_pac << 1) - 1 - i : i;
	const int c = (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}
// Score of ksw_align2(qlen, query, tlen, target, 5, mat, ..., KSW_XSTART, 0) (src/ksw.c:389-740 through ksw_i16 :539-665): the striped
// SSE2 kernel walked by one thread, lane by lane -- eight 16-bit lanes, lane l owning the query positions l slen .. l slen + slen - 1,
// E taken before the lazy-F correction, the lazy-F loop with its exit test, as csrc/local_sw.cpp (the host form, pinned to the
// reference binary's records) and csrc/pair_kernels.hip (sixteen GPU lanes per alignment) do.  Only the score is wanted here.
#define CH_SW_MAXQ 200          // MEM_SHORT_LEN: mem_seed_sw gives up on windows of 200 bases or more
// (the rows live in the thread's private memory: one 16-byte vector per segment position -- the eight lanes of the SSE register it
// stands for -- so that a position costs one load or store instead of eight; on the device that memory is what the walk waits for)
struct ch_sw_v8 { int16_t v[8]; } __attribute__((aligned(16)));
struct ch_sw_q8 { int8_t v[8]; } __attribute__((aligned(8)));
template <class QB, class TB> CH_HD inline int seed_sw_score(const bmh_chain_opt_t &o, int qlen, QB qbase, int tlen, TB tbase)
{
	constexpr int L = 8, SMAX = (CH_SW_MAXQ + L - 1) / L;
	const int slen = (qlen + L - 1) / L;
	ch_sw_v8 H0[SMAX], H1[SMAX], Ev[SMAX]; ch_sw_q8 Q[SMAX];
	for (int j = 0; j < slen; ++j) {
		ch_sw_v8 z; ch_sw_q8 q;
#pragma unroll
		for (int l = 0; l < L; ++l) { const int k = j + l * slen; z.v[l] = 0; q.v[l] = (int8_t)(k < qlen ? qbase(k) : 5); }
		H0[j] = z; H1[j] = z; Ev[j] = z; Q[j] = q;
	}
	const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
	auto sat0 = [](int v) { return v < 0 ? 0 : v; };
	ch_sw_v8 *h0 = H0, *h1 = H1;
	int gmax = 0;
	for (int i = 0; i < tlen; ++i) {
		const int t = tbase(i);
		int hv[L], f[L], mxv[L];
		{
			const ch_sw_v8 last = h0[slen - 1];
#pragma unroll
			for (int l = 0; l < L; ++l) { hv[l] = l ? last.v[l - 1] : 0; f[l] = 0; mxv[l] = 0; }
		}
		for (int j = 0; j < slen; ++j) {
			const ch_sw_q8 q8 = Q[j]; const ch_sw_v8 e8 = Ev[j], p8 = h0[j];
			ch_sw_v8 hn, en;
#pragma unroll
			for (int l = 0; l < L; ++l) {
				const int q = q8.v[l];
				const int sc = q == 5 ? 0 : (t > 3 || q > 3) ? -1 : (t == q ? o.a : -o.b);
				int h = hv[l] + sc; h = h > 32767 ? 32767 : h < -32768 ? -32768 : h;
				const int e = e8.v[l];
				h = h > e ? h : e; h = h > f[l] ? h : f[l];
				mxv[l] = mxv[l] > h ? mxv[l] : h;
				hn.v[l] = (int16_t)h;
				const int hu = h & 0xffff;                               // _mm_subs_epu16 reads the 16 bits as unsigned
				const int e1 = sat0(e - o.e_del), e2 = sat0(hu - oe_del);
				en.v[l] = (int16_t)(e1 > e2 ? e1 : e2);
				const int f1 = sat0(f[l] - o.e_ins), f2 = sat0(hu - oe_ins);
				f[l] = f1 > f2 ? f1 : f2;
				hv[l] = p8.v[l];
			}
			h1[j] = hn; Ev[j] = en;
		}
		for (int k = 0; k < 16; ++k) {                                   // lazy F (src/ksw.c:627-638)
#pragma unroll
			for (int l = L - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			bool done = false;
			for (int j = 0; j < slen; ++j) {
				ch_sw_v8 h8 = h1[j];
				bool any = false;
#pragma unroll
				for (int l = 0; l < L; ++l) {
					int h = h8.v[l]; h = h > f[l] ? h : f[l];
					h8.v[l] = (int16_t)h;
					h = sat0((h & 0xffff) - oe_ins);
					f[l] = sat0(f[l] - o.e_ins);
					if (f[l] > h) any = true;
				}
				h1[j] = h8;
				if (!any) { done = true; break; }
			}
			if (done) break;
		}
		int imax = 0;
#pragma unroll
		for (int l = 0; l < L; ++l) imax = imax > mxv[l] ? imax : mxv[l];
		if (imax > gmax) gmax = imax;
		ch_sw_v8 *tmp = h0; h0 = h1; h1 = tmp;
	}
	return gmax;
}
// mem_seed_sw :774-807: local alignment score of the seed's neighbourhood (50 bases either side), -1 when the seed or its window is
// long enough to be trusted as it is
template <class ST> CH_HD inline int seed_sw(const ch_ctx_t &x, uint32_t r, int l_query, const ST &s)
{
	const int SHORT_EXT = 50, SHORT_LEN = CH_SW_MAXQ;                    // MEM_SHORT_EXT, MEM_SHORT_LEN
	const int64_t l_pac = x.l_pac;
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
		const int rid = pos2rid<false>(x, depos(x, mid, &is_rev));
		int64_t far_beg = x.n_contigs > 1 ? x.ctg_off[rid] : 0, far_end = far_beg + (x.n_contigs > 1 ? x.ctg_len[rid] : l_pac);
		if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
		rb = rb > far_beg ? rb : far_beg;
		re = re < far_end ? re : far_end;
	}
	const uint8_t *query = x.reads + x.read_offs[r] + qb;
	const uint8_t *pac = x.pac;
	return seed_sw_score(x.o, qe - qb, [&](int k) { return ch_nt4(query[k]); }, (int)(re - rb), [&](int i) { return ch_text_base(pac, l_pac, rb + i); });
}

CH_HD inline ch_scr_t global_scratch(const ch_ctx_t &x, uint32_t r)
{
	const uint32_t b = x.prefix[r];
	ch_scr_t s;
	s.S = x.g.S + b; s.CH = x.g.CH + b; s.order = x.g.order + b; s.opos = x.g.opos + b; s.klist = x.g.klist + b; s.srt = x.g.srt + b;
	s.cidx = x.g.cidx + b; s.E = x.g.E + b;
	return s;
}

// The read: returns through x.regs (slot order = creation order), x.regs_per_read[r], x.jobs_per_read[r].
// (W: lanes of the group that works on the read in the cooperative form -- the wave, or a 16-lane row with its own scratch, ch_grp)
// STG (cooperative device forms whose read has at most as many located seeds as its scratch has entries): the read's seed arrays -- group sizes, query
// intervals, positions -- are copied into the scratch (the space of E, not in use before mem_chain_flt) by the whole group in ONE round trip to global
// memory; without it every SMEM group costs a chain of dependent global loads (its size, its interval, its positions), which was most of the time a
// read of a dozen seeds took.
// SITE: a tag that gives a call site an instantiation of its own -- the wave kernel calls the same form with scratch pointers of different address spaces
// (all global; the arrays of the sequential phases in LDS), and ONE shared copy of the function makes every access through them a flat one.
template <bool COOP, bool LDSX = false, bool FLT = false, int W = 64, bool STG = false, int SITE = 0, class SCR = ch_scr_t> CH_HD CH_INLINE void chain_read(const ch_ctx_t &x, uint32_t r, const SCR &sc)
{
	typedef typename SCR::ty TY;                                  // the records' widths: ch_wide_ty or ch_compact_ty
	typedef typename TY::seed_t seed_t; typedef typename TY::chain_t chain_t; typedef typename TY::est_t est_t; typedef typename TY::idx_t idx_t;
	constexpr uint32_t NIL = TY::NIL;                             // end of a chain's seed list
	const bmh_chain_opt_t &o = x.o;
	const uint32_t base = x.prefix[r];
	const int n = (int)x.n_ref[r];
	const int l_query = (int)x.read_lens[r];
	const int64_t l_pac = x.l_pac;
	seed_t *S = sc.S; chain_t *CH = sc.CH; idx_t *order = sc.order; int64_t *opos = sc.opos;
	idx_t *klist = sc.klist; uint64_t *srt = sc.srt; idx_t *cidx = sc.cidx; est_t *E = sc.E; ch_reg_t *R = x.regs + base;
	x.regs_per_read[r] = 0; x.jobs_per_read[r] = 0; x.frac_rep[r] = 0.f;
	if (n == 0 || l_query < o.min_seed_len) return;
	// the reference's seed filter (mem_flt_chained_seeds, src/bwamem.c:970-991) applies to a read with (W ? 1.1f W : 5.5 ln l) <= 0.05f l,
	// i.e. a small -W or more than ~730 bp: the FLT forms run it (below, between mem_chain_flt and mem_chain2aln), the others refuse
	// the read, and so does every form a read longer than the extension kernels' classes reach
	int min_HSP_score = 0;
	const bool flt_on = seed_filter_applies(o, l_query, &min_HSP_score);
	if (l_query > CH_MAX_READ_LEN || (flt_on && !FLT)) { *x.err = 1; return; }
	const uint64_t *g_rbeg; const int32_t *g_qbeg; const uint32_t *g_score;
#if defined(__HIP_DEVICE_COMPILE__)
	if constexpr (STG) {
		static_assert(COOP, "staged seeds: cooperative forms only");
		const uint64_t *s_rbeg = x.rbeg + base; const int32_t *s_qbeg = x.qbeg + 2 * (size_t)base; const uint32_t *s_score = x.score + base;
		uint32_t *ls = (uint32_t *)E; int32_t *lq = (int32_t *)(ls + n); uint64_t *lr = (uint64_t *)(ls + ((3 * n + 1) & ~1));      // 20 n + 4 bytes of E's 32 n
		const int lane = ch_grp<W>::lane();
		for (int k = lane; k < n; k += W) { ls[k] = s_score[k]; lr[k] = s_rbeg[k]; }
		for (int k = lane; k < 2 * n; k += W) lq[k] = s_qbeg[k];
		ch_wave_fence<false>();                                    // (the loads must have arrived: vmcnt and lgkmcnt 0)
		g_rbeg = lr; g_qbeg = lq; g_score = ls;
	} else
#endif
	{ g_rbeg = x.rbeg + base; g_qbeg = x.qbeg + 2 * (size_t)base; g_score = x.score + base; }

	// ---------------------------------------------------------------- mem_chain
	CH_STAMP(0);
	{
		int b = 0, e = 0, l_rep = 0;
		for (int i = 0; i < n;) {
			const uint32_t cnt = g_score[i];
			if (cnt == 0) break;
			if (cnt > (uint32_t)o.max_occ) {
				const int sb = g_qbeg[2 * i], se = g_qbeg[2 * i + 1];
				if (sb > e) { l_rep += e - b; b = sb; e = se; }
				else e = e > se ? e : se;
			}
			i += (int)cnt;
		}
		l_rep += e - b;
		x.frac_rep[r] = (float)l_rep / l_query;
	}
	int nc = 0, ns = 0;
	for (int i = 0; i < n;) {
		const uint32_t cnt = g_score[i];
		if (cnt == 0) break;                               // malformed group head; cannot happen with bmh_seed_batch output
		const int sb = g_qbeg[2 * i], slen = g_qbeg[2 * i + 1] - sb;
		const int step = cnt > (uint32_t)o.max_occ ? (int)(cnt / o.max_occ) : 1;
		// occurrences k = 0, step, 2 step, ... while k < cnt, at most max_occ of them
		const int n_it = (int)(((int64_t)cnt + step - 1) / step) < o.max_occ ? (int)(((int64_t)cnt + step - 1) / step) : o.max_occ;
		// one occurrence of the SMEM (sb, slen) at reference position rb, the reference's way: closest chain at or below it, merge or new
		auto seq_one = [&](const int64_t rb) {
			const int rid = intv2rid<COOP, W>(x, rb, rb + slen);
			if (rid < 0) return;
			// closest chain at or below the seed: upper bound over opos[0..nc), then one back
			const int lo = upper_bound_pos<COOP, W>(opos, nc, rb);
			bool to_add = true;
			if (lo > 0) {
				chain_t &c = CH[order[lo - 1]];
				const seed_t first = S[c.head], last = S[c.tail];
				const int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len;
				if (rid == c.rid) {                            // test_and_merge
					if (sb >= first.qbeg && sb + slen <= qend && rb >= first.rbeg && rb + slen <= rend) to_add = false;   // contained
					else if (!((last.rbeg < l_pac || first.rbeg < l_pac) && rb >= l_pac)) {
						const int64_t xq = sb - last.qbeg, y = rb - last.rbeg;
						if (y >= 0 && xq - y <= o.w && y - xq <= o.w && xq - last.len < o.max_chain_gap && y - last.len < o.max_chain_gap) {
							seed_t s; s.rbeg = rb; s.qbeg = sb; s.len = slen; s.next = NIL; s.pad = 0;
							S[ns] = s; S[c.tail].next = (uint32_t)ns; c.tail = (uint32_t)ns; ++c.n; ++ns;
							to_add = false;
						}
					}
				}
			}
			if (to_add) {
				seed_t s; s.rbeg = rb; s.qbeg = sb; s.len = slen; s.next = NIL; s.pad = 0;
				S[ns] = s;
				chain_t c; c.head = c.tail = (uint32_t)ns; c.n = 1; c.rid = rid; c.w = 0; c.first = -1; c.beg = c.end = 0; c.kept = 0; c.pad = 0;
				CH[nc] = c;
				sorted_insert<COOP, LDSX, W>(order, opos, nc, lo, (uint32_t)nc, rb);
				++ns; ++nc;
			}
		};
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) {
			// 64 occurrences of the SMEM at a time, one per lane.  Each lane SPECULATES its occurrence against the chains as they
			// stand before the batch (own binary search, own merge test).  The speculation is the sequential outcome unless an earlier
			// occurrence of the batch changes what a later one sees: (a) it starts a new chain that becomes the later one's closest
			// chain -- which matters if the later one was going to merge / be contained, or lies within w of it (all occurrences of
			// a batch share the query interval, so only then can it merge into that one-seed chain); (b) it merges into the chain the
			// later one is tested against.  Lanes look for such a pair among themselves (64 register reads each); a batch without
			// one -- the rule on a seed-rich read, whose occurrences lie at loci of their own -- is committed by all lanes at once and
			// its new chains are merged into the sorted chain index in one pass; any other batch takes the sequential path below.
			const int lane = ch_grp<W>::lane();
			for (int c0 = 0; c0 < n_it; c0 += W) {
				const int cc = c0 + lane;
				const bool act = cc < n_it;
				const long long rbl = act ? (long long)g_rbeg[i + (int64_t)cc * step] : 0;
				bool done = false;
				CH_ACC_BEGIN();
				if (n_it - c0 >= 8) {
					int rid = -1;
					if (act) rid = intv2rid<false>(x, rbl, rbl + slen);
					const bool valid = act && rid >= 0;
					int lo = 0, kind = 0;                               // kind: 0 nothing, 1 new chain, 2 merge, 3 contained
					uint32_t pc = 0xFFFFFFFFu, ptail = 0; long long ppos = -1;
					if (valid) {
						lo = upper_bound_pos<false>(opos, nc, rbl);
						kind = 1;
						if (lo > 0) {
							pc = order[lo - 1]; ppos = opos[lo - 1];
							const chain_t c = CH[pc];
							const seed_t first = S[c.head], last = S[c.tail];
							ptail = c.tail;
							const int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len;
							if (rid == c.rid) {
								if (sb >= first.qbeg && sb + slen <= qend && rbl >= first.rbeg && rbl + slen <= rend) kind = 3;
								else if (!((last.rbeg < l_pac || first.rbeg < l_pac) && rbl >= l_pac)) {
									const int64_t xq = sb - last.qbeg, y = rbl - last.rbeg;
									if (y >= 0 && xq - y <= o.w && y - xq <= o.w && xq - last.len < o.max_chain_gap && y - last.len < o.max_chain_gap) kind = 2;
								}
							}
						}
					}
					bool inter = false; int below = 0;
					const int rlo = (int)(unsigned)(rbl & 0xFFFFFFFFll), rhi = (int)(rbl >> 32);
					for (int v = 0; v < W; ++v) {
						const int kv = ch_grp<W>::bcast(kind, v);
						if (kv != 1 && kv != 2) continue;                 // (uniform over the group)
						const long long rv = (long long)(unsigned)ch_grp<W>::bcast(rlo, v) | ((long long)ch_grp<W>::bcast(rhi, v) << 32);
						const uint32_t pv = (uint32_t)ch_grp<W>::bcast((int)pc, v);
						if (valid) {
							if (v < lane) {
								if (kv == 1 && rv <= rbl && (lo == 0 || rv >= ppos) && (kind != 1 || rbl - rv <= o.w)) inter = true;
								if (kv == 2 && lo > 0 && pv == pc) inter = true;
							}
							if (kv == 1 && kind == 1 && rv < rbl) ++below;
						}
					}
					if (!ch_grp<W>::any(inter)) {
						const bool isnew = kind == 1, ismrg = kind == 2;
						const unsigned long long lt = (1ull << lane) - 1, mnew = ch_grp<W>::ballot(isnew), madd = ch_grp<W>::ballot(isnew || ismrg);
						const uint32_t my_ns = (uint32_t)ns + (uint32_t)__builtin_popcountll(madd & lt), my_nc = (uint32_t)nc + (uint32_t)__builtin_popcountll(mnew & lt);
						if (isnew || ismrg) { seed_t sd; sd.rbeg = rbl; sd.qbeg = sb; sd.len = slen; sd.next = NIL; sd.pad = 0; S[my_ns] = sd; }
						if (ismrg) { S[ptail].next = my_ns; CH[pc].tail = my_ns; CH[pc].n += 1; }
						if (isnew) {
							chain_t c; c.head = c.tail = my_ns; c.n = 1; c.rid = rid; c.w = 0; c.first = -1; c.beg = c.end = 0; c.kept = 0; c.pad = 0;
							CH[my_nc] = c;
						}
						const int K = (int)__builtin_popcountll(mnew);
						if (K) {
							// the K new chains into the sorted index: old entry j moves up by the number of new chains that go at or below it
							// (chunks from the top, every chunk read before it is written; moved entries never collide: the shift grows with j)
							int minlo = 0x7FFFFFFF;
							for (unsigned long long mm = mnew; mm; mm &= mm - 1) { const int lv = ch_grp<W>::bcast(lo, (int)__builtin_ctzll(mm)); minlo = lv < minlo ? lv : minlo; }
							for (int hi = nc; hi > minlo; hi -= W) {
								const int jj = hi - 1 - lane;
								uint32_t ov = 0; int64_t op = 0;
								if (jj >= minlo) { ov = order[jj]; op = opos[jj]; }
								// new chains that go below this round's entries shift all of them alike (one count); only those that go INSIDE the round's window are
								// compared lane by lane -- K comparisons over all rounds instead of K per round (a 1500-entry read: 29 rounds of up to 64)
								int sh = (int)__builtin_popcountll(ch_grp<W>::ballot(isnew && lo <= hi - W - 1));
								for (unsigned long long mm = ch_grp<W>::ballot(isnew && lo >= hi - W && lo <= hi - 1); mm; mm &= mm - 1) { const int lv = ch_grp<W>::bcast(lo, (int)__builtin_ctzll(mm)); sh += lv <= jj ? 1 : 0; }
								ch_wave_fence<LDSX>();
								if (jj >= minlo && sh) { order[jj + sh] = ov; opos[jj + sh] = op; }
								ch_wave_fence<LDSX>();
							}
							if (isnew) { order[lo + below] = my_nc; opos[lo + below] = rbl; }
						}
						nc += K; ns += (int)__builtin_popcountll(madd);
						ch_wave_fence<LDSX>();
						done = true;
					}
				}
				CH_ACC_END(6, 9);
				if (!done) {
					const int m = n_it - c0 < W ? n_it - c0 : W;
					CH_ACC_BEGIN();
					for (int u = 0; u < m; ++u) seq_one((int64_t)ch_grp<W>::bcast64(rbl, u));
					CH_ACC_END(7, 8);
				}
			}
		} else
#endif
		for (int count = 0; count < n_it; ++count) seq_one((int64_t)g_rbeg[i + (int64_t)count * step]);
		i += (int)cnt;
	}
	if (nc == 0) return;

	// ---------------------------------------------------------------- mem_chain_flt
	CH_STAMP(1);
	int na = 0;
	auto weigh = [&](uint32_t ci) {                                       // mem_chain_weight + the chain's query span
		chain_t &c = CH[ci];
		int64_t end = 0; int w = 0;
		for (uint32_t p = c.head; p != NIL; p = S[p].next) {       // query cover
			const seed_t s = S[p];
			if (s.qbeg >= end) w += s.len;
			else if (s.qbeg + s.len > end) w += (int)(s.qbeg + s.len - end);
			end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
		}
		const int tmp = w; w = 0; end = 0;
		for (uint32_t p = c.head; p != NIL; p = S[p].next) {       // reference cover
			const seed_t s = S[p];
			if (s.rbeg >= end) w += s.len;
			else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
			end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
		}
		w = w < tmp ? w : tmp;
		w = w < 1 << 30 ? w : (1 << 30) - 1;
		c.w = w; c.first = -1; c.kept = 0;
		c.beg = S[c.head].qbeg; c.end = S[c.tail].qbeg + S[c.tail].len;
		return w;
	};
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) {                                                           // one chain per lane, kept in order
		const int lane = ch_grp<W>::lane();
		for (int b = 0; b < nc; b += W) {
			const int i = b + lane;
			uint32_t ci = 0; int w = -1;
			if (i < nc) { ci = order[i]; w = weigh(ci); opos[i] |= (int64_t)ci << 40; }       // (position rank -> chain, for the isolation pass below)
			const bool keep = i < nc && w >= o.min_chain_weight;
			const unsigned long long m = ch_grp<W>::ballot(keep);
			if (keep) srt[na + __builtin_popcountll(m & ((1ull << lane) - 1))] = (uint64_t)(uint32_t)w << 32 | ci;
			na += __builtin_popcountll(m);
		}
		ch_wave_fence<LDSX>();
	} else
#endif
	for (int i = 0; i < nc; ++i) {
		const uint32_t ci = order[i];
		const int w = weigh(ci);
		if (w >= o.min_chain_weight) srt[na++] = (uint64_t)(uint32_t)w << 32 | ci;
	}
	if (na == 0) return;
	CH_STAMP(2);
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) {
		if (!w_introsort_coop<LDSX, W>(srt, order, na)) { *x.err = 2; return; }
		if (na < 65536) w_place_coop<LDSX, W>(srt, order, na);
		else {                                                        // (more chains than the 16-bit index of the rank keys holds: a huge -c)
			w_insertion(srt, 0, na);
			for (int i = 0; i < na; ++i) order[i] = (uint32_t)srt[i];
			ch_wave_fence<LDSX>();
		}
	} else
#endif
	{
		if (!w_introsort(srt, na)) { *x.err = 2; return; }
		for (int i = 0; i < na; ++i) order[i] = (uint32_t)srt[i];
	}
	CH_STAMP(3);
	int nk = 0;
	bool kept_done = false;
#if defined(__HIP_DEVICE_COMPILE__)
	// seed-rich reads: span classes instead of a scan of the kept list per chain (srt is free between the sort and mem_chain2aln:
	// its 8 bytes per entry hold the kept weights and the class links)
#ifndef CH_NO_CLASSES
	// (with ALT contigs whether two chains "overlap" also depends on which of them is ALT, src/bwamem.c:518: not a function of the spans alone)
	// (the whole wave only: a class is owned by a lane and the minimum runs over the wave's DPP rows; a row's reads have at most 64 chains)
	if (COOP && W == 64 && na > 48 && !x.ctg_alt) kept_done = ch_kept_by_classes<LDSX>(o, CH, order, klist, (int32_t *)srt, (uint32_t *)srt + na, cidx, na, nk);
	if (COOP && W == 64 && na > 48 && !x.ctg_alt && !kept_done) ch_wave_fence<LDSX>();
#endif
#endif
	// kept chains: klist[k] = index in the sorted array, ks[k] = {beg, end, w, chain} so the scan reads one entry per k
	// (E is not in use before mem_chain2aln and has room for it)
	struct ks_t { int32_t beg, end, w; uint32_t chain; };                 // chain: index | is_alt << 31
	ks_t *ks = (ks_t *)E;
	if (!kept_done) {
		{ const chain_t c0 = CH[order[0]]; ks_t e; e.beg = c0.beg; e.end = c0.end; e.w = c0.w; e.chain = order[0] | chain_is_alt(x, c0.rid) << 31; ks[0] = e; }
		CH[order[0]].kept = 3; klist[nk++] = 0;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) ch_wave_fence<LDSX>();
#endif
	for (int i = 1; i < na && !kept_done; ++i) {
		const uint32_t ci = order[i];
		const chain_t ai = CH[ci];
		const uint32_t ai_alt = chain_is_alt(x, ai.rid);
		bool large_ovlp = false, broke = false;
		auto test = [&](const ks_t &aj, bool &ovl, bool &brk) {
			ovl = brk = false;
			const int b_max = aj.beg > ai.beg ? aj.beg : ai.beg;
			const int e_min = aj.end < ai.end ? aj.end : ai.end;
			if (e_min > b_max && (!(aj.chain >> 31) || ai_alt)) {        // (an overlap where the kept chain is ALT and this one is not does not count, :518)
				const int li = ai.end - ai.beg, lj = aj.end - aj.beg;
				const int min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * o.mask_level && min_l < o.max_chain_gap) {
					ovl = true;
					if (ai.w < aj.w * o.drop_ratio && aj.w - ai.w >= o.min_seed_len << 1) brk = true;
				}
			}
		};
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) {
			// 4 W kept chains per round, four per lane with independent loads; the scan ends behind the first chain that drops
			// chain i (src/bwamem.c:520-535), the marks of `first` stop there too
			const int lane = ch_grp<W>::lane();
			constexpr int U = W == 64 ? 4 : 1;
			for (int b = 0; b < nk && !broke; b += U * W) {
				ks_t aj[U]; bool ovl[U]; unsigned long long mb[U], mo[U];
#pragma unroll
				for (int u = 0; u < U; ++u) { const int k = b + W * u + lane; aj[u] = ks[k < nk ? k : nk - 1]; }   // U loads in flight
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const int k = b + W * u + lane;
					bool brk = false;
					test(aj[u], ovl[u], brk);
					ovl[u] = ovl[u] && k < nk; brk = brk && k < nk;
					mb[u] = ch_grp<W>::ballot(brk); mo[u] = ch_grp<W>::ballot(ovl[u]);
				}
#pragma unroll
				for (int u = 0; u < U; ++u) {
					if (broke) break;
					unsigned long long vm = ~0ull;
					if (mb[u]) { const int f = (int)__builtin_ctzll(mb[u]); vm = f == 63 ? ~0ull : ((1ull << (f + 1)) - 1); broke = true; }
					if (ovl[u] && ((vm >> lane) & 1)) { chain_t &cj = CH[aj[u].chain & 0x7FFFFFFFu]; if (cj.first < 0) cj.first = i; }
					if (mo[u] & vm) large_ovlp = true;
				}
			}
		} else
#endif
		{
			for (int k = 0; k < nk; ++k) {
				bool ovl, brk;
				const ks_t aj = ks[k];
				test(aj, ovl, brk);
				if (ovl) { large_ovlp = true; chain_t &cj = CH[aj.chain & 0x7FFFFFFFu]; if (cj.first < 0) cj.first = i; }
				if (brk) { broke = true; break; }
			}
		}
		if (!broke) {
			ks_t e; e.beg = ai.beg; e.end = ai.end; e.w = ai.w; e.chain = ci | ai_alt << 31; ks[nk] = e;
			klist[nk++] = (uint32_t)i; CH[ci].kept = large_ovlp ? 2 : 3;
		}
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) ch_wave_fence<LDSX>();
#endif
	}
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) {                                                           // (the marks are independent of each other: one kept chain per lane)
		for (int k = ch_grp<W>::lane(); k < nk; k += W) { const int f = CH[order[klist[k]]].first; if (f >= 0) CH[order[f]].kept = 1; }
		ch_wave_fence<LDSX>();
	} else
#endif
	for (int k = 0; k < nk; ++k) { const int f = CH[order[klist[k]]].first; if (f >= 0) CH[order[f]].kept = 1; }
	if (na >= o.max_chain_extend) {                                       // (fewer chains than max_chain_extend -- always, with the default of 2^30: the count below cannot reach it)
		int i, k;
		for (i = k = 0; i < na; ++i) {
			const uint32_t kp = CH[order[i]].kept;
			if (kp == 0 || kp == 3) continue;
			if (++k >= o.max_chain_extend) break;
		}
		for (; i < na; ++i) if (CH[order[i]].kept < 3) CH[order[i]].kept = 0;
	}

	// ---------------------------------------------------------------- mem_flt_chained_seeds :970-991
	if (FLT && flt_on) {
		// the seeds of the kept chains, flat (cidx is free until mem_chain2aln); every seed's score is independent of the others
		int nsd = 0;
		for (int ia = 0; ia < na; ++ia) {
			const chain_t c = CH[order[ia]];
			if (c.kept == 0) continue;
			for (uint32_t p = c.head; p != NIL; p = S[p].next) cidx[nsd++] = p;
		}
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) {
			ch_wave_fence<LDSX>();
			for (int k = ch_grp<W>::lane(); k < nsd; k += W) { const uint32_t p = cidx[k]; S[p].pad = (uint32_t)seed_sw(x, r, l_query, S[p]); }
			ch_wave_fence<LDSX>();
		} else
#endif
		for (int k = 0; k < nsd; ++k) { const uint32_t p = cidx[k]; S[p].pad = (uint32_t)seed_sw(x, r, l_query, S[p]); }
		// the weak seeds leave their chains (:979-987); a chain may end up empty: mem_chain2aln returns at once for it (:1187)
		for (int ia = 0; ia < na; ++ia) {
			chain_t c = CH[order[ia]];
			if (c.kept == 0) continue;
			uint32_t head = NIL, last = NIL, cnt = 0;
			for (uint32_t p = c.head; p != NIL;) {
				const uint32_t nx = S[p].next;
				const int scv = (int)S[p].pad;
				if (scv < 0 || scv >= min_HSP_score) {
					S[p].pad = (uint32_t)(scv < 0 ? S[p].len * o.a : scv);
					if (last == NIL) head = p; else S[last].next = p;
					last = p; ++cnt;
				}
				p = nx;
			}
			if (last != NIL) S[last].next = NIL;
			if (cnt != c.n) {
				if (cnt == 0) c.kept = 0; else { c.head = head; c.tail = last; }
				c.n = cnt;
				CH[order[ia]] = c;
			}
		}
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) ch_wave_fence<LDSX>();
#endif
	}

	// ---------------------------------------------------------------- mem_chain2aln, chain by chain in filtered order
	CH_STAMP(4);
#if defined(__HIP_DEVICE_COMPILE__)
	if (COOP) {
		// Which chains are ISOLATED on the reference?  A region made from seed t covers seed s only if s.rbeg lies within l_query of
		// t.rbeg (rb_est >= t.rbeg - 0.85 (t.qbeg + 1), re_est <= t.rbeg + t.len + 0.85 (l_query - t.qbeg - t.len), :1235-1256), and
		// the seeds of a chain lie between its position and its last seed (rbeg never decreases along a chain, test_and_merge
		// :337-364).  So a chain whose neighbours in position order are farther than 2 l_query from its ends can neither cover nor be
		// covered by a region of ANOTHER chain: the "made before?" scan of its seeds needs the chain's own regions only.  On a
		// seed-rich read almost every chain is a single seed at its own locus, and the scan over all earlier regions was half of
		// this phase.  opos still holds the chains in position order (with the chain index in its high bits since the weights).
		const int lane = ch_grp<W>::lane();
		const int64_t PMASK = ((int64_t)1 << 40) - 1;
		int64_t run_max = -((int64_t)1 << 60);
		for (int b = 0; b < nc; b += W) {
			const int r = b + lane;
			int64_t pos = 0, tail = -((int64_t)1 << 60), nextpos = (int64_t)1 << 60; uint32_t ci = 0;
			if (r < nc) {
				const int64_t v = opos[r]; pos = v & PMASK; ci = (uint32_t)(v >> 40);
				tail = S[CH[ci].tail].rbeg;
				if (r + 1 < nc) nextpos = opos[r + 1] & PMASK;
			}
			int64_t incl = tail;
#pragma unroll
			for (int d = 1; d < W; d <<= 1) { const int64_t t = ch_grp<W>::shfl_up64(incl, d); if (lane >= d && t > incl) incl = t; }
			int64_t excl = ch_grp<W>::shfl_up64(incl, 1);
			if (lane == 0 || excl < run_max) excl = run_max;
			const bool iso = excl + 2 * (int64_t)l_query < pos && nextpos > tail + 2 * (int64_t)l_query;
			if (r < nc) CH[ci].pad = iso ? 1u : 0u;
			const int64_t tot = ch_grp<W>::bcast64(incl, W - 1);
			run_max = tot > run_max ? tot : run_max;
		}
		ch_wave_fence<LDSX>();
	}
#endif
	int n_regs = 0, n_jobs = 0;
	for (int ia = 0; ia < na; ++ia) {
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) {
			// A run of chains that need no look at any other region -- dropped ones (kept == 0) and ISOLATED chains of a single seed
			// (their "made before?" scan is empty: no other chain's region can cover the seed and the chain has no earlier region of
			// its own) -- is handled one chain per lane: the region of such a chain is a function of its seed alone, and its slot is
			// its rank in the run.  The chains of a seed-rich read are almost all of this kind.
			const int lane = ch_grp<W>::lane();
			const int i = ia + lane;
			const bool in = i < na;
			chain_t cl; cl.kept = 0; cl.n = 0; cl.pad = 0; cl.head = 0;
			if (in) cl = CH[order[i]];
			const unsigned long long im = ch_grp<W>::ballot(in), slow = im & ~ch_grp<W>::ballot(in && (cl.kept == 0 || (cl.pad != 0 && cl.n == 1)));
			const int run = slow ? (int)__builtin_ctzll(slow) : (int)__builtin_popcountll(im);
			if (run > 0) {
				const bool me = lane < run && cl.kept != 0;
				const unsigned long long mm = ch_grp<W>::ballot(me);
				bool jl = false, jr = false;
				if (me) {
					const int idx = n_regs + (int)__builtin_popcountll(mm & ((1ull << lane) - 1));
					const seed_t t = S[cl.head];
					int64_t rmax0 = t.rbeg - (t.qbeg + cal_max_gap(o, t.qbeg));
					int64_t rmax1 = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + cal_max_gap(o, l_query - t.qbeg - t.len));
					rmax0 = rmax0 > 0 ? rmax0 : 0;
					rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
					if (rmax0 < l_pac && l_pac < rmax1) { if (t.rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
					{
						int is_rev;
						const int rid = pos2rid<false>(x, depos(x, t.rbeg, &is_rev));
						int64_t far_beg = x.n_contigs > 1 ? x.ctg_off[rid] : 0, far_end = far_beg + (x.n_contigs > 1 ? x.ctg_len[rid] : l_pac);
						if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
						rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
						rmax1 = rmax1 < far_end ? rmax1 : far_end;
					}
					ch_reg_t a; est_t e;
					const int fwd = (int)(0.85 * (l_query - (t.qbeg + t.len)));
					e.qe_est = (t.qbeg + t.len) + fwd < l_query ? (t.qbeg + t.len) + fwd : l_query;
					e.re_est = (t.rbeg + t.len) + fwd < l_pac << 1 ? (t.rbeg + t.len) + fwd : l_pac << 1;
					const int back = (int)(0.85 * (t.qbeg + 1));
					e.qb_est = (t.qbeg - back) > 0 ? (t.qbeg - back) : 0;
					e.rb_est = (t.rbeg - back) > 0 ? (t.rbeg - back) : 0;
					if (e.rb_est < l_pac && l_pac < e.qe_est) { if (t.rbeg < l_pac) e.re_est = l_pac; else e.rb_est = l_pac; }
					e.seedlen0 = t.len; e.pad = 0;
					E[idx] = e;
					a.seed_rbeg = t.rbeg; a.seed_qbeg = t.qbeg; a.seedlen0 = t.len; a.rmax0 = rmax0;
					a.lr = (int)(t.rbeg - rmax0);
					a.rq = l_query - (t.qbeg + t.len);
					a.rr = (int)(rmax1 - rmax0) - (a.lr + t.len);
					a.pad = 0;
					R[idx] = a;
					jl = t.qbeg > 0; jr = a.rq > 0;
				}
				n_regs += (int)__builtin_popcountll(mm);
				n_jobs += (int)__builtin_popcountll(ch_grp<W>::ballot(jl)) + (int)__builtin_popcountll(ch_grp<W>::ballot(jr));
				ia += run - 1;                                                    // (the loop adds the last one)
				ch_wave_fence<LDSX>();
				continue;
			}
		}
#endif
		const chain_t c = CH[order[ia]];
		if (c.kept == 0) continue;
		const int cn = (int)c.n;
		const int scan_from = c.pad ? n_regs : 0;                               // isolated chain: only its own regions can cover its seeds
		int64_t rmax0 = l_pac << 1, rmax1 = 0;
		{
			int i = 0;
			for (uint32_t p = c.head; p != NIL; p = S[p].next, ++i) {
				const seed_t t = S[p];
				const int64_t b = t.rbeg - (t.qbeg + cal_max_gap(o, t.qbeg));
				const int64_t e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + cal_max_gap(o, l_query - t.qbeg - t.len));
				rmax0 = rmax0 < b ? rmax0 : b;
				rmax1 = rmax1 > e ? rmax1 : e;
				cidx[i] = p;
				srt[i] = (uint64_t)(FLT && flt_on ? t.pad : (uint32_t)t.len) << 32 | (uint32_t)i;   // score == len unless the seed filter re-scored it
			}
		}
		rmax0 = rmax0 > 0 ? rmax0 : 0;
		rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
		const seed_t s0 = S[c.head];
		if (rmax0 < l_pac && l_pac < rmax1) { if (s0.rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
		{   // bns_fetch_seq clips the window to the contig of the first seed (src/bntseq.c:531-556)
			int is_rev;
			const int rid = pos2rid<COOP, W>(x, depos(x, s0.rbeg, &is_rev));
			int64_t far_beg = x.n_contigs > 1 ? x.ctg_off[rid] : 0, far_end = far_beg + (x.n_contigs > 1 ? x.ctg_len[rid] : l_pac);
			if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
			rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
			rmax1 = rmax1 < far_end ? rmax1 : far_end;
		}
#if defined(__HIP_DEVICE_COMPILE__)
		if (COOP) ch_wave_fence<LDSX>();
#endif
		sort_distinct<COOP, LDSX, W>(srt, (uint64_t *)(opos), cn);    // opos is free by now (8 bytes per entry)
		for (int k = cn - 1; k >= 0; --k) {
			const seed_t s = S[cidx[(uint32_t)srt[k]]];
			auto covered = [&](const est_t &p) {                                 // extension (estimated) made before? :1235-1256
				if (s.rbeg < p.rb_est || s.rbeg + s.len > p.re_est || s.qbeg < p.qb_est || s.qbeg + s.len > p.qe_est) return false;
				if (s.len - p.seedlen0 > .1 * l_query) return false;
				int qd = s.qbeg - p.qb_est; int64_t rd = s.rbeg - p.rb_est;
				int max_gap = cal_max_gap(o, qd < rd ? qd : (int)rd);
				int w = max_gap < o.w ? max_gap : o.w;
				if (qd - rd < w && rd - qd < w) return true;
				qd = p.qe_est - (s.qbeg + s.len); rd = p.re_est - (s.rbeg + s.len);
				max_gap = cal_max_gap(o, qd < rd ? qd : (int)rd);
				w = max_gap < o.w ? max_gap : o.w;
				return qd - rd < w && rd - qd < w;
			};
			int hit = n_regs;
#if defined(__HIP_DEVICE_COMPILE__)
			if (COOP) {
				// 4 W regions per round; the four entries of a lane are loaded before any is tested, so the loads overlap
				const int lane = ch_grp<W>::lane();
				constexpr int U = W == 64 ? 4 : 1;
				for (int b = scan_from; b < n_regs && hit == n_regs; b += U * W) {
					est_t p4[U]; unsigned long long m[U];
#pragma unroll
					for (int u = 0; u < U; ++u) { const int i = b + W * u + lane; p4[u] = E[i < n_regs ? i : n_regs - 1]; }
#pragma unroll
					for (int u = 0; u < U; ++u) { const int i = b + W * u + lane; m[u] = ch_grp<W>::ballot(i < n_regs && covered(p4[u])); }
#pragma unroll
					for (int u = U - 1; u >= 0; --u) if (m[u]) hit = b + W * u + (int)__builtin_ctzll(m[u]);
				}
			} else
#endif
			for (int i = 0; i < n_regs; ++i) if (covered(E[i])) { hit = i; break; }
			if (hit < n_regs) {                                                   // :1258-1276
				const int j = first_true<COOP, W>(k + 1, cn, [&](int j) {
					if (srt[j] == 0) return false;
					const seed_t t = S[cidx[(uint32_t)srt[j]]];
					if (t.len < s.len * .95) return false;
					if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) return true;
					if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) return true;
					return false;
				});
				if (j == cn) {
					srt[k] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
					if (COOP) ch_wave_fence<LDSX>();
#endif
					continue;
				}
			}
			ch_reg_t a; est_t e;
			const int fwd = (int)(0.85 * (l_query - (s.qbeg + s.len)));           // FILTER_COEF, :52, :1285-1298
			e.qe_est = (s.qbeg + s.len) + fwd < l_query ? (s.qbeg + s.len) + fwd : l_query;
			e.re_est = (s.rbeg + s.len) + fwd < l_pac << 1 ? (s.rbeg + s.len) + fwd : l_pac << 1;
			const int back = (int)(0.85 * (s.qbeg + 1));
			e.qb_est = (s.qbeg - back) > 0 ? (s.qbeg - back) : 0;
			e.rb_est = (s.rbeg - back) > 0 ? (s.rbeg - back) : 0;
			if (e.rb_est < l_pac && l_pac < e.qe_est) { if (s.rbeg < l_pac) e.re_est = l_pac; else e.rb_est = l_pac; }   // (sic) qe_est, :1292
			e.seedlen0 = s.len; e.pad = 0;
			E[n_regs] = e;
			a.seed_rbeg = s.rbeg; a.seed_qbeg = s.qbeg; a.seedlen0 = s.len; a.rmax0 = rmax0;
			a.lr = (int)(s.rbeg - rmax0);
			a.rq = l_query - (s.qbeg + s.len);
			a.rr = (int)(rmax1 - rmax0) - (a.lr + s.len);
			a.pad = 0;
			R[n_regs++] = a;
			n_jobs += (s.qbeg > 0) + (a.rq > 0);
#if defined(__HIP_DEVICE_COMPILE__)
			if (COOP) ch_wave_fence<LDSX>();
#endif
		}
	}
	CH_STAMP(5);
	x.regs_per_read[r] = (uint32_t)n_regs; x.jobs_per_read[r] = (uint32_t)n_jobs;
}

} // namespace chain_core
