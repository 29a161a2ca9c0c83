// The region tail on the device (SURVEY.md 8f rank 1 tail + the selection of rank 4): what the reference's host threads do with
// the regions of a read after the extension -- mem_sort_dedup_patch, mem_patch_reg, mem_mark_primary_se, mem_approx_mapq_se, the
// selection of mem_reg2sam (/root/reference/src/bwamem.c:581-680, 685-760, 1690-1717, 1721-1770) -- on the regions where the
// merge kernel left them in HBM, so that reads -> reportable alignments never leaves the device (bmh_finalize_regs is the host
// form).  The per-read logic is csrc/regs_core.h, the same code the CPU test builds.
//   fin_lane_kernel   lane per read, records in registers / private memory: reads with at most FIN_NL regions (95 % of the reads
//                     of an hg38-like batch); a read whose patch test reaches its global alignment (16 per million there) is
//                     handed on
//   fin_wave_kernel   wave per read, records in LDS (up to FIN_NMAX regions; in HBM beyond): the three sorts are klib's introsort
//                     EXACTLY, done by the wave -- every record gets the number of records that sort ahead of it as its rank
//                     (64 comparisons per step), the ranks go through the cooperative introsort of the chaining stage
//                     (chain_core.h: w_introsort_coop / w_place_coop make the exchanges ks_introsort would make, so ties fall as
//                     in the reference), the records are moved once; the loops between the sorts run on lane 0
//   fin_compact_kernel  per-read slots -> one array in read order
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>
#include "bmh_internal.h"
#include "chain_core.h"
#include "regs_core.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

using namespace regs_core;

#define FIN_NL 8          // regions of a read the lane kernel takes
#define FIN_NMAX 512      // regions of a read the wave kernels keep in LDS (classes of 32, 128 and 512 records; in HBM beyond)
#define FIN_NCLS 5
#define FIN_DPCAP 1024    // query columns (+2) of the patch test's alignment (scratch in HBM, one pair of rows per block: the alignment is rare)
#define FIN_WAVE_GRID 2048
#define FIN_CTG_LDS 512     // contig tables up to this size are copied to LDS by the wave kernels
#define FIN_NLOG 65536

#define FIN_NCLS_PROF 5
// optional phase stamps of the wave kernels (100 MHz ticks summed over the reads of a class), for tuning: compile with -DFIN_PROFILE
// (collected in LDS by the block's one wave, flushed when the block ends)
#if defined(FIN_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
__shared__ unsigned long long g_lprof[16];
#define FIN_STAMP(k) do { const long long t_ = (long long)wall_clock64(); if (chain_core::ch_lane() == 0) g_lprof[k] += (unsigned long long)(t_ - t_prev); t_prev = (long long)wall_clock64(); } while (0)
#define FIN_STAMP_BEGIN() long long t_prev = (long long)wall_clock64()
#else
#define FIN_STAMP(k) do { } while (0)
#define FIN_STAMP_BEGIN() do { } while (0)
#endif
__device__ unsigned long long g_fin_prof[FIN_NCLS_PROF][16];

struct fin_args_t {
	ctx_t x;
	const uint8_t *reads; const uint32_t *read_offs;
	const int32_t *regs_in; const uint32_t *rpr, *in_off; const float *frac_rep;
	int32_t *work, *work2;            // [n_regs][16] records at in_off; work2 + the arrays below: scratch of reads beyond FIN_NMAX regions
	uint64_t *g_keys, *g_k128; uint32_t *g_tmp, *g_order; int32_t *g_z;
	uint32_t *opr; uint32_t n_reads;
	uint32_t *defer, *ctr;            // defer[c * n_reads ..]: reads handed to wave class c; ctr[c] their number, ctr[8 + c] next to be drawn, ctr[16] error
	int32_t *g_dp;                    // [FIN_NCLS][FIN_WAVE_GRID][2][FIN_DPCAP]
	int32_t *dedup_out;               // optional [n_regs][16] at in_off: the regions as mem_sort_dedup_patch leaves them (before the marking sorts them), [0] = read
};
__device__ __forceinline__ int fin_class(int n) { return n <= 32 ? 0 : n <= 128 ? 1 : n <= 256 ? 2 : n <= FIN_NMAX ? 3 : 4; }

__global__ void __launch_bounds__(256) fin_lane_kernel(fin_args_t A)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= A.n_reads) return;
	const int n_in = (int)A.rpr[r];
	const uint32_t off = A.in_off[r];
	if (n_in == 0) { A.opr[r] = 0; return; }
	bool defer = n_in > FIN_NL;
	if (!defer) {
		rec_t a[FIN_NL]; int32_t z[FIN_NL];
		for (int i = 0; i < n_in; ++i) {
			const int4 *src = (const int4 *)(A.regs_in + 8 * (size_t)(off + i));
			const int4 u = src[0], v = src[1];
			a[i].v[0] = u.x; a[i].v[1] = u.y; a[i].v[2] = u.z; a[i].v[3] = u.w; a[i].v[4] = v.x; a[i].v[5] = v.y; a[i].v[6] = v.z; a[i].v[7] = v.w;
		}
		int n;
		if (!A.dedup_out) n = finalize_read<true, 1>(A.x, A.reads + A.read_offs[r], r, A.x.po.id0 + r, A.frac_rep ? A.frac_rep[r] : 0.f, n_in, a, z);
		else {                                                      // (finalize_read with a copy of the regions between its two halves)
			for (int i = 0; i < n_in; ++i) init_one(A.x, a[i]);
			n = sort_dedup_patch<true, 1>(A.x, A.reads + A.read_offs[r], n_in, a);
			if (n >= 0) {
				for (int i = 0; i < n; ++i) {
					int4 *dst = (int4 *)(A.dedup_out + 16 * (size_t)(off + i));
					dst[0] = make_int4((int)r, a[i].v[1], a[i].v[2], a[i].v[3]); dst[1] = make_int4(a[i].v[4], a[i].v[5], a[i].v[6], a[i].v[7]);
					dst[2] = make_int4(a[i].v[8], a[i].v[9], a[i].v[10], a[i].v[11]); dst[3] = make_int4(a[i].v[12], r_seq(a[i]), a[i].v[14], a[i].v[15]);   // (the sequence without the ALT bit)
				}
				mark_primary<1>(A.x, n, a, A.x.po.id0 + r, z);
				n = emit_all(A.x, r, A.frac_rep ? A.frac_rep[r] : 0.f, n, a, z);
			}
		}
		if (n == -NEED_DP) defer = true;
		else if (n < 0) { A.ctr[16] = (uint32_t)-n; A.opr[r] = 0; return; }
		else {
			for (int i = 0; i < n; ++i) {
				int4 *dst = (int4 *)(A.work + 16 * (size_t)(off + i));
				dst[0] = make_int4(a[i].v[0], a[i].v[1], a[i].v[2], a[i].v[3]); dst[1] = make_int4(a[i].v[4], a[i].v[5], a[i].v[6], a[i].v[7]);
				dst[2] = make_int4(a[i].v[8], a[i].v[9], a[i].v[10], a[i].v[11]); dst[3] = make_int4(a[i].v[12], a[i].v[13], a[i].v[14], a[i].v[15]);
			}
			A.opr[r] = (uint32_t)n;
		}
	}
	if (defer) { const int c = fin_class(n_in); A.defer[(size_t)c * A.n_reads + atomicAdd(&A.ctr[c], 1u)] = r; }
}

// ---- the wave form
struct wptr_t { rec_t *a, *b; uint64_t *keys, *k128; uint32_t *tmp, *order; int32_t *z; };

enum { KEY_RE = 0, KEY_SCORE_RB_QB = 1, KEY_SCORE_HASH = 2, KEY_ALT_SCORE_HASH = 3 };
// 128-bit key whose unsigned lexicographic order is the comparator's strict weak order (lt_re / lt_score_rb_qb / lt_score_hash)
template <int WHICH> __device__ __forceinline__ void fin_key(const rec_t &r, uint64_t &hi, uint64_t &lo)
{
	if (WHICH == KEY_RE) { hi = 0; lo = (uint64_t)r_re(r); }
	else if (WHICH == KEY_SCORE_RB_QB) { hi = (uint64_t)(0x7FFFFFFFll - (long long)r.v[1]); lo = ((uint64_t)r_rb(r) << 24) | (uint64_t)(uint32_t)(r.v[2] & 0xFFFFFF); }
	else if (WHICH == KEY_SCORE_HASH) { hi = (uint64_t)(0x7FFFFFFFll - (long long)r.v[1]) << 1 | (uint64_t)r_alt(r); lo = r_hash(r); }      // lt_score_hash: score, primary assembly first, hash
	else { hi = (uint64_t)r_alt(r) << 32 | (uint64_t)(0x7FFFFFFFll - (long long)r.v[1]); lo = r_hash(r); }                                      // lt_alt_score_hash
}

// a[0..n) sorted as r_introsort<lt> would leave it (same permutation: ties included).  All 64 lanes.  Every record gets its rank,
// the number of records that sort strictly ahead of it (64 comparisons per step; with the records in HBM the keys are staged through
// `stage`, FIN_STAGE of them at a time).  Without two equal keys -- always so for the hash-tie-broken order of
// mem_mark_primary_se, nearly always for the other two -- the rank IS the place; with ties the ranks go through klib's introsort as
// the wave does it in the chaining stage, which exchanges what ks_introsort would exchange.  Returns false on an internal limit
// (stack of the introsort).
#define FIN_STAGE 1024
template <int WHICH, bool STAGED> __device__ bool fin_wave_sort(wptr_t &P, const int n, uint64_t *stage)
{
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	if (n < 2) return true;
	const int lane = ch_lane();
	FIN_STAMP_BEGIN();
	// One 64-bit key per record where the order allows it: the end position (KEY_RE); (score, rb, qb) packed when every score of the
	// read lies in 0 .. 16383 (KEY_SCORE_RB_QB); the score and the top 33 bits of the hash (KEY_SCORE_HASH: two records that agree in
	// those are told apart by the full keys in a second round, which a batch of a million reads does not see).  Half the LDS traffic and
	// comparisons of the 128-bit form.
	bool small = true;
	for (int x = lane; x < n; x += 64) {
		uint64_t h, l; fin_key<WHICH>(P.a[x], h, l);
		P.k128[2 * x] = h; P.k128[2 * x + 1] = l;
		const int sc_ = P.a[x].v[1];
		uint64_t k64 = l;
		if (WHICH == KEY_SCORE_RB_QB) { small = small && sc_ >= 0 && sc_ < 16384 && (uint32_t)P.a[x].v[2] < 65536u; k64 = ((uint64_t)(16383 - sc_) << 50) | ((uint64_t)r_rb(P.a[x]) << 16) | (uint64_t)(uint32_t)(P.a[x].v[2] & 0xFFFF); }
		if (WHICH == KEY_SCORE_HASH) { small = small && sc_ >= 0; k64 = (h << 32) | (l >> 32); }
		if (WHICH == KEY_ALT_SCORE_HASH) { small = small && sc_ >= 0; k64 = (h << 31) | (l >> 33); }
		P.keys[x] = k64;
	}
	small = !__any(!small);
	ch_wave_fence<false>();
	bool ties = false;
	bool wide = !small;                                          // compare the 128-bit keys
	for (int round = 0; round < 2; ++round) {
		ties = false;
		for (int xb = 0; xb < n; xb += 64 * 4) {
			uint64_t mh[4], ml[4]; int rk[4], eq[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int x = xb + 64 * u + lane, xc = x < n ? x : n - 1;
				mh[u] = wide ? P.k128[2 * xc] : 0; ml[u] = wide ? P.k128[2 * xc + 1] : P.keys[xc]; rk[u] = 0; eq[u] = 0;
			}
			for (int y0 = 0; y0 < n; y0 += FIN_STAGE) {
				const int ny = n - y0 < FIN_STAGE ? n - y0 : FIN_STAGE;
				const uint64_t *src = wide ? P.k128 + 2 * (size_t)y0 : P.keys + y0;
				if (STAGED) {
					ch_wave_fence<false>();
					for (int k = lane; k < (wide ? 2 : 1) * ny; k += 64) stage[k] = src[k];
					ch_wave_fence<false>();
					src = stage;
				}
				if (wide) {
					for (int y = 0; y < ny; ++y) {
						const uint64_t yh = src[2 * y], yl = src[2 * y + 1];
#pragma unroll
						for (int u = 0; u < 4; ++u) { rk[u] += (yh < mh[u] || (yh == mh[u] && yl < ml[u])) ? 1 : 0; eq[u] += (yh == mh[u] && yl == ml[u]) ? 1 : 0; }
					}
				} else {
					int y = 0;
					for (; y + 8 <= ny; y += 8) {
						uint64_t yk[8];
#pragma unroll
						for (int v = 0; v < 8; ++v) yk[v] = src[y + v];
#pragma unroll
						for (int v = 0; v < 8; ++v) {
#pragma unroll
							for (int u = 0; u < 4; ++u) { rk[u] += yk[v] < ml[u] ? 1 : 0; eq[u] += yk[v] == ml[u] ? 1 : 0; }
						}
					}
					for (; y < ny; ++y) {
						const uint64_t yl = src[y];
#pragma unroll
						for (int u = 0; u < 4; ++u) { rk[u] += yl < ml[u] ? 1 : 0; eq[u] += yl == ml[u] ? 1 : 0; }
					}
				}
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int x = xb + 64 * u + lane;
				if (x < n) { P.tmp[x] = (uint32_t)rk[u]; ties = ties || eq[u] > 1; }
			}
		}
		ties = __any(ties);
		// equal short keys of the hash order are not ties of the order itself: once more with the full keys
		if (ties && !wide && (WHICH == KEY_SCORE_HASH || WHICH == KEY_ALT_SCORE_HASH)) { wide = true; continue; }
		break;
	}
	ch_wave_fence<false>();
	for (int x = lane; x < n; x += 64) P.keys[x] = ((uint64_t)(uint32_t)(n - (int)P.tmp[x]) << 32) | (uint32_t)x;
	ch_wave_fence<false>();
	FIN_STAMP(10);
	if (!ties) {
		for (int x = lane; x < n; x += 64) P.order[n - (int)(P.keys[x] >> 32)] = (uint32_t)x;
		ch_wave_fence<false>();
	} else {
		if (!w_introsort_coop<false>(P.keys, P.tmp, n)) return false;
		FIN_STAMP(11);
		if (n < 65536) w_place_coop<false>(P.keys, P.order, n);
		else {
			if (lane == 0) { w_insertion(P.keys, 0, n); for (int i = 0; i < n; ++i) P.order[i] = (uint32_t)P.keys[i]; }
			ch_wave_fence<false>();
		}
	}
	FIN_STAMP(12);
	for (int pos = lane; pos < n; pos += 64) P.b[pos] = P.a[P.order[pos]];
	ch_wave_fence<false>();
	FIN_STAMP(13);
	rec_t *t = P.a; P.a = P.b; P.b = t;
#endif
	return true;
}

__device__ __forceinline__ int fin_bcast(int v) { return __builtin_amdgcn_readfirstlane(v); }

// a[i] with keep(i) -> b[0..m) in order, a and b change places; returns m.  All 64 lanes.
template <class KEEP> __device__ int fin_wave_compact(wptr_t &P, const int n, KEEP keep)
{
	int m = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	const int lane = ch_lane();
	for (int b = 0; b < n; b += 64) {
		const int i = b + lane;
		const bool k = i < n && keep(i);
		const unsigned long long mk = __ballot(k);
		if (k) P.b[m + (int)__builtin_popcountll(mk & ((1ull << lane) - 1ull))] = P.a[i];
		m += (int)__builtin_popcountll(mk);
	}
	ch_wave_fence<false>();
	rec_t *t = P.a; P.a = P.b; P.b = t;
#endif
	return m;
}

// The patch test's global alignment (regs_core.h: gen_score) by the wave, one target row per step: M of a row depends on the previous
// row only, E on the column above, and F(j) = max over k < j of M(k) - oe_ins - e_ins (j - 1 - k) is a max-plus prefix scan along
// the row (ksw_global2 opens its gaps from M, src/ksw.c:1190-1200), so the band's columns are spread over the lanes.  Values that
// the sequential code carries at "minus infinity" (-0x40000000 less a few gap extensions) may differ here by such small amounts;
// they never reach the result: every step adds less than 2^20 to a value, a band of w >= |rlen - qlen| + 3 always holds a real path.
__device__ int fin_wave_gen_score(const ctx_t &x, int w_, int l_query, const uint8_t *query, int64_t rb, int64_t re, int *err)
{
	int result = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	const bmh_ext_params_t &p = x.ep;
	const int lane = ch_lane();
	int rlen = 0, w = 0; bool flip = false;
	const int plan = gen_plan(x, w_, l_query, rb, re, &rlen, &flip, &w);
	if (plan == 0) return 0;
	auto tb = [&](int i) { return text_base(x.pac, x.l_pac, flip ? re - 1 - i : rb + i); };
	auto qb = [&](int i) { return qbase<true>(query, flip ? l_query - 1 - i : i); };
	if (plan == 1) {
		int sum = 0;
		for (int i = lane; i < l_query; i += 64) sum += sc(p, tb(i), qb(i));
		for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d);
		return sum;
	}
	if (l_query + 2 > x.dp_cap) { *err = E_DPCAP; return 0; }
	if (2 * w + 1 > 512) {                                        // a band beyond eight columns per lane: lane 0 alone, as in the core
		int sc1 = 0, e1 = 0;
		if (lane == 0) sc1 = gen_score<true>(x, w_, l_query, query, rb, re, &e1);
		*err = __builtin_amdgcn_readfirstlane(e1);
		ch_wave_fence<false>();
		return __builtin_amdgcn_readfirstlane(sc1);
	}
	const int NEG = -0x40000000, oe_del = p.o_del + p.e_del, oe_ins = p.o_ins + p.e_ins, qlen = l_query;
	int32_t *Hd = x.dp_h, *E = x.dp_e;
	for (int j = lane; j <= qlen; j += 64) { Hd[j] = j == 0 ? 0 : (j <= w ? -(p.o_ins + p.e_ins * j) : NEG); E[j] = NEG; }
	ch_wave_fence<false>();
	for (int i = 0; i < rlen; ++i) {
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int ti = tb(i);
		const int C = (end - beg + 63) >> 6;                       // columns per lane (<= 8), contiguous: lane l owns beg + l C .. beg + l C + C - 1
		const int j0 = beg + lane * C;
		int mbuf[8], ebuf[8];
		int gown = NEG;                                           // largest g = M + e_ins k of the lane's columns: what they send to the columns on their right
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const int j = j0 + c;
			mbuf[c] = NEG; ebuf[c] = NEG;
			if (c < C && j < end) {
				mbuf[c] = Hd[j] + sc(p, ti, qb(j)); ebuf[c] = E[j];
				const int gk = mbuf[c] + p.e_ins * j; gown = gown > gk ? gown : gk;
			}
		}
		int incl = gown;                                          // inclusive max-scan over the lanes, then the exclusive value
		for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl = incl > t ? incl : t; }
		int G = __shfl_up(incl, 1);
		if (lane == 0) G = NEG;
		ch_wave_fence<false>();                                   // every lane has read its Hd / E: Hd[j + 1] below is the next lane's first column
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const int j = j0 + c;
			if (c < C && j < end) {
				const int m = mbuf[c];
				int e = ebuf[c];
				const int f = G <= NEG ? NEG : G - oe_ins - p.e_ins * (j - 1);
				int h = m >= e ? m : e;
				h = h >= f ? h : f;
				Hd[j + 1] = h;
				const int y = m - oe_del; e -= p.e_del; E[j] = e > y ? e : y;
				const int gk = m + p.e_ins * j; G = G > gk ? G : gk;
			}
		}
		if (lane == 0) { Hd[beg] = beg == 0 ? -(p.o_del + p.e_del * (i + 1)) : NEG; E[end] = NEG; }
		ch_wave_fence<false>();
	}
	result = Hd[qlen];
#endif
	return result;
}

// Iteration i of mem_sort_dedup_patch's scan (regs_core.h: dedup_one) with the regions of its window looked at 64 at a time.  What
// happens to a pair (i, j) depends on the other pairs only through region i itself, and region i changes only when it is emptied
// (the scan ends) or when the patch test reaches its global alignment (rare): up to the first such pair of a step every lane settles
// its own j -- the redundant lower-scoring j's are emptied --, that pair is done by lane 0 with the sequential code, and the scan
// goes on behind it with region i as it then is.  (A read inside a tandem array has all its regions within max_chain_gap of one
// another: sequentially this scan was most of the tail of such reads.)
__device__ int fin_wave_dedup_one(const ctx_t &x, const uint8_t *query, const int i, rec_t *a)
{
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	const int lane = ch_lane();
	int jtop = i - 1;
	while (jtop >= 0) {
		const int j = jtop - lane;
		const rec_t p = a[i];
		rec_t q = p;
		if (j >= 0) q = a[j];
		const bool inwin = j >= 0 && dedup_in_window(x, p, q);
		const unsigned long long outm = __ballot(!inwin);
		const int first_out = outm ? (int)__builtin_ctzll(outm) : 64;
		const bool act = lane < first_out && q.v[3] != q.v[2];
		bool red = false, pdies = false, pev = false;
		if (act) {
			int w;
			red = dedup_redundant(x, p, q);
			pdies = red && p.v[1] < q.v[1];
			pev = !red && r_rb(q) < r_rb(p) && patch_pre(x, q, p, &w);
		}
		const unsigned long long evm = __ballot(act && (pdies || pev));
		const int first_ev = evm ? (int)__builtin_ctzll(evm) : 64;
		if (act && red && !pdies && lane < first_ev) a[j].v[3] = a[j].v[2];
		ch_wave_fence<false>();
		if (first_ev < 64) {
			const int jev = jtop - first_ev;
			const bool ev_dies = __builtin_amdgcn_readlane(pdies ? 1 : 0, first_ev) != 0;
			if (ev_dies) {                                            // region i is the lower-scoring one of a redundant pair: emptied, its scan ends
				if (lane == 0) a[i].v[3] = a[i].v[2];
				ch_wave_fence<false>();
				return 0;
			}
			// the patch test reaches its global alignment: the alignment by the whole wave, the verdict and the merge by lane 0
			int w = 0, err = 0;
			const rec_t pe = a[i], qe = a[jev];
			patch_pre(x, qe, pe, &w);
			const int score = fin_wave_gen_score(x, w, pe.v[3] - qe.v[2], query + qe.v[2], r_rb(qe), r_re(pe), &err);
			if (err) return err;
			if (lane == 0 && patch_accept(qe, pe, score) > 0) dedup_patch_apply(&a[i], &a[jev], score, w);
			ch_wave_fence<false>();
			jtop = jev - 1;
			continue;
		}
		if (first_out < 64) return 0;
		jtop -= 64;
	}
#endif
	return 0;
}

// mem_mark_primary_se's scan (regs_core.h: mark_loop) 64 regions at a time.  Whether region i overlaps one of the primaries found
// so far does not depend on the regions between them and i; only a region that overlaps none becomes a primary itself and can then
// claim later regions of its own step -- those few are settled one new primary at a time.  The updates of a primary (its
// sub-optimal score = the first region that overlaps it, the count of near-optimal ones) are taken in the order of the list.
__device__ void fin_wave_mark(const ctx_t &x, const int n, rec_t *a, int32_t *z)
{
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	if (n == 0) return;
	const int lane = ch_lane();
	const int tmp = mark_tmp(x);
	int nz = 1;
	if (lane == 0) z[0] = 0;
	ch_wave_fence<false>();
	for (int b = 1; b < n; b += 64) {
		const int i = b + lane;
		const bool have = i < n;
		rec_t me; if (have) me = a[i];
		int hit = -1;                                            // index in z of the first primary that overlaps me
		for (int k = 0; k < nz; ++k) { const int j = z[k]; if (have && hit < 0 && mark_overlap(x, a[j], me)) hit = k; }
		unsigned long long open = __ballot(have && hit < 0);      // regions that overlap no primary yet, in list order
		while (open) {
			const int l0 = (int)__builtin_ctzll(open);            // the first of them is a primary
			if (lane == l0) { z[nz] = i; hit = -2; }
			ch_wave_fence<false>();
			const int j = z[nz];
			if (have && hit == -1 && lane > l0 && mark_overlap(x, a[j], me)) hit = nz;
			++nz;
			open = __ballot(have && hit == -1);
		}
		// effects on the primaries, in list order: per primary hit in this step
		unsigned long long todo = __ballot(have && hit >= 0);
		while (todo) {
			const int l1 = (int)__builtin_ctzll(todo);
			const int k = __builtin_amdgcn_readlane(hit, l1);
			const int j = z[k];
			const unsigned long long mine = __ballot(have && hit == k);
			const int sj = a[j].v[1];
			const unsigned long long nzs = __ballot(have && hit == k && me.v[1] != 0);
			const int aj = r_alt(a[j]);
			const unsigned long long near = __ballot(have && hit == k && sj - me.v[1] <= tmp && (aj || !r_alt(me)));
			if (lane == l1) {
				if (a[j].v[10] == 0 && nzs) a[j].v[10] = __builtin_amdgcn_readlane(me.v[1], (int)__builtin_ctzll(nzs));
				a[j].v[11] += (int)__builtin_popcountll(near);
			}
			if (have && hit == k) a[i].v[12] = j;
			todo &= ~mine;
		}
		ch_wave_fence<false>();
	}
#endif
}

// the output records (regs_core.h: emit_all) 64 at a time: every record by itself, then the two things that depend on the records
// before it -- a primary record after the first reported one is supplementary and its MAPQ is capped by that first one's
__device__ int fin_wave_emit(const ctx_t &x, uint32_t read, float frac_rep, const int n, rec_t *a, int32_t *z)
{
	int err_any = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	const int lane = ch_lane();
	int first = -1, mapq0 = 0;                                   // first reported record and its MAPQ (wave-uniform)
	const bool altm = x.ctg_alt != nullptr;
	if (altm) {                                                  // (regs_core.h: emit_all) the ALT score of every record while every record still says whether it is one
		for (int k = lane; k < n; k += 64) z[k] = alt_score_of(a, k);
		ch_wave_fence<false>();
	}
	for (int b = 0; b < n; b += 64) {
		const int k = b + lane;
		const bool have = k < n;
		int mapq = 0, flag = 0, rep = 0, err = 0;
		const int alt = have ? r_alt(a[k]) : 0;
		if (have) emit_one(x, frac_rep, a, k, &mapq, &flag, &rep, &err);
		const unsigned long long em = __ballot(err != 0);
		if (em) { err_any = __builtin_amdgcn_readlane(err, (int)__builtin_ctzll(em)); break; }
		const unsigned long long rm = __ballot(have && rep);
		if (first < 0 && rm) { const int l0 = (int)__builtin_ctzll(rm); first = b + l0; mapq0 = __builtin_amdgcn_readlane(mapq, l0); }
		if (have && rep && first >= 0 && k > first) {
			if (a[k].v[12] < 0) flag |= x.po.no_multi ? 0x10000 : 0x800;
			if (!alt && mapq > mapq0) mapq = mapq0;
		}
		ch_wave_fence<false>();                                  // (every lane has read what it needs of the other records: their [1] and [12] are not written here)
		if (have) {
			if (altm) { if (!x.alt_keep_sub_n) a[k].v[11] = a[k].v[0]; else mapq |= (a[k].v[0] + 1) << 8; rep |= alt << 1 | (z[k] > 0 ? z[k] << 2 : 0); }
			a[k].v[0] = (int32_t)read; a[k].v[13] = mapq; a[k].v[14] = flag; a[k].v[15] = rep;
		}
	}
	ch_wave_fence<false>();
#endif
	return err_any ? -err_any : n;
}

// the tail of one read by a wave; the result is in P.a[0..n) (P.a / P.b may have changed places)
template <bool STAGED> __device__ int fin_wave_read(const ctx_t &x, const uint8_t *query, uint32_t read, int64_t id, float frac_rep, int n_in, wptr_t &P, uint64_t *stage,
                                                    rec_t *dedup_out = nullptr)
{
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	const int lane = ch_lane();
	FIN_STAMP_BEGIN();
	for (int i = lane; i < n_in; i += 64) init_one(x, P.a[i]);
	ch_wave_fence<false>();
	FIN_STAMP(0);
	int n = n_in;
	if (n > 1) {
		if (!fin_wave_sort<KEY_RE, STAGED>(P, n, stage)) return -E_DPCAP;
		FIN_STAMP(1);
		// the scan of mem_sort_dedup_patch: only a region with a close predecessor does anything (dedup_near: evaluated for 64 regions
		// at a time, on fields the scan does not change ahead of itself); those take their turn one by one, in list order
		for (int b = 1; b < n; b += 64) {
			const int i = b + lane;
			unsigned long long near = __ballot(i < n && dedup_near(x, P.a, i));
			while (near) {
				const int l0 = (int)__builtin_ctzll(near);
				near &= near - 1;
				const int err = fin_wave_dedup_one(x, query, b + l0, P.a);
				if (err) return -err;
			}
		}
		FIN_STAMP(2);
		{ rec_t *a = P.a; n = fin_wave_compact(P, n, [a](int i) { return a[i].v[3] > a[i].v[2]; }); }
		FIN_STAMP(3);
		if (!fin_wave_sort<KEY_SCORE_RB_QB, STAGED>(P, n, stage)) return -E_DPCAP;
		FIN_STAMP(4);
		for (int i = 1 + lane; i < n; i += 64) if (dedup_same(P.a, i)) P.a[i].v[3] = P.a[i].v[2];      // (reads [1], rb, [2]; writes [3])
		ch_wave_fence<false>();
		{ rec_t *a = P.a; n = fin_wave_compact(P, n, [a](int i) { return i == 0 || a[i].v[3] > a[i].v[2]; }); }
		FIN_STAMP(5);
	}
	if (x.dedup_only) {
		for (int i = lane; i < n; i += 64) P.a[i].v[0] = (int32_t)read;
		ch_wave_fence<false>();
		return n;
	}
	if (dedup_out) for (int i = lane; i < n; i += 64) { rec_t t = P.a[i]; t.v[0] = (int32_t)read; t.v[13] = r_seq(t); dedup_out[i] = t; }     // (the sequence without the ALT bit)
	for (int i = lane; i < n; i += 64) mark_init_one(P.a[i], id, i);
	ch_wave_fence<false>();
	if (!fin_wave_sort<KEY_SCORE_HASH, STAGED>(P, n, stage)) return -E_DPCAP;
	FIN_STAMP(6);
	fin_wave_mark(x, n, P.a, P.z);
	if (x.ctg_alt) {                                                // the second round (regs_core.h: mark_second_round), 64 regions at a time
		int n_pri = 0;
		for (int b = 0; b < n; b += 64) n_pri += (int)__builtin_popcountll(__ballot(b + lane < n && !r_alt(P.a[b + lane])));
		if (n_pri == n) { for (int i = lane; i < n; i += 64) P.a[i].v[0] = P.a[i].v[12]; ch_wave_fence<false>(); }
		else {
			for (int i = lane; i < n; i += 64) P.a[i].v[0] = i;
			ch_wave_fence<false>();
			if (n_pri > 0 && !fin_wave_sort<KEY_ALT_SCORE_HASH, STAGED>(P, n, stage)) return -E_DPCAP;
			for (int i = lane; i < n; i += 64) P.z[P.a[i].v[0]] = i;
			ch_wave_fence<false>();
			for (int i = lane; i < n; i += 64) {
				rec_t &p = P.a[i];
				if (p.v[12] >= 0) { p.v[0] = P.z[p.v[12]]; if (r_alt(p)) p.v[12] = 0x7FFFFFFF; }
				else p.v[0] = -1;
			}
			ch_wave_fence<false>();
			if (n_pri > 0) {
				for (int i = lane; i < n_pri; i += 64) { P.a[i].v[10] = 0; P.a[i].v[12] = -1; }
				ch_wave_fence<false>();
				fin_wave_mark(x, n_pri, P.a, P.z);
			}
		}
	}
	FIN_STAMP(7);
	const int rc = fin_wave_emit(x, read, frac_rep, n, P.a, P.z);
	FIN_STAMP(8);
	return rc;
#else
	return 0;
#endif
}

// NMAX records per read in LDS (0: the read's records stay in HBM, for the rare read beyond FIN_NMAX regions); CLS: its list
template <int NMAX, int CLS>
__global__ void __launch_bounds__(64) fin_wave_kernel(fin_args_t A)
{
#if defined(__HIP_DEVICE_COMPILE__)
	using namespace chain_core;
	constexpr int NL = NMAX ? NMAX : 1;
	__shared__ __attribute__((aligned(16))) rec_t la[NL], lb[NL];
	__shared__ uint64_t lkeys[NL], lk128[2 * NL];
	__shared__ uint32_t ltmp[NL], lorder[NL];
	__shared__ int32_t lz[NL];
	__shared__ uint64_t lstage[NMAX ? 1 : 2 * FIN_STAGE];         // keys of a read whose records are in HBM, on their way through the rank loop
	__shared__ int64_t lctg[FIN_CTG_LDS];                          // a small contig table is read from LDS (the first step of every read looks its regions up)
	const int lane = ch_lane();
#ifdef FIN_PROFILE
	if (lane < 16) g_lprof[lane] = 0;
	ch_wave_fence<false>();
	const long long t_block0 = (long long)wall_clock64();
#endif
	ctx_t x = A.x;
	if (x.n_contigs > 1 && x.n_contigs <= FIN_CTG_LDS) {
		for (int k = lane; k < x.n_contigs; k += 64) lctg[k] = A.x.ctg_off[k];
		ch_wave_fence<false>();
		x.ctg_off = lctg;
	}
	x.dp_h = A.g_dp + ((size_t)CLS * FIN_WAVE_GRID + blockIdx.x) * 2 * FIN_DPCAP; x.dp_e = x.dp_h + FIN_DPCAP; x.dp_cap = FIN_DPCAP;
	const uint32_t n_def = A.ctr[CLS];
	const uint32_t *list = A.defer + (size_t)CLS * A.n_reads;
	for (;;) {
		uint32_t t = 0;
		if (lane == 0) t = atomicAdd(&A.ctr[8 + CLS], 1u);
		t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
		if (t >= n_def) break;
		const uint32_t r = list[t];
		const int n_in = (int)A.rpr[r];
		const uint32_t off = A.in_off[r];
		wptr_t P;
		if (NMAX) { P.a = la; P.b = lb; P.keys = lkeys; P.k128 = lk128; P.tmp = ltmp; P.order = lorder; P.z = lz; }
		else {
			P.a = (rec_t *)(A.work + 16 * (size_t)off); P.b = (rec_t *)(A.work2 + 16 * (size_t)off);
			P.keys = A.g_keys + off; P.k128 = A.g_k128 + 2 * (size_t)off; P.tmp = A.g_tmp + off; P.order = A.g_order + off; P.z = A.g_z + off;
		}
		for (int i = lane; i < n_in; i += 64) {
			const int4 *src = (const int4 *)(A.regs_in + 8 * (size_t)(off + i));
			const int4 u = src[0], v = src[1];
			rec_t &d = P.a[i];
			d.v[0] = u.x; d.v[1] = u.y; d.v[2] = u.z; d.v[3] = u.w; d.v[4] = v.x; d.v[5] = v.y; d.v[6] = v.z; d.v[7] = v.w;
		}
		ch_wave_fence<false>();
		const int n = fin_wave_read<NMAX == 0>(x, A.reads + A.read_offs[r], r, x.po.id0 + r, A.frac_rep ? A.frac_rep[r] : 0.f, n_in, P, lstage,
		                                       A.dedup_out ? (rec_t *)(A.dedup_out + 16 * (size_t)off) : nullptr);
		if (n < 0) { if (lane == 0) { A.ctr[16] = (uint32_t)-n; A.opr[r] = 0; } continue; }
		rec_t *dst = (rec_t *)(A.work + 16 * (size_t)off);
		if (P.a != dst) for (int i = lane; i < n; i += 64) dst[i] = P.a[i];
		if (lane == 0) A.opr[r] = (uint32_t)n;
		ch_wave_fence<false>();
	}
#ifdef FIN_PROFILE
	ch_wave_fence<false>();
	if (lane == 0) g_lprof[15] = (unsigned long long)((long long)wall_clock64() - t_block0);
	ch_wave_fence<false>();
	if (lane < 16) atomicAdd(&g_fin_prof[CLS][lane], g_lprof[lane]);
	if (lane == 0) atomicMax(&g_fin_prof[CLS][14], g_lprof[15]);
#endif
#endif
}

// record k of read r: work[in_off[r] + k] -> out[out_off[r] + k]; one thread per INPUT slot (the slots beyond a read's count are skipped)
__global__ void __launch_bounds__(256) fin_compact_kernel(const int32_t *__restrict__ work, const uint32_t *__restrict__ in_off, const uint32_t *__restrict__ out_off,
                                                          const uint32_t *__restrict__ opr, uint32_t n_reads, int32_t *__restrict__ out)
{
	// (sixteen lanes a read, a quarter record each: a lane a read wrote 64 bytes at a time to 64 different places)
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 4, l = t & 15u;
	if (r >= n_reads) return;
	const uint32_t n = opr[r], si = in_off[r], so = out_off[r];
	const int4 *s = (const int4 *)(work + 16 * (size_t)si);
	int4 *d = (int4 *)(out + 16 * (size_t)so);
	for (uint32_t q = l; q < 4 * n; q += 16) d[q] = s[q];
}

// ---- host side: scratch per (device, stream), grown on demand
struct fin_scratch_t {
	uint32_t *in_off, *out_off, *opr_tmp, *defer, *ctr; size_t cap_reads; int32_t *g_dp;
	hipStream_t side[3]; hipEvent_t fork, join[3];
	int32_t *work, *work2; uint64_t *g_keys, *g_k128; uint32_t *g_tmp, *g_order; int32_t *g_z; size_t cap_regs;
	int32_t *dedup; size_t cap_dedup;       // the regions between the two halves of the tail, for callers that ask (bmh_finalize_regs_device_ex)
	uint8_t *ctg_alt; int cap_alt;          // the ALT table on the device
	void *scan_tmp; size_t scan_bytes;
	double *logtab; int64_t *ctg; int cap_ctg;
	uint32_t *h_pin;
	hipEvent_t ev0, ev1;
	bool inited;             // the one-time part below went through completely
};
static std::mutex g_fin_mu;
static std::map<std::pair<int, void *>, fin_scratch_t *> g_fin_map;
static thread_local fin_scratch_t *g_fin_last = nullptr;

template <class T> static int fin_grow(T *&p, size_t n) { if (p) (void)hipFree(p); p = nullptr; return hipMalloc((void **)&p, sizeof(T) * n) == hipSuccess ? BMH_OK : BMH_ENOMEM; }

// the (device, stream) scratch of bmh_finalize_regs_device (device buffers, three side streams, events, pinned words): freed when
// the caller retires the stream (stream idle, its device current)
extern "C" void bmh_finalize_release(void *stream_)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return;
	fin_scratch_t *S = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_fin_mu);
		auto it = g_fin_map.find(std::make_pair(dev, stream_));
		if (it == g_fin_map.end()) return;
		S = it->second;
		g_fin_map.erase(it);
	}
	if (g_fin_last == S) g_fin_last = nullptr;
	void *ps[] = {S->in_off, S->out_off, S->opr_tmp, S->defer, S->ctr, S->g_dp, S->work, S->work2, S->g_keys, S->g_k128, S->g_tmp, S->g_order, S->g_z, S->scan_tmp, S->logtab, S->ctg, S->dedup, S->ctg_alt};
	for (void *q : ps) if (q) (void)hipFree(q);
	if (S->h_pin) (void)hipHostFree(S->h_pin);
	if (S->ev0) (void)hipEventDestroy(S->ev0);
	if (S->ev1) (void)hipEventDestroy(S->ev1);
	if (S->fork) (void)hipEventDestroy(S->fork);
	for (int i = 0; i < 3; ++i) { if (S->side[i]) (void)hipStreamDestroy(S->side[i]); if (S->join[i]) (void)hipEventDestroy(S->join[i]); }
	free(S);
}

extern "C" float bmh_finalize_regs_device_last_ms(void)
{
	if (!g_fin_last) return -1.f;
	float ms = -1.f;
	if (hipEventSynchronize(g_fin_last->ev1) != hipSuccess) return -1.f;
	if (hipEventElapsedTime(&ms, g_fin_last->ev0, g_fin_last->ev1) != hipSuccess) return -1.f;
	return ms;
}

static int64_t finalize_regs_device_impl(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                                         const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                                         const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read, const float *d_frac_rep,
                                         int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream_, int dedup_only,
                                         bmh_fin_extra_t *extra = nullptr);

// bmh_finalize_regs_device that also leaves (csrc/align_pipeline.hip: interleaved pairs) the regions as mem_sort_dedup_patch left them, the first record of
// every read and the device's copies of the logarithm table and the contig offsets
int64_t bmh_finalize_regs_device_ex(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                                    const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                                    const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read, const float *d_frac_rep,
                                    int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream_, bmh_fin_extra_t *extra)
{
	return finalize_regs_device_impl(idx, copt, ep, popt, d_reads, d_offs, n_reads, d_regs, n_regs, d_regs_per_read, d_frac_rep, n_contigs, contig_offset, d_out, d_out_per_read, stream_, 0, extra);
}

extern "C" int64_t bmh_finalize_regs_device(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                                            const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                                            const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read, const float *d_frac_rep,
                                            int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream_)
{
	return finalize_regs_device_impl(idx, copt, ep, popt, d_reads, d_offs, n_reads, d_regs, n_regs, d_regs_per_read, d_frac_rep, n_contigs, contig_offset, d_out, d_out_per_read, stream_, 0);
}

// mem_sort_dedup_patch alone (the first step of the region tail, the step interleaved pairs share with single-end reads): see the header
extern "C" int64_t bmh_dedup_regs_device(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                                         const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                                         const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read,
                                         int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream_)
{
	return finalize_regs_device_impl(idx, copt, ep, popt, d_reads, d_offs, n_reads, d_regs, n_regs, d_regs_per_read, nullptr, n_contigs, contig_offset, d_out, d_out_per_read, stream_, 1);
}

static int64_t finalize_regs_device_impl(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                                         const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                                         const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read, const float *d_frac_rep,
                                         int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream_, int dedup_only,
                                         bmh_fin_extra_t *extra)
{
	if (!idx || !copt || !ep || !popt || (n_reads && (!d_reads || !d_offs || !d_regs_per_read || (!d_frac_rep && !dedup_only) || !d_out_per_read)) || (n_regs && (!d_regs || !d_out))) {
		bmh_set_error("bmh_finalize_regs_device: null argument"); return BMH_EINVAL;
	}
	if (!idx->dev.pac) { bmh_set_error("bmh_finalize_regs_device: the index has no 2-bit reference"); return BMH_EINVAL; }
	if (!dedup_only && !(popt->mapQ_coef_len > 0)) { bmh_set_error("bmh_finalize_regs_device: mapQ_coef_len <= 0 (the seed-coverage form of MAPQ) is not restated"); return BMH_EINVAL; }
	if (n_contigs > 1 && !contig_offset) { bmh_set_error("bmh_finalize_regs_device: null contig table"); return BMH_EINVAL; }
	if (n_regs >> 31) { bmh_set_error("bmh_finalize_regs_device: 2^31 regions or more in one batch"); return BMH_ECAPACITY; }
	if (n_reads == 0) return 0;
	const bool alt_mode = popt->contig_is_alt && !dedup_only;      // ALT contigs: the table goes to the device (below), the marking gets its second round
	if (getenv("BMH_FIN_FORCE_ECAPACITY")) { bmh_set_error("bmh_finalize_regs_device: capacity error forced by BMH_FIN_FORCE_ECAPACITY (test hook of the callers' host fallback)"); return BMH_ECAPACITY; }
	hipStream_t st = (hipStream_t)stream_;
	int dev = 0;
	HIPCK(hipGetDevice(&dev));
	fin_scratch_t *S;
	{
		std::lock_guard<std::mutex> lk(g_fin_mu);
		auto key = std::make_pair(dev, stream_);
		auto it = g_fin_map.find(key);
		if (it == g_fin_map.end()) { S = (fin_scratch_t *)calloc(1, sizeof(fin_scratch_t)); g_fin_map[key] = S; }
		else S = it->second;
	}
	g_fin_last = S;
	if (!S->inited) {
		// one-time part of a (device, stream) pair; a step that fails leaves `inited` false and the next call starts over (what was
		// created so far is kept and not created twice)
		if (!S->logtab) {
			// log(k) by the host's libm (the reference's MAPQ is computed there): the device reads these values, it never calls log()
			double *h = (double *)malloc(sizeof(double) * FIN_NLOG);
			if (!h) { bmh_set_error("bmh_finalize_regs_device: out of host memory"); return BMH_ENOMEM; }
			for (int k = 0; k < FIN_NLOG; ++k) h[k] = log((double)k);
			double *d_log = nullptr;
			const bool ok = fin_grow(d_log, FIN_NLOG) == BMH_OK && hipMemcpy(d_log, h, sizeof(double) * FIN_NLOG, hipMemcpyHostToDevice) == hipSuccess;
			free(h);
			if (!ok) { if (d_log) (void)hipFree(d_log); bmh_set_error("bmh_finalize_regs_device: log table: %s", hipGetErrorString(hipGetLastError())); return BMH_ENOMEM; }
			S->logtab = d_log;
		}
		if (!S->ev0) HIPCK(hipEventCreate(&S->ev0));
		if (!S->ev1) HIPCK(hipEventCreate(&S->ev1));
		if (!S->h_pin) HIPCK(hipHostMalloc((void **)&S->h_pin, 128));
		if (!S->ctr && fin_grow(S->ctr, 32) != BMH_OK) return BMH_ENOMEM;
		if (!S->g_dp && fin_grow(S->g_dp, (size_t)FIN_NCLS * FIN_WAVE_GRID * 2 * FIN_DPCAP) != BMH_OK) return BMH_ENOMEM;
		if (!S->fork) HIPCK(hipEventCreateWithFlags(&S->fork, hipEventDisableTiming));
		for (int i = 0; i < 3; ++i) {
			if (!S->side[i]) HIPCK(hipStreamCreateWithFlags(&S->side[i], hipStreamNonBlocking));
			if (!S->join[i]) HIPCK(hipEventCreateWithFlags(&S->join[i], hipEventDisableTiming));
		}
		S->inited = true;
	}
	if ((size_t)n_reads + 1 > S->cap_reads) {
		const size_t c = (size_t)n_reads + n_reads / 4 + 1024;
		if (fin_grow(S->in_off, c) != BMH_OK || fin_grow(S->out_off, c) != BMH_OK || fin_grow(S->defer, FIN_NCLS * c) != BMH_OK) return BMH_ENOMEM;
		size_t t1 = 0;
		(void)rocprim::exclusive_scan(nullptr, t1, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, c, rocprim::plus<uint32_t>(), 0);
		if (S->scan_tmp) (void)hipFree(S->scan_tmp);
		S->scan_tmp = nullptr;
		HIPCK(hipMalloc(&S->scan_tmp, t1 + 256)); S->scan_bytes = t1 + 256;
		S->cap_reads = c;
	}
	if (n_regs + 1 > S->cap_regs) {
		const size_t c = n_regs + n_regs / 4 + 1024;
		if (fin_grow(S->work, 16 * c) != BMH_OK || fin_grow(S->work2, 16 * c) != BMH_OK || fin_grow(S->g_keys, c) != BMH_OK || fin_grow(S->g_k128, 2 * c) != BMH_OK ||
		    fin_grow(S->g_tmp, c) != BMH_OK || fin_grow(S->g_order, c) != BMH_OK || fin_grow(S->g_z, c) != BMH_OK) return BMH_ENOMEM;
		S->cap_regs = c;
	}
	if (n_contigs > 1) {
		if (n_contigs > S->cap_ctg) { if (fin_grow(S->ctg, (size_t)n_contigs) != BMH_OK) return BMH_ENOMEM; S->cap_ctg = n_contigs; }
		HIPCK(hipMemcpyAsync(S->ctg, contig_offset, sizeof(int64_t) * n_contigs, hipMemcpyHostToDevice, st));
	}
	fin_args_t A;
	memset(&A, 0, sizeof(A));
	A.x.co = *copt; A.x.ep = *ep; A.x.po = *popt; A.x.l_pac = (int64_t)idx->dev.l_pac; A.x.pac = idx->dev.pac;
	A.x.n_contigs = n_contigs > 1 ? n_contigs : 1; A.x.ctg_off = n_contigs > 1 ? S->ctg : nullptr;
	A.x.logtab = S->logtab; A.x.n_log = FIN_NLOG; A.x.dp_h = A.x.dp_e = nullptr; A.x.dp_cap = 0; A.x.dedup_only = dedup_only;
	A.x.po.contig_is_alt = nullptr;                               // (a host pointer: never followed on the device)
	A.x.ctg_alt = nullptr; A.x.alt_keep_sub_n = extra ? extra->alt_keep_sub_n : 0;
	if (alt_mode) {
		const int nc = n_contigs > 1 ? n_contigs : 1;
		if (nc > S->cap_alt) { if (fin_grow(S->ctg_alt, (size_t)nc) != BMH_OK) return BMH_ENOMEM; S->cap_alt = nc; }
		HIPCK(hipMemcpyAsync(S->ctg_alt, popt->contig_is_alt, (size_t)nc, hipMemcpyHostToDevice, st));
		A.x.ctg_alt = S->ctg_alt;
	}
	A.reads = d_reads; A.read_offs = d_offs; A.regs_in = d_regs; A.rpr = d_regs_per_read; A.in_off = S->in_off; A.frac_rep = d_frac_rep;
	A.work = S->work; A.work2 = S->work2; A.g_keys = S->g_keys; A.g_k128 = S->g_k128; A.g_tmp = S->g_tmp; A.g_order = S->g_order; A.g_z = S->g_z;
	A.opr = d_out_per_read; A.n_reads = n_reads; A.defer = S->defer; A.ctr = S->ctr; A.g_dp = S->g_dp;
	A.dedup_out = nullptr;
	if (extra && extra->d_dedup_out) {
		if (n_regs + 1 > S->cap_dedup) { const size_t c = n_regs + n_regs / 4 + 1024; if (fin_grow(S->dedup, 16 * c) != BMH_OK) return BMH_ENOMEM; S->cap_dedup = c; }
		A.dedup_out = S->dedup;
	}
	HIPCK(hipEventRecord(S->ev0, st));
	HIPCK(hipMemsetAsync(S->ctr, 0, 128, st));
	size_t tb = S->scan_bytes;
	HIPCK(rocprim::exclusive_scan(S->scan_tmp, tb, d_regs_per_read, S->in_off, 0u, (size_t)n_reads, rocprim::plus<uint32_t>(), st));
	static const bool want_phases = getenv("BMH_FIN_PHASES") != nullptr;      // debug: time of the kernels of every call
	static thread_local hipEvent_t ph[4] = {nullptr, nullptr, nullptr, nullptr};
	if (want_phases && !ph[0]) for (hipEvent_t &e : ph) HIPCK(hipEventCreate(&e));
	if (want_phases) HIPCK(hipEventRecord(ph[0], st));
	fin_lane_kernel<<<(n_reads + 255) / 256, 256, 0, st>>>(A);
	if (want_phases) HIPCK(hipEventRecord(ph[1], st));
	// the wave classes side by side (their lists are short and uneven: the class of the largest reads is a handful of long jobs)
	static thread_local hipEvent_t pc[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	if (want_phases) {                       // (debug: one class after the other, each timed)
		if (!pc[0]) for (hipEvent_t &e : pc) HIPCK(hipEventCreate(&e));
		HIPCK(hipEventRecord(pc[0], st));
		fin_wave_kernel<32, 0><<<FIN_WAVE_GRID, 64, 0, st>>>(A); HIPCK(hipEventRecord(pc[1], st));
		fin_wave_kernel<128, 1><<<FIN_WAVE_GRID, 64, 0, st>>>(A); HIPCK(hipEventRecord(pc[2], st));
		fin_wave_kernel<256, 2><<<FIN_WAVE_GRID, 64, 0, st>>>(A); HIPCK(hipEventRecord(pc[3], st));
		fin_wave_kernel<FIN_NMAX, 3><<<FIN_WAVE_GRID, 64, 0, st>>>(A); HIPCK(hipEventRecord(pc[4], st));
		fin_wave_kernel<0, 4><<<FIN_WAVE_GRID, 64, 0, st>>>(A); HIPCK(hipEventRecord(pc[5], st));
	} else {
		HIPCK(hipEventRecord(S->fork, st));
		for (int i = 0; i < 3; ++i) HIPCK(hipStreamWaitEvent(S->side[i], S->fork, 0));
		fin_wave_kernel<FIN_NMAX, 3><<<FIN_WAVE_GRID, 64, 0, S->side[0]>>>(A);
		fin_wave_kernel<0, 4><<<FIN_WAVE_GRID, 64, 0, S->side[1]>>>(A);
		fin_wave_kernel<256, 2><<<FIN_WAVE_GRID, 64, 0, S->side[2]>>>(A);
		fin_wave_kernel<128, 1><<<FIN_WAVE_GRID, 64, 0, S->side[0]>>>(A);
		fin_wave_kernel<32, 0><<<FIN_WAVE_GRID, 64, 0, st>>>(A);
		for (int i = 0; i < 3; ++i) { HIPCK(hipEventRecord(S->join[i], S->side[i])); HIPCK(hipStreamWaitEvent(st, S->join[i], 0)); }
	}
	if (want_phases) HIPCK(hipEventRecord(ph[2], st));
	tb = S->scan_bytes;
	HIPCK(rocprim::exclusive_scan(S->scan_tmp, tb, d_out_per_read, S->out_off, 0u, (size_t)n_reads, rocprim::plus<uint32_t>(), st));
	fin_compact_kernel<<<(unsigned)(((size_t)n_reads * 16 + 255) / 256), 256, 0, st>>>(S->work, S->in_off, S->out_off, d_out_per_read, n_reads, d_out);
	if (extra) {
		if (extra->d_dedup_out) fin_compact_kernel<<<(unsigned)(((size_t)n_reads * 16 + 255) / 256), 256, 0, st>>>(A.dedup_out, S->in_off, S->out_off, d_out_per_read, n_reads, extra->d_dedup_out);
		if (extra->d_out_off) HIPCK(hipMemcpyAsync(extra->d_out_off, S->out_off, 4 * (size_t)n_reads, hipMemcpyDeviceToDevice, st));
		extra->d_logtab = S->logtab; extra->n_log = FIN_NLOG; extra->d_ctg_off = n_contigs > 1 ? S->ctg : nullptr;
	}
	HIPCK(hipMemcpyAsync(S->h_pin, S->out_off + (n_reads - 1), 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(S->h_pin + 1, d_out_per_read + (n_reads - 1), 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(S->h_pin + 2, S->ctr, 68, hipMemcpyDeviceToHost, st));
	HIPCK(hipEventRecord(S->ev1, st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	if (want_phases) {
		float a = 0, b = 0, c = 0;
		(void)hipEventElapsedTime(&a, ph[0], ph[1]); (void)hipEventElapsedTime(&b, ph[1], ph[2]); (void)hipEventElapsedTime(&c, ph[2], S->ev1);
		float q[5] = {0, 0, 0, 0, 0};
		for (int i = 0; i < 5; ++i) (void)hipEventElapsedTime(&q[i], pc[i], pc[i + 1]);
#ifdef FIN_PROFILE
		{
			unsigned long long hh[5][16];
			(void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_fin_prof), sizeof(hh));
			unsigned long long z16[5][16] = {{0}};
			(void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_prof), z16, sizeof(z16));
			for (int c = 0; c < 5; ++c) {
				const unsigned long long *h = hh[c];
				fprintf(stderr, "[finalize] class %d wave-ms: blocks alive %.1f (longest %.3f) | init %.1f | sort1 %.1f dedup %.1f compact %.1f | sort2 %.1f equal+compact %.1f | sort3 %.1f mark %.1f emit %.1f || in the sorts: keys+ranks %.1f introsort %.1f place %.1f permute %.1f\n", c,
				        h[15] / 1e5, h[14] / 1e5, h[0] / 1e5, h[1] / 1e5, h[2] / 1e5, h[3] / 1e5, h[4] / 1e5, h[5] / 1e5, h[6] / 1e5, h[7] / 1e5, h[8] / 1e5, h[10] / 1e5, h[11] / 1e5, h[12] / 1e5, h[13] / 1e5);
			}
		}
#endif
		fprintf(stderr, "[finalize] wave classes one after the other: %.3f / %.3f / %.3f / %.3f / %.3f ms\n", q[0], q[1], q[2], q[3], q[4]);
		fprintf(stderr, "[finalize] %u reads, %llu regions: lane kernel %.3f ms, wave kernels %.3f ms (%u / %u / %u / %u / %u reads with up to 32 / 128 / 256 / 512 / more regions), scan + compaction %.3f ms\n", n_reads, (unsigned long long)n_regs, a, b, S->h_pin[2], S->h_pin[3], S->h_pin[4], S->h_pin[5], S->h_pin[6], c);
	}
	if (S->h_pin[18] != 0) {
		const uint32_t e = S->h_pin[18];
		bmh_set_error("bmh_finalize_regs_device: %s", e == E_LOG ? "a region longer than the logarithm table (65535)" : e == E_DPCAP ? "a patch alignment beyond the kernel's capacity (query side of 1022 bases) or a sort beyond its stack" : "internal error");
		return BMH_ECAPACITY;
	}
	return (int64_t)S->h_pin[0] + (int64_t)S->h_pin[1];
}
