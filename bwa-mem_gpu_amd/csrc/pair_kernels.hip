// Mate rescue's local alignments on the device (BASELINE configs[3]): the ksw_align2 calls of mem_matesw
// (/root/reference/src/bwamem_pair.c:119-188; ksw_align2 with KSW_XSUBO | KSW_XSTART [| KSW_XBYTE], src/ksw.c:389-740) as one batch of
// jobs.  pair_post.cpp walks the pairs twice -- once to collect the alignments mem_matesw is going to ask for, once to take their
// results -- and this kernel computes them in between; on the host they were 98 % of bmh_finalize_pairs on an hg38-like batch
// (107 000 alignments of 150 x 480 cells per million reads: 2.7 s on 16 threads).
//
// The reference computes a striped SSE2 kernel (Farrar) whose results depend on the striping in two places (local_sw.cpp): E(i+1,j) is
// taken from H(i,j) BEFORE the lazy-F correction, and zero-score padding lanes take part in the row maximum.  To return the same
// numbers a job runs on 16 GPU lanes that ARE the 16 byte lanes of the SSE register (8 of them the 16-bit lanes of ksw_i16): lane l
// owns query positions l * slen .. l * slen + slen - 1, walks its slen segments in the reference's order, takes H of the previous
// segment row from lane l - 1 (the byte shift of the SSE code) and runs the lazy-F loop with the reference's exit test.  The lane's
// columns of H / E live in LDS (lane-private: no synchronisation), four jobs per wave.
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include "bmh_internal.h"
#include "local_sw.h"
#include "pair_kernels.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

#define MSW_SLEN 40            // segments per lane the kernel keeps in LDS: queries up to 16 * 40 columns in byte mode (which ends at 249), 8 * 40 = 320 in 16-bit mode
#define MSW_JOBS_PER_BLOCK 8   // 128 threads
#define MSW_TBUF 64            // target rows staged per refill
#define MSW_ROW (16 * MSW_JOBS_PER_BLOCK)   // elements of one LDS row: the block's jobs, sixteen lanes each

struct msw_args_t {
	const bmh_msw_job_t *jobs; uint32_t n_jobs;
	const uint8_t *reads; const uint32_t *read_offs;      // ASCII reads of the batch
	const uint8_t *pac; long long l_pac;
	int a, b, o_del, e_del, o_ins, e_ins;
	uint32_t *blist;                                      // per job: room for tlen entries (val << 16 | row) of the second-best bookkeeping
	int32_t *out;                                         // [n_jobs][7] = {score, te, qe, score2, te2, tb, qb}
};

__device__ __forceinline__ int msw_text(const msw_args_t &A, long long p)
{
	const bool rev = p >= A.l_pac;
	const long long f = rev ? (A.l_pac << 1) - 1 - p : p;
	const int c = (A.pac[f >> 2] >> ((~f & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}
__device__ __forceinline__ int msw_nt4(uint8_t c) { c &= 0xDF; return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
__device__ __forceinline__ int msw_sat0(int v) { return v < 0 ? 0 : v; }
// value of lane l - 1 of the 16-lane group (lane 0 of the group: 0)
__device__ __forceinline__ int msw_shl(int v, int l) { const int t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); return l == 0 ? 0 : t; }
__device__ __forceinline__ int msw_gmax(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
	return v;
}

struct msw_res_t { int score, te, qe, score2, te2; };

// One pass of the striped kernel for the job of this 16-lane group (sw_pass of local_sw.cpp).  All lanes of the wave call it together;
// `on`: this group has a job.  query position k of the pass = qbase[k * qstep] through qmap (ASCII read, optionally complemented);
// target row i = trow(i).  xtra as in ksw_align2.
// BYTE_ALL: every job of the batch runs in byte mode (mates of up to 249 bases at a = 1: the usual case) -- the word-mode arithmetic and the per-lane choice between
// the two are compiled out of the loops.
template <bool BYTE_ALL, class QF, class TF>
__device__ void msw_pass(const msw_args_t &A, const bool on, const int lanes, const int qlen, const int tlen, QF qcode, TF trow, const int xtra,
                         uint16_t *H0, uint16_t *H1, uint16_t *E, uint16_t *Hm, uint8_t *qc, uint8_t *tbuf, uint32_t *blist, msw_res_t &R)
{
	const int lane = threadIdx.x & 63, l = lane & 15;
	const unsigned long long gmask = 0xFFFFull << (lane & 48);
	const bool byte = BYTE_ALL ? true : lanes == 16;
	const bool lact = on && l < lanes;                         // this lane is one of the job's SSE lanes
	const int slen = on ? (qlen + lanes - 1) / lanes : 0;
	const int minsc = (xtra & BMH_SW_XSUBO) ? xtra & 0xffff : 0x10000, endsc = (xtra & BMH_SW_XSTOP) ? xtra & 0xffff : 0x10000;
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	int mn = min(min(A.a, -A.b), -1), mx = max(max(A.a, -A.b), -1);
	const int shift = byte ? (256 - (mn & 0xff)) & 0xff : 0;
	// lane-private columns: element j of lane l at [j * MSW_ROW + l]
	for (int j = 0; j < slen; ++j) {
		const int k = j + l * slen;
		H0[j * MSW_ROW + l] = 0; H1[j * MSW_ROW + l] = 0; E[j * MSW_ROW + l] = 0; Hm[j * MSW_ROW + l] = 0;
		qc[j * MSW_ROW + l] = (uint8_t)((lact && k < qlen) ? qcode(k) : 5);      // 5: padding, scores 0 against everything
	}
	int gmax = 0, te = -1;
	int bl_n = 0, bl_val = 0, bl_i = -2;                       // second-best bookkeeping (lane 0 of the group): entries written so far, the open one
	uint16_t *h0 = H0, *h1 = H1;
	bool alive = on && tlen > 0;
	const int slen_w = __builtin_amdgcn_readfirstlane(max(max(__shfl(slen, 0), __shfl(slen, 16)), max(__shfl(slen, 32), __shfl(slen, 48))));
	for (int i = 0; __any(alive); ++i) {
		if ((i & (MSW_TBUF - 1)) == 0) {                        // next MSW_TBUF target rows of every job, four per lane
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			for (int u = l; u < MSW_TBUF; u += 16) tbuf[u] = (uint8_t)((alive && i + u < tlen) ? trow(i + u) : 4);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		}
		const bool run = alive && i < tlen;
		const int t = tbuf[i & (MSW_TBUF - 1)];
		// the row's target base against a query code: a match scores a, an N on either side -1, padding (5) nothing -- without branches: teq never equals a
		// query code when the target base is an N, mis is what a mismatch with THIS target base costs
		const int teq = t > 3 ? 99 : t, mis = t > 3 ? -1 : -A.b;
		int hv = msw_shl(run ? (int)h0[(slen > 0 ? slen - 1 : 0) * MSW_ROW + l] : 0, l);
		int f = 0, mxv = 0;
		for (int j = 0; j < slen_w; ++j) {
			if (run && j < slen) {
				const int q = qc[j * MSW_ROW + l];
				int e = E[j * MSW_ROW + l];
				const int hnext = h0[j * MSW_ROW + l];                   // (the three loads of the segment in flight together)
				int S = q == teq ? A.a : mis;
				S = q > 3 ? (q & 1) - 1 : S;
				int h = byte ? msw_sat0(min(hv + S + shift, 255) - shift) : max(min(hv + S, 32767), -32768);
				h = max(h, e); h = max(h, f);
				mxv = max(mxv, h);
				h1[j * MSW_ROW + l] = (uint16_t)h;
				e = max(msw_sat0(e - A.e_del), msw_sat0(h - oe_del));
				E[j * MSW_ROW + l] = (uint16_t)e;
				f = max(msw_sat0(f - A.e_ins), msw_sat0(h - oe_ins));
				hv = hnext;
			}
		}
		// lazy F (ksw.c:497-511, 627-638): at most 16 rounds; a round ends the whole loop at the first segment where no lane's F can still raise H
		{
			int k = 0, j = 0;
			bool lz = run && slen > 0;
			f = msw_shl(f, l);
			while (__any(lz)) {
				bool more = false;
				if (lz) {
					const int h = max((int)h1[j * MSW_ROW + l], f);
					h1[j * MSW_ROW + l] = (uint16_t)h;
					const int hh = msw_sat0(h - oe_ins);
					f = msw_sat0(f - A.e_ins);
					more = lact && f > hh;
				}
				const bool any = (__ballot(more) & gmask) != 0;
				bool shiftf = false;
				if (lz) {
					if (!any) lz = false;
					else if (++j == slen) { j = 0; if (++k == 16) lz = false; else shiftf = true; }
				}
				const int fs = msw_shl(f, l);
				f = shiftf ? fs : f;
			}
		}
		int imax = msw_gmax(lact ? mxv : 0);
		if (run) {
			if (imax >= minsc && l == 0) {                       // ksw.c:519-527: consecutive rows keep one entry, at the row of their maximum so far
				if (bl_i + 1 != i) { if (bl_i >= 0) blist[bl_n++] = (uint32_t)bl_val << 16 | (uint32_t)bl_i; bl_val = imax; bl_i = i; }
				else if (bl_val < imax) { bl_val = imax; bl_i = i; }
			}
			if (imax > gmax) {
				gmax = imax; te = i;
				for (int j = 0; j < slen; ++j) Hm[j * MSW_ROW + l] = h1[j * MSW_ROW + l];
				if ((byte && gmax + shift >= 255) || gmax >= endsc) alive = false;
			}
			uint16_t *tsw = h0; h0 = h1; h1 = tsw;
			if (i + 1 >= tlen) alive = false;
		}
	}
	if (l == 0 && bl_i >= 0) blist[bl_n++] = (uint32_t)bl_val << 16 | (uint32_t)bl_i;
	bl_n = __shfl(bl_n, lane & 48);
	R.score = byte ? (gmax + shift < 255 ? gmax : 255) : gmax;
	R.te = te; R.qe = -1; R.score2 = -1; R.te2 = -1;
	if (on && (!byte || R.score != 255)) {
		// end of the query: the largest H of the best row, the smallest position among equals (ksw.c:540-547)
		int bv = -1, bp = 0x7FFFFFFF;
		for (int j = 0; j < slen; ++j) if (l < lanes) { const int v = Hm[j * MSW_ROW + l], pos = j + l * slen; if (v > bv || (v == bv && pos < bp)) { bv = v; bp = pos; } }
		int key = l < lanes ? (bv << 12 | (0xFFF - min(bp, 0xFFF))) : -1;        // positions < 4096 (MSW_SLEN * 16 = 640)
		key = msw_gmax(key);
		R.qe = 0xFFF - (key & 0xFFF);
		if (bl_n > 0) {
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_s_waitcnt(0);
			const int d = (R.score + mx - 1) / mx, low = te - d, high = te + d;
			// best entry outside [low, high], the first of equals: key = val << 16 | (0xFFFF - index)
			int best = -1;
			for (int u = l; u < bl_n; u += 16) {
				const uint32_t x = blist[u];
				const int e = (int)(x & 0xFFFF), v = (int)(x >> 16);
				if (e < low || e > high) best = max(best, (v << 16) | (0xFFFF - min(u, 0xFFFF)));
			}
			best = msw_gmax(best);
			if (best >= 0) { R.score2 = best >> 16; R.te2 = (int)(blist[0xFFFF - (best & 0xFFFF)] & 0xFFFF); }
		}
	}
}

// ---- byte mode with the lane's columns in REGISTERS (round 5): the same numbers without walking the striped kernel's loops.
// What the lazy-F loop leaves in H is the ordinary F recurrence over the query positions in order -- its exit test (no lane's F above
// H - oe_ins at a segment) only fires once every lane's inflow is dominated by an F chain that has been applied already, given
// o_ins > 0 -- so H after the loop is: first sweep (F starting at 0 at the lane's first column) + the F that flows in from the lanes
// to the left, a max-plus scan over the group's sixteen lanes of what each lane's sweep sends out, exactly as in the extension
// kernels.  The two places where the striping shows stay as they are: E(i+1, j) and the row maximum are taken from the FIRST
// sweep's H (before the inflow), and the padding columns (score 0) are columns like any other.  The 255 cap of the byte
// arithmetic never acts before the row whose maximum ends the pass (gmax + shift >= 255), and that row's maximum is at least as
// large uncapped, so plain integers return the same score (255), row and nothing else.  Checked against local_sw.cpp on 2.4 M
// random windows (substitutions, gaps, repeats, N, six scorings, both exit flags) before the kernel was written.
// A lane holds SL columns, the job's slen of them right-aligned (register r = column r - (SL - slen)): the column its right
// neighbour's diagonal comes from is always register SL - 1.  No LDS but the staged target rows: 102 / 140 registers (both forms of the
// row -- with and without a test per column -- are in the kernel), no waits for lane-private LDS columns (the form above spends its time there),
// a row of ten columns is ~170 instructions for four jobs.  The kernel against the form above, word for word: scripts/msw_bench.py (MSW_HARD),
// in the GPU suite.
__device__ __forceinline__ int msw_scan_max16(int v)          // inclusive max-scan over the 16 lanes of a DPP row
{
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
	return v;
}
template <int SL, class QF, class TF>
__device__ void msw_pass_reg(const msw_args_t &A, const bool on, const int qlen, const int tlen, QF qcode, TF trow, const int xtra,
                             uint8_t *tbuf, uint32_t *blist, msw_res_t &R)
{
	constexpr int NQ = (SL + 3) / 4;
	const int lane = threadIdx.x & 63, l = lane & 15;
	const int slen = on ? (qlen + 15) / 16 : 0;
	const int r0 = SL - slen;                                  // first register in use
	const int minsc = (xtra & BMH_SW_XSUBO) ? xtra & 0xffff : 0x10000, endsc = (xtra & BMH_SW_XSTOP) ? xtra & 0xffff : 0x10000;
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	const int mn = min(min(A.a, -A.b), -1), mx = max(max(A.a, -A.b), -1);
	const int shift = (256 - (mn & 0xff)) & 0xff;
	const int eS = A.e_ins * slen;
	int H[SL], E[SL], Hm[SL];
	uint32_t qsel[NQ];
#pragma unroll
	for (int u = 0; u < NQ; ++u) qsel[u] = 0x05050505u;
#pragma unroll
	for (int r = 0; r < SL; ++r) {
		H[r] = E[r] = Hm[r] = 0;
		const int k = l * slen + (r - r0);
		const uint32_t c = (on && r >= r0 && k < qlen) ? (uint32_t)qcode(k) : 5u;       // 5: padding, scores 0 against everything
		qsel[r >> 2] = (qsel[r >> 2] & ~(0xFFu << (8 * (r & 3)))) | (c << (8 * (r & 3)));
	}
	// scores of a row as a byte table: codes 0..3 from tbl_lo (match a, mismatch -b; a target N: -1 everywhere), N (4) -1 and padding (5) 0 from tbl_hi
	const uint32_t mis4 = (uint32_t)(-A.b & 0xFF) * 0x01010101u, xab = (uint32_t)((A.a ^ -A.b) & 0xFF);
	int gmax = 0, te = -1;
	int bl_n = 0, bl_val = 0, bl_i = -2;
	bool alive = on && tlen > 0;
	int i = 0, mxv = 0;
	// every job of the wave fills its lanes' SL columns (the usual case: mates of one length): the row without a test per column
	const bool full = !__any(on && slen != SL);
	// one row: first sweep, the inflow from the left, the row's maximum over the first sweep's H (FULL: no column test)
	auto row = [&](auto full_c) {
		constexpr bool FULL = decltype(full_c)::value;
		const int t = tbuf[i & (MSW_TBUF - 1)];
		const uint32_t tbl_lo = t > 3 ? 0xFFFFFFFFu : mis4 ^ (xab << (8 * t));
		int hv = msw_shl(H[SL - 1], l);
		int f = 0;
		mxv = 0;
#pragma unroll
		for (int r = 0; r < SL; ++r) {
			if (FULL || r >= r0) {
				const uint32_t sc4 = __builtin_amdgcn_perm(0x000000FFu, tbl_lo, qsel[r >> 2]);
				const int S = (int)(int8_t)(sc4 >> (8 * (r & 3)));
				const int h = max(max(hv + S, E[r]), f);            // >= 0: E and F are
				mxv = max(mxv, h);
				hv = H[r];
				H[r] = h;
				E[r] = max(max(E[r] - A.e_del, h - oe_del), 0);
				f = max(max(f - A.e_ins, h - oe_ins), 0);
			}
		}
		// what flows in from the left: lane l' sends f out, e_ins per column further on
		// (lane 0 receives a large negative value, not a select behind the shift: around `l == 0 ? 0 : ...` the compiler built a branch and folded the DPP read
		// INTO it -- with lane 0 masked off, lane 1 read a zero for what lane 0 sends out and insertions that cross from the first lane into the second lost
		// their F: 2 of 200 000 hard windows, found by scripts/msw_bench.py's kernel-against-kernel run)
		const int Y = msw_scan_max16(f + eS * l);
		const int fin = max(__builtin_amdgcn_update_dpp(-(1 << 28), Y, 0x111, 0xf, 0xf, false) - eS * (l - 1), 0);
		if (__any(fin > 0)) {
			int fv = fin + A.e_ins * r0;                            // fin - e_ins * (r - r0) at register r
#pragma unroll
			for (int r = 0; r < SL; ++r) {
				if (FULL || r >= r0) H[r] = max(H[r], fv);
				fv -= A.e_ins;
			}
		}
	};
	for (; __any(alive); ++i) {
		if ((i & (MSW_TBUF - 1)) == 0) {
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			for (int u = l; u < MSW_TBUF; u += 16) tbuf[u] = (uint8_t)((alive && i + u < tlen) ? trow(i + u) : 4);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		}
		const bool run = alive && i < tlen;
		if (run) {                                              // (all sixteen lanes of a group agree: the DPP steps below stay inside it)
			if (full) row(std::true_type()); else row(std::false_type());
			const int imax = msw_gmax(mxv);
			if (imax >= minsc && l == 0) {                       // ksw.c:519-527: consecutive rows keep one entry, at the row of their maximum so far
				if (bl_i + 1 != i) { if (bl_i >= 0) blist[bl_n++] = (uint32_t)bl_val << 16 | (uint32_t)bl_i; bl_val = imax; bl_i = i; }
				else if (bl_val < imax) { bl_val = imax; bl_i = i; }
			}
			if (imax > gmax) {
				gmax = imax; te = i;
#pragma unroll
				for (int r = 0; r < SL; ++r) Hm[r] = H[r];
				if (gmax + shift >= 255 || gmax >= endsc) alive = false;
			}
			if (i + 1 >= tlen) alive = false;
		}
	}
	if (l == 0 && bl_i >= 0) blist[bl_n++] = (uint32_t)bl_val << 16 | (uint32_t)bl_i;
	bl_n = __shfl(bl_n, lane & 48);
	R.score = gmax + shift < 255 ? gmax : 255;
	R.te = te; R.qe = -1; R.score2 = -1; R.te2 = -1;
	if (on && R.score != 255) {
		// end of the query: the largest H of the best row, the smallest position among equals (ksw.c:540-547)
		int key = -1;
#pragma unroll
		for (int r = 0; r < SL; ++r) if (r >= r0) key = max(key, Hm[r] << 12 | (0xFFF - (l * slen + r - r0)));
		key = msw_gmax(key);
		R.qe = 0xFFF - (key & 0xFFF);
		if (bl_n > 0) {
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_s_waitcnt(0);
			const int d = (R.score + mx - 1) / mx, low = te - d, high = te + d;
			int best = -1;
			for (int u = l; u < bl_n; u += 16) {
				const uint32_t x = blist[u];
				const int e = (int)(x & 0xFFFF), v = (int)(x >> 16);
				if (e < low || e > high) best = max(best, (v << 16) | (0xFFFF - min(u, 0xFFFF)));
			}
			best = msw_gmax(best);
			if (best >= 0) { R.score2 = best >> 16; R.te2 = (int)(blist[0xFFFF - (best & 0xFFFF)] & 0xFFFF); }
		}
	}
}

template <int SL> __global__ void __launch_bounds__(16 * MSW_JOBS_PER_BLOCK) msw_reg_kernel(msw_args_t A)
{
	__shared__ uint8_t st[MSW_JOBS_PER_BLOCK][MSW_TBUF];
	const int g = threadIdx.x >> 4;
	const uint32_t jid = blockIdx.x * MSW_JOBS_PER_BLOCK + g;
	const bool on = jid < A.n_jobs;
	bmh_msw_job_t J;
	memset(&J, 0, sizeof(J));
	if (on) J = A.jobs[jid];
	const uint8_t *rd = A.reads + (on ? A.read_offs[J.read] : 0);
	const int l_ms = J.l_ms, is_rev = J.is_rev;
	const long long rb = J.rb;
	const int tlen = (int)(J.re - J.rb);
	auto qfwd = [&](int k) { const int c = msw_nt4(rd[is_rev ? l_ms - 1 - k : k]); return is_rev ? (c < 4 ? 3 - c : 4) : c; };
	auto tfwd = [&](int i) { return msw_text(A, rb + i); };
	msw_res_t R1;
	uint32_t *bl = A.blist + (on ? J.bl_off : 0);
	msw_pass_reg<SL>(A, on, l_ms, tlen, qfwd, tfwd, J.xtra, st[g], bl, R1);
	int tb = -1, qb = -1;
	const bool second = on && (J.xtra & BMH_SW_XSTART) && !((J.xtra & BMH_SW_XSUBO) && R1.score < (J.xtra & 0xffff));
	if (__any(second)) {
		const int qe = R1.qe, te = R1.te;
		auto qrev = [&](int k) { return qfwd(qe - k); };
		auto trev = [&](int i) { return i <= te ? msw_text(A, rb + (te - i)) : msw_text(A, rb + i); };
		msw_res_t R2;
		msw_pass_reg<SL>(A, second, qe + 1, tlen, qrev, trev, BMH_SW_XSTOP | R1.score, st[g], bl, R2);
		if (second && R1.score == R2.score) { tb = R1.te - R2.te; qb = R1.qe - R2.qe; }
	}
	if (on && (threadIdx.x & 15) == 0) {
		int32_t *o = A.out + 7 * (size_t)jid;
		o[0] = R1.score; o[1] = R1.te; o[2] = R1.qe; o[3] = R1.score2; o[4] = R1.te2; o[5] = tb; o[6] = qb;
	}
}

// LDS: four 16-bit arrays and the query codes, each [cap][MSW_JOBS_PER_BLOCK][16]: segment j of the block's eight jobs side by side,
// so that a wave's four jobs read one row of 64 dwords -- every bank once (with the jobs' columns one after the other all four hit
// the same banks) -- and `cap`, the longest column of the batch, sizes the block: 12 KB for 150 bp mates instead of 46 KB for the
// longest the kernel takes, four times the waves per CU for a loop that waits for its LDS round trips.
template <bool BYTE_ALL> __global__ void __launch_bounds__(16 * MSW_JOBS_PER_BLOCK) msw_kernel(msw_args_t A, const int cap)
{
	extern __shared__ __align__(16) uint8_t msw_lds[];
	__shared__ uint8_t st[MSW_JOBS_PER_BLOCK][MSW_TBUF];
	const int g = threadIdx.x >> 4;
	const uint32_t jid = blockIdx.x * MSW_JOBS_PER_BLOCK + g;
	const bool on = jid < A.n_jobs;
	bmh_msw_job_t J;
	memset(&J, 0, sizeof(J));
	if (on) J = A.jobs[jid];
	const int lanes = (J.xtra & BMH_SW_XBYTE) ? 16 : 8;
	const uint8_t *rd = A.reads + (on ? A.read_offs[J.read] : 0);
	const int l_ms = J.l_ms, is_rev = J.is_rev;
	const long long rb = J.rb;
	const int tlen = (int)(J.re - J.rb);
	// the mate as mem_matesw aligns it: itself, or its reverse complement
	auto qfwd = [&](int k) { const int c = msw_nt4(rd[is_rev ? l_ms - 1 - k : k]); return is_rev ? (c < 4 ? 3 - c : 4) : c; };
	auto tfwd = [&](int i) { return msw_text(A, rb + i); };
	msw_res_t R1;
	uint32_t *bl = A.blist + (on ? J.bl_off : 0);
	uint16_t *sH0 = (uint16_t *)msw_lds + g * 16, *sH1 = sH0 + (size_t)cap * MSW_ROW, *sE = sH1 + (size_t)cap * MSW_ROW, *sHm = sE + (size_t)cap * MSW_ROW;
	uint8_t *sq = msw_lds + (size_t)cap * MSW_ROW * 8 + g * 16;
	msw_pass<BYTE_ALL>(A, on, lanes, l_ms, tlen, qfwd, tfwd, J.xtra, sH0, sH1, sE, sHm, sq, st[g], bl, R1);
	int tb = -1, qb = -1;
	const bool second = on && (J.xtra & BMH_SW_XSTART) && !((J.xtra & BMH_SW_XSUBO) && R1.score < (J.xtra & 0xffff));
	if (__any(second)) {
		// start positions: the same pass over the reversed prefixes, stopped at the score (ksw.c:722-736; the target keeps its full length,
		// its rows beyond the reversed prefix are the original ones)
		const int qe = R1.qe, te = R1.te;
		auto qrev = [&](int k) { return qfwd(qe - k); };
		auto trev = [&](int i) { return i <= te ? msw_text(A, rb + (te - i)) : msw_text(A, rb + i); };
		msw_res_t R2;
		msw_pass<BYTE_ALL>(A, second, lanes, qe + 1, tlen, qrev, trev, BMH_SW_XSTOP | R1.score, sH0, sH1, sE, sHm, sq, st[g], bl, R2);
		if (second && R1.score == R2.score) { tb = R1.te - R2.te; qb = R1.qe - R2.qe; }
	}
	if (on && (threadIdx.x & 15) == 0) {
		int32_t *o = A.out + 7 * (size_t)jid;
		o[0] = R1.score; o[1] = R1.te; o[2] = R1.qe; o[3] = R1.score2; o[4] = R1.te2; o[5] = tb; o[6] = qb;
	}
}

// ---- the rescue's windows found on the device (the first of pair_post.cpp's two walks): mem_matesw's tests up to the ksw_align2 call
// (/root/reference/src/bwamem_pair.c:119-150) for every pair, on the regions mem_sort_dedup_patch left on the device.  A pair is
// walked twice: pass 0 counts the alignments its mem_matesw calls would ask for (and the words of second-best bookkeeping they need), pass 1 -- after a
// scan of the counts -- writes the jobs and their keys in pair order, the order the host's walk collected them in.  The reference calls
// mem_sort_dedup_patch on the mate's list after a window was reached, which can remove a hit and with it change what a LATER call of the same pair
// skips; the lists are taken as they are here.  That is exact where it matters: `active` (a call of the pair got as far as a window) is decided before
// the first such call, and for an active pair the host's walk looks its alignments up by (end, hit, orientation) and computes the ones it does not
// find itself (pair_post.cpp: matesw, sw_mode 2) -- a list that differs costs host time or a wasted job, never a different record.
struct rj_pes_t { int low, high, failed; };
struct rj_args_t {
	long long l_pac; int n_contigs; const int64_t *ctg_off;
	rj_pes_t pes[4]; int pen_unpaired, max_matesw, min_seed_len, a;
	const int32_t *ded; const uint32_t *opr, *off, *lens;      // [..][16] regions behind mem_sort_dedup_patch, their number and first record per read, read lengths
	uint32_t n_pairs;
	uint32_t *cnt, *blw;                                        // [n_pairs + 1]: pass 0 counts, pass 1 reads the scanned offsets
	uint8_t *active;
	uint32_t *stat;                                             // [0] longest lane-private column in segments, [1] != 0: a job in 16-bit mode, [2..3] words of bookkeeping (64 bits)
	bmh_msw_job_t *jobs; bmh_msw_key_t *keys;
};

__device__ __forceinline__ int rj_infer_dir(long long l_pac, long long b1, long long b2, long long *dist)      // mem_infer_dir
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const long long p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
__device__ __forceinline__ long long rj_rb(const int32_t *g) { return (long long)(uint32_t)g[4] | (long long)g[5] << 32; }
__device__ __forceinline__ int rj_pos2rid(const rj_args_t &A, long long pos_f)          // bns_pos2rid
{
	if (pos_f >= A.l_pac) return -1;
	if (A.n_contigs <= 1) return 0;
	int left = 0, mid = 0, right = A.n_contigs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= A.ctg_off[mid]) {
			if (mid == A.n_contigs - 1) break;
			if (pos_f < A.ctg_off[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
__device__ __forceinline__ bool rj_takes(int l_ms, long long tlen, int xtra)        // bmh_matesw_device_takes
{
	const int lanes = (xtra & BMH_SW_XBYTE) ? 16 : 8;
	return l_ms > 0 && (l_ms + lanes - 1) / lanes <= MSW_SLEN && tlen > 0 && tlen < 65536;
}

// OR over the 16 lanes of a row (every lane gets it)
__device__ __forceinline__ int rj_row_or(int v)
{
	v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
	v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
	v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
	v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
	return v;
}

// One pair per ROW of 16 lanes (four pairs a wave): the calls of a pair are walked in order by the row, the mate's hits of a call are tested 64 at a time,
// four by each lane (a pair of repeat-rich reads is 50 calls x hundreds of hits: one lane per pair took 15 ms a pass for the longest of them).
template <int PASS> __global__ void __launch_bounds__(64) rescue_jobs_kernel(rj_args_t A)
{
	const uint32_t l = threadIdx.x & 15u, p = blockIdx.x * 4u + (threadIdx.x >> 4);
	if (p >= A.n_pairs) return;                                       // (a whole row leaves: the DPP steps above never cross rows)
	const long long l_pac = A.l_pac;
	uint32_t nj = 0, nb = 0;
	const uint32_t at = PASS ? A.cnt[p] : 0u, bl = PASS ? A.blw[p] : 0u;
	int seg_max = 0, wide = 0;
	bool active = false;
	if (PASS && A.cnt[p + 1] == at) return;
	for (int i = 0; i < 2; ++i) {
		const uint32_t ra = 2 * p + (uint32_t)i, rm = 2 * p + (uint32_t)!i;
		const int na = (int)A.opr[ra], nm = (int)A.opr[rm];
		if (na == 0) continue;
		const int32_t *a = A.ded + 16 * (size_t)A.off[ra], *m = A.ded + 16 * (size_t)A.off[rm];
		const int l_ms = (int)A.lens[rm];
		const int thr = a[1] - A.pen_unpaired;
		int j = 0;
		for (int k = 0; k < na && j < A.max_matesw; ++k) {
			const int32_t *g = a + 16 * k;
			if (g[1] < thr) continue;
			const long long arb = rj_rb(g);
			// ---- one mem_matesw call: which orientations the mate's hits leave open
			int skip = 0;
			for (int r = 0; r < 4; ++r) skip |= (A.pes[r].failed ? 1 : 0) << r;
			for (int t0 = 0; t0 < nm && skip != 15; t0 += 64) {
				int bits = 0;
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const int t = t0 + u * 16 + (int)l;
					if (t < nm) {
						long long dist;
						const int r = rj_infer_dir(l_pac, arb, rj_rb(m + 16 * t), &dist);
						if (dist >= A.pes[r].low && dist <= A.pes[r].high) bits |= 1 << r;
					}
				}
				skip |= rj_row_or(bits);
			}
			for (int r = 0; r < 4 && skip != 15; ++r) {
				if (skip >> r & 1) continue;
				const int is_rev = (r >> 1 != (r & 1)), is_larger = !(r >> 1);
				long long rb, re;
				if (!is_rev) {
					rb = is_larger ? arb + A.pes[r].low : arb - A.pes[r].high;
					re = (is_larger ? arb + A.pes[r].high : arb - A.pes[r].low) + l_ms;
				} else {
					rb = (is_larger ? arb + A.pes[r].low : arb - A.pes[r].high) - l_ms;
					re = is_larger ? arb + A.pes[r].high : arb - A.pes[r].low;
				}
				if (rb < 0) rb = 0;
				if (re > l_pac << 1) re = l_pac << 1;
				if (rb >= re) continue;
				// bns_fetch_seq's clipping to the sequence (and strand) of the window's middle
				const long long mid = (rb + re) >> 1;
				const bool mrev = mid >= l_pac;
				const int rid = rj_pos2rid(A, mrev ? (l_pac << 1) - 1 - mid : mid);
				long long far_beg = A.n_contigs > 1 ? A.ctg_off[rid] : 0, far_end = A.n_contigs > 1 ? (rid + 1 < A.n_contigs ? A.ctg_off[rid + 1] : l_pac) : l_pac;
				if (mrev) { const long long t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
				rb = rb > far_beg ? rb : far_beg;
				re = re < far_end ? re : far_end;
				if (rb >= re || g[13] != rid || re - rb < A.min_seed_len) continue;
				active = true;
				const int xtra = BMH_SW_XSUBO | BMH_SW_XSTART | (l_ms * A.a < 250 ? BMH_SW_XBYTE : 0) | (A.min_seed_len * A.a);
				if (!rj_takes(l_ms, re - rb, xtra)) continue;
				const uint32_t words = (uint32_t)((re - rb) / 2 + 2);
				if (PASS) {
					if (l == 0) {
						bmh_msw_job_t jb;
						jb.rb = rb; jb.re = re; jb.read = rm; jb.l_ms = l_ms; jb.is_rev = is_rev; jb.xtra = xtra; jb.bl_off = bl + nb; jb.pad = 0;
						A.jobs[at + nj] = jb;
						bmh_msw_key_t ky; ky.pair = p; ky.j = (uint16_t)j; ky.i = (uint8_t)i; ky.r = (uint8_t)r;
						A.keys[at + nj] = ky;
					}
				} else {
					const int lanes = (xtra & BMH_SW_XBYTE) ? 16 : 8, sl = (l_ms + lanes - 1) / lanes;
					seg_max = sl > seg_max ? sl : seg_max; wide |= lanes == 8;
				}
				++nj; nb += words;
			}
			++j;
		}
	}
	if (!PASS && l == 0) {
		A.cnt[p] = nj; A.blw[p] = nb; A.active[p] = active ? 1 : 0;
		if (nj) { atomicMax(A.stat, (uint32_t)seg_max); if (wide) atomicOr(A.stat + 1, 1u); atomicAdd((unsigned long long *)(A.stat + 2), (unsigned long long)nb); }
	}
}

// ---- host side
struct msw_scratch_t {
	bmh_msw_job_t *d_jobs; int32_t *d_out; uint32_t *d_bl; size_t cap_jobs, cap_bl;
	// the rescue's windows found on the device
	uint32_t *d_cnt, *d_blw, *d_stat; uint8_t *d_active; bmh_msw_key_t *d_keys; void *d_scan; size_t cap_pairs, cap_keys, cap_scan;
	uint32_t rj_n_jobs, rj_cap; uint64_t rj_bl; bool rj_byte_all; rj_args_t rj;
};
static std::mutex g_msw_mu;
static std::map<std::pair<int, void *>, msw_scratch_t *> g_msw_map;

// the (device, stream) scratch of bmh_matesw_batch_device / bmh_rescue_*_device: freed when the caller retires the stream (stream idle, its device current)
extern "C" void bmh_matesw_release(void *stream_)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return;
	msw_scratch_t *S = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_msw_mu);
		auto it = g_msw_map.find(std::make_pair(dev, stream_));
		if (it == g_msw_map.end()) return;
		S = it->second;
		g_msw_map.erase(it);
	}
	void *ps[] = {S->d_jobs, S->d_out, S->d_bl, S->d_cnt, S->d_blw, S->d_stat, S->d_active, S->d_keys, S->d_scan};
	for (void *q : ps) if (q) (void)hipFree(q);
	delete S;
}

// can the kernel take this job? (the host computes the others itself)
extern "C" int bmh_matesw_device_takes(int l_ms, int64_t tlen, int xtra)
{
	const int lanes = (xtra & BMH_SW_XBYTE) ? 16 : 8;
	return l_ms > 0 && (l_ms + lanes - 1) / lanes <= MSW_SLEN && tlen > 0 && tlen < 65536;
}

static int msw_scratch_of(void *stream_, msw_scratch_t **out)
{
	int dev = 0;
	HIPCK(hipGetDevice(&dev));
	std::lock_guard<std::mutex> lk(g_msw_mu);
	auto key = std::make_pair(dev, stream_);
	auto it = g_msw_map.find(key);
	if (it == g_msw_map.end()) { *out = new msw_scratch_t(); memset((void *)*out, 0, sizeof(msw_scratch_t)); g_msw_map[key] = *out; }
	else *out = it->second;
	return BMH_OK;
}
static int msw_room(msw_scratch_t *S, uint64_t n_jobs, uint64_t bl)
{
	if (n_jobs > S->cap_jobs) {
		if (S->d_jobs) (void)hipFree(S->d_jobs);
		if (S->d_out) (void)hipFree(S->d_out);
		S->d_jobs = nullptr; S->d_out = nullptr; S->cap_jobs = 0;
		const size_t c = n_jobs + n_jobs / 4 + 1024;
		HIPCK(hipMalloc((void **)&S->d_jobs, sizeof(bmh_msw_job_t) * c)); HIPCK(hipMalloc((void **)&S->d_out, sizeof(int32_t) * 7 * c));
		S->cap_jobs = c;
	}
	if (bl > S->cap_bl) {
		if (S->d_bl) (void)hipFree(S->d_bl);
		S->d_bl = nullptr; S->cap_bl = 0;
		const size_t c = bl + bl / 4 + 1024;
		HIPCK(hipMalloc((void **)&S->d_bl, 4 * c));
		S->cap_bl = c;
	}
	return BMH_OK;
}
// the jobs in S->d_jobs (bl_off set) -> S->d_out
static int msw_launch(msw_scratch_t *S, const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const bmh_ext_params_t *ep, uint64_t n_jobs, int cap, bool byte_all, hipStream_t st)
{
	msw_args_t A;
	A.jobs = S->d_jobs; A.n_jobs = (uint32_t)n_jobs; A.reads = d_reads; A.read_offs = d_offs; A.pac = idx->dev.pac; A.l_pac = (long long)idx->dev.l_pac;
	A.a = ep->a; A.b = ep->b; A.o_del = ep->o_del; A.e_del = ep->e_del; A.o_ins = ep->o_ins; A.e_ins = ep->e_ins;
	A.blist = S->d_bl; A.out = S->d_out;
	// the register form: every job in byte mode, columns of at most 16 segments, gap opens that cost something (its F is the plain recurrence only then)
	const unsigned nblk = (unsigned)((n_jobs + MSW_JOBS_PER_BLOCK - 1) / MSW_JOBS_PER_BLOCK);
	const bool reg_form = byte_all && cap <= 16 && ep->o_ins > 0 && ep->e_ins > 0 && ep->a < 100 && ep->b < 100 && bmh_tune("MSW_REG", 1) != 0;
	if (reg_form && cap <= 10) msw_reg_kernel<10><<<nblk, 16 * MSW_JOBS_PER_BLOCK, 0, st>>>(A);
	else if (reg_form) msw_reg_kernel<16><<<nblk, 16 * MSW_JOBS_PER_BLOCK, 0, st>>>(A);
	else if (byte_all) msw_kernel<true><<<nblk, 16 * MSW_JOBS_PER_BLOCK, (size_t)cap * MSW_ROW * 9, st>>>(A, cap);
	else msw_kernel<false><<<nblk, 16 * MSW_JOBS_PER_BLOCK, (size_t)cap * MSW_ROW * 9, st>>>(A, cap);
	HIPCK(hipGetLastError());
	return BMH_OK;
}

extern "C" int bmh_matesw_batch_device(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const bmh_ext_params_t *ep,
                                       bmh_msw_job_t *jobs, uint64_t n_jobs, int32_t *out, void *stream_)
{
	if (!idx || !idx->dev.pac || !d_reads || !d_offs || !ep || (n_jobs && (!jobs || !out))) { bmh_set_error("bmh_matesw_batch_device: null argument"); return BMH_EINVAL; }
	if (n_jobs == 0) return BMH_OK;
	if (n_jobs >> 31) { bmh_set_error("bmh_matesw_batch_device: too many jobs"); return BMH_ECAPACITY; }
	hipStream_t st = (hipStream_t)stream_;
	msw_scratch_t *S;
	{ const int rc = msw_scratch_of(stream_, &S); if (rc != BMH_OK) return rc; }
	uint64_t bl = 0;
	int cap = 1;                                             // the longest lane-private column of the batch, in segments
	bool byte_all = true;
	for (uint64_t k = 0; k < n_jobs; ++k) { const int lanes = (jobs[k].xtra & BMH_SW_XBYTE) ? 16 : 8; const int sl = (jobs[k].l_ms + lanes - 1) / lanes; cap = sl > cap ? sl : cap; byte_all = byte_all && lanes == 16; }
	if (cap > MSW_SLEN) { bmh_set_error("bmh_matesw_batch_device: a job the kernel does not take (see bmh_matesw_device_takes)"); return BMH_EINVAL; }
	for (uint64_t k = 0; k < n_jobs; ++k) { jobs[k].bl_off = (uint32_t)bl; bl += (uint64_t)(jobs[k].re - jobs[k].rb) / 2 + 2; if (bl >> 32) { bmh_set_error("bmh_matesw_batch_device: windows too long"); return BMH_ECAPACITY; } }
	{ const int rc = msw_room(S, n_jobs, bl); if (rc != BMH_OK) return rc; }
	HIPCK(hipMemcpyAsync(S->d_jobs, jobs, sizeof(bmh_msw_job_t) * n_jobs, hipMemcpyHostToDevice, st));
	{ const int rc = msw_launch(S, idx, d_reads, d_offs, ep, n_jobs, cap, byte_all, st); if (rc != BMH_OK) return rc; }
	HIPCK(hipMemcpyAsync(out, S->d_out, sizeof(int32_t) * 7 * n_jobs, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	return BMH_OK;
}

// The rescue's windows of a batch of interleaved pairs found on the device (rescue_jobs_kernel), in two calls on one stream:
// bmh_rescue_count_device: pair_off [n_reads / 2 + 1] receives the first job of every pair (and the number of jobs at the end), active [n_reads / 2] whether a
// mem_matesw call of the pair reached a window; returns the number of jobs.  bmh_rescue_run_device: the jobs' keys [n_jobs] and results [n_jobs][7]
// (as bmh_matesw_batch_device).  pes [4][5] as bmh_finalize_pairs' pes_out.  Both wait for the stream.
extern "C" int64_t bmh_rescue_count_device(const bmh_index_t *idx, const bmh_rescue_in_t *in, const bmh_ext_params_t *ep, int min_seed_len, const bmh_pe_opt_t *pe,
                                           const double *pes, uint32_t n_reads, uint32_t *pair_off, uint8_t *active, void *stream_)
{
	if (!idx || !idx->dev.pac || !in || !in->d_dedup || !in->d_opr || !in->d_roff || !in->d_lens || !ep || !pe || !pes || !pair_off || !active || (in->n_contigs > 1 && !in->d_ctg_off)) {
		bmh_set_error("bmh_rescue_count_device: null argument"); return BMH_EINVAL;
	}
	const uint32_t np = n_reads / 2;
	hipStream_t st = (hipStream_t)stream_;
	msw_scratch_t *S;
	{ const int rc = msw_scratch_of(stream_, &S); if (rc != BMH_OK) return rc; }
	S->rj_n_jobs = 0;
	pair_off[0] = 0;
	if (np == 0) return 0;
	if ((size_t)np + 1 > S->cap_pairs) {
		void *ps[] = {S->d_cnt, S->d_blw, S->d_active};
		for (void *q : ps) if (q) (void)hipFree(q);
		S->d_cnt = S->d_blw = nullptr; S->d_active = nullptr; S->cap_pairs = 0;
		const size_t c = (size_t)np + np / 4 + 1024;
		HIPCK(hipMalloc((void **)&S->d_cnt, 4 * c)); HIPCK(hipMalloc((void **)&S->d_blw, 4 * c)); HIPCK(hipMalloc((void **)&S->d_active, c));
		S->cap_pairs = c;
	}
	if (!S->d_stat) HIPCK(hipMalloc((void **)&S->d_stat, 16));
	const size_t sb = bmh_pair_scan_bytes(np + 1);
	if (sb > S->cap_scan) { if (S->d_scan) (void)hipFree(S->d_scan); S->d_scan = nullptr; S->cap_scan = 0; HIPCK(hipMalloc(&S->d_scan, sb)); S->cap_scan = sb; }
	rj_args_t &A = S->rj;
	memset(&A, 0, sizeof(A));
	A.l_pac = (long long)idx->dev.l_pac; A.n_contigs = in->n_contigs > 1 ? in->n_contigs : 1; A.ctg_off = in->n_contigs > 1 ? in->d_ctg_off : nullptr;
	for (int d = 0; d < 4; ++d) { A.pes[d].low = (int)pes[5 * d]; A.pes[d].high = (int)pes[5 * d + 1]; A.pes[d].failed = (int)pes[5 * d + 2]; }
	A.pen_unpaired = pe->pen_unpaired; A.max_matesw = pe->max_matesw; A.min_seed_len = min_seed_len; A.a = ep->a;
	A.ded = in->d_dedup; A.opr = in->d_opr; A.off = in->d_roff; A.lens = in->d_lens; A.n_pairs = np;
	A.cnt = S->d_cnt; A.blw = S->d_blw; A.active = S->d_active; A.stat = S->d_stat;
	HIPCK(hipMemsetAsync(S->d_stat, 0, 16, st));
	HIPCK(hipMemsetAsync(S->d_cnt + np, 0, 4, st)); HIPCK(hipMemsetAsync(S->d_blw + np, 0, 4, st));
	rescue_jobs_kernel<0><<<(np + 3) / 4, 64, 0, st>>>(A);
	HIPCK(hipGetLastError());
	{ const int rc = bmh_pair_scan(S->d_cnt, S->d_cnt, np + 1, S->d_scan, S->cap_scan, st); if (rc != BMH_OK) return rc; }
	uint32_t stat[4] = {0, 0, 0, 0}, last_bl = 0;
	HIPCK(hipMemcpyAsync(pair_off, S->d_cnt, 4 * ((size_t)np + 1), hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(active, S->d_active, np, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(stat, S->d_stat, 16, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	const uint32_t nj = pair_off[np];
	if (nj >> 31) { bmh_set_error("bmh_rescue_count_device: too many jobs"); return BMH_ECAPACITY; }
	if (stat[3]) { bmh_set_error("bmh_rescue_count_device: windows too long"); return BMH_ECAPACITY; }      // (the words of bookkeeping: offsets of 32 bits, like bmh_matesw_batch_device)
	{ const int rc = bmh_pair_scan(S->d_blw, S->d_blw, np + 1, S->d_scan, S->cap_scan, st); if (rc != BMH_OK) return rc; }
	HIPCK(hipMemcpyAsync(&last_bl, S->d_blw + np, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	S->rj_n_jobs = nj; S->rj_bl = last_bl; S->rj_cap = stat[0] ? stat[0] : 1; S->rj_byte_all = stat[1] == 0;
	return (int64_t)nj;
}

extern "C" int bmh_rescue_run_device(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const bmh_ext_params_t *ep, bmh_msw_key_t *keys, int32_t *out, void *stream_)
{
	if (!idx || !idx->dev.pac || !d_reads || !d_offs || !ep) { bmh_set_error("bmh_rescue_run_device: null argument"); return BMH_EINVAL; }
	hipStream_t st = (hipStream_t)stream_;
	msw_scratch_t *S;
	{ const int rc = msw_scratch_of(stream_, &S); if (rc != BMH_OK) return rc; }
	const uint64_t nj = S->rj_n_jobs;
	if (nj == 0) return BMH_OK;
	if (!keys || !out) { bmh_set_error("bmh_rescue_run_device: null argument"); return BMH_EINVAL; }
	if ((int)S->rj_cap > MSW_SLEN) { bmh_set_error("bmh_rescue_run_device: internal error: a job the kernel does not take"); return BMH_EINVAL; }
	{ const int rc = msw_room(S, nj, S->rj_bl); if (rc != BMH_OK) return rc; }
	if (nj > S->cap_keys) {
		if (S->d_keys) (void)hipFree(S->d_keys);
		S->d_keys = nullptr; S->cap_keys = 0;
		const size_t c = nj + nj / 4 + 1024;
		HIPCK(hipMalloc((void **)&S->d_keys, sizeof(bmh_msw_key_t) * c));
		S->cap_keys = c;
	}
	rj_args_t A = S->rj;
	A.jobs = S->d_jobs; A.keys = S->d_keys;
	rescue_jobs_kernel<1><<<(A.n_pairs + 3) / 4, 64, 0, st>>>(A);
	HIPCK(hipGetLastError());
	{ const int rc = msw_launch(S, idx, d_reads, d_offs, ep, nj, (int)S->rj_cap, S->rj_byte_all, st); if (rc != BMH_OK) return rc; }
	HIPCK(hipMemcpyAsync(keys, S->d_keys, sizeof(bmh_msw_key_t) * nj, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(out, S->d_out, sizeof(int32_t) * 7 * nj, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	return BMH_OK;
}
