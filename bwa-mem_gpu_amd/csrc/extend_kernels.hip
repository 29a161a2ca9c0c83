// Zero-trimmed affine-gap seed extension (ksw_extend2) on gfx950 -- hand-written HIP, wave64.
//
// Contract = ksw_extend2 of the reference, opt_ext == 0 path
// (/root/reference/src/ksw.c:864-986) followed by the local-vs-to-end rule of
// decoy_cpu_align (src/bwamem.c:1893-1901); results are bit-identical, including
// the row-sequential parts of the algorithm: the [beg,end) trimming driven by the
// previous row's zeros (:963-970), the m==0 break (:946), z-drop (:951-959), the
// "last column on ties" row maximum (:928) and the "last row on ties" gscore (:943).
//
// Mapping: one target row per step, lanes own the query columns (C consecutive
// columns per lane, a template parameter).  Two kernels share the row algebra:
//   extend16_kernel<C>   qlen <= 16*C <= 288: FOUR alignments per wave, one per 16-lane
//                        DPP row (row_shr scans and quad_perm/row_mirror butterflies are
//                        natively 16 wide, so the four rows never interact); all per-
//                        alignment control (beg/end/max/alive) is vector state replicated
//                        over the row's lanes; jobs are sorted by (class, tlen) so the four
//                        alignments of a wave end together.
//   extend_wide_kernel<C> 288 < qlen <= 512: one alignment per wave, 64 lanes x C columns.  Because ksw_extend2 opens gaps from the diagonal value M and not
// from H (:929-938), M of a whole row depends only on the previous row, and
// F(i,j+1) = max(F(i,j)-e, max(M(i,j)-oe,0)) is a max-plus prefix scan along the
// row: F(i,j) = max_{j'<j}(t(j') + e*j') - e*(j-1).  The scan and the row maximum
// run on DPP row shifts/broadcasts (no LDS); beg/end/mj come from __ballot.
// A lane-per-column ANTI-DIAGONAL sweep cannot reproduce `end` exactly: whether
// cell (i,j) is inside the trimmed range depends on cells (i-1, j'>j) that such a
// sweep has not computed yet ("why rows, not anti-diagonals").
//
// All row state lives in VGPRs; per alignment the wave reads qlen+tlen bytes and
// writes 12 (or 36) bytes, so the kernel is integer-VALU bound, not HBM bound.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>
#include "bmh_internal.h"
#include "wtrace.h"

#define NEG_INF (-(1 << 29))
#define EXT_T_CAP 1024        // target bases of an alignment staged in LDS by extend16_kernel
#ifndef EXT_DRAW_CHUNK
#define EXT_DRAW_CHUNK 16      // jobs a wave takes from its class counter per atomic (8 / 16 / 32 / 64 measured: 33.1 / 33.1 / 33.4 / 34.1 ms per batch)
#endif

// inclusive max-scan over the 64 lanes (Kogge-Stone inside 16-lane rows on DPP
// row_shr, then row_bcast:15 / row_bcast:31 across rows); lane 63 ends with the total
__device__ __forceinline__ int wave_scan_max(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x111, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x112, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x114, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x118, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x142, 0xa, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x143, 0xc, 0xf, false); v = max(v, t);
	return v;
}

// value of lane-1 (wave_shr:1); lane 0 receives `fill`
__device__ __forceinline__ int wave_shr1(int v, int fill)
{
	return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}

struct ext_args_t {
	const uint8_t *q, *t;
	const uint32_t *qoff, *qlen, *toff, *tlen, *h0;
	// descriptor mode (desc != 0; jobs of the device job builder, nothing materialised): query bases are read from the
	// ASCII reads at reads + jq_src[id], target bases from the 2-bit reference at text position jt0[id] + i; LEFT jobs
	// (job_side[id] == 0) run both backwards (src/bwamem.c:1328-1334).  q/t/qoff/toff are unused then.
	int desc;
	const uint8_t *reads, *pac; long long l_pac;
	const uint32_t *jq_src, *job_side; const long long *jt0;
	const uint32_t *ids;          // alignment ids of this class
	const uint4 *recs;            // the same list as 32-byte job records (ext_scatter_kernel), two uint4 per job: {id, qlen | tlen << 16, h0, jq_src or qoff} {job_side, toff, jt0 lo, jt0 hi}
	const uint32_t *count;        // how many
	uint32_t *ctr;                // next unassigned job of this class (extend16_kernel draws from it)
	int32_t *out, *raw;
	int a, b, o_del, e_del, o_ins, e_ins, zdrop, end_bonus;
	unsigned long long *stats;    // debug (BMH_EXT_STATS): [0] rows executed, [1] sum of tlen, [2] alignments, [3] wave-rows
};

// where the bases of one job come from
struct job_src_t { const uint8_t *qp; int qstep; const uint8_t *tp; long long t0; int tdir; };

__device__ __forceinline__ job_src_t ext_job_src(const ext_args_t &A, uint32_t id, bool have, int qlen, int tlen)
{
	job_src_t s;
	if (A.desc) {
		const bool left = have && A.job_side[id] == 0;
		s.qp = A.reads + (have ? A.jq_src[id] : 0) + (left ? qlen - 1 : 0); s.qstep = left ? -1 : 1;
		s.t0 = (have ? A.jt0[id] : 0) + (left ? tlen - 1 : 0); s.tdir = left ? -1 : 1; s.tp = nullptr;
	} else {
		s.qp = A.q + (have ? A.qoff[id] : 0); s.qstep = 1; s.tp = A.t + (have ? A.toff[id] : 0); s.t0 = 0; s.tdir = 1;
	}
	return s;
}
__device__ __forceinline__ int ext_q_at(const ext_args_t &A, const job_src_t &s, int i)      // query code 0..4
{
	int v = (int)s.qp[(long)i * s.qstep];
	if (A.desc) { v &= 0xDF; v = v == 'A' ? 0 : v == 'C' ? 1 : v == 'G' ? 2 : v == 'T' ? 3 : 4; }
	return v;
}
__device__ __forceinline__ int ext_t_at(const ext_args_t &A, const job_src_t &s, int i)      // target code 0..4
{
	if (!A.desc) return (int)s.tp[i];
	const long long p = s.t0 + (long long)i * s.tdir;
	const bool rev = p >= A.l_pac;
	const long long f = rev ? (A.l_pac << 1) - 1 - p : p;
	const int c = (A.pac[f >> 2] >> ((~f & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}

template <int C>
__global__ void __launch_bounds__(256) extend_wide_kernel(ext_args_t A)
{
	const int lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t n = A.count[0];
	const uint32_t *ids = A.ids + A.count[1];      // count[1] = offset of this class in the sorted id list
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	for (uint32_t w = wave; w < n; w += n_waves) {
		const uint32_t id = ids[w];
		const int qlen = (int)A.qlen[id], tlen = (int)A.tlen[id], h0 = (int)A.h0[id];
		const job_src_t src = ext_job_src(A, id, true, qlen, tlen);
		int H[C], E[C], qb[C];
#pragma unroll
		for (int c = 0; c < C; ++c) {
			int j = lane * C + c;
			qb[c] = j < qlen ? ext_q_at(A, src, j) : 4;
			int v = h0 - oe_ins - j * A.e_ins;            // H(-1,j), ksw.c:880-883
			H[c] = (j < qlen && v > 0) ? v : 0;
			E[c] = 0;
		}
		int beg = 0, end = qlen, mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
		int tchunk = 0;
		for (int i = 0; i < tlen; ++i) {
			if ((i & 63) == 0) tchunk = (i + lane < tlen) ? ext_t_at(A, src, i + lane) : 4;
			const int ti = __builtin_amdgcn_readlane(tchunk, i & 63);
			// H(i-1,-1): the first-column value of the previous row (ksw.c:909-914, :880)
			const int hm1 = i == 0 ? h0 : max(0, h0 - (A.o_del + A.e_del * i));
			const int left = wave_shr1(H[C - 1], beg == 0 ? hm1 : 0);
			int M[C], g[C];
			int agg = NEG_INF;
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const int j = lane * C + c;
				const bool act = j >= beg && j < end;
				const int hd = c == 0 ? left : H[c - 1];
				const int sc = (ti > 3 || qb[c] > 3) ? -1 : (ti == qb[c] ? A.a : -A.b);
				const int m = (act && hd) ? hd + sc : 0;
				M[c] = m;
				const int tins = max(m - oe_ins, 0);
				g[c] = act ? tins + A.e_ins * j : NEG_INF;
				agg = max(agg, g[c]);
			}
			const int incl = wave_scan_max(agg);
			int run = wave_shr1(incl, NEG_INF);           // max of g over all columns left of this lane
			int lm = 0, lmj = -1;                           // lane-local row maximum and its last column (m = 0, mj = -1 start, ksw.c:900)
			bool nzH = false, nzE = false;
			int firstH = 1 << 20, lastH = -1, firstE = 1 << 20, lastE = -1;
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const int j = lane * C + c;
				const bool act = j >= beg && j < end;
				const int f = max(0, run - A.e_ins * (j - 1));
				run = max(run, g[c]);
				int h = max(max(M[c], E[c]), f);
				int e = max(E[c] - A.e_del, max(M[c] - oe_del, 0));
				h = act ? h : 0;
				e = act ? e : 0;
				H[c] = h; E[c] = e;
				if (act && h >= lm) { lm = h; lmj = j; }
				if (h) { nzH = true; firstH = min(firstH, j); lastH = j; }
				if (e) { nzE = true; firstE = min(firstE, j); lastE = j; }
			}
			// row maximum m and mj = last column holding it (ksw.c:928-929)
			const int m = __builtin_amdgcn_readlane(wave_scan_max(lm), 63);
			if (end == qlen) {                              // ksw.c:942-945
				const int jl = qlen - 1;
				int h1 = 0;
				if (qlen > 0) {
					int src = 0;
#pragma unroll
					for (int c = 0; c < C; ++c) if (jl % C == c) src = H[c];
					h1 = __builtin_amdgcn_readlane(src, jl / C);
					if (!(jl >= beg)) h1 = 0;
				} else {
					h1 = beg == 0 ? max(0, h0 - (A.o_del + A.e_del * (i + 1))) : 0;
				}
				if (!(gscore > h1)) max_ie = i;
				gscore = max(gscore, h1);
			}
			if (m == 0) break;
			const unsigned long long em = __ballot(lm == m);
			const int mj = __builtin_amdgcn_readlane(lmj, 63 - __builtin_clzll(em));
			if (m > mx) {
				mx = m; max_i = i; max_j = mj;
				max_off = max(max_off, abs(mj - i));
			} else if (A.zdrop > 0) {
				if (i - max_i > mj - max_j) {
					if (mx - m - ((i - max_i) - (mj - max_j)) * A.e_del > A.zdrop) break;
				} else {
					if (mx - m - ((mj - max_j) - (i - max_i)) * A.e_ins > A.zdrop) break;
				}
			}
			// next row's [beg,end): first / last non-zero of eh[beg..end] (ksw.c:963-970), where
			// eh[j] = {H(i,j-1), E(i+1,j)}, eh[beg].h = the first-column value, eh[end].e = 0
			const int h1i = beg == 0 ? max(0, h0 - (A.o_del + A.e_del * (i + 1))) : 0;
			const unsigned long long bh = __ballot(nzH), be = __ballot(nzE);
			int fidx = 1 << 20, lidx = -1;
			if (h1i) { fidx = beg; lidx = beg; }
			if (bh) {
				const int fl = __builtin_ctzll(bh), ll = 63 - __builtin_clzll(bh);
				fidx = min(fidx, __builtin_amdgcn_readlane(firstH, fl) + 1);
				lidx = max(lidx, __builtin_amdgcn_readlane(lastH, ll) + 1);
			}
			if (be) {
				const int fl = __builtin_ctzll(be), ll = 63 - __builtin_clzll(be);
				fidx = min(fidx, __builtin_amdgcn_readlane(firstE, fl));
				lidx = max(lidx, __builtin_amdgcn_readlane(lastE, ll));
			}
			const int nbeg = min(fidx, end);
			const int nend = min(qlen, max(lidx, nbeg - 1) + 2);
			beg = nbeg; end = nend;
		}
		if (lane == 0) {
			const int qle = max_j + 1, tle = max_i + 1, gtle = max_ie + 1;
			int32_t *o = A.out + 3 * (size_t)id;
			if (gscore <= 0 || gscore <= mx - A.end_bonus) { o[0] = mx; o[1] = qle; o[2] = tle; }
			else { o[0] = gscore; o[1] = qlen; o[2] = gtle; }
			if (A.raw) {
				int32_t *r = A.raw + 6 * (size_t)id;
				r[0] = mx; r[1] = qle; r[2] = tle; r[3] = gtle; r[4] = gscore; r[5] = max_off;
			}
		}
	}
}


// ------------------------------------------------------------------ 16-lane rows

typedef short short2_t __attribute__((ext_vector_type(2)));

// inclusive max-scan inside each 16-lane row
__device__ __forceinline__ int row_scan_max(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x111, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x112, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x114, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(NEG_INF, v, 0x118, 0xf, 0xf, false); v = max(v, t);
	return v;
}
// lane-1 inside the row (row_shr:1); lane 0 of each row receives `fill`
__device__ __forceinline__ int row_shr1(int v, int fill)
{
	return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false);
}
// all-reduce max over each 16-lane row: xor-1, xor-2 (quad_perm), row_half_mirror, row_mirror
__device__ __forceinline__ int row_allmax(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
	return v;
}
__device__ __forceinline__ int pk_max(int a, int b)
{
	short2_t x = __builtin_bit_cast(short2_t, a), y = __builtin_bit_cast(short2_t, b);
	return __builtin_bit_cast(int, __builtin_elementwise_max(x, y));
}
// same butterfly on two packed signed 16-bit values
__device__ __forceinline__ int row_allmax_pk(int v)
{
	v = pk_max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
	v = pk_max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
	v = pk_max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
	v = pk_max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
	return v;
}

// fused DPP forms: one VALU op per step (dst = max(dst, dpp(dst))), wait states for the VALU-write ->
// DPP-read hazard inside the string (hipcc does not pad inline asm)
__device__ __forceinline__ int row_allmax_f(int v)
{
	asm volatile("s_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "+v"(v));
	return v;
}
__device__ __forceinline__ int row_scan_max_f(int v)     // inclusive; lanes without a source keep their value
{
	asm volatile("s_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "+v"(v));
	return v;
}

// three independent all-reduce max butterflies over each 16-lane row, interleaved so that the VALU-write -> DPP-read
// hazard of one string is covered by the two others (no wait states inside)
__device__ __forceinline__ void row_allmax3_f(int &x, int &y, int &z)
{
	asm volatile("s_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_i32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "+v"(x), "+v"(y), "+v"(z));
}

// Substitution scores as a bit-field table (LUT form, |a|,|b| < 32): five 6-bit signed fields, one per query code;
// a row's table is the all-mismatch table with the field of its target base flipped to +a (all -1 when the target
// base is N), so the score of a cell is ONE v_bfe_i32 with the column's precomputed shift.
struct ext_lut_t { int base, flip, alln; };
__device__ __forceinline__ ext_lut_t ext_lut(const ext_args_t &A)
{
	ext_lut_t L;
	const int mb = (-A.b) & 63;
	L.base = mb | (mb << 6) | (mb << 12) | (mb << 18) | (63 << 24);
	L.flip = (mb ^ A.a) & 63;
	L.alln = 63 | (63 << 6) | (63 << 12) | (63 << 18) | (63 << 24);
	return L;
}

// per-alignment control state, replicated over the 16 lanes of the row
struct ext_rs_t { int beg, end, mx, max_i, max_j, max_ie, gscore, max_off; };

// One DP row of up to four alignments (one per 16-lane row).  Cells left of `beg` need no mask: their inputs are
// zero (that is why beg moved past them) and zero inputs give zero outputs; cells at or right of `end` are masked
// (M = 0, which with their stored E = 0 keeps E at 0, and H = 0 whatever F flows in from the left).
// F is carried unclamped: max(F,0) is what the reference holds, and H = max(M,E,F) with E >= 0 absorbs the clamp.
// Returns the new `alive`.
template <int C, bool LUT>
__device__ __forceinline__ bool ext_row(const ext_args_t &A, const ext_lut_t &L, const int oe_del, const int oe_ins,
                                        int (&H)[C], int (&E)[C], const int (&qv)[C], const int (&mmv)[C],
                                        const int ti, const bool run, const int hfc, const int hnx, const int i, const int qlen,
                                        const int j0, const int ej0, const int jl_lane, ext_rs_t &S, bool alive, const bool bound, const int rl)
{
	const int wend = run ? S.end - j0 : 0;                      // cells c < wend of this lane are left of `end`
	const int left = row_shr1(H[C - 1], S.beg == 0 ? hfc : 0);
	const bool tN = ti > 3;
	int tbl = 0;
	if (LUT) tbl = tN ? L.alln : (L.base ^ (L.flip << (6 * ti)));
	int M[C];
	int agg = NEG_INF;
	const int kc0 = ej0 - oe_ins;
#pragma unroll
	for (int c = 0; c < C; ++c) {
		const int hd = c == 0 ? left : H[c - 1];
		int sc;
		if (LUT) sc = __builtin_amdgcn_sbfe(tbl, (unsigned)qv[c], 6u);
		else sc = ti == qv[c] ? A.a : (tN ? -1 : mmv[c]);
		const int m = (c < wend && hd != 0) ? hd + sc : 0;          // masked at and right of `end`: E of those cells stays 0 by itself
		M[c] = m;
		agg = max(agg, m + kc0 + A.e_ins * c);                  // (M - oe_ins) + e_ins * j
	}
	int f = row_shr1(row_scan_max_f(agg), NEG_INF) - (ej0 - A.e_ins);      // F of this lane's first column
	int key = 0;                                                // (h << 16) | cell: row maximum, last column on ties
	int nzm = 0, nze = 0;                                       // bit (C-1-c): H != 0; bit 16 + (C-1-c) (C > 16: of nze): E != 0
	int hl = 0;                                                 // H of this lane's last cell left of `end`
#pragma unroll
	for (int c = 0; c < C; ++c) {
		const bool act = c < wend;
		const int hraw = max(max(M[c], E[c]), f);
		const int e = max(max(E[c] - A.e_del, M[c] - oe_del), 0);
		f = max(f - A.e_ins, M[c] - oe_ins);
		const int h = act ? hraw : 0;
		hl = act ? hraw : hl;
		H[c] = h; E[c] = e;
		key = max(key, (h << 16) | c);
		if (C <= 16) {
			const short2_t one = {1, 1};
			const int nz = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(short2_t, (e << 16) | h), one));
			nzm = (nzm << 1) | nz;
		} else {                                                // more than 16 cells per lane: one mask each
			nzm = (nzm << 1) | min(h, 1);
			nze = (nze << 1) | min(e, 1);
		}
	}
	key += j0;
	// E(i+1,j) != 0 implies H(i,j) != 0 (H >= E(i,j) and H >= M), so the non-zero span of eh[] follows from the non-zero H
	// columns alone: last index = last column + 1, first = first column (+1 if its E is 0)
	const unsigned hm = C <= 16 ? ((unsigned)nzm & 0xFFFFu) : (unsigned)nzm;
	const int p_hi = 31 - __clz((int)hm), p_lo = __ffs((int)hm) - 1;          // hm == 0: unused
	const int enz = C <= 16 ? ((nzm >> (16 + p_hi)) & 1) : ((nze >> (p_hi & 31)) & 1);
	int nfirst = hm ? -(j0 + (C - 1 - p_hi) + (enz ? 0 : 1)) : NEG_INF;
	int nlast = hm ? j0 + (C - 1 - p_lo) + 1 : -1;
	row_allmax3_f(key, nfirst, nlast);
	const int m = key >> 16, mj = key & 0xFFFF;
	// gscore: H(i, qlen-1) when the row reaches the query end (ksw.c:942-945)
	{
		int h1 = __builtin_amdgcn_ds_bpermute(jl_lane, hl);
		h1 = qlen == 0 ? (S.beg == 0 ? hnx : 0) : h1;
		const bool ge = run && S.end == qlen;
		S.max_ie = (ge && !(S.gscore > h1)) ? i : S.max_ie;
		S.gscore = ge ? max(S.gscore, h1) : S.gscore;
	}
	const bool upd = run && m != 0;
	alive = alive && !(run && m == 0);                          // ksw.c:946
	const bool better = upd && m > S.mx;
	S.max_off = better ? max(S.max_off, abs(mj - i)) : S.max_off;
	S.max_i = better ? i : S.max_i;
	S.max_j = better ? mj : S.max_j;
	if (A.zdrop > 0) {                                          // wave-uniform branch (ksw.c:951-959)
		const int di = i - S.max_i, dj = mj - S.max_j;
		const int pen = di > dj ? (di - dj) * A.e_del : (dj - di) * A.e_ins;
		alive = alive && !(upd && !better && S.mx - m - pen > A.zdrop);
	}
	S.mx = better ? m : S.mx;
	// next row's [beg,end) (ksw.c:963-970): first / last non-zero of eh[beg..end]
	{
		const int h1i = S.beg == 0 ? hnx : 0;
		const bool none = nfirst == NEG_INF;
		int fidx = none ? (1 << 20) : -nfirst;
		int lidx = none ? -1 : nlast;
		fidx = h1i ? min(fidx, S.beg) : fidx;
		lidx = h1i ? max(lidx, S.beg) : lidx;
		const int nbeg = min(fidx, S.end);
		const int nend = min(qlen, max(lidx, nbeg - 1) + 2);
		S.beg = upd ? nbeg : S.beg;
		S.end = upd ? nend : S.end;
	}
	// Exact early stop.  Phi(v at column c) = v + a*(qlen-1-c) never increases along a DP transition
	// (diagonal: +s <= +a and one column right; E: same column minus a gap cost; F: right minus a gap
	// cost), and only a NON-ZERO cell starts anything (M == 0 stays 0, ksw.c:924; E(i+1,j) != 0 implies H(i,j) != 0), so
	// every H of every later row is <= U = max Phi over the non-zero cells of this row's frontier {H(i,j), E(i+1,j),
	// first-column value}; a path also gains at most a per remaining target row, so U <= (row maximum) + a * rows left.
	// Once U <= max and U < gscore no later row can change max/max_i/max_j/max_off (strict >, ksw.c:948) nor
	// gscore/max_ie (>=, ksw.c:943): the remaining rows are dead work.  (Round 2 let zero cells carry a*(qlen-1-c): U never
	// fell below a*(qlen-1-beg) and the rule fired for near-perfect flanks only.)  Callers that take only the three
	// GASAL2 results (no raw 6-tuple) stop as well once the local-vs-to-end rule (src/bwamem.c:1893-1901) is decided:
	// U <= max - end_bonus and gscore <= max - end_bonus mean that no later gscore reaches max - end_bonus either.
	if (bound) {                                                // wave-uniform
		const int aq = A.a * (qlen - 1 - j0);
		int u = 0;
#pragma unroll
		for (int c = 0; c < C; ++c) { const int v = max(H[c], E[c]); u = max(u, v ? v + (aq - A.a * c) : 0); }
		const int h1n = S.beg == 0 ? hnx : 0;
		u = max(u, h1n ? h1n + A.a * qlen : 0);
		u = row_allmax_f(u);
		u = min(u, max(m, h1n) + A.a * rl);
		const bool fin = u < S.gscore || (A.raw == nullptr && u <= S.mx - A.end_bonus && S.gscore <= S.mx - A.end_bonus);
		alive = alive && !(u <= S.mx && fin);
	}
	return alive;
}

// column state of a new alignment; query codes above 3 are N (code 4); the target's N is 5 so that N never "matches"
// (mat[4][4] = -1, bwa.c:99-108)
template <int C, bool LUT>
__device__ __forceinline__ void ext_cols_init(const ext_args_t &A, const job_src_t &src, const int j0, const int qlen, const int h0, const int oe_ins,
                                              int (&H)[C], int (&E)[C], int (&qv)[C], int (&mmv)[C])
{
#pragma unroll
	for (int c = 0; c < C; ++c) {
		const int j = j0 + c;
		int qb = j < qlen ? ext_q_at(A, src, j) : 4;
		qb = qb > 3 ? 4 : qb;
		qv[c] = LUT ? 6 * qb : qb;
		mmv[c] = qb > 3 ? -1 : -A.b;                             // mismatch score of this column (generic form)
		const int v = h0 - oe_ins - j * A.e_ins;
		H[c] = (j < qlen && v > 0) ? v : 0;
		E[c] = 0;
	}
}

// Two forms of the 16-lane-row kernel.  extend16_static_kernel: a wave takes four consecutive jobs of the sorted list
// and runs them side by side to the end of the longest -- one set of loads per four jobs, which suits the short
// alignments of the small classes (few rows per job, the fetch latency of a job is a large part of it).
// extend16_kernel: every row draws its next job from a counter as soon as its alignment ends -- no waiting for the
// slowest of four (3.0 -> 3.8 live alignments per wave-row), which pays once a job is long enough to hide the draw.
// LUT: substitution scores from the bit-field table (|a|,|b| < 32); the generic form compares and selects.
template <int C, bool LUT>
__global__ void __launch_bounds__(256) extend16_static_kernel(ext_args_t A)
{
	const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
	const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t n = A.count[0];
	const uint32_t *ids = A.ids + A.count[1];
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	const int j0 = l16 * C;                                    // first column of this lane
	const int ej0 = A.e_ins * j0;
	const int bp_base = (lane & 48) << 2;                      // byte address of this row's lane 0 for ds_bpermute
	const ext_lut_t L = ext_lut(A);
	for (uint32_t w = wave * 4; w < n; w += n_waves * 4) {
		const bool have = w + grp < n;
		const uint32_t id = have ? ids[w + grp] : 0;
		const int qlen = have ? (int)A.qlen[id] : 0, tlen = have ? (int)A.tlen[id] : 0, h0 = have ? (int)A.h0[id] : 1;
		const job_src_t src = ext_job_src(A, id, have, qlen, tlen);
		int H[C], E[C], qv[C], mmv[C];
		ext_cols_init<C, LUT>(A, src, j0, qlen, h0, oe_ins, H, E, qv, mmv);
		const int jl = qlen - 1, jl_lane = bp_base + (((jl < 0 ? 0 : jl) / C) << 2);
		ext_rs_t S = {0, qlen, h0, -1, -1, -1, -1, 0};
		bool alive = have;
		int tchunk = 5;
		int rows_done = 0, wave_rows = 0;
		for (int i = 0; __any(alive && i < tlen); ++i) {
			++wave_rows;
			if ((i & 15) == 0) { const int tb = (alive && i + l16 < tlen) ? ext_t_at(A, src, i + l16) : 5; tchunk = tb > 3 ? 5 : tb; }
			const int ti = __builtin_amdgcn_ds_bpermute(bp_base + ((i & 15) << 2), tchunk);
			const bool run = alive && i < tlen;
			rows_done += run ? 1 : 0;
			const int hfc = max(0, h0 - (i == 0 ? 0 : A.o_del + A.e_del * i));        // H(i-1,-1)
			const int hnx = max(0, h0 - (A.o_del + A.e_del * (i + 1)));
			alive = ext_row<C, LUT>(A, L, oe_del, oe_ins, H, E, qv, mmv, ti, run, hfc, hnx, i, qlen, j0, ej0, jl_lane, S, alive, (i & 3) == 3, tlen - 1 - i);
		}
		if (have && l16 == 0) {
			const int qle = S.max_j + 1, tle = S.max_i + 1, gtle = S.max_ie + 1;
			int32_t *o = A.out + 3 * (size_t)id;
			if (S.gscore <= 0 || S.gscore <= S.mx - A.end_bonus) { o[0] = S.mx; o[1] = qle; o[2] = tle; }
			else { o[0] = S.gscore; o[1] = qlen; o[2] = gtle; }
			if (A.raw) {
				int32_t *r = A.raw + 6 * (size_t)id;
				r[0] = S.mx; r[1] = qle; r[2] = tle; r[3] = gtle; r[4] = S.gscore; r[5] = S.max_off;
			}
			if (A.stats) { atomicAdd(A.stats, (unsigned long long)rows_done); atomicAdd(A.stats + 1, (unsigned long long)tlen); atomicAdd(A.stats + 2, 1ull); if (grp == 0) atomicAdd(A.stats + 3, (unsigned long long)wave_rows); }
		}
	}
}


template <int C, bool LUT>
#ifndef EXT_MIN_WAVES
#define EXT_MIN_WAVES 1
#endif
__global__ void __launch_bounds__(256, EXT_MIN_WAVES) extend16_kernel(ext_args_t A)
{
	const int lane = threadIdx.x & 63, l16 = lane & 15;
	const uint32_t n = A.count[0];
	const uint32_t *ids = A.ids + A.count[1];
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	const int j0 = l16 * C;                                    // first column of this lane
	const int ej0 = A.e_ins * j0;
	const int bp_base = (lane & 48) << 2;                      // byte address of this row's lane 0 for ds_bpermute
	const ext_lut_t L = ext_lut(A);
	// Each 16-lane row works on its own alignment and row index; a row whose alignment has ended takes the next job
	// of the class from a global counter (longest first), so the four rows of a wave never wait for the slowest.
	// the target of a row's alignment is staged in LDS when the row takes the job (EXT_T_CAP bases; longer targets
	// never reach this kernel: ext_bin_kernel), so the row loop itself has no global loads to wait for
	__shared__ uint8_t t_lds[16][EXT_T_CAP];
	uint8_t *tl = t_lds[threadIdx.x >> 4];
	bool have = false, alive = false;
	uint32_t id = 0;
	int qlen = 0, tlen = 0, h0 = 1, i = 0;
	int hfc = 0, dn = 0;          // first-column value H(i-1,-1) of the current row; o_del + e_del*(i+1)
	const uint8_t *tp = tl;
	job_src_t src = ext_job_src(A, 0, false, 0, 0);
	int H[C], E[C], qv[C], mmv[C];
#pragma unroll
	for (int c = 0; c < C; ++c) { H[c] = E[c] = 0; qv[c] = LUT ? 24 : 4; mmv[c] = -1; }
	int jl_lane = bp_base;
	ext_rs_t S = {0, 0, 0, -1, -1, -1, -1, 0};
	int rows_done = 0, wave_rows = 0;
	bool more = true, more_g = true;                           // jobs left for this wave / on the class counter (wave-uniform)
	uint32_t qn = 0, qe = 0;
	for (;;) {
		// one bit per row that has something to do here (results to write and / or a job to draw), at lanes 0/16/32/48
		const unsigned long long nb = __ballot(!alive && l16 == 0 && (more || have));
		if (nb) {
			if (!alive && have && l16 == 0) {                   // results of the alignment that just ended
				const int qle = S.max_j + 1, tle = S.max_i + 1, gtle = S.max_ie + 1;
				int32_t *o = A.out + 3 * (size_t)id;
				if (S.gscore <= 0 || S.gscore <= S.mx - A.end_bonus) { o[0] = S.mx; o[1] = qle; o[2] = tle; }
				else { o[0] = S.gscore; o[1] = qlen; o[2] = gtle; }
				if (A.raw) {
					int32_t *r = A.raw + 6 * (size_t)id;
					r[0] = S.mx; r[1] = qle; r[2] = tle; r[3] = gtle; r[4] = S.gscore; r[5] = S.max_off;
				}
				if (A.stats) { atomicAdd(A.stats, (unsigned long long)rows_done); atomicAdd(A.stats + 1, (unsigned long long)tlen); atomicAdd(A.stats + 2, 1ull); }
			}
			// (wave-local job range refilled (EXT_DRAW_CHUNK / 4) at a time, see extpk_kernel)
			if (qn == qe && more_g) {
				uint32_t b0 = 0;
				if (lane == 0) b0 = atomicAdd(A.ctr, (uint32_t)(EXT_DRAW_CHUNK / 4));
				qn = __builtin_amdgcn_readfirstlane(b0);
				qe = qn + (EXT_DRAW_CHUNK / 4) < n ? qn + (EXT_DRAW_CHUNK / 4) : n;
				if (qn >= n) { qn = qe = n; }
				more_g = qe < n;
			}
			const uint32_t base = qn;
			{
				const uint32_t cnt = (uint32_t)__builtin_popcountll(nb), avail = qe - qn;
				qn += cnt < avail ? cnt : avail;
			}
			more = more_g || qn < qe;
			if (!alive) {
				const uint32_t k = base + (uint32_t)__builtin_popcountll(nb & ((1ull << (lane & 48)) - 1));
				have = k < qe;
				id = have ? ids[n - 1 - k] : 0;
				qlen = have ? (int)A.qlen[id] : 0; tlen = have ? (int)A.tlen[id] : 0; h0 = have ? (int)A.h0[id] : 1;
				src = ext_job_src(A, id, have, qlen, tlen);
				ext_cols_init<C, LUT>(A, src, j0, qlen, h0, oe_ins, H, E, qv, mmv);
				const int jl = qlen - 1;
				jl_lane = bp_base + (((jl < 0 ? 0 : jl) / C) << 2);
				S.beg = 0; S.end = qlen; S.mx = h0; S.max_i = -1; S.max_j = -1; S.max_ie = -1; S.gscore = -1; S.max_off = 0;
				i = 0; rows_done = 0; hfc = h0; dn = oe_del; tp = tl;
				for (int k = l16; k < tlen && k < EXT_T_CAP; k += 16) { const int tb = ext_t_at(A, src, k); tl[k] = (uint8_t)(tb > 3 ? 5 : tb); }
				alive = have && tlen > 0;
				if (have && tlen == 0) {
					// a job without target rows (its window was clipped away at a sequence end) is answered on the spot: the
					// loop below may end before this row comes back to the result writer above
					if (l16 == 0) {
						int32_t *o = A.out + 3 * (size_t)id;
						o[0] = h0; o[1] = 0; o[2] = 0;                   // ksw_extend2 with no rows: max = h0, nothing consumed, gscore -1
						if (A.raw) { int32_t *r = A.raw + 6 * (size_t)id; r[0] = h0; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = -1; r[5] = 0; }
						if (A.stats) atomicAdd(A.stats + 2, 1ull);
					}
					have = false;
				}
			}
		}
		if (!__any(alive)) {
			if (!more) break;
			continue;                                           // every row drew a zero-row job: draw again
		}
		++wave_rows;
		const int ti = (int)*tp;
		const bool run = alive;
		rows_done += run ? 1 : 0;
		const int hnx = max(0, h0 - dn);                        // first-column value of the next row: max(0, h0 - (o_del + e_del*(i+1)))
		// (the early-stop bound is evaluated every fourth iteration of the wave for all its rows: it is exact at any row)
		alive = ext_row<C, LUT>(A, L, oe_del, oe_ins, H, E, qv, mmv, ti, run, hfc, hnx, i, qlen, j0, ej0, jl_lane, S, alive, (wave_rows & 3) == 0, tlen - 1 - i);
		if (run) { ++i; ++tp; hfc = hnx; dn += A.e_del; }
		alive = alive && i < tlen;
	}
	if (A.stats && lane == 0) atomicAdd(A.stats + 3, (unsigned long long)wave_rows);
}


// classes: 0 = unsupported length (query longer than 768 bases: all three outputs INT32_MIN, counted, see
// bmh_extend_last_unsupported); 1..18 = extend16_kernel<C>; 19..26 = extend_wide_kernel<5..12>
#define EXT_WIDE_MAX_C 12
#define EXT_N_CLS 49
#define EXT_DONE_CLS 27     // decided by the closed-form prefilter: no DP
#define EXT16_MAX_C 18
// 28..34 = extpk_kernel<4, P>, P = 4, 6, .. 16 (queries up to 8 P columns); 35..36 = extpk_kernel<8, P>, P = 9, 10 (up to 16 P); 37..39 = extpk_kernel<4, P>, P = 24, 28, 32
// (up to 8 P; extpk_kernel<8, P>, P = 12, 14, 16 in builds with PK_WIDE4 0 and in the persistent kernel); 40 = extpk_kernel<8, 18> (up to 288; extpk_kernel<16, 9> in builds with PK_WIDE18 0 and in the persistent kernel)
#define EXT_PK_BASE 28
#define EXT_PK_MAXQ 288

// (G = 4, P = 17: queries of 129..136 columns -- the flank of a 150 bp read whose seed sits at its very end -- on four lanes
// instead of eight: 26 instead of 34 wave-instructions per alignment row, for 15 % of the extension's time on 150 bp reads)
#define EXT_PK17_CLS (EXT_PK_BASE + 13)
// 42..48 = extpk_kernel<2, P>, P = 8, 12, .. 32 (queries up to 4 P columns, targets up to 8 P rows)
#define EXT_PK2_BASE 42
constexpr int ext_pk_cls_of(int G, int P) { return G == 2 ? EXT_PK2_BASE + P / 4 - 2 : G == 4 && P == 17 ? EXT_PK17_CLS : G == 4 && P > 18 ? EXT_PK_BASE + P / 4 + 3 : EXT_PK_BASE + (G == 4 ? P / 2 - 2 : G == 16 ? 12 : P == 9 ? 7 : P == 10 ? 8 : P / 2 + 3); }      // (G = 4, P = 24 / 28 / 32: the places of G = 8, P = 12 / 14 / 16)

#include "extpk_dev.h"

// ------------------------------------------------------------------ closed-form prefilter

// packed 16-bit class of a query length (0: none)
__device__ __forceinline__ int ext_pk_class(uint32_t ql)
{
	if (ql > EXT_PK_MAXQ) return 0;
	if (ql <= 128) { const int h = (int)((ql + 15) / 16); return EXT_PK_BASE + (h < 2 ? 2 : h) - 2; }       // P = 2 h
	return EXT_PK_BASE + (ql <= 144 ? 7 : ql <= 160 ? 8 : ql <= 192 ? 9 : ql <= 224 ? 10 : ql <= 256 ? 11 : 12);
}
__device__ __forceinline__ int ext_class(uint32_t ql)
{
	if (ql <= 16 * EXT16_MAX_C) return ql <= 16 ? 1 : (int)((ql + 15) / 16);
	const int wc = (int)((ql + 63) / 64);
	return wc <= EXT_WIDE_MAX_C ? 19 + (wc - 5) : 0;
}
// The jobs of a batch are grouped by class and, inside a class, by target length in steps of 32 rows (the class kernels draw from
// the long end, so that their tails are short jobs): a counting sort over EXT_N_BINS = classes x 16 bins -- the bin of every job and
// the histogram come out of the prefilter kernel below, then offsets and a scatter -- instead of a radix sort of (class << 20 | tlen)
// keys behind a key kernel (seven launches, 0.95 ms of every extension pass for 2 M jobs).  The order inside a bin is whatever the
// blocks' atomics make it; every job is independent.
#define EXT_TL_BINS 16
#define EXT_N_BINS (EXT_N_CLS * EXT_TL_BINS)
// class of a job the prefilter has not decided (pk_a > 0: packed 16-bit kernels allowed, match score)
__device__ __forceinline__ int ext_route(uint32_t ql, uint32_t tl, uint32_t h0, int pk_a, const uint32_t g2)
{
	int cls = ext_class(ql);
	if (cls >= 1 && cls <= EXT16_MAX_C && tl > EXT_T_CAP) cls = 19;     // very long target: wide kernel (streams it)
	if (pk_a > 0 && h0 + ql * (uint32_t)pk_a < PK_HMAX) {
		const int pc = ext_pk_class(ql);
		if (pc && tl <= (uint32_t)(ql <= 128 ? PK_TCAP(4) : ql <= 256 ? PK_TCAP(8) : PK_TCAP(16)) && (!PK_WIDE18 || ql <= 256 || h0 + ql * (uint32_t)pk_a < PK_HMAX17) && (!PK_WIDE4 || ql <= 160 || ql > 256 || h0 + ql * (uint32_t)pk_a < PK_HMAX17)) cls = pc;
		if (pc && ql > 128 && ql <= 136 && tl <= (uint32_t)PK_TCAP(4) && h0 + ql * (uint32_t)pk_a < PK_HMAX17) cls = EXT_PK17_CLS;
#if PK_G2
		if (g2 && ql <= 128 && ql > 16u * (g2 - 1u)) {                                            // two lanes of 4 h pairs where the target fits the class's rows
			const uint32_t h = ql <= 32 ? 2u : (ql + 15u) / 16u;
			if (tl <= 32u * h && (h <= 4u || h0 + ql * (uint32_t)pk_a < PK_HMAX17)) cls = EXT_PK2_BASE + (int)h - 2;
		}
#endif
	}
	return cls;
}


__device__ __forceinline__ int row_allsum_f(int v)
{
	asm volatile("s_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "+v"(v));
	return v;
}
__device__ __forceinline__ int row_allor_f(int v)
{
	asm volatile("s_nop 1\n\t"
	             "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_or_b32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_or_b32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "+v"(v));
	return v;
}

// Closed form for flanks whose main diagonal has at most TWO substitutions and that need no gap.
// H(i,j) is bounded by the best score of a path from the origin (value h0) made of diagonal steps (+a / -b)
// and gaps (o + L*e); the zero floor, the "M == 0 stays 0" rule and the [beg,end) trimming only lower values,
// and a positive diagonal cell is always inside [beg,end) (eh[i] is non-zero in row i-1).  Write
// delta = a + b, D(i) for the value of the gap-free path at (i,i), and let the main diagonal mismatch only at
// columns p1 < p2 (no N anywhere, tlen >= qlen).
//  * A path with two or more gaps loses >= 2*min(oe) against the gap-free path and can win back at most
//    2*delta: with delta < min(oe) it stays strictly below every comparison target.
//  * A path with ONE gap of length L that leaves the diagonal after (r,r) reaches its end cell with at most as
//    many diagonal steps as the gap-free path to the comparison target; it ties or beats the target only if
//    delta*(k' - m_alt) >= o + e*L + a*(lost steps), k' = main mismatches it bypasses, m_alt = mismatches on its
//    own steps.  With one bypassed mismatch (k' <= 1) that is impossible (delta < o + e).  With k' = 2 it needs
//    m_alt = 0 and L <= Lmax = (2*delta - o)/e: the path leaves before p1, runs on the diagonal shifted by
//    d = +-L, |d| <= Lmax, and ALL its steps match, in particular those in rows [p1 + Lmax, p2].
//  * gscore compares H(i, qlen-1) of EVERY row with D(qlen-1): a cell above the diagonal in that column ends a
//    right-shifted diagonal early (it loses L steps: a*L more), so only L <= Lg = max{L: o + (e+a)L < 2*delta} can
//    beat D(qlen-1), and only if its steps up to row qlen-1-L all match.
// Hence: with <= 1 mismatch the diagonal is the strict maximum of every row; with 2 mismatches the same holds
// as soon as every shifted diagonal |d| <= Lmax has a mismatch somewhere in rows [p1 + Lmax, p2] (for right
// shifts: before the diagonal leaves the query) -- eight short byte comparisons for the default scoring, which random sequence passes and
// tandem repeats fail (those go to the DP).  The targets are D(i) for cells of row i < qlen, and D(qlen-1) for
// the rows below the diagonal and for the gscore column, so max/max_i/max_j follow the gap-free values
// (strict > updates: the end of each mismatch-free segment that exceeds all earlier ones), max_off = 0,
// gscore = D(qlen-1), max_ie = qlen-1.  z-drop (ksw.c:951-959) is handled by requiring the draw-down at the
// mismatch rows to stay within zdrop.  A flank starts right after a maximal exact match, so its first base is
// usually the mismatch that ended the seed; most 150 bp flanks at 1 % error have at most one more.
// Eight lanes per job, eight columns per lane and step (the target codes of eight rows decoded at once from the 2-bit text, the query
// bytes decoded four to a register; one column per lane and step, on sixteen lanes, this prefilter took 15 % of the stage);
// jobs decided here get done[id] = 1 and never reach the DP kernels.
__device__ __forceinline__ int grp8_allsum(int v)
{
	asm volatile("s_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "+v"(v));
	return v;
}
__device__ __forceinline__ int grp8_allor(int v)
{
	asm volatile("s_nop 1\n\t"
	             "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_or_b32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
	             : "+v"(v));
	return v;
}
// query codes of columns j0..j0+7 as two dwords (byte u of `lo` = column j0+u; columns at and beyond qlen: code 0); bad: one of the
// columns below qlen is not A, C, G or T (descriptor jobs decode the ASCII reads: code = t ^ t >> 1 of t = bits 2:1 of the letter,
// checked by looking the letter up again) -- array jobs carry their codes, anything above 3 shows in the bytes themselves
__device__ __forceinline__ void pk_q8(const ext_args_t &A, const job_src_t &s, const int j0, const int qlen, uint32_t &lo, uint32_t &hi, bool &bad)
{
	uint32_t w[2] = {0, 0};
	if (j0 + 8 <= qlen) {                 // all eight columns exist: two unaligned dword loads, byte-swapped when the job runs backwards
		const uint8_t *p0 = s.qp + (long)j0 * s.qstep;
		if (s.qstep > 0) { __builtin_memcpy(&w[0], p0, 4); __builtin_memcpy(&w[1], p0 + 4, 4); }
		else { uint32_t a, b; __builtin_memcpy(&a, p0 - 3, 4); __builtin_memcpy(&b, p0 - 7, 4); w[0] = __builtin_bswap32(a); w[1] = __builtin_bswap32(b); }
	} else {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int j = j0 + u;
			const uint32_t b = j < qlen ? (uint32_t)s.qp[(long)j * s.qstep] : (A.desc ? 0x41u : 0u);
			w[u >> 2] |= b << (8 * (u & 3));
		}
	}
	bad = false;
	if (A.desc) {
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint32_t W = w[k] & 0xDFDFDFDFu, t = (W >> 1) & 0x03030303u, code = t ^ ((t >> 1) & 0x01010101u);
			bad = bad || __builtin_amdgcn_perm(0u, 0x54474341u, code) != W;
			w[k] = code;
		}
	}
	lo = w[0]; hi = w[1];
}
__global__ void __launch_bounds__(256) ext_closed_form_kernel(ext_args_t A, uint32_t n, uint32_t *__restrict__ bin_of, uint32_t *__restrict__ bin_cnt, int pk_a, int g2)
{
	wtrace_scope_t wt_(WT_EXT_CLOSED);
	__shared__ uint32_t hist[EXT_N_BINS];
	for (int k = threadIdx.x; k < EXT_N_BINS; k += 256) hist[k] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63, l8 = lane & 7, grp = lane >> 3;
	const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	const int min_oe = oe_del < oe_ins ? oe_del : oe_ins;
	const int min_o = A.o_del < A.o_ins ? A.o_del : A.o_ins, min_e = A.e_del < A.e_ins ? A.e_del : A.e_ins;
	const int delta = A.a + A.b;
	// longest single gap that two bypassed mismatches could pay for (ties count); -1: two-mismatch form disabled
	const int lmax = (min_e > 0 && 2 * delta >= min_o) ? (2 * delta - min_o) / min_e : (2 * delta < min_o + min_e ? 0 : -1);
	const bool two_ok = lmax >= 0 && lmax <= 6;
	// longest insertion whose end cell in the gscore column can still beat D(qlen-1): o + (e + a)*L < 2*delta
	const int lg = (2 * delta - min_o - 1 >= 0 && min_e + A.a > 0) ? (2 * delta - min_o - 1) / (min_e + A.a) : 0;
	const bool params_ok = A.a > 0 && delta < min_oe && (A.zdrop <= 0 || A.b <= A.zdrop);
	// the codes of the compared columns are kept in LDS (one row pair per job): the shifted-diagonal test of the two-mismatch
	// form re-reads up to seven of them per row, which as fresh fetches (address arithmetic, ASCII / 2-bit decoding) was most
	// of this kernel's instructions
	__shared__ __attribute__((aligned(16))) uint8_t q_lds[32][512], t_lds[32][512];
	uint8_t *qs = q_lds[threadIdx.x >> 3], *ts = t_lds[threadIdx.x >> 3];
	for (uint32_t w = wave * 8; w < n; w += n_waves * 8) {
		const uint32_t id = w + grp;
		const bool have = id < n;
		const int qlen = have ? (int)A.qlen[id] : 0, tlen = have ? (int)A.tlen[id] : 0, h0 = have ? (int)A.h0[id] : 1;
		const job_src_t src = ext_job_src(A, id, have, qlen, tlen);
		const bool elig = have && qlen > 0 && qlen <= 512 && tlen >= qlen && params_ok;     // (512: the LDS rows; longer queries have no DP class either)
		int hi = -1, lo = -0x7000, cnt = 0;   // largest / (negated) smallest mismatching column, mismatch count of this lane
		for (int j0 = 8 * l8; __any(elig && j0 < qlen); j0 += 64) {
			if (elig && j0 < qlen) {
				uint32_t tlo, thi, qlo, qhi; bool bad;
				pk_t8(A, src, j0, tlen, tlo, thi);
				pk_q8(A, src, j0, qlen, qlo, qhi, bad);
				*(uint2 *)(qs + j0) = make_uint2(qlo, qhi); *(uint2 *)(ts + j0) = make_uint2(tlo, thi);
				const int nv = qlen - j0 < 8 ? qlen - j0 : 8;                  // columns of this step below qlen
				const unsigned long long vm = nv == 8 ? ~0ull : (1ull << (8 * nv)) - 1ull;
				const unsigned long long q64 = (unsigned long long)qhi << 32 | qlo, t64 = (unsigned long long)thi << 32 | tlo;
				bad = bad || ((q64 | t64) & 0xFCFCFCFCFCFCFCFCull & vm) != 0ull;      // N on either side: not eligible
				unsigned long long x = (q64 ^ t64) & vm;
				x = (x | x >> 1 | x >> 2) & 0x0101010101010101ull;            // one bit per mismatching column
				if (x) { hi = max(hi, j0 + ((63 - (int)__builtin_clzll(x)) >> 3)); lo = max(lo, -(j0 + ((int)__builtin_ctzll(x) >> 3))); }
				cnt += (int)__builtin_popcountll(x) + (bad ? 1 : 0);
				hi = bad ? 0x7000 : hi; lo = bad ? 0 : lo;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");           // the codes in LDS before other lanes read them below
		hi = grp_allmax<8>(hi); lo = grp_allmax<8>(lo); cnt = grp8_allsum(cnt);
		const bool clean = hi < 0x7000;
		const int p1 = -lo, p2 = hi;
		// two mismatches: every diagonal shifted by 0 < |d| <= lmax needs a mismatch in rows [p1+lmax, p2].  A diagonal
		// shifted to the right (d = +L, reached by an insertion) ends at row qlen-1-L in the gscore column: if it ends
		// before row p2 it cannot beat its own row's diagonal cell (only one mismatch bypassed), but for L <= lg it can
		// still beat D(qlen-1) -- the gscore comparison of ksw.c:942-945 -- so there the mismatch must come before its end.
		bool two = elig && clean && cnt == 2 && two_ok && p2 - p1 >= (lmax > 0 ? lmax : 1);
		if (__any(two) && lmax > 0) {
			int bits = 0;
			for (int i = p1 + lmax + l8; __any(two && i <= p2); i += 8) {
				if (two && i <= p2) {
					const int tb = (int)ts[i];
					for (int L = 1; L <= lmax; ++L) {
						const int cm = i - L, cp = i + L;
						const bool mm = cm < 0 || (int)qs[cm] != tb;                // d = -L
						const bool mp = cp < qlen && (int)qs[cp] != tb;             // d = +L: rows past the end of that diagonal do not count
						bits |= (mm ? 1 : 0) << (2 * (L - 1)) | (mp ? 1 : 0) << (2 * (L - 1) + 1);
					}
				}
			}
			bits = grp8_allor(bits);
			for (int L = 1; L <= lmax; ++L) {
				const bool minus_ok = (bits >> (2 * (L - 1))) & 1;
				const bool plus_ok = ((bits >> (2 * (L - 1) + 1)) & 1) || (L > lg && p2 > qlen - 1 - L);
				two = two && minus_ok && plus_ok;
			}
		}
		const bool none = elig && clean && cnt == 0, one = elig && clean && cnt == 1;
		// gap-free values: V1 = value before row p1, V2 = before row p2, V3 = D(qlen-1)
		const int pa = none ? qlen : p1;
		const int V1 = h0 + pa * A.a;
		const int V2 = (one || two) ? V1 - A.b + ((two ? p2 : qlen) - p1 - 1) * A.a : V1;      // one: this is already D(qlen-1)
		const int V3 = two ? V2 - A.b + (qlen - 1 - p2) * A.a : V2;
		bool ok = none || one || two;
		ok = ok && (none || V1 - A.b > 0) && (!two || V2 - A.b > 0);
		if (A.zdrop > 0 && two) ok = ok && (max(V1, V2) - V2 + A.b <= A.zdrop);
		if (have && l8 == 0) {
			{   // the job's bin: decided here (no DP), or its DP class by query / target length and score range
				const int cls = ok ? EXT_DONE_CLS : ext_route((uint32_t)qlen, (uint32_t)tlen, (uint32_t)h0, pk_a, (uint32_t)g2);
				if (cls == 0) A.out[3 * (size_t)id] = A.out[3 * (size_t)id + 1] = A.out[3 * (size_t)id + 2] = INT32_MIN;
				const uint32_t tb = (uint32_t)tlen >> 5;
				const uint32_t bin = (uint32_t)cls * EXT_TL_BINS + (tb < EXT_TL_BINS - 1 ? tb : EXT_TL_BINS - 1);
				bin_of[id] = bin;
				atomicAdd(&hist[bin], 1u);
			}
			if (ok) {
				// running maximum with strict updates: start (h0, -1); segment ends (V1, p1-1), (V2, p2-1) [two only], (V3, qlen-1)
				int mx = h0, mi = -1;
				if (pa > 0) { mx = V1; mi = pa - 1; }
				if (two && p2 - p1 - 1 > 0 && V2 > mx) { mx = V2; mi = p2 - 1; }
				const int lastseg = none ? 0 : qlen - 1 - (two ? p2 : p1);
				if (lastseg > 0 && V3 > mx) { mx = V3; mi = qlen - 1; }
				const int gscore = V3, qle = mi + 1, tle = mi + 1, gtle = qlen;
				int32_t *o = A.out + 3 * (size_t)id;
				if (gscore <= 0 || gscore <= mx - A.end_bonus) { o[0] = mx; o[1] = qle; o[2] = tle; }
				else { o[0] = gscore; o[1] = qlen; o[2] = gtle; }
				if (A.raw) {
					int32_t *r = A.raw + 6 * (size_t)id;
					r[0] = mx; r[1] = qle; r[2] = tle; r[3] = gtle; r[4] = gscore; r[5] = 0;
				}
			}
		}
	}
	__syncthreads();
	for (int k = threadIdx.x; k < EXT_N_BINS; k += 256) if (hist[k]) atomicAdd(&bin_cnt[k], hist[k]);
}

// ------------------------------------------------------------------ host side


// bin_cnt -> bin_base (start of every bin in the grouped list); counts[2c] = size of class c, counts[2c+1] = its offset
__global__ void ext_offsets_kernel(uint32_t *counts, const uint32_t *__restrict__ bin_cnt, uint32_t *__restrict__ bin_base)
{
	if (threadIdx.x == 0) {
		uint32_t acc = 0;
		for (int c = 0; c < EXT_N_CLS; ++c) {
			counts[2 * c + 1] = acc;
			for (int b = 0; b < EXT_TL_BINS; ++b) { bin_base[c * EXT_TL_BINS + b] = acc; acc += bin_cnt[c * EXT_TL_BINS + b]; }
			counts[2 * c] = acc - counts[2 * c + 1];
		}
	}
}

// every block takes a range of each of its bins with one atomic, its jobs their places in it with LDS atomics; beside its id every job
// leaves a 32-byte RECORD of what a kernel needs to start it (lengths, seed score, where its bases are) at its place in the grouped list:
// a wave that draws jobs reads them in one coalesced load per chunk instead of chasing ids[] -> qlen / tlen / h0 -> jq_src / jt0 / job_side
// through three dependent gathers per job (extpk_dev.h: what the packed kernels' waves waited for)
__global__ void __launch_bounds__(256) ext_scatter_kernel(ext_args_t A, const uint32_t *__restrict__ bin_of, uint32_t n, const uint32_t *__restrict__ bin_base,
                                                          uint32_t *__restrict__ bin_cur, uint32_t *__restrict__ ids, uint4 *__restrict__ recs)
{
	__shared__ uint32_t hist[EXT_N_BINS], base[EXT_N_BINS];
	for (int k = threadIdx.x; k < EXT_N_BINS; k += 256) hist[k] = 0;
	__syncthreads();
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t bin = 0, my = 0;
	if (t < n) { bin = bin_of[t]; my = atomicAdd(&hist[bin], 1u); }
	__syncthreads();
	for (int k = threadIdx.x; k < EXT_N_BINS; k += 256) if (hist[k]) base[k] = bin_base[k] + atomicAdd(&bin_cur[k], hist[k]);
	__syncthreads();
	if (t < n) {
		const uint32_t pos = base[bin] + my;
		ids[pos] = t;
		if (bin / EXT_TL_BINS == EXT_DONE_CLS) return;          // decided by the prefilter: nobody draws it
		const uint32_t ql = A.qlen[t], tl = A.tlen[t];
		uint4 r0, r1;
		r0.x = t; r0.y = (ql & 0xFFFFu) | ((tl < 0xFFFFu ? tl : 0xFFFFu) << 16); r0.z = A.h0[t];
		if (A.desc) { const long long t0 = A.jt0[t]; r0.w = A.jq_src[t]; r1.x = A.job_side[t]; r1.y = 0; r1.z = (uint32_t)t0; r1.w = (uint32_t)((unsigned long long)t0 >> 32); }
		else { r0.w = A.qoff[t]; r1.x = 1; r1.y = A.toff[t]; r1.z = r1.w = 0; }
		recs[2 * (size_t)pos] = r0; recs[2 * (size_t)pos + 1] = r1;
	}
}

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

// scratch for the sorted job list: one per (device, stream), grown on demand and reused across calls, so
// that batches in flight on different streams never share it
struct ext_scratch_t {
	uint32_t *keys, *vals2, *counts, *bins; uint4 *recs; size_t cap; int dev;      // keys: bin of every job; vals2: job ids grouped by bin; bins: [3][EXT_N_BINS] count / base / cursor
	hipEvent_t ev0, ev1; bool have_ev;
	hipStream_t side[4]; hipEvent_t fork, join[4];     // class kernels run concurrently on side streams
};
static std::mutex g_scr_mu;
static std::map<std::pair<int, void *>, ext_scratch_t *> g_scr_map;
static thread_local ext_scratch_t *g_last = nullptr;

static ext_scratch_t *scratch_for(int dev, void *stream)
{
	std::lock_guard<std::mutex> lk(g_scr_mu);
	auto key = std::make_pair(dev, stream);
	auto it = g_scr_map.find(key);
	if (it != g_scr_map.end()) return it->second;
	ext_scratch_t *s = (ext_scratch_t *)calloc(1, sizeof(ext_scratch_t));
	s->dev = dev;
	g_scr_map[key] = s;
	return s;
}

// room for n jobs; grown by a quarter beyond the request, so that batches of slowly growing size do not reallocate every time
// (hipFree waits for the whole device: a reallocation between the two extension passes of bmh_chain_extend_merge would serialise them)
static int scratch_reserve(ext_scratch_t &g_scr, int dev, size_t n)
{
	if (g_scr.cap >= n) return BMH_OK;
	const size_t c = n + n / 4 + 1024;
	void *ps[] = {g_scr.keys, g_scr.vals2, g_scr.counts, g_scr.bins, g_scr.recs};
	for (void *q : ps) if (q) (void)hipFree(q);
	g_scr.keys = g_scr.vals2 = g_scr.counts = g_scr.bins = nullptr; g_scr.recs = nullptr; g_scr.cap = 0;
	HIPCK(hipMalloc((void **)&g_scr.keys, 4 * c)); HIPCK(hipMalloc((void **)&g_scr.vals2, 4 * c));
	HIPCK(hipMalloc((void **)&g_scr.recs, 32 * c));
	HIPCK(hipMalloc((void **)&g_scr.counts, 4 * 3 * EXT_N_CLS));
	HIPCK(hipMalloc((void **)&g_scr.bins, 4 * 3 * EXT_N_BINS));
	g_scr.cap = c; g_scr.dev = dev;
	return BMH_OK;
}
// (internal, bmh_internal.h) the scratch of (current device, stream) sized for batches of up to n jobs before any of them is launched
int bmh_extend_reserve(void *stream_, uint64_t n)
{
	int dev = 0;
	HIPCK(hipGetDevice(&dev));
	return scratch_reserve(*scratch_for(dev, stream_), dev, (size_t)n);
}

// Frees the scratch (sorted job list, side streams, events) that bmh_extend_batch keeps per (device, stream); call it before
// destroying a stream that was used for extensions (a recycled stream handle would otherwise inherit stale scratch).  The
// stream must be idle.  A stream must not be used for extensions by two host threads at once: they would share this scratch.
extern "C" void bmh_finalize_release(void *stream_);
extern "C" void bmh_matesw_release(void *stream_);
extern "C" void bmh_cigar_release(void *stream_);
extern "C" void bmh_extend_release(void *stream_)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return;
	bmh_finalize_release(stream_); bmh_matesw_release(stream_); bmh_cigar_release(stream_);      // the per-stream scratch of the stages after the extension goes with it
	ext_scratch_t *s = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_scr_mu);
		auto it = g_scr_map.find(std::make_pair(dev, stream_));
		if (it == g_scr_map.end()) return;
		s = it->second;
		g_scr_map.erase(it);
	}
	if (g_last == s) g_last = nullptr;
	void *ps[] = {s->keys, s->vals2, s->counts, s->bins, s->recs};
	for (void *q : ps) if (q) (void)hipFree(q);
	if (s->have_ev) {
		(void)hipEventDestroy(s->ev0); (void)hipEventDestroy(s->ev1); (void)hipEventDestroy(s->fork);
		for (int i = 0; i < 4; ++i) { (void)hipStreamDestroy(s->side[i]); (void)hipEventDestroy(s->join[i]); }
	}
	free(s);
}

// device time of the last bmh_extend_batch issued by this thread (HIP events on its stream)
extern "C" float bmh_extend_last_ms(void)
{
	if (!g_last || !g_last->have_ev) return -1.f;
	float ms = -1.f;
	if (hipEventSynchronize(g_last->ev1) != hipSuccess) return -1.f;
	if (hipEventElapsedTime(&ms, g_last->ev0, g_last->ev1) != hipSuccess) return -1.f;
	return ms;
}

// number of jobs of the thread's last bmh_extend_batch whose query was longer than the kernels support (768 bases); their
// outputs are INT32_MIN.  Waits for the batch.
extern "C" int64_t bmh_extend_last_unsupported(void)
{
	if (!g_last || !g_last->have_ev) return 0;
	uint32_t n0 = 0;
	if (hipEventSynchronize(g_last->ev1) != hipSuccess) return -1;
	if (hipMemcpy(&n0, g_last->counts, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (int64_t)n0;
}

// Occupancy cap for co-scheduling: a block that reserves `g_ext_lds` bytes of (unused) dynamic LDS limits the DP
// kernels to 160 KiB / g_ext_lds blocks per CU, which leaves wave slots for the gather-bound seeding kernels of
// another stream to become resident beside them (BMH_EXT_LDS_KB; 0 = no cap).
static unsigned g_ext_lds = [] { const char *e = getenv("BMH_EXT_LDS_KB"); return e ? (unsigned)atoi(e) * 1024u : 0u; }();

// packed 16-bit kernels for the jobs that qualify (bmh_extend_set_packed; BMH_EXT_PACKED=0 starts with them off)
static int g_ext_packed = [] { const char *e = getenv("BMH_EXT_PACKED"); return e ? atoi(e) : 1; }();
extern "C" int bmh_extend_set_packed(int on) { const int was = g_ext_packed; g_ext_packed = on ? 1 : 0; return was; }

// classes with at least this many columns per lane use the job-drawing form (BMH_EXT_REFILL_FROM; 19 = never)
static int g_ext_refill_from = [] { const char *e = getenv("BMH_EXT_REFILL_FROM"); return e ? atoi(e) : 5; }();

template <int C>
static void launch16(const ext_args_t &base, hipStream_t st, unsigned grid)
{
	ext_args_t a = base;
	a.count = base.count + 2 * C;
	a.ctr = base.ctr + C;
	const bool lut = a.a >= 0 && a.a < 32 && a.b >= 0 && a.b < 32;
	if (!lut) extend16_static_kernel<C, false><<<grid, 256, g_ext_lds, st>>>(a);
	else if (C >= g_ext_refill_from) extend16_kernel<C, true><<<grid, 256, g_ext_lds, st>>>(a);
	else extend16_static_kernel<C, true><<<grid, 256, g_ext_lds, st>>>(a);
}
template <int G, int P>
static void launch_pk(const ext_args_t &base, hipStream_t st, unsigned grid)
{
	ext_args_t a = base;
	const int cls = ext_pk_cls_of(G, P);
	a.count = base.count + 2 * cls;
	a.ctr = base.ctr + cls;
	// (BMH_EXT_LDS_PAD=bytes: experiment knob -- unused dynamic LDS per block, i.e. fewer extension waves per CU, room for other kernels' waves)
	static const unsigned lds_pad = [] { const char *e = getenv("BMH_EXT_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
	if (a.o_ins + a.e_ins == a.o_del + a.e_del) extpk_kernel<G, P, true><<<grid, 256, lds_pad, st>>>(a);
	else extpk_kernel<G, P, false><<<grid, 256, lds_pad, st>>>(a);
}
template <int C>
static void launch_wide(const ext_args_t &base, hipStream_t st, unsigned grid)
{
	ext_args_t a = base;
	a.count = base.count + 2 * (19 + C - 5);
	extend_wide_kernel<C><<<grid, 256, 0, st>>>(a);
}

static int extend_launch(const uint8_t *d_q, const uint32_t *d_qoff, const uint32_t *d_qlen, const uint8_t *d_t,
                         const uint32_t *d_toff, const uint32_t *d_tlen, const uint32_t *d_h0, uint32_t n,
                         const bmh_ext_params_t *p, int32_t *d_out, int32_t *d_raw, void *stream_, const bmh_ext_desc_t *desc);

extern "C" int bmh_extend_batch(const uint8_t *d_q, const uint32_t *d_qoff, const uint32_t *d_qlen, const uint8_t *d_t,
                                const uint32_t *d_toff, const uint32_t *d_tlen, const uint32_t *d_h0, uint32_t n,
                                const bmh_ext_params_t *p, int32_t *d_out, int32_t *d_raw, void *stream_)
{
	if (!p || (n && (!d_q || !d_t || !d_qoff || !d_qlen || !d_toff || !d_tlen || !d_h0 || !d_out))) {
		bmh_set_error("bmh_extend_batch: null argument"); return BMH_EINVAL;
	}
	return extend_launch(d_q, d_qoff, d_qlen, d_t, d_toff, d_tlen, d_h0, n, p, d_out, d_raw, stream_, nullptr);
}

// descriptor form, used by bmh_chain_extend (chain_kernels.hip)
int bmh_extend_batch_desc(const bmh_ext_desc_t *desc, const uint32_t *d_qlen, const uint32_t *d_tlen, const uint32_t *d_h0, uint32_t n,
                          const bmh_ext_params_t *p, int32_t *d_out, int32_t *d_raw, void *stream_)
{
	if (!p || !desc || (n && (!d_qlen || !d_tlen || !d_h0 || !d_out || !desc->reads || !desc->pac || !desc->jq_src || !desc->job_side || !desc->jt0))) {
		bmh_set_error("bmh_extend_batch_desc: null argument"); return BMH_EINVAL;
	}
	return extend_launch(nullptr, nullptr, d_qlen, nullptr, nullptr, d_tlen, d_h0, n, p, d_out, d_raw, stream_, desc);
}

static int extend_launch(const uint8_t *d_q, const uint32_t *d_qoff, const uint32_t *d_qlen, const uint8_t *d_t,
                         const uint32_t *d_toff, const uint32_t *d_tlen, const uint32_t *d_h0, uint32_t n,
                         const bmh_ext_params_t *p, int32_t *d_out, int32_t *d_raw, void *stream_, const bmh_ext_desc_t *desc)
{
	if (p->e_del < 0 || p->e_ins < 0 || p->o_del < 0 || p->o_ins < 0) { bmh_set_error("bmh_extend_batch: negative gap penalty"); return BMH_EINVAL; }
	if (n == 0) return BMH_OK;
	hipStream_t st = (hipStream_t)stream_;
	int dev = 0;
	HIPCK(hipGetDevice(&dev));
	ext_scratch_t &g_scr = *scratch_for(dev, stream_);
	g_last = &g_scr;
	{ const int rc = scratch_reserve(g_scr, dev, n); if (rc != BMH_OK) return rc; }
	if (!g_scr.have_ev) {
		HIPCK(hipEventCreate(&g_scr.ev0)); HIPCK(hipEventCreate(&g_scr.ev1));
		HIPCK(hipEventCreateWithFlags(&g_scr.fork, hipEventDisableTiming));
		for (int i = 0; i < 4; ++i) {
			HIPCK(hipStreamCreateWithFlags(&g_scr.side[i], hipStreamNonBlocking));
			HIPCK(hipEventCreateWithFlags(&g_scr.join[i], hipEventDisableTiming));
		}
		g_scr.have_ev = true;
	}
	ext_args_t a;
	a.q = d_q; a.t = d_t; a.qoff = d_qoff; a.qlen = d_qlen; a.toff = d_toff; a.tlen = d_tlen; a.h0 = d_h0;
	a.desc = desc ? 1 : 0;
	a.reads = desc ? desc->reads : nullptr; a.pac = desc ? desc->pac : nullptr; a.l_pac = desc ? desc->l_pac : 0;
	a.jq_src = desc ? desc->jq_src : nullptr; a.job_side = desc ? desc->job_side : nullptr; a.jt0 = desc ? (const long long *)desc->jt0 : nullptr;
	a.ids = g_scr.vals2; a.recs = g_scr.recs; a.count = g_scr.counts; a.ctr = g_scr.counts + 2 * EXT_N_CLS; a.out = d_out; a.raw = d_raw;
	a.a = p->a; a.b = p->b; a.o_del = p->o_del; a.e_del = p->e_del; a.o_ins = p->o_ins; a.e_ins = p->e_ins;
	a.zdrop = p->zdrop; a.end_bonus = p->end_bonus;
	a.stats = nullptr;
	static const bool want_stats = getenv("BMH_EXT_STATS") != nullptr;
	static unsigned long long *d_stats = nullptr;
	if (want_stats) {
		if (!d_stats) HIPCK(hipMalloc((void **)&d_stats, 64));
		HIPCK(hipMemsetAsync(d_stats, 0, 64, st));
		a.stats = d_stats;
	}
	HIPCK(hipEventRecord(g_scr.ev0, st));
	static const bool want_phases = getenv("BMH_EXT_PHASES") != nullptr;      // debug: time of the prefilter / key / sort phases of every call
	static thread_local hipEvent_t ph[4] = {nullptr, nullptr, nullptr, nullptr};
	if (want_phases && !ph[0]) for (hipEvent_t &e : ph) HIPCK(hipEventCreate(&e));
	if (want_phases) HIPCK(hipEventRecord(ph[0], st));
	HIPCK(hipMemsetAsync(g_scr.counts, 0, 4 * 3 * EXT_N_CLS, st));
	HIPCK(hipMemsetAsync(g_scr.bins, 0, 4 * 3 * EXT_N_BINS, st));
	// packed 16-bit rows need 1 <= b, a + b <= 255 (byte score table), a >= 0 and gap penalties that fit the 16-bit lanes
	const bool pk_ok = g_ext_packed && p->a > 0 && p->b >= 1 && p->a + p->b <= 255 && p->o_del + p->e_del < 4096 && p->o_ins + p->e_ins < 4096 && p->e_ins * 32 < 4096;
	const int persist = pk_ok ? bmh_tune("EXT_PERSIST", PK_PERSIST_DEFAULT) : 0;
	{
		unsigned gp = (unsigned)((n + 31) / 32);
		if (gp > 4096) gp = 4096;
		ext_closed_form_kernel<<<gp, 256, 0, st>>>(a, n, g_scr.keys, g_scr.bins, pk_ok ? p->a : 0, persist == 0 ? bmh_tune("EXT_G2", 1) : 0);      // (EXT_G2 = k: the two-lane classes for queries beyond 16 (k - 1) columns, 0: none; the persistent kernel knows them not)
	}
	if (want_phases) { HIPCK(hipEventRecord(ph[1], st)); HIPCK(hipEventRecord(ph[2], st)); }
	ext_offsets_kernel<<<1, 64, 0, st>>>(g_scr.counts, g_scr.bins, g_scr.bins + EXT_N_BINS);
	ext_scatter_kernel<<<(n + 255) / 256, 256, 0, st>>>(a, g_scr.keys, n, g_scr.bins + EXT_N_BINS, g_scr.bins + 2 * EXT_N_BINS, g_scr.vals2, g_scr.recs);
	// class sizes stay on the device (no host sync): every class kernel is launched with a grid
	// that covers the whole batch and its waves stride over the class's slice of the sorted list
	unsigned g16 = (unsigned)((n + 15) / 16), gw = (unsigned)((n + 3) / 4);
	static const unsigned max_grid = [] { const char *e = getenv("BMH_EXT_GRID"); unsigned v = e ? (unsigned)atoi(e) : 0; return v ? v : 256u * 4; }();
	if (g16 > max_grid) g16 = max_grid;
	if (gw > max_grid) gw = max_grid;
	// the class kernels are independent: fork them over four side streams so that the tail of one
	// class overlaps the body of the next, then join back into the caller's stream
	if (want_phases) HIPCK(hipEventRecord(ph[3], st));
	HIPCK(hipEventRecord(g_scr.fork, st));
	for (int i = 0; i < 4; ++i) HIPCK(hipStreamWaitEvent(g_scr.side[i], g_scr.fork, 0));
	hipStream_t *S = g_scr.side;
	// (a caller that knows a bound of the query lengths -- the device job builder: the longest read -- spares the launches of the
	// classes beyond it: they would find their lists empty, but each occupies its side stream until it has had its turn on the chip)
	const uint32_t mq = desc && desc->max_qlen ? desc->max_qlen : 0xFFFFFFFFu;
	// the packed classes hold the bulk of the jobs when they are enabled: they go first, widest (longest running) first
	// Persistent form (EXT_PERSIST = blocks per CU, 0 = one kernel per class): one launch works all packed classes off and keeps a fixed
	// share of every SIMD for the whole pass (extpk_dev.h: extpk_persist_kernel)
	if (persist > 0) {
		static thread_local int n_cu = 0;
		if (!n_cu) { hipDeviceProp_t prop; HIPCK(hipGetDeviceProperties(&prop, dev)); n_cu = prop.multiProcessorCount; }
		unsigned gp = (unsigned)(n_cu * (persist > 3 ? 3 : persist));
		const unsigned need = (unsigned)((n + 15) / 16 + 3) / 4;            // blocks that 16 jobs a wave would fill
		if (gp > need) gp = need ? need : 1;
		const unsigned lds_pad = (unsigned)bmh_tune("EXT_LDS_PAD", 0);
		if (a.o_ins + a.e_ins == a.o_del + a.e_del) extpk_persist_kernel<true><<<gp, 256, lds_pad, S[0]>>>(a);
		else extpk_persist_kernel<false><<<gp, 256, lds_pad, S[0]>>>(a);
	} else if (pk_ok) {
		// grids sized to what is resident at once (the waves draw their jobs): 4 waves per SIMD for the classes of PK_WAVES4 (extpk_dev.h), 3 for the others (768 blocks)
		unsigned g4 = (unsigned)((n + 63) / 64), g8 = (unsigned)((n + 31) / 32);
		if (g4 > max_grid) g4 = max_grid;
		if (g8 > max_grid) g8 = max_grid;
		const unsigned g4w = g4 > max_grid * 3 / 4 ? max_grid * 3 / 4 : g4, g8w = g8 > max_grid * 3 / 4 ? max_grid * 3 / 4 : g8, g16w = g16 > max_grid * 3 / 4 ? max_grid * 3 / 4 : g16;
#if PK_G2
		{
			unsigned g2 = (unsigned)((n + 127) / 128);
			if (g2 > max_grid) g2 = max_grid;
			const unsigned g2w = g2 > max_grid * 3 / 4 ? max_grid * 3 / 4 : g2, g2h = g2 > max_grid / 2 ? max_grid / 2 : g2;
			if (mq > 112) launch_pk<2, 32>(a, S[0], g2h);
			if (mq > 96) launch_pk<2, 28>(a, S[1], g2h);
			if (mq > 80) launch_pk<2, 24>(a, S[2], g2h);
			if (mq > 64) launch_pk<2, 20>(a, S[3], g2h);
			if (mq > 48) launch_pk<2, 16>(a, S[0], g2w);
			if (mq > 32) launch_pk<2, 12>(a, S[1], g2w);
			launch_pk<2, 8>(a, S[2], g2);
		}
#endif
		if (mq > 128) launch_pk<4, 17>(a, S[2], PK_WAVES4(4, 17) ? g4 : g4w);
		if (mq > 112) launch_pk<4, 16>(a, S[3], PK_WAVES4(4, 16) ? g4 : g4w);
		if (mq > 96) launch_pk<4, 14>(a, S[0], PK_WAVES4(4, 14) ? g4 : g4w);
		if (mq > 80) launch_pk<4, 12>(a, S[1], PK_WAVES4(4, 12) ? g4 : g4w);
		if (mq > 64) launch_pk<4, 10>(a, S[2], PK_WAVES4(4, 10) ? g4 : g4w);
		if (mq > 48) launch_pk<4, 8>(a, S[3], g4);
		if (mq > 32) launch_pk<4, 6>(a, S[0], g4);
		launch_pk<4, 4>(a, S[1], g4);
		if (mq > 128) launch_pk<8, 9>(a, S[2], PK_WAVES4(8, 9) ? g8 : g8w);
		if (mq > 144) launch_pk<8, 10>(a, S[3], PK_WAVES4(8, 10) ? g8 : g8w);
#if PK_WIDE4
		const unsigned g4h = g4 > max_grid / 2 ? max_grid / 2 : g4;       // (two waves per SIMD)
		if (mq > 160) launch_pk<4, 24>(a, S[0], g4h);
		if (mq > 192) launch_pk<4, 28>(a, S[1], g4h);
		if (mq > 224) launch_pk<4, 32>(a, S[2], g4h);
#else
		if (mq > 160) launch_pk<8, 12>(a, S[0], PK_WAVES4(8, 12) ? g8 : g8w);
		if (mq > 192) launch_pk<8, 14>(a, S[1], PK_WAVES4(8, 14) ? g8 : g8w);
		if (mq > 224) launch_pk<8, 16>(a, S[2], PK_WAVES4(8, 16) ? g8 : g8w);
#endif
#if PK_WIDE18
		(void)g16w;
		if (mq > 256) launch_pk<8, 18>(a, S[3], g8w);
#else
		if (mq > 256) launch_pk<16, 9>(a, S[3], PK_WAVES4(16, 9) ? g16 : g16w);
#endif
	}
	// (narrow classes first measured better than widest first: 10.8 vs 11.1 ms)
	// class C of extend16 holds the queries of 16 (C - 1) + 1 .. 16 C columns; wide class C those of 64 (C - 1) + 1 .. 64 C, and class 5
	// also every job whose target is longer than EXT_T_CAP rows
	#define L16(C, s_) do { if (mq > 16u * ((C) - 1)) launch16<C>(a, S[s_], g16); } while (0)
	L16(1, 0); L16(2, 1); L16(3, 2); L16(4, 3); L16(5, 0); L16(6, 1); L16(7, 2); L16(8, 3); L16(9, 0); L16(10, 1); L16(11, 2); L16(12, 3);
	L16(13, 0); L16(14, 1); L16(15, 2); L16(16, 3); L16(17, 0); L16(18, 1);
	#undef L16
	launch_wide<5>(a, S[2], gw);
	if (mq > 320) launch_wide<6>(a, S[3], gw);
	if (mq > 384) launch_wide<7>(a, S[2], gw);
	if (mq > 448) launch_wide<8>(a, S[3], gw);
	if (mq > 512) launch_wide<9>(a, S[0], gw);
	if (mq > 576) launch_wide<10>(a, S[1], gw);
	if (mq > 640) launch_wide<11>(a, S[2], gw);
	if (mq > 704) launch_wide<12>(a, S[3], gw);
	for (int i = 0; i < 4; ++i) { HIPCK(hipEventRecord(g_scr.join[i], g_scr.side[i])); HIPCK(hipStreamWaitEvent(st, g_scr.join[i], 0)); }
	HIPCK(hipEventRecord(g_scr.ev1, st));
	HIPCK(hipGetLastError());
	if (want_phases) {
		float a = 0, b = 0, c = 0, d = 0;
		HIPCK(hipEventSynchronize(g_scr.ev1));
		(void)hipEventElapsedTime(&a, ph[0], ph[1]); (void)hipEventElapsedTime(&b, ph[1], ph[2]); (void)hipEventElapsedTime(&c, ph[2], ph[3]); (void)hipEventElapsedTime(&d, ph[3], g_scr.ev1);
		fprintf(stderr, "[ext] %u jobs: prefilter %.3f ms, keys %.3f, offsets + sort %.3f, class kernels %.3f\n", n, a, b, c, d);
	}
	if (want_stats) {
		unsigned long long h[8];
		HIPCK(hipStreamSynchronize(st));
		HIPCK(hipMemcpy(h, d_stats, 64, hipMemcpyDeviceToHost));
		fprintf(stderr, "[ext] packed kernels: wave-rows with a running alignment %llu, of them with every running alignment at end == qlen %llu (%.1f%%)\n", h[4], h[5], 100.0 * h[5] / (h[4] ? h[4] : 1));
		fprintf(stderr, "[ext] packed kernels (-DPK_STATS builds): idle group-rows while the wave still had jobs to draw %llu, in the wave's drain %llu\n", h[6], h[7]);
		{
			uint32_t hc[2 * EXT_N_CLS];
			HIPCK(hipMemcpy(hc, g_scr.counts, sizeof(hc), hipMemcpyDeviceToHost));
			fprintf(stderr, "[ext] jobs per class:");
			for (int c = 0; c < EXT_N_CLS; ++c) if (hc[2 * c]) fprintf(stderr, " %d:%u", c, hc[2 * c]);
			fprintf(stderr, "\n");
		}
		fprintf(stderr, "[ext] alignments %llu, rows executed %llu of %llu target rows (%.1f%%), wave-rows %llu (%.2f alignments per wave-row)\n", h[2], h[0], h[1], 100.0 * h[0] / (h[1] ? h[1] : 1), h[3], (double)h[0] / (h[3] ? h[3] : 1));
	}
	return BMH_OK;
}

// ------------------------------------------------------------------ calibration of the integer-VALU roofline

// What the chip sustains on the instruction kinds the DP kernels are made of, measured at a given occupancy: `iters` rounds of
// 128 instructions per wave, inline asm so that nothing is folded.  mode 0: v_max_i32 / v_add_u32 on eight independent registers
// (the issue ceiling), 1: one dependent chain of the same, 2: dependent v_max_i32_dpp row_shr:1 (the scans), 3: the same on four
// independent registers, 4: packed 16-bit v_pk_add_i16 / v_pk_max_i16 on eight registers (two columns per lane-op),
// 5: v_bfe_i32 on eight registers.  The peak they are held against: 256 CUs x 4 SIMD-32 x 2.4 GHz = 7.86e13 lane-ops/s (a
// wave64 instruction occupies its SIMD for 2 cycles, MI355X_MICROARCH.md).
template <int MODE>
__global__ void __launch_bounds__(256) calib_valu_kernel(int iters, int *sink, unsigned long long *place)
{
	const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	if (place && (threadIdx.x & 63) == 0) {                 // where this wave runs: HW_ID (wave / SIMD / CU / SH / SE) and the XCC
		unsigned hw, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
		place[blockIdx.x * 4 + (threadIdx.x >> 6)] = ((unsigned long long)xcc << 32) | hw;
	}
	double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;    // register pairs of mode 7
	int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
	const int k = MODE == 5 ? 3 : (blockIdx.x | 1);
	for (int i = 0; i < iters; ++i) {                      // 128 instructions per trip, nothing else in the loop but its counter
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (MODE == 0)
				asm volatile("v_max_i32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_max_i32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
				             "v_max_i32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_max_i32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t"
				             "v_add_u32 %0, %0, %8\n\tv_max_i32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_max_i32 %3, %3, %8\n\t"
				             "v_add_u32 %4, %4, %8\n\tv_max_i32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_max_i32 %7, %7, %8"
				             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k));
			else if (MODE == 1)
				asm volatile("v_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
				             "v_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
				             "v_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
				             "v_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(r0) : "v"(k));
			else if (MODE == 2)
				asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
				             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
				             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
				             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
				             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
				             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
				             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
				             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(r0));
			else if (MODE == 3)
				asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
				             "v_max_i32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
				             "v_max_i32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
				             "v_max_i32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
				             "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
				             "v_max_i32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
				             "v_max_i32_dpp %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
				             "v_max_i32_dpp %6, %6, %6 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_max_i32_dpp %7, %7, %7 row_shr:2 row_mask:0xf bank_mask:0xf"
				             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
			else if (MODE == 4)
				asm volatile("v_pk_add_i16 %0, %0, %8\n\tv_pk_max_i16 %1, %1, %8\n\tv_pk_add_i16 %2, %2, %8\n\tv_pk_max_i16 %3, %3, %8\n\t"
				             "v_pk_add_i16 %4, %4, %8\n\tv_pk_max_i16 %5, %5, %8\n\tv_pk_add_i16 %6, %6, %8\n\tv_pk_max_i16 %7, %7, %8\n\t"
				             "v_pk_max_i16 %0, %0, %8\n\tv_pk_add_i16 %1, %1, %8\n\tv_pk_max_i16 %2, %2, %8\n\tv_pk_add_i16 %3, %3, %8\n\t"
				             "v_pk_max_i16 %4, %4, %8\n\tv_pk_add_i16 %5, %5, %8\n\tv_pk_max_i16 %6, %6, %8\n\tv_pk_add_i16 %7, %7, %8"
				             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k));
			else if (MODE == 5)
				asm volatile("v_bfe_i32 %0, %0, %8, 6\n\tv_bfe_i32 %1, %1, %8, 6\n\tv_bfe_i32 %2, %2, %8, 6\n\tv_bfe_i32 %3, %3, %8, 6\n\t"
				             "v_bfe_i32 %4, %4, %8, 6\n\tv_bfe_i32 %5, %5, %8, 6\n\tv_bfe_i32 %6, %6, %8, 6\n\tv_bfe_i32 %7, %7, %8, 6\n\t"
				             "v_bfe_i32 %0, %0, %8, 6\n\tv_bfe_i32 %1, %1, %8, 6\n\tv_bfe_i32 %2, %2, %8, 6\n\tv_bfe_i32 %3, %3, %8, 6\n\t"
				             "v_bfe_i32 %4, %4, %8, 6\n\tv_bfe_i32 %5, %5, %8, 6\n\tv_bfe_i32 %6, %6, %8, 6\n\tv_bfe_i32 %7, %7, %8, 6"
				             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k));
			else if (MODE == 7)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n\tv_pk_fma_f32 %1, %1, %8, %8\n\tv_pk_fma_f32 %2, %2, %8, %8\n\tv_pk_fma_f32 %3, %3, %8, %8\n\t"
				             "v_pk_fma_f32 %4, %4, %8, %8\n\tv_pk_fma_f32 %5, %5, %8, %8\n\tv_pk_fma_f32 %6, %6, %8, %8\n\tv_pk_fma_f32 %7, %7, %8, %8\n\t"
				             "v_pk_fma_f32 %0, %0, %8, %8\n\tv_pk_fma_f32 %1, %1, %8, %8\n\tv_pk_fma_f32 %2, %2, %8, %8\n\tv_pk_fma_f32 %3, %3, %8, %8\n\t"
				             "v_pk_fma_f32 %4, %4, %8, %8\n\tv_pk_fma_f32 %5, %5, %8, %8\n\tv_pk_fma_f32 %6, %6, %8, %8\n\tv_pk_fma_f32 %7, %7, %8, %8"
				             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(d7 + 1.0));
			else if (MODE == 8)           // the DP kernels' own mix: v_perm, packed u16 mad / saturating sub / max, plain and, v_bfi
				asm volatile("v_pk_mad_u16 %0, %0, %8, %8\n\tv_pk_sub_u16 %1, %1, %8 clamp\n\tv_pk_max_u16 %2, %2, %8\n\tv_perm_b32 %3, %3, %8, %8\n\t"
				             "v_pk_max_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_pk_min_u16 %7, %7, %8\n\t"
				             "v_pk_sub_u16 %0, %0, %8 clamp\n\tv_pk_max_u16 %1, %1, %8\n\tv_pk_mad_u16 %2, %2, %8, %8\n\tv_pk_max_u16 %3, %3, %8\n\t"
				             "v_pk_add_u16 %4, %4, %8\n\tv_perm_b32 %5, %5, %8, %8\n\tv_pk_min_u16 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
				             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k));
			else
				asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
				             "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8\n\t"
				             "v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
				             "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
				             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k));
		}
	}
	const int acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (int)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 0.125);
	if (acc == 0x7fffffff) sink[0] = acc;
	if (blockIdx.x == 0 && threadIdx.x == 0) {      // shader cycles and 100 MHz ticks of this wave: cycles per instruction, and the clock the chip ran at
		const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
		((unsigned long long *)sink)[1] = c1 - c0; ((unsigned long long *)sink)[2] = w1 - w0;
	}
}

static void calib_launch(int mode, unsigned grid, int iters, int *sink, unsigned long long *place, hipStream_t st)
{
	switch (mode) {
	case 0: calib_valu_kernel<0><<<grid, 256, 0, st>>>(iters, sink, place); break;
	case 1: calib_valu_kernel<1><<<grid, 256, 0, st>>>(iters, sink, place); break;
	case 2: calib_valu_kernel<2><<<grid, 256, 0, st>>>(iters, sink, place); break;
	case 3: calib_valu_kernel<3><<<grid, 256, 0, st>>>(iters, sink, place); break;
	case 4: calib_valu_kernel<4><<<grid, 256, 0, st>>>(iters, sink, place); break;
	case 5: calib_valu_kernel<5><<<grid, 256, 0, st>>>(iters, sink, place); break;
	case 6: calib_valu_kernel<6><<<grid, 256, 0, st>>>(iters, sink, place); break;
	case 7: calib_valu_kernel<7><<<grid, 256, 0, st>>>(iters, sink, place); break;
	default: calib_valu_kernel<8><<<grid, 256, 0, st>>>(iters, sink, place); break;
	}
}

// waves_per_simd resident waves on every SIMD of the chip (256-thread blocks, one wave per SIMD each); *ms = kernel time,
// *lane_ops = 64 lanes x 128 instructions x iters per wave, summed
// bmh_calib_valu_placed: the same, and place[grid * 4] (host memory, may be NULL) receives for every wave of the timed launch
// (xcc_id << 32 | HW_ID): which SIMD of which CU it ran on -- the evidence that the launch covered every SIMD evenly.
extern "C" int bmh_calib_valu_placed(int mode, int waves_per_simd, int iters, void *stream_, float *ms, double *lane_ops, unsigned long long *place, unsigned *n_place);
// The shader clock during the calling thread's last calibration launch, measured IN the kernel: shader cycles (s_memtime) over 100 MHz
// ticks (s_memrealtime) of wave 0 around its instruction loop, and the shader cycles one wave64 instruction of that wave took.  A
// VALU-bound stage scales with this clock: the bench line carries it so that two runs on two boxes can be told apart from two builds.
static thread_local double g_calib_mhz = 0.0, g_calib_cpi = 0.0;
extern "C" int bmh_calib_last_clock(double *mhz, double *cycles_per_instr)
{
	if (mhz) *mhz = g_calib_mhz;
	if (cycles_per_instr) *cycles_per_instr = g_calib_cpi;
	return g_calib_mhz > 0.0 ? BMH_OK : BMH_EINVAL;
}
extern "C" int bmh_calib_valu(int mode, int waves_per_simd, int iters, void *stream_, float *ms, double *lane_ops)
{
	return bmh_calib_valu_placed(mode, waves_per_simd, iters, stream_, ms, lane_ops, nullptr, nullptr);
}
extern "C" int bmh_calib_valu_placed(int mode, int waves_per_simd, int iters, void *stream_, float *ms, double *lane_ops, unsigned long long *place, unsigned *n_place)
{
	if (mode < 0 || mode > 8 || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !ms || !lane_ops) { bmh_set_error("bmh_calib_valu: bad argument"); return BMH_EINVAL; }
	hipStream_t st = (hipStream_t)stream_;
	static thread_local int *sink = nullptr;
	static thread_local hipEvent_t e0 = nullptr, e1 = nullptr;
	static thread_local int n_cu = 0;
	if (!sink) {
		HIPCK(hipMalloc((void **)&sink, 64)); HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
		int dev = 0; hipDeviceProp_t prop; HIPCK(hipGetDevice(&dev)); HIPCK(hipGetDeviceProperties(&prop, dev)); n_cu = prop.multiProcessorCount;
	}
	const unsigned grid = (unsigned)(n_cu * waves_per_simd);
	unsigned long long *d_place = nullptr;
	if (place) HIPCK(hipMalloc((void **)&d_place, (size_t)grid * 4 * 8));
	calib_launch(mode, grid, 16, sink, nullptr, st);                     // warm-up (clocks, code fetch)
	HIPCK(hipEventRecord(e0, st));
	calib_launch(mode, grid, iters, sink, d_place, st);
	HIPCK(hipEventRecord(e1, st));
	HIPCK(hipEventSynchronize(e1));
	HIPCK(hipEventElapsedTime(ms, e0, e1));
	HIPCK(hipGetLastError());
	if (place) {
		HIPCK(hipMemcpy(place, d_place, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost));
		HIPCK(hipFree(d_place));
		if (n_place) *n_place = grid * 4;
	}
	*lane_ops = (double)grid * 256.0 * 128.0 * (double)iters;
	unsigned long long h[3];
	HIPCK(hipMemcpy(h, sink, 24, hipMemcpyDeviceToHost));
	g_calib_cpi = (double)h[1] / (128.0 * iters);
	g_calib_mhz = h[2] ? (double)h[1] / (double)h[2] * 100.0 : 0.0;
	if (getenv("BMH_CALIB_VERBOSE")) {
		fprintf(stderr, "[calib] mode %d, %d waves/SIMD: %.2f shader cycles per instruction of one wave, shader clock %.0f MHz\n", mode, waves_per_simd,
		        (double)h[1] / (128.0 * iters), h[2] ? (double)h[1] / (double)h[2] * 100.0 : 0.0);
	}
	return BMH_OK;
}

// wave residency trace (wtrace.h): this translation unit's copy of the trace symbols
WTRACE_DEFINE_SETTER(bmh_wtrace_set_extend)
