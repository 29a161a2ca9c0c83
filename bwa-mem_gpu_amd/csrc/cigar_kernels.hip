// Region -> CIGAR / NM / MD on the device: the step right after the hot path.
//   ksw_global2      /root/reference/src/ksw.c:1120-1241   banded global alignment, 6-bit direction matrix, traceback
//   bwa_gen_cigar2   src/bwa.c:111-216                     band width, reverse-strand flip, NM and MD
//   mem_reg2aln      src/bwamem.c:2344-2440 (+ infer_bw :1486-1494)   retry with doubled band, squeeze, soft clips, position
// One wave per region.  Rows of the DP matrix are computed by the 64 lanes at once (C columns per lane): gaps open
// from the diagonal value M, so F along a row is a max-plus prefix scan exactly as in the extension kernels; cells
// outside the band carry the reference's MINUS_INF arithmetic so that every in-band cell holds the reference's value.
// The direction matrix (one byte per in-band cell) lives in LDS when it fits, else in a per-wave slab in HBM; the
// traceback, the run-length CIGAR, NM and the MD string are produced by the wave in lock step (one lane writes).
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <utility>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include "bmh_internal.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

#define G_NEG (-0x40000000)          // MINUS_INF of ksw.c:997
#define G_SENT (-0x60000000)         // below every value the reference arithmetic can produce; only ever max()ed and shifted a little

__device__ __forceinline__ int g_wave_scan_max(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x111, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x112, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x114, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x118, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x142, 0xa, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x143, 0xc, 0xf, false); v = max(v, t);
	return v;
}
__device__ __forceinline__ int g_wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }

struct cigar_args_t {
	const uint8_t *reads; const uint32_t *offs, *lens;
	const uint8_t *pac; long long l_pac;
	const int32_t *regs;            // [n][stride]: {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi} (+ {truesc, w, ...} when stride == 16)
	int stride;
	const uint32_t *sel; uint32_t n; // optional list of region indices; n = number of jobs
	int a, b, o_del, e_del, o_ins, e_ins, opt_w;
	int max_cigar, md_cap;
	uint32_t *cigar; int32_t *aln; char *md;
	uint32_t z_lds_bytes;           // direction-matrix bytes available in LDS per wave
	uint8_t *z_slab; unsigned long long z_slab_stride;   // per-wave fallback in HBM
	uint32_t max_len;               // LDS room for query / target bases (each)
	// fast path (cigar_classify_kernel -> cigar_dp16_kernel -> cigar_finish_kernel); the wave-per-region kernel then
	// only takes the jobs of list[CG_SLOW]
	struct cg_job_t *jobs;          // per job: band, direction-matrix offset, score, kind
	uint32_t *lists; uint32_t *list_n;   // lists[k * n + i]: jobs of kind k; list_n[k] their number
	unsigned long long *z_total;    // bytes of direction matrices of the fast jobs (running sum while classifying)
	uint8_t *z; uint32_t *rev;      // direction matrices; reversed CIGAR scratch [n][max_cigar]
	int use_list;                   // wave-per-region kernel: 0 = all jobs, 1 = list[CG_SLOW]
};

enum { CG_REJECT = 0, CG_TRIVIAL = 1, CG_F2 = 2, CG_F3 = 3, CG_F5 = 4, CG_F10 = 5, CG_F16 = 6, CG_SLOW = 7, CG_NKIND = 8 };
struct cg_job_t { int32_t w, n_col, score, kind; unsigned long long zoff; };

__device__ __forceinline__ int g_code(uint8_t ch) { ch &= 0xDF; return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; }
__device__ __forceinline__ int g_text(const cigar_args_t &A, long long p)
{
	const bool rev = p >= A.l_pac;
	const long long f = rev ? (A.l_pac << 1) - 1 - p : p;
	const int c = (A.pac[f >> 2] >> ((~f & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}
__device__ __forceinline__ int g_sc(const cigar_args_t &A, int t, int q) { return (t > 3 || q > 3) ? -1 : (t == q ? A.a : -A.b); }

__device__ __forceinline__ int g_infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	const int d = l1 > l2 ? l1 - l2 : l2 - l1;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	const int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	return w < d ? d : w;
}

// banded global DP of the wave's job: fills z, returns the score.  qs/ts = bases in LDS.
template <int C>
__device__ int g_fill(const cigar_args_t &A, const uint8_t *qs, const uint8_t *ts, int qlen, int rlen, int w, uint8_t *z, int n_col, int lane)
{
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	int Hc[C], E[C], qv[C];
#pragma unroll
	for (int c = 0; c < C; ++c) {
		const int j = lane * C + c;
		qv[c] = j < qlen ? (int)qs[j] : 4;
		Hc[c] = (j + 1 <= w) ? -(A.o_ins + A.e_ins * (j + 1)) : G_NEG;      // H(-1, j)
		E[c] = G_NEG;
	}
	for (int i = 0; i < rlen; ++i) {
		const int ti = (int)ts[i];
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int fill0 = i == 0 ? 0 : (i - 1 > w ? G_NEG : -(A.o_del + A.e_del * i));           // H(i-1, -1)
		const int left = g_wave_shr1(Hc[C - 1], fill0);
		int M[C], g[C];
		int agg = G_SENT;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const int j = lane * C + c;
			const bool act = j >= beg && j < end;
			const int hd = c == 0 ? left : Hc[c - 1];
			const int m = hd + g_sc(A, ti, qv[c]);
			M[c] = m;
			g[c] = act ? m - oe_ins + A.e_ins * j : G_SENT;
			agg = max(agg, g[c]);
		}
		const int incl = g_wave_scan_max(agg);
		int run = max(g_wave_shr1(incl, G_SENT), G_NEG + A.e_ins * (beg - 1));      // the seed f(beg) = MINUS_INF in the same form
		uint8_t *zr = z + (size_t)i * n_col - beg;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const int j = lane * C + c;
			const bool act = j >= beg && j < end;
			const int f = run - A.e_ins * (j - 1);
			run = max(run, g[c]);
			const int m = M[c], e = E[c];
			int d = m >= e ? 0 : 1;
			int h = max(m, e);
			d = h >= f ? d : 2;
			h = max(h, f);
			int t = m - oe_del, e2 = e - A.e_del;
			d |= e2 > t ? 1 << 2 : 0;
			e2 = max(e2, t);
			t = m - oe_ins;
			d |= (f - A.e_ins) > t ? 2 << 4 : 0;
			if (act) { E[c] = e2; zr[j] = (uint8_t)d; }
			Hc[c] = h;
		}
	}
	const int jl = qlen - 1;
	int src = 0;
#pragma unroll
	for (int c = 0; c < C; ++c) if (jl % C == c) src = Hc[c];
	return __builtin_amdgcn_readlane(src, jl / C);
}

struct md_out_t { char *p; int len, cap; };
__device__ __forceinline__ void md_putc(md_out_t &m, char c, int lane) { if (m.len + 1 < m.cap && lane == 0) m.p[m.len] = c; ++m.len; }
__device__ __forceinline__ void md_putw(md_out_t &m, int v, int lane)
{
	// decimal digits without an indexed local array (lengths are < 2^20 < 10^7): a register array indexed inside
	// divergent loops made the compiler's lane-by-lane indexing loop spin forever in cigar_finish_kernel
	bool started = false;
#pragma unroll
	for (int d = 1000000; d >= 1; d /= 10) {
		const int q = v / d;
		v -= q * d;
		started = started || q != 0 || d == 1;
		if (started) md_putc(m, (char)('0' + q), lane);
	}
}

template <int C, int CLO>       // handles jobs with CLO < ceil(qlen / 64) <= C
__global__ void __launch_bounds__(64) cigar_kernel(cigar_args_t A)
{
	extern __shared__ __align__(16) uint8_t g_lds[];
	const int lane = threadIdx.x;
	uint8_t *qs = g_lds, *ts = g_lds + A.max_len;
	uint32_t *rev = (uint32_t *)(g_lds + 2 * (size_t)A.max_len);
	uint8_t *z_l = g_lds + 2 * (size_t)A.max_len + 4 * (size_t)A.max_cigar;
	uint8_t *z_g = A.z_slab + (size_t)blockIdx.x * A.z_slab_stride;
	const uint32_t n_iter = A.use_list ? A.list_n[CG_SLOW] : A.n;
	for (uint32_t it_ = blockIdx.x; it_ < n_iter; it_ += gridDim.x) {
		const uint32_t job = A.use_list ? A.lists[(size_t)CG_SLOW * A.n + it_] : it_;
		const uint32_t id = A.sel ? A.sel[job] : job;
		const int32_t *R = A.regs + (size_t)A.stride * id;
		const uint32_t read = (uint32_t)R[0];
		const int truesc = A.stride >= 16 ? R[8] : R[1], reg_w = A.stride >= 16 ? R[9] : A.opt_w, qb = R[2], qe = R[3];
		const long long rb = (long long)(uint32_t)R[4] | (long long)R[5] << 32, re = (long long)(uint32_t)R[6] | (long long)R[7] << 32;
		const int qlen = qe - qb, l_query = (int)A.lens[read];
		const int need = (qlen + 63) >> 6;
		if (need <= CLO || need > C) { if (!(CLO == 0 && need <= 0)) continue; }
		int32_t *out = A.aln + 8 * (size_t)job;
		uint32_t *cg = A.cigar + (size_t)job * A.max_cigar;
		md_out_t md; md.p = A.md ? A.md + (size_t)job * A.md_cap : nullptr; md.len = 0; md.cap = A.md ? A.md_cap : 0;
		const long long rlen_ll = re - rb;
		// rejected by bwa_gen_cigar2 (empty, or bridging the strands) -- or beyond this build's limits
		if (qlen <= 0 || rb >= re || (rb < A.l_pac && re > A.l_pac) || rlen_ll > (long long)A.max_len || qlen > (int)A.max_len) {
			if (lane == 0) { out[0] = out[1] = 0; out[2] = 0; out[3] = 0; out[4] = -1; out[5] = 0; out[6] = 0; out[7] = (qlen > (int)A.max_len || rlen_ll > (long long)A.max_len) ? 4 : 2; if (md.p) md.p[0] = 0; }
			continue;
		}
		const int rlen = (int)rlen_ll;
		const bool flip = rb >= A.l_pac;          // reverse strand: align the reversed sequences so that gaps go leftmost
		{
			const uint8_t *rp = A.reads + A.offs[read] + qb;
			for (int k = lane; k < qlen; k += 64) qs[k] = (uint8_t)g_code(rp[flip ? qlen - 1 - k : k]);
			for (int k = lane; k < rlen; k += 64) ts[k] = (uint8_t)g_text(A, flip ? re - 1 - k : rb + k);
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0);
		int w2 = g_infer_bw(qlen, rlen, truesc, A.a, A.o_ins, A.e_ins);
		{ const int tmp = g_infer_bw(qlen, rlen, truesc, A.a, A.o_del, A.e_del); w2 = w2 > tmp ? w2 : tmp; }
		if (w2 > A.opt_w) w2 = w2 < reg_w ? w2 : reg_w;              // ar->w: opt->w (src/bwamem.c:1280) unless the region was patched
		int score = 0, last_sc = -(1 << 30), n_ops = 0, flags = 0, it = 0;
		bool lds_z = true;
		do {
			if (w2 > A.opt_w << 2) w2 = A.opt_w << 2;
			n_ops = 0;
			if (qlen == rlen && w2 == 0) {                          // no gap possible: one M run
				int s = 0;
				for (int k = lane; k < qlen; k += 64) s += g_sc(A, ts[k], qs[k]);
				for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
				score = s;
				if (lane == 0) rev[0] = (uint32_t)qlen << 4;
				n_ops = 1;
			} else {
				int max_ins = (int)((double)(((qlen + 1) >> 1) * A.a - A.o_ins) / A.e_ins + 1.);
				int max_del = (int)((double)(((qlen + 1) >> 1) * A.a - A.o_del) / A.e_del + 1.);
				int max_gap = max_ins > max_del ? max_ins : max_del;
				max_gap = max_gap > 1 ? max_gap : 1;
				const int diff = rlen > qlen ? rlen - qlen : qlen - rlen;
				int w = (max_gap + diff + 1) >> 1;
				w = w < w2 ? w : w2;
				w = w > diff + 3 ? w : diff + 3;
				const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
				lds_z = (size_t)n_col * rlen <= A.z_lds_bytes;
				uint8_t *z = lds_z ? z_l : z_g;
				if (!lds_z && (unsigned long long)n_col * rlen > A.z_slab_stride) { flags |= 4; break; }
				score = g_fill<C>(A, qs, ts, qlen, rlen, w, z, n_col, lane);
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0);
				// traceback (ksw.c:1213-1235): state 0 = H (bits 0-1), 1 = E (bits 2-3), 2 = F (bits 4-5); ops come out reversed
				int i = rlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, state = 0, cur_op = -1, cur_len = 0;
#define G_PUSH(op_, len_) do { if ((op_) == cur_op) cur_len += (len_); else { if (cur_op >= 0) { if (n_ops < A.max_cigar - 2) { if (lane == 0) rev[n_ops] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; } else flags |= 1; ++n_ops; } cur_op = (op_); cur_len = (len_); } } while (0)
				while (i >= 0 && k >= 0) {
					const int beg = i > w ? i - w : 0;
					state = (int)z[(size_t)i * n_col + (k - beg)] >> (state << 1) & 3;
					const int op = state == 0 ? 0 : state == 1 ? 2 : 1;
					G_PUSH(op, 1);
					i -= state != 2; k -= state != 1;
				}
				if (i >= 0) G_PUSH(2, i + 1);
				if (k >= 0) G_PUSH(1, k + 1);
				G_PUSH(-2, 0);                                      // flush
#undef G_PUSH
			}
			if (score == last_sc || w2 == A.opt_w << 2) break;
			last_sc = score;
			w2 <<= 1;
		} while (++it < 3 && score < truesc - A.a);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0);
		if (n_ops > A.max_cigar - 2) n_ops = A.max_cigar - 2;
		// NM and MD along the CIGAR in forward order (op k = rev[n_ops-1-k]); bwa.c:171-204
		int n_mm = 0, n_gap = 0, x = 0, y = 0, u = 0;
		const bool fwd = rb < A.l_pac;
		for (int k = 0; k < n_ops; ++k) {
			const uint32_t c = rev[n_ops - 1 - k];
			const int op = (int)(c & 0xf), len = (int)(c >> 4);
			if (op == 0) {
				for (int b = 0; b < len; b += 64) {
					const int i = b + lane;
					const bool mis = i < len && qs[x + i] != ts[y + i];
					unsigned long long mm = __ballot(mis);
					int prev = b;
					while (mm) {
						const int bit = __builtin_ctzll(mm); mm &= mm - 1;
						const int pos = b + bit;
						u += pos - prev; prev = pos + 1;
						const int tb = (int)ts[y + pos];
						md_putw(md, u, lane);
						md_putc(md, fwd ? "ACGTN"[tb] : "TGCAN"[tb], lane);
						++n_mm; u = 0;
					}
					const int lim = b + 64 < len ? b + 64 : len;
					u += lim - prev;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < n_ops - 1) {
					md_putw(md, u, lane); md_putc(md, '^', lane);
					for (int i = 0; i < len; ++i) { const int tb = (int)ts[y + i]; md_putc(md, fwd ? "ACGTN"[tb] : "TGCAN"[tb], lane); }
					u = 0; n_gap += len;
				}
				y += len;
			} else { x += len; n_gap += len; }
		}
		md_putw(md, u, lane);
		if (md.p && lane == 0) md.p[md.len < md.cap ? md.len : md.cap - 1] = 0;
		if (md.len + 1 > md.cap && md.cap) flags |= 8;
		// mem_reg2aln tail: position, squeeze of a leading / trailing deletion, soft clips
		const long long xx = rb < A.l_pac ? rb : re - 1;
		const int is_rev = xx >= A.l_pac;
		long long pos = is_rev ? (A.l_pac << 1) - 1 - xx : xx;
		int first = 0, last = n_ops;                              // forward-order op range kept
		if (n_ops > 0) {
			const uint32_t c0 = rev[n_ops - 1], c1 = rev[0];
			if ((c0 & 0xf) == 2) { pos += c0 >> 4; first = 1; }
			else if ((c1 & 0xf) == 2) last = n_ops - 1;
		}
		int n_out = 0;
		const int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
		if (clip5) { if (lane == 0) cg[n_out] = (uint32_t)clip5 << 4 | 3; ++n_out; }
		for (int k = first + lane; k < last; k += 64) cg[n_out + (k - first)] = rev[n_ops - 1 - k];
		n_out += last - first;
		if (clip3) { if (lane == 0) cg[n_out] = (uint32_t)clip3 << 4 | 3; ++n_out; }
		if (lane == 0) {
			out[0] = (int32_t)(uint32_t)pos; out[1] = (int32_t)(pos >> 32); out[2] = is_rev; out[3] = n_out;
			out[4] = n_mm + n_gap; out[5] = score; out[6] = md.len; out[7] = flags;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0);
	}
}

// ------------------------------------------------------------------------------------------------ fast path
// Most regions need no DP at all (equal lengths and a score that leaves no room for two gaps: one M run) and nearly all
// others need one narrow band.  cigar_classify_kernel sorts the jobs into lists; cigar_dp16_kernel fills the direction
// matrices of four banded alignments per wave (one per 16-lane row, band-relative columns so that only the band is
// computed); cigar_finish_kernel does traceback, CIGAR, NM, MD and the mem_reg2aln tail with one lane per region.
// Whatever does not fit (a band wider than 256 columns, a region whose first band scores below the local score so that
// the reference retries with a doubled band) goes to the wave-per-region kernel above.

struct cg_geom_t { int qb, qe, qlen, rlen, l_query, truesc, reg_w; long long rb, re; uint32_t read; bool reject, flip; };

__device__ __forceinline__ cg_geom_t cg_geom(const cigar_args_t &A, uint32_t job)
{
	const uint32_t id = A.sel ? A.sel[job] : job;
	const int32_t *R = A.regs + (size_t)A.stride * id;
	cg_geom_t g;
	g.read = (uint32_t)R[0];
	g.truesc = A.stride >= 16 ? R[8] : R[1]; g.reg_w = A.stride >= 16 ? R[9] : A.opt_w; g.qb = R[2]; g.qe = R[3];
	g.rb = (long long)(uint32_t)R[4] | (long long)R[5] << 32; g.re = (long long)(uint32_t)R[6] | (long long)R[7] << 32;
	g.qlen = g.qe - g.qb; g.l_query = (int)A.lens[g.read];
	const long long rl = g.re - g.rb;
	g.reject = g.qlen <= 0 || g.rb >= g.re || (g.rb < A.l_pac && g.re > A.l_pac) || rl > (1 << 20) || g.qlen > (1 << 20);
	g.rlen = g.reject ? 0 : (int)rl;
	g.flip = g.rb >= A.l_pac;
	return g;
}
__device__ __forceinline__ int cg_q(const cigar_args_t &A, const cg_geom_t &g, int k)     // k-th base of the (possibly reversed) query
{
	return g_code(A.reads[(size_t)A.offs[g.read] + g.qb + (g.flip ? g.qlen - 1 - k : k)]);
}
__device__ __forceinline__ int cg_t(const cigar_args_t &A, const cg_geom_t &g, int k) { return g_text(A, g.flip ? g.re - 1 - k : g.rb + k); }
__device__ __forceinline__ int cg_first_w2(const cigar_args_t &A, const cg_geom_t &g)
{
	int w2 = g_infer_bw(g.qlen, g.rlen, g.truesc, A.a, A.o_ins, A.e_ins);
	const int tmp = g_infer_bw(g.qlen, g.rlen, g.truesc, A.a, A.o_del, A.e_del);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > A.opt_w) w2 = w2 < g.reg_w ? w2 : g.reg_w;
	if (w2 > A.opt_w << 2) w2 = A.opt_w << 2;
	return w2;
}

__global__ void __launch_bounds__(256) cigar_classify_kernel(cigar_args_t A)
{
	const uint32_t job = blockIdx.x * 256u + threadIdx.x;
	const int lane = threadIdx.x & 63;
	int kind = -1; unsigned long long zb = 0; int w = 0, n_col = 0;
	if (job < A.n) {
		const cg_geom_t g = cg_geom(A, job);
		if (g.reject) kind = CG_REJECT;
		else {
			const int w2 = cg_first_w2(A, g);
			if (g.qlen == g.rlen && w2 == 0) kind = CG_TRIVIAL;
			else {
				int max_ins = (int)((double)(((g.qlen + 1) >> 1) * A.a - A.o_ins) / A.e_ins + 1.);
				int max_del = (int)((double)(((g.qlen + 1) >> 1) * A.a - A.o_del) / A.e_del + 1.);
				int max_gap = max_ins > max_del ? max_ins : max_del;
				max_gap = max_gap > 1 ? max_gap : 1;
				const int diff = g.rlen > g.qlen ? g.rlen - g.qlen : g.qlen - g.rlen;
				w = (max_gap + diff + 1) >> 1;
				w = w < w2 ? w : w2;
				w = w > diff + 3 ? w : diff + 3;
				n_col = g.qlen < 2 * w + 1 ? g.qlen : 2 * w + 1;
				const int slots = n_col;                        // columns alive in one row: end - beg <= min(qlen, 2w+1)
				// (bands of up to 32 and 48 columns have kernels of their own: three to five mismatches in a read without a gap ask for 23 .. 43 columns, and most of the
				// regions that need a DP at all are such reads)
				kind = slots <= 32 ? CG_F2 : slots <= 48 ? CG_F3 : slots <= 80 ? CG_F5 : slots <= 160 ? CG_F10 : slots <= 256 ? CG_F16 : CG_SLOW;
				if (kind != CG_SLOW) zb = (unsigned long long)n_col * (unsigned long long)g.rlen;
			}
		}
	}
	// direction-matrix offsets: wave-level exclusive scan of the sizes, one atomic per wave
	unsigned long long incl = zb;
	for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = (unsigned long long)__shfl_up((long long)incl, o); if (lane >= o) incl += t; }
	const unsigned long long tot = (unsigned long long)__shfl((long long)incl, 63);
	unsigned long long base = 0;
	if (lane == 0 && tot) base = atomicAdd(A.z_total, tot);
	base = (unsigned long long)__shfl((long long)base, 0);
	if (job < A.n) { cg_job_t j; j.w = w; j.n_col = n_col; j.score = 0; j.kind = kind; j.zoff = base + incl - zb; A.jobs[job] = j; }
	// per-kind lists, one atomic per wave and kind
	for (int k = CG_F2; k <= CG_SLOW; ++k) {
		const unsigned long long m = __ballot(kind == k);
		if (!m) continue;
		uint32_t b = 0;
		if (lane == (int)__builtin_ctzll(m)) b = atomicAdd(A.list_n + k, (uint32_t)__builtin_popcountll(m));
		b = (uint32_t)__shfl((int)b, (int)__builtin_ctzll(m));
		if (kind == k) A.lists[(size_t)k * A.n + b + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1))] = job;
	}
}

__device__ __forceinline__ int cg_row_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false); }   // row_shr:1
__device__ __forceinline__ int cg_row_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x101, 0xf, 0xf, false); }   // row_shl:1
__device__ __forceinline__ int cg_row_scan_max(int v)      // inclusive max-scan inside the 16-lane row
{
	int t;
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x111, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x112, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x114, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(G_SENT, v, 0x118, 0xf, 0xf, false); v = max(v, t);
	return v;
}

// Four banded global alignments per wave.  Slot o of a row holds column beg(i) + o, beg(i) = max(0, i - w): while the
// band still starts at column 0 the diagonal neighbour H(i-1,j-1) sits one slot to the left; once the band slides
// (i > w) it sits in the same slot, and E(i,j) and the query base arrive from one slot to the right.
template <int C>
__global__ void __launch_bounds__(256) cigar_dp16_kernel(cigar_args_t A, int kind)
{
	const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t n_list = A.list_n[kind];
	const uint32_t li = wave * 4 + grp;
	const bool have = li < n_list;
	if (wave * 4 >= n_list) return;
	const uint32_t job = have ? A.lists[(size_t)kind * A.n + li] : 0;
	const int oe_del = A.o_del + A.e_del, oe_ins = A.o_ins + A.e_ins;
	const int bp_base = (lane & 48) << 2;
	cg_geom_t g = cg_geom(A, job);
	cg_job_t J = A.jobs[job];
	const int w = J.w, n_col = J.n_col, qlen = have ? g.qlen : 0, rlen = have ? g.rlen : 0;
	uint8_t *z = A.z + J.zoff;
	const int o0 = l16 * C;                                    // first slot of this lane
	int Hs[C], E[C], qv[C];
#pragma unroll
	for (int c = 0; c < C; ++c) {
		const int j = o0 + c;
		qv[c] = j < qlen ? cg_q(A, g, j) : 4;
		Hs[c] = (j + 1 <= w) ? -(A.o_ins + A.e_ins * (j + 1)) : G_NEG;      // H(-1, j)
		E[c] = G_NEG;
	}
	int tchunk = 4, qchunk = 4;
	int final_h = 0;
	const int max_rlen = max(max(__shfl(rlen, 0), __shfl(rlen, 16)), max(__shfl(rlen, 32), __shfl(rlen, 48)));
	for (int i = 0; i < max_rlen; ++i) {
		const bool run = i < rlen;
		if ((i & 15) == 0) tchunk = (run && i + l16 < rlen) ? cg_t(A, g, i + l16) : 4;
		const int ti = __builtin_amdgcn_ds_bpermute(bp_base + ((i & 15) << 2), tchunk);
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const bool slide = i > w;                               // beg(i) = beg(i-1) + 1
		// sliding band: E and the query bases move one slot left; the base entering the last slot is q[beg + 16C - 1]
		{
			const int fi = beg + 16 * C - 1;
			if (slide && ((fi & 15) == 0 || i == w + 1)) { const int k = (fi & ~15) + l16; qchunk = (run && k < qlen) ? cg_q(A, g, k) : 4; }
			const int qin = __builtin_amdgcn_ds_bpermute(bp_base + ((fi & 15) << 2), qchunk);
			const int e_in = cg_row_shl1(E[0], G_NEG), q_in = cg_row_shl1(qv[0], qin);
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const int en = c == C - 1 ? e_in : E[c + 1], qn = c == C - 1 ? q_in : qv[c + 1];
				E[c] = slide ? en : E[c]; qv[c] = slide ? qn : qv[c];
			}
		}
		const int fill0 = i == 0 ? 0 : (i - 1 > w ? G_NEG : -(A.o_del + A.e_del * i));           // H(i-1, -1)
		const int left = cg_row_shr1(Hs[C - 1], fill0);
		const unsigned wdt = (unsigned)(end - beg);
		int M[C], gg[C];
		int agg = G_SENT;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const bool act = run && (unsigned)(o0 + c) < wdt;
			const int hd = slide ? Hs[c] : (c == 0 ? left : Hs[c - 1]);
			const int m = hd + g_sc(A, ti, qv[c]);
			M[c] = m;
			gg[c] = act ? m - oe_ins + A.e_ins * (beg + o0 + c) : G_SENT;
			agg = max(agg, gg[c]);
		}
		int runm = max(cg_row_shr1(cg_row_scan_max(agg), G_SENT), G_NEG + A.e_ins * (beg - 1));
		uint8_t *zr = z + (size_t)i * n_col;
		int Hn[C];
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const int o = o0 + c, j = beg + o;
			const bool act = run && (unsigned)o < wdt;
			const int f = runm - A.e_ins * (j - 1);
			runm = max(runm, gg[c]);
			const int m = M[c], e = E[c];
			int d = m >= e ? 0 : 1;
			int h = max(m, e);
			d = h >= f ? d : 2;
			h = max(h, f);
			int t = m - oe_del, e2 = e - A.e_del;
			d |= e2 > t ? 1 << 2 : 0;
			e2 = max(e2, t);
			t = m - oe_ins;
			d |= (f - A.e_ins) > t ? 2 << 4 : 0;
			if (act) { E[c] = e2; zr[o] = (uint8_t)d; }
			Hn[c] = h;
		}
#pragma unroll
		for (int c = 0; c < C; ++c) Hs[c] = run ? Hn[c] : Hs[c];
		if (__any(run && i == rlen - 1)) {                      // H(rlen-1, qlen-1), the global score
			const int so = max(qlen - 1 - beg, 0);               // its slot in this row
			int src = Hs[0];
#pragma unroll
			for (int c = 1; c < C; ++c) src = (so % C) == c ? Hs[c] : src;
			const int v = __builtin_amdgcn_ds_bpermute(bp_base + ((so / C) << 2), src);
			final_h = (run && i == rlen - 1) ? v : final_h;
		}
	}
	if (have && l16 == 0) {
		A.jobs[job].score = final_h;
		// the reference retries with a doubled band when the global score falls short of the local one (src/bwamem.c:2386)
		const int w2 = cg_first_w2(A, g);
		if (final_h < g.truesc - A.a && w2 != A.opt_w << 2) {
			A.jobs[job].kind = CG_SLOW;
			A.lists[(size_t)CG_SLOW * A.n + atomicAdd(A.list_n + CG_SLOW, 1u)] = job;
		}
	}
}

// 16 lanes per region: the row's first lane walks the direction matrix back (fast jobs) and writes the results; the
// M runs are compared 16 bases at a time by the whole row (NM, MD, and the score of the one-run jobs)
__global__ void __launch_bounds__(256) cigar_finish_kernel(cigar_args_t A)
{
	const int lane = threadIdx.x & 63, l16 = lane & 15, sh = lane & 48;
	const uint32_t job = (blockIdx.x * 256u + threadIdx.x) >> 4;
	const bool have = job < A.n;
	cg_job_t J; J.kind = CG_SLOW; J.w = J.n_col = J.score = 0; J.zoff = 0;
	if (have) J = A.jobs[job];
	const bool work = have && J.kind != CG_SLOW;
	if (!__any(work)) return;
	const cg_geom_t g = cg_geom(A, have ? job : 0);
	const bool lead = work && l16 == 0;
	int32_t *out = A.aln + 8 * (size_t)job;
	uint32_t *cg = A.cigar + (size_t)job * A.max_cigar;
	char *mdp = A.md ? A.md + (size_t)job * A.md_cap : nullptr;
	const int md_cap = A.md ? A.md_cap : 0;
	uint32_t *rev = A.rev + (size_t)(have ? job : 0) * A.max_cigar;
	const bool rej = work && J.kind == CG_REJECT;
	if (lead && rej) {
		out[0] = out[1] = 0; out[2] = 0; out[3] = 0; out[4] = -1; out[5] = 0; out[6] = 0; out[7] = 2;
		if (mdp) mdp[0] = 0;
	}
	const bool go = work && !rej;
	const int qlen = g.qlen, rlen = g.rlen;
	int n_ops = 0, flags = 0;
	// The walk back through the direction matrix is a chain of dependent byte loads -- one round trip to HBM per step, 150-300 us for a read with a gap while the
	// row's other fifteen lanes waited (round 6: the kernel was 4.6 ms per million alignments, the largest of the rows behind the path).  The path only ever moves
	// up and to the left, one row and / or one column a step, so the next sixteen steps lie in the 16 x 16 window whose lower right corner is the current cell:
	// the row's lanes fetch that window -- lane l the sixteen bytes of row i - l, all loads in flight at once -- into LDS, and ALL of them walk it (the same values
	// in every lane; only the first one writes the operations), then the next window.
	__shared__ uint32_t zwin[4][4][16 * 4 + 4];                                  // [wave of the block][row of the wave][16 rows x 4 dwords]
	const bool dp = go && J.kind != CG_TRIVIAL;
	if (lead && !rej && J.kind == CG_TRIVIAL) { rev[0] = (uint32_t)qlen << 4; n_ops = 1; }
	{
		const int w = J.w, n_col = J.n_col;
		const uint8_t *z = A.z + J.zoff;
		uint32_t *win = zwin[threadIdx.x >> 6][(lane >> 4) & 3];
		int i = rlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, state = 0, cur_op = -1, cur_len = 0;
#define G_PUSH(op_, len_) do { if ((op_) == cur_op) cur_len += (len_); else { if (cur_op >= 0) { if (n_ops < A.max_cigar - 2) { if (l16 == 0) rev[n_ops] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; } else flags |= 1; ++n_ops; } cur_op = (op_); cur_len = (len_); } } while (0)
		while (dp && i >= 0 && k >= 0) {                                          // (uniform over the row's sixteen lanes)
			const int i0 = i, k0 = k;
			{
				const int row = i0 - l16;
				uint32_t b4[4] = {0u, 0u, 0u, 0u};
				if (row >= 0) {
					const int beg = row > w ? row - w : 0;
					const uint8_t *zr = z + (size_t)row * n_col - beg;                // zr[column] for the columns of the row's band
#pragma unroll
					for (int c = 0; c < 16; ++c) {
						const int kk = k0 - 15 + c, rel = kk - beg;
						const uint32_t v = (kk >= 0 && rel >= 0 && rel < n_col) ? (uint32_t)zr[kk] : 0u;
						b4[c >> 2] |= v << ((c & 3) << 3);
					}
				}
#pragma unroll
				for (int c = 0; c < 4; ++c) win[l16 * 4 + c] = b4[c];
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xC07F);      // (the window is in LDS: lgkmcnt 0)
			while (i >= 0 && k >= 0 && i0 - i < 16 && k0 - k < 16) {
				const int byte = (int)(((const uint8_t *)win)[(i0 - i) * 16 + (k - (k0 - 15))]);
				state = byte >> (state << 1) & 3;
				const int op = state == 0 ? 0 : state == 1 ? 2 : 1;
				G_PUSH(op, 1);
				i -= state != 2; k -= state != 1;
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xC07F);      // (read before the next window overwrites it)
		}
		if (dp) {
			if (i >= 0) G_PUSH(2, i + 1);
			if (k >= 0) G_PUSH(1, k + 1);
			G_PUSH(-2, 0);
			if (n_ops > A.max_cigar - 2) n_ops = A.max_cigar - 2;
		}
#undef G_PUSH
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0);
	n_ops = __shfl(n_ops, sh); flags = __shfl(flags, sh);
	// NM, MD (bwa.c:171-204) and, for the one-run jobs, the score
	int n_mm = 0, n_gap = 0, x = 0, y = 0, u = 0, md_len = 0, sc_part = 0;
	const bool fwd = g.rb < A.l_pac;
#define MD_PUTC(ch_) do { if (md_len + 1 < md_cap && lead) mdp[md_len] = (ch_); ++md_len; } while (0)
#define MD_PUTW(v_) do { int v__ = (v_); bool st__ = false; _Pragma("unroll") for (int d__ = 1000000; d__ >= 1; d__ /= 10) { const int q__ = v__ / d__; v__ -= q__ * d__; st__ = st__ || q__ != 0 || d__ == 1; if (st__) MD_PUTC((char)('0' + q__)); } } while (0)      /* lengths are < 2^20 < 10^7; no indexed local array */
	if (n_ops > A.max_cigar - 2) n_ops = A.max_cigar - 2;
	const int max_ops = max(max(__shfl(go ? n_ops : 0, 0), __shfl(go ? n_ops : 0, 16)), max(__shfl(go ? n_ops : 0, 32), __shfl(go ? n_ops : 0, 48)));
	for (int k = 0; k < max_ops; ++k) {
		const bool on = go && k < n_ops;
		const uint32_t c = on ? rev[n_ops - 1 - k] : 0;
		const int op = on ? (int)(c & 0xf) : 1, len = on ? (int)(c >> 4) : 0;
		if (__any(on && op == 0)) {
			const int mlen = (on && op == 0) ? min(len, min(qlen - x, rlen - y)) : 0;       // (a run never exceeds what is left)
			const int max_len = max(max(__shfl(mlen, 0), __shfl(mlen, 16)), max(__shfl(mlen, 32), __shfl(mlen, 48)));
			for (int b = 0; b < max_len; b += 16) {
				const int i = b + l16;
				bool mis = false; int tb = 0;
				if (i < mlen) { const int qb_ = cg_q(A, g, x + i); tb = cg_t(A, g, y + i); sc_part += g_sc(A, tb, qb_); mis = qb_ != tb; }
				const unsigned long long bal = __ballot(mis);
				const unsigned mm = (unsigned)((bal >> sh) & 0xFFFFull);
				int prev = b;
				if (bal) {                                      // some row of the wave has a mismatch in this chunk
					for (int bit = 0; bit < 16; ++bit) {
						const int tbb = __shfl(tb, sh + bit);
						if ((mm >> bit) & 1) {
							const int pos = b + bit;
							u += pos - prev; prev = pos + 1;
							MD_PUTW(u); MD_PUTC(fwd ? "ACGTN"[tbb] : "TGCAN"[tbb]);
							++n_mm; u = 0;
						}
					}
				}
				const int lim = b + 16 < mlen ? b + 16 : mlen;
				u += lim > prev ? lim - prev : 0;
			}
		}
		if (on && op == 0) { x += len; y += len; }
		else if (on && op == 2) {
			if (k > 0 && k < n_ops - 1) {
				MD_PUTW(u); MD_PUTC('^');
				for (int i = 0; i < len && y + i < rlen; ++i) { const int tb = cg_t(A, g, y + i); MD_PUTC(fwd ? "ACGTN"[tb] : "TGCAN"[tb]); }
				u = 0; n_gap += len;
			}
			y += len;
		} else if (on) { x += len; n_gap += len; }
	}
	if (go) MD_PUTW(u);
#undef MD_PUTW
#undef MD_PUTC
	// sum of the per-lane score parts over the row
	for (int o = 8; o; o >>= 1) sc_part += __shfl_xor(sc_part, o);
	if (!(lead && !rej)) return;
	if (mdp) mdp[md_len < md_cap ? md_len : md_cap - 1] = 0;
	if (md_len + 1 > md_cap && md_cap) flags |= 8;
	const int score = J.kind == CG_TRIVIAL ? sc_part : J.score;
	const long long xx = g.rb < A.l_pac ? g.rb : g.re - 1;
	const int is_rev = xx >= A.l_pac;
	long long pos = is_rev ? (A.l_pac << 1) - 1 - xx : xx;
	int first = 0, last = n_ops;
	if (n_ops > 0) {
		const uint32_t c0 = rev[n_ops - 1], c1 = rev[0];
		if ((c0 & 0xf) == 2) { pos += c0 >> 4; first = 1; }
		else if ((c1 & 0xf) == 2) last = n_ops - 1;
	}
	int n_out = 0;
	const int clip5 = is_rev ? g.l_query - g.qe : g.qb, clip3 = is_rev ? g.qb : g.l_query - g.qe;
	if (clip5) cg[n_out++] = (uint32_t)clip5 << 4 | 3;
	for (int k = first; k < last; ++k) cg[n_out++] = rev[n_ops - 1 - k];
	if (clip3) cg[n_out++] = (uint32_t)clip3 << 4 | 3;
	out[0] = (int32_t)(uint32_t)pos; out[1] = (int32_t)(pos >> 32); out[2] = is_rev; out[3] = n_out;
	out[4] = n_mm + n_gap; out[5] = score; out[6] = md_len; out[7] = flags;
}

// largest direction matrix (the whole rectangle bounds every retry) and longest sequence of the batch
__global__ void __launch_bounds__(256) cigar_size_kernel(const int32_t *regs, int stride, const uint32_t *sel, uint32_t n, unsigned long long *out)
{
	const uint32_t job = blockIdx.x * 256u + threadIdx.x;
	unsigned long long zb = 0, ml = 0;
	if (job < n) {
		const int32_t *R = regs + (size_t)stride * (sel ? sel[job] : job);
		const long long rb = (long long)(uint32_t)R[4] | (long long)R[5] << 32, re = (long long)(uint32_t)R[6] | (long long)R[7] << 32;
		const long long ql = (long long)R[3] - R[2], rl = re - rb;
		if (ql > 0 && rl > 0 && ql < (1 << 20) && rl < (1 << 20)) { zb = (unsigned long long)(ql * rl); ml = (unsigned long long)(ql > rl ? ql : rl); }
	}
	for (int o = 32; o; o >>= 1) { zb = max(zb, (unsigned long long)__shfl_xor((long long)zb, o)); ml = max(ml, (unsigned long long)__shfl_xor((long long)ml, o)); }
	if ((threadIdx.x & 63) == 0) { atomicMax(out, zb); atomicMax(out + 1, ml); }
}

// the scratch of bmh_cigar_batch per (device, stream), like the other stages': kept between calls (the direction matrices of a million 300 bp alignments are
// gigabytes: allocating them per run of a worker thread cost more than the kernels), dropped by bmh_cigar_release(stream) when the caller retires the stream.
// A stream belongs to one thread at a time (the library's rule for all per-stream scratch).
struct cigar_scratch_t {
	unsigned long long *d_sizes = nullptr;       // [0] largest rectangle [1] longest sequence [2] fast-path matrix bytes; then CG_NKIND list counts (u32)
	uint8_t *slab = nullptr; size_t slab_bytes = 0;
	cg_job_t *jobs = nullptr; uint32_t *lists = nullptr, *rev = nullptr; size_t cap_n = 0, cap_rev = 0; uint8_t *z = nullptr; size_t z_bytes = 0;
	void drop()
	{
		void *ps[] = {d_sizes, slab, jobs, lists, rev, z};
		for (void *q : ps) if (q) (void)hipFree(q);
		d_sizes = nullptr; slab = nullptr; jobs = nullptr; lists = rev = nullptr; z = nullptr;
		slab_bytes = cap_n = cap_rev = z_bytes = 0;
	}
};
static std::mutex g_cs_mu;
static std::map<std::pair<int, void *>, cigar_scratch_t *> g_cs_map;

extern "C" void bmh_cigar_release(void *stream_)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return;
	cigar_scratch_t *S = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_cs_mu);
		auto it = g_cs_map.find(std::make_pair(dev, stream_));
		if (it == g_cs_map.end()) return;
		S = it->second;
		g_cs_map.erase(it);
	}
	S->drop();
	delete S;
}

template <int C, int CLO>
static int launch_cigar(const cigar_args_t &a, unsigned grid, size_t lds, hipStream_t st)
{
	HIPCK(hipFuncSetAttribute((const void *)cigar_kernel<C, CLO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	cigar_kernel<C, CLO><<<grid, 64, lds, st>>>(a);
	return BMH_OK;
}

template <class T> static int cg_grow(T *&p, size_t bytes)
{
	if (p) { (void)hipFree(p); p = nullptr; }
	if (hipMalloc((void **)&p, bytes) != hipSuccess) { bmh_set_error("bmh_cigar_batch: hipMalloc of %zu bytes failed", bytes); return BMH_ENOMEM; }
	return BMH_OK;
}

extern "C" int bmh_cigar_batch(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const uint32_t *d_lens,
                               const int32_t *d_regs, int reg_stride, const uint32_t *d_sel, uint32_t n, const bmh_ext_params_t *p, int opt_w,
                               int max_cigar, uint32_t *d_cigar, int32_t *d_aln, int md_cap, char *d_md, void *stream_)
{
	if (!idx || !p || (n && (!d_reads || !d_offs || !d_lens || !d_regs || !d_cigar || !d_aln))) { bmh_set_error("bmh_cigar_batch: null argument"); return BMH_EINVAL; }
	if (!idx->dev.pac || idx->dev.l_pac == 0) { bmh_set_error("bmh_cigar_batch: the index was uploaded without the 2-bit reference (pac)"); return BMH_EINVAL; }
	if ((reg_stride != 8 && reg_stride != 16) || max_cigar < 4 || (d_md && md_cap < 2) || p->e_del < 1 || p->e_ins < 1 || opt_w < 1) { bmh_set_error("bmh_cigar_batch: bad argument"); return BMH_EINVAL; }
	if (n == 0) return BMH_OK;
	hipStream_t st = (hipStream_t)stream_;
	static const bool slow_only = getenv("BMH_CIGAR_SLOW") != nullptr;
	cigar_scratch_t *gs;
	{
		int dev = 0;
		HIPCK(hipGetDevice(&dev));
		std::lock_guard<std::mutex> lk(g_cs_mu);
		auto key = std::make_pair(dev, stream_);
		auto it = g_cs_map.find(key);
		if (it == g_cs_map.end()) { gs = new cigar_scratch_t(); g_cs_map[key] = gs; }
		else gs = it->second;
	}
	cigar_scratch_t &g_cs = *gs;
	if (!g_cs.d_sizes) HIPCK(hipMalloc((void **)&g_cs.d_sizes, 64));
	HIPCK(hipMemsetAsync(g_cs.d_sizes, 0, 64, st));
	cigar_size_kernel<<<(n + 255) / 256, 256, 0, st>>>(d_regs, reg_stride, d_sel, n, g_cs.d_sizes);
	cigar_args_t a;
	memset(&a, 0, sizeof(a));
	a.reads = d_reads; a.offs = d_offs; a.lens = d_lens; a.pac = idx->dev.pac; a.l_pac = (long long)idx->dev.l_pac;
	a.regs = d_regs; a.stride = reg_stride; a.sel = d_sel; a.n = n;
	a.a = p->a; a.b = p->b; a.o_del = p->o_del; a.e_del = p->e_del; a.o_ins = p->o_ins; a.e_ins = p->e_ins; a.opt_w = opt_w;
	a.max_cigar = max_cigar; a.md_cap = d_md ? md_cap : 0; a.cigar = d_cigar; a.aln = d_aln; a.md = d_md;
	if (!slow_only) {
		if (g_cs.cap_n < n) {
			const size_t c = (size_t)n + n / 4 + 1024;
			if (cg_grow(g_cs.jobs, sizeof(cg_job_t) * c) != BMH_OK || cg_grow(g_cs.lists, 4 * CG_NKIND * c) != BMH_OK) return BMH_ENOMEM;
			g_cs.cap_n = c;
		}
		a.jobs = g_cs.jobs; a.lists = g_cs.lists; a.list_n = (uint32_t *)(g_cs.d_sizes + 3); a.z_total = g_cs.d_sizes + 2;
		cigar_classify_kernel<<<(n + 255) / 256, 256, 0, st>>>(a);
	}
	unsigned long long h[8];
	HIPCK(hipMemcpyAsync(h, g_cs.d_sizes, 64, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	const uint32_t *h_list_n = (const uint32_t *)(h + 3);
	if (!slow_only && getenv("BMH_CIGAR_STATS"))
		fprintf(stderr, "[cigar] %u regions: band<=32 %u, <=48 %u, <=80 %u, <=160 %u, <=256 %u, wide/other %u; direction matrices %.1f MB\n", n, h_list_n[CG_F2], h_list_n[CG_F3], h_list_n[CG_F5], h_list_n[CG_F10],
		        h_list_n[CG_F16], h_list_n[CG_SLOW], (double)h[2] / 1e6);
	const uint32_t max_len = (uint32_t)((h[1] + 15) & ~15ull);
	if (max_len > 704) { bmh_set_error("bmh_cigar_batch: a region spans %llu bases (limit 704)", h[1]); return BMH_ECAPACITY; }
	a.max_len = max_len;
	if (!slow_only) {
		// fast path: direction matrices of all single-band jobs at once (HBM is large), then one lane per region
		if (g_cs.z_bytes < h[2] + 256) { const size_t c = (size_t)h[2] + (size_t)h[2] / 4 + 4096; if (cg_grow(g_cs.z, c) != BMH_OK) return BMH_ENOMEM; g_cs.z_bytes = c; }
		if (g_cs.cap_rev < (size_t)n * max_cigar) { const size_t c = (size_t)n * max_cigar + 1024; if (cg_grow(g_cs.rev, 4 * c) != BMH_OK) return BMH_ENOMEM; g_cs.cap_rev = c; }
		a.z = g_cs.z; a.rev = g_cs.rev;
		if (h_list_n[CG_F2]) cigar_dp16_kernel<2><<<(h_list_n[CG_F2] + 15) / 16, 256, 0, st>>>(a, CG_F2);
		if (h_list_n[CG_F3]) cigar_dp16_kernel<3><<<(h_list_n[CG_F3] + 15) / 16, 256, 0, st>>>(a, CG_F3);
		if (h_list_n[CG_F5]) cigar_dp16_kernel<5><<<(h_list_n[CG_F5] + 15) / 16, 256, 0, st>>>(a, CG_F5);
		if (h_list_n[CG_F10]) cigar_dp16_kernel<10><<<(h_list_n[CG_F10] + 15) / 16, 256, 0, st>>>(a, CG_F10);
		if (h_list_n[CG_F16]) cigar_dp16_kernel<16><<<(h_list_n[CG_F16] + 15) / 16, 256, 0, st>>>(a, CG_F16);
		cigar_finish_kernel<<<(n + 15) / 16, 256, 0, st>>>(a);
		a.use_list = 1;
	}
	// LDS per wave: bases + reversed ops + the direction matrix of a first try (band ~ qlen/2); larger matrices go to HBM
	size_t z_lds = (size_t)h[0] * 6 / 10 + 256;
	const size_t fixed = 2 * (size_t)max_len + 4 * (size_t)max_cigar;
	if (z_lds + fixed > 40 * 1024) z_lds = 40 * 1024 > fixed ? 40 * 1024 - fixed : 0;
	z_lds &= ~(size_t)15;
	a.z_lds_bytes = (uint32_t)z_lds;
	const size_t lds = fixed + z_lds;
	unsigned grid = 256u * (unsigned)(lds ? (160 * 1024) / lds : 16);
	if (grid > 256u * 16) grid = 256u * 16;
	if (grid > n) grid = n;
	a.z_slab_stride = (h[0] + 255) & ~255ull;
	const size_t slab = (size_t)a.z_slab_stride * grid;
	if (g_cs.slab_bytes < slab) {
		if (g_cs.slab) (void)hipFree(g_cs.slab);
		g_cs.slab = nullptr; g_cs.slab_bytes = 0;
		if (hipMalloc((void **)&g_cs.slab, slab) != hipSuccess) { bmh_set_error("bmh_cigar_batch: hipMalloc of %zu bytes failed", slab); return BMH_ENOMEM; }
		g_cs.slab_bytes = slab;
	}
	a.z_slab = g_cs.slab;
	int rc = BMH_OK;
	rc = launch_cigar<1, 0>(a, grid, lds, st);
	if (rc == BMH_OK && max_len > 64) rc = launch_cigar<2, 1>(a, grid, lds, st);
	if (rc == BMH_OK && max_len > 128) rc = launch_cigar<3, 2>(a, grid, lds, st);
	if (rc == BMH_OK && max_len > 192) rc = launch_cigar<5, 3>(a, grid, lds, st);
	if (rc == BMH_OK && max_len > 320) rc = launch_cigar<8, 5>(a, grid, lds, st);
	if (rc == BMH_OK && max_len > 512) rc = launch_cigar<11, 8>(a, grid, lds, st);
	if (rc != BMH_OK) return rc;
	HIPCK(hipGetLastError());
	return BMH_OK;
}
