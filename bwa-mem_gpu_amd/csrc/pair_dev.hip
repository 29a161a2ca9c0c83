// mem_pair and the per-pair choices of mem_sam_pe ON THE DEVICE (/root/reference/src/bwamem_pair.c:190-397) for the pairs
// the mate rescue does not touch -- nineteen in twenty -- next to msw_kernel, which aligns the rescue's windows for the others.
//
// A pair without a rescued region goes through mem_sam_pe with the regions its reads leave mem_sort_dedup_patch with; mem_mark_primary_se
// of either read and the selection of mem_reg2sam are then what the single-end tail computes (same regions, same tie-break hash: the
// read's index in the run), so the kernel starts from the records of bmh_finalize_regs_device and changes what pairing changes: mem_pair
// over the hits of both reads (sorted by position; the best and the second-best pair, the number of close runners-up), the pair's MAPQ,
// which hit of either read is reported, its MAPQ, the pair flags, the XA group of a chosen secondary hit.  One pair per lane.
//
// Two things stay with the host (bmh_finalize_pairs_split: the host walks of csrc/pair_post.cpp on a subset of the pairs): the pairs whose
// reads hold more hits together than PD_NMAX (their position sort would be a lane's serial work for milliseconds), and the pairs for which
// a pair score falls within 1e-6 of an integer before it is truncated (.721 log(2 erfc(|z| / sqrt 2)): the device's erfc / log are not
// glibc's to the last bit).  The kernel marks both in `todo`.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "bmh_internal.h"
#include "regs_core.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

#define PD_NMAX 64          // hits of a pair's two reads together the kernel takes (a lane's private arrays)

using namespace regs_core;

namespace {

struct pd_pes_t { int low, high, failed; double avg, std; };
struct pd_args_t {
	ctx_t x; bmh_pe_opt_t pe; pd_pes_t pes[4];
	int32_t *fin; const uint32_t *opr; const uint32_t *off; const float *frac_rep;
	uint32_t n_pairs; int64_t id0;
	int32_t *h_rec, *unflag; uint8_t *todo;
	int alt_mode;                   // the index has ALT contigs: a pair with a hit on one is the host's (todo = 3); the others leave as ALT-mode records ([11] = [12])
};

__device__ __forceinline__ int pd_infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)      // mem_infer_dir
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
__device__ __forceinline__ int pd_raw_mapq(int diff, int a)
{
#pragma clang fp contract(off)
	return (int)(6.02 * diff / a + .499);
}
__device__ __forceinline__ bool pd_lt(uint64_t ax, uint64_t ay, uint64_t bx, uint64_t by) { return ax < bx || (ax == bx && ay < by); }

__global__ void __launch_bounds__(64) pair_kernel(pd_args_t A)
{
#pragma clang fp contract(off)
	const uint32_t p = blockIdx.x * 64u + threadIdx.x;
	if (p >= A.n_pairs) return;
	const ctx_t &x = A.x;
	const uint32_t r0 = 2 * p;
	const int n[2] = {(int)A.opr[r0], (int)A.opr[r0 + 1]};
	rec_t *a[2] = {(rec_t *)(A.fin + 16 * (size_t)A.off[r0]), (rec_t *)(A.fin + 16 * (size_t)A.off[r0 + 1])};
	if (n[0] + n[1] > PD_NMAX) { A.todo[p] = 2; return; }
	// ALT contigs: the single-end tail ran with the table (ctx_t::alt_keep_sub_n): the hits of the primary assembly come first (n_pri of them: mem_pair and the
	// is_multi test look at those only, src/bwamem_pair.c:287-295), [15] carries is_alt (bit 1) and alt_sc, [12] is `secondary` of the second marking round and
	// secondary_all + 1 rides above the MAPQ in [13]
	int n_pri[2] = {n[0], n[1]};
	if (A.alt_mode) for (int i = 0; i < 2; ++i) { n_pri[i] = 0; for (int j = 0; j < n[i]; ++j) n_pri[i] += !(a[i][j].v[15] & 2); }
	const int64_t l_pac = x.l_pac;
	const uint64_t id = (uint64_t)(A.id0 / 2) + p;
	int z[2] = {0, 0}, o = 0, subo = 0, n_sub = 0, extra_flag = 1;
	bool paired = false, uncertain = false;
	if (!A.pe.no_pairing && n_pri[0] && n_pri[1]) {
		// ---- mem_pair: the hits of both reads by position
		uint64_t vx[PD_NMAX], vy[PD_NMAX];
		int nv = 0;
		for (int r = 0; r < 2; ++r)
			for (int i = 0; i < n_pri[r]; ++i) {
				const rec_t &e = a[r][i];
				const int64_t rb = r_rb(e), re = r_re(e);
				const int rid = pos2rid(x, rb < l_pac ? rb : (l_pac << 1) - 1 - (re - 1));
				uint64_t kx = (uint64_t)(rb < l_pac ? rb : (l_pac << 1) - 1 - rb);
				kx = (uint64_t)rid << 32 | (kx - (uint64_t)(x.n_contigs > 1 ? x.ctg_off[rid] : 0));
				const uint64_t ky = (uint64_t)e.v[1] << 32 | (uint64_t)(i << 2) | (uint64_t)((rb >= l_pac) << 1) | (uint64_t)r;
				int k = nv++;                                         // insertion sort: the keys are distinct (ky holds the hit)
				while (k > 0 && pd_lt(kx, ky, vx[k - 1], vy[k - 1])) { vx[k] = vx[k - 1]; vy[k] = vy[k - 1]; --k; }
				vx[k] = kx; vy[k] = ky;
			}
		// the pairs within an orientation's insert-size bounds: the best, the second best, then the runners-up close to the second best
		uint64_t bx = 0, by = 0, sx = 0, sy = 0; int nu = 0;
		for (int pass = 0; pass < 2; ++pass) {
			int y[4] = {-1, -1, -1, -1};
			const int tmp = mark_tmp(x);
			const int sub_q = nu > 1 ? (int)(sx >> 32) : 0;
			for (int i = 0; i < nv; ++i) {
				for (int r = 0; r < 2; ++r) {
					const int dir = r << 1 | (int)(vy[i] >> 1 & 1);
					if (A.pes[dir].failed) continue;
					const int which = r << 1 | (int)((vy[i] & 1) ^ 1);
					if (y[which] < 0) continue;
					for (int k = y[which]; k >= 0; --k) {
						if ((int)(vy[k] & 3) != which) continue;
						const int64_t dist = (int64_t)vx[i] - (int64_t)vx[k];
						if (dist > A.pes[dir].high) break;
						if (dist < A.pes[dir].low) continue;
						const double ns = (dist - A.pes[dir].avg) / A.pes[dir].std;
						const double val = (double)((vy[i] >> 32) + (vy[k] >> 32)) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * x.ep.a + .499;
						const double fr = val - floor(val);
						if (fr < 1e-6 || fr > 1. - 1e-6 || !(val == val)) uncertain = true;
						int q = (int)val;
						if (q < 0) q = 0;
						const uint64_t py = (uint64_t)k << 32 | (uint64_t)i;
						const uint64_t px = (uint64_t)q << 32 | (hash64(py ^ (uint64_t)((int)id << 8)) & 0xffffffffU);
						if (pass == 0) {
							++nu;
							if (nu == 1 || pd_lt(bx, by, px, py)) { sx = bx; sy = by; bx = px; by = py; if (nu == 1) { sx = 0; sy = 0; } }
							else if (nu == 2 || pd_lt(sx, sy, px, py)) { sx = px; sy = py; }
						} else if (!(px == bx && py == by) && sub_q - q <= tmp) ++n_sub;
					}
				}
				y[vy[i] & 3] = i;
			}
			if (nu <= 1) break;                                      // (the second pass counts the pairs beside the best one: none)
		}
		if (nu > 0) {
			const int i = (int)(by >> 32), k = (int)(by << 32 >> 32);
			z[vy[i] & 1] = (int)(vy[i] << 32 >> 34);
			z[vy[k] & 1] = (int)(vy[k] << 32 >> 34);
			o = (int)(bx >> 32);
			subo = nu > 1 ? (int)(sx >> 32) : 0;
		}
		if (o > 0) {
			int is_multi[2];
			for (int i = 0; i < 2; ++i) {
				int j;
				for (j = 1; j < n_pri[i]; ++j) if (a[i][j].v[12] < 0 && a[i][j].v[1] >= x.po.T) break;
				is_multi[i] = j < n_pri[i] ? 1 : 0;
			}
			if (!is_multi[0] && !is_multi[1]) {
				paired = true;
				int q_pe, q_se[2], err = 0;
				const int score_un = a[0][0].v[1] + a[1][0].v[1] - A.pe.pen_unpaired;
				subo = subo > score_un ? subo : score_un;
				q_pe = pd_raw_mapq(o - subo, x.ep.a);
				if (n_sub > 0) { if (n_sub + 1 >= x.n_log) { A.todo[p] = 2; return; } q_pe -= (int)(4.343 * x.logtab[n_sub + 1] + .499); }
				if (q_pe < 0) q_pe = 0;
				if (q_pe > 60) q_pe = 60;
				q_pe = (int)(q_pe * (1. - .5 * (A.frac_rep[r0] + A.frac_rep[r0 + 1])) + .499);
				if (o > score_un) {
					for (int i = 0; i < 2; ++i) {
						rec_t &cc = a[i][z[i]];
						if (cc.v[12] >= 0) cc.v[10] = a[i][cc.v[12]].v[1];       // (its secondary becomes -2 on the host: nothing that is written reads it)
						q_se[i] = approx_mapq(x, cc, A.frac_rep[r0 + i], &err);
					}
					q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
					q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
					extra_flag |= 2;
					for (int i = 0; i < 2; ++i) { const int cap = pd_raw_mapq(a[i][z[i]].v[1], x.ep.a); q_se[i] = q_se[i] < cap ? q_se[i] : cap; }   // (csub is 0 without a rescue)
				} else {
					z[0] = z[1] = 0;
					q_se[0] = approx_mapq(x, a[0][0], A.frac_rep[r0], &err);
					q_se[1] = approx_mapq(x, a[1][0], A.frac_rep[r0 + 1], &err);
				}
				// the best ALT hit of a read goes along as a supplementary record (src/bwamem_pair.c:349-356)
				int supp_q[2] = {-1, -1};
				if (A.alt_mode) {
					for (int i = 0; i < 2; ++i) {
						if (n_pri[i] >= n[i]) continue;
						const rec_t &q = a[i][n_pri[i]];
						if (!(q.v[1] < x.po.T || q.v[12] >= 0 || !(q.v[15] & 2))) supp_q[i] = approx_mapq(x, q, A.frac_rep[r0 + i], &err);
					}
					// from here on [12] is secondary_all, as in the records the host's mem_sam_pe leaves (`secondary` has been read for the last time)
					for (int i = 0; i < 2; ++i) for (int j = 0; j < n[i]; ++j) { a[i][j].v[12] = (a[i][j].v[13] >> 8) - 1; a[i][j].v[13] &= 0xFF; }
				}
				if (err) { A.todo[p] = 2; return; }
				for (int i = 0; i < 2; ++i) {                              // the chosen hit becomes the primary of its group (for the XA tag)
					const int k = a[i][z[i]].v[12];
					if (k >= 0 && k < n_pri[i]) {
						for (int j = 0; j < n[i]; ++j) if (a[i][j].v[12] == k || j == k) a[i][j].v[12] = z[i];
						a[i][z[i]].v[12] = -1;
					}
				}
				for (int i = 0; i < 2; ++i) {
					for (int j = 0; j < n[i]; ++j) { a[i][j].v[13] = 0; a[i][j].v[14] = 0; a[i][j].v[15] &= ~1; }     // (is_alt and alt_sc stay; without a table [15] is `reported` alone)
					rec_t &c = a[i][z[i]];
					c.v[15] |= 1; c.v[13] = q_se[i]; c.v[14] = 0x40 << i | extra_flag;
					if (supp_q[i] >= 0) { rec_t &q = a[i][n_pri[i]]; q.v[15] |= 1; q.v[13] = supp_q[i]; q.v[14] = 0x800 | 0x40 << i | extra_flag; }
					A.h_rec[r0 + i] = z[i]; A.unflag[r0 + i] = 0;
				}
			}
		}
	}
	if (uncertain) { A.todo[p] = 1; return; }                         // (what was written is overwritten by the host's records)
	if (!paired) {
		// the alignment a read shows its mate (src/bwamem_pair.c:376-385): its best hit, or -- when that one (the best of the primary assembly) is below the
		// threshold -- its best ALT hit; the orientation test below still takes the FIRST hits' positions (:389)
		int hh[2];
		for (int i = 0; i < 2; ++i) {
			hh[i] = -1;
			if (n[i]) {
				if (a[i][0].v[1] >= x.po.T) hh[i] = 0;
				else if (n_pri[i] < n[i] && a[i][n_pri[i]].v[1] >= x.po.T) hh[i] = n_pri[i];
			}
		}
		if (!A.pe.no_pairing && hh[0] >= 0 && hh[1] >= 0) {            // src/bwamem_pair.c:386
			const int64_t rb0 = r_rb(a[0][0]), rb1 = r_rb(a[1][0]);
			const int64_t hb0 = r_rb(a[0][hh[0]]), hb1 = r_rb(a[1][hh[1]]);
			const int rid0 = pos2rid(x, hb0 < l_pac ? hb0 : (l_pac << 1) - 1 - (r_re(a[0][hh[0]]) - 1)), rid1 = pos2rid(x, hb1 < l_pac ? hb1 : (l_pac << 1) - 1 - (r_re(a[1][hh[1]]) - 1));
			if (rid0 == rid1) {
				int64_t dist;
				const int d = pd_infer_dir(l_pac, rb0, rb1, &dist);
				if (!A.pes[d].failed && dist >= A.pes[d].low && dist <= A.pes[d].high) extra_flag |= 2;
			}
		}
		for (int i = 0; i < 2; ++i) {                                  // mem_reg2sam's selection is the single-end tail's; the pair flags go on top
			const int extra = (i ? 0x81 : 0x41) | extra_flag;
			bool any = false;
			for (int j = 0; j < n[i]; ++j) if (a[i][j].v[15] & 1) { a[i][j].v[14] |= extra; any = true; }
			A.h_rec[r0 + i] = hh[i]; A.unflag[r0 + i] = any ? 0 : extra;
		}
	}
	if (A.alt_mode) {                                               // ALT-mode records: [11] = [12] = secondary_all (the XA tag's key), [13] the MAPQ alone
		for (int i = 0; i < 2; ++i)
			for (int j = 0; j < n[i]; ++j) {
				rec_t &q = a[i][j];
				if (!paired) { q.v[12] = (q.v[13] >> 8) - 1; q.v[13] &= 0xFF; }
				q.v[11] = q.v[12];
			}
	}
	A.todo[p] = 0;
}

// ---- the records of the pairs the host walked take the place of the device's: the final arrays in read order
struct pm_args_t {
	const int32_t *fin_dev; const uint32_t *opr_dev, *off_dev; const int32_t *h_dev, *uf_dev;
	const int32_t *fin_host; const uint32_t *opr_host, *off_host; const int32_t *h_host, *uf_host;     // compact: the reads of the host's pairs, in their order
	const int32_t *slot;            // [n_reads] place of the read in the host's arrays or -1
	uint32_t n_reads;
	uint32_t *opr; const uint32_t *off; int32_t *fin, *h, *uf;
};
__global__ void __launch_bounds__(256) pair_scatter_slot_kernel(const uint32_t *todo_pairs, uint32_t n_todo, int32_t *slot)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t >= n_todo) return;
	slot[2 * todo_pairs[t]] = (int32_t)(2 * t); slot[2 * todo_pairs[t] + 1] = (int32_t)(2 * t + 1);
}
__global__ void __launch_bounds__(256) pair_counts_kernel(pm_args_t A)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r >= A.n_reads) return;
	const int32_t s = A.slot[r];
	A.opr[r] = s >= 0 ? A.opr_host[s] : A.opr_dev[r];
	A.h[r] = s >= 0 ? A.h_host[s] : A.h_dev[r];
	A.uf[r] = s >= 0 ? A.uf_host[s] : A.uf_dev[r];
}
__global__ void __launch_bounds__(256) pair_merge_kernel(pm_args_t A)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x, r = t >> 4, l = t & 15u;      // sixteen lanes a read, a quarter record each
	if (r >= A.n_reads) return;
	const int32_t s = A.slot[r];
	const uint32_t n = A.opr[r];
	const int4 *src = (const int4 *)(s >= 0 ? A.fin_host + 16 * (size_t)A.off_host[s] : A.fin_dev + 16 * (size_t)A.off_dev[r]);
	int4 *dst = (int4 *)(A.fin + 16 * (size_t)A.off[r]);
	for (uint32_t k = l; k < 4 * n; k += 16) dst[k] = src[k];
}

}   // namespace

// pes[4][5] = {low, high, failed, avg, std} (bmh_finalize_pairs' pes_out); d_fin [m][16] / d_opr / d_off: the records of bmh_finalize_regs_device, their
// numbers per read and the first record of every read; d_logtab: log(k) of the host's libm for k < n_log.  Asynchronous.
int bmh_pair_device(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe, const double *pes, int64_t l_pac,
                    int n_contigs, const int64_t *d_ctg_off, const double *d_logtab, int n_log, int32_t *d_fin, const uint32_t *d_opr, const uint32_t *d_off,
                    const float *d_frac_rep, uint32_t n_reads, int32_t *d_h_rec, int32_t *d_unflag, uint8_t *d_todo, void *stream)
{
	if (n_reads == 0) return BMH_OK;
	pd_args_t A;
	memset(&A, 0, sizeof(A));
	A.x.co = *copt; A.x.ep = *ep; A.x.po = *popt; A.x.po.contig_is_alt = nullptr; A.x.co.contig_is_alt = nullptr; A.x.l_pac = l_pac;
	A.x.n_contigs = n_contigs > 1 ? n_contigs : 1; A.x.ctg_off = n_contigs > 1 ? d_ctg_off : nullptr; A.x.logtab = d_logtab; A.x.n_log = n_log;
	A.pe = *pe;
	for (int d = 0; d < 4; ++d) { A.pes[d].low = (int)pes[5 * d]; A.pes[d].high = (int)pes[5 * d + 1]; A.pes[d].failed = (int)pes[5 * d + 2]; A.pes[d].avg = pes[5 * d + 3]; A.pes[d].std = pes[5 * d + 4]; }
	A.fin = d_fin; A.opr = d_opr; A.off = d_off; A.frac_rep = d_frac_rep; A.n_pairs = n_reads / 2; A.id0 = popt->id0;
	A.h_rec = d_h_rec; A.unflag = d_unflag; A.todo = d_todo; A.alt_mode = popt->contig_is_alt != nullptr;
	pair_kernel<<<(A.n_pairs + 63) / 64, 64, 0, (hipStream_t)stream>>>(A);
	HIPCK(hipGetLastError());
	return BMH_OK;
}

int bmh_pair_limit(void) { return PD_NMAX; }

// d_slot [n_reads] int32 scratch; the host's compact arrays (already on the device): d_todo_pairs [n_todo], d_fin_host, d_opr_host / d_off_host / d_h_host /
// d_uf_host [2 n_todo].  First call: counts (d_opr, d_h, d_uf final); the caller scans d_opr into d_off; second call: the records.  Asynchronous.
int bmh_pair_merge_counts(uint32_t n_reads, const uint32_t *d_todo_pairs, uint32_t n_todo, int32_t *d_slot, const uint32_t *d_opr_dev, const int32_t *d_h_dev, const int32_t *d_uf_dev,
                          const uint32_t *d_opr_host, const int32_t *d_h_host, const int32_t *d_uf_host, uint32_t *d_opr, int32_t *d_h, int32_t *d_uf, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	if (n_reads == 0) return BMH_OK;
	HIPCK(hipMemsetAsync(d_slot, 0xFF, 4 * (size_t)n_reads, st));
	if (n_todo) pair_scatter_slot_kernel<<<(n_todo + 255) / 256, 256, 0, st>>>(d_todo_pairs, n_todo, d_slot);
	pm_args_t A;
	memset(&A, 0, sizeof(A));
	A.opr_dev = d_opr_dev; A.h_dev = d_h_dev; A.uf_dev = d_uf_dev; A.opr_host = d_opr_host; A.h_host = d_h_host; A.uf_host = d_uf_host; A.slot = d_slot; A.n_reads = n_reads;
	A.opr = d_opr; A.h = d_h; A.uf = d_uf;
	pair_counts_kernel<<<(n_reads + 255) / 256, 256, 0, st>>>(A);
	HIPCK(hipGetLastError());
	return BMH_OK;
}
int bmh_pair_merge_records(uint32_t n_reads, const int32_t *d_slot, const int32_t *d_fin_dev, const uint32_t *d_off_dev, const int32_t *d_fin_host, const uint32_t *d_off_host,
                           const uint32_t *d_opr, const uint32_t *d_off, int32_t *d_fin, void *stream)
{
	if (n_reads == 0) return BMH_OK;
	pm_args_t A;
	memset(&A, 0, sizeof(A));
	A.fin_dev = d_fin_dev; A.off_dev = d_off_dev; A.fin_host = d_fin_host; A.off_host = d_off_host; A.slot = d_slot; A.n_reads = n_reads;
	A.opr = (uint32_t *)d_opr; A.off = d_off; A.fin = d_fin;
	pair_merge_kernel<<<(unsigned)(((size_t)n_reads * 16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(A);
	HIPCK(hipGetLastError());
	return BMH_OK;
}

// an exclusive scan of n words for the callers above (d_tmp: bmh_pair_scan_bytes(n) bytes of device memory)
size_t bmh_pair_scan_bytes(uint32_t n)
{
	size_t t = 0;
	(void)rocprim::exclusive_scan(nullptr, t, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), 0);
	return t + 256;
}
int bmh_pair_scan(const uint32_t *d_in, uint32_t *d_out, uint32_t n, void *d_tmp, size_t tmp_bytes, void *stream)
{
	if (n == 0) return BMH_OK;
	size_t tb = tmp_bytes;
	HIPCK(rocprim::exclusive_scan(d_tmp, tb, d_in, d_out, 0u, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t)stream));
	return BMH_OK;
}
