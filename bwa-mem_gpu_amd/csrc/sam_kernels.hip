// What the SAM writer needs of a batch, chosen and packed ON THE DEVICE (the host side of gase_aln around the
// device path).  Until round 4 the records of the region tail went to the host, one host pass chose the records that need a CIGAR
// (bmh_sam_need_cigar: the reported ones and the XA candidates of mem_gen_alt, /root/reference/src/bwamem_extra.c:97-150), the list went
// back to the device, and the CIGAR / MD arrays came home as fixed slots of 64 + 96 bytes an alignment of which a typical alignment
// uses 4 + 4.  Both are a few microseconds of device work:
//   * bmh_sam_select_device: the same selection, one lane per read, over the records where bmh_finalize_regs_device left them;
//   * bmh_cigar_pack_sizes / bmh_cigar_pack: the operations and the MD string of every alignment back to back in one array of words.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdint>
#include <cstdlib>
#include "bmh_internal.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

namespace {

struct sel_args_t {
	const int32_t *fin; const uint32_t *opr; const uint32_t *off; const int32_t *h_rec;
	uint32_t n_reads; uint32_t *cnt, *flag;
	int flag_all, sa, max_XA_hits, max_XA_hits_alt; double drop;
};

// bmh_sam_need_cigar (csrc/sam_format.cpp), one RECORD per lane (a record carries its read in field [0]; a lane per read spent its time in the
// serial loops of the few reads with hundreds of records): first cnt[k] = hits listed under primary k (bit 31: one of them on an ALT contig),
// then need = reported, or an XA candidate whose primary lists no more hits than the XA limits allow (src/bwamem_extra.c:125)
__device__ __forceinline__ int sel_pri(const sel_args_t &A, const int32_t *a, int i)
{
	const int k = a[16 * i + A.sa];
	return (k >= 0 && (double)a[16 * i + 1] >= (double)a[16 * k + 1] * A.drop) ? k : -1;
}
__global__ void __launch_bounds__(256) sam_count_kernel(sel_args_t A, uint64_t m)
{
	const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (g >= m) return;
	const uint32_t r = (uint32_t)A.fin[16 * g];
	const uint64_t base = A.off[r];
	const int k = sel_pri(A, A.fin + 16 * base, (int)(g - base));
	if (k >= 0) {
		atomicAdd(A.cnt + base + (uint64_t)k, 1u);
		if (A.fin[16 * g + 15] & 2) atomicOr(A.cnt + base + (uint64_t)k, 0x80000000u);
	}
}
__global__ void __launch_bounds__(256) sam_select_kernel(sel_args_t A, uint64_t m)
{
	const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (g >= m) return;
	const uint32_t r = (uint32_t)A.fin[16 * g];
	const uint64_t base = A.off[r];
	const int i = (int)(g - base);
	uint32_t need = (A.fin[16 * g + 15] & 1) ? 1u : 0u;
	if (!A.flag_all) {
		const int k = sel_pri(A, A.fin + 16 * base, i);
		if (k >= 0) {
			const uint32_t c = A.cnt[base + (uint64_t)k];
			const int n = (int)(c & 0x7FFFFFFFu); const bool has_alt = c >> 31;
			if (!(n > A.max_XA_hits_alt || (!has_alt && n > A.max_XA_hits))) need = 1u;
		}
	}
	if (A.h_rec && A.h_rec[r] == i) need = 1u;                   // pairs: the read's own alignment lends its mate the mate fields
	A.flag[g] = need;
}

__global__ void __launch_bounds__(256) sam_sel_scatter_kernel(const uint32_t *flag, const uint32_t *pos, uint64_t m, uint32_t *sel, int32_t *slot, uint32_t *total)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (i >= m) return;
	const uint32_t f = flag[i], p = pos[i];
	slot[i] = f ? (int32_t)p : -1;
	if (f) sel[p] = (uint32_t)i;
	if (i == m - 1) *total = p + f;
}

// words an alignment takes in the packed array: its operations, then its MD string with the NUL, padded to a word; an alignment that overflowed
// the fixed slots (flags 1, 8: the caller redoes it with larger ones) or was rejected takes none
__global__ void __launch_bounds__(256) cigar_words_kernel(const int32_t *aln, uint32_t n, int with_md, uint32_t *words)
{
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	if (k > n) return;
	uint32_t w = 0;
	if (k < n) {
		const int32_t *a = aln + 8 * (size_t)k;
		if (!(a[7] & ~2)) w = (uint32_t)a[3] + (with_md ? ((uint32_t)a[6] + 4u) >> 2 : 0u);
	}
	words[k] = w;
}

__global__ void __launch_bounds__(256) cigar_pack_kernel(const int32_t *aln, const uint32_t *cigar, int max_cigar, const char *md, int md_cap, uint32_t n,
                                                         const uint32_t *off, uint32_t *packed)
{
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	if (k >= n) return;
	const int32_t *a = aln + 8 * (size_t)k;
	if (a[7] & ~2) return;
	uint32_t *o = packed + off[k];
	const int nc = a[3];
	const uint32_t *cg = cigar + (size_t)max_cigar * k;
	for (int i = 0; i < nc; ++i) o[i] = cg[i];
	if (md) {
		const uint32_t *ms = (const uint32_t *)(md + (size_t)md_cap * k);       // (md_cap is a multiple of 4: checked by the host)
		const int nw = (a[6] + 4) >> 2;
		for (int i = 0; i < nw; ++i) o[nc + i] = ms[i];
	}
}

size_t scan_bytes(size_t n)
{
	size_t t = 0;
	(void)rocprim::exclusive_scan(nullptr, t, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n, rocprim::plus<uint32_t>(), 0);
	return (t + 255) & ~(size_t)255;
}
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}   // namespace

extern "C" size_t bmh_sam_select_work(uint32_t n_reads, uint64_t m)
{
	return al256(4 * ((size_t)n_reads + 1)) + 3 * al256(4 * (size_t)(m + 1)) + scan_bytes((size_t)(m > n_reads ? m : n_reads) + 1) + 512;
}

extern "C" int64_t bmh_sam_select_device(const bmh_post_opt_t *popt, const int32_t *d_fin, const uint32_t *d_fin_per_read, const int32_t *d_h_rec,
                                         uint32_t n_reads, uint64_t m, uint32_t *d_sel, int32_t *d_slot, void *d_work, size_t work_bytes, void *stream_)
{
	if (!popt || (n_reads && !d_fin_per_read) || (m && (!d_fin || !d_sel || !d_slot)) || !d_work) { bmh_set_error("bmh_sam_select_device: null argument"); return BMH_EINVAL; }
	if (work_bytes < bmh_sam_select_work(n_reads, m)) { bmh_set_error("bmh_sam_select_device: %zu bytes of work space, bmh_sam_select_work asks for %zu", work_bytes, bmh_sam_select_work(n_reads, m)); return BMH_EINVAL; }
	if (m >> 31) { bmh_set_error("bmh_sam_select_device: 2^31 records or more in one batch"); return BMH_ECAPACITY; }
	if (n_reads == 0 || m == 0) return 0;
	hipStream_t st = (hipStream_t)stream_;
	uint8_t *w = (uint8_t *)d_work;
	uint32_t *off = (uint32_t *)w; w += al256(4 * ((size_t)n_reads + 1));
	uint32_t *cnt = (uint32_t *)w; w += al256(4 * (size_t)(m + 1));
	uint32_t *flag = (uint32_t *)w; w += al256(4 * (size_t)(m + 1));
	uint32_t *pos = (uint32_t *)w; w += al256(4 * (size_t)(m + 1));
	uint32_t *total = (uint32_t *)w; w += 256;
	void *tmp = w;
	size_t tb = scan_bytes((size_t)(m > n_reads ? m : n_reads) + 1);
	HIPCK(rocprim::exclusive_scan(tmp, tb, d_fin_per_read, off, 0u, (size_t)n_reads, rocprim::plus<uint32_t>(), st));
	sel_args_t A;
	A.fin = d_fin; A.opr = d_fin_per_read; A.off = off; A.h_rec = d_h_rec; A.n_reads = n_reads; A.cnt = cnt; A.flag = flag;
	A.flag_all = popt->flag_all; A.sa = popt->contig_is_alt ? 11 : 12; A.max_XA_hits = popt->max_XA_hits; A.max_XA_hits_alt = popt->max_XA_hits_alt;
	A.drop = (double)popt->XA_drop_ratio;
	// (field [0] of a record is its read's index in the batch: bmh_finalize_regs*, bmh_finalize_pairs*)
	HIPCK(hipMemsetAsync(cnt, 0, 4 * (size_t)m, st));
	if (!A.flag_all) sam_count_kernel<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(A, m);
	sam_select_kernel<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(A, m);
	tb = scan_bytes((size_t)(m > n_reads ? m : n_reads) + 1);
	HIPCK(rocprim::exclusive_scan(tmp, tb, flag, pos, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
	sam_sel_scatter_kernel<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(flag, pos, m, d_sel, d_slot, total);
	uint32_t h = 0;
	HIPCK(hipMemcpyAsync(&h, total, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	return (int64_t)h;
}

extern "C" size_t bmh_cigar_pack_work(uint32_t n) { return al256(4 * ((size_t)n + 2)) + scan_bytes((size_t)n + 2) + 256; }

extern "C" int64_t bmh_cigar_pack_sizes(const int32_t *d_aln, uint32_t n, int with_md, uint32_t *d_off, void *d_work, size_t work_bytes, void *stream_)
{
	if ((n && !d_aln) || !d_off || !d_work) { bmh_set_error("bmh_cigar_pack_sizes: null argument"); return BMH_EINVAL; }
	if (work_bytes < bmh_cigar_pack_work(n)) { bmh_set_error("bmh_cigar_pack_sizes: %zu bytes of work space, bmh_cigar_pack_work asks for %zu", work_bytes, bmh_cigar_pack_work(n)); return BMH_EINVAL; }
	hipStream_t st = (hipStream_t)stream_;
	uint8_t *w = (uint8_t *)d_work;
	uint32_t *words = (uint32_t *)w; w += al256(4 * ((size_t)n + 2));
	cigar_words_kernel<<<(n + 1 + 255) / 256, 256, 0, st>>>(d_aln, n, with_md, words);
	size_t tb = scan_bytes((size_t)n + 2);
	HIPCK(rocprim::exclusive_scan((void *)w, tb, words, d_off, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st));      // d_off[n] = the total
	uint32_t h = 0;
	HIPCK(hipMemcpyAsync(&h, d_off + n, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	return (int64_t)h;
}

extern "C" int bmh_cigar_pack(const int32_t *d_aln, const uint32_t *d_cigar, int max_cigar, const char *d_md, int md_cap, uint32_t n, const uint32_t *d_off,
                              uint32_t *d_packed, void *stream_)
{
	if (n == 0) return BMH_OK;
	if (!d_aln || !d_cigar || !d_off || !d_packed || max_cigar < 1 || (d_md && (md_cap < 4 || (md_cap & 3)))) { bmh_set_error("bmh_cigar_pack: bad argument (md_cap must be a multiple of 4)"); return BMH_EINVAL; }
	cigar_pack_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(d_aln, d_cigar, max_cigar, d_md, md_cap, n, d_off, d_packed);
	HIPCK(hipGetLastError());
	return BMH_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The SAM text itself on the device: bmh_format_sam / bmh_format_sam_pe (csrc/sam_format.cpp; mem_aln2sam, /root/reference/src/bwamem.c:
// 1506-1683; mem_gen_alt, src/bwamem_extra.c:97-150), one read per lane, two passes over the same code: the first counts the bytes of every
// read's records, a scan places them, the second writes.  On the host the text of a million reads was 0.9 s of CPU time (14 ms on 64
// threads) and needed the records, the alignments and the CIGAR / MD arrays in host memory: 300 MB of copies for 230 MB of text.  Here the
// text is the only thing that leaves the device.  Without ALT contigs (their soft clips and the pa:f tag stay with the host formatter).
namespace {

struct sam_args_t {
	bmh_sam_dev_t d;
	const uint32_t *rec_off;       // [n_reads] first record of every read
	uint32_t *len; const uint64_t *text_off; char *text; uint32_t *err;
	int flag_all, max_XA_hits, max_XA_hits_alt, no_multi, softclip; double drop;
	int sa;                        // field of a record that holds the XA tag's key: [12] secondary, with ALT contigs [11] secondary_all (bmh_post_opt_t)
	int rg_len; char rg[256];      // read group id (bmh_post_opt_t::rg_id), 0 = none
};

// where the text goes: W = 0 counts the bytes, 1 writes them to global memory, 2 into the wave's image in LDS (copied out in dwords afterwards:
// a lane's byte stores to global memory are one memory transaction each, 230 M of them for a million reads)
typedef __attribute__((address_space(3))) char sam_lds_char;
template <int W> struct sam_ptr_t { typedef char *type; };
template <> struct sam_ptr_t<2> { typedef sam_lds_char *type; };
template <int W> struct sam_out_t {
	typename sam_ptr_t<W>::type p; uint32_t n;
	__device__ __forceinline__ void ch(char c) { if (W) *p++ = c; else ++n; }
	__device__ __forceinline__ void str(const char *s, int len) { if (W) { for (int i = 0; i < len; ++i) *p++ = s[i]; } else n += (uint32_t)len; }   // len bytes, from global memory
	template <int N> __device__ __forceinline__ void lit(const char (&s)[N]) { for (int i = 0; i < N - 1; ++i) ch(s[i]); }
	__device__ void num(long long v)                                                             // put_int
	{
		unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
		if (v < 0) ch('-');
		int nd = 1;
		for (unsigned long long t = u; t >= 10; t /= 10) ++nd;
		if (W) { typename sam_ptr_t<W>::type q = p + nd; do { *--q = (char)('0' + u % 10); u /= 10; } while (u); p += nd; }
		else n += (uint32_t)nd;
	}
};

// a / b as printf("%.3f") writes it (the pa:f tag, src/bwamem.c:1663): the double nearest to a / b, rounded to three decimals from its EXACT value, ties to even
// -- m x 2^e x 1000 in integers
template <int W> __device__ void sam_fmt3(sam_out_t<W> &o, int a, int b)
{
#pragma clang fp contract(off)
	const double x = (double)a / (double)b;
	unsigned long long q = 0;
	if (x > 0.) {
		const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
		const int ex = (int)(bits >> 52 & 0x7FF);
		const unsigned long long m = ex ? (bits & 0xFFFFFFFFFFFFFull) | 1ull << 52 : (bits & 0xFFFFFFFFFFFFFull);
		const int e = (ex ? ex : 1) - 1075;                         // x = m * 2^e
		const unsigned long long N = m * 1000ull;                   // < 2^63
		if (e >= 0) q = N << e;                                     // (not reached for a ratio of two scores)
		else if (-e >= 64) q = 0;
		else {
			const int sft = -e;
			q = N >> sft;
			const unsigned long long rem = N & ((1ull << sft) - 1), half = 1ull << (sft - 1);
			if (rem > half || (rem == half && (q & 1))) ++q;
		}
	}
	o.num((long long)(q / 1000));
	o.ch('.');
	const int f = (int)(q % 1000);
	o.ch((char)('0' + f / 100)); o.ch((char)('0' + f / 10 % 10)); o.ch((char)('0' + f % 10));
}

struct sam_rec_t { const int32_t *fin; const int32_t *aln; const uint32_t *cigar; const char *md; };

__device__ __forceinline__ sam_rec_t sam_rec(const sam_args_t &A, uint64_t base, int i)
{
	sam_rec_t x; x.fin = A.d.d_fin + 16 * (base + (uint64_t)i);
	const int32_t s = A.d.d_slot[base + (uint64_t)i];
	x.aln = nullptr; x.cigar = nullptr; x.md = nullptr;
	if (s >= 0) {
		x.aln = A.d.d_aln + 8 * (size_t)s;
		if (x.aln[7] & ~2) { x.aln = nullptr; return x; }       // (an alignment bmh_cigar_batch flagged has no words in the packed array)
		x.cigar = A.d.d_packed + A.d.d_cig_off[s]; x.md = (const char *)(x.cigar + x.aln[3]);
	}
	return x;
}
__device__ __forceinline__ long long sam_pos(const int32_t *a) { return (long long)(uint32_t)a[0] | (long long)a[1] << 32; }
__device__ __forceinline__ int sam_rid(const sam_args_t &A, long long pos)
{
	if (A.d.n_contigs <= 1) return 0;
	int lo = 0, hi = A.d.n_contigs;
	while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.d.d_contig_offset[mid] <= pos) lo = mid; else hi = mid; }
	return lo;
}
__device__ __forceinline__ long long sam_ctg0(const sam_args_t &A, int rid) { return A.d.n_contigs > 1 ? A.d.d_contig_offset[rid] : 0; }
__device__ __forceinline__ const char *sam_ctg(const sam_args_t &A, int rid) { return A.d.d_contig_names + A.d.d_contig_name_off[rid]; }
__device__ __forceinline__ int sam_ctg_len(const sam_args_t &A, int rid) { return (int)(A.d.d_contig_name_off[rid + 1] - A.d.d_contig_name_off[rid]) - 1; }
template <int W> __device__ void sam_cigar(sam_out_t<W> &o, const sam_rec_t &r, bool hard)
{
	const int n = r.aln[3];
	for (int i = 0; i < n; ++i) {
		int c = (int)(r.cigar[i] & 0xf);
		if (hard && (c == 3 || c == 4)) c = 4;
		o.num(r.cigar[i] >> 4);
		o.ch("MIDSH"[c]);
	}
}
__device__ __forceinline__ int sam_ref_len(int n, const uint32_t *cg)
{
	int l = 0;
	for (int k = 0; k < n; ++k) { const int op = (int)(cg[k] & 0xf); if (op == 0 || op == 2) l += (int)(cg[k] >> 4); }
	return l;
}
__device__ __forceinline__ int sam_nt4(uint8_t c)            // nst_nt4_table as far as the text needs it: A C G T in either case, everything else N
{
	c &= 0xDF;
	return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}
// the letter SEQ shows for a read's letter: upper case, N for everything that is not A C G T; complemented on the reverse strand
__device__ __forceinline__ char sam_letter(uint8_t c, bool rev)
{
	c &= 0xDF;
	const bool a = c == 'A', cc = c == 'C', g = c == 'G', t = c == 'T';
	if (!rev) return (a || cc || g || t) ? (char)c : 'N';
	return a ? 'T' : cc ? 'G' : g ? 'C' : t ? 'A' : 'N';
}
template <int W> __device__ void sam_seq(sam_out_t<W> &o, const uint8_t *seq, int qb, int qe, bool rev)
{
	if (qe <= qb) return;
	if (!W) { o.n += (uint32_t)(qe - qb); return; }
	// (eight letters in flight at a time, and the whole wave copying read after read in coalesced steps, were both tried: the writing pass is bound
	// by its two waves per SIMD -- 16 KB of LDS per wave -- and the dependent loads of the records, not by these bytes)
	if (!rev) for (int k = qb; k < qe; ++k) *o.p++ = sam_letter(seq[k], false);
	else for (int k = qe - 1; k >= qb; --k) *o.p++ = sam_letter(seq[k], true);
}

struct sam_mate_t { int rid; long long pos; int is_rev, n_cigar; const uint32_t *cigar; };

template <int W> __device__ void sam_mate_fields(const sam_args_t &A, sam_out_t<W> &o, bool pe, int p_rid, long long p_pos, int p_rev, int p_ncig, const uint32_t *p_cig,
                                                  bool mate_mapped, int m_rid, long long m_pos, int m_rev, int m_ncig, const uint32_t *m_cig)
{
	if (pe && mate_mapped) {
		if (p_rid == m_rid) o.ch('='); else o.str(sam_ctg(A, m_rid), sam_ctg_len(A, m_rid));
		o.ch('\t'); o.num(m_pos - sam_ctg0(A, m_rid) + 1); o.ch('\t');
		if (p_rid == m_rid) {
			const long long p0 = p_pos + (p_rev ? sam_ref_len(p_ncig, p_cig) - 1 : 0), p1 = m_pos + (m_rev ? sam_ref_len(m_ncig, m_cig) - 1 : 0);
			if (m_ncig == 0 || p_ncig == 0) o.ch('0');
			else o.num(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else o.ch('0');
	} else o.lit("*\t0\t0");
	o.ch('\t');
}

// the records of read r (the body of bmh_format_sam_parts' loop); false: a record the text needs has no alignment
template <int W> __device__ bool sam_read(const sam_args_t &A, uint32_t r, sam_out_t<W> &out)
{
	const uint64_t base = A.rec_off[r];
	const int n = (int)A.d.d_fin_per_read[r];
	const int32_t *a = A.d.d_fin + 16 * base;
	const bool pe = A.d.d_h_rec != nullptr;
	sam_mate_t m; m.rid = -1; m.pos = 0; m.is_rev = 0; m.n_cigar = 0; m.cigar = nullptr;
	if (pe) {
		const uint32_t mr = r ^ 1u;
		const int h = A.d.d_h_rec[mr];
		if (h >= 0) {
			const sam_rec_t y = sam_rec(A, A.rec_off[mr], h);
			if (!y.aln) return false;
			m.pos = sam_pos(y.aln); m.rid = sam_rid(A, m.pos); m.is_rev = y.aln[2]; m.n_cigar = y.aln[3]; m.cigar = y.cigar;
		}
	}
	const char *name = A.d.d_names + A.d.d_name_off[r];
	const int name_len = (int)(A.d.d_name_off[r + 1] - A.d.d_name_off[r]) - 1;     // (names are NUL-terminated back to back: no strlen)
	const uint8_t *seq = A.d.d_reads + A.d.d_offs[r];
	const int l_seq = (int)A.d.d_lens[r];
	int n_rep = 0;
	for (int i = 0; i < n; ++i) n_rep += a[16 * i + 15] & 1;
	if (n_rep == 0) {                                           // unmapped record (mem_reg2sam's aa.n == 0 branch)
		int flag = 4 | (A.d.d_unflag ? A.d.d_unflag[r] : 0);
		const bool mm = pe && m.rid >= 0;
		if (pe && m.rid < 0) flag |= 8;
		const int p_rev = mm ? m.is_rev : 0;
		if (p_rev) flag |= 0x10;
		if (mm && m.is_rev) flag |= 0x20;
		out.str(name, name_len); out.ch('\t'); out.num(flag); out.ch('\t');
		if (mm) { out.str(sam_ctg(A, m.rid), sam_ctg_len(A, m.rid)); out.ch('\t'); out.num(m.pos - sam_ctg0(A, m.rid) + 1); out.lit("\t0\t*\t"); }
		else out.lit("*\t0\t0\t*\t");
		sam_mate_fields<W>(A, out, pe, mm ? m.rid : -1, m.pos, p_rev, 0, nullptr, mm, m.rid, m.pos, m.is_rev, m.n_cigar, m.cigar);
		sam_seq<W>(out, seq, 0, l_seq, p_rev != 0);
		out.lit("\t*\tAS:i:0\tXS:i:0");
		if (A.rg_len) { out.lit("\tRG:Z:"); out.str(A.rg, A.rg_len); }
		out.ch('\n');
		return true;
	}
	auto pri = [&](int i) { const int k = a[16 * i + A.sa]; return (k >= 0 && (double)a[16 * i + 1] >= (double)a[16 * k + 1] * A.drop) ? k : -1; };
	int which = 0;
	for (int i = 0; i < n; ++i) {
		if (!(a[16 * i + 15] & 1)) continue;
		const sam_rec_t x = sam_rec(A, base, i);
		if (!x.aln) return false;
		const long long pos = sam_pos(x.aln);
		const int rid = sam_rid(A, pos);
		const bool mate_mapped = pe && m.rid >= 0;
		const int m_rid = mate_mapped ? m.rid : rid; const long long m_pos = mate_mapped ? m.pos : pos; const int m_rev = mate_mapped ? m.is_rev : (x.aln[2] ? 1 : 0);
		int flag = (x.aln[2] ? 0x10 : 0) | x.fin[14];
		if (pe) { if (m.rid < 0) flag |= 8; if (m_rev) flag |= 0x20; }
		const bool hard = which > 0 && !A.softclip && !(x.fin[15] & 2);
		out.str(name, name_len); out.ch('\t'); out.num((flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)); out.ch('\t');
		out.str(sam_ctg(A, rid), sam_ctg_len(A, rid)); out.ch('\t'); out.num(pos - sam_ctg0(A, rid) + 1); out.ch('\t');
		out.num(x.fin[13]); out.ch('\t');
		if (x.aln[3]) sam_cigar<W>(out, x, hard); else out.ch('*');
		out.ch('\t');
		sam_mate_fields<W>(A, out, pe, rid, pos, x.aln[2] ? 1 : 0, x.aln[3], x.cigar, pe, m_rid, m_pos, m_rev, mate_mapped ? m.n_cigar : 0, mate_mapped ? m.cigar : nullptr);
		if (flag & 0x100) out.lit("*\t*");
		else {
			int qb = 0, qe = l_seq;
			const int nc = x.aln[3];
			if (nc && hard) {
				const int c0 = (int)(x.cigar[0] & 0xf), c1 = (int)(x.cigar[nc - 1] & 0xf);
				if (!x.aln[2]) { if (c0 == 3 || c0 == 4) qb += x.cigar[0] >> 4; if (c1 == 3 || c1 == 4) qe -= x.cigar[nc - 1] >> 4; }
				else { if (c0 == 3 || c0 == 4) qe -= x.cigar[0] >> 4; if (c1 == 3 || c1 == 4) qb += x.cigar[nc - 1] >> 4; }
			}
			sam_seq<W>(out, seq, qb, qe, x.aln[2] != 0);
			out.lit("\t*");
		}
		if (x.aln[3]) { out.lit("\tNM:i:"); out.num(x.aln[4]); out.lit("\tMD:Z:"); out.str(x.md, x.aln[6]); }
		if (x.fin[1] >= 0) { out.lit("\tAS:i:"); out.num(x.fin[1]); }
		if (!(flag & 0x100) && x.fin[10] >= 0) { out.lit("\tXS:i:"); out.num(x.fin[10]); }
		if (A.rg_len) { out.lit("\tRG:Z:"); out.str(A.rg, A.rg_len); }      // src/bwamem.c:1631-1634
		if (!(flag & 0x100)) {
			bool other = false;
			for (int j = 0; j < n; ++j) if (j != i && (a[16 * j + 15] & 1) && !(a[16 * j + 14] & 0x100)) other = true;
			if (other) {
				out.lit("\tSA:Z:");
				for (int j = 0; j < n; ++j) {
					if (j == i || !(a[16 * j + 15] & 1) || (a[16 * j + 14] & 0x100)) continue;
					const sam_rec_t y = sam_rec(A, base, j);
					if (!y.aln) return false;
					const long long p2 = sam_pos(y.aln);
					const int rid2 = sam_rid(A, p2);
					out.str(sam_ctg(A, rid2), sam_ctg_len(A, rid2)); out.ch(','); out.num(p2 - sam_ctg0(A, rid2) + 1); out.ch(',');
					out.ch("+-"[y.aln[2] ? 1 : 0]); out.ch(',');
					sam_cigar<W>(out, y, false);
					out.ch(','); out.num(y.fin[13]); out.ch(','); out.num(y.aln[4]); out.ch(';');
				}
			}
		}
		if (!(flag & 0x100) && (x.fin[15] >> 2) > 0) { out.lit("\tpa:f:"); sam_fmt3<W>(out, x.fin[1], x.fin[15] >> 2); }      // score / score of the ALT hit that shadows it
		if (!A.flag_all) {                                       // the XA tag of this record: the hits listed under it (mem_gen_alt)
			int cnt = 0; bool has_alt = false;
			for (int j = 0; j < n; ++j) if (pri(j) == i) { ++cnt; has_alt = has_alt || (a[16 * j + 15] & 2); }
			if (cnt > 0 && !(cnt > A.max_XA_hits_alt || (!has_alt && cnt > A.max_XA_hits))) {         // src/bwamem_extra.c:125
				out.lit("\tXA:Z:");
				for (int j = 0; j < n; ++j) {
					if (pri(j) != i) continue;
					const sam_rec_t y = sam_rec(A, base, j);
					if (!y.aln) return false;
					const long long p2 = sam_pos(y.aln);
					const int rid2 = sam_rid(A, p2);
					out.str(sam_ctg(A, rid2), sam_ctg_len(A, rid2)); out.ch(','); out.ch("+-"[y.aln[2] ? 1 : 0]); out.num(p2 - sam_ctg0(A, rid2) + 1); out.ch(',');
					sam_cigar<W>(out, y, false);
					out.ch(','); out.num(y.aln[4]); out.ch(';');
				}
			}
		}
		out.ch('\n');
		++which;
	}
	return true;
}

// first pass: the bytes of every read's records
__global__ void __launch_bounds__(256) sam_text_count_kernel(sam_args_t A)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r >= A.d.n_reads) return;
	sam_out_t<0> o; o.p = nullptr; o.n = 0;
	const bool ok = sam_read<0>(A, r, o);
	if (!ok) atomicOr(A.err, 1u);
	A.len[r] = ok ? o.n : 0u;
}

// second pass: a wave writes the records of its 64 reads -- one piece of the text -- into LDS, every lane its read's, and copies the piece out in
// dwords.  A piece beyond the wave's share of LDS (300 bp reads: 64 records are 24 KB) is cut into halves, quarters, .. of the wave's reads, written one after
// the other (the lanes of the others wait: still LDS stores and one coalesced copy instead of records written to global memory byte by byte -- 300 bp reads:
// 14.6 -> 11.1 ms per million reads beside the other lane's kernels); only a single read whose records exceed the share (long XA lists) writes to global memory directly
#define SAM_LDS_WAVE 16384
__global__ void __launch_bounds__(256) sam_text_write_kernel(sam_args_t A)
{
	__shared__ __attribute__((aligned(16))) char lds[4][SAM_LDS_WAVE];
	const uint32_t n = A.d.n_reads;
	const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t r0 = blockIdx.x * 256u + wv * 64u, r = r0 + lane;
	if (r0 >= n) return;
	const uint32_t r1 = r0 + 64u < n ? r0 + 64u : n;
	// reads per part: the largest power of two whose parts all fit
	uint32_t per = 64u;
	for (;;) {
		bool fits = true;
		for (uint32_t b = r0; b < r1 && fits; b += per) {
			const uint32_t e = b + per < r1 ? b + per : r1;
			fits = A.text_off[e] - A.text_off[b] + 3u <= SAM_LDS_WAVE;
		}
		if (fits || per == 1u) break;
		per >>= 1;
	}
	for (uint32_t b = r0; b < r1; b += per) {                                         // (wave-uniform)
		const uint32_t e = b + per < r1 ? b + per : r1;
		const uint64_t t0 = A.text_off[b], t1 = A.text_off[e];
		const uint32_t a = (uint32_t)((uintptr_t)(A.text + t0) & 3u);              // the image starts where its first byte sits in a dword of the text
		const bool mine = r >= b && r < e;
		if (t1 - t0 + a <= SAM_LDS_WAVE) {
			sam_lds_char *img = (sam_lds_char *)lds[wv] + a;
			if (mine) {
				sam_out_t<2> o; o.n = 0;
				o.p = img + (uint32_t)(A.text_off[r] - t0);
				const bool ok = sam_read<2>(A, r, o);
				if (ok && (uint32_t)(o.p - img) != (uint32_t)(A.text_off[r + 1] - t0)) atomicOr(A.err, 2u);   // (the two passes disagree: internal error)
				if (!ok) { atomicOr(A.err, 1u); }
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_s_waitcnt(0xC07F);                                       // (the wave's LDS stores are done: lgkmcnt 0)
			const uint32_t len = (uint32_t)(t1 - t0);
			char *dst = A.text + t0;
			const uint32_t head = len < ((4u - a) & 3u) ? len : ((4u - a) & 3u);
			if (lane < head) dst[lane] = img[lane];
			const uint32_t body = (len - head) >> 2;
			const uint32_t *src32 = (const uint32_t *)(lds[wv] + a + head);           // (a + head is a multiple of 4, or the piece ended)
			uint32_t *dst32 = (uint32_t *)(dst + head);
			for (uint32_t i = lane; i < body; i += 64u) dst32[i] = src32[i];
			const uint32_t tail0 = head + 4u * body;
			if (tail0 + lane < len) dst[tail0 + lane] = img[tail0 + lane];
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_s_waitcnt(0xC07F);                                       // (the copy has read the image before the next part overwrites it)
		} else if (mine) {
			sam_out_t<1> o; o.p = A.text + A.text_off[r]; o.n = 0;
			const bool ok = sam_read<1>(A, r, o);
			if (!ok) atomicOr(A.err, 1u);
			else if ((uint64_t)(o.p - A.text) != A.text_off[r + 1]) atomicOr(A.err, 2u);
		}
	}
}

size_t scan64_bytes(size_t n)
{
	size_t t = 0;
	(void)rocprim::exclusive_scan(nullptr, t, (uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0, n, rocprim::plus<uint64_t>(), 0);
	return (t + 255) & ~(size_t)255;
}

int sam_args(const bmh_post_opt_t *popt, const bmh_sam_dev_t *d, const char *fn, sam_args_t &A)
{
	if (!popt || !d) { bmh_set_error("%s: null argument", fn); return BMH_EINVAL; }
	if (d->n_reads && (!d->d_names || !d->d_name_off || !d->d_reads || !d->d_offs || !d->d_lens || !d->d_contig_names || !d->d_contig_name_off || !d->d_fin_per_read ||
	                   !d->d_slot || !d->d_aln || !d->d_cig_off || !d->d_packed || (d->n_contigs > 1 && !d->d_contig_offset) || (d->d_h_rec && (!d->d_unflag || (d->n_reads & 1))))) {
		bmh_set_error("%s: null argument", fn); return BMH_EINVAL;
	}
	memset(&A, 0, sizeof(A));
	A.d = *d;
	A.flag_all = popt->flag_all; A.max_XA_hits = popt->max_XA_hits; A.max_XA_hits_alt = popt->max_XA_hits_alt; A.no_multi = popt->no_multi; A.softclip = popt->softclip;
	A.drop = (double)popt->XA_drop_ratio;
	A.sa = popt->contig_is_alt ? 11 : 12;
	if (popt->rg_id && popt->rg_id[0]) {
		const size_t l = strlen(popt->rg_id);
		if (l > 255) { bmh_set_error("%s: the read group id is longer than 255 characters", fn); return BMH_EINVAL; }
		A.rg_len = (int)l; memcpy(A.rg, popt->rg_id, l);
	}
	return BMH_OK;
}

// ALT contigs in the native pipeline (csrc/align_pipeline.hip): the device tail runs without the table; its records become ALT-mode records in place ([11] = secondary_all,
// which without a hit on an ALT contig is the secondary), and the reads the host has redone with the table take theirs
__global__ void __launch_bounds__(256) alt_records_kernel(int32_t *fin, uint64_t m)
{
	const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (k < m) fin[16 * k + 11] = fin[16 * k + 12];
}
__global__ void __launch_bounds__(256) alt_scatter_kernel(int32_t *fin, const uint32_t *rec_off, const uint32_t *ids, const uint32_t *sub_off, const int32_t *sub, uint32_t ns)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x, j = t >> 4, l = t & 15u;
	if (j >= ns) return;
	const uint32_t n = sub_off[j + 1] - sub_off[j];
	const int4 *src = (const int4 *)(sub + 16 * (size_t)sub_off[j]);
	int4 *dst = (int4 *)(fin + 16 * (size_t)rec_off[ids[j]]);
	for (uint32_t q = l; q < 4 * n; q += 16) dst[q] = src[q];
}

}   // namespace

int bmh_alt_records_device(int32_t *d_fin, uint64_t m, const uint32_t *d_rec_off, const uint32_t *d_ids, const uint32_t *d_sub_off, const int32_t *d_sub, uint32_t ns, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	if (m) alt_records_kernel<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(d_fin, m);
	if (ns) alt_scatter_kernel<<<(unsigned)(((size_t)ns * 16 + 255) / 256), 256, 0, st>>>(d_fin, d_rec_off, d_ids, d_sub_off, d_sub, ns);
	HIPCK(hipGetLastError());
	return BMH_OK;
}

extern "C" size_t bmh_sam_text_work(uint32_t n_reads)
{
	return 2 * al256(4 * ((size_t)n_reads + 2)) + scan_bytes((size_t)n_reads + 2) + scan64_bytes((size_t)n_reads + 2) + 512;
}

extern "C" int64_t bmh_sam_text_sizes(const bmh_post_opt_t *popt, const bmh_sam_dev_t *d, uint64_t *d_text_off, void *d_work, size_t work_bytes, void *stream_)
{
	sam_args_t A;
	const int rc = sam_args(popt, d, "bmh_sam_text_sizes", A);
	if (rc != BMH_OK) return rc;
	if (!d_text_off || !d_work || work_bytes < bmh_sam_text_work(d->n_reads)) { bmh_set_error("bmh_sam_text_sizes: the work space is smaller than bmh_sam_text_work asks for"); return BMH_EINVAL; }
	const uint32_t n = d->n_reads;
	hipStream_t st = (hipStream_t)stream_;
	if (n == 0) return 0;
	uint8_t *w = (uint8_t *)d_work;
	uint32_t *rec_off = (uint32_t *)w; w += al256(4 * ((size_t)n + 2));
	uint32_t *len = (uint32_t *)w; w += al256(4 * ((size_t)n + 2));
	uint32_t *err = (uint32_t *)w; w += 256;
	void *tmp = w;
	size_t tb = scan_bytes((size_t)n + 2);
	HIPCK(rocprim::exclusive_scan(tmp, tb, d->d_fin_per_read, rec_off, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
	HIPCK(hipMemsetAsync(err, 0, 4, st));
	HIPCK(hipMemsetAsync(len + n, 0, 4, st));
	A.rec_off = rec_off; A.len = len; A.err = err;
	sam_text_count_kernel<<<(n + 255) / 256, 256, 0, st>>>(A);
	tb = scan64_bytes((size_t)n + 2);
	HIPCK(rocprim::exclusive_scan((void *)((uint8_t *)tmp + scan_bytes((size_t)n + 2)), tb, len, d_text_off, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), st));
	uint64_t total = 0; uint32_t h_err = 0;
	HIPCK(hipMemcpyAsync(&total, d_text_off + n, 8, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	if (h_err) { bmh_set_error("bmh_sam_text_sizes: a record the text needs has no CIGAR (see bmh_sam_select_device), or one that bmh_cigar_batch flagged"); return BMH_EINVAL; }
	return (int64_t)total;
}

extern "C" int bmh_sam_text_write(const bmh_post_opt_t *popt, const bmh_sam_dev_t *d, const uint64_t *d_text_off, char *d_text, void *d_work, size_t work_bytes, void *stream_)
{
	sam_args_t A;
	const int rc = sam_args(popt, d, "bmh_sam_text_write", A);
	if (rc != BMH_OK) return rc;
	if (d->n_reads == 0) return BMH_OK;
	if (!d_text_off || !d_text || !d_work || work_bytes < bmh_sam_text_work(d->n_reads)) { bmh_set_error("bmh_sam_text_write: null argument, or a work space smaller than bmh_sam_text_work asks for"); return BMH_EINVAL; }
	const uint32_t n = d->n_reads;
	uint8_t *w = (uint8_t *)d_work;
	A.rec_off = (const uint32_t *)w; w += 2 * al256(4 * ((size_t)n + 2));          // (left there by bmh_sam_text_sizes)
	A.err = (uint32_t *)w;
	A.text_off = d_text_off; A.text = d_text;
	sam_text_write_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(A);
	HIPCK(hipGetLastError());
	return BMH_OK;
}

// 1 = the write pass met an inconsistency (reads d_work's flag; waits for the stream)
extern "C" int bmh_sam_text_check(const void *d_work, uint32_t n_reads, void *stream_)
{
	const uint8_t *w = (const uint8_t *)d_work + 2 * al256(4 * ((size_t)n_reads + 2));
	uint32_t h = 0;
	HIPCK(hipMemcpyAsync(&h, w, 4, hipMemcpyDeviceToHost, (hipStream_t)stream_));
	HIPCK(hipStreamSynchronize((hipStream_t)stream_));
	if (h) { bmh_set_error("bmh_sam_text_write: internal error (flags %u): the text of a read differs from its counted size", h); return BMH_EINVAL; }
	return BMH_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The alignments whose fixed slots overflowed (flags 1, 8), without a trip to the host (csrc/align_pipeline.hip): their list, and -- once
// the caller has redone them with large slots -- their records and words put in place behind the packed array.
namespace {

__global__ void __launch_bounds__(256) cigar_over_kernel(const int32_t *aln, uint32_t n, const uint32_t *sel, uint32_t *over, uint32_t *sel2, uint32_t *counter)
{
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	if (k >= n) return;
	if (aln[8 * (size_t)k + 7] & 9) { const uint32_t t = atomicAdd(counter, 1u); over[t] = k; sel2[t] = sel ? sel[k] : k; }
}

// the redone alignments take their places: words of each (cigar_patch_words), a scan, then one wave per alignment copies
__global__ void __launch_bounds__(256) cigar_patch_words_kernel(const int32_t *aln2, uint32_t n_over, uint32_t *words)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t > n_over) return;
	uint32_t w = 0;
	if (t < n_over) { const int32_t *a = aln2 + 8 * (size_t)t; if (!(a[7] & ~2)) w = (uint32_t)a[3] + (((uint32_t)a[6] + 4u) >> 2); }
	words[t] = w;
}
__global__ void __launch_bounds__(64) cigar_patch_kernel(int32_t *aln, uint32_t *off, uint32_t *packed, const uint32_t *over, uint32_t n_over,
                                                         const int32_t *aln2, const uint32_t *cigar2, int mc2, const char *md2, int mdc2, const uint32_t *start)
{
	const uint32_t t = blockIdx.x, lane = threadIdx.x;
	if (t >= n_over) return;
	const int32_t *a = aln2 + 8 * (size_t)t;
	const uint32_t k = over[t];
	if (lane < 8) aln[8 * (size_t)k + lane] = a[lane];
	if (a[7] & ~2) return;
	if (lane == 0) off[k] = start[t];
	const uint32_t nc = (uint32_t)a[3], nw = ((uint32_t)a[6] + 4u) >> 2;
	uint32_t *o = packed + start[t];
	const uint32_t *cg = cigar2 + (size_t)mc2 * t; const uint32_t *ms = (const uint32_t *)(md2 + (size_t)mdc2 * t);
	for (uint32_t i = lane; i < nc; i += 64) o[i] = cg[i];
	for (uint32_t i = lane; i < nw; i += 64) o[nc + i] = ms[i];
}

}   // namespace

// d_over / d_sel2 [n]: the overflowed alignments and the records they belong to (d_sel: the list bmh_cigar_batch ran on, NULL = identity); d_counter: one
// word of device memory.  Returns their number (waits for the stream).
int64_t bmh_cigar_overflowed(const int32_t *d_aln, uint32_t n, const uint32_t *d_sel, uint32_t *d_over, uint32_t *d_sel2, uint32_t *d_counter, void *stream_)
{
	hipStream_t st = (hipStream_t)stream_;
	if (n == 0) return 0;
	HIPCK(hipMemsetAsync(d_counter, 0, 4, st));
	cigar_over_kernel<<<(n + 255) / 256, 256, 0, st>>>(d_aln, n, d_sel, d_over, d_sel2, d_counter);
	uint32_t h = 0;
	HIPCK(hipMemcpyAsync(&h, d_counter, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	return (int64_t)h;
}

// the redone alignments (d_aln2 / d_cigar2 [n_over][mc2] / d_md2 [n_over][mdc2], in d_over's order) take their places: d_aln[k] is replaced, the words
// are appended behind the `words` packed ones (the caller left room for n_over * (mc2 + mdc2 / 4) more) and d_off[k] points at them; d_work:
// bmh_cigar_patch_work(n_over) bytes.  Returns the new number of words (waits for the stream).
size_t bmh_cigar_patch_work(uint32_t n_over) { return 2 * al256(4 * ((size_t)n_over + 2)) + scan_bytes((size_t)n_over + 2) + 256; }
int64_t bmh_cigar_patch(int32_t *d_aln, uint32_t *d_off, uint32_t *d_packed, uint64_t words, const uint32_t *d_over, uint32_t n_over,
                        const int32_t *d_aln2, const uint32_t *d_cigar2, int mc2, const char *d_md2, int mdc2, void *d_work, size_t work_bytes, void *stream_)
{
	hipStream_t st = (hipStream_t)stream_;
	if (n_over == 0) return (int64_t)words;
	if (!d_work || work_bytes < bmh_cigar_patch_work(n_over)) { bmh_set_error("bmh_cigar_patch: work space too small"); return BMH_EINVAL; }
	uint8_t *w = (uint8_t *)d_work;
	uint32_t *wd = (uint32_t *)w; w += al256(4 * ((size_t)n_over + 2));
	uint32_t *start = (uint32_t *)w; w += al256(4 * ((size_t)n_over + 2));
	cigar_patch_words_kernel<<<(n_over + 1 + 255) / 256, 256, 0, st>>>(d_aln2, n_over, wd);
	size_t tb = scan_bytes((size_t)n_over + 2);
	HIPCK(rocprim::exclusive_scan((void *)w, tb, wd, start, (uint32_t)words, (size_t)n_over + 1, rocprim::plus<uint32_t>(), st));      // start[n_over] = the new total
	cigar_patch_kernel<<<n_over, 64, 0, st>>>(d_aln, d_off, d_packed, d_over, n_over, d_aln2, d_cigar2, mc2, d_md2, mdc2, start);
	uint32_t h = 0;
	HIPCK(hipMemcpyAsync(&h, start + n_over, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	return (int64_t)h;
}
