// FMD-index construction on the MI355X: suffix array of fwd . revcomp(fwd), BWT with Occ blocks and the
// suffix-array samples, all in HBM, for texts beyond 2^32 symbols (hg38: 6.2e9 rows).
//
// What it replaces: the reference builds its index offline on the host -- `bwa index` with the bwtsw
// algorithm for large genomes (/root/reference/bwa_index/bwtindex.c:287-358), the re-blocking into the
// GPU layout (bwt_bwtupdate_core_occ_32, bwa_index/bwtindex.c:174-197) and the sampled suffix array
// (bwt_cal_sa, bwa_index/bwt.c:63-148) -- hours of single-thread CPU for hg38.  The result here is the same
// index, value for value (tests/test_index_builder.py compares with files written by the reference's CLI);
// the method is ours and has nothing in common with bwtsw's incremental BWT merging:
//
//   round 0   suffixes are bucketed by their first 6 symbols; groups of buckets of at most CAP suffixes are
//             collected with their first 32 symbols as a 64-bit key and radix-sorted (rocPRIM) -> SA ordered
//             by 32 symbols, ISA[s] = position of the head of s's group, U = positions of unresolved
//             suffixes (group size > 1).  Positions past the end of the text read as 'A'; the rounds below
//             put suffixes shorter than the depth in their place (key = length, below every rank).
//   doubling  (Larsson-Sadakane on the device) with depth h = 32, 64, ...: the unresolved groups, CAP
//             positions at a time and never cutting a group, are sorted by (group, ISA[s + h]); new group
//             heads become the ranks, singletons leave U.  Ends when U is empty: O(log maxLCP) rounds,
//             each only over what is still unresolved.
//   output    BWT symbol of every row gathered from the 2-bit text, packed MSB-first into the reference's
//             32-byte blocks {u32 occ[4]; u32 bwt[4]} with a rocPRIM scan for the Occ columns; SA samples
//             of every sa_intv-th row as 32 bits + 1 packed high bit (seed_gen.cu:1386-1436 loader format).
//   verify    (optional) every adjacent pair of rows is compared symbol by symbol and ISA[SA[r]] == r is
//             checked for all r: a complete proof that SA is the suffix array.
//
// HBM: SA and ISA as u64 (16 n bytes), U (8 n), sort buffers 41 * CAP bytes: 175 GB for hg38 with CAP = 2^29
// -- sized for the 288 GB of this GPU, which is the point: no external-memory or incremental scheme needed.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "bmh_internal.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("index build: %s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

typedef unsigned long long u64;

// ---------------------------------------------------------------- text

// T = fwd . revcomp(fwd) as 2-bit symbols, 32 per u64 word, first symbol in the top bits; symbols >= n are 0
__global__ void __launch_bounds__(256) ib_text_kernel(const uint8_t *__restrict__ pac, u64 l_pac, u64 n_words, u64 *__restrict__ tw)
{
	const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words) return;
	const u64 n = 2 * l_pac;
	u64 v = 0;
	for (int j = 0; j < 32; ++j) {
		const u64 i = w * 32 + (u64)j;
		int c = 0;
		if (i < n) {
			const bool rev = i >= l_pac;
			const u64 p = rev ? n - 1 - i : i;
			c = (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
			if (rev) c = 3 - c;
		}
		v = (v << 2) | (u64)c;
	}
	tw[w] = v;
}

// the 32 symbols from position i on (tw is padded by two words)
__device__ __forceinline__ u64 ib_key32(const u64 *__restrict__ tw, u64 i)
{
	const u64 w0 = tw[i >> 5], w1 = tw[(i >> 5) + 1];
	const int sh = (int)(i & 31) * 2;
	return sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;
}
__device__ __forceinline__ int ib_sym(const u64 *__restrict__ tw, u64 i) { return (int)(tw[i >> 5] >> (62 - 2 * (int)(i & 31))) & 3; }

// ---------------------------------------------------------------- round 0

#define IB_BUCKET_BITS 12
#define IB_N_BUCKETS (1 << IB_BUCKET_BITS)

__global__ void __launch_bounds__(256) ib_hist_kernel(const u64 *__restrict__ tw, u64 n, u64 *__restrict__ hist)
{
	__shared__ unsigned int h[IB_N_BUCKETS];
	for (int i = threadIdx.x; i < IB_N_BUCKETS; i += blockDim.x) h[i] = 0;
	__syncthreads();
	// a block walks a contiguous span so that its counts stay below 2^32
	const u64 span = (n + gridDim.x - 1) / gridDim.x;
	const u64 lo = (u64)blockIdx.x * span, hi = lo + span < n ? lo + span : n;
	for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&h[ib_key32(tw, i) >> (64 - IB_BUCKET_BITS)], 1u);
	__syncthreads();
	for (int i = threadIdx.x; i < IB_N_BUCKETS; i += blockDim.x) if (h[i]) atomicAdd(&hist[i], (u64)h[i]);
}

// suffixes whose bucket lies in [blo, bhi): (key, position) appended in any order (the sort follows).  A wave takes
// IB_ITEMS * 64 consecutive positions at a time and reserves its output with ONE atomic (a single counter sustains
// only ~90 atomics per microsecond on this chip: one per 64 positions cost 0.8 s per pass over 4.4e9 positions).
#define IB_ITEMS 16
__global__ void __launch_bounds__(256) ib_collect_kernel(const u64 *__restrict__ tw, u64 n, unsigned blo, unsigned bhi,
                                                         u64 *__restrict__ keys, u64 *__restrict__ vals, u64 *__restrict__ cursor)
{
	const int lane = __lane_id();
	const u64 n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
	const u64 tile = (u64)IB_ITEMS * 64;
	for (u64 base = (((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * tile; base < n; base += n_waves * tile) {    // wave-uniform
		u64 k[IB_ITEMS];
		unsigned inm = 0;
#pragma unroll
		for (int it = 0; it < IB_ITEMS; ++it) {
			const u64 i = base + (u64)it * 64 + (u64)lane;
			k[it] = 0;
			if (i < n) {
				k[it] = ib_key32(tw, i);
				const unsigned b = (unsigned)(k[it] >> (64 - IB_BUCKET_BITS));
				if (b >= blo && b < bhi) inm |= 1u << it;
			}
		}
		// exclusive prefix of the per-lane counts over the wave
		const unsigned cnt = (unsigned)__popc(inm);
		unsigned inc = cnt;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(inc, d); if (lane >= d) inc += t; }
		const unsigned total = __shfl(inc, 63);
		if (!total) continue;
		u64 at = 0;
		if (lane == 0) at = atomicAdd(cursor, (u64)total);
		at = __shfl(at, 0) + (u64)(inc - cnt);
#pragma unroll
		for (int it = 0; it < IB_ITEMS; ++it)
			if (inm & (1u << it)) { keys[at] = k[it]; vals[at] = base + (u64)it * 64 + (u64)lane; ++at; }
	}
}

// head marks of a sorted key array: hv[t] = t at the first element of a run of equal keys, else 0 (max-scan -> head index)
__global__ void __launch_bounds__(256) ib_headmark_kernel(const u64 *__restrict__ keys, u64 m, uint32_t *__restrict__ hv)
{
	const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	hv[t] = (t == 0 || keys[t] != keys[t - 1]) ? (uint32_t)t : 0u;
}

// round 0: sorted chunk -> SA, ISA (head position of the group), unresolved flags
__global__ void __launch_bounds__(256) ib_finish0_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ vals, const uint32_t *__restrict__ hd,
                                                         u64 m, u64 base, u64 *__restrict__ SA, u64 *__restrict__ ISA, uint8_t *__restrict__ uf)
{
	const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	const u64 k = keys[t], s = vals[t];
	const bool head = t == 0 || keys[t - 1] != k, tail = t + 1 == m || keys[t + 1] != k;
	SA[base + t] = s;
	ISA[s] = base + (u64)hd[t];
	uf[t] = !(head && tail);
}

// ---------------------------------------------------------------- doubling

// keys of one chunk of unresolved positions: low 34 bits = rank of the suffix h symbols further on (suffixes that end
// before that: their length, which is below every rank and orders them among themselves: the shorter is the smaller);
// gh[t] = 1 at group heads (scan -> group number inside the chunk, the high 30 bits of the key)
#define IB_K2_BITS 34
__global__ void __launch_bounds__(256) ib_dkey_kernel(const u64 *__restrict__ U, u64 m, const u64 *__restrict__ SA, const u64 *__restrict__ ISA,
                                                      u64 n, u64 h, u64 *__restrict__ keys, u64 *__restrict__ vals, uint32_t *__restrict__ gh)
{
	const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	const u64 p = U[t], s = SA[p];
	gh[t] = ISA[s] == p ? 1u : 0u;
	const u64 s2 = s + h;
	keys[t] = s2 < n ? ISA[s2] + n + 1 : n - s;
	vals[t] = s;
}
__global__ void __launch_bounds__(256) ib_dseg_kernel(u64 *__restrict__ keys, const uint32_t *__restrict__ seg, u64 m)
{
	const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	keys[t] |= (u64)(seg[t] - 1u) << IB_K2_BITS;
}
// sorted chunk -> SA, ISA of the refined groups, unresolved flags
__global__ void __launch_bounds__(256) ib_dfinish_kernel(const u64 *__restrict__ U, const u64 *__restrict__ keys, const u64 *__restrict__ vals,
                                                         const uint32_t *__restrict__ hd, u64 m, u64 *__restrict__ SA, u64 *__restrict__ ISA,
                                                         uint8_t *__restrict__ uf)
{
	const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= m) return;
	const u64 k = keys[t], s = vals[t];
	const bool head = t == 0 || keys[t - 1] != k, tail = t + 1 == m || keys[t + 1] != k;
	SA[U[t]] = s;
	ISA[s] = U[hd[t]];
	uf[t] = !(head && tail);
}
// where the group that holds U[e] starts in U (positions of a group are consecutive in U)
__global__ void ib_group_start_kernel(const u64 *__restrict__ U, u64 e, const u64 *__restrict__ SA, const u64 *__restrict__ ISA, u64 *__restrict__ out)
{
	const u64 p = U[e];
	out[0] = e - (p - ISA[SA[p]]);
}

// ---------------------------------------------------------------- output

struct cnt4_t { uint32_t c[4]; };
struct cnt4_plus { __host__ __device__ cnt4_t operator()(const cnt4_t &a, const cnt4_t &b) const { cnt4_t r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] + b.c[i]; return r; } };

__device__ __forceinline__ uint32_t ib_spread16(uint32_t x)      // bit i -> bit 2i
{
	x &= 0xFFFFu;
	x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
	return x;
}

// one wave per 64-symbol block (grid-stride): the BWT symbol of every row (the row of the whole text, `primary`, is left
// out as in the reference's files, is.c / bwt_gen), packed into the block's four words; per-block symbol counts for the scan
__global__ void __launch_bounds__(256) ib_bwt_kernel(const u64 *__restrict__ tw, const u64 *__restrict__ SA, u64 n, u64 primary, u64 n_blk,
                                                     uint32_t *__restrict__ blocks, cnt4_t *__restrict__ cnt)
{
	const int lane = __lane_id();
	const u64 n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
	for (u64 b = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < n_blk; b += n_waves) {
		const u64 j = b * 64 + (u64)lane;
		int c = 0;
		const bool valid = j < n;
		if (valid) {
			const u64 r = j < primary ? j : j + 1;                    // row of the (n+1)-row matrix; row 0 is the empty suffix
			const u64 s = r == 0 ? n : SA[r - 1];
			c = ib_sym(tw, s - 1);                                     // s == 0 only at row primary, which is skipped
		}
		const u64 b0 = __ballot(valid && (c & 1)), b1 = __ballot(valid && (c & 2)), vm = __ballot(valid);
		if (lane < 4) {
			const uint32_t lo = (uint32_t)(b0 >> (16 * lane)), hi = (uint32_t)(b1 >> (16 * lane));
			blocks[b * 8 + 4 + lane] = __brev(ib_spread16(hi) | (ib_spread16(lo) << 1));     // symbol t of the word at bits 31-2t, 30-2t
		}
		if (lane == 0) {
			cnt4_t q;
			q.c[0] = (uint32_t)__popcll(vm & ~b0 & ~b1); q.c[1] = (uint32_t)__popcll(b0 & ~b1);
			q.c[2] = (uint32_t)__popcll(b1 & ~b0); q.c[3] = (uint32_t)__popcll(b0 & b1);
			cnt[b] = q;
		}
	}
}
__global__ void __launch_bounds__(256) ib_occ_kernel(const cnt4_t *__restrict__ occ, u64 n_blk1, uint32_t *__restrict__ blocks)
{
	const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blk1) return;
	const cnt4_t q = occ[b];
	*(uint4 *)&blocks[b * 8] = make_uint4(q.c[0], q.c[1], q.c[2], q.c[3]);
}
// samples of rows 0, intv, 2 intv, ... of the (n+1)-row matrix: low 32 bits + one packed high bit; sa[0] = -1 (seed_gen.cu:1423)
__global__ void __launch_bounds__(256) ib_sample_kernel(const u64 *__restrict__ SA, u64 n_sa, int shift, u64 hi0, uint32_t *__restrict__ sa, uint32_t *__restrict__ bits)
{
	const int lane = __lane_id();
	const u64 stride = (u64)gridDim.x * blockDim.x;                              // a multiple of 64: the trip count is wave-uniform
	for (u64 j0 = (u64)blockIdx.x * blockDim.x + threadIdx.x - lane; j0 < n_sa; j0 += stride) {
		const u64 j = j0 + (u64)lane;
		u64 v = 0;
		if (j < n_sa) { v = j == 0 ? (0xFFFFFFFFull | (hi0 << 32)) : SA[(j << shift) - 1]; sa[j] = (uint32_t)v; }
		const u64 m = __ballot(j < n_sa && ((v >> 32) & 1));
		if ((lane & 31) == 0 && j < n_sa) bits[j >> 5] = (uint32_t)(m >> lane);  // j is a multiple of 32 at lanes 0 and 32
	}
}

// ---------------------------------------------------------------- verification

// err[0]: rows out of order, err[1]: ISA[SA[r]] != r or SA[r] out of range
__device__ void ib_verify_row(const u64 *__restrict__ tw, const u64 *__restrict__ SA, const u64 *__restrict__ ISA, u64 n, u64 r, u64 *__restrict__ err)
{
	const u64 a = SA[r];
	if (a >= n || ISA[a] != r) { atomicAdd(&err[1], 1ull); return; }
	if (r + 1 >= n) return;
	const u64 b = SA[r + 1];
	if (b >= n) return;
	// suffix a must be smaller than suffix b; a suffix that ends first is the smaller
	u64 d = 0;
	for (;;) {
		const u64 la = n - (a + d), lb = n - (b + d);                 // symbols left (neither is 0 here: a + d, b + d < n)
		const u64 ka = ib_key32(tw, a + d), kb = ib_key32(tw, b + d);
		const u64 lim = la < lb ? la : lb;
		if (lim >= 32) {
			if (ka != kb) { if (ka > kb) atomicAdd(&err[0], 1ull); return; }
			d += 32;
			if (a + d >= n || b + d >= n) { if (!(a + d >= n)) atomicAdd(&err[0], 1ull); return; }   // the one that ran out must be a
			continue;
		}
		const int sh = 64 - 2 * (int)lim;
		const u64 pa = ka >> sh, pb = kb >> sh;
		if (pa != pb) { if (pa > pb) atomicAdd(&err[0], 1ull); return; }
		if (!(la < lb)) atomicAdd(&err[0], 1ull);                     // equal up to the end of the shorter: a must be the shorter
		return;
	}
}
__global__ void __launch_bounds__(256) ib_verify_kernel(const u64 *__restrict__ tw, const u64 *__restrict__ SA, const u64 *__restrict__ ISA, u64 n, u64 *__restrict__ err)
{
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) ib_verify_row(tw, SA, ISA, n, r, err);
}

// ---------------------------------------------------------------- host

static inline unsigned ib_nblk(u64 n, unsigned b) { return (unsigned)((n + b - 1) / b); }

struct ib_bufs_t {
	u64 *tw = nullptr, *SA = nullptr, *ISA = nullptr, *U = nullptr, *kA = nullptr, *kB = nullptr, *vA = nullptr, *vB = nullptr, *small = nullptr;
	uint32_t *h1 = nullptr, *h2 = nullptr; uint8_t *uf = nullptr; void *tmp = nullptr; cnt4_t *cnt = nullptr, *occ = nullptr;
	~ib_bufs_t() { void *ps[] = {tw, SA, ISA, U, kA, kB, vA, vB, small, h1, h2, uf, tmp, cnt, occ}; for (void *p : ps) if (p) (void)hipFree(p); }
};

extern "C" int bmh_index_build(const uint8_t *d_pac, uint64_t l_pac, int sa_intv, uint32_t *d_bwt_words, uint32_t *d_sa, uint32_t *d_sa_bits,
                               uint64_t *primary_out, uint64_t L2_out[5], int flags, bmh_build_stats_t *stats)
{
	if (!d_pac || !d_bwt_words || !d_sa || !d_sa_bits || !primary_out || !L2_out) { bmh_set_error("bmh_index_build: null argument"); return BMH_EINVAL; }
	if (l_pac == 0 || (l_pac >> 32)) { bmh_set_error("bmh_index_build: l_pac %llu outside (0, 2^32): the text must stay below 2^33 symbols", (u64)l_pac); return BMH_EINVAL; }
	if (sa_intv < 1 || (sa_intv & (sa_intv - 1))) { bmh_set_error("bmh_index_build: sa_intv %d is not a power of two", sa_intv); return BMH_EINVAL; }
	if (((uintptr_t)d_bwt_words & 31) != 0) { bmh_set_error("bmh_index_build: bwt words must be 32-byte aligned"); return BMH_EINVAL; }
	const auto T0 = std::chrono::steady_clock::now();
	auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count(); };
	const u64 n = 2 * (u64)l_pac;
	int cap_log2 = 29;
	if (const char *e = getenv("BMH_BUILD_CAP_LOG2")) cap_log2 = atoi(e);
	if (cap_log2 < 8 || cap_log2 > 30) { bmh_set_error("bmh_index_build: BMH_BUILD_CAP_LOG2 outside [8, 30]"); return BMH_EINVAL; }
	const u64 CAP = std::min<u64>(1ull << cap_log2, n);
	const bool verbose = getenv("BMH_BUILD_VERBOSE") != nullptr;
	hipStream_t st = nullptr;
	ib_bufs_t B;
	const u64 n_tw = n / 32 + 3;
	HIPCK(hipMalloc((void **)&B.tw, n_tw * 8)); HIPCK(hipMalloc((void **)&B.SA, n * 8)); HIPCK(hipMalloc((void **)&B.ISA, n * 8));
	HIPCK(hipMalloc((void **)&B.U, n * 8));
	HIPCK(hipMalloc((void **)&B.kA, CAP * 8)); HIPCK(hipMalloc((void **)&B.kB, CAP * 8)); HIPCK(hipMalloc((void **)&B.vA, CAP * 8)); HIPCK(hipMalloc((void **)&B.vB, CAP * 8));
	HIPCK(hipMalloc((void **)&B.h1, CAP * 4)); HIPCK(hipMalloc((void **)&B.h2, CAP * 4)); HIPCK(hipMalloc((void **)&B.uf, CAP));
	HIPCK(hipMalloc((void **)&B.small, (IB_N_BUCKETS + 16) * 8));
	size_t t_sort = 0, t_scan = 0, t_sel = 0;
	HIPCK(rocprim::radix_sort_pairs(nullptr, t_sort, B.kA, B.kB, B.vA, B.vB, (size_t)CAP, 0, 64, st));
	HIPCK(rocprim::inclusive_scan(nullptr, t_scan, B.h1, B.h2, (size_t)CAP, rocprim::maximum<uint32_t>(), st));
	HIPCK(rocprim::select(nullptr, t_sel, rocprim::counting_iterator<u64>(0), B.uf, B.kA, B.small, (size_t)CAP, st));
	size_t tmp_bytes = std::max(t_sort, std::max(t_scan, t_sel)) + 256;
	HIPCK(hipMalloc(&B.tmp, tmp_bytes));

	ib_text_kernel<<<ib_nblk(n_tw, 256), 256, 0, st>>>(d_pac, l_pac, n_tw, B.tw);
	HIPCK(hipGetLastError());
	// ---- round 0
	u64 *d_hist = B.small, *d_cursor = B.small + IB_N_BUCKETS, *d_count = B.small + IB_N_BUCKETS + 1, *d_gs = B.small + IB_N_BUCKETS + 2;
	HIPCK(hipMemsetAsync(B.small, 0, (IB_N_BUCKETS + 16) * 8, st));
	ib_hist_kernel<<<4096, 256, 0, st>>>(B.tw, n, d_hist);
	HIPCK(hipGetLastError());
	std::vector<u64> hist(IB_N_BUCKETS);
	HIPCK(hipMemcpy(hist.data(), d_hist, IB_N_BUCKETS * 8, hipMemcpyDeviceToHost));
	u64 n_unres = 0, base = 0;
	int n_pass = 0;
	for (unsigned blo = 0; blo < IB_N_BUCKETS;) {
		u64 m = 0; unsigned bhi = blo;
		while (bhi < IB_N_BUCKETS && m + hist[bhi] <= CAP) m += hist[bhi++];
		if (bhi == blo) { bmh_set_error("bmh_index_build: %llu suffixes share their first 6 symbols, more than the chunk capacity %llu (raise BMH_BUILD_CAP_LOG2)", hist[blo], CAP); return BMH_ECAPACITY; }
		if (m) {
			HIPCK(hipMemsetAsync(d_cursor, 0, 8, st));
			ib_collect_kernel<<<8192, 256, 0, st>>>(B.tw, n, blo, bhi, B.kA, B.vA, d_cursor);
			HIPCK(hipGetLastError());
			size_t tb = tmp_bytes;
			HIPCK(rocprim::radix_sort_pairs(B.tmp, tb, B.kA, B.kB, B.vA, B.vB, (size_t)m, 0, 64, st));
			ib_headmark_kernel<<<ib_nblk(m, 256), 256, 0, st>>>(B.kB, m, B.h1);
			HIPCK(hipGetLastError());
			tb = tmp_bytes;
			HIPCK(rocprim::inclusive_scan(B.tmp, tb, B.h1, B.h2, (size_t)m, rocprim::maximum<uint32_t>(), st));
			ib_finish0_kernel<<<ib_nblk(m, 256), 256, 0, st>>>(B.kB, B.vB, B.h2, m, base, B.SA, B.ISA, B.uf);
			HIPCK(hipGetLastError());
			tb = tmp_bytes;
			HIPCK(rocprim::select(B.tmp, tb, rocprim::counting_iterator<u64>(base), B.uf, B.U + n_unres, d_count, (size_t)m, st));
			u64 cnt = 0;
			HIPCK(hipMemcpy(&cnt, d_count, 8, hipMemcpyDeviceToHost));
			n_unres += cnt; base += m; ++n_pass;
		}
		blo = bhi;
	}
	if (base != n) { bmh_set_error("bmh_index_build: internal error: %llu of %llu suffixes collected", base, n); return BMH_EINVAL; }
	if (verbose) fprintf(stderr, "[index build] n = %llu, round 0: %d passes, %llu unresolved (%.1f%%), %.2f s\n", n, n_pass, n_unres, 100.0 * n_unres / n, secs());
	if (stats) { stats->round0_passes = n_pass; stats->unresolved_after_round0 = n_unres; stats->round0_seconds = secs(); }
	// ---- doubling
	int rounds = 0;
	for (u64 h = 32; n_unres; h *= 2, ++rounds) {
		if (h > 2 * n) { bmh_set_error("bmh_index_build: internal error: doubling did not converge"); return BMH_EINVAL; }
		u64 wcur = 0;
		for (u64 cs = 0; cs < n_unres;) {
			u64 ce = std::min(cs + CAP, n_unres);
			if (ce < n_unres) {
				ib_group_start_kernel<<<1, 1, 0, st>>>(B.U, ce, B.SA, B.ISA, d_gs);
				HIPCK(hipGetLastError());
				HIPCK(hipMemcpy(&ce, d_gs, 8, hipMemcpyDeviceToHost));
				if (ce <= cs) { bmh_set_error("bmh_index_build: a group of equal %llu-symbol prefixes exceeds the chunk capacity %llu (raise BMH_BUILD_CAP_LOG2)", h, CAP); return BMH_ECAPACITY; }
			}
			const u64 m = ce - cs;
			const u64 *Uc = B.U + cs;
			ib_dkey_kernel<<<ib_nblk(m, 256), 256, 0, st>>>(Uc, m, B.SA, B.ISA, n, h, B.kA, B.vA, B.h1);
			HIPCK(hipGetLastError());
			size_t tb = tmp_bytes;
			HIPCK(rocprim::inclusive_scan(B.tmp, tb, B.h1, B.h2, (size_t)m, rocprim::plus<uint32_t>(), st));
			ib_dseg_kernel<<<ib_nblk(m, 256), 256, 0, st>>>(B.kA, B.h2, m);
			HIPCK(hipGetLastError());
			tb = tmp_bytes;
			HIPCK(rocprim::radix_sort_pairs(B.tmp, tb, B.kA, B.kB, B.vA, B.vB, (size_t)m, 0, 64, st));
			ib_headmark_kernel<<<ib_nblk(m, 256), 256, 0, st>>>(B.kB, m, B.h1);
			HIPCK(hipGetLastError());
			tb = tmp_bytes;
			HIPCK(rocprim::inclusive_scan(B.tmp, tb, B.h1, B.h2, (size_t)m, rocprim::maximum<uint32_t>(), st));
			ib_dfinish_kernel<<<ib_nblk(m, 256), 256, 0, st>>>(Uc, B.kB, B.vB, B.h2, m, B.SA, B.ISA, B.uf);
			HIPCK(hipGetLastError());
			tb = tmp_bytes;
			HIPCK(rocprim::select(B.tmp, tb, Uc, B.uf, B.kA, d_count, (size_t)m, st));        // survivors -> kA, then behind the write cursor of U
			u64 cnt = 0;
			HIPCK(hipMemcpy(&cnt, d_count, 8, hipMemcpyDeviceToHost));
			if (cnt) HIPCK(hipMemcpyAsync(B.U + wcur, B.kA, cnt * 8, hipMemcpyDeviceToDevice, st));
			wcur += cnt;
			cs = ce;
		}
		if (verbose) fprintf(stderr, "[index build] depth %llu -> %llu: %llu unresolved left, %.2f s\n", h, 2 * h, wcur, secs());
		n_unres = wcur;
	}
	if (stats) { stats->doubling_rounds = rounds; stats->sa_seconds = secs(); }
	// the sort buffers are no longer needed: make room for the block counts
	(void)hipFree(B.kA); (void)hipFree(B.kB); (void)hipFree(B.vA); (void)hipFree(B.vB); B.kA = B.kB = B.vA = B.vB = nullptr;
	(void)hipFree(B.U); B.U = nullptr;
	// ---- verification (optional)
	if (flags & BMH_BUILD_VERIFY) {
		u64 *d_err = B.small;
		HIPCK(hipMemsetAsync(d_err, 0, 16, st));
		ib_verify_kernel<<<(unsigned)std::min<u64>(ib_nblk(n, 256), 1u << 20), 256, 0, st>>>(B.tw, B.SA, B.ISA, n, d_err);
		HIPCK(hipGetLastError());
		u64 err[2] = {0, 0};
		HIPCK(hipMemcpy(err, d_err, 16, hipMemcpyDeviceToHost));
		if (stats) stats->verify_seconds = secs() - stats->sa_seconds;
		if (err[0] || err[1]) { bmh_set_error("bmh_index_build: verification failed: %llu rows out of order, %llu rows not a permutation", err[0], err[1]); return BMH_EINVAL; }
		if (stats) stats->verified = 1;
	}
	// ---- output: primary, BWT blocks + Occ, samples
	u64 pos0 = 0;
	HIPCK(hipMemcpy(&pos0, B.ISA, 8, hipMemcpyDeviceToHost));
	const u64 primary = pos0 + 1;
	(void)hipFree(B.ISA); B.ISA = nullptr;
	const u64 n_blk = (n + 63) / 64;
	HIPCK(hipMalloc((void **)&B.cnt, (n_blk + 1) * sizeof(cnt4_t))); HIPCK(hipMalloc((void **)&B.occ, (n_blk + 1) * sizeof(cnt4_t)));
	HIPCK(hipMemsetAsync(d_bwt_words, 0, (n_blk + 1) * 32, st));
	HIPCK(hipMemsetAsync(B.cnt + n_blk, 0, sizeof(cnt4_t), st));
	ib_bwt_kernel<<<16384, 256, 0, st>>>(B.tw, B.SA, n, primary, n_blk, d_bwt_words, B.cnt);
	HIPCK(hipGetLastError());
	{
		size_t tb = 0; cnt4_t zero = {{0, 0, 0, 0}};
		HIPCK(rocprim::exclusive_scan(nullptr, tb, B.cnt, B.occ, zero, (size_t)n_blk + 1, cnt4_plus(), st));
		if (tb > tmp_bytes) { (void)hipFree(B.tmp); B.tmp = nullptr; HIPCK(hipMalloc(&B.tmp, tb)); tmp_bytes = tb; }
		HIPCK(rocprim::exclusive_scan(B.tmp, tb, B.cnt, B.occ, zero, (size_t)n_blk + 1, cnt4_plus(), st));
	}
	ib_occ_kernel<<<ib_nblk(n_blk + 1, 256), 256, 0, st>>>(B.occ, n_blk + 1, d_bwt_words);
	HIPCK(hipGetLastError());
	cnt4_t tot;
	HIPCK(hipMemcpy(&tot, B.occ + n_blk, sizeof(tot), hipMemcpyDeviceToHost));
	// per-symbol counts above 2^32 would have wrapped in the 32-bit Occ columns of this layout: recount in 64 bits from the histogram
	u64 c64[4] = {0, 0, 0, 0};
	for (unsigned b = 0; b < IB_N_BUCKETS; ++b) c64[b >> (IB_BUCKET_BITS - 2)] += hist[b];
	L2_out[0] = 0;
	for (int c = 0; c < 4; ++c) {
		if (c64[c] >> 32) { bmh_set_error("bmh_index_build: %llu occurrences of base %d do not fit the 32-bit Occ columns of the reference's GPU layout", c64[c], c); return BMH_EINVAL; }
		if (c64[c] != tot.c[c]) { bmh_set_error("bmh_index_build: internal error: symbol counts of text and BWT differ"); return BMH_EINVAL; }
		L2_out[c + 1] = L2_out[c] + c64[c];
	}
	int shift = 0;
	while ((1 << shift) < sa_intv) ++shift;
	const u64 n_sa = (n + (u64)sa_intv) / (u64)sa_intv;
	HIPCK(hipMemsetAsync(d_sa_bits, 0, (n_sa / 32 + 1) * 4, st));
	ib_sample_kernel<<<(unsigned)std::min<u64>(ib_nblk(n_sa, 256), 1u << 20), 256, 0, st>>>(B.SA, n_sa, shift, (n >> 32) & 1, d_sa, d_sa_bits);
	HIPCK(hipGetLastError());
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	*primary_out = primary;
	if (stats) stats->total_seconds = secs();
	if (verbose) fprintf(stderr, "[index build] done: primary %llu, %d doubling rounds, %.2f s\n", primary, rounds, secs());
	return BMH_OK;
}
