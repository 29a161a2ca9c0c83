// SMEM seeding over the FMD index on gfx950 -- hand-written HIP, wave64.
//
// What it computes (bit-identical to the reference's CPU statement):
//   per read, every SMEM of the first seeding pass of BWA-MEM
//   (bwt_smem1, /root/reference/src/bwt.c:483-566, driven as
//   bwa_index/bwamem.c:114-131) with length >= min_seed_len
//   (src/bwamem.c:260-263), and the text position of every occurrence
//   (bwt_sa, src/bwt.c:105-115), laid out as the reference's mem_seed_v_gpu
//   (src/GPUSeed/seed_gen.h:68-75).
//
// How (our own decomposition; the reference's kernels are seed_gen.cu:868-1085,
// 520-662, 704-780 -- we keep their candidate/backward-search idea but none of
// their structure):
//   pack      ASCII -> 2-bit words + N mask, transposed [word][read] so that a
//             wave reading word w of 64 consecutive reads is one coalesced load
//   forward   one lane per read: bidirectional forward extension; at every
//             interval-size change with end >= min_seed_len append a candidate
//             {read, ordinal, start, end, k, s} to a global list (one
//             wave-aggregated atomic per append round)
//   backward  one lane per candidate: unidirectional backward search from
//             start-1 to the maximal begin; result written at
//             cand_base[read]+ordinal, i.e. already sorted by (read, end)
//   filter    one lane per result: drop length < min_seed_len and results whose
//             longer same-pass neighbour has the same begin (contained match;
//             src/bwt.c:535-541), emit per-result occurrence counts
//   scan      rocPRIM exclusive scans (candidate bases, occurrence offsets)
//   expand    SA rows of every kept SMEM, qbeg/score columns of the output
//   locate    one lane per occurrence: LF walk to a sampled row + SA sample
// Every rank query is one 32-byte block = two 16-byte loads from one 64-byte
// sector; all state is in registers, there is no LDS use in these kernels
// (the index is far larger than LDS and each block is used once).
#include <cstring>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "bmh_internal.h"
#include "wtrace.h"
#include "fmd_dev.h"

// ---------------------------------------------------------------- read access

struct read_view_t {
	const uint32_t *pk;   // [word][read] 2-bit, base i at bits 2*(i&15)
	const uint32_t *nm;   // [word32][read] N mask, base i at bit i&31
	uint32_t n_reads;
};

__device__ __forceinline__ int read_base(const read_view_t &v, uint32_t r, int i)
{
	uint32_t w = v.pk[(size_t)(i >> 4) * v.n_reads + r];
	uint32_t m = v.nm[(size_t)(i >> 5) * v.n_reads + r];
	return ((m >> (i & 31)) & 1) ? 4 : (int)((w >> ((i & 15) << 1)) & 3);
}

__device__ __forceinline__ int ascii_code(uint8_t ch)
{
	ch &= 0xDF;
	return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
}

// 64 consecutive reads per 256-thread block.  The block's ASCII span (reads are back to back in
// the input, seed_gen.cu:1715-1716) is staged into LDS with coalesced dword loads, then each
// thread packs (read, 32-base group) items from LDS: two 2-bit words and one N-mask word,
// stored transposed [word][read].  Reads that are not contiguous fall back to global byte loads.
#define PACK_READS_PER_BLOCK 64
__global__ void __launch_bounds__(256) pack_reads_kernel(const uint8_t *__restrict__ ascii, const uint32_t *__restrict__ offs,
                                                         const uint32_t *__restrict__ lens, uint32_t n_reads, uint32_t n_grp,
                                                         uint32_t lds_bytes, uint32_t *__restrict__ pk, uint32_t *__restrict__ nm)
{
	wtrace_scope_t wt_(WT_PACK);
	extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
	const uint32_t r0 = blockIdx.x * PACK_READS_PER_BLOCK;
	const uint32_t nr = min((uint32_t)PACK_READS_PER_BLOCK, n_reads - r0);
	const uint32_t first = offs[r0], last = offs[r0 + nr - 1] + lens[r0 + nr - 1];
	const uint32_t abase = first & ~3u;                      // dword-aligned start of the span
	const bool staged = last >= first && (last - abase) + 4 <= lds_bytes && (((uintptr_t)ascii) & 3) == 0;
	if (staged) {
		const uint32_t nw = (last - abase + 3) >> 2;
		const uint32_t *src = (const uint32_t *)(ascii + abase);
		for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) ((uint32_t *)stage)[i] = src[i];
	}
	__syncthreads();
	for (uint32_t it = threadIdx.x; it < nr * n_grp; it += blockDim.x) {
		const uint32_t rr = it % nr, g = it / nr, r = r0 + rr;
		const uint32_t len = lens[r], o = offs[r];
		const bool in_lds = staged && o >= first && o + len <= last;
		const uint8_t *p = in_lds ? stage + (o - abase) : ascii + o;
		uint32_t w0 = 0, w1 = 0, m = 0;
		for (int t = 0; t < 32; ++t) {
			const uint32_t i = g * 32 + t;
			const int c = i < len ? ascii_code(p[i]) : 4;
			if (c > 3) m |= 1u << t;
			else if (t < 16) w0 |= (uint32_t)c << (2 * t);
			else w1 |= (uint32_t)c << (2 * (t - 16));
		}
		pk[(size_t)(2 * g) * n_reads + r] = w0;
		pk[(size_t)(2 * g + 1) * n_reads + r] = w1;
		nm[(size_t)g * n_reads + r] = m;
	}
}

// ---------------------------------------------------------------- forward

struct cand_t { uint32_t read, xe, j, s; };   // xe = start<<16 | end

// Candidate list appends.  A single global counter bumped once per wave per append round
// saturates (one word takes ~90 atomics/us on this chip), so each wave reserves CHUNK
// slots at a time with one atomic and hands them out with a wave-uniform cursor and a
// ballot prefix; the unused tail of a chunk is filled with invalid entries
// (read == CAND_INVALID) that the backward kernel skips.
#define CAND_CHUNK 1024u
#define CAND_INVALID 0xFFFFFFFFu

struct cand_cursor_t { unsigned long long cur, end; };   // wave-uniform

__device__ __forceinline__ void cand_fill_invalid(const cand_cursor_t &cc, cand_t *out_a, uint64_t cap)
{
	const int lane = __lane_id();
	for (unsigned long long p = cc.cur + lane; p < cc.end; p += 64)
		if (p < cap) { cand_t inv = {CAND_INVALID, 0, 0, 0}; out_a[p] = inv; }
}

__device__ __forceinline__ void cand_append(bool want, const cand_t &c, uint64_t k, cand_t *out_a, uint64_t *out_k,
                                            unsigned long long *counter, uint64_t cap, cand_cursor_t &cc)
{
	unsigned long long mask = __ballot(want);
	if (!mask) return;
	const int lane = __lane_id();
	const unsigned n = (unsigned)__popcll(mask);
	if (cc.cur + n > cc.end) {                 // wave-uniform branch: take a fresh chunk
		cand_fill_invalid(cc, out_a, cap);
		unsigned long long base = 0;
		if (lane == 0) base = atomicAdd(counter, (unsigned long long)CAND_CHUNK);
		cc.cur = __shfl(base, 0);
		cc.end = cc.cur + CAND_CHUNK;
	}
	if (want) {
		uint64_t pos = cc.cur + __popcll(mask & ((1ull << lane) - 1));
		if (pos < cap) { out_a[pos] = c; out_k[pos] = k; }
	}
	cc.cur += n;
}

#ifndef FWD_UNIQ_WORDS
#define FWD_UNIQ_WORDS 1
#endif
__global__ void __launch_bounds__(256) smem_forward_kernel(fmd_dev_t f, read_view_t rv, const uint32_t *__restrict__ lens,
                                                           int min_seed_len, cand_t *__restrict__ out_a, uint64_t *__restrict__ out_k,
                                                           unsigned long long *counter, uint64_t cap, uint32_t *__restrict__ n_cand,
                                                           unsigned long long *__restrict__ fst, const uint32_t *__restrict__ deal, uint32_t *__restrict__ iters_out)
{
	wtrace_scope_t wt_(WT_FORWARD);
	fmd_wave_prio(f.wave_prio);
	// (round 5, measured and dropped: reads PULLED from a cursor by a resident grid -- a wave deals 64-read chunks to its lanes as they
	// finish, 8 idle lanes at a time -- instead of 64 consecutive reads per wave: 2.66-2.71 -> 2.84 ms; the kernel's time is its gathers
	// and the candidate appends, not the lanes that wait for their wave's slowest read)
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	bool live = r < rv.n_reads;
	// (experiment, knob SEED_FWD_SORT: the reads dealt in the order of a list -- the previous call's reads by their iteration counts, costliest first --, so that
	// the 64 reads of a wave cost alike; the candidates do not depend on the deal: they are placed by (read, ordinal) afterwards)
	if (deal && live) r = deal[r];
	int len = live ? (int)lens[r] : 0;
	int i = 0, x = 0;
	uint32_t j = 0;
	uint64_t k = 0, l = 0, s = 0;
	// state machine: one rank-pair (or one candidate append) per iteration, so the
	// lanes of a wave stay convergent on the loads
	// ST_UNIQ (needs the 2-bit text, f.pac): once the interval holds ONE suffix, extending the match is comparing the read
	// with the text behind that suffix -- its position is one suffix-array gather (bmh_index_densify_sa), the text comes 16
	// symbols per load -- instead of two rank gathers per base; k does not move while a unique match grows forward, so the
	// candidate pushed at the end is the one the rank walk would push (src/bwt.c:505-519 with x[2] == 1)
	enum { ST_START, ST_EXT, ST_DONE, ST_TAIL, ST_UNIQ };
	uint64_t tp = 0;                 // ST_UNIQ: text index that pairs with read position i
	int st = live && len > 0 ? ST_START : ST_DONE;
	cand_cursor_t cc = {0, 0};
	unsigned it_wave = 0, it_mine = 0;                   // (BMH_SEED_STATS: iterations of the wave / in which this lane's read was still at work)
	while (__any(st != ST_DONE)) {
		++it_wave; it_mine += st != ST_DONE ? 1u : 0u;
		bool want = false;
		cand_t c = {r, 0, 0, 0};
		uint64_t ck = 0;
		if (st == ST_START) {
			// skip ambiguous bases, open a pass at the first A/C/G/T
			int b = read_base(rv, r, i);
			++i;
			if (b < 4) {
				x = i - 1;
				k = fmd_L2(f, b) + 1; s = fmd_L2(f, b + 1) - fmd_L2(f, b); l = fmd_L2(f, 3 - b) + 1;
				if (i == len) {           // pass that starts on the last base
					want = i >= min_seed_len;
					c.xe = ((uint32_t)x << 16) | (uint32_t)i; c.j = j; c.s = (uint32_t)s; ck = k;
					st = ST_DONE;
				} else st = ST_EXT;
			} else if (i == len) st = ST_DONE;
		} else if (st == ST_EXT) {
			int b = read_base(rv, r, i);
			if (b < 4) {
				int cb = 3 - b;
				uint64_t ak[4], al[4], as[4];
				fmd_forward_ext(f, k, l, s, ak, al, as);
				const uint64_t ns = cb == 0 ? as[0] : cb == 1 ? as[1] : cb == 2 ? as[2] : as[3];
				const uint64_t nk = cb == 0 ? ak[0] : cb == 1 ? ak[1] : cb == 2 ? ak[2] : ak[3];
				const uint64_t nl = cb == 0 ? al[0] : cb == 1 ? al[1] : cb == 2 ? al[2] : al[3];
				if (ns != s) {
					want = i >= min_seed_len;
					c.xe = ((uint32_t)x << 16) | (uint32_t)i; c.j = j; c.s = (uint32_t)s; ck = k;
				}
				if (ns == 0) st = ST_START;            // next pass starts at i (src/bwt.c:519 ret)
				else {
					k = nk; l = nl; s = ns; ++i;
					if (i == len) {                       // reached the end: push the last interval
						// the interval just computed is itself a candidate; it is appended on the
						// next iteration through ST_TAIL below
						st = ST_TAIL;
					} else if (s == 1 && f.pac) {
						tp = fmd_sa(f, k) + (uint64_t)(i - x);
						st = ST_UNIQ;
					}
				}
			} else {                                      // ambiguous base ends the pass
				want = i >= min_seed_len;
				c.xe = ((uint32_t)x << 16) | (uint32_t)i; c.j = j; c.s = (uint32_t)s; ck = k;
				st = ST_START;
			}
		} else if (st == ST_TAIL) {
			want = i >= min_seed_len;
			c.xe = ((uint32_t)x << 16) | (uint32_t)i; c.j = j; c.s = (uint32_t)s; ck = k;
			st = ST_DONE;
		} else if (st == ST_UNIQ) {
			// the rest of this read word against the text: stop at the first differing symbol, the first N, the end of the text;
			// up to FWD_UNIQ_WORDS read words per iteration (an iteration costs the wave the rank step of its other lanes too)
			bool stop = false;
			for (int rep = 0; rep < FWD_UNIQ_WORDS && !stop && i < len; ++rep) {
				const int w16 = i & 15;
				const uint32_t rw = rv.pk[(size_t)(i >> 4) * rv.n_reads + r] >> (2 * w16);
				const uint32_t rm = rv.nm[(size_t)(i >> 5) * rv.n_reads + r] >> (i & 31);
				int n_av = min(16 - w16, len - i);
				const uint64_t room = tp < f.seq_len ? f.seq_len - tp : 0;
				n_av = (uint64_t)n_av < room ? n_av : (int)room;
				const uint32_t diff = rw ^ fmd_text16(f, tp);
				const uint32_t dp = (diff | (diff >> 1)) & 0x55555555u;
				const int fd = dp ? (__ffs((int)dp) - 1) >> 1 : 16;
				const int fn = (rm & 0xFFFFu) ? __ffs((int)(rm & 0xFFFFu)) - 1 : 16;
				const int m = min(min(fd, fn), n_av);
				stop = m < n_av || n_av == 0;                     // mismatch, N, or nothing left of the text: the pass ends at i
				i += m; tp += (uint64_t)m;
			}
			if (stop) {
				want = i >= min_seed_len;
				c.xe = ((uint32_t)x << 16) | (uint32_t)i; c.j = j; c.s = 1u; ck = k;
				st = ST_START;
			} else if (i == len) st = ST_TAIL;
		}
		if (want) ++j;
		cand_append(want, c, ck, out_a, out_k, counter, cap, cc);
	}
	cand_fill_invalid(cc, out_a, cap);
	if (live) n_cand[r] = j;
	if (iters_out && live) iters_out[r] = it_mine;
	if (fst) {       // [0] wave-iterations [1] lane-iterations [2] waves [3] longest wave; [8..40) reads by iterations / 8; [40..64) waves by iterations / 8
		if (live) { atomicAdd(fst + 1, (unsigned long long)it_mine); atomicAdd(fst + 8 + (it_mine / 8u < 31u ? it_mine / 8u : 31u), 1ull); }
		if ((threadIdx.x & 63) == 0) { atomicAdd(fst, (unsigned long long)it_wave); atomicAdd(fst + 2, 1ull); atomicMax(fst + 3, (unsigned long long)it_wave); atomicAdd(fst + 40 + (it_wave / 8u < 23u ? it_wave / 8u : 23u), 1ull); }
	}
}

// ---------------------------------------------------------------- backward

struct res_t { uint32_t read, be, s, pad; };   // be = begin<<16 | end; s == 0: dropped

// list order -> (read, ordinal) order: slot cand_base[read] + ordinal receives the list index of its candidate (4 bytes
// scattered instead of the 24-byte candidate; the backward kernel fetches the candidate through it).  In that order the
// candidates of one forward pass (same read, same start) sit in adjacent slots, shortest first.
__global__ void __launch_bounds__(256) cand_scatter_kernel(const cand_t *__restrict__ in_a, uint64_t n_list, const uint32_t *__restrict__ cand_base,
                                                           uint32_t *__restrict__ perm)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_list) return;
	const uint2 h = *(const uint2 *)&in_a[t];               // read, xe
	const uint32_t j = in_a[t].j;
	if (h.x == CAND_INVALID) return;
	perm[(size_t)cand_base[h.x] + j] = (uint32_t)t;
}

// One lane per candidate, in (read, ordinal) order: slot t takes its candidate through perm[t] (cand_scatter_kernel)
// and holds its result on exit.  Unidirectional backward search from start-1 to the maximal begin.
// Contained-match early exit (the `ok[c].x[2] != curr->a[curr->n-1].x[2]` test of bwt_smem1,
// src/bwt.c:543): the candidates of one pass walk back in lockstep in adjacent lanes; when a
// candidate's interval size equals that of the nearest still-active longer candidate of its pass,
// both have the same occurrences from here on, so it ends at the same begin and the filter would
// drop it -- it stops now and is marked dropped.  Lanes of a pass split across two waves simply
// miss this shortcut (the filter still drops them).
// Phases (round 3): the kernel is bound by instruction issue (197 instructions per wave step at 38 % lane utilisation: a wave steps
// until its longest walk ends, 22.7 times for 8.8 steps per candidate), so the walk is cut into phases: after max_iter steps the
// lanes that are still searching park their state (32 bytes) in a list -- in lane order, so the candidates of a pass stay
// adjacent and in step -- and the next launch (RESUME) continues that list with full waves.
// The list is BWD_NSUB lists, block b appending to list b mod BWD_NSUB (one counter each: appends to a single address run at ~90 per
// microsecond, 650 000 waves would take 7 ms); a list holds at most the lanes of the blocks that feed it (sub_cap).
__global__ void __launch_bounds__(256) iota_kernel(uint32_t *__restrict__ v, uint32_t n) { const uint32_t i = blockIdx.x * 256u + threadIdx.x; if (i < n) v[i] = i; }
#define BWD_NSUB 64
struct bwd_state_t { uint64_t lo, hi; uint32_t read, t; uint16_t x, end, i, beg; };     // 32 B (a parked lane has i >= 0)
template <bool RESUME>
__global__ void __launch_bounds__(256) smem_backward_kernel(fmd_dev_t f, read_view_t rv, uint64_t n_cands, int min_seed_len,
                                                            const cand_t *__restrict__ cand_a, const uint64_t *__restrict__ cand_k,
                                                            const uint32_t *__restrict__ perm,
                                                            res_t *__restrict__ res_a, uint64_t *__restrict__ res_k,
                                                            unsigned long long *__restrict__ stats,
                                                            const bwd_state_t *__restrict__ in_state, const uint32_t *__restrict__ in_count,
                                                            bwd_state_t *__restrict__ out_state, uint32_t *__restrict__ out_count, const uint32_t sub_cap, const int max_iter)
{
	wtrace_scope_t wt_(WT_BACKWARD);
	fmd_wave_prio(f.wave_prio);
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = __lane_id();
	// L2[b] for a per-lane b: one LDS read per step instead of a conditional-move tree over four 64-bit scalars
	__shared__ uint64_t l2_lds[4];
	if (threadIdx.x < 4) l2_lds[threadIdx.x] = f.L2[threadIdx.x];
	__syncthreads();
	const uint32_t sub = blockIdx.x % BWD_NSUB;                              // the list this block appends to (and, resumed, reads from)
	if (RESUME) {
		n_cands = in_count[sub];
		t = (uint64_t)(blockIdx.x / BWD_NSUB) * blockDim.x + threadIdx.x;    // index inside the list
		in_state += (size_t)sub * sub_cap;
		if ((uint64_t)(blockIdx.x / BWD_NSUB) * blockDim.x >= n_cands) return;   // (a resumed launch is sized by the lists' upper bound)
	}
	if (out_state) { out_state += (size_t)sub * sub_cap; out_count += sub; }
	bool live = t < n_cands;
	cand_t c = {CAND_INVALID, 0, 0, 0};
	uint64_t lo = 0, hi = 0;
	int x = 0, end = 0, i = -1, beg = 0;
	uint32_t t_out = (uint32_t)t;
	if (RESUME) {
		if (live) { const bwd_state_t s = in_state[t]; lo = s.lo; hi = s.hi; c.read = s.read; t_out = s.t; x = s.x; end = s.end; i = s.i; beg = s.beg; c.s = (uint32_t)(hi - lo + 1); }
	} else {
		if (live) { const uint32_t src = perm[t]; c = cand_a[src]; lo = cand_k[src]; hi = lo + c.s - 1; }
		x = (int)(c.xe >> 16); end = (int)(c.xe & 0xFFFF);
		i = x - 1; beg = x;
	}
	bool act = live && i >= 0, dropped = false;
	// last lane of my pass segment inside this wave
	// (a parked list holds the waves' survivors in the order the waves finished: two parts of one pass may meet in it the wrong way
	// round, and the lanes above must be the LONGER candidates -- so a segment also ends where the original index does not grow)
	const uint32_t nread = __shfl_down(c.read, 1), nx = __shfl_down((uint32_t)x, 1), nt = __shfl_down(t_out, 1);
	const bool last_of_seg = lane == 63 || !live || nread != c.read || nx != (uint32_t)x || nt <= t_out;
	const unsigned long long segb = __ballot(last_of_seg);
	const int seg_end = lane + __builtin_ctzll(segb >> lane);      // >= lane, bit 63 is always set
	const unsigned long long above = seg_end > lane ? ((~0ull >> (63 - (seg_end - lane - 1))) << 1 << lane) : 0ull;  // lanes lane+1..seg_end
	uint32_t size = c.s;
	unsigned st_mine = 0;                                  // (stats) rank steps of this lane's candidate
	unsigned st_iter = 0, st_steps = 0, st_uniq = 0, st_u0 = (!RESUME && live && i >= 0 && c.s == 1) ? 1u : 0u, st_x0 = (!RESUME && live && i < 0) ? 1u : 0u;
	// the read's packed words are re-fetched only when the walk crosses into the next one (16 / 32 bases)
	uint32_t rw = 0, rm = 0;
	if (act) { rw = rv.pk[(size_t)(i >> 4) * rv.n_reads + c.read]; rm = rv.nm[(size_t)(i >> 5) * rv.n_reads + c.read]; }
	for (int it = 0; it < max_iter && __any(act); ++it) {
		if (stats) { ++st_iter; st_steps += act ? 1u : 0u; st_mine += act ? 1u : 0u; st_uniq += (act && size == 1) ? 1u : 0u; }
		if (act) {
			const int b = (int)((rw >> ((i & 15) << 1)) & 3);
			const bool isn = (rm >> (i & 31)) & 1;
			uint64_t ol, ou;
			fmd_occ1_pair<false>(f, lo - 1, hi, b, ol, ou);       // lo >= 1: first row of a non-empty pattern's interval
			const uint64_t L2b = l2_lds[b];
			const uint64_t nl = L2b + ol + 1, nu = L2b + ou;
			const bool ok = !isn && nl <= nu;
			lo = ok ? nl : lo; hi = ok ? nu : hi; beg = ok ? i : beg;
			size = ok ? (uint32_t)(nu - nl + 1) : size;
			i -= ok ? 1 : 0;
			act = ok && i >= 0;
			if (act && (i & 15) == 15) {
				rw = rv.pk[(size_t)(i >> 4) * rv.n_reads + c.read];
				if ((i & 31) == 31) rm = rv.nm[(size_t)(i >> 5) * rv.n_reads + c.read];
			}
		}
		// nearest longer candidate of my pass that is still searching, after this step
		const unsigned long long am = __ballot(act) & above;
		const int nxt = am ? __builtin_ctzll(am) : lane;
		const uint32_t nsize = __shfl(size, nxt);
		if (act && am && nsize == size) { act = false; dropped = true; }
	}
	if (stats) {
		if (live && !RESUME) atomicAdd(stats + 8 + (st_mine < 39u ? st_mine : 39u), 1ull);      // histogram of rank steps per candidate (0 .. 38, 39+)
		atomicAdd(stats + 1, (unsigned long long)st_steps); atomicAdd(stats + 4, (unsigned long long)st_uniq); atomicAdd(stats + 5, (unsigned long long)st_u0); atomicAdd(stats + 6, (unsigned long long)st_x0); { const bool nowalk = !__any(live && !st_x0); if (!RESUME && lane == 0 && nowalk) atomicAdd(stats + 7, 1ull); }
		if (lane == 0) { atomicAdd(stats, (unsigned long long)st_iter); atomicAdd(stats + 2, 1ull); atomicMax(stats + 3, (unsigned long long)st_iter); }
	}
	// still searching when the phase ends: parked for the next launch, in lane order
	const unsigned long long sm = __ballot(act);
	if (sm) {
		const int first = (int)__builtin_ctzll(sm);
		uint32_t base = 0;
		if (lane == first) base = atomicAdd(out_count, (uint32_t)__builtin_popcountll(sm));
		base = __shfl(base, first);
		if (act) {
			bwd_state_t s; s.lo = lo; s.hi = hi; s.read = c.read; s.t = t_out; s.x = (uint16_t)x; s.end = (uint16_t)end; s.i = (uint16_t)i; s.beg = (uint16_t)beg;
			out_state[base + (uint32_t)__builtin_popcountll(sm & ((1ull << lane) - 1ull))] = s;
		}
	}
	if (live && !act) {
		res_t o;
		o.read = c.read; o.be = ((uint32_t)beg << 16) | (uint32_t)end;
		o.s = (!dropped && end - beg >= min_seed_len) ? (uint32_t)(hi - lo + 1) : 0u;
		o.pad = 0;
		res_a[t_out] = o;
		res_k[t_out] = lo;
	}
}

// ---------------------------------------------------------------- filter

// keep result t unless the next VALID result of the same read has the same begin (results dropped by
// the backward kernel's early exit, or too short, are skipped: an early-dropped candidate ends where
// the next longer one does, so the comparison partner is the first survivor after it)
__global__ void __launch_bounds__(256) smem_filter_kernel(const res_t *__restrict__ res_a, uint64_t n, uint32_t *__restrict__ occ)
{
	wtrace_scope_t wt_(WT_FILTER);
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	res_t e = {0xFFFFFFFEu, 0, 0, 0};
	if (t < n) e = res_a[t];
	bool k = e.s > 0;
	// comparison partner = next valid result; inside the wave it comes from a ballot + shuffle, behind the wave's last
	// valid result from the first valid result of the block's later waves (LDS), and only behind the block's last one
	// from a walk through global memory
	__shared__ uint32_t fv_has[4], fv_read[4], fv_be[4];
	const int lane = __lane_id(), wv = (int)(threadIdx.x >> 6);
	const unsigned long long vm = __ballot(k);
	if (lane == 0) fv_has[wv] = vm ? 1u : 0u;
	if (vm && lane == (int)__builtin_ctzll(vm)) { fv_read[wv] = e.read; fv_be[wv] = e.be; }
	__syncthreads();
	const unsigned long long ab = lane < 63 ? (vm >> (lane + 1)) << (lane + 1) : 0ull;
	const int nl = ab ? __builtin_ctzll(ab) : lane;
	const uint32_t nread = __shfl(e.read, nl), nbe = __shfl(e.be, nl);
	if (k) {
		if (ab) { if (nread == e.read && (nbe >> 16) == (e.be >> 16)) k = false; }
		else {
			int w2 = wv + 1;
			while (w2 < 4 && !fv_has[w2]) ++w2;
			if (w2 < 4) { if (fv_read[w2] == e.read && (fv_be[w2] >> 16) == (e.be >> 16)) k = false; }
			else {
				for (uint64_t u = (uint64_t)(blockIdx.x + 1) * blockDim.x; u < n; ++u) {
					res_t nx = res_a[u];
					if (nx.read != e.read) break;
					if (nx.s == 0) continue;
					if ((nx.be >> 16) == (e.be >> 16)) k = false;
					break;
				}
			}
		}
	}
	if (t <= n) occ[t] = k ? e.s : 0u;                  // (a kept result has at least one occurrence: occ != 0 is the keep flag)
}

// The scan over the occurrence counts carries the number of kept results in its high bits (one pass instead of a second array and
// a reduction): element t contributes occ[t] | (occ[t] != 0) << 36; totals stay below 2^32 occurrences (checked) / 2^28 results.
#define OCC_OFF_SHIFT 36
#define OCC_OFF_MASK ((1ull << OCC_OFF_SHIFT) - 1ull)
struct occ_keep_in {
	const uint32_t *occ;
	__device__ uint64_t operator()(uint64_t t) const { const uint32_t v = occ[t]; return (uint64_t)v | ((uint64_t)(v != 0u) << OCC_OFF_SHIFT); }
};
__global__ void __launch_bounds__(256) per_read_counts_kernel(const uint32_t *__restrict__ cand_base, const uint64_t *__restrict__ occ_off,
                                                              uint32_t n_reads, uint32_t *__restrict__ n_ref_pos, uint32_t *__restrict__ prefix)
{
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	uint64_t a = occ_off[cand_base[r]] & OCC_OFF_MASK, b = occ_off[cand_base[r + 1]] & OCC_OFF_MASK;
	prefix[r] = (uint32_t)a;
	n_ref_pos[r] = (uint32_t)(b - a);
}

// ---------------------------------------------------------------- expand

// rows k..k+s-1 of each kept SMEM, plus the qbeg/score columns.  Small groups are written
// by their own lane; large groups by the whole wave, one group at a time.
__global__ void __launch_bounds__(256) expand_kernel(const res_t *__restrict__ res_a, const uint64_t *__restrict__ res_k,
                                                     const uint32_t *__restrict__ occ, const uint64_t *__restrict__ occ_off, uint64_t n,
                                                     uint64_t *__restrict__ rows, int2 *__restrict__ qbeg, uint32_t *__restrict__ score)
{
	wtrace_scope_t wt_(WT_EXPAND);
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t s = 0;
	uint64_t k = 0, off = 0;
	int2 qb = make_int2(0, 0);
	if (t < n) {
		s = occ[t];
		if (s) { res_t e = res_a[t]; k = res_k[t]; off = occ_off[t] & OCC_OFF_MASK; qb = make_int2((int)(e.be >> 16), (int)(e.be & 0xFFFF)); }
	}
	const uint32_t SMALL = 4;
	if (s && s <= SMALL) {
		for (uint32_t u = 0; u < s; ++u) { rows[off + u] = k + u; qbeg[off + u] = qb; score[off + u] = u ? 0u : s; }
	}
	unsigned long long big = __ballot(s > SMALL);
	int lane = __lane_id();
	while (big) {
		int src = __ffsll((long long)big) - 1;
		big &= big - 1;
		uint32_t ss = __shfl(s, src);
		uint64_t kk = __shfl(k, src), oo = __shfl(off, src);
		int2 q2 = make_int2(__shfl(qb.x, src), __shfl(qb.y, src));
		for (uint32_t u = lane; u < ss; u += 64) { rows[oo + u] = kk + u; qbeg[oo + u] = q2; score[oo + u] = u ? 0u : ss; }
	}
}

// ---------------------------------------------------------------- locate

// Each wave owns LOCATE_PER_WAVE consecutive occurrences and keeps its 64 lanes busy: a lane whose
// walk reaches a sampled row stores its position and takes the next unclaimed occurrence of the
// wave's chunk (wave-uniform cursor + ballot prefix, no atomics), so lanes do not idle behind the
// longest walk of the wave (walk lengths are 0..sa_intv-1 and unpredictable).
#ifndef LOCATE_PER_WAVE
#define LOCATE_PER_WAVE 512
#endif
__global__ void __launch_bounds__(256) locate_kernel(fmd_dev_t f, uint64_t *__restrict__ rows, uint64_t n)
{
	wtrace_scope_t wt_(WT_LOCATE);
	fmd_wave_prio(f.wave_prio);
	const int lane = __lane_id();
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint64_t c0 = wave * LOCATE_PER_WAVE;
	if (c0 >= n) return;
	const uint64_t c1 = min(n, c0 + (uint64_t)LOCATE_PER_WAVE);
	const uint64_t mask = (1ull << f.sa_shift) - 1;
	uint64_t cur = c0;                 // wave-uniform: next unclaimed occurrence
	uint64_t t = 0, k = 0, steps = 0;
	bool busy = false;
	for (;;) {
		// finished or idle lanes claim new work
		const bool need = !busy;
		const unsigned long long nm = __ballot(need);
		if (nm) {
			const uint64_t mine = cur + __popcll(nm & ((1ull << lane) - 1));
			if (need && mine < c1) { t = mine; k = rows[t]; steps = 0; busy = true; }
			cur += __popcll(nm);
		}
		if (!__any(busy)) break;
		if (busy) {
			if (k & mask) { k = fmd_inv_psi(f, k); ++steps; }
			else {
				uint64_t idx = k >> f.sa_shift, pos;
				if (idx == 0) pos = steps - 1;
				else {
					uint64_t hb = (f.sa_bits[idx >> 5] >> (idx & 31)) & 1u;
					pos = ((uint64_t)f.sa[idx] | (hb << 32)) + steps;
				}
				rows[t] = pos;
				busy = false;
			}
		}
	}
}

// ---------------------------------------------------------------- fused forward + backward

// One lane per read runs the WHOLE SMEM search of its read: forward pass, then the backward search of
// that pass's candidates (longest first), then the next pass.  The candidates of the current pass live
// in a per-read scratch column in HBM ([slot][read], 16 bytes each: written once, read once, mostly
// served by L2 / Infinity Cache) instead of going through a global list, a scan, a scatter and a second
// kernel; only the SMEMs themselves (~2-3 per read instead of ~35 candidates) leave the kernel.
//
// Every loop iteration a lane issues at most ONE rank pair (the only HBM-missing access), and the rank
// pair is computed by code common to both phases, so the lanes of a wave stay convergent on the
// 32-byte block gathers whatever phase each lane is in.
//
// Contained matches (src/bwt.c:535-543): candidates are searched longest first, so a lane needs only an
// O(1) summary of the last longer candidate that finished its own search: final interval size s_fin,
// the position p1 from which it had that size, and its begin b.  A shorter candidate whose interval
// size equals s_fin at a position in [b, p1] has the same occurrences from there on, ends at the same
// begin and is dropped on the spot (bwt_smem1 drops it at the first equal size, never later than p1).
// A candidate that finishes at the same begin as that summary is contained too (both stopped at an
// ambiguous base or at the read start).
enum { FS_OPEN = 0, FS_FWD, FS_BINIT, FS_BWD, FS_IDLE };

// Work distribution: a persistent grid (one resident wave per wave slot) pulls 64-read chunks from a
// global queue; inside a wave every lane that finishes its read immediately takes the next read of
// the wave's chunk (wave-uniform cursor + ballot prefix).  Reads differ a lot in cost (reads from
// repeat families have many more candidates), and a static read-per-lane mapping leaves most lanes
// of every wave idle behind its slowest read.
#define FUSED_CHUNK 64u

__global__ void __launch_bounds__(256) smem_fused_kernel(fmd_dev_t f, read_view_t rv, const uint32_t *__restrict__ lens, int min_seed_len,
                                                         uint4 *__restrict__ scratch, cand_t *__restrict__ out_a, uint64_t *__restrict__ out_k,
                                                         unsigned long long *counter, uint64_t cap, uint32_t *__restrict__ n_ref_pos,
                                                         unsigned long long *n_cand_total, unsigned int *queue)
{
	const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;      // scratch columns are per resident lane
	const int lane = __lane_id();
	uint32_t r = 0;
	int len = 0;
	int st = FS_IDLE;
	int i = 0, x = 0;                 // forward / resume position, pass start
	uint64_t k = 0, l = 0, s = 0;     // forward bi-interval (k, l, s); backward: [k, l] = SA interval
	int nc = 0, cidx = -1;            // candidates of the pass; next one to search backward
	int bi = 0, beg = 0, cend = 0, p1 = 0;
	uint32_t size = 0;
	bool ref_valid = false; uint32_t ref_s = 0; int ref_p1 = 0, ref_b = 0;
	uint32_t occ_sum = 0, cand_sum = 0;
	cand_cursor_t cc = {0, 0};
	uint32_t ch_cur = 0, ch_end = 0;  // wave-uniform: unclaimed reads of the wave's current chunk
	bool drained = false;             // wave-uniform: the global queue is empty
	unsigned long long n_iter = 0, n_rank = 0;
#define FS_PUSH(END_) do { if ((END_) >= min_seed_len) { \
		scratch[(size_t)nc * stride + gtid] = make_uint4((uint32_t)k, (uint32_t)s, (uint32_t)(k >> 32) | ((uint32_t)(END_) << 16), 0u); ++nc; } } while (0)
	for (;;) {
		// ---- refill: idle lanes take the next reads of the chunk; an empty chunk is replaced from the queue
		{
			const unsigned long long im = __ballot(st == FS_IDLE);
			if (im) {
				if (ch_cur == ch_end && !drained) {
					uint32_t b0 = 0;
					if (lane == 0) b0 = atomicAdd(queue, FUSED_CHUNK);
					b0 = __shfl(b0, 0);
					if (b0 >= rv.n_reads) drained = true;
					else { ch_cur = b0; ch_end = min(b0 + FUSED_CHUNK, rv.n_reads); }
				}
				const uint32_t avail = ch_end - ch_cur;
				if (avail) {
					const uint32_t rank = (uint32_t)__popcll(im & ((1ull << lane) - 1));
					if (st == FS_IDLE && rank < avail) {
						r = ch_cur + rank; len = (int)lens[r];
						i = 0; occ_sum = 0;
						if (len > 0) st = FS_OPEN; else n_ref_pos[r] = 0;
					}
					ch_cur += min(avail, (uint32_t)__popcll(im));
				}
			}
			if (drained && ch_cur == ch_end && !__any(st != FS_IDLE)) break;
		}
		// ---- A: at most one base fetch (+ one scratch fetch) per iteration, issued together
		if (st == FS_BINIT && cidx < 0) st = FS_OPEN;                 // pass finished: resume the forward scan at i
		const bool fetch_cand = st == FS_BINIT;
		const int pos = fetch_cand ? x - 1 : (st == FS_BWD ? bi : i);
		const bool at_end = st == FS_OPEN && i >= len;
		const bool need_base = (st == FS_OPEN && !at_end) || st == FS_FWD || ((st == FS_BWD || fetch_cand) && pos >= 0);
		int base = 4;
		if (need_base) base = read_base(rv, r, pos);
		if (fetch_cand) {
			const uint4 c = scratch[(size_t)cidx * stride + gtid];
			k = (uint64_t)c.x | ((uint64_t)(c.z & 0xFFFFu) << 32);
			size = c.y; cend = (int)(c.z >> 16);
			l = k + size - 1;
			bi = x - 1; beg = x; p1 = x;
			st = FS_BWD;
		}
		bool want_rank = false, fin = false, merged = false;
		uint64_t q1 = 0, q2 = 0;
		if (st == FS_OPEN) {
			if (at_end) { n_ref_pos[r] = occ_sum; st = FS_IDLE; }
			else {
				++i;
				if (base < 4) {
					x = i - 1;
					k = fmd_L2(f, base) + 1; s = fmd_L2(f, base + 1) - fmd_L2(f, base); l = fmd_L2(f, 3 - base) + 1;
					nc = 0;
					if (i == len) { FS_PUSH(i); cidx = nc - 1; cand_sum += (uint32_t)nc; ref_valid = false; st = FS_BINIT; }
					else st = FS_FWD;
				}
			}
		} else if (st == FS_FWD) {
			if (base < 4) { want_rank = true; q1 = l - 1; q2 = l - 1 + s; }
			else { FS_PUSH(i); cidx = nc - 1; cand_sum += (uint32_t)nc; ref_valid = false; st = FS_BINIT; }   // next pass opens at i
		} else if (st == FS_BWD) {
			if (bi < 0 || base > 3) fin = true;
			else { want_rank = true; q1 = k - 1; q2 = l; }
		}
		// ---- B: the rank pair, common to both phases
		uint64_t tk[4] = {0, 0, 0, 0}, tl[4] = {0, 0, 0, 0};
		if (want_rank) fmd_occ4_pair(f, q1, q2, tk, tl);
		++n_iter; n_rank += (unsigned long long)__popcll(__ballot(want_rank));
		// ---- C: apply
		if (want_rank && st == FS_FWD) {
			const int cb = 3 - base;
			uint64_t os[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) os[q] = tl[q] - tk[q];
			const uint64_t nk3 = k + ((l <= f.primary) & (l + s - 1 >= f.primary));
			const uint64_t nk2 = nk3 + os[3], nk1 = nk2 + os[2], nk0 = nk1 + os[1];
			const uint64_t ns = cb == 0 ? os[0] : cb == 1 ? os[1] : cb == 2 ? os[2] : os[3];
			const uint64_t nk = cb == 0 ? nk0 : cb == 1 ? nk1 : cb == 2 ? nk2 : nk3;
			const uint64_t nl = fmd_L2(f, cb) + 1 + (cb == 0 ? tk[0] : cb == 1 ? tk[1] : cb == 2 ? tk[2] : tk[3]);
			bool pass_end = false;
			if (ns != s) FS_PUSH(i);                                   // interval size changes: [x, i) is a candidate
			if (ns == 0) pass_end = true;                              // next pass opens at i (src/bwt.c:519)
			else {
				k = nk; l = nl; s = ns; ++i;
				if (i == len) { FS_PUSH(i); pass_end = true; }          // reached the read end: push the last interval
			}
			if (pass_end) { cidx = nc - 1; cand_sum += (uint32_t)nc; ref_valid = false; st = FS_BINIT; }
		}
		bool keep = false;
		cand_t o = {r, 0, 0, 0};
		uint64_t ok = 0;
		if (st == FS_BWD) {
			if (want_rank) {
				const uint64_t ol = base == 0 ? tk[0] : base == 1 ? tk[1] : base == 2 ? tk[2] : tk[3];
				const uint64_t ou = base == 0 ? tl[0] : base == 1 ? tl[1] : base == 2 ? tl[2] : tl[3];
				const uint64_t nl = fmd_L2(f, base) + ol + 1, nu = fmd_L2(f, base) + ou;
				if (nl > nu) fin = true;
				else {
					const uint32_t nsz = (uint32_t)(nu - nl + 1);
					if (nsz != size) p1 = bi;
					k = nl; l = nu; size = nsz; beg = bi;
					if (ref_valid && bi >= ref_b && bi <= ref_p1 && size == ref_s) { fin = true; merged = true; }
					else { --bi; if (bi < 0) fin = true; }
				}
			}
			if (fin) {
				keep = !merged && (cend - beg >= min_seed_len) && !(ref_valid && beg == ref_b);
				if (!merged) { ref_valid = true; ref_s = size; ref_p1 = p1; ref_b = beg; }
				o.xe = ((uint32_t)beg << 16) | (uint32_t)cend; o.s = size; ok = k;
				if (keep) occ_sum += size;
				--cidx;
				st = FS_BINIT;
			}
		}
		cand_append(keep, o, ok, out_a, out_k, counter, cap, cc);
	}
#undef FS_PUSH
	cand_fill_invalid(cc, out_a, cap);
	// candidate count statistic: one atomic per wave
	unsigned long long tot = cand_sum;
	for (int off = 32; off; off >>= 1) tot += __shfl_down(tot, off);
	if (lane == 0 && tot) atomicAdd(n_cand_total, tot);
	if (lane == 0) { atomicAdd(n_cand_total + 3, n_iter); atomicAdd(n_cand_total + 4, n_rank); atomicAdd(n_cand_total + 5, 1ull); }
}

// sort key of an SMEM record: (read << 16) | end; fillers sort to the end
__global__ void __launch_bounds__(256) smem_key_kernel(const cand_t *__restrict__ in_a, uint64_t n_list, uint64_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals, unsigned long long *n_valid)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool v = false;
	if (t < n_list) {
		cand_t c = in_a[t];
		v = c.read != CAND_INVALID;
		keys[t] = v ? (((uint64_t)c.read << 16) | (c.xe & 0xFFFFu)) : ~0ull;
		vals[t] = (uint32_t)t;
	}
	unsigned long long m = __ballot(v);
	if (__lane_id() == 0 && m) atomicAdd(n_valid, (unsigned long long)__popcll(m));
}

// sorted order -> result arrays + occurrence counts
__global__ void __launch_bounds__(256) smem_gather_kernel(const cand_t *__restrict__ in_a, const uint64_t *__restrict__ in_k,
                                                          const uint32_t *__restrict__ vals, uint64_t n_valid, res_t *__restrict__ res_a,
                                                          uint64_t *__restrict__ res_k, uint32_t *__restrict__ occ)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t > n_valid) return;
	if (t == n_valid) { occ[t] = 0; return; }
	const uint32_t src = vals[t];
	const cand_t c = in_a[src];
	res_t o = {c.read, c.xe, c.s, 0};
	res_a[t] = o;
	res_k[t] = in_k[src];
	occ[t] = c.s;
}

// ---------------------------------------------------------------- host side

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

struct bmh_seed_ws {
	uint32_t max_reads; uint64_t max_bases, max_cands, max_occ;
	uint32_t *pk, *nm;
	uint32_t n_grp_cap;
	cand_t *cand_a; uint64_t *cand_k;
	res_t *res_a; uint64_t *res_k;
	uint32_t *n_cand, *cand_base;       // [max_reads+1]
	uint32_t *occ;                      // [max_cands+1]
	uint64_t *occ_off;                  // [max_cands+1]
	uint64_t *rows; int2 *qbeg; uint32_t *score;   // [max_occ]
	uint32_t *n_ref_pos, *prefix;       // [max_reads]
	unsigned long long *counter;
	uint4 *scratch; size_t scratch_entries;       // fused kernel: [slot][read] candidate columns
	uint64_t *skeys, *skeys2; uint32_t *svals, *svals2;   // SMEM sort
	bwd_state_t *bwd_state[2]; uint64_t bwd_state_cap;   // parked walks between the phases of the backward search (allocated on first use)
	uint32_t *bwd_cnt;                                   // [8][BWD_NSUB] list lengths per phase
	uint32_t *fwd_iters; uint32_t fwd_iters_n;           // experiment SEED_FWD_SORT: iterations of every read in the last forward search
	void *scan_tmp; size_t scan_tmp_bytes;
	hipEvent_t ev[8];
	float ms[7];
	// the seeding kernels run on a stream of the highest priority (ordered behind and before the caller's by two events): their waves wait on HBM most of the
	// time, and beside another batch's extension -- whose waves fill the register files -- they are the ones that should get the slots that become free
	hipStream_t st_hi; hipEvent_t ev_in, ev_out;
};

extern "C" bmh_seed_ws_t *bmh_seed_ws_create(uint32_t max_reads, uint64_t max_bases, uint64_t max_cands, uint64_t max_occ)
{
	if (max_reads == 0 || max_bases == 0) { bmh_set_error("bmh_seed_ws_create: empty capacity"); return nullptr; }
	bmh_seed_ws *w = (bmh_seed_ws *)calloc(1, sizeof(bmh_seed_ws));
	w->max_reads = max_reads; w->max_bases = max_bases;
	w->max_cands = max_cands ? max_cands : (uint64_t)max_reads * 16 + max_bases * 2 / 5;
	w->max_cands += ((uint64_t)max_reads / 64 + 1) * (CAND_CHUNK + 64);   // chunked appends: one open chunk per wave
	w->max_occ = max_occ ? max_occ : (uint64_t)max_reads * 64;
	// packed-read buffers are sized by the longest read of a batch: allocated on first use
	w->n_grp_cap = 0;
	bool ok = true;
#define A(p, n) ok = ok && hipMalloc((void **)&(p), (size_t)(n)) == hipSuccess
	A(w->cand_a, sizeof(cand_t) * w->max_cands); A(w->cand_k, 8 * w->max_cands);
	A(w->res_a, sizeof(res_t) * (w->max_cands + 1)); A(w->res_k, 8 * (w->max_cands + 1));
	A(w->n_cand, 4 * ((size_t)max_reads + 1)); A(w->cand_base, 4 * ((size_t)max_reads + 1));
	A(w->occ, 4 * (w->max_cands + 1));
	A(w->occ_off, 8 * (w->max_cands + 1));
	A(w->rows, 8 * w->max_occ); A(w->qbeg, 8 * w->max_occ); A(w->score, 4 * w->max_occ);
	A(w->n_ref_pos, 4 * (size_t)max_reads); A(w->prefix, 4 * (size_t)max_reads);
	A(w->counter, 1024);
	A(w->bwd_cnt, 4 * 8 * BWD_NSUB);
	A(w->skeys, 8 * w->max_cands); A(w->skeys2, 8 * w->max_cands); A(w->svals, 4 * w->max_cands); A(w->svals2, 4 * w->max_cands);
	size_t t1 = 0, t2 = 0, t3 = 0;
	rocprim::radix_sort_pairs(nullptr, t3, w->skeys, w->skeys2, w->svals, w->svals2, (size_t)w->max_cands, 0, 64, 0);
	rocprim::exclusive_scan(nullptr, t1, w->n_cand, w->cand_base, 0u, (size_t)max_reads + 1, rocprim::plus<uint32_t>(), 0);
	rocprim::exclusive_scan(nullptr, t2, w->occ, w->occ_off, (uint64_t)0, w->max_cands + 1, rocprim::plus<uint64_t>(), 0);
	w->scan_tmp_bytes = t1 > t2 ? t1 : t2;
	if (t3 > w->scan_tmp_bytes) w->scan_tmp_bytes = t3;
	A(w->scan_tmp, w->scan_tmp_bytes + 256);
#undef A
	for (int i = 0; i < 8; ++i) ok = ok && hipEventCreate(&w->ev[i]) == hipSuccess;
	{
		const char *pe = getenv("BMH_SEED_PRIO");                    // (BMH_SEED_PRIO=normal: A/B -- the kernels on the caller's stream)
		if (!(pe && pe[0] == 'n')) {
			int prio_lo = 0, prio_hi = 0;
			(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
			// (BMH_SEED_CUS=n, measurement knob: the seeding stage on a stream confined to n of the chip's compute units -- a pseudo-random subset, so that
			// whatever order the driver hands mask bits to XCDs and shader engines in, every one of them keeps its share.  The gather-bound kernels
			// lose less than proportionally on fewer units (what they wait for is memory), and the units they do not take stay with the other
			// batch's extension, whose blocks cannot be dislodged by a flood of small waves there: DESIGN.md section 5)
			const char *cue = getenv("BMH_SEED_CUS");
			const int n_cus = cue ? atoi(cue) : 0;
			if (n_cus > 0 && n_cus < 256) {
				uint32_t cumask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
				for (int i = 0; i < n_cus; ++i) { const int b = (i * 167 + 13) & 255; cumask[b >> 5] |= 1u << (b & 31); }
				ok = ok && hipExtStreamCreateWithCUMask(&w->st_hi, 8, cumask) == hipSuccess;
			} else
			ok = ok && hipStreamCreateWithPriority(&w->st_hi, hipStreamNonBlocking, prio_hi) == hipSuccess;
			ok = ok && hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&w->ev_out, hipEventDisableTiming) == hipSuccess;
		}
	}
	if (!ok) { bmh_set_error("bmh_seed_ws_create: hipMalloc failed (%s)", hipGetErrorString(hipGetLastError())); bmh_seed_ws_free(w); return nullptr; }
	return w;
}

extern "C" void bmh_seed_ws_free(bmh_seed_ws_t *w)
{
	if (!w) return;
	void *ps[] = {w->pk, w->nm, w->cand_a, w->cand_k, w->res_a, w->res_k, w->n_cand, w->cand_base, w->occ,
	              w->occ_off, w->rows, w->qbeg, w->score, w->n_ref_pos, w->prefix, w->counter, w->scan_tmp,
	              w->scratch, w->skeys, w->skeys2, w->svals, w->svals2, w->bwd_state[0], w->bwd_state[1], w->bwd_cnt, w->fwd_iters};
	for (void *p : ps) if (p) (void)hipFree(p);
	for (int i = 0; i < 8; ++i) if (w->ev[i]) (void)hipEventDestroy(w->ev[i]);
	if (w->st_hi) (void)hipStreamDestroy(w->st_hi);
	if (w->ev_in) (void)hipEventDestroy(w->ev_in);
	if (w->ev_out) (void)hipEventDestroy(w->ev_out);
	free(w);
}

extern "C" void bmh_seed_last_timing(const bmh_seed_ws_t *w, float ms[7]) { memcpy(ms, w->ms, sizeof(float) * 7); }

static inline unsigned nblk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

// room for the walks a phase of the backward search parks (at most every candidate), two lists used in turn
static int bwd_state_reserve(bmh_seed_ws *w, uint64_t n)
{
	if (n <= w->bwd_state_cap) return BMH_OK;
	for (int k = 0; k < 2; ++k) { if (w->bwd_state[k]) (void)hipFree(w->bwd_state[k]); w->bwd_state[k] = nullptr; }
	w->bwd_state_cap = 0;
	const uint64_t cap = n + n / 8 + 1024;
	for (int k = 0; k < 2; ++k)
		if (hipMalloc((void **)&w->bwd_state[k], sizeof(bwd_state_t) * cap) != hipSuccess) { bmh_set_error("bmh_seed_batch: no memory for %llu parked walks of the backward search", (unsigned long long)cap); return BMH_ENOMEM; }
	w->bwd_state_cap = cap;
	return BMH_OK;
}

// the occurrences of a batch are known before anything is written to the three output arrays: when they do not fit, the arrays
// are replaced by larger ones (the default capacity, 64 per read, is a guess -- reads from high-copy repeats have thousands)
static int grow_occ(bmh_seed_ws *w, uint64_t need, bmh_seeds_t *out)
{
	if (need <= w->max_occ) return BMH_OK;
	(void)hipFree(w->rows); (void)hipFree(w->qbeg); (void)hipFree(w->score);
	w->rows = nullptr; w->qbeg = nullptr; w->score = nullptr; w->max_occ = 0;
	const uint64_t cap = need + need / 4 + 1024;
	if (hipMalloc((void **)&w->rows, 8 * cap) != hipSuccess || hipMalloc((void **)&w->qbeg, 8 * cap) != hipSuccess || hipMalloc((void **)&w->score, 4 * cap) != hipSuccess) {
		bmh_set_error("bmh_seed_batch: %llu occurrences: no memory for the output arrays (%s)", (unsigned long long)need, hipGetErrorString(hipGetLastError()));
		return BMH_ECAPACITY;
	}
	w->max_occ = cap;
	out->d_rbeg = w->rows; out->d_qbeg = (const int32_t *)w->qbeg; out->d_score = w->score;
	return BMH_OK;
}

static int seed_batch_on(bmh_seed_ws_t *w, const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs,
                         const uint32_t *d_lens, uint32_t n_reads, int min_seed_len, void *stream_, bmh_seeds_t *out);
extern "C" int bmh_seed_batch(bmh_seed_ws_t *w, const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs,
                              const uint32_t *d_lens, uint32_t n_reads, int min_seed_len, void *stream_, bmh_seeds_t *out)
{
	if (!w || !idx || !out) { bmh_set_error("bmh_seed_batch: null argument"); return BMH_EINVAL; }
	if (!w->st_hi) return seed_batch_on(w, idx, d_reads, d_offs, d_lens, n_reads, min_seed_len, stream_, out);
	// behind what the caller's stream holds -- the HOST waits for it (the call waits for the stream several times anyway): a barrier packet that waits in the
	// high-priority queue was measured to cost what the priority gains (34.1 against 35.3 Mreads/s) --, on the workspace's own stream, and the caller's
	// stream behind it again (whatever the outcome)
	hipStream_t su = (hipStream_t)stream_;
	HIPCK(hipStreamSynchronize(su));
	const int rc = seed_batch_on(w, idx, d_reads, d_offs, d_lens, n_reads, min_seed_len, (void *)w->st_hi, out);
	HIPCK(hipEventRecord(w->ev_out, w->st_hi));
	HIPCK(hipStreamWaitEvent(su, w->ev_out, 0));
	return rc;
}
static int seed_batch_on(bmh_seed_ws_t *w, const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs,
                         const uint32_t *d_lens, uint32_t n_reads, int min_seed_len, void *stream_, bmh_seeds_t *out)
{
	memset(out, 0, sizeof(*out));
	if (n_reads > w->max_reads) { bmh_set_error("bmh_seed_batch: %u reads > workspace capacity %u", n_reads, w->max_reads); return BMH_ECAPACITY; }
	if (min_seed_len < 1) { bmh_set_error("bmh_seed_batch: min_seed_len < 1"); return BMH_EINVAL; }
	hipStream_t st = (hipStream_t)stream_;
	out->d_rbeg = w->rows; out->d_qbeg = (const int32_t *)w->qbeg; out->d_score = w->score;
	out->d_n_ref_pos = w->n_ref_pos; out->d_prefix = w->prefix;
	memset(w->ms, 0, sizeof(w->ms));
	if (n_reads == 0) return BMH_OK;
	fmd_dev_t f = idx->dev;
	f.wave_prio = bmh_tune("SEED_SETPRIO", 0);
	const unsigned lds_pad = (unsigned)bmh_tune("SEED_LDS_PAD", 0);      // (measurement knob: unused dynamic LDS per block of the gather-bound kernels = a cap on their blocks per CU)
	// longest read -> packed geometry (host needs it; one small D2H reduce)
	uint32_t max_len = 0;
	{
		uint32_t *d_max = (uint32_t *)w->counter + 2;
		size_t tb = w->scan_tmp_bytes;
		HIPCK(rocprim::reduce(w->scan_tmp, tb, d_lens, d_max, 0u, (size_t)n_reads, rocprim::maximum<uint32_t>(), st));
		HIPCK(hipMemcpyAsync(&max_len, d_max, 4, hipMemcpyDeviceToHost, st));
		HIPCK(hipStreamSynchronize(st));
	}
	if (max_len > 65535) { bmh_set_error("bmh_seed_batch: read longer than 65535 bases"); return BMH_EINVAL; }
	uint32_t n_grp = (max_len + 31) / 32;
	if (n_grp == 0) n_grp = 1;
	if (n_grp > w->n_grp_cap) {     // grow the packed-read buffers (first batch, or a longer read length)
		if (w->pk) (void)hipFree(w->pk);
		if (w->nm) (void)hipFree(w->nm);
		w->pk = w->nm = nullptr; w->n_grp_cap = 0;
		HIPCK(hipMalloc((void **)&w->pk, (size_t)8 * n_grp * w->max_reads));
		HIPCK(hipMalloc((void **)&w->nm, (size_t)4 * n_grp * w->max_reads));
		w->n_grp_cap = n_grp;
	}
	read_view_t rv = {w->pk, w->nm, n_reads};

	HIPCK(hipEventRecord(w->ev[0], st));
	{
		const uint32_t lds_bytes = PACK_READS_PER_BLOCK * n_grp * 32 + 64;   // 64 reads of the longest length, + alignment slack
		pack_reads_kernel<<<nblk(n_reads, PACK_READS_PER_BLOCK), 256, lds_bytes <= 65536 ? lds_bytes : 0, st>>>(d_reads, d_offs, d_lens, n_reads, n_grp, lds_bytes <= 65536 ? lds_bytes : 0, w->pk, w->nm);
	}
	// Two pipelines produce identical output (both are run by the GPU parity tests):
	//   default      pack | forward | scan | scatter | backward (lane per candidate) | filter | scans | expand | locate
	//   BMH_SEED_FUSED=1  pack | fused forward+backward (lane per read, persistent grid) | sort | scans | expand | locate
	// The fused form moves ~15x fewer bytes between kernels but keeps fewer gathers in flight per CU; on MI355X
	// both are bound by the random 32-byte gather rate and the split form is currently faster.
	static const bool use_fused = getenv("BMH_SEED_FUSED") != nullptr;
	if (use_fused) {
		// ---- fused pipeline: pack | fused forward+backward | sort SMEMs by (read, end) | scans | expand | locate
		const uint32_t depth = max_len >= (uint32_t)min_seed_len ? max_len - (uint32_t)min_seed_len + 1 : 1;
		// persistent grid: as many 256-thread blocks as the chip keeps resident, never more than the reads need
		static int blocks_per_cu = 0, n_cu = 0;
		if (!blocks_per_cu) {
			int dev = 0; hipDeviceProp_t prop;
			HIPCK(hipGetDevice(&dev)); HIPCK(hipGetDeviceProperties(&prop, dev));
			n_cu = prop.multiProcessorCount;
			HIPCK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, smem_fused_kernel, 256, 0));
			if (blocks_per_cu < 1) blocks_per_cu = 1;
		}
		unsigned grid = (unsigned)(n_cu * blocks_per_cu);
		if (grid > nblk(n_reads, 256)) grid = nblk(n_reads, 256);
		const size_t want = (size_t)depth * grid * 256;
		if (want > w->scratch_entries) {
			if (w->scratch) (void)hipFree(w->scratch);
			w->scratch = nullptr; w->scratch_entries = 0;
			HIPCK(hipMalloc((void **)&w->scratch, want * sizeof(uint4)));
			w->scratch_entries = want;
		}
		HIPCK(hipEventRecord(w->ev[1], st));
		HIPCK(hipMemsetAsync(w->counter, 0, 64, st));
		smem_fused_kernel<<<grid, 256, 0, st>>>(f, rv, d_lens, min_seed_len, w->scratch, w->cand_a, w->cand_k, w->counter,
		                                        w->max_cands, w->n_ref_pos, w->counter + 1, (unsigned int *)(w->counter + 3));
		HIPCK(hipEventRecord(w->ev[2], st));
		unsigned long long cnt[8] = {0};
		HIPCK(hipMemcpyAsync(cnt, w->counter, 64, hipMemcpyDeviceToHost, st));
		HIPCK(hipStreamSynchronize(st));
		const unsigned long long n_list = cnt[0];
		out->n_cands = cnt[1];
		if (getenv("BMH_SEED_STATS")) fprintf(stderr, "[fused] grid %u blocks, waves %llu, wave-iterations %llu (%.0f per wave), rank requests %llu (%.1f per iteration), cands %llu\n", grid, cnt[6], cnt[4], (double)cnt[4] / (double)(cnt[6] ? cnt[6] : 1), cnt[5], (double)cnt[5] / (double)(cnt[4] ? cnt[4] : 1), cnt[1]);
		if (n_list > w->max_cands) { bmh_set_error("bmh_seed_batch: %llu SMEM slots > capacity %llu", n_list, (unsigned long long)w->max_cands); return BMH_ECAPACITY; }
		unsigned long long n_valid = 0;
		if (n_list) {
			smem_key_kernel<<<nblk(n_list, 256), 256, 0, st>>>(w->cand_a, n_list, w->skeys, w->svals, w->counter + 2);
			size_t tb = w->scan_tmp_bytes;
			HIPCK(rocprim::radix_sort_pairs(w->scan_tmp, tb, w->skeys, w->skeys2, w->svals, w->svals2, (size_t)n_list, 0, 64, st));
			HIPCK(hipMemcpyAsync(&n_valid, w->counter + 2, 8, hipMemcpyDeviceToHost, st));
			HIPCK(hipStreamSynchronize(st));
		}
		HIPCK(hipEventRecord(w->ev[3], st));
		smem_gather_kernel<<<nblk(n_valid + 1, 256), 256, 0, st>>>(w->cand_a, w->cand_k, w->svals2, n_valid, w->res_a, w->res_k, w->occ);
		{
			size_t tb = w->scan_tmp_bytes;
			HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, w->occ, w->occ_off, (uint64_t)0, (size_t)n_valid + 1, rocprim::plus<uint64_t>(), st));
			tb = w->scan_tmp_bytes;
			HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, w->n_ref_pos, w->prefix, 0u, (size_t)n_reads, rocprim::plus<uint32_t>(), st));
		}
		uint64_t tot = 0;
		HIPCK(hipMemcpyAsync(&tot, w->occ_off + n_valid, 8, hipMemcpyDeviceToHost, st));
		HIPCK(hipStreamSynchronize(st));
		out->n_seeds = tot; out->n_smems = n_valid;
		if (grow_occ(w, tot, out) != BMH_OK) return BMH_ECAPACITY;
		if (tot >> 32) { bmh_set_error("bmh_seed_batch: more than 2^32 occurrences in one batch"); return BMH_ECAPACITY; }
		HIPCK(hipEventRecord(w->ev[4], st));
		if (n_valid)
			expand_kernel<<<nblk(n_valid, 256), 256, 0, st>>>(w->res_a, w->res_k, w->occ, w->occ_off, n_valid, w->rows, w->qbeg, w->score);
		HIPCK(hipEventRecord(w->ev[5], st));
		if (tot)
			locate_kernel<<<nblk(nblk(tot, LOCATE_PER_WAVE) * 64ull, 256), 256, 0, st>>>(f, w->rows, tot);
		HIPCK(hipEventRecord(w->ev[6], st));
		HIPCK(hipStreamSynchronize(st));
		HIPCK(hipGetLastError());
		for (int i = 0; i < 6; ++i) (void)hipEventElapsedTime(&w->ms[i], w->ev[i], w->ev[i + 1]);
		(void)hipEventElapsedTime(&w->ms[6], w->ev[0], w->ev[6]);
		return BMH_OK;
	}
	HIPCK(hipEventRecord(w->ev[1], st));
	HIPCK(hipMemsetAsync(w->counter, 0, 8, st));
	HIPCK(hipMemsetAsync(w->n_cand + n_reads, 0, 4, st));
	{
		static const bool fwd_stats = getenv("BMH_SEED_STATS") != nullptr;
		unsigned long long *d_fst = fwd_stats ? (unsigned long long *)w->counter + 64 : nullptr;
		if (fwd_stats) HIPCK(hipMemsetAsync(d_fst, 0, 64 * 8, st));
		// experiment SEED_FWD_SORT (1: deal by the previous call's iteration counts -- the same batch again is then an ideally sorted deal)
		const uint32_t *d_deal = nullptr; uint32_t *d_iters = nullptr;
		if (bmh_tune("SEED_FWD_SORT", 0)) {
			if (!w->fwd_iters) HIPCK(hipMalloc((void **)&w->fwd_iters, 4 * ((size_t)w->max_reads + 1)));
			uint32_t *k0 = (uint32_t *)w->skeys, *k1 = (uint32_t *)w->skeys2;
			if (w->fwd_iters_n == n_reads) {
				iota_kernel<<<nblk(n_reads, 256), 256, 0, st>>>(w->svals, n_reads);
				HIPCK(hipMemcpyAsync(k0, w->fwd_iters, 4 * (size_t)n_reads, hipMemcpyDeviceToDevice, st));
				size_t tb = w->scan_tmp_bytes;
				HIPCK(rocprim::radix_sort_pairs_desc(w->scan_tmp, tb, k0, k1, w->svals, w->svals2, (size_t)n_reads, 0, 32, st));
				d_deal = w->svals2;
			}
			d_iters = w->fwd_iters; w->fwd_iters_n = n_reads;
		}
		smem_forward_kernel<<<nblk(n_reads, 256), 256, lds_pad, st>>>(f, rv, d_lens, min_seed_len, w->cand_a, w->cand_k, w->counter, w->max_cands, w->n_cand, d_fst, d_deal, d_iters);
		if (fwd_stats) {
			unsigned long long h[64];
			HIPCK(hipStreamSynchronize(st));
			HIPCK(hipMemcpy(h, d_fst, sizeof(h), hipMemcpyDeviceToHost));
			fprintf(stderr, "[forward] reads %u, waves %llu, wave-iterations %llu (%.1f per wave, longest %llu), lane-iterations %llu (%.1f per read): lane utilisation %.1f%%\n", n_reads, h[2], h[0],
			        (double)h[0] / (h[2] ? h[2] : 1), h[3], h[1], (double)h[1] / (n_reads ? n_reads : 1), 100.0 * h[1] / (64.0 * (h[0] ? h[0] : 1)));
			fprintf(stderr, "[forward] reads by iterations (0-7, 8-15, .. 248+):"); for (int q = 0; q < 32; ++q) fprintf(stderr, " %llu", h[8 + q]);
			fprintf(stderr, "\n[forward] waves by iterations (0-7, .. 184+):"); for (int q = 0; q < 24; ++q) fprintf(stderr, " %llu", h[40 + q]);
			fprintf(stderr, "\n");
		}
	}
	HIPCK(hipEventRecord(w->ev[2], st));
	{
		size_t tb = w->scan_tmp_bytes;
		HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, w->n_cand, w->cand_base, 0u, (size_t)n_reads + 1, rocprim::plus<uint32_t>(), st));
	}
	unsigned long long n_list = 0;     // slots handed out in the candidate list (chunks, incl. invalid fillers)
	uint32_t n_cands32 = 0;            // true number of candidates = size of the (read, ordinal)-sorted result array
	HIPCK(hipMemcpyAsync(&n_list, w->counter, 8, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(&n_cands32, w->cand_base + n_reads, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	const unsigned long long n_cands = n_cands32;
	out->n_cands = n_cands;
	if (n_list > w->max_cands) { bmh_set_error("bmh_seed_batch: %llu candidate slots > capacity %llu", n_list, (unsigned long long)w->max_cands); return BMH_ECAPACITY; }
	if (n_list)
		cand_scatter_kernel<<<nblk(n_list, 256), 256, 0, st>>>(w->cand_a, n_list, w->cand_base, w->svals);
	{
		static const bool want_stats = getenv("BMH_SEED_STATS") != nullptr;
		unsigned long long *d_st = want_stats ? (unsigned long long *)w->counter + 8 : nullptr;
		if (want_stats) HIPCK(hipMemsetAsync(d_st, 0, 64 + 40 * 8, st));
		if (n_cands) {
			// phases of the walk (BMH_SEED_BWD_PHASES="k1,k2,..": steps per phase, the last phase runs to the end).  Default: one launch --
			// measured on the bench workload, "6,12,24,48" takes the wave-iterations from 14.7 M to 9.2 M (lane utilisation 38.5 -> 61.7 %)
			// and the kernel from 6.14 to 6.69 ms: what it waits for is its gathers, not its instructions
			const std::vector<int> phases = [] {
				std::vector<int> v; const char *e = getenv("BMH_SEED_BWD_PHASES"); std::string t = e ? e : "";
				for (size_t p = 0; p < t.size();) { size_t q = t.find(',', p); if (q == std::string::npos) q = t.size(); const int k = atoi(t.substr(p, q - p).c_str()); if (k > 0) v.push_back(k); p = q + 1; }
				return v; }();
			const int np = (int)(phases.size() < 7 ? phases.size() : 7);
			const unsigned nb = nblk(n_cands, 256);
			const uint32_t sub_cap = ((nb + BWD_NSUB - 1) / BWD_NSUB) * 256u;        // lanes of the blocks that feed one list
			if (np && bwd_state_reserve(w, (uint64_t)sub_cap * BWD_NSUB) != BMH_OK) return BMH_ENOMEM;
			uint32_t *cnt = w->bwd_cnt;                                              // [np][BWD_NSUB] survivors of every phase
			if (np) HIPCK(hipMemsetAsync(cnt, 0, 4 * (size_t)np * BWD_NSUB, st));
			const int big = 0x7FFFFFFF;
			smem_backward_kernel<false><<<nb, 256, lds_pad, st>>>(f, rv, n_cands, min_seed_len, w->cand_a, w->cand_k, w->svals, w->res_a, w->res_k, d_st,
			                                                nullptr, nullptr, np ? w->bwd_state[0] : nullptr, np ? cnt : nullptr, sub_cap, np ? phases[0] : big);
			// (a resumed launch is sized by an upper bound of its lists -- every lane of the first launch -- and the blocks beyond a list leave at once)
			for (int ph = 1; ph <= np; ++ph)
				smem_backward_kernel<true><<<(sub_cap / 256u) * BWD_NSUB, 256, 0, st>>>(f, rv, n_cands, min_seed_len, w->cand_a, w->cand_k, w->svals, w->res_a, w->res_k, d_st,
				                                                                       w->bwd_state[(ph - 1) & 1], cnt + (size_t)(ph - 1) * BWD_NSUB, ph < np ? w->bwd_state[ph & 1] : nullptr,
				                                                                       ph < np ? cnt + (size_t)ph * BWD_NSUB : nullptr, sub_cap, ph < np ? phases[ph] : big);
		}
		if (want_stats) {
			unsigned long long h[48];
			HIPCK(hipStreamSynchronize(st));
			HIPCK(hipMemcpy(h, d_st, 64 + 40 * 8, hipMemcpyDeviceToHost));
			fprintf(stderr, "[backward] rank steps per candidate (0, 1, .. 38, 39+):");
			for (int q = 0; q < 40; ++q) fprintf(stderr, " %llu", h[8 + q]);
			fprintf(stderr, "\n");
			fprintf(stderr, "[backward] lane-steps on a one-row interval %llu (%.1f%%), candidates that start on one %llu (%.1f%%)\n", h[4], 100.0 * h[4] / (h[1] ? h[1] : 1), h[5], 100.0 * h[5] / (n_cands ? n_cands : 1));
			fprintf(stderr, "[backward] candidates of a pass that starts at read position 0 (no walk) %llu (%.1f%%); waves made of such only %llu\n", h[6], 100.0 * h[6] / (n_cands ? n_cands : 1), h[7]);
			fprintf(stderr, "[backward] candidates %llu, waves %llu (all phases), wave-iterations %llu (%.1f per wave, max %llu), lane-steps %llu (%.1f per candidate): lane utilisation %.1f%%\n",
			        (unsigned long long)n_cands, h[2], h[0], (double)h[0] / (h[2] ? h[2] : 1), h[3], h[1], (double)h[1] / (n_cands ? n_cands : 1), 100.0 * h[1] / (64.0 * (h[0] ? h[0] : 1)));
		}
	}
	HIPCK(hipEventRecord(w->ev[3], st));
	smem_filter_kernel<<<nblk(n_cands + 1, 256), 256, 0, st>>>(w->res_a, n_cands, w->occ);
	{
		size_t tb = w->scan_tmp_bytes;
		occ_keep_in fin; fin.occ = w->occ;
		HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), fin), w->occ_off, (uint64_t)0,
		                              (size_t)n_cands + 1, rocprim::plus<uint64_t>(), st));
	}
	uint64_t tot[2] = {0, 0};
	HIPCK(hipMemcpyAsync(&tot[0], w->occ_off + n_cands, 8, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	tot[1] = tot[0] >> OCC_OFF_SHIFT; tot[0] &= OCC_OFF_MASK;
	if (n_cands >> 28) { bmh_set_error("bmh_seed_batch: more than 2^28 candidates in one batch"); return BMH_ECAPACITY; }
	out->n_seeds = tot[0]; out->n_smems = tot[1];
	if (grow_occ(w, tot[0], out) != BMH_OK) return BMH_ECAPACITY;
	if (tot[0] >> 32) { bmh_set_error("bmh_seed_batch: more than 2^32 occurrences in one batch"); return BMH_ECAPACITY; }
	per_read_counts_kernel<<<nblk(n_reads, 256), 256, 0, st>>>(w->cand_base, w->occ_off, n_reads, w->n_ref_pos, w->prefix);
	HIPCK(hipEventRecord(w->ev[4], st));
	if (n_cands)
		expand_kernel<<<nblk(n_cands, 256), 256, 0, st>>>(w->res_a, w->res_k, w->occ, w->occ_off, n_cands, w->rows, w->qbeg, w->score);
	HIPCK(hipEventRecord(w->ev[5], st));
	if (tot[0])
		locate_kernel<<<nblk(nblk(tot[0], LOCATE_PER_WAVE) * 64ull, 256), 256, lds_pad, st>>>(f, w->rows, tot[0]);
	HIPCK(hipEventRecord(w->ev[6], st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	for (int i = 0; i < 6; ++i) (void)hipEventElapsedTime(&w->ms[i], w->ev[i], w->ev[i + 1]);
	(void)hipEventElapsedTime(&w->ms[6], w->ev[0], w->ev[6]);
	return BMH_OK;
}

// ---------------------------------------------------------------- calibration

// Random 32-byte block gathers over the index with a known count: the access pattern of the
// seeding kernels without their arithmetic.  Used (a) to calibrate rocprofv3's FETCH_SIZE for this
// pattern (MI355X_MICROARCH.md, HBM: the counter is only calibrated for wide streams) and (b) to
// measure the practical ceiling of random block gathers that the seeding kernels are held against.
// dependent != 0 chains each address on the previous block's contents, like a rank walk.
// dependent bits 8..: the gather's width in 32-byte blocks (0 or 1: one block; 2: an aligned 64-byte pair; 4: an aligned 128-byte line) --
// does the chip pay for random REQUESTS or for the bytes they move?  (what a wider rank block would cost); dependent bit 16: 16-byte requests, one load each
__global__ void __launch_bounds__(256) calib_gather_kernel(fmd_dev_t f, uint64_t n_blocks, int iters, int dependent, uint32_t *sink)
{
	uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
	uint32_t acc = 0;
	const int width = ((dependent >> 8) & 0xFF) > 1 ? ((dependent >> 8) & 0xFF) : 1;
	const bool half = (dependent >> 16) & 1;     // bit 16: a request is ONE 16-byte load (half a block): what a request costs per load instruction
	dependent &= 1;
	for (int i = 0; i < iters; ++i) {
		x = x * 6364136223846793005ull + 1442695040888963407ull;
		const uint64_t b0 = ((x >> 20) % n_blocks) & ~(uint64_t)(width - 1);
		if (half) {
			const uint4 h = f.blocks[2 * b0 + ((x >> 19) & 1)];
			acc += h.x ^ h.w;
			if (dependent) x ^= (uint64_t)(h.y + h.z) << 24;
			continue;
		}
		blk_t b = fmd_load_block(f, b0);
		acc += b.occ.x ^ (uint32_t)(b.hi >> 32);
		for (int w = 1; w < width; ++w) { const blk_t c = fmd_load_block(f, b0 + (uint64_t)w); acc += c.occ.x ^ (uint32_t)(c.hi >> 32); }
		if (dependent) x ^= (uint64_t)(b.occ.y + (uint32_t)b.lo) << 24;
	}
	if (acc == 0x12345678u) sink[0] = acc;    // keeps the loads alive
}

extern "C" int bmh_calib_gather(const bmh_index_t *idx, uint64_t n_lanes, int iters, int dependent, void *stream_, float *ms)
{
	if (!idx || !ms || iters < 1) { bmh_set_error("bmh_calib_gather: bad argument"); return BMH_EINVAL; }
	hipStream_t st = (hipStream_t)stream_;
	// no allocation / free in here: hipFree is a device-wide synchronisation point
	static thread_local uint32_t *sink = nullptr;
	static thread_local hipEvent_t e0 = nullptr, e1 = nullptr;
	if (!sink) { HIPCK(hipMalloc((void **)&sink, 64)); HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1)); }
	const uint64_t n_blocks = ((idx->dev.seq_len + 63) / 64) & ~3ull;
	HIPCK(hipEventRecord(e0, st));
	calib_gather_kernel<<<nblk(n_lanes, 256), 256, 0, st>>>(idx->dev, n_blocks, iters, dependent, sink);
	HIPCK(hipEventRecord(e1, st));
	HIPCK(hipEventSynchronize(e1));
	HIPCK(hipEventElapsedTime(ms, e0, e1));
	return BMH_OK;
}

// wave residency trace (wtrace.h): this translation unit's copy of the trace symbols
WTRACE_DEFINE_SETTER(bmh_wtrace_set_seed)
