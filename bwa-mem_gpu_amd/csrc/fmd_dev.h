// Device-side FMD-index arithmetic for gfx950 (wave64).
//
// Index FILES and the arrays that cross the C ABI keep the reference's GPU layout
// (/root/reference/src/GPUSeed/seed_gen.cu:28-48): one 32-byte block per 64 BWT
// symbols = u32 occ[4] (counts of A,C,G,T before the block) + u32 bwt[4] (16 symbols
// per word, 2 bits, MSB first).  In HBM the blocks are re-encoded once, when an index
// handle is made (fmd_native_blocks_kernel, c_api.hip), into the MI355X-native form of
// the same 32 bytes:
//     u32 occ[4];  u64 lo;  u64 hi;          bit t of lo / hi = low / high bit of symbol t
// so that a rank inside a block is  occ[c] + popcount((hi ^ ~Hc) & (lo ^ ~Lc) & mask64(off))
// -- one 64-bit mask and ~12 vector instructions instead of four 16-symbol words with a
// mask, a shift, an xor and two ands each (~48).  Same sector count (a block is still one
// 32-byte-aligned unit = one 64-byte HBM sector), same values.
//
// Rank semantics follow the CPU statement of the reference
// (src/bwt.c:235-261 bwt_occ, :363-405 bwt_2occ4, :64-70 bwt_invPsi), not the
// off-by-one GPU form of seed_gen.cu:403-405.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct fmd_dev_t {
	uint64_t primary;
	uint64_t L2[5];
	uint64_t seq_len;
	const uint4 *blocks;      // NATIVE blocks, 2 x uint4 each: [2b] = occ, [2b+1] = {lo[31:0], lo[63:32], hi[31:0], hi[63:32]}
	const uint32_t *sa;       // sa[0] = 0xFFFFFFFF
	const uint32_t *sa_bits;  // 33rd bit of each sample
	uint64_t n_sa;
	int sa_shift;             // log2(sa_intv)
	// optional 2-bit forward-strand text (pac, 4 bases per byte, MSB first) for the
	// unique-interval shortcut; 0 when absent
	const uint8_t *pac;
	uint64_t l_pac;
	int wave_prio;            // measurement knob SEED_SETPRIO (0 = off): s_setprio level of the seeding kernels' waves
};

// the wave's issue priority (the SIMD arbitrates by priority, then age): the gather-bound kernels run short bursts between loads
__device__ __forceinline__ void fmd_wave_prio(const int p)
{
	if (p == 1) __builtin_amdgcn_s_setprio(1);
	else if (p == 2) __builtin_amdgcn_s_setprio(2);
	else if (p >= 3) __builtin_amdgcn_s_setprio(3);
}

struct blk_t { uint4 occ; uint64_t lo, hi; };

// v[c] for a per-lane c in 0..3 as two levels of conditional moves on the bits of c (a chain of ?: on c == 0, c == 1, ...
// is compiled into divergent branches around single moves)
__device__ __forceinline__ uint32_t sel4(int c, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3)
{
	const bool b0 = c & 1, b1 = c & 2;
	const uint32_t lo = b0 ? v1 : v0, hi = b0 ? v3 : v2;
	return b1 ? hi : lo;
}
__device__ __forceinline__ uint64_t sel4(int c, uint64_t v0, uint64_t v1, uint64_t v2, uint64_t v3)
{
	const bool b0 = c & 1, b1 = c & 2;
	const uint64_t lo = b0 ? v1 : v0, hi = b0 ? v3 : v2;
	return b1 ? hi : lo;
}

// L2[c] for a per-lane c without indexing the kernel-argument struct dynamically
__device__ __forceinline__ uint64_t fmd_L2(const fmd_dev_t &f, int c)
{
	const uint64_t r = sel4(c, f.L2[0], f.L2[1], f.L2[2], f.L2[3]);
	return c > 3 ? f.L2[4] : r;
}

__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

__device__ __forceinline__ blk_t fmd_load_block(const fmd_dev_t &f, uint64_t b)
{
	blk_t r;
	r.occ = f.blocks[2 * b];
	const uint4 p = f.blocks[2 * b + 1];
	r.lo = u64_of(p.x, p.y); r.hi = u64_of(p.z, p.w);
	return r;
}

// reference word (16 symbols, symbol t at bits 31-2t : 30-2t) -> its 16 low bits and 16 high bits, symbol t at bit t
__host__ __device__ __forceinline__ uint32_t fmd_even_bits16(uint32_t x)
{
	x &= 0x55555555u;
	x = (x | (x >> 1)) & 0x33333333u;
	x = (x | (x >> 2)) & 0x0F0F0F0Fu;
	x = (x | (x >> 4)) & 0x00FF00FFu;
	x = (x | (x >> 8)) & 0x0000FFFFu;
	return x;
}
__host__ __device__ __forceinline__ uint32_t fmd_brev32(uint32_t x)
{
	x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
	x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
	x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
	x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
	return (x >> 16) | (x << 16);
}
// the four reference words of a block -> the two bit planes
__host__ __device__ __forceinline__ void fmd_words_to_planes(const uint32_t w[4], uint64_t &lo, uint64_t &hi)
{
	lo = 0; hi = 0;
	for (int i = 0; i < 4; ++i) {
		const uint32_t r = fmd_brev32(w[i]);              // symbol t: high bit at 2t, low bit at 2t+1
		hi |= (uint64_t)fmd_even_bits16(r) << (16 * i);
		lo |= (uint64_t)fmd_even_bits16(r >> 1) << (16 * i);
	}
}

// bits 0..off set, off in 0..63 (M1: off may also be -1 -> no bit)
template <bool M1 = true>
__device__ __forceinline__ uint64_t mask_incl(int off)
{
	const uint64_t m = ~0ull >> (63 - off);
	return (M1 && off < 0) ? 0ull : m;
}

// counts of A,C,G,T among symbols [0..off] (inclusive) of the block, plus the block's occ
template <bool M1 = true>
__device__ __forceinline__ void blk_occ4(const blk_t &b, int off, uint32_t cnt[4])
{
	const uint64_t m = mask_incl<M1>(off);
	const uint64_t h = b.hi & m, l = b.lo & m, both = h & l;
	const uint32_t c3 = (uint32_t)__popcll(both), c2 = (uint32_t)__popcll(h ^ both), c1 = (uint32_t)__popcll(l ^ both);
	cnt[0] = b.occ.x + (uint32_t)(off + 1) - c1 - c2 - c3;
	cnt[1] = b.occ.y + c1;
	cnt[2] = b.occ.z + c2;
	cnt[3] = b.occ.w + c3;
}

// count of symbol c among symbols [0..off] of the planes, plus occ_c
template <bool M1 = true>
__device__ __forceinline__ uint32_t planes_occ1(uint64_t lo, uint64_t hi, uint32_t occ_c, int off, int c)
{
	const uint64_t xl = (c & 1) ? 0ull : ~0ull, xh = (c & 2) ? 0ull : ~0ull;
	return occ_c + (uint32_t)__popcll((hi ^ xh) & (lo ^ xl) & mask_incl<M1>(off));
}

// count of symbol c among symbols [0..off] of the block, plus the block's occ[c]
__device__ __forceinline__ uint32_t blk_occ1(const blk_t &b, int off, int c)
{
	return planes_occ1(b.lo, b.hi, sel4(c, b.occ.x, b.occ.y, b.occ.z, b.occ.w), off, c);
}

__device__ __forceinline__ int blk_sym(const blk_t &b, int off)
{
	return (int)(((b.hi >> off) & 1) << 1 | ((b.lo >> off) & 1));
}

// Occ for all four symbols at full-matrix row k (src/bwt.c:235-261 semantics)
__device__ __forceinline__ void fmd_occ4(const fmd_dev_t &f, uint64_t k, uint64_t cnt[4])
{
	if (k == f.seq_len) {
#pragma unroll
		for (int c = 0; c < 4; ++c) cnt[c] = f.L2[c + 1] - f.L2[c];
		return;
	}
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= f.primary);
	blk_t b = fmd_load_block(f, k >> 6);
	uint32_t c4[4];
	blk_occ4(b, (int)(k & 63), c4);
#pragma unroll
	for (int c = 0; c < 4; ++c) cnt[c] = c4[c];
}

// Occ(k,.) and Occ(l,.) for k <= l, l a valid row, k possibly (uint64_t)-1 (all counts 0; M1 = false: the caller knows k is a
// row -- the searches' k is `first row of a non-empty interval - 1` >= 0).  Written without divergent paths: row seq_len needs
// no special case (after the primary adjustment it is the last BWT symbol, whose inclusive count is the total), row -1 becomes
// "nothing of block 0" (whose occ is 0), and the second block is only fetched when it differs from the first -- the arithmetic
// is the same two block counts for every lane.
template <bool M1 = true>
__device__ __forceinline__ void fmd_occ4_pair(const fmd_dev_t &f, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	const bool km1 = M1 && k == (uint64_t)-1;
	const uint64_t k2 = km1 ? 0 : k - (k >= f.primary), l2 = l - (l >= f.primary);
	const int koff = km1 ? -1 : (int)(k2 & 63), loff = (int)(l2 & 63);
	const uint64_t kb = k2 >> 6, lb = l2 >> 6;
	blk_t A = fmd_load_block(f, kb), B = A;
	if (lb != kb) B = fmd_load_block(f, lb);
	uint32_t a4[4], b4[4];
	blk_occ4<M1>(A, koff, a4);
	blk_occ4<false>(B, loff, b4);
#pragma unroll
	for (int c = 0; c < 4; ++c) { ck[c] = a4[c]; cl[c] = b4[c]; }
}

__device__ __forceinline__ uint64_t fmd_occ1(const fmd_dev_t &f, uint64_t k, int c)
{
	if (k == f.seq_len) return fmd_L2(f, c + 1) - fmd_L2(f, c);
	if (k == (uint64_t)-1) return 0;
	k -= (k >= f.primary);
	blk_t b = fmd_load_block(f, k >> 6);
	return blk_occ1(b, (int)(k & 63), c);
}

// Occ(k, c) and Occ(l, c) (as fmd_occ4_pair, one symbol): of every block only the counter of c (one dword at its place) and the
// two planes are fetched -- no selection among four counters afterwards
template <bool M1 = true>
__device__ __forceinline__ void fmd_occ1_pair(const fmd_dev_t &f, uint64_t k, uint64_t l, int c, uint64_t &ok, uint64_t &ol)
{
	const bool km1 = M1 && k == (uint64_t)-1;
	const uint64_t k2 = km1 ? 0 : k - (k >= f.primary), l2 = l - (l >= f.primary);
	const int koff = km1 ? -1 : (int)(k2 & 63), loff = (int)(l2 & 63);
	const uint64_t kb = k2 >> 6, lb = l2 >> 6;
	const uint32_t *base = (const uint32_t *)f.blocks;
	uint32_t oa = base[kb * 8 + (uint32_t)c];
	uint4 pa = f.blocks[2 * kb + 1];
	uint32_t ob = oa; uint4 pb = pa;
	if (lb != kb) {
		ob = base[lb * 8 + (uint32_t)c]; pb = f.blocks[2 * lb + 1];
		asm volatile("" : "+v"(pb.x), "+v"(pb.y), "+v"(pb.z), "+v"(pb.w));     // one 16-byte load (left alone the compiler hoists half of it out of the branch as dword loads)
	}
	const uint64_t xl = (c & 1) ? 0ull : ~0ull, xh = (c & 2) ? 0ull : ~0ull;
	const uint64_t ma = (u64_of(pa.z, pa.w) ^ xh) & (u64_of(pa.x, pa.y) ^ xl) & mask_incl<M1>(koff);
	const uint64_t mb = (u64_of(pb.z, pb.w) ^ xh) & (u64_of(pb.x, pb.y) ^ xl) & mask_incl<false>(loff);
	ok = oa + (uint32_t)__popcll(ma);
	ol = ob + (uint32_t)__popcll(mb);
}

// forward extension of the bi-interval (k, l, s) by every symbol (bwt_extend on the swapped interval, src/bwt.c:428-448):
// the interval of the pattern followed by read symbol b is (nk[3-b], nl[3-b], ns[3-b])
__device__ __forceinline__ void fmd_forward_ext(const fmd_dev_t &f, uint64_t k, uint64_t l, uint64_t s, uint64_t nk[4], uint64_t nl[4], uint64_t ns[4])
{
	uint64_t tk[4], tl[4];
	fmd_occ4_pair<false>(f, l - 1, l - 1 + s, tk, tl);      // l >= 1: the first row of an interval of a non-empty pattern
#pragma unroll
	for (int q = 0; q < 4; ++q) { ns[q] = tl[q] - tk[q]; nl[q] = fmd_L2(f, q) + 1 + tk[q]; }
	nk[3] = k + ((l <= f.primary) & (l + s - 1 >= f.primary));
	nk[2] = nk[3] + ns[3]; nk[1] = nk[2] + ns[2]; nk[0] = nk[1] + ns[1];
}

// LF step, CPU form (src/bwt.c:64-70)
__device__ __forceinline__ uint64_t fmd_inv_psi(const fmd_dev_t &f, uint64_t k)
{
	if (k == f.primary) return 0;
	uint64_t x = k - (k > f.primary);
	blk_t b = fmd_load_block(f, x >> 6);
	int off = (int)(x & 63);
	int c = blk_sym(b, off);
	return fmd_L2(f, c) + blk_occ1(b, off, c);
}

// SA value of row k (src/bwt.c:105-115; sample = 32 bits + packed 33rd bit, seed_gen.cu:653-657)
__device__ __forceinline__ uint64_t fmd_sa(const fmd_dev_t &f, uint64_t k)
{
	uint64_t steps = 0, mask = (1ull << f.sa_shift) - 1;
	while (k & mask) { ++steps; k = fmd_inv_psi(f, k); }
	uint64_t idx = k >> f.sa_shift;
	if (idx == 0) return steps - 1;            // sa[0] == (bwtint_t)-1 on the CPU path
	uint64_t hi = (f.sa_bits[idx >> 5] >> (idx & 31)) & 1u;
	return ((uint64_t)f.sa[idx] | (hi << 32)) + steps;
}

// 16 symbols T[t0 .. t0+15] of the indexed text, packed like the reads (symbol j at bits 2j+1:2j).  Two loads when the
// window lies inside one strand (the reverse strand is the complement of the forward bytes read in their own order:
// pac is MSB-first, so the mirrored order is already there); symbols at or beyond seq_len read as 0.
__device__ __forceinline__ int fmd_text(const fmd_dev_t &f, uint64_t i);
__device__ __forceinline__ uint32_t fmd_text16(const fmd_dev_t &f, uint64_t t0)
{
	const uint64_t L = f.l_pac;
	const bool fwd = t0 + 16 <= L, rev = t0 >= L && t0 + 16 <= 2 * L;
	if (fwd || rev) {
		const uint64_t a = fwd ? t0 : 2 * L - 16 - t0;              // forward-strand symbols a .. a+15
		// two aligned words cover the 5 bytes that hold the 16 symbols (the allocation is padded: bmh_index_upload / callers)
		const uint64_t byte0 = a >> 2;
		const uint32_t *wp = (const uint32_t *)(f.pac + (byte0 & ~3ull));
		const uint64_t two = ((uint64_t)__builtin_bswap32(wp[0]) << 32) | (uint64_t)__builtin_bswap32(wp[1]);    // bytes in text order, first on top
		const uint32_t v = (uint32_t)(two >> (32 - 8 * (int)(byte0 & 3) - 2 * (int)(a & 3)));   // symbol a in the top two bits
		if (rev) return ~v;                                           // T[t0+j] = 3 - fwd[a+15-j] sits at bits 2j+1:2j of v already
		const uint32_t r = __brev(v);
		return ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
	}
	uint32_t w = 0;
	for (int j = 0; j < 16; ++j) if (t0 + (uint64_t)j < f.seq_len) w |= (uint32_t)fmd_text(f, t0 + (uint64_t)j) << (2 * j);
	return w;
}

// symbol i of the indexed text T = fwd . revcomp(fwd) from the 2-bit pac
__device__ __forceinline__ int fmd_text(const fmd_dev_t &f, uint64_t i)
{
	bool rev = i >= f.l_pac;
	uint64_t p = rev ? 2 * f.l_pac - 1 - i : i;
	int c = (f.pac[p >> 2] >> ((~p & 3) << 1)) & 3;
	return rev ? 3 - c : c;
}
