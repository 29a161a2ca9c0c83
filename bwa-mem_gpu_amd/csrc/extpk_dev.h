// Packed 16-bit rows of the seed-extension DP (ksw_extend2, /root/reference/src/ksw.c:864-986) -- included by extend_kernels.hip.
//
// Same contract and the same row-at-a-time mapping as extend16_kernel, rebuilt for instruction count (the stage runs at the
// VALU issue ceiling, so its time IS its instruction count):
//   * two query columns per VGPR: H, E, M, F live as unsigned 16-bit pairs and move through v_pk_{mad,sub(clamp),max,min}_u16.
//     The reference clamps M-oe, E and F at zero, which is what an unsigned saturating subtract does for free.
//     Lane l of a group owns the 2P columns [l*2P, l*2P+2P): low halves = its first P columns, high halves = the next P, so the
//     diagonal neighbour of pair p is simply pair p-1 of the previous row (pair 0: one DPP shift + v_alignbit).
//   * G = 4 lanes per alignment up to 128 query columns (8 up to 256, 16 up to 288): SIXTEEN alignments per wave share the per-row
//     control code (scan, reductions, end / maximum bookkeeping), which is what a row costs besides its cells.
//   * substitution scores by ONE v_perm_b32 per pair from an 8-byte row table {score(t, code) + b}: M = (hd != 0) * scb + hd -sat b.
//   * no `beg` bookkeeping: the first-column value max(0, h0 - o_del - e_del*(i+1)) applies in every row -- beg > 0 implies it has
//     reached zero for good, and cells left of beg have zero inputs (scripts/extpk_model.py checks this algebra against the oracle).
//   * only H is masked at and right of `end` (M and E are zero there by themselves); the masks come from a small LDS table.
//   * F: per-chain local recurrence, 32-bit max-plus scan over the group's lanes (DPP), exact inflow for both chains of a lane.
//   * row maximum and its last column from max over (h << 4 | pair) keys; new end = last non-zero H column + 3.
//   * H(i, qlen-1) for gscore: the lanes park their H pairs in LDS, the group reads the one halfword it needs (VALU-free).
// Eligibility (ext_route): 1 <= b, a + b <= 255, h0 + qlen*a < 4096, qlen <= 288, tlen <= PK_TCAP(G); everything else takes the
// 32-bit kernels.  Results are bit-identical to those (tests/test_gpu_parity.py runs both).
#pragma once

// target bases of an alignment staged in LDS, by group width (the 4-lane groups are the many small queries: 64 of them per block)
#define PK_TCAP(G) ((G) == 4 ? 384 : (G) == 8 ? 512 : 640)
#ifndef PK_WAVES4_MAXP
#define PK_WAVES4_MAXP 10     // classes of up to this many pairs per lane run four waves per SIMD (registers and grid; 120 VGPRs at 10 pairs), the larger ones three (8 -> 10: -0.5 % at 150 bp, -0.7 % at 300 bp)
#endif
#ifndef PK_BOUND_MASK
#define PK_BOUND_MASK 15      // the exact early-stop bound is evaluated every (mask + 1)-th row of a wave (every 2nd / 4th / 8th / 16th / 32nd: 17.6 / 17.3 / 17.2 / 17.1 / 17.4 ms)
#endif
#define PK_HMAX 4096          // scores stay below this (keys are h << 4 | pair in 16 bits)
#define PK_HMAX17 2048        // ... in the 17-pair class (keys h << 5 | pair)
#define PK_NEG (-(1 << 28))

// The pair steps are spelled out as asm blocks: left to itself the compiler turns min(x, 1) and the 0/1 multiply into per-half
// compares and selects (five instructions instead of one), and reorders the in-place updates so that every loop-carried pair needs
// a copy at the end of the row.  Plain `asm` (not volatile): pure functions of their operands.  Wave-uniform operands sit in SGPRs
// (a VOP3P instruction takes one); small constants are inline operands applied to both halves (op_sel_hi 0).
__device__ __forceinline__ uint32_t pk_min1(uint32_t a) { uint32_t d; asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t pk_maxs(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_subiK(uint32_t a, uint32_t k) { uint32_t d; asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k)); return d; }
__device__ __forceinline__ uint32_t pk_addi(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_madK(uint32_t a, uint32_t k, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c)); return d; }
__device__ __forceinline__ uint32_t pk_splat(int v) { return ((uint32_t)v & 0xFFFFu) * 0x10001u; }

__device__ __forceinline__ uint32_t pk_subsK(uint32_t a, uint32_t k) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "s"(k)); return d; }
__device__ __forceinline__ unsigned long long pk_readlane64(unsigned long long v, int lane)
{
	return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane) << 32);
}
// byte u of the result = bits 2u+1:2u of z (four 2-bit symbols spread into four bytes)
__device__ __forceinline__ uint32_t pk_spread4(uint32_t z)
{
	const uint32_t a = (z | (z << 12)) & 0x000F000Fu;
	return (a | (a << 6)) & 0x03030303u;
}
// Target codes of rows k0..k0+7 of one job as two dwords (byte u of `lo` = row k0+u; rows at and beyond tlen: don't care).
// Descriptor jobs decode them from three bytes of the 2-bit text at once (the rows of a job walk ONE strand up or down:
// chain2aln clips a window that would cross the strand boundary, src/bwamem.c:1261-1264); array jobs read their bytes.
__device__ __forceinline__ void pk_t8(const ext_args_t &A, const job_src_t &s, const int k0, const int tlen, uint32_t &lo, uint32_t &hi)
{
	if (!A.desc) {
		lo = hi = 0;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int k = k0 + u;
			const uint32_t b = k < tlen ? (uint32_t)s.tp[k] : 5u, c = b > 3u ? 5u : b;
			if (u < 4) lo |= c << (8 * u); else hi |= c << (8 * (u - 4));
		}
		return;
	}
	const long long p0 = s.t0 + (long long)k0 * s.tdir;
	const bool rev = p0 >= A.l_pac;
	const long long f0 = rev ? (A.l_pac << 1) - 1 - p0 : p0;          // forward-strand position of row k0
	const bool asc = rev ? s.tdir < 0 : s.tdir > 0;                  // the rows walk the forward strand upwards
	long long fa = asc ? f0 : f0 - 7;                                // lowest position of the eight
	int D = 7;                                                       // descending: row u is symbol D - u of the window at fa
	if (fa < 0) { D = (int)f0; fa = 0; }
	// three bytes of the text as ONE unaligned dword load (the 64 lanes of a staging step read 64 different lines: what such a step
	// costs is its number of load instructions; pac is readable 9 bytes past its end)
	uint32_t raw; __builtin_memcpy(&raw, A.pac + (fa >> 2), 4);
	const uint32_t v = __builtin_bswap32(raw) >> 8;
	const uint32_t y = (v << (8 + 2 * ((int)fa & 3))) >> 16;          // eight symbols, position fa+j at bits 15-2j:14-2j
	const uint32_t z = asc ? y : y >> (2 * (7 - D));
	const uint32_t w0 = pk_spread4(z & 0xFFu), w1 = pk_spread4((z >> 8) & 0xFFu);
	lo = asc ? __builtin_bswap32(w1) : w0;
	hi = asc ? __builtin_bswap32(w0) : w1;
	if (rev) { lo ^= 0x03030303u; hi ^= 0x03030303u; }
}

// first pass of a pair: M = hd ? max(hd + score, 0) : 0 and the chain's local F, i.e. what the chain's own cells send out of its last
// column if nothing flows in: max(0, max_k (M_k - oe - e (P-1-k))) -- the recurrence F = max(F - e, max(M - oe, 0)) unrolled; kept as
// max_k (M_k + kf_k) with kf_k = 0x4000 - oe - e (P-1-k), ONE scalar register stepped by e from pair to pair inside the block (sixteen
// constants of their own do not fit the scalar registers: the compiler parks them in a vector register and reads them back lane by
// lane, one v_readlane per pair), the 0x4000 taken off with a saturating subtraction after the last pair.  5 vector instructions
// (6 as a recurrence) + 1 scalar
__device__ __forceinline__ void pk_pair1(uint32_t &M, uint32_t &agg, uint32_t &kf, const uint32_t hd, const uint32_t mask, const uint32_t sel,
                                         const uint32_t tbl_hi, const uint32_t tbl_lo, const uint32_t b2, const uint32_t ei2)
{
	uint32_t t;
	asm("v_perm_b32 %[t], %[thi], %[tlo], %[sel]\n\t"
	    "v_pk_mad_u16 %[M], %[mask], %[t], %[hd]\n\t"
	    "v_pk_sub_u16 %[M], %[M], %[b2] clamp\n\t"
	    "v_pk_add_u16 %[t], %[M], %[kf]\n\t"
	    "s_add_u32 %[kf], %[kf], %[ei]\n\t"
	    "v_pk_max_u16 %[agg], %[agg], %[t]"
	    : [M] "=&v"(M), [agg] "+v"(agg), [t] "=&v"(t), [kf] "+s"(kf)
	    : [hd] "v"(hd), [mask] "v"(mask), [sel] "v"(sel), [thi] "s"(tbl_hi), [tlo] "v"(tbl_lo), [b2] "s"(b2), [ei] "s"(ei2)
	    : "scc");
}
// second pass of a pair: H = max(M, E, F) masked at `end`, E and F for the next cells, non-zero bits, row-maximum key (12 / 13 instructions).
// H and NZ are declared read-write although their old values are dead: that ties the new values to the same registers, so the
// loop-carried pairs are updated in place instead of being copied back at the end of every row.
template <int SLOT, bool SAME_OE, int KMUL>
__device__ __forceinline__ void pk_pair2(uint32_t &H, uint32_t &E, uint32_t &NZ, uint32_t &f, uint32_t &key, uint32_t &nzb, const uint32_t M, const uint32_t em,
                                         const uint32_t ei2, const uint32_t ed2, const uint32_t oei2, const uint32_t oed2)
{
	uint32_t t, u;
	if (SAME_OE)
		asm("v_pk_max_u16 %[H], %[M], %[E]\n\t"
		    "v_pk_sub_u16 %[t], %[M], %[oei] clamp\n\t"
		    "v_pk_max_u16 %[H], %[H], %[f]\n\t"
		    "v_pk_sub_u16 %[E], %[E], %[ed] clamp\n\t"
		    "v_pk_sub_u16 %[f], %[f], %[ei] clamp\n\t"
		    "v_and_b32 %[H], %[H], %[em]\n\t"
		    "v_pk_max_u16 %[E], %[E], %[t]\n\t"
		    "v_pk_max_u16 %[f], %[f], %[t]\n\t"
		    "v_pk_min_u16 %[NZ], %[H], 1 op_sel_hi:[1,0]\n\t"
		    "v_pk_mad_u16 %[u], %[H], %[kmul], %[slot] op_sel_hi:[1,0,0]\n\t"
		    "v_lshl_or_b32 %[nzb], %[nzb], 1, %[NZ]\n\t"
		    "v_pk_max_u16 %[key], %[key], %[u]"
		    : [H] "+v"(H), [E] "+v"(E), [NZ] "+v"(NZ), [f] "+v"(f), [key] "+v"(key), [nzb] "+v"(nzb), [t] "=&v"(t), [u] "=&v"(u)
		    : [M] "v"(M), [em] "v"(em), [ei] "s"(ei2), [ed] "s"(ed2), [oei] "s"(oei2), [slot] "n"(SLOT), [kmul] "n"(KMUL));
	else
		asm("v_pk_max_u16 %[H], %[M], %[E]\n\t"
		    "v_pk_sub_u16 %[t], %[M], %[oed] clamp\n\t"
		    "v_pk_max_u16 %[H], %[H], %[f]\n\t"
		    "v_pk_sub_u16 %[E], %[E], %[ed] clamp\n\t"
		    "v_pk_sub_u16 %[f], %[f], %[ei] clamp\n\t"
		    "v_pk_max_u16 %[E], %[E], %[t]\n\t"
		    "v_pk_sub_u16 %[t], %[M], %[oei] clamp\n\t"
		    "v_and_b32 %[H], %[H], %[em]\n\t"
		    "v_pk_max_u16 %[f], %[f], %[t]\n\t"
		    "v_pk_min_u16 %[NZ], %[H], 1 op_sel_hi:[1,0]\n\t"
		    "v_pk_mad_u16 %[u], %[H], %[kmul], %[slot] op_sel_hi:[1,0,0]\n\t"
		    "v_lshl_or_b32 %[nzb], %[nzb], 1, %[NZ]\n\t"
		    "v_pk_max_u16 %[key], %[key], %[u]"
		    : [H] "+v"(H), [E] "+v"(E), [NZ] "+v"(NZ), [f] "+v"(f), [key] "+v"(key), [nzb] "+v"(nzb), [t] "=&v"(t), [u] "=&v"(u)
		    : [M] "v"(M), [em] "v"(em), [ei] "s"(ei2), [ed] "s"(ed2), [oei] "s"(oei2), [oed] "s"(oed2), [slot] "n"(SLOT), [kmul] "n"(KMUL));
}

// ---- group primitives: G = 16 is one DPP row, G = 8 half of one, G = 4 a quad
template <int G> __device__ __forceinline__ int grp_shr1(int v, int fill, bool g0)      // lane-1 of the group; its lane 0 receives `fill`
{
	int t = __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false);
	if (G < 16) t = g0 ? fill : t;
	return t;
}
template <int G> __device__ __forceinline__ int grp_scan_max(int v)                     // inclusive max-scan over the group
{
	if (G == 16) return row_scan_max_f(v);
	if (G == 4) {
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
		             : "+v"(v));
		return v;
	}
	int t;
	// inside the quads by quad_perm (a lane without a left neighbour reads itself), then the first quad's total into the second
	// one of each half row: row_shr:4 written only in banks 1 and 3, so nothing crosses from one group into the next
	asm volatile("s_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %0, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_mov_b32_dpp %1, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
	             "v_max_i32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n\ts_nop 1"
	             : "+v"(v), "=&v"(t));
	return v;
}
template <int G> __device__ __forceinline__ void grp_allmax2(int &x, int &y)           // two all-reduce max butterflies, interleaved
{
	if (G == 16)
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
		             "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
		             "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
		             : "+v"(x), "+v"(y));
	else if (G == 8)
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
		             "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
		             : "+v"(x), "+v"(y));
	else
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
		             : "+v"(x), "+v"(y));
}
template <int G> __device__ __forceinline__ int grp_allmax(int v)
{
	if (G == 16) return row_allmax_f(v);
	if (G == 8)
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
		             : "+v"(v));
	else
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
		             : "+v"(v));
	return v;
}

// scoring constants, wave-uniform (SGPRs)
template <int P> struct pk_consts_t {
	uint32_t kf0;                          // 0x4000 - oe_ins - e_ins (P-1), splatted: pk_pair1's F constant of pair 0
	uint32_t b2, oei2, oed2, ei2, ed2;     // splatted
	uint32_t ab, nrow, tbl_hi;             // a+b; (b-1) in all four bytes (rows whose target base is N); table bytes 4..7: code 4 (N) = b-1, pad = 0
	int eP, eC, a;                         // e_ins * P, e_ins * 2P
	uint32_t a2;                           // a splatted
	bool same_oe;
};
template <int P> __device__ __forceinline__ pk_consts_t<P> pk_consts(const ext_args_t &A)
{
	pk_consts_t<P> K;
	K.kf0 = pk_splat(0x4000 - (A.o_ins + A.e_ins) - A.e_ins * (P - 1));
	K.b2 = pk_splat(A.b); K.oei2 = pk_splat(A.o_ins + A.e_ins); K.oed2 = pk_splat(A.o_del + A.e_del);
	K.ei2 = pk_splat(A.e_ins); K.ed2 = pk_splat(A.e_del);
	K.ab = (uint32_t)(A.a + A.b); K.nrow = (uint32_t)(A.b - 1) * 0x01010101u; K.tbl_hi = (uint32_t)(A.b - 1);
	K.eP = A.e_ins * P; K.eC = A.e_ins * 2 * P; K.a = A.a; K.a2 = pk_splat(A.a);
	K.same_oe = A.o_ins + A.e_ins == A.o_del + A.e_del;
	return K;
}

// per-alignment control state, replicated over the lanes of the group
struct pk_rs_t { int end, mx, max_i, max_j, max_ie, gscore, max_off; };

template <int P, bool SAME_OE, int PP, int... Is>
__device__ __forceinline__ void pk_pass2(uint32_t (&H)[P], uint32_t (&E)[P], uint32_t (&NZ)[P], const uint32_t (&M)[P], const uint32_t (&em)[PP],
                                         uint32_t &f, uint32_t &key, uint32_t &nzb, uint32_t &nzb2, const pk_consts_t<P> &K, std::integer_sequence<int, Is...>)
{
	// (keys are h << 4 | pair up to 16 pairs, h << 5 | pair beyond; the non-zero bits of pairs 16.. go to a second register)
	(pk_pair2<Is, SAME_OE, (P > 16 ? 32 : 16)>(H[Is], E[Is], NZ[Is], f, key, Is < 16 ? nzb : nzb2, M[Is], em[Is], K.ei2, K.ed2, K.oei2, K.oed2), ...);
}

// One DP row of the wave's alignments.  em_tab: the end masks, [2P+1][PP] dwords in LDS; hrow: the group's PP dwords of the H
// parking area, written by the lane that owns column qlen-1 (`owner`); h16[goff]: where the group finds H(i, qlen-1) in it.
// Returns the new `alive`.
template <int G, int P, bool SAME_OE>
__device__ __forceinline__ bool pk_row(const pk_consts_t<P> &K, const int zdrop, uint32_t (&H)[P], uint32_t (&E)[P], uint32_t (&NZ)[P], const uint32_t (&sel)[P],
                                       const int ti, const bool run, const int hfc, const int hnx, const int i, const int qlen,
                                       const int j0, const int eCl, const bool g0, const uint32_t phi0,
                                       const uint32_t *em_tab, uint32_t *hrow, const bool owner, const uint16_t *h16, const int goff,
                                       pk_rs_t &S, bool alive, const bool bound, const int rl, const bool o3only, const int end_bonus)
{
	constexpr int C = 2 * P, PP = (P + 3) & ~3, PS = PP + 4;      // PS: LDS row stride in dwords (odd multiple of 4: rows start in different banks)
	// end masks of this lane: cells [0, wend) of the lane are left of `end`
	const int wend = run ? S.end - j0 : 0;
	const int wc = min(max(wend, 0), C);
	uint32_t em[PP];
	{
		const uint4 *src = (const uint4 *)(em_tab + wc * PS);
#pragma unroll
		for (int k = 0; k < PP / 4; ++k) { const uint4 v = src[k]; em[4 * k] = v.x; em[4 * k + 1] = v.y; em[4 * k + 2] = v.z; em[4 * k + 3] = v.w; }
	}
	// score + b of this row's target base against query codes 0..3 (bytes of tbl_lo); N rows: b-1 everywhere
	const uint32_t tbl_lo = ti < 4 ? (K.ab << (8 * ti)) : K.nrow;
	// diagonal input of pair 0: (last column of the left lane | first-column value, own column P-1)
	const uint32_t leftv = (uint32_t)grp_shr1<G>((int)H[P - 1], hfc << 16, g0);
	uint32_t hd = __builtin_amdgcn_alignbit(H[P - 1], leftv, 16);
	uint32_t mask = pk_min1(hd);
	uint32_t M[P];
	uint32_t agg = 0;                                          // F leaving each chain if nothing flowed in
	uint32_t kf = K.kf0;
#pragma unroll
	for (int p = 0; p < P; ++p) {
		pk_pair1(M[p], agg, kf, hd, mask, sel[p], K.tbl_hi, tbl_lo, K.b2, K.ei2);
		hd = H[p]; mask = NZ[p];
	}
	agg = pk_subsK(agg, 0x40004000u);
	// F entering the lane: max-plus scan over the lanes of T + e*C*lane, T = what the lane's own cells send to its right neighbour
	uint32_t f;
	{
		const int fout_lo = (int)(agg & 0xFFFFu), fout_hi = (int)(agg >> 16);
		int X = max(fout_lo - K.eP, fout_hi) + eCl;
		X = grp_scan_max<G>(X);
		const int ex = grp_shr1<G>(X, PK_NEG, g0);
		const int fin_lo = max(ex - (eCl - K.eC), 0);
		const int fin_hi = max(fin_lo - K.eP, fout_lo);
		f = (uint32_t)fin_lo | ((uint32_t)fin_hi << 16);
	}
	uint32_t key = 0, nzb = 0, nzb2 = 0;
	pk_pass2<P, SAME_OE>(H, E, NZ, M, em, f, key, nzb, nzb2, K, std::make_integer_sequence<int, P>());
	// last non-zero H column of the lane, + 1 (0: none): pair p sits at bit P-1-p of its half of nzb (branch-free)
	int nlast;
	{
		if (P <= 16) {
			const uint32_t nh = nzb >> 16;
			const uint32_t pick = nh ? nh : (nzb & 0xFFFFu);
			const int base = nh ? j0 + 2 * P : j0 + P;                   // column of pair 0 + P
			nlast = nzb ? base - (int)__builtin_ctz(pick | 0x10000u) : 0;
		} else {                                                     // pairs 16.. (nzb2: pair p at bit P-1-p of its half) below pairs 0..15
			const uint32_t nh = ((nzb >> 16) << (P - 16)) | (nzb2 >> 16), nl = ((nzb & 0xFFFFu) << (P - 16)) | (nzb2 & 0xFFFFu);
			const uint32_t pick = nh ? nh : nl;
			const int base = nh ? j0 + 2 * P : j0 + P;
			nlast = pick ? base - (int)__builtin_ctz(pick) : 0;
		}
	}
	// lane key (h << 16 | column): the high chain wins ties (its columns are the larger ones)
	int kk;
	{
		const uint32_t kl = key & 0xFFFFu, kh = key >> 16;
		constexpr int KSH = P > 16 ? 5 : 4;
		const int Kl = (int)(((kl << (16 - KSH)) & 0xFFFF0000u) | (uint32_t)(j0 + (int)(kl & ((1u << KSH) - 1u))));
		const int Kh = (int)(((kh << (16 - KSH)) & 0xFFFF0000u) | (uint32_t)(j0 + P + (int)(kh & ((1u << KSH) - 1u))));
		kk = max(Kl, Kh);
	}
	grp_allmax2<G>(kk, nlast);
	const int m = kk >> 16, mj = kk & 0xFFFF;
	// gscore: H(i, qlen-1) when the row reaches the query end (ksw.c:942-945)
	const bool ge = run && S.end == qlen;
	if (__any(ge)) {                                            // wave-uniform
		if (owner) {
			uint4 *dst = (uint4 *)hrow;
#pragma unroll
			for (int k = 0; k < PP / 4; ++k) {
				uint4 v;
				v.x = H[4 * k < P ? 4 * k : 0]; v.y = H[4 * k + 1 < P ? 4 * k + 1 : 0]; v.z = H[4 * k + 2 < P ? 4 * k + 2 : 0]; v.w = H[4 * k + 3 < P ? 4 * k + 3 : 0];
				dst[k] = v;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // LDS operations of a wave execute in order; this keeps the compiler from moving the read up
		int h1 = (int)h16[goff];
		h1 = qlen == 0 ? hnx : h1;
		S.max_ie = (ge && !(S.gscore > h1)) ? i : S.max_ie;
		S.gscore = ge ? max(S.gscore, h1) : S.gscore;
	}
	const bool upd = run && m != 0;
	alive = alive && !(run && m == 0);                          // ksw.c:946
	const bool better = upd && m > S.mx;
	S.max_off = better ? max(S.max_off, abs(mj - i)) : S.max_off;
	S.max_i = better ? i : S.max_i;
	S.max_j = better ? mj : S.max_j;
	if (zdrop > 0) {                                            // wave-uniform (ksw.c:951-959)
		const int di = i - S.max_i, dj = mj - S.max_j;
		const int pen = di > dj ? (di - dj) * (int)(K.ed2 & 0xFFFFu) : (dj - di) * (int)(K.ei2 & 0xFFFFu);
		alive = alive && !(upd && !better && S.mx - m - pen > zdrop);
	}
	S.mx = better ? m : S.mx;
	S.end = upd ? min(qlen, nlast + 2) : S.end;                 // ksw.c:963-970: last non-zero index of eh[] is the column + 1
	// Exact early stop (see ext_row): Phi = H + a*(qlen-1-column) over the NON-ZERO cells of the frontier; E(i+1,j) <= H(i,j), so H
	// alone carries it.  Zero cells are lifted out of the maximum by an offset that the non-zero ones carry (NZ * 0x4000).
	if (bound) {                                                // wave-uniform
		// max over the pairs of H[p] - a*p, as a Horner-style chain from the last pair down (one constant instead of P)
#ifdef PK_BOUND_R2      // (A/B builds: round 2's rule, zero cells carry potential too)
		uint32_t u2 = H[P - 1];
#pragma unroll
		for (int p = P - 2; p >= 0; --p) u2 = pk_maxs(H[p], pk_subiK(u2, K.a2));
		u2 = pk_addi(u2, phi0);
		int u = max((int)(short)(u2 & 0xFFFFu), (int)u2 >> 16);
		u = max(u, hnx + K.a * qlen);
		u = grp_allmax<G>(u);
		alive = alive && !(u <= S.mx && u < S.gscore);
#else
		uint32_t u2 = pk_madK(NZ[P - 1], 0x40004000u, H[P - 1]);
#pragma unroll
		for (int p = P - 2; p >= 0; --p) u2 = pk_maxs(pk_madK(NZ[p], 0x40004000u, H[p]), pk_subiK(u2, K.a2));
		u2 = pk_subiK(pk_addi(u2, phi0), 0x40004000u);
		int u = max(max((int)(short)(u2 & 0xFFFFu), (int)u2 >> 16), 0);
		u = max(u, hnx ? hnx + K.a * qlen : 0);
		u = grp_allmax<G>(u);
		u = min(u, max(m, hnx) + K.a * rl);
		const bool fin = u < S.gscore || (o3only && u <= S.mx - end_bonus && S.gscore <= S.mx - end_bonus);
		alive = alive && !(u <= S.mx && fin);
#endif
	}
	return alive;
}

// The kernel: every group of G lanes runs its own alignment and row index and draws its next job from the class counter when
// the alignment ends (as extend16_kernel does with its four rows).
template <int G, int P, bool SAME_OE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(P > PK_WAVES4_MAXP ? 3 : 4))) extpk_kernel(ext_args_t A)
{
	constexpr int C = 2 * P, PP = (P + 3) & ~3, PS = PP + 4, NG = 256 / G;
	const int lane = threadIdx.x & 63, l = lane & (G - 1);
	const bool g0 = l == 0;
	const uint32_t n = A.count[0];
	const uint32_t *ids = A.ids + A.count[1];
	const int oe_ins = A.o_ins + A.e_ins, oe_del = A.o_del + A.e_del;
	const int j0 = l * C;
	const pk_consts_t<P> K = pk_consts<P>(A);
	const int eCl = K.eC * l;
	constexpr int TCAP = PK_TCAP(G);
	// (row strides padded so that the groups of a wave, which read the same offsets of their own rows, hit different banks:
	// unpadded, the 16 groups' target bytes sat in two banks and every row's read was an 8-way conflict)
	__shared__ __attribute__((aligned(16))) uint8_t t_lds[NG][TCAP + 4];
	__shared__ __attribute__((aligned(16))) uint8_t q_lds[4][(C * G + 15) & ~15];      // the query codes of the job a wave is staging
	__shared__ __attribute__((aligned(16))) uint32_t em_tab[(C + 1) * PS];
	__shared__ __attribute__((aligned(16))) uint32_t h_lds[NG * PS];
	for (int k = threadIdx.x; k < (C + 1) * PS; k += 256) {
		const int w = k / PS, p = k % PS;
		em_tab[k] = p < P ? ((p < w ? 0xFFFFu : 0u) | (P + p < w ? 0xFFFF0000u : 0u)) : 0u;
	}
	__syncthreads();
	uint8_t *tl = t_lds[threadIdx.x / G];
	uint8_t *qw = q_lds[threadIdx.x >> 6];
	const int hgrp = (int)(threadIdx.x / G) * PS;              // the group's dwords in h_lds
	uint32_t *hrow = h_lds + hgrp;
	bool owner = false;                                        // this lane holds column qlen-1
	bool have = false, alive = false;
	uint32_t id = 0;
	int qlen = 0, tlen = 0, h0 = 1, i = 0;
	int hfc = 0, dn = 0;
	const uint8_t *tp = tl;
	int goff = 0;                                              // halfword of h_lds that holds H(i, qlen-1)
	uint32_t phi0 = 0;
	uint32_t H[P], E[P], NZ[P], sel[P];
#pragma unroll
	for (int p = 0; p < P; ++p) { H[p] = E[p] = NZ[p] = 0; sel[p] = 0x0C070C07u; }
	pk_rs_t S = {0, 0, -1, -1, -1, -1, 0};
	int rows_done = 0, wave_rows = 0;
	bool more = true, more_g = true;                           // jobs left for this wave / on the class counter (wave-uniform)
	uint32_t qn = 0, qe = 0;
	for (;;) {
		const unsigned long long nb = __ballot(!alive && g0 && (more || have));
		if (nb) {
			if (!alive && have && g0) {                          // results of the alignment that just ended
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
			// jobs come from a wave-local range [qn, qe) refilled EXT_DRAW_CHUNK at a time: one atomic per job on the class counter
			// (a single address: ~90 atomics/us for the whole chip) was what short jobs waited for
			if (qn == qe && more_g) {
				uint32_t b0 = 0;
				if (lane == 0) b0 = atomicAdd(A.ctr, (uint32_t)EXT_DRAW_CHUNK);
				qn = __builtin_amdgcn_readfirstlane(b0);
				qe = qn + EXT_DRAW_CHUNK < n ? qn + EXT_DRAW_CHUNK : n;
				if (qn >= n) { qn = qe = n; }
				more_g = qe < n;
			}
			const uint32_t base = qn;
			{
				const uint32_t cnt = (uint32_t)__builtin_popcountll(nb), avail = qe - qn;
				qn += cnt < avail ? cnt : avail;            // groups beyond `avail` stay idle this round and ask again
			}
			more = more_g || qn < qe;
			const bool drawing = !alive;
			job_src_t src = ext_job_src(A, 0, false, 0, 0);
			if (drawing) {
				const uint32_t k = base + (uint32_t)__builtin_popcountll(nb & ((1ull << (lane & ~(G - 1))) - 1));
				have = k < qe;
				id = have ? ids[n - 1 - k] : 0;
				qlen = have ? (int)A.qlen[id] : 0; tlen = have ? (int)A.tlen[id] : 0; h0 = have ? (int)A.h0[id] : 1;
				src = ext_job_src(A, id, have, qlen, tlen);
				{
					const int jq = qlen > 0 ? qlen - 1 : 0, lq = jq / C, r = jq % C;       // lane, chain and pair of column qlen-1
					goff = 2 * (hgrp + (r >= P ? r - P : r)) + (r >= P ? 1 : 0); owner = l == lq;
					const int pl = K.a * (qlen - 1 - j0), ph = K.a * (qlen - 1 - j0 - P);
					phi0 = ((uint32_t)pl & 0xFFFFu) | ((uint32_t)ph << 16);
				}
				S.end = qlen; S.mx = h0; S.max_i = -1; S.max_j = -1; S.max_ie = -1; S.gscore = -1; S.max_off = 0;
				i = 0; rows_done = 0; hfc = h0; dn = oe_del; tp = tl;
			}
			// The bases of the new jobs, staged by ALL 64 lanes of the wave one job at a time (eight target rows and four query
			// columns per lane): done by the four lanes of the drawing group alone, a draw cost the wave two to three DP rows
			// of instructions during which its other fifteen alignments stood still.
			for (unsigned long long todo = __ballot(drawing && have && g0 && tlen > 0); todo; todo &= todo - 1) {
				const int gl = (int)__builtin_ctzll(todo);                      // first lane of the group (wave-uniform)
				const int tlen_g = __builtin_amdgcn_readlane(tlen, gl), qlen_g = __builtin_amdgcn_readlane(qlen, gl);
				job_src_t sg;
				sg.qp = (const uint8_t *)pk_readlane64((unsigned long long)src.qp, gl); sg.qstep = __builtin_amdgcn_readlane(src.qstep, gl);
				sg.tp = (const uint8_t *)pk_readlane64((unsigned long long)src.tp, gl);
				sg.t0 = (long long)pk_readlane64((unsigned long long)src.t0, gl); sg.tdir = __builtin_amdgcn_readlane(src.tdir, gl);
				uint8_t *tg = t_lds[(threadIdx.x >> 6) * (64 / G) + gl / G];
				for (int k0 = 8 * lane; k0 < tlen_g; k0 += 512) {
					uint32_t lo, hi;
					pk_t8(A, sg, k0, tlen_g, lo, hi);
					*(uint32_t *)(tg + k0) = lo;
					if (k0 + 4 < tlen_g) *(uint32_t *)(tg + k0 + 4) = hi;
				}
				for (int c0 = 4 * lane; c0 < C * G; c0 += 256) {
					uint32_t w = 0;
#pragma unroll
					for (int u = 0; u < 4; ++u) { const int j = c0 + u; w |= (uint32_t)(j < qlen_g ? min(ext_q_at(A, sg, j), 4) : 7) << (8 * u); }      // 4 = N, 7 = pad
					*(uint32_t *)(qw + c0) = w;
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (LDS operations of a wave execute in order; the fences hold the compiler to it)
				if ((lane & ~(G - 1)) == gl) {
#pragma unroll
					for (int p = 0; p < P; ++p) sel[p] = 0x0C000C00u | (uint32_t)qw[j0 + p] | ((uint32_t)qw[j0 + P + p] << 16);
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			}
			if (drawing) {
				// H(-1, j) = max(0, h0 - o_ins - e_ins*(j+1)) left of qlen (ksw.c:880-883): one saturating packed subtract per pair
				const int wq = min(max(qlen - j0, 0), C);
				const uint4 *mrow = (const uint4 *)(em_tab + wq * PS);
				uint32_t em[PP];
#pragma unroll
				for (int k = 0; k < PP / 4; ++k) { const uint4 v = mrow[k]; em[4 * k] = v.x; em[4 * k + 1] = v.y; em[4 * k + 2] = v.z; em[4 * k + 3] = v.w; }
				uint32_t X = (uint32_t)max(h0 - oe_ins - j0 * A.e_ins, 0) | ((uint32_t)max(h0 - oe_ins - (j0 + P) * A.e_ins, 0) << 16);
#pragma unroll
				for (int p = 0; p < P; ++p) {
					H[p] = X & em[p];
					NZ[p] = pk_min1(H[p]);
					E[p] = 0;
					X = pk_subsK(X, K.ei2);
					if (!have) sel[p] = 0x0C070C07u;
				}
				alive = have && tlen > 0;
				if (have && tlen == 0) {
					if (g0) {
						int32_t *o = A.out + 3 * (size_t)id;
						o[0] = h0; o[1] = 0; o[2] = 0;
						if (A.raw) { int32_t *r = A.raw + 6 * (size_t)id; r[0] = h0; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = -1; r[5] = 0; }
						if (A.stats) atomicAdd(A.stats + 2, 1ull);
					}
					have = false;
				}
			}
		}
		if (!__any(alive) && !more) break;
		// (no `continue` past the row when every group drew a job without rows: a second path around the row body makes the
		// compiler keep two copies of all loop-carried pairs; the idle row is harmless and rare)
		++wave_rows;
		const int ti = (int)*tp;
		const bool run = alive;
		if (A.stats) {          // (BMH_EXT_STATS) wave-rows in which every running alignment has reached the query end: candidates of a row without end masks
			const bool all_at_end = !__any(run && S.end != qlen), any_run = __any(run);
			if (lane == 0 && any_run) { atomicAdd(A.stats + 4, 1ull); if (all_at_end) atomicAdd(A.stats + 5, 1ull); }
		}
		rows_done += run ? 1 : 0;
		const int hnx = max(0, h0 - dn);
		alive = pk_row<G, P, SAME_OE>(K, A.zdrop, H, E, NZ, sel, ti, run, hfc, hnx, i, qlen, j0, eCl, g0, phi0, em_tab, hrow, owner, (const uint16_t *)h_lds, goff, S, alive, (wave_rows & PK_BOUND_MASK) == 0, tlen - 1 - i, A.raw == nullptr, A.end_bonus);
		if (run) { ++i; ++tp; hfc = hnx; dn += A.e_del; }
		alive = alive && i < tlen;
	}
	if (A.stats && lane == 0) atomicAdd(A.stats + 3, (unsigned long long)wave_rows);
}
