// Packed 16-bit rows of the seed-extension DP (ksw_extend2, /root/reference/src/ksw.c:864-986) -- included by extend_kernels.hip.
//
// Same contract and the same row-at-a-time mapping as extend16_kernel, rebuilt for instruction count (the stage runs at the
// VALU issue ceiling, so its time IS its instruction count):
//   * two query columns per VGPR: H, E, M, F live as unsigned 16-bit pairs and move through v_pk_{mad,sub(clamp),max,min}_u16.
//     The reference clamps M-oe, E and F at zero, which is what an unsigned saturating subtract does for free.
//     Lane l of a group owns the 2P columns [l*2P, l*2P+2P): low halves = its first P columns, high halves = the next P, so the
//     diagonal neighbour of pair p is simply pair p-1 of the previous row (pair 0: one DPP shift + v_alignbit).
//   * G = 4 lanes per alignment up to 136 query columns and, with 24 / 28 / 32 pairs a lane, from 161 to 256 (8 lanes for 137 .. 160 and, 18 pairs a lane, 257 .. 288): SIXTEEN
//     alignments per wave share the per-row control code (scan, reductions, end / maximum bookkeeping), which is what a row costs besides its cells.
//   * substitution scores by ONE v_perm_b32 per pair from an 8-byte row table {score(t, code) + b}: M = (hd != 0) * scb + hd -sat b.
//   * no `beg` bookkeeping: the first-column value max(0, h0 - o_del - e_del*(i+1)) applies in every row -- beg > 0 implies it has
//     reached zero for good, and cells left of beg have zero inputs (scripts/extpk_model.py checks this algebra against the oracle).
//   * only H is masked at and right of `end` (M and E are zero there by themselves); the masks come from a small LDS table.
//   * F: per-chain local recurrence, 32-bit max-plus scan over the group's lanes (DPP), exact inflow for both chains of a lane.
//   * row maximum and its last column from max over (h << 4 | pair) keys; new end = last non-zero H column + 3.
//   * H(i, qlen-1) for gscore: the lanes park their H pairs in LDS, the group reads the one halfword it needs (VALU-free).
//   * jobs (round 5): 32-byte records in list order, taken PK_CHUNK at a time into the wave's LDS; in the four-lane classes the NEXT job's
//     bases are always on their way (two dwords per lane in flight across rows, decoded into a spare LDS row), so that a draw costs no
//     round trip to memory; one wave-uniform branch per row around all of it (extpk_body).
// Eligibility (ext_route): 1 <= b, a + b <= 255, h0 + qlen*a < 4096, qlen <= 288, tlen <= PK_TCAP(G); everything else takes the
// 32-bit kernels.  Results are bit-identical to those (tests/test_gpu_parity.py runs both).
#pragma once

// target bases of an alignment staged in LDS, by group width (the 4-lane groups are the many small queries: 64 of them per block)
#ifndef PK_TCAP4
#define PK_TCAP4 384
#define PK_TCAP8 512
#define PK_TCAP16 640
#endif
#define PK_TCAP(G) ((G) <= 4 ? PK_TCAP4 : (G) == 8 ? PK_TCAP8 : PK_TCAP16)
#ifndef PK_WIDE18
#define PK_WIDE18 1           // the queries of 257 .. 288 columns on EIGHT lanes of 18 pairs (eight alignments per wave, 161 registers) instead of sixteen lanes of 9 (four):
                              // 46 instead of 54 wave-instructions per alignment row; 300 bp reads: extension 65.2 -> 62.2 ms, step 90.9 -> 87.7 ms (three interleaved pairs)
#endif
#ifndef PK_WIDE4
#define PK_WIDE4 1            // the queries of 161 .. 256 columns on FOUR lanes of 24 / 28 / 32 pairs (sixteen alignments per wave; 216 / 237 / 256 registers: two waves per SIMD, which still
                              // issue at the SIMD's rate) instead of eight lanes of 12 / 14 / 16: 300 bp reads: extension 62.4 -> 60.9 ms, step 88.2 -> 87.0 ms (three interleaved pairs)
#endif
#ifndef PK_G2
#define PK_G2 0               // the queries of up to 128 columns on TWO lanes of 8 .. 32 pairs (thirty-two alignments per wave) where their targets are short enough (see PK_TCAPGP)
#endif
// (the many-pair classes take the wider groups' jobs: their target rows.  The two-lane classes hold thirty-two target rows a wave: rows of 8 P = twice the class's longest
// query -- what mem_chain2aln's windows come to under the default scoring --, so that two blocks of the widest and four of the others share a CU's LDS; longer targets stay with four lanes)
#define PK_TCAPGP(G, P) ((G) == 2 ? 8 * (P) : (G) == 8 && (P) == 18 ? PK_TCAP16 : (G) == 4 && (P) > 18 ? PK_TCAP8 : PK_TCAP(G))
#define PK_WAVES2(G, P) ((P) > 18)
#ifndef PK_WAVES4_MAXP
#define PK_WAVES4_MAXP 8      // classes of up to this many pairs per lane run four waves per SIMD (registers and grid; 120 VGPRs at 10 pairs), the larger ones three (8 -> 10: -0.5 % at 150 bp, -0.7 % at 300 bp)
#endif
#ifndef PK_BOUND_MASK
#define PK_BOUND_MASK 15      // the exact early-stop bound is evaluated every (mask + 1)-th row of a wave (every 2nd / 4th / 8th / 16th / 32nd: 17.6 / 17.3 / 17.2 / 17.1 / 17.4 ms)
#endif
#ifndef PK_EM_EAGER
#define PK_EM_EAGER(G, P) 0      // (A/B knob: classes whose end masks are all loaded before the row's first pass, round 4's order, instead of one uint4 at a time inside the second)
#endif
#define PK_HMAX 4096          // scores stay below this (keys are h << 4 | pair in 16 bits)
#define PK_HMAX17 2048        // ... in the 17-pair class (keys h << 5 | pair)
#define PK_NEG (-(1 << 28))
#ifndef PK_PERSIST_DEFAULT
#define PK_PERSIST_DEFAULT 0  // blocks per CU of the persistent packed kernel (0: one kernel per class); knob EXT_PERSIST
#endif

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
// Where the eight target rows k0..k0+7 of a descriptor job sit in the 2-bit text, and their decoding from the dword loaded there --
// split so that the load can be issued long before its result is needed (extpk_body fetches the next job's bases ahead).
struct pk_t8w_t { long long fa; int D; bool asc, rev; };
__device__ __forceinline__ pk_t8w_t pk_t8_where(const long long l_pac, const long long t0, const int tdir, const int k0)
{
	pk_t8w_t w;
	const long long p0 = t0 + (long long)k0 * tdir;
	w.rev = p0 >= l_pac;
	const long long f0 = w.rev ? (l_pac << 1) - 1 - p0 : p0;        // forward-strand position of row k0
	w.asc = w.rev ? tdir < 0 : tdir > 0;                            // the rows walk the forward strand upwards
	w.fa = w.asc ? f0 : f0 - 7;                                     // lowest position of the eight
	w.D = 7;                                                        // descending: row u is symbol D - u of the window at fa
	if (w.fa < 0) { w.D = (int)f0; w.fa = 0; }
	return w;
}
__device__ __forceinline__ void pk_t8_decode(const uint32_t raw, const pk_t8w_t &w, uint32_t &lo, uint32_t &hi)
{
	const uint32_t v = __builtin_bswap32(raw) >> 8;
	const uint32_t y = (v << (8 + 2 * ((int)w.fa & 3))) >> 16;      // eight symbols, position fa+j at bits 15-2j:14-2j
	const uint32_t z = w.asc ? y : y >> (2 * (7 - w.D));
	const uint32_t w0 = pk_spread4(z & 0xFFu), w1 = pk_spread4((z >> 8) & 0xFFu);
	lo = w.asc ? __builtin_bswap32(w1) : w0;
	hi = w.asc ? __builtin_bswap32(w0) : w1;
	if (w.rev) { lo ^= 0x03030303u; hi ^= 0x03030303u; }
}
// Four query columns c0..c0+3 of a descriptor job from ONE dword of the ASCII reads.  The dword is the four bytes at the columns' lowest
// address, moved into the query segment where it would reach outside (the lane at the segment's end: a load must not leave the
// reads' buffer); sh = how many bytes it was moved.  Needs qlen >= 4.
// (pk_q4_offset: the same as plain integers relative to qp -- the load address is qp + offset, and the decoding recomputes sh from the offsets alone)
__device__ __forceinline__ int pk_q4_offset(const int qstep, const int qlen, const int c0, int &sh)
{
	const int lo = qstep > 0 ? c0 : -c0 - 3;                   // lowest address of the four columns, relative to qp
	const int seg = qstep > 0 ? 0 : -(qlen - 1);               // ... of the segment
	const int ld = lo < seg ? seg : (lo > seg + (qlen - 4) ? seg + (qlen - 4) : lo);
	sh = lo - ld;
	return ld;
}
__device__ __forceinline__ const uint8_t *pk_q4_where(const uint8_t *qp, const int qstep, const int qlen, const int c0, int &sh)
{
	return qp + pk_q4_offset(qstep, qlen, c0, sh);
}
// ... and its decoding: codes 0..3, 4 = anything else (N), 7 = pad at and beyond qlen, column c0+u in byte u (what ext_q_at gives byte by byte)
__device__ __forceinline__ uint32_t pk_q4_decode(const uint32_t raw, const int sh, const int qstep, const int qlen, const int c0)
{
	uint32_t v = sh >= 0 ? raw >> (8 * sh) : raw << (8 * -sh);       // byte u = the byte at (lowest address + u)
	if (qstep < 0) v = __builtin_bswap32(v);
	const uint32_t W = v & 0xDFDFDFDFu, t = (W >> 1) & 0x03030303u;
	uint32_t code = t ^ ((t >> 1) & 0x01010101u);
	const uint32_t x = __builtin_amdgcn_perm(0u, 0x54474341u, code) ^ W;                    // non-zero byte: not the letter the code stands for
	const uint32_t m = (((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u) >> 7) * 0xFFu;
	code = (code & ~m) | (0x04040404u & m);
	const int nv = qlen - c0;
	const uint32_t vm = nv >= 4 ? 0xFFFFFFFFu : nv <= 0 ? 0u : (1u << (8 * nv)) - 1u;
	return (code & vm) | (0x07070707u & ~vm);
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
	const pk_t8w_t w = pk_t8_where(A.l_pac, s.t0, s.tdir, k0);
	// three bytes of the text as ONE unaligned dword load (the 64 lanes of a staging step read 64 different lines: what such a step
	// costs is its number of load instructions; pac is readable 9 bytes past its end)
	uint32_t raw; __builtin_memcpy(&raw, A.pac + (w.fa >> 2), 4);
	pk_t8_decode(raw, w, lo, hi);
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

// ---- group primitives: G = 16 is one DPP row, G = 8 half of one, G = 4 a quad, G = 2 half a quad
template <int G> __device__ __forceinline__ int grp_shr1(int v, int fill, bool g0)      // lane-1 of the group; its lane 0 receives `fill`
{
	int t = __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false);
	if (G < 16) t = g0 ? fill : t;
	return t;
}
template <int G> __device__ __forceinline__ int grp_scan_max(int v)                     // inclusive max-scan over the group
{
	if (G == 16) return row_scan_max_f(v);
	if (G == 2) {
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
		             : "+v"(v));
		return v;
	}
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
	else if (G == 2)
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
		             "v_max_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
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
	else if (G == 2)
		asm volatile("s_nop 1\n\t"
		             "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
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

// second pass over four pairs (one uint4 of end masks); FIRST = the chunk's first pair
template <int P, bool SAME_OE, int FIRST, int... Is>
__device__ __forceinline__ void pk_pass2_chunk(uint32_t (&H)[P], uint32_t (&E)[P], uint32_t (&NZ)[P], const uint32_t (&M)[P], const uint4 em,
                                               uint32_t &f, uint32_t &key, uint32_t &nzb, uint32_t &nzb2, const pk_consts_t<P> &K, std::integer_sequence<int, Is...>)
{
	// (keys are h << 4 | pair up to 16 pairs, h << 5 | pair beyond; the non-zero bits of pairs 16.. go to a second register)
	const uint32_t e[4] = {em.x, em.y, em.z, em.w};
	(pk_pair2<FIRST + Is, SAME_OE, (P > 16 ? 32 : 16)>(H[FIRST + Is], E[FIRST + Is], NZ[FIRST + Is], f, key, FIRST + Is < 16 ? nzb : nzb2, M[FIRST + Is], e[Is], K.ei2, K.ed2, K.oei2, K.oed2), ...);
}
// the whole second pass: the end masks of the lane come from LDS one uint4 (four pairs) at a time, each asked for while the four pairs
// before it are worked on -- all PP of them held at once were the registers that the job pipeline's loads in flight needed
template <int P, bool SAME_OE, int K0>
__device__ __forceinline__ void pk_pass2(uint32_t (&H)[P], uint32_t (&E)[P], uint32_t (&NZ)[P], const uint32_t (&M)[P], const uint4 *src, const uint4 cur,
                                         uint32_t &f, uint32_t &key, uint32_t &nzb, uint32_t &nzb2, const pk_consts_t<P> &K)
{
	constexpr int PP = (P + 3) & ~3, NCH = PP / 4, CNT = (K0 + 1) * 4 <= P ? 4 : P - K0 * 4;
	if constexpr (K0 + 1 < NCH) {
		const uint4 nxt = src[K0 + 1];
		pk_pass2_chunk<P, SAME_OE, K0 * 4>(H, E, NZ, M, cur, f, key, nzb, nzb2, K, std::make_integer_sequence<int, CNT>());
		pk_pass2<P, SAME_OE, K0 + 1>(H, E, NZ, M, src, nxt, f, key, nzb, nzb2, K);
	} else {
		pk_pass2_chunk<P, SAME_OE, K0 * 4>(H, E, NZ, M, cur, f, key, nzb, nzb2, K, std::make_integer_sequence<int, CNT>());
	}
}

// ... and with all masks in registers before the row's first pass (classes with registers to spare)
template <int P, bool SAME_OE, int K0>
__device__ __forceinline__ void pk_pass2_eager(uint32_t (&H)[P], uint32_t (&E)[P], uint32_t (&NZ)[P], const uint32_t (&M)[P], const uint4 (&em)[((P + 3) & ~3) / 4],
                                               uint32_t &f, uint32_t &key, uint32_t &nzb, uint32_t &nzb2, const pk_consts_t<P> &K)
{
	constexpr int PP = (P + 3) & ~3, NCH = PP / 4, CNT = (K0 + 1) * 4 <= P ? 4 : P - K0 * 4;
	pk_pass2_chunk<P, SAME_OE, K0 * 4>(H, E, NZ, M, em[K0], f, key, nzb, nzb2, K, std::make_integer_sequence<int, CNT>());
	if constexpr (K0 + 1 < NCH) pk_pass2_eager<P, SAME_OE, K0 + 1>(H, E, NZ, M, em, f, key, nzb, nzb2, K);
}

// One DP row of the wave's alignments.  em_tab: the end masks, [2P+1][PP] dwords in LDS; hrow: the group's PP dwords of the H
// parking area, written by the lane that owns column qlen-1 (`owner`); h16[goff]: where the group finds H(i, qlen-1) in it.
// Returns the new `alive`.
template <int G, int P, bool SAME_OE>
__device__ __forceinline__ bool pk_row(const pk_consts_t<P> &K, const int zdrop, uint32_t (&H)[P], uint32_t (&E)[P], uint32_t (&NZ)[P], const uint32_t (&sel)[P],
                                       const int ti, const bool run, const int hfc, const int hnx, const int i, const int qlen,
                                       const int j0, const int eCl, const bool g0,
                                       const uint32_t *em_tab, uint32_t *hrow, const bool owner, const uint16_t *h16, const int goff,
                                       pk_rs_t &S, bool alive, const bool bound, const int rl, const bool o3only, const int end_bonus)
{
	constexpr int C = 2 * P, PP = (P + 3) & ~3, PS = PP + 4;      // PS: LDS row stride in dwords (odd multiple of 4: rows start in different banks)
	// end masks of this lane: cells [0, wend) of the lane are left of `end`
	const int wend = run ? S.end - j0 : 0;
	const int wc = min(max(wend, 0), C);
	const uint4 *em_src = (const uint4 *)(em_tab + wc * PS);
	const uint4 em0 = em_src[0];
	uint4 em_all[PP / 4];
	if constexpr (PK_EM_EAGER(G, P)) {
#pragma unroll
		for (int k = 0; k < PP / 4; ++k) em_all[k] = em_src[k];
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
	if constexpr (PK_EM_EAGER(G, P)) pk_pass2_eager<P, SAME_OE, 0>(H, E, NZ, M, em_all, f, key, nzb, nzb2, K);
	else pk_pass2<P, SAME_OE, 0>(H, E, NZ, M, em_src, em0, f, key, nzb, nzb2, K);
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
		// potential of the lane's first column of either chain: a * (columns right of it)
		const uint32_t phi0 = ((uint32_t)(K.a * (qlen - 1 - j0)) & 0xFFFFu) | ((uint32_t)(K.a * (qlen - 1 - j0 - P)) << 16);
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

#ifndef PK_CHUNK
#define PK_CHUNK 16           // jobs a wave takes from its class counter per atomic (their records: one coalesced load into the wave's LDS); 16 / 32 / 64: extension 13.9-14.1 / 14.3-14.5 / 14.6-14.8 ms (the last chunks of a class are its tail); 8 / 4: 14.1 / 14.5-15.4 (the draws); the END of a list in smaller chunks (what is left shared out among 256 / 1024 waves, a look at the counter before the atomic): 14.8-15.3 / 15.2-15.6
#endif
// classes whose next job's bases are fetched AHEAD (loads in flight across rows): the four-lane classes, whose jobs are short and whose draws are
// frequent.  The eight- and sixteen-lane classes (300 bp reads) draw rarely, and the two to four registers the loads in flight occupy cost their
// nine- and ten-pair kernels the fourth wave per SIMD: measured at 300 bp with everything fetched ahead, the extension 66.5 -> 70.0 ms
#define PK_AHEAD(G) ((G) <= 4)
#define PK_WAVES4(G, P) ((G) <= 4 ? (P) <= PK_WAVES4_MAXP : (P) <= 10)
#ifndef PK_PREFETCH_AGE
#define PK_PREFETCH_AGE 3     // rows after which the bases fetched ahead are taken out of their registers (they have long arrived by then)
#endif
// LDS of one wave, by class: the target rows of its 64 / G groups + the one the next job is staged in (row strides padded so that the
// groups of a wave, which read the same offsets of their own rows, hit different banks: unpadded, the 16 groups' target bytes sat in
// two banks and every row's read was an 8-way conflict), the query codes of the job the wave is staging, the H parking rows, the
// records of its chunk of jobs; the end-mask table is per block in the per-class kernels and per wave in the persistent one
template <int G, int P> struct pk_lds_t {
	static constexpr int C = 2 * P, PP = (P + 3) & ~3, PS = PP + 4, NGW = 64 / G, TCAP = PK_TCAPGP(G, P);
	static constexpr int T_BYTES = ((NGW + 1) * (TCAP + 4) + 15) & ~15, Q_BYTES = (C * G + 15) & ~15, EM_DWORDS = (C + 1) * PS, H_DWORDS = NGW * PS, RC_BYTES = PK_CHUNK * 32;
	static constexpr int WAVE_BYTES = T_BYTES + Q_BYTES + 4 * EM_DWORDS + 4 * H_DWORDS + RC_BYTES;      // (all parts are multiples of 16 bytes)
};
template <int P> __device__ __forceinline__ void pk_em_init(uint32_t *em_tab, const int first, const int step)
{
	constexpr int C = 2 * P, PP = (P + 3) & ~3, PS = PP + 4;
	for (int k = first; k < (C + 1) * PS; k += step) {
		const int w = k / PS, p = k % PS;
		em_tab[k] = p < P ? ((p < w ? 0xFFFFu : 0u) | (P + p < w ? 0xFFFF0000u : 0u)) : 0u;
	}
}

// The body of a class: every group of G lanes of the wave runs its own alignment and row index and takes its next job when the
// alignment ends; returns when the class is drained and the wave's alignments are through.  t_wave: the wave's 64 / G + 1 target rows
// of TCAP + 4 bytes; qw: its query staging row; em_tab: the end masks; h_wave: its H parking rows; rc_wave: PK_CHUNK job records.
// A.count / A.ctr / A.recs are the class's.
//
// The job pipeline (round 5).  A wave used to stand still at every draw for a chain of dependent round trips to memory -- the class
// counter, ids[], qlen / tlen / h0, the descriptor, the bases -- with all sixteen alignments waiting: alone on its SIMD a wave spent
// half its time there (3 850 cycles per row of which ~1 950 are the row), and beside the gather-bound seeding kernels of another batch,
// whose requests fill the memory system's queues, far more: that, not the register file, was why the two did not share the chip
// (profiles/r05_corun.txt).  Now: (1) the jobs of a class lie as 32-byte RECORDS in list order (ext_scatter_kernel); a wave takes
// PK_CHUNK of them with one atomic and one coalesced load into its LDS; (2) ONE JOB IS ALWAYS STAGED AHEAD: as soon as the staged job
// is handed to a group the next record is read from LDS and the loads of its bases are issued -- one dword of the 2-bit text per eight
// rows and one dword of the reads per four columns, per lane -- into registers that nobody looks at for the next PK_PREFETCH_AGE rows
// (or until a group asks); decoding them into the spare target row and the query row costs no wait then.  A group that ends takes the
// spare row as its own and leaves its old one as the next spare.  Only when several groups end within the same few rows (the start of
// a class, very short jobs) does the wave wait for a load, as before.
template <int G, int P, bool SAME_OE>
__device__ __forceinline__ void extpk_body(const ext_args_t &A, uint8_t *t_wave, uint8_t *qw, const uint32_t *em_tab, uint32_t *h_wave, uint4 *rc_wave)
{
	constexpr int C = 2 * P, PP = (P + 3) & ~3, PS = PP + 4, TCAP = PK_TCAPGP(G, P), TROW = TCAP + 4, NGW = 64 / G;
	constexpr int NT = (TCAP + 511) / 512, NQ = (C * G + 255) / 256;      // dwords a lane fetches ahead: eight target rows / four query columns each
	const int lane = threadIdx.x & 63, l = lane & (G - 1);
	const bool g0 = l == 0;
	const uint32_t n = A.count[0];
	const uint4 *recs = A.recs + 2 * (size_t)A.count[1];
	const int oe_ins = A.o_ins + A.e_ins, oe_del = A.o_del + A.e_del;
	const int j0 = l * C;
	const pk_consts_t<P> K = pk_consts<P>(A);
	const int eCl = K.eC * l;
	int trow = (lane / G) * TROW;                              // the group's target row in t_wave (rows change hands at every draw)
	int spare = NGW * TROW;                                    // the row the next job is staged in (wave-uniform)
	const int hgrp = (lane / G) * PS;                          // the group's dwords in h_wave
	uint32_t *hrow = h_wave + hgrp;
	bool owner = false;                                        // this lane holds column qlen-1
	bool have = false, alive = false;
	int qlen = 0, tlen = 0, i = 0;
	int hfc = 0, hnx = 0;                                      // H(i-1, -1) and H(i, -1): max(0, h0 - o_del - e_del (i + 1)), stepped row by row (once 0, 0 for good)
	int goff = 0;                                              // halfword of h_wave that holds H(i, qlen-1)
	uint32_t H[P], E[P], NZ[P], sel[P];
#pragma unroll
	for (int p = 0; p < P; ++p) { H[p] = E[p] = NZ[p] = 0; sel[p] = 0x0C070C07u; }
	pk_rs_t S = {0, 0, -1, -1, -1, -1, 0};
	int wave_rows = 0;                                         // (rows an alignment has executed = its i)
	// the job pipeline (all wave-uniform but the raw dwords)
	uint32_t qn = 0, qe = 0, cbase = 0;                        // the wave's chunk [qn, qe) of the class list; rc_wave holds the records of [cbase, qe)
	bool more_g = n > 0;                                       // jobs left on the class counter
	bool pre = false, pend = false, p_async = false;           // the next job is staged / its bases are on their way (as loads in flight)
	int age = 0;
	uint32_t p_slot = 0;                                       // its record's place in rc_wave (the record itself is re-read from LDS where it is needed: eight scalars carried
	                                                           // through the row loop were eight more spilled into vector lanes and fetched back row after row)
	uint32_t raw_t[NT], raw_q[NQ];
#pragma unroll
	for (int u = 0; u < NT; ++u) raw_t[u] = 0;
#pragma unroll
	for (int u = 0; u < NQ; ++u) raw_q[u] = 0;
	for (;;) {
		unsigned long long reqs = __ballot(!alive && g0);        // groups without a running alignment
		// ONE wave-uniform branch around everything that is not a DP row: in the steady state -- a job staged ahead, every group running -- a row pays the ballot and this
		// test (the nine- and ten-pair rows of the eight- and sixteen-lane classes are short: five tests a row, with their scalar arithmetic, were 3 % of the 300 bp extension)
		if (reqs || !pre) {
			if (reqs) {
				if (!alive && have && g0) {                          // results of the alignment that just ended
					const int qle = S.max_j + 1, tle = S.max_i + 1, gtle = S.max_ie + 1;
					const uint32_t id = hrow[PP];                    // (the job's id waits in a padding word of the group's parking row)
					int32_t *o = A.out + 3 * (size_t)id;
					if (S.gscore <= 0 || S.gscore <= S.mx - A.end_bonus) { o[0] = S.mx; o[1] = qle; o[2] = tle; }
					else { o[0] = S.gscore; o[1] = qlen; o[2] = gtle; }
					if (A.raw) {
						int32_t *r = A.raw + 6 * (size_t)id;
						r[0] = S.mx; r[1] = qle; r[2] = tle; r[3] = gtle; r[4] = S.gscore; r[5] = S.max_off;
					}
					if (A.stats) { atomicAdd(A.stats, (unsigned long long)i); atomicAdd(A.stats + 1, (unsigned long long)tlen); atomicAdd(A.stats + 2, 1ull); }
				}
				have = have && alive;
			}
			// serve the groups that ask, one job each, then leave the next job on its way
			for (;;) {
				const bool waiting = reqs != 0;
				if (!pre) {
					if (!pend) {
						if (qn == qe && more_g) {                        // the next chunk: one atomic on the class counter, its records in one coalesced load
							uint32_t b0 = 0;
							if (lane == 0) b0 = atomicAdd(A.ctr, (uint32_t)PK_CHUNK);
							qn = __builtin_amdgcn_readfirstlane(b0);
							qe = qn + PK_CHUNK < n ? qn + PK_CHUNK : n;
							if (qn >= n) { qn = qe = n; }
							more_g = qe < n;
							cbase = qn;
							if ((uint32_t)lane < qe - qn) {               // (the class kernels draw from the long end of the list)
								const uint4 *r = recs + 2 * (size_t)(n - 1 - (qn + (uint32_t)lane));
								rc_wave[2 * lane] = r[0]; rc_wave[2 * lane + 1] = r[1];
							}
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (LDS operations of a wave execute in order; the fences hold the compiler to it)
						}
						if (qn < qe) {
							p_slot = qn - cbase;
							const uint4 r0 = rc_wave[2 * p_slot], r1 = rc_wave[2 * p_slot + 1];
							++qn;
							const uint32_t p_qt = __builtin_amdgcn_readfirstlane(r0.y), p_w3 = __builtin_amdgcn_readfirstlane(r0.w), p_side = __builtin_amdgcn_readfirstlane(r1.x);
							const uint32_t p_t0l = __builtin_amdgcn_readfirstlane(r1.z), p_t0h = __builtin_amdgcn_readfirstlane(r1.w);
							const int ql = (int)(p_qt & 0xFFFFu), tln = (int)(p_qt >> 16);
							p_async = PK_AHEAD(G) && A.desc && ql >= 4;
							if (p_async) {                               // the bases: loads only, nobody waits for them here
								const bool left = p_side == 0;
								const long long t0 = (long long)(((unsigned long long)p_t0h << 32) | p_t0l) + (left ? tln - 1 : 0);
								const uint8_t *qp = A.reads + p_w3 + (left ? ql - 1 : 0);
	#pragma unroll
								for (int u = 0; u < NT; ++u) {
									const int k0 = 8 * lane + 512 * u;
									if (k0 < tln) { const pk_t8w_t w = pk_t8_where(A.l_pac, t0, left ? -1 : 1, k0); __builtin_memcpy(&raw_t[u], A.pac + (w.fa >> 2), 4); }
								}
	#pragma unroll
								for (int u = 0; u < NQ; ++u) {
									const int c0 = 4 * lane + 256 * u;
									if (c0 < ql) { int sh; const uint8_t *a = pk_q4_where(qp, left ? -1 : 1, ql, c0, sh); __builtin_memcpy(&raw_q[u], a, 4); }
								}
							}
							pend = true; age = 0;
						}
					}
					if (pend && (waiting || age >= PK_PREFETCH_AGE)) {   // into the spare row and the query row
						const uint4 r0 = rc_wave[2 * p_slot], r1 = rc_wave[2 * p_slot + 1];
						const uint32_t p_qt = __builtin_amdgcn_readfirstlane(r0.y), p_w3 = __builtin_amdgcn_readfirstlane(r0.w), p_side = __builtin_amdgcn_readfirstlane(r1.x);
						const uint32_t p_w5 = __builtin_amdgcn_readfirstlane(r1.y), p_t0l = __builtin_amdgcn_readfirstlane(r1.z), p_t0h = __builtin_amdgcn_readfirstlane(r1.w);
						const int ql = (int)(p_qt & 0xFFFFu), tln = (int)(p_qt >> 16);
						const bool left = A.desc && p_side == 0;
						uint8_t *tg = t_wave + spare;
						if (p_async) {
							const long long t0 = (long long)(((unsigned long long)p_t0h << 32) | p_t0l) + (left ? tln - 1 : 0);
	#pragma unroll
							for (int u = 0; u < NT; ++u) {
								const int k0 = 8 * lane + 512 * u;
								if (k0 < tln) {
									uint32_t lo, hi;
									pk_t8_decode(raw_t[u], pk_t8_where(A.l_pac, t0, left ? -1 : 1, k0), lo, hi);
									*(uint32_t *)(tg + k0) = lo;
									if (k0 + 4 < tln) *(uint32_t *)(tg + k0 + 4) = hi;
								}
							}
	#pragma unroll
							for (int u = 0; u < NQ; ++u) {
								const int c0 = 4 * lane + 256 * u;
								if (c0 < C * G) {
									int sh = 0;
									if (c0 < ql) (void)pk_q4_offset(left ? -1 : 1, ql, c0, sh);
									*(uint32_t *)(qw + c0) = c0 < ql ? pk_q4_decode(raw_q[u], sh, left ? -1 : 1, ql, c0) : 0x07070707u;
								}
							}
						} else {                                         // array jobs and queries of under four columns: fetched here and now
							job_src_t sg;
							if (A.desc) { sg.qp = A.reads + p_w3 + (left ? ql - 1 : 0); sg.qstep = left ? -1 : 1; sg.tp = nullptr; sg.t0 = (long long)(((unsigned long long)p_t0h << 32) | p_t0l) + (left ? tln - 1 : 0); sg.tdir = left ? -1 : 1; }
							else { sg.qp = A.q + p_w3; sg.qstep = 1; sg.tp = A.t + p_w5; sg.t0 = 0; sg.tdir = 1; }
							for (int k0 = 8 * lane; k0 < tln; k0 += 512) {
								uint32_t lo, hi;
								pk_t8(A, sg, k0, tln, lo, hi);
								*(uint32_t *)(tg + k0) = lo;
								if (k0 + 4 < tln) *(uint32_t *)(tg + k0 + 4) = hi;
							}
							for (int c0 = 4 * lane; c0 < C * G; c0 += 256) {
								uint32_t w = 0;
	#pragma unroll
								for (int u = 0; u < 4; ++u) { const int j = c0 + u; w |= (uint32_t)(j < ql ? min(ext_q_at(A, sg, j), 4) : 7) << (8 * u); }      // 4 = N, 7 = pad
								*(uint32_t *)(qw + c0) = w;
							}
						}
						__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
						pend = false; pre = true;
					}
				}
				if (!waiting || !pre) break;
				{   // the staged job goes to the first group that asks; its row becomes the next spare
					const int gl = (int)__builtin_ctzll(reqs);           // first lane of the group (wave-uniform)
					const int old = __builtin_amdgcn_readlane(trow, gl);
					if ((lane & ~(G - 1)) == gl) {
						const uint4 r0 = rc_wave[2 * p_slot];
						const uint32_t id = r0.x, p_qt = r0.y; const int h0 = (int)r0.z;
						qlen = (int)(p_qt & 0xFFFFu); tlen = (int)(p_qt >> 16);
						if (g0) hrow[PP] = id;
						trow = spare;
	#pragma unroll
						for (int p = 0; p < P; ++p) sel[p] = 0x0C000C00u | (uint32_t)qw[j0 + p] | ((uint32_t)qw[j0 + P + p] << 16);
						{
							const int jq = qlen > 0 ? qlen - 1 : 0, lq = jq / C, r = jq % C;       // lane, chain and pair of column qlen-1
							goff = 2 * (hgrp + (r >= P ? r - P : r)) + (r >= P ? 1 : 0); owner = l == lq;
						}
						S.end = qlen; S.mx = h0; S.max_i = -1; S.max_j = -1; S.max_ie = -1; S.gscore = -1; S.max_off = 0;
						i = 0; hfc = h0; hnx = max(0, h0 - oe_del);
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
						}
						have = tlen > 0; alive = have;
						if (tlen == 0 && g0) {                           // no target rows (a window clipped away): the answer is (h0, 0, 0)
							int32_t *o = A.out + 3 * (size_t)id;
							o[0] = h0; o[1] = 0; o[2] = 0;
							if (A.raw) { int32_t *r = A.raw + 6 * (size_t)id; r[0] = h0; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = -1; r[5] = 0; }
							if (A.stats) atomicAdd(A.stats + 2, 1ull);
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // qw has been read before the next job is staged into it
					spare = old;
					pre = false;
					reqs &= reqs - 1;
				}
			}
			if (pend) ++age;
			if (!__any(alive) && !pre && !pend && qn == qe && !more_g) break;
		}
		// (no `continue` past the row when no group runs: a second path around the row body makes the compiler keep two copies of all
		// loop-carried pairs; the idle row is harmless and rare)
		++wave_rows;
		const int ti = (int)t_wave[trow + i];
		const bool run = alive;
#ifdef PK_STATS            // (a -DPK_STATS build + BMH_EXT_STATS) wave-rows in which every running alignment has reached the query end: candidates of a row without end
		if (A.stats) {          // masks.  Not in the shipped build: the row loop carries no test that is not a DP row's
			const bool all_at_end = !__any(run && S.end != qlen), any_run = __any(run);
			if (lane == 0 && any_run) { atomicAdd(A.stats + 4, 1ull); if (all_at_end) atomicAdd(A.stats + 5, 1ull); }
			// groups without a running alignment in this row: while the class still has jobs for the wave (a draw that waited for its bases) / in the wave's drain
			const int idle = (int)__builtin_popcountll(__ballot(!run && g0));
			const bool work_left = pre || pend || qn != qe || more_g;
			if (lane == 0 && idle) atomicAdd(A.stats + (work_left ? 6 : 7), (unsigned long long)idle);
		}
#endif
		alive = pk_row<G, P, SAME_OE>(K, A.zdrop, H, E, NZ, sel, ti, run, hfc, hnx, i, qlen, j0, eCl, g0, em_tab, hrow, owner, (const uint16_t *)h_wave, goff, S, alive, (wave_rows & PK_BOUND_MASK) == 0, tlen - 1 - i, A.raw == nullptr, A.end_bonus);
		if (run) { ++i; hfc = hnx; hnx = max(0, hnx - A.e_del); }
		alive = alive && i < tlen;
	}
	if (A.stats && lane == 0) atomicAdd(A.stats + 3, (unsigned long long)wave_rows);
}

// One kernel per class (the form the side streams of extend_launch run when the persistent kernel is off): the end masks are the block's.
template <int G, int P, bool SAME_OE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PK_WAVES2(G, P) ? 2 : PK_WAVES4(G, P) ? 4 : 3))) extpk_kernel(ext_args_t A)
{
	wtrace_scope_t wt_(WT_EXT_PK, (uint32_t)(G << 8 | P));
	using L = pk_lds_t<G, P>;
	__shared__ __attribute__((aligned(16))) uint8_t t_lds[4][L::T_BYTES];
	__shared__ __attribute__((aligned(16))) uint8_t q_lds[4][L::Q_BYTES];      // the query codes of the job a wave is staging
	__shared__ __attribute__((aligned(16))) uint32_t em_tab[L::EM_DWORDS];
	__shared__ __attribute__((aligned(16))) uint32_t h_lds[4][L::H_DWORDS];
	__shared__ __attribute__((aligned(16))) uint4 rc_lds[4][2 * PK_CHUNK];
	pk_em_init<P>(em_tab, threadIdx.x, 256);
	__syncthreads();
	const int w = threadIdx.x >> 6;
	extpk_body<G, P, SAME_OE>(A, t_lds[w], q_lds[w], em_tab, h_lds[w], rc_lds[w]);
}

// The persistent form: ONE launch for all packed classes of a pass.  A wave works the classes off one after the other, widest
// first, each with the class's own body (registers: the largest class's; LDS: a slice per wave carved per class, so that no
// block-wide barrier ties a wave to its block mates).  Its blocks stay resident from the first job of the pass to the last: the
// extension holds a FIXED share of every SIMD's registers (grid = blocks per CU x CUs, chosen by the host) instead of taking and
// giving up the whole chip class by class, and the gather-bound kernels of the other batch in flight run in what it leaves.
template <int G, int P> constexpr int pk_wave_lds_max(int m) { return pk_lds_t<G, P>::WAVE_BYTES > m ? pk_lds_t<G, P>::WAVE_BYTES : m; }
constexpr int PK_PERSIST_WAVE_LDS =
	pk_wave_lds_max<16, 9>(pk_wave_lds_max<8, 16>(pk_wave_lds_max<8, 14>(pk_wave_lds_max<8, 12>(pk_wave_lds_max<8, 10>(pk_wave_lds_max<8, 9>(
	pk_wave_lds_max<4, 17>(pk_wave_lds_max<4, 16>(pk_wave_lds_max<4, 14>(pk_wave_lds_max<4, 12>(pk_wave_lds_max<4, 10>(pk_wave_lds_max<4, 8>(
	pk_wave_lds_max<4, 6>(pk_wave_lds_max<4, 4>(0))))))))))))));
template <int G, int P, bool SAME_OE>
__device__ __forceinline__ void pk_persist_class(const ext_args_t &A0, uint8_t *base)
{
	using L = pk_lds_t<G, P>;
	ext_args_t a = A0;
	constexpr int cls = ext_pk_cls_of(G, P);
	a.count = A0.count + 2 * cls;
	a.ctr = A0.ctr + cls;
	if (a.count[0] == 0) return;                               // (wave-uniform)
	uint8_t *t_wave = base, *qw = base + L::T_BYTES;
	uint32_t *em_tab = (uint32_t *)(qw + L::Q_BYTES), *h_wave = em_tab + L::EM_DWORDS;
	uint4 *rc_wave = (uint4 *)(h_wave + L::H_DWORDS);
	pk_em_init<P>(em_tab, threadIdx.x & 63, 64);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // (LDS operations of a wave execute in order; the fences hold the compiler to it)
	extpk_body<G, P, SAME_OE>(a, t_wave, qw, em_tab, h_wave, rc_wave);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}
template <bool SAME_OE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) extpk_persist_kernel(ext_args_t A)
{
	wtrace_scope_t wt_(WT_EXT_PERSIST);
	__shared__ __attribute__((aligned(16))) uint8_t lds[4][PK_PERSIST_WAVE_LDS];
	uint8_t *base = lds[threadIdx.x >> 6];
	// longest rows first: what is left at the end of the pass are the short jobs of the narrow classes
	pk_persist_class<16, 9, SAME_OE>(A, base);
	pk_persist_class<8, 16, SAME_OE>(A, base); pk_persist_class<8, 14, SAME_OE>(A, base); pk_persist_class<8, 12, SAME_OE>(A, base);
	pk_persist_class<8, 10, SAME_OE>(A, base); pk_persist_class<8, 9, SAME_OE>(A, base);
	pk_persist_class<4, 17, SAME_OE>(A, base); pk_persist_class<4, 16, SAME_OE>(A, base); pk_persist_class<4, 14, SAME_OE>(A, base);
	pk_persist_class<4, 12, SAME_OE>(A, base); pk_persist_class<4, 10, SAME_OE>(A, base); pk_persist_class<4, 8, SAME_OE>(A, base);
	pk_persist_class<4, 6, SAME_OE>(A, base); pk_persist_class<4, 4, SAME_OE>(A, base);
}
