// Device job builder: the stage between the two kernels of the hot path, on
// the GPU so that a batch goes reads -> seeds -> extension jobs -> regions without leaving HBM.
//   chain_lane_kernel / chain_wave_kernel   per-read chaining core (chain_core.h): mem_chain, mem_chain_flt,
//                                           mem_chain2aln of /root/reference/src/bwamem.c:404-477, 487-559, 1170-1479
//   emit_kernel                             regions and job descriptors in batch order (per read, per region,
//                                           LEFT then RIGHT -- the order of bmh_build_jobs)
//   materialize_kernel                      on-device reference fetch: query bases from the reads, target bases from
//                                           the 2-bit pac incl. the reverse strand and the LEFT reversal
//                                           (bns_get_seq src/bntseq.c:558-580; bwamem.c:1328-1334)
//   merge_kernel                            extension results -> regions (src/bwamem.c:2297-2303)
// The batch it produces is byte-identical to bmh_build_jobs' (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include "bmh_internal.h"
#include "wtrace.h"
#include "chain_core.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("%s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)

struct ch_outreg_t { int64_t seed_rbeg; int32_t seed_qbeg, seedlen0, job0, job1; uint32_t read; int32_t l_query; };   // 32 B

struct chain_args_t {
	ch_ctx_t x;
	uint32_t n_reads, heavy_thresh;
	uint32_t *heavy_list; uint32_t *heavy_n;
	uint32_t *need;               // [n_reads] seed occurrences the chaining core will sample
	uint32_t *light_list, *light_n;   // the reads of the lane kernel by need bin: [CH_N_BINS][n_reads], [CH_N_BINS]
	unsigned long long *need_sum;     // sum of need over the reads beyond heavy_thresh
};

// What a read costs the chaining core is the number of seed occurrences mem_chain SAMPLES (at most max_occ per SMEM,
// src/bwamem.c:430-436), not the number located: on an hg38-like genome 0.25 % of the reads carry SMEMs with thousands of
// occurrences (90 % of all located seeds) of which 500 each are used.  need[r] = that number; it sizes the read's scratch.
// Reads that need more than heavy_thresh entries go to the wave kernels, one list per LDS size class.
#ifndef CH_HEAVY_DEFAULT
#define CH_HEAVY_DEFAULT 8u
#endif
#define CH_N_CLASSES 11
#define CH_HYBRID_CLASS 6        // classes from here on keep only the arrays of the sequential phases in LDS
#define CH_N_SUB 3               // classes 0 .. 2 (reads of heavy_thresh+1 .. 16, .. 32, .. 64 entries: 115 000 of the 124 000 seed-rich reads of a million at hg38 scale):
                                 // FOUR READS PER WAVE, a 16-lane row each with its scratch in LDS (chain_sub_kernel, round 6); knob CHAIN_SUB, bit c clear: class c in its
                                 // round-5 form -- a lane per read over the class's list with global scratch (classes 0, 1), a wave per read (class 2)
#ifndef CH_SUB_CAP0
#define CH_SUB_CAP0 16u          // (even numbers: the arrays of a row's slice start on 8-byte boundaries)
#define CH_SUB_CAP1 32u
#endif
__device__ __forceinline__ int ch_class_of(uint32_t need) { return need <= CH_SUB_CAP0 ? 0 : need <= CH_SUB_CAP1 ? 1 : need <= 64u ? 2 : need <= 128u ? 3 : need <= 256u ? 4 : need <= 384u ? 5 : need <= 512u ? 6 : need <= 620u ? 7 : need <= 1250u ? 8 : need <= 1860u ? 9 : 10; }
// LDS entries per class; the hybrid classes (512 entries and up) keep only the arrays of the sequential phases in LDS (seeds, chains,
// the sorted chain index, the sort keys: 84 bytes per entry instead of 124) and the rest in the read's slice of the global scratch:
// these classes are LDS-bound -- three reads of 620 entries per CU instead of two was worth 1 ms of a 10 ms stage; the last class
// (a read that samples more than 1860 occurrences: four SMEMs of 465+ hits each) works in global memory altogether
// (a read that samples one SMEM of 500+ occurrences plus a few more seeds needs 500-600 entries: on the hg38-like genome most of
// the reads beyond 256 entries sit there, hence the 512 and 620 classes)
static const uint32_t CH_CLASS_CAP[CH_N_CLASSES] = {CH_SUB_CAP0, CH_SUB_CAP1, 64u, 128u, 256u, 384u, 512u, 620u, 1250u, 1860u, 0u};
#ifndef CH_WAVE_GRIDS
#define CH_WAVE_GRIDS 4096u, 2048u, 1024u, 1024u, 768u, 512u, 256u, 256u
#endif
static const uint32_t CH_CLASS_GRID[CH_N_CLASSES] = {0u, 0u, 8192u, CH_WAVE_GRIDS};
// blocks (one wave, four reads at a time) of chain_sub_kernel per class: about what the chip holds at once (LDS: 4 x cap x 124 bytes a block), the reads dealt round robin
#ifndef CH_SUB_GRIDS
#define CH_SUB_GRIDS 4096u, 3584u, 1792u     // (2560 and 1280 for the last two until the compact records: 24.2 against 24.5 ms per step)
#endif
static const uint32_t CH_SUB_GRID[CH_N_SUB] = {CH_SUB_GRIDS};

// need, the wave / lane-list class of a seed-rich read -- and, for the reads of the lane kernel, a BIN by need (<= 2, <= 4, <= 8,
// the rest): the lane kernel walks the bins' compacted lists, so the 64 reads of a wave cost about the same.  In read order every
// wave had a handful of 9-16 entry reads and took as long as those: 15 600 waves of ~1 ms each for reads most of which need 0.1 ms.
// (appends aggregated per block: one atomic per bin and block on the global counters)
#define CH_N_BINS 4
__device__ __forceinline__ int ch_bin_of(uint32_t need) { return need <= 2u ? 0 : need <= 4u ? 1 : need <= 8u ? 2 : 3; }
__global__ void __launch_bounds__(256) chain_classify_kernel(chain_args_t A)
{
	wtrace_scope_t wt_(WT_CHAIN_CLASSIFY);
	// (bins CH_N_BINS.. of the block counters: the wave / lane-list classes -- 50 000 appends to two counters, one atomic each, were
	// 0.4 ms of same-address atomics)
	__shared__ uint32_t l_cnt[CH_N_BINS + CH_N_CLASSES], l_base[CH_N_BINS + CH_N_CLASSES];
	__shared__ uint32_t l_maxlen;
	__shared__ uint32_t l_need;          // sampled occurrences of the block's wave / lane-list reads (the bound of their regions: need_sum)
	if (threadIdx.x < CH_N_BINS + CH_N_CLASSES) l_cnt[threadIdx.x] = 0;
	if (threadIdx.x == 0) { l_need = 0; l_maxlen = 0; }
	__syncthreads();
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	int bin = -1; uint32_t my = 0;
	if (r < A.n_reads) {
		const uint32_t n = A.x.n_ref[r];
		uint32_t need = n;
		if (n > (uint32_t)A.x.o.max_occ) {          // only then can a group exceed max_occ
			const uint32_t *sc = A.x.score + A.x.prefix[r];
			need = 0;
			for (uint32_t i = 0; i < n;) {
				const uint32_t cnt = sc[i];
				if (cnt == 0) break;
				need += cnt < (uint32_t)A.x.o.max_occ ? cnt : (uint32_t)A.x.o.max_occ;
				i += cnt;
			}
		}
		A.need[r] = need;
		if (need > A.heavy_thresh) {
			// (the four-per-wave classes stage a read's LOCATED seeds in its scratch: a read that samples few of many -- a small -c -- goes by that count,
			// to a wave class if need be)
			int c = ch_class_of(need);
			if (c < CH_N_SUB && n > need) { const int c2 = ch_class_of(n); c = c2 < CH_N_SUB ? c2 : CH_N_SUB; }
			bin = CH_N_BINS + c;
		} else bin = ch_bin_of(need);
		my = atomicAdd(&l_cnt[bin], 1u);
		if (need > A.heavy_thresh) atomicAdd(&l_need, need);
		atomicMax(&l_maxlen, A.x.read_lens[r]);
	}
	__syncthreads();
	if (threadIdx.x == 0 && l_need) atomicAdd(A.need_sum, (unsigned long long)l_need);
	if (threadIdx.x == 0) atomicMax(A.light_n + CH_N_BINS, l_maxlen);      // the longest read of the batch: a bound of every query length of its jobs
	if (threadIdx.x < CH_N_BINS + CH_N_CLASSES && l_cnt[threadIdx.x])
		l_base[threadIdx.x] = atomicAdd(threadIdx.x < CH_N_BINS ? A.light_n + threadIdx.x : A.heavy_n + (threadIdx.x - CH_N_BINS), l_cnt[threadIdx.x]);
	__syncthreads();
	if (bin >= CH_N_BINS) A.heavy_list[(size_t)(bin - CH_N_BINS) * A.n_reads + l_base[bin] + my] = r;
	else if (bin >= 0) A.light_list[(size_t)bin * A.n_reads + l_base[bin] + my] = r;
}

// A lane's scratch as PRIVATE memory (CAP entries of every array): the hardware interleaves private memory dword by dword over the
// lanes of a wave, so the 64 lanes' accesses to entry i of an array are consecutive addresses -- 256 bytes in one to four cache
// lines -- where the per-read slices of the global scratch are 64 lines per wave instruction (round 2 measured 3.4 GB of the chain
// family's 5.7 GB of HBM traffic there, for arrays nobody reads afterwards).  Only the regions (x.regs) leave the kernel.
// Scratch records of the cooperative kernels and of the lane kernels' private scratch: compact (chain_core.h: ch_compact_ty, 82 / 58 bytes per entry) unless the
// launch carries the reference's seed filter; knob -DCH_WIDE_SCRATCH: the wide records everywhere (124 / 84 bytes: rounds 2-5).
#ifdef CH_WIDE_SCRATCH
template <bool FLT> struct ch_coop_ty { typedef ch_wide_ty ty; };
#else
template <bool FLT> struct ch_coop_ty { typedef ch_compact_ty ty; };
template <> struct ch_coop_ty<true> { typedef ch_wide_ty ty; };
#endif
template <int CAP, class TY> struct ch_private_t {
	typename TY::seed_t S[CAP]; typename TY::chain_t CH[CAP]; typename TY::est_t E[CAP]; int64_t opos[CAP]; uint64_t srt[CAP]; typename TY::idx_t order[CAP], klist[CAP], cidx[CAP];
	__device__ __forceinline__ ch_scr<TY> scr() { ch_scr<TY> L; L.S = S; L.CH = CH; L.E = E; L.opos = opos; L.srt = srt; L.order = order; L.klist = klist; L.cidx = cidx; return L; }
};

// one read per lane, the bins from the costliest down as one sequence (CAP > 0: private scratch of CAP entries -- the caller makes sure
// no read of the launch needs more; CAP == 0: the read's slice of the global scratch)
template <bool FLT, int CAP>
__global__ void __launch_bounds__(256) chain_lane_kernel(chain_args_t A, const uint32_t skip_bins)
{
	wtrace_scope_t wt_(WT_CHAIN_LANE);
	uint32_t t = blockIdx.x * 256u + threadIdx.x;
	int bin = CH_N_BINS - 1;
	for (; bin >= 0; --bin) { const uint32_t c = (skip_bins >> bin & 1u) ? 0u : A.light_n[bin]; if (t < c) break; t -= c; }
	if (bin < 0) return;
	const uint32_t r = A.light_list[(size_t)bin * A.n_reads + t];
	if (CAP > 0 && A.need[r] > (uint32_t)CAP) { *A.x.err = 3; return; }      // (a launch whose bins do not fit its private arrays: an internal error, never an overrun)
	if (CAP == 8 && skip_bins == 0u) {
		// ONE launch, the private arrays sized by the read's bin (the waves are bin-pure but for the seams: a wave of 2-entry reads dirties a quarter of the
		// scratch lines an 8-entry frame would): private memory is written back like any other, and with eight entries for every lane this kernel moved
		// 0.98 GB out and 0.83 GB in per million reads -- more than all the other chaining kernels together -- for 44 MB of regions
		typedef typename ch_coop_ty<FLT>::ty TY;
		if (bin == 0) { ch_private_t<2, TY> P; chain_core::chain_read<false, false, FLT, 64, false, 2>(A.x, r, P.scr()); }
		else if (bin == 1) { ch_private_t<4, TY> P; chain_core::chain_read<false, false, FLT, 64, false, 4>(A.x, r, P.scr()); }
		else { ch_private_t<8, TY> P; chain_core::chain_read<false, false, FLT, 64, false, 8>(A.x, r, P.scr()); }
	} else if (CAP > 0) {
		ch_private_t<(CAP > 0 ? CAP : 1), typename ch_coop_ty<FLT>::ty> P;
		chain_core::chain_read<false, false, FLT>(A.x, r, P.scr());
	} else chain_core::chain_read<false, false, FLT>(A.x, r, chain_core::global_scratch(A.x, r));
}

#define CH_LDS_BYTES_PER_ENTRY (8 + 8 + sizeof(ch_est_t) + sizeof(ch_chain_t) + sizeof(ch_seed_t) + 4 + 4 + 4)
// The same with the lane's scratch in LDS (bins of at most CAP entries a read): the lane form's global scratch is one cache line
// per lane and access -- 64 lines per wave instruction, 3.4 GB of HBM traffic per million reads for arrays nobody reads afterwards.
// A lane's slice holds its eight arrays back to back; its stride is an odd number of 8-byte words, so that the 64 lanes of an
// access spread over the banks (2-way at worst for 4-byte accesses).  Only the regions (x.regs) leave the kernel.
#define CH_LANE_LDS_SLICE(cap) ((((uint32_t)(cap) * (uint32_t)CH_LDS_BYTES_PER_ENTRY + 7u) / 8u | 1u) * 8u)
template <int CAP>
__global__ void __launch_bounds__(64) chain_lane_lds_kernel(chain_args_t A, int bin)
{
	extern __shared__ __align__(16) uint8_t ch_lds[];
	const uint32_t n = A.light_n[bin];
	ch_scr_t L;
	{
		uint8_t *p = ch_lds + (size_t)threadIdx.x * CH_LANE_LDS_SLICE(CAP);
		L.opos = (int64_t *)p; p += 8 * CAP;
		L.srt = (uint64_t *)p; p += 8 * CAP;
		L.CH = (ch_chain_t *)p; p += sizeof(ch_chain_t) * CAP;
		L.S = (ch_seed_t *)p; p += sizeof(ch_seed_t) * CAP;
		L.E = (ch_est_t *)p; p += sizeof(ch_est_t) * CAP;
		L.order = (uint32_t *)p; p += 4 * CAP;
		L.klist = (uint32_t *)p; p += 4 * CAP;
		L.cidx = (uint32_t *)p;
	}
	for (uint32_t t = blockIdx.x * 64u + threadIdx.x; t < n; t += gridDim.x * 64u)
		chain_core::chain_read<false>(A.x, A.light_list[(size_t)bin * A.n_reads + t], L);
}

// Lane form over the list of class 0.  The reads just above the lane kernel's threshold (9 .. 32 entries: most of the "heavy"
// reads) cost a lane a few milliseconds of dependent steps, too long for the lane kernel, whose waves would all wait for their one
// such read -- but compacted into their own list they are 64 reads of similar cost per wave, a few hundred waves that leave the
// CUs (and all of the LDS) to the wave kernels of the larger classes.  One wave per read, they took a third of the stage.
template <bool FLT, int CAP>
__global__ void __launch_bounds__(256) chain_lane_list_kernel(chain_args_t A, uint32_t cls, const uint32_t lanes)
{
	wtrace_scope_t wt_(WT_CHAIN_LIST, cls);
	// `lanes` reads per wave (knob CHAIN_LIST_LANES; 64 = every lane): a wave of this kernel executes the union of its lanes' paths through
	// the sequential chaining core, so fewer reads per wave are shorter waves -- and there are wave slots to spare (100 000 reads are 1 563 full waves)
	const uint32_t lane = threadIdx.x & 63u;
	if (lane >= lanes) return;
	const uint32_t i = ((blockIdx.x * 256u + threadIdx.x) >> 6) * lanes + lane;
	if (i >= A.heavy_n[cls]) return;
	const uint32_t r = A.heavy_list[(size_t)cls * A.n_reads + i];
	if (CAP > 0 && A.need[r] > (uint32_t)CAP) { *A.x.err = 3; return; }
	if (CAP > 0) {
		ch_private_t<(CAP > 0 ? CAP : 1), typename ch_coop_ty<FLT>::ty> P;
		chain_core::chain_read<false, false, FLT>(A.x, r, P.scr());
	} else chain_core::chain_read<false, false, FLT>(A.x, r, chain_core::global_scratch(A.x, r));
}

// one wave per heavy read of one size class (list filled by chain_classify_kernel), on a side stream beside chain_lane_kernel.
// The read's scratch lives in LDS (lds_cap entries of CH_LDS_BYTES_PER_ENTRY bytes each; lds_cap == 0: the read's slice of the
// global scratch): the wave form is a chain of dependent accesses, so their latency is its run time -- and the LDS a block asks
// for decides how many of these waves a CU runs side by side (64 entries: 20, 128: 10, 256: 5, 512: 2, 1250: 1), which is why the classes exist.
#define CH_LDS_BYTES_PER_ENTRY_HYBRID (8 + 8 + sizeof(ch_chain_t) + sizeof(ch_seed_t) + 4)
#define CH_LDS_CONTIGS 256         // contig tables up to this size are copied into LDS (3 KB)
#ifndef CH_B_SIDE
#define CH_B_SIDE 1                // bmh_chain_extend_merge: pass B's emit + extension on the second stream, beside pass A's (knob CHAIN_B_SIDE)
#endif
#ifndef CH_LIST_LANES
#define CH_LIST_LANES 64           // reads per wave of chain_lane_list_kernel (knob CHAIN_LIST_LANES)
#endif
#ifndef CH_WAVE_ATTR
#define CH_WAVE_ATTR
#endif
// bytes of LDS per entry (hybrid: everything but the region estimates, which only the last phase reads -- until round 6 the kept list and the seed list lay in
// global memory too, and every kept chain cost the wave a round trip there: 0.2 ms of a 600-entry read's 0.67)
template <class TY> constexpr size_t ch_lds_entry_bytes(bool hybrid)
{
	// (wide records: the hybrid classes keep the two lists in global memory as before -- 1860 entries of 92 bytes would not fit the LDS)
	return 8 + 8 + sizeof(typename TY::seed_t) + sizeof(typename TY::chain_t) + (hybrid && sizeof(typename TY::idx_t) == 4 ? 1 : 3) * sizeof(typename TY::idx_t) + (hybrid ? 0 : sizeof(typename TY::est_t));
}
// a slice of LDS as a read's scratch (cap even; 8-byte arrays first, every array aligned for either record width); hybrid: E is set by the caller
template <class TY> __device__ __forceinline__ ch_scr<TY> ch_carve(uint8_t *p, const size_t cap, const bool hybrid)
{
	ch_scr<TY> L;
	L.opos = (int64_t *)p; p += 8 * cap;
	L.srt = (uint64_t *)p; p += 8 * cap;
	L.S = (typename TY::seed_t *)p; p += sizeof(typename TY::seed_t) * cap;
	L.E = (typename TY::est_t *)p; if (!hybrid) p += sizeof(typename TY::est_t) * cap;
	L.CH = (typename TY::chain_t *)p; p += sizeof(typename TY::chain_t) * cap;
	L.order = (typename TY::idx_t *)p; p += sizeof(typename TY::idx_t) * cap;
	L.klist = (typename TY::idx_t *)p; p += sizeof(typename TY::idx_t) * cap;
	L.cidx = (typename TY::idx_t *)p;
	return L;
}
// (CTG_LDS: entries of the contig table's copy in LDS -- 0: none, the table stays in global memory; 64: 768 bytes; CH_LDS_CONTIGS: 3 KB, which cost the
// 512-entry class its fifth block per CU)
template <int CTG_LDS, bool FLT>
__global__ void __launch_bounds__(64) CH_WAVE_ATTR chain_wave_kernel(chain_args_t A, uint32_t cls, uint32_t lds_cap, int hybrid, int prio)
{
	wtrace_scope_t wt_(WT_CHAIN_WAVE, cls);
	if (prio) __builtin_amdgcn_s_setprio(3);             // (knob CHAIN_WAVE_PRIO: the few long-lived waves of the largest classes ahead of their SIMD's other waves)
	extern __shared__ __align__(16) uint8_t ch_lds[];
	const uint32_t nh = A.heavy_n[cls];
	const uint32_t *list = A.heavy_list + (size_t)cls * A.n_reads;
	typedef typename ch_coop_ty<FLT>::ty TY;
	const ch_scr<TY> L = ch_carve<TY>(ch_lds, (size_t)lds_cap, hybrid != 0);
	// the contig table is looked up twice per seed occurrence (bns_intv2rid): a copy in LDS instead of dependent global loads
	__shared__ int64_t ctg_off_l[CTG_LDS > 0 ? CTG_LDS : 1];
	__shared__ int32_t ctg_len_l[CTG_LDS > 0 ? CTG_LDS : 1];
	if (CTG_LDS > 0) {                  // (a template parameter: the table pointer must have one address space per instantiation)
		for (int c = (int)threadIdx.x; c < A.x.n_contigs; c += 64) { ctg_off_l[c] = A.x.ctg_off[c]; ctg_len_l[c] = A.x.ctg_len[c]; }
		A.x.ctg_off = ctg_off_l; A.x.ctg_len = ctg_len_l;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0);
	}
	for (uint32_t i = blockIdx.x; i < nh; i += gridDim.x) {
		const uint32_t r = list[i];
		// three call sites so that every pointer of a call has ONE address space the compiler can see (a pointer that may be
		// LDS or global becomes a flat access, several times the latency of ds_read on the LDS side)
		if (lds_cap && !hybrid) chain_core::chain_read<true, true, FLT>(A.x, r, L);
		else if (lds_cap) {
			// (the region estimates in the read's slice of the global scratch, as records of this form's width: the slices are sized for the wide ones)
			const ch_scr_t G = chain_core::global_scratch(A.x, r);
			ch_scr<TY> H = L; H.E = (typename TY::est_t *)G.E;
			if (sizeof(typename TY::idx_t) == 4) { H.klist = (typename TY::idx_t *)G.klist; H.cidx = (typename TY::idx_t *)G.cidx; }
			chain_core::chain_read<true, false, FLT, 64, false, 1>(A.x, r, H);
		} else chain_core::chain_read<true, false, FLT>(A.x, r, chain_core::global_scratch(A.x, r));
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_s_waitcnt(0);                           // the read's output stores before the scratch is reused
	}
}

// FOUR reads per wave, one per 16-lane row, for the classes of at most 64 entries: the cooperative core with W = 16 (chain_core.h: ch_grp<16>) -- every
// ballot, broadcast and scan stays inside the row, the rows of a wave diverge freely --, each row's scratch in its own slice of the block's LDS (cap entries of
// CH_LDS_BYTES_PER_ENTRY bytes), global memory only for the read's seeds and its regions.  These reads used to take a lane each with their scratch in global memory
// (2.5 ms of dependent round trips per lane, 3 x the stage's algorithmic traffic) or a whole wave for loops that are sixteen wide at most.
// CTGN > 0: the contig table (at most CTGN sequences) is copied into LDS.
#define CH_SUB_LDS_CONTIGS 64
#ifndef CH_SUB_ATTR
#define CH_SUB_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))      // 128 registers (148 unconstrained, four spilled): four waves per SIMD; the class of 16 entries alone 0.67 -> 0.47 ms
#endif
template <int CTGN, bool FLT>
__global__ void __launch_bounds__(64) CH_SUB_ATTR chain_sub_kernel(chain_args_t A, uint32_t cls, uint32_t cap)
{
	wtrace_scope_t wt_(WT_CHAIN_SUB, cls);
	extern __shared__ __align__(16) uint8_t ch_lds[];
	const uint32_t nh = A.heavy_n[cls];
	if (blockIdx.x * 4u >= nh) return;
	const uint32_t *list = A.heavy_list + (size_t)cls * A.n_reads;
	const uint32_t g = threadIdx.x >> 4;
	typedef typename ch_coop_ty<FLT>::ty TY;
	const ch_scr<TY> L = ch_carve<TY>(ch_lds + (size_t)g * cap * ch_lds_entry_bytes<TY>(false), (size_t)cap, false);   // (cap is a multiple of four: a slice starts on an 8-byte boundary)
	__shared__ int64_t ctg_off_l[CTGN > 0 ? CTGN : 1];
	__shared__ int32_t ctg_len_l[CTGN > 0 ? CTGN : 1];
	if (CTGN > 0) {
		for (int c = (int)threadIdx.x; c < A.x.n_contigs; c += 64) { ctg_off_l[c] = A.x.ctg_off[c]; ctg_len_l[c] = A.x.ctg_len[c]; }
		A.x.ctg_off = ctg_off_l; A.x.ctg_len = ctg_len_l;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0);
	}
	for (uint32_t i = blockIdx.x * 4u + g; i < nh; i += gridDim.x * 4u) {
		// (the read's seed arrays are staged in its scratch: chain_classify_kernel sends no read here that has more located seeds than the class has entries)
		chain_core::chain_read<true, true, FLT, 16, true>(A.x, list[i], L);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                    // (LDS operations of a wave execute in order; the region stores hold their data already)
	}
}

struct emit_args_t {
	const ch_reg_t *regs; const uint32_t *prefix, *regs_per_read, *reg_off, *job_off, *read_offs, *read_lens;
	uint32_t n_reads;
	uint32_t reg_base, job_base;      // added to reg_off / job_off (second pass of bmh_chain_extend_merge)
	const uint32_t *need; uint32_t thresh; int pass;      // need != nullptr: only the reads of one pass count (pass 1: need > thresh, pass 0: the others)
	ch_outreg_t *outregs;
	uint32_t *qlen, *tlen, *h0, *job_read, *job_reg, *job_side, *jq_src; int64_t *jt0;
};

// (reads with more regions than this are emitted by a wave each, emit_wave_kernel: a repetitive read has up to max_occ regions per
// SMEM, and one lane writing a thousand job descriptors one after the other was what the second extension pass waited for)
#define CH_EMIT_LANE_MAX 32
__global__ void __launch_bounds__(256) emit_kernel(emit_args_t A)
{
	wtrace_scope_t wt_(WT_CHAIN_EMIT);
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r >= A.n_reads) return;
	const uint32_t nr = (A.need && (A.need[r] > A.thresh) != (A.pass == 1)) ? 0u : A.regs_per_read[r];
	if (nr == 0 || nr > CH_EMIT_LANE_MAX) return;
	const ch_reg_t *R = A.regs + A.prefix[r];
	uint32_t g = A.reg_base + A.reg_off[r], j = A.job_base + A.job_off[r];
	const uint32_t roff = A.read_offs[r]; const int lq_ = (int)A.read_lens[r];
	for (uint32_t i = 0; i < nr; ++i, ++g) {
		const ch_reg_t a = R[i];
		ch_outreg_t o; o.seed_rbeg = a.seed_rbeg; o.seed_qbeg = a.seed_qbeg; o.seedlen0 = a.seedlen0; o.job0 = o.job1 = -1; o.read = r; o.l_query = lq_;
		if (a.seed_qbeg > 0) {
			o.job0 = (int32_t)j;
			A.qlen[j] = (uint32_t)a.seed_qbeg; A.tlen[j] = (uint32_t)a.lr; A.h0[j] = (uint32_t)a.seedlen0;
			A.job_read[j] = r; A.job_reg[j] = g; A.job_side[j] = 0; A.jq_src[j] = roff; A.jt0[j] = a.rmax0; ++j;
		}
		if (a.rq > 0) {
			o.job1 = (int32_t)j;
			A.qlen[j] = (uint32_t)a.rq; A.tlen[j] = (uint32_t)a.rr; A.h0[j] = (uint32_t)a.seedlen0;
			A.job_read[j] = r; A.job_reg[j] = g; A.job_side[j] = 1; A.jq_src[j] = roff + (uint32_t)(a.seed_qbeg + a.seedlen0);
			A.jt0[j] = a.seed_rbeg + a.seedlen0; ++j;
		}
		A.outregs[g] = o;
	}
}

// the same for the reads of the wave classes' lists that have more than CH_EMIT_LANE_MAX regions (fewer sampled seeds than that
// cannot make that many): one wave per read, one region per lane, the jobs' places by ballot counts
__global__ void __launch_bounds__(256) emit_wave_kernel(emit_args_t A, const uint32_t *__restrict__ lists, const uint32_t *__restrict__ list_n, int cls_lo, int cls_hi)
{
	wtrace_scope_t wt_(WT_CHAIN_EMIT, 1);
	const int lane = threadIdx.x & 63;
	const uint32_t gw = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
	for (int cls = cls_lo; cls <= cls_hi; ++cls) {
		const uint32_t nh = list_n[cls];
		const uint32_t *list = lists + (size_t)cls * A.n_reads;
		for (uint32_t k = gw; k < nh; k += n_waves) {
			const uint32_t r = list[k];
			const uint32_t nr = (A.need && (A.need[r] > A.thresh) != (A.pass == 1)) ? 0u : A.regs_per_read[r];
			if (nr <= CH_EMIT_LANE_MAX) continue;
			const ch_reg_t *R = A.regs + A.prefix[r];
			const uint32_t g0 = A.reg_base + A.reg_off[r];
			uint32_t j0 = A.job_base + A.job_off[r];
			const uint32_t roff = A.read_offs[r]; const int lq_ = (int)A.read_lens[r];
			for (uint32_t b = 0; b < nr; b += 64) {
				const uint32_t i = b + (uint32_t)lane;
				const bool v = i < nr;
				ch_reg_t a = R[v ? i : nr - 1];
				const bool has0 = v && a.seed_qbeg > 0, has1 = v && a.rq > 0;
				const unsigned long long m0 = __ballot(has0), m1 = __ballot(has1);
				uint32_t j = j0 + (uint32_t)__builtin_popcountll(m0 & lt) + (uint32_t)__builtin_popcountll(m1 & lt);
				if (v) {
					const uint32_t g = g0 + i;
					ch_outreg_t o; o.seed_rbeg = a.seed_rbeg; o.seed_qbeg = a.seed_qbeg; o.seedlen0 = a.seedlen0; o.job0 = o.job1 = -1; o.read = r; o.l_query = lq_;
					if (has0) {
						o.job0 = (int32_t)j;
						A.qlen[j] = (uint32_t)a.seed_qbeg; A.tlen[j] = (uint32_t)a.lr; A.h0[j] = (uint32_t)a.seedlen0;
						A.job_read[j] = r; A.job_reg[j] = g; A.job_side[j] = 0; A.jq_src[j] = roff; A.jt0[j] = a.rmax0; ++j;
					}
					if (has1) {
						o.job1 = (int32_t)j;
						A.qlen[j] = (uint32_t)a.rq; A.tlen[j] = (uint32_t)a.rr; A.h0[j] = (uint32_t)a.seedlen0;
						A.job_read[j] = r; A.job_reg[j] = g; A.job_side[j] = 1; A.jq_src[j] = roff + (uint32_t)(a.seed_qbeg + a.seedlen0);
						A.jt0[j] = a.seed_rbeg + a.seedlen0;
					}
					A.outregs[g] = o;
				}
				j0 += (uint32_t)__builtin_popcountll(m0) + (uint32_t)__builtin_popcountll(m1);
			}
		}
	}
}

__device__ __forceinline__ int ch_ascii_code(uint8_t ch)      // nst_nt4_table for the letters a read can hold
{
	ch &= 0xDF;
	return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
}

struct mat_args_t {
	const uint8_t *reads; const uint8_t *pac; int64_t l_pac;
	const uint32_t *qlen, *tlen, *job_side, *jq_src; const int64_t *jt0; const uint64_t *qoff64, *toff64;
	uint32_t n_jobs;
	uint8_t *q, *t; uint32_t *qoff, *toff;
};

// 16 lanes per job
__global__ void __launch_bounds__(256) materialize_kernel(mat_args_t A)
{
	const uint32_t l16 = threadIdx.x & 15u;
	for (uint64_t j = (uint64_t)blockIdx.x * 16u + (threadIdx.x >> 4); j < A.n_jobs; j += (uint64_t)gridDim.x * 16u) {
		const int qn = (int)A.qlen[j], tn = (int)A.tlen[j];
		const bool left = A.job_side[j] == 0;
		const uint64_t qo = A.qoff64[j], to = A.toff64[j];
		if (l16 == 0) { A.qoff[j] = (uint32_t)qo; A.toff[j] = (uint32_t)to; }
		const uint8_t *src = A.reads + A.jq_src[j];
		for (int i = (int)l16; i < qn; i += 16) A.q[qo + i] = (uint8_t)ch_ascii_code(src[left ? qn - 1 - i : i]);
		const int64_t t0 = A.jt0[j];
		for (int i = (int)l16; i < tn; i += 16) {
			const int64_t p = left ? t0 + tn - 1 - i : t0 + i;           // position in fwd . revcomp(fwd)
			const bool rev = p >= A.l_pac;
			const int64_t f = rev ? (A.l_pac << 1) - 1 - p : p;
			const int c = (A.pac[f >> 2] >> ((~f & 3) << 1)) & 3;
			A.t[to + i] = (uint8_t)(rev ? 3 - c : c);
		}
	}
}

// score of a region whose seed spans the whole read (no extension on either side): a->score = s->score (src/bwamem.c:1435), which is the
// seed's length (mem_chain :444) unless the seed filter re-scored it -- to len * a: the window is too long for the local alignment
// (-1 -> s->len * opt->a, :985) or the read matches it end to end (:774-807)
__device__ __forceinline__ int ch_bare_seed_score(const bmh_chain_opt_t &o, int seedlen0, int l_query)
{
	return o.min_chain_weight > 0 && chain_core::seed_filter_applies(o, l_query, nullptr) ? seedlen0 * o.a : seedlen0;
}
__global__ void __launch_bounds__(256) merge_kernel(const ch_outreg_t *__restrict__ regs, uint32_t n_regs, const int32_t *__restrict__ out3, int32_t *__restrict__ regs_out, const bmh_chain_opt_t copt)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_regs) return;
	const ch_outreg_t a = regs[i];
	int score, qb, qe; int64_t rb, re;
	const int sides = (a.job0 >= 0) + (a.job1 >= 0);
	if (sides > 0) {
		int ls = 0, lq = 0, lt = 0, rs = 0, rq = 0, rt = 0;
		if (a.job0 >= 0) { ls = out3[3 * (size_t)a.job0]; lq = out3[3 * (size_t)a.job0 + 1]; lt = out3[3 * (size_t)a.job0 + 2]; }
		if (a.job1 >= 0) { rs = out3[3 * (size_t)a.job1]; rq = out3[3 * (size_t)a.job1 + 1]; rt = out3[3 * (size_t)a.job1 + 2]; }
		score = ls + rs - (sides == 2 ? a.seedlen0 : 0);
		qb = a.seed_qbeg - lq; qe = a.seed_qbeg + a.seedlen0 + rq;
		rb = a.seed_rbeg - lt; re = a.seed_rbeg + a.seedlen0 + rt;
	} else {
		score = ch_bare_seed_score(copt, a.seedlen0, a.l_query); qb = 0; qe = a.l_query; rb = a.seed_rbeg; re = a.seed_rbeg + a.seedlen0;
	}
	int32_t *o = regs_out + 8 * (size_t)i;
	o[0] = (int32_t)a.read; o[1] = score; o[2] = qb; o[3] = qe;
	o[4] = (int32_t)(uint32_t)rb; o[5] = (int32_t)(rb >> 32); o[6] = (int32_t)(uint32_t)re; o[7] = (int32_t)(re >> 32);
}

// ------------------------------------------------------------------------------------------------ workspace / API

struct bmh_chain_ws {
	uint32_t max_reads; uint64_t max_seeds;
	// per-seed scratch
	ch_seed_t *seeds; ch_chain_t *chains; uint32_t *order; int64_t *opos; uint32_t *klist; uint64_t *srt; uint32_t *cidx; ch_reg_t *regs; ch_est_t *est;
	// per read
	uint32_t *regs_per_read, *jobs_per_read, *reg_off, *job_off, *heavy_list, *need; float *frac_rep;
	uint32_t *counters;            // [0..CH_N_CLASSES) heavy_n per size class  [CH_N_CLASSES] err  [12..32) profile stamps  [32..36) reads per need bin of the lane kernel  [36] longest read
	// contigs
	int n_contigs; int64_t *ctg_off; int32_t *ctg_len; uint8_t *ctg_alt;
	// outputs, grown on demand
	uint64_t cap_regs, cap_jobs, cap_q, cap_t;
	ch_outreg_t *outregs;
	uint32_t *qlen, *tlen, *h0, *job_read, *job_reg, *job_side, *jq_src, *qoff, *toff; int64_t *jt0; uint64_t *qoff64, *toff64;
	uint8_t *q, *t;
	void *scan_tmp; size_t scan_tmp_bytes;
	uint64_t n_regs, n_jobs;
	uint32_t *h_pin;               // pinned host words for the small D2H copies
	bmh_chain_opt_t last_opt;      // the options of the last batch (the merge kernels' score of a bare seed depends on them)
	hipStream_t side; hipEvent_t ev_fork, ev_join;   // the wave kernels run beside the lane kernel
	hipStream_t st_hi;                // bmh_chain_extend_merge: classification, lane kernel and counts (what the first extension pass waits for) at the highest priority
	hipStream_t cls_stream[CH_N_CLASSES]; hipEvent_t cls_done[CH_N_CLASSES]; // ... and beside each other, one stream per size class
	hipEvent_t ev_t[8]; float ms[8]; uint32_t heavy_per_class[CH_N_CLASSES];
	// bmh_chain_extend_merge: the batch in two passes (reads of the lane kernel, reads of the wave kernels)
	uint32_t *off2[4];               // [0] regions A [1] jobs A [2] regions B [3] jobs B: exclusive scans of the per-read counts of a pass
	uint64_t *need_sum;              // [0] sampled occurrences of the wave-kernel reads (bound of their regions)
	int32_t *out3; uint64_t cap_out3;
	hipStream_t side2; hipEvent_t ev_x[7];     // [0..1] extension of pass A  [2] counts of pass B  [3], [6] pass B's emit + extension on its stream  [4] join  [5] end of the merge
	uint64_t n_regs_a, n_jobs_a;                 // kernel times of the last batch (bmh_chain_last_timing)
	int materialize;               // 1: bmh_chain_batch also writes the base arrays q/t (+ qoff/toff)
	const uint8_t *last_reads, *last_pac; uint64_t last_l_pac;   // sources of the last batch, for bmh_chain_extend
};

extern "C" void bmh_chain_ws_free(bmh_chain_ws_t *w)
{
	if (!w) return;
	void *ps[] = {w->seeds, w->chains, w->order, w->opos, w->klist, w->srt, w->cidx, w->regs, w->est, w->regs_per_read, w->jobs_per_read, w->reg_off,
	              w->job_off, w->heavy_list, w->need, w->frac_rep, w->counters, w->ctg_off, w->ctg_len, w->ctg_alt, w->outregs, w->qlen, w->tlen, w->h0, w->job_read, w->job_reg,
	              w->job_side, w->jq_src, w->qoff, w->toff, w->jt0, w->qoff64, w->toff64, w->q, w->t, w->scan_tmp};
	for (void *p : ps) if (p) (void)hipFree(p);
	if (w->h_pin) (void)hipHostFree(w->h_pin);
	if (w->side) (void)hipStreamDestroy(w->side);
	if (w->st_hi) (void)hipStreamDestroy(w->st_hi);
	if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
	if (w->ev_join) (void)hipEventDestroy(w->ev_join);
	for (hipEvent_t e : w->ev_t) if (e) (void)hipEventDestroy(e);
	for (hipStream_t q : w->cls_stream) if (q) (void)hipStreamDestroy(q);
	for (int i = 0; i < 4; ++i) if (w->off2[i]) (void)hipFree(w->off2[i]);
	if (w->need_sum) (void)hipFree(w->need_sum);
	if (w->out3) (void)hipFree(w->out3);
	if (w->side2) { bmh_extend_release((void *)w->side2); (void)hipStreamDestroy(w->side2); }      // (the extension keeps scratch per stream: pass B may have run on this one)
	for (hipEvent_t e : w->ev_x) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : w->cls_done) if (e) (void)hipEventDestroy(e);
	free(w);
}

static int prio_lo_early() { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); return lo; }

extern "C" bmh_chain_ws_t *bmh_chain_ws_create(uint32_t max_reads, uint64_t max_seeds)
{
	if (max_reads == 0 || max_seeds == 0) { bmh_set_error("bmh_chain_ws_create: empty capacity"); return nullptr; }
	bmh_chain_ws *w = (bmh_chain_ws *)calloc(1, sizeof(bmh_chain_ws));
	w->max_reads = max_reads; w->max_seeds = max_seeds;
	bool ok = true;
#define A(p, n) ok = ok && hipMalloc((void **)&(p), (size_t)(n)) == hipSuccess
	const size_t S = (size_t)max_seeds + 1;
	A(w->seeds, sizeof(ch_seed_t) * S); A(w->chains, sizeof(ch_chain_t) * S); A(w->order, 4 * S); A(w->opos, 8 * S); A(w->klist, 4 * S);
	A(w->srt, 8 * S); A(w->cidx, 4 * S); A(w->regs, sizeof(ch_reg_t) * S); A(w->est, sizeof(ch_est_t) * S);
	const size_t Rn = (size_t)max_reads + 1;
	A(w->regs_per_read, 4 * Rn); A(w->jobs_per_read, 4 * Rn); A(w->reg_off, 4 * Rn); A(w->job_off, 4 * Rn); A(w->heavy_list, (CH_N_CLASSES + CH_N_BINS) * 4 * Rn); A(w->need, 4 * Rn); A(w->frac_rep, 4 * Rn);
	A(w->counters, 256);
	for (int i = 0; i < 4; ++i) A(w->off2[i], 4 * Rn);
	A(w->need_sum, 16);
	size_t t1 = 0, t2 = 0;
	rocprim::exclusive_scan(nullptr, t1, w->regs_per_read, w->reg_off, 0u, Rn, rocprim::plus<uint32_t>(), 0);
	rocprim::exclusive_scan(nullptr, t2, (uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0, 2 * S + 1, rocprim::plus<uint64_t>(), 0);
	w->scan_tmp_bytes = t1 > t2 ? t1 : t2;
	A(w->scan_tmp, w->scan_tmp_bytes + 256);
#undef A
	ok = ok && hipHostMalloc((void **)&w->h_pin, 256) == hipSuccess;
	ok = ok && hipStreamCreateWithPriority(&w->side, hipStreamNonBlocking, prio_lo_early()) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming) == hipSuccess;
	for (hipEvent_t &e : w->ev_t) ok = ok && hipEventCreate(&e) == hipSuccess;
	for (hipEvent_t &e : w->ev_x) ok = ok && hipEventCreate(&e) == hipSuccess;
	ok = ok && hipStreamCreateWithFlags(&w->side2, hipStreamNonBlocking) == hipSuccess;
	// the wave kernels are background work: low-priority streams get hardware queues of their own, so that the short kernels of the
	// caller's stream (lane kernel, scans, the first pass's extension) are neither queued behind them nor starved of wave slots
	int prio_lo = 0, prio_hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
	// (BMH_CHAIN_PRIO=hi | normal: experiment knob -- on workloads where these kernels are the critical path of the stage)
	const char *pe = getenv("BMH_CHAIN_PRIO");
	const int cls_prio = pe && pe[0] == 'h' ? prio_hi : pe && pe[0] == 'n' ? 0 : prio_lo;
	// (BMH_CHAIN_CUS=n: experiment knob -- the wave kernels' streams confined to n of the chip's compute units, evenly spread, so that
	// their LDS-hungry waves displace extension waves on those units only; measured in DESIGN.md section 5)
	// bmh_chain_extend_merge: classification, lane kernel and the counts of the first pass -- short kernels the batch's first extension pass waits for -- on a stream
	// of the highest priority: beside another batch's extension they were stretched from 1.5 to 3-4 ms (measured: 35.1-35.2 -> 35.8-36.1 Mreads/s on top of the
	// seeding's priority stream; BMH_CHAIN_LIGHT_PRIO=normal: A/B)
	{ const char *le = getenv("BMH_CHAIN_LIGHT_PRIO"); if (!(le && le[0] == 'n')) ok = ok && hipStreamCreateWithPriority(&w->st_hi, hipStreamNonBlocking, prio_hi) == hipSuccess; }
	const char *cue = getenv("BMH_CHAIN_CUS");
	const int n_cus = cue ? atoi(cue) : 0;
	uint32_t cumask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	if (n_cus > 0 && n_cus < 256) for (int i = 0; i < 256; ++i) if ((long)i * n_cus / 256 != (long)(i + 1) * n_cus / 256) cumask[i >> 5] |= 1u << (i & 31);
	for (int c = 0; c < CH_N_CLASSES; ++c) {
		if (n_cus > 0 && n_cus < 256) ok = ok && hipExtStreamCreateWithCUMask(&w->cls_stream[c], 8, cumask) == hipSuccess;
		else ok = ok && hipStreamCreateWithPriority(&w->cls_stream[c], hipStreamNonBlocking, cls_prio) == hipSuccess;
		ok = ok && hipEventCreate(&w->cls_done[c]) == hipSuccess;
	}
	if (!ok) { bmh_set_error("bmh_chain_ws_create: hipMalloc failed (%s)", hipGetErrorString(hipGetLastError())); bmh_chain_ws_free(w); return nullptr; }
	w->n_contigs = 1;
	w->materialize = 1;
	return w;
}

extern "C" void bmh_chain_last_timing(const bmh_chain_ws_t *w, float ms[8])
{
	for (int i = 0; i < 4; ++i) ms[i] = w->ms[i];
	ms[4] = w->ms[4]; ms[5] = w->ms[5];
	ms[6] = ms[7] = 0.f;                                       // reads chained by a wave: up to 512 entries / beyond
	for (int c = 0; c < CH_N_CLASSES; ++c) ms[c <= 6 ? 6 : 7] += (float)w->heavy_per_class[c];
}

extern "C" int bmh_chain_set_contigs(bmh_chain_ws_t *w, int n_contigs, const int64_t *offset, const int32_t *len)
{
	if (!w) { bmh_set_error("bmh_chain_set_contigs: null workspace"); return BMH_EINVAL; }
	if (w->ctg_off) { (void)hipFree(w->ctg_off); w->ctg_off = nullptr; }
	if (w->ctg_len) { (void)hipFree(w->ctg_len); w->ctg_len = nullptr; }
	if (w->ctg_alt) { (void)hipFree(w->ctg_alt); w->ctg_alt = nullptr; }            // (a new table: its ALT flags are set again, bmh_chain_set_alt)
	w->n_contigs = n_contigs > 1 ? n_contigs : 1;
	if (n_contigs <= 1) return BMH_OK;
	if (!offset || !len) { bmh_set_error("bmh_chain_set_contigs: null arrays"); return BMH_EINVAL; }
	HIPCK(hipMalloc((void **)&w->ctg_off, 8 * (size_t)n_contigs)); HIPCK(hipMalloc((void **)&w->ctg_len, 4 * (size_t)n_contigs));
	HIPCK(hipMemcpy(w->ctg_off, offset, 8 * (size_t)n_contigs, hipMemcpyHostToDevice));
	HIPCK(hipMemcpy(w->ctg_len, len, 4 * (size_t)n_contigs, hipMemcpyHostToDevice));
	return BMH_OK;
}

// which sequences of the table are ALT contigs (the .alt file, src/bntseq.c:179-200): mem_chain_flt treats an overlap with a kept ALT
// chain differently (src/bwamem.c:518).  is_alt[n_contigs], n_contigs as in bmh_chain_set_contigs; null or all zero: none.
extern "C" int bmh_chain_set_alt(bmh_chain_ws_t *w, int n_contigs, const uint8_t *is_alt)
{
	if (!w) { bmh_set_error("bmh_chain_set_alt: null workspace"); return BMH_EINVAL; }
	if (w->ctg_alt) { (void)hipFree(w->ctg_alt); w->ctg_alt = nullptr; }
	bool any = false;
	for (int c = 0; is_alt && c < n_contigs; ++c) any = any || is_alt[c] != 0;
	if (!any) return BMH_OK;
	if (n_contigs != w->n_contigs) { bmh_set_error("bmh_chain_set_alt: %d flags for a table of %d sequences (bmh_chain_set_contigs first)", n_contigs, w->n_contigs); return BMH_EINVAL; }
	HIPCK(hipMalloc((void **)&w->ctg_alt, (size_t)n_contigs));
	HIPCK(hipMemcpy(w->ctg_alt, is_alt, (size_t)n_contigs, hipMemcpyHostToDevice));
	return BMH_OK;
}

extern "C" int bmh_chain_set_materialize(bmh_chain_ws_t *w, int on)
{
	if (!w) { bmh_set_error("bmh_chain_set_materialize: null workspace"); return BMH_EINVAL; }
	w->materialize = on ? 1 : 0;
	return BMH_OK;
}

template <class T> static int grow(T *&p, uint64_t need_elems)
{
	if (p) { (void)hipFree(p); p = nullptr; }
	if (hipMalloc((void **)&p, sizeof(T) * (size_t)need_elems) != hipSuccess) { bmh_set_error("bmh_chain_batch: hipMalloc of %llu bytes failed", (unsigned long long)(sizeof(T) * need_elems)); return BMH_ENOMEM; }
	return BMH_OK;
}

static inline unsigned nblk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

// arguments of the chaining kernels for one batch
static void chain_fill_args(bmh_chain_ws *w, chain_args_t &A, const bmh_chain_opt_t *opt, const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs,
                            const uint32_t *d_lens, uint32_t n_reads, const bmh_seeds_t *seeds)
{
	memset(&A, 0, sizeof(A));
	A.x.reads = d_reads; A.x.read_offs = d_offs; A.x.pac = idx->dev.pac;
	w->last_opt = *opt;
	A.x.o = *opt; A.x.l_pac = (int64_t)idx->dev.l_pac; A.x.n_contigs = w->n_contigs; A.x.ctg_off = w->ctg_off; A.x.ctg_len = w->ctg_len; A.x.ctg_alt = w->ctg_alt;
	A.x.rbeg = seeds->d_rbeg; A.x.qbeg = seeds->d_qbeg; A.x.score = seeds->d_score; A.x.n_ref = seeds->d_n_ref_pos; A.x.prefix = seeds->d_prefix;
	A.x.read_lens = d_lens;
	A.x.g.S = w->seeds; A.x.g.CH = w->chains; A.x.g.order = w->order; A.x.g.opos = w->opos; A.x.g.klist = w->klist; A.x.g.srt = w->srt; A.x.g.cidx = w->cidx;
	A.x.g.E = w->est; A.x.regs = w->regs; A.x.regs_per_read = w->regs_per_read; A.x.jobs_per_read = w->jobs_per_read; A.x.frac_rep = w->frac_rep; A.x.err = (int *)(w->counters + CH_N_CLASSES);
	A.n_reads = n_reads;
	const char *ht = getenv("BMH_CHAIN_HEAVY");
	// (round 3: 8 -- with the 9..16-entry reads on the lane-list stream, beside the first extension pass instead of ahead of it, the step was 2.5 % shorter than
	// with 16; round 6: with those reads chained four per wave and wide scratch records 12 was better (25.4-25.6 against 25.7-26.1 ms per step), with the compact
	// records 8 is again: 24.8-25.1 against 25.1-25.3 ms, five interleaved pairs of 40 steps)
	A.heavy_thresh = ht ? (uint32_t)atoi(ht) : CH_HEAVY_DEFAULT;
	A.heavy_list = w->heavy_list; A.heavy_n = w->counters; A.need = w->need;
	A.light_list = w->heavy_list + (size_t)CH_N_CLASSES * w->max_reads; A.light_n = w->counters + 32;
	A.need_sum = (unsigned long long *)w->need_sum;
#ifdef CH_PROFILE
	{
		const char *pr = getenv("BMH_CHAIN_PROF_READ");
		if (pr) { A.x.prof = (long long *)(w->counters + 12); A.x.prof_read = (uint32_t)atoi(pr); }
	}
#endif
}

// classify, then the lane kernel on st and the wave kernels beside it: every size class on its own stream (the classes differ in
// LDS per block, so they fill different gaps of the CUs, and the largest reads -- the longest chains of dependent steps --
// start at once); w->ev_join is recorded when all of them are through.  join: st waits for it.
template <bool FLT>
static int chain_launch_t(bmh_chain_ws *w, const chain_args_t &A, hipStream_t st, bool join)
{
	const uint32_t n_reads = A.n_reads;
	HIPCK(hipMemsetAsync(w->counters, 0, 256, st));
	HIPCK(hipMemsetAsync(w->need_sum, 0, 16, st));
	HIPCK(hipMemsetAsync(w->regs_per_read + n_reads, 0, 4, st));
	HIPCK(hipMemsetAsync(w->jobs_per_read + n_reads, 0, 4, st));
	HIPCK(hipEventRecord(w->ev_t[0], st));
	chain_classify_kernel<<<nblk(n_reads, 256), 256, 0, st>>>(A);
	HIPCK(hipEventRecord(w->ev_t[1], st));
	HIPCK(hipEventRecord(w->ev_fork, st));
	HIPCK(hipStreamWaitEvent(w->side, w->ev_fork, 0));
	static const int lane_lds_env = [] { const char *e = getenv("BMH_CHAIN_LANE_LDS"); return e ? atoi(e) : 0; }();    // (experiment knob)
	// lane forms with their scratch in private memory; BMH_CHAIN_LANE_PRIVATE bit 0: lane kernel, bit 1: lane-list kernel (A/B runs).  Measured on
	// the bench workload (round 4): the lane kernel (reads of at most 8 entries, lanes nearly in step) 1.10 -> 0.74 ms and the step 29.1-29.7 ->
	// 28.7 ms; the lane-list kernel (9..32 entries, 4.5 KB per lane) 4.2-4.5 -> 5.9-6.0 ms -- its lanes walk their arrays out of step, and
	// in the interleaved layout the dwords of ONE lane's entry lie 256 bytes apart (a 24-byte seed is six lines), so it keeps the global slices
	static const int lane_private_env = [] { const char *e = getenv("BMH_CHAIN_LANE_PRIVATE"); return e ? atoi(e) : 1; }();
	const bool lane_private = (lane_private_env & 1) != 0, list_private = (lane_private_env & 2) != 0;
	if (lane_lds_env && A.heavy_thresh <= 16u && !FLT) {                // the bins from the costliest down, each with the LDS its reads need
		static const uint32_t grid_cap[CH_N_BINS] = {1u << 20, 1u << 20, 2048u, 1024u};
		for (int bin = CH_N_BINS - 1; bin >= 0; --bin) {
			if (!(lane_lds_env >> bin & 1)) continue;
			const uint32_t g = nblk(n_reads, 64) < grid_cap[bin] ? nblk(n_reads, 64) : grid_cap[bin];
			if (bin == CH_N_BINS - 1) HIPCK(hipFuncSetAttribute((const void *)chain_lane_lds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * CH_LANE_LDS_SLICE(16))));
			switch (bin) {
			case 0: chain_lane_lds_kernel<2><<<g, 64, 64 * CH_LANE_LDS_SLICE(2), st>>>(A, bin); break;
			case 1: chain_lane_lds_kernel<4><<<g, 64, 64 * CH_LANE_LDS_SLICE(4), st>>>(A, bin); break;
			case 2: chain_lane_lds_kernel<8><<<g, 64, 64 * CH_LANE_LDS_SLICE(8), st>>>(A, bin); break;
			default: chain_lane_lds_kernel<16><<<g, 64, 64 * CH_LANE_LDS_SLICE(16), st>>>(A, bin); break;
			}
		}
		if ((lane_lds_env & 15) != 15) chain_lane_kernel<false, 0><<<nblk(n_reads, 256), 256, 0, st>>>(A, (uint32_t)lane_lds_env);
	} else if (lane_private && !FLT && A.heavy_thresh <= 8u) {
		// (knob CHAIN_LANE_SPLIT=1: one launch per scratch size instead of one launch that sizes the arrays by the read's bin -- measured: the same traffic, but
		// three launches one behind the other on the path the first extension pass waits for: +0.1 ms per step)
		if (bmh_tune("CHAIN_LANE_SPLIT", 0)) {
			// (ch_bin_of: bin 0 = at most 2 entries, 1 = at most 4, 2 = at most 8, 3 = the rest up to the threshold; the argument is the mask of the bins a launch SKIPS)
			static_assert(CH_N_BINS == 4, "the masks below name four bins");
			chain_lane_kernel<FLT, 8><<<nblk(n_reads, 256), 256, 0, st>>>(A, 0x3u);        // bins 3 and 2
			chain_lane_kernel<FLT, 4><<<nblk(n_reads, 256), 256, 0, st>>>(A, 0xDu);        // bin 1
			chain_lane_kernel<FLT, 2><<<nblk(n_reads, 256), 256, 0, st>>>(A, 0xEu);        // bin 0
		} else chain_lane_kernel<FLT, 8><<<nblk(n_reads, 256), 256, 0, st>>>(A, 0u);
	}
	else if (lane_private && !FLT && A.heavy_thresh <= 12u) chain_lane_kernel<FLT, 12><<<nblk(n_reads, 256), 256, 0, st>>>(A, 0u);
	else if (lane_private && !FLT && A.heavy_thresh <= 16u) chain_lane_kernel<FLT, 16><<<nblk(n_reads, 256), 256, 0, st>>>(A, 0u);
	else chain_lane_kernel<FLT, 0><<<nblk(n_reads, 256), 256, 0, st>>>(A, 0u);
	HIPCK(hipEventRecord(w->ev_t[2], st));
	HIPCK(hipEventRecord(w->ev_t[3], w->side));
	const int ctg_lds = w->n_contigs <= 1 ? 0 : w->n_contigs <= 64 ? 64 : w->n_contigs <= CH_LDS_CONTIGS ? CH_LDS_CONTIGS : 0;
	const void *wave_fn = ctg_lds == 64 ? (const void *)chain_wave_kernel<64, FLT> : ctg_lds ? (const void *)chain_wave_kernel<CH_LDS_CONTIGS, FLT> : (const void *)chain_wave_kernel<0, FLT>;
	typedef typename ch_coop_ty<FLT>::ty TY;
	HIPCK(hipFuncSetAttribute(wave_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CH_CLASS_CAP[CH_N_CLASSES - 2] * ch_lds_entry_bytes<ch_wide_ty>(true))));
	static const bool serial = getenv("BMH_CHAIN_SERIAL") != nullptr;          // (measurement: one class after the other, so that BMH_CHAIN_STATS shows what each costs alone)
	// Ablation knob CHAIN_REPLAY_HEAVY (bit c = size class c): the class's kernel is NOT launched -- the regions, counts and repeat fractions its reads left in
	// this workspace the last time stand.  Only meaningful when the workspace saw the same batch in its previous call (bench.py --distinct-batches 2 with two
	// batches in flight): the step measured without a class is the ceiling of what a faster kernel for that class can gain.
	const unsigned replay = (unsigned)bmh_tune("CHAIN_REPLAY_HEAVY", 0);
	const unsigned wave_prio = (unsigned)bmh_tune("CHAIN_WAVE_PRIO", 0);      // bit c: the waves of size class c raise their priority (s_setprio 3)
	const unsigned sub_mask = (unsigned)bmh_tune("CHAIN_SUB", (1 << CH_N_SUB) - 1);
	const bool sub_ctg = w->n_contigs > 1 && w->n_contigs <= CH_SUB_LDS_CONTIGS;
	if (sub_mask) {
		const int mx = (int)(4 * (size_t)CH_CLASS_CAP[CH_N_SUB - 1] * ch_lds_entry_bytes<ch_wide_ty>(false));
		HIPCK(hipFuncSetAttribute(sub_ctg ? (const void *)chain_sub_kernel<CH_SUB_LDS_CONTIGS, FLT> : (const void *)chain_sub_kernel<0, FLT>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
	}
	for (int cls = CH_N_CLASSES - 1; cls >= 0; --cls) {
		const uint32_t lds_cap = CH_CLASS_CAP[cls];
		const int hybrid = cls >= CH_HYBRID_CLASS && lds_cap != 0;
		const size_t lds_bytes = (size_t)lds_cap * ch_lds_entry_bytes<TY>(hybrid != 0);
		HIPCK(hipStreamWaitEvent(w->cls_stream[cls], w->ev_fork, 0));
		if (serial && cls < CH_N_CLASSES - 1) HIPCK(hipStreamWaitEvent(w->cls_stream[cls], w->cls_done[cls + 1], 0));
		if (replay >> cls & 1u) { }                                                                                       // (ablation: the class's stored output stands)
		else if (cls < CH_N_SUB && (sub_mask >> cls & 1u)) {                                                              // four reads per wave (blocks beyond the list leave at once)
			const size_t sub_lds = 4 * (size_t)lds_cap * ch_lds_entry_bytes<TY>(false);
			if (sub_ctg) chain_sub_kernel<CH_SUB_LDS_CONTIGS, FLT><<<CH_SUB_GRID[cls], 64, sub_lds, w->cls_stream[cls]>>>(A, (uint32_t)cls, lds_cap);
			else chain_sub_kernel<0, FLT><<<CH_SUB_GRID[cls], 64, sub_lds, w->cls_stream[cls]>>>(A, (uint32_t)cls, lds_cap);
		}
		else if (cls < 2) {                                                                                               // (round-5 form: a lane per read over the class's list)
			const int ll = bmh_tune("CHAIN_LIST_LANES", CH_LIST_LANES);
			const uint32_t lanes = (uint32_t)(ll < 1 ? 1 : ll > 64 ? 64 : ll);
			const unsigned lgrid = nblk((uint64_t)nblk(n_reads, lanes) * 64u, 256);
			if (list_private && !FLT) chain_lane_list_kernel<FLT, 32><<<lgrid, 256, 0, w->cls_stream[cls]>>>(A, (uint32_t)cls, lanes);
			else chain_lane_list_kernel<FLT, 0><<<lgrid, 256, 0, w->cls_stream[cls]>>>(A, (uint32_t)cls, lanes);
		}
		else if (ctg_lds == 64) chain_wave_kernel<64, FLT><<<CH_CLASS_GRID[cls], 64, lds_bytes, w->cls_stream[cls]>>>(A, (uint32_t)cls, lds_cap, hybrid, (int)(wave_prio >> cls & 1u));
		else if (ctg_lds) chain_wave_kernel<CH_LDS_CONTIGS, FLT><<<CH_CLASS_GRID[cls], 64, lds_bytes, w->cls_stream[cls]>>>(A, (uint32_t)cls, lds_cap, hybrid, (int)(wave_prio >> cls & 1u));
		else chain_wave_kernel<0, FLT><<<CH_CLASS_GRID[cls], 64, lds_bytes, w->cls_stream[cls]>>>(A, (uint32_t)cls, lds_cap, hybrid, (int)(wave_prio >> cls & 1u));
		HIPCK(hipEventRecord(w->cls_done[cls], w->cls_stream[cls]));
		HIPCK(hipStreamWaitEvent(w->side, w->cls_done[cls], 0));
	}
	HIPCK(hipEventRecord(w->ev_t[4], w->side));
	HIPCK(hipEventRecord(w->ev_join, w->side));
	if (join) HIPCK(hipStreamWaitEvent(st, w->ev_join, 0));
	HIPCK(hipGetLastError());
	return BMH_OK;
}

// The forms with the reference's seed filter (mem_flt_chained_seeds) are launched only when the options let it apply to a read the
// device path takes at all (a -W small enough for some read of at most CH_MAX_READ_LEN bases; without -W it starts beyond ~730 bp):
// they carry the local alignment's rows in private memory.
static bool chain_filter_possible(const bmh_chain_opt_t &o) { return o.min_chain_weight > 0 && chain_core::seed_filter_applies(o, CH_MAX_READ_LEN, nullptr); }
static int chain_launch(bmh_chain_ws *w, const chain_args_t &A, hipStream_t st, bool join)
{
	return chain_filter_possible(A.x.o) ? chain_launch_t<true>(w, A, st, join) : chain_launch_t<false>(w, A, st, join);
}

static int chain_check_args(const char *fn, bmh_chain_ws *w, const bmh_chain_opt_t *opt, const bmh_index_t *idx, uint32_t n_reads, const bmh_seeds_t *seeds)
{
	if (!idx->dev.pac || idx->dev.l_pac == 0) { bmh_set_error("%s: the index was uploaded without the 2-bit reference (pac)", fn); return BMH_EINVAL; }
	if (idx->dev.l_pac * 2 != idx->dev.seq_len) { bmh_set_error("%s: l_pac does not match the index", fn); return BMH_EINVAL; }
	if (n_reads > w->max_reads) { bmh_set_error("%s: %u reads > workspace capacity %u", fn, n_reads, w->max_reads); return BMH_ECAPACITY; }
	if (seeds->n_seeds > w->max_seeds) { bmh_set_error("%s: %llu seeds > workspace capacity %llu", fn, (unsigned long long)seeds->n_seeds, (unsigned long long)w->max_seeds); return BMH_ECAPACITY; }
	if (opt->max_occ < 1 || opt->e_del < 1 || opt->e_ins < 1) { bmh_set_error("%s: bad options", fn); return BMH_EINVAL; }
	return BMH_OK;
}

static void chain_print_stats(bmh_chain_ws *w)
{
	fprintf(stderr, "[chain] lane kernel %.3f ms; wave kernels %.3f ms beside it:", w->ms[1], w->ms[2]);
	for (int c = 0; c < CH_N_CLASSES; ++c) {
		float t = 0; (void)hipEventElapsedTime(&t, w->ev_t[3], w->cls_done[c]);
		fprintf(stderr, " class %d (%u entries) %u reads done at %.3f ms;", c, CH_CLASS_CAP[c], w->heavy_per_class[c], t);
	}
	fprintf(stderr, "\n");
}

extern "C" int bmh_chain_batch(bmh_chain_ws_t *w, const bmh_chain_opt_t *opt, const bmh_index_t *idx, const uint8_t *d_reads,
                               const uint32_t *d_offs, const uint32_t *d_lens, uint32_t n_reads, const bmh_seeds_t *seeds,
                               void *stream_, bmh_dev_jobs_t *out)
{
	if (!w || !opt || !idx || !seeds || !out) { bmh_set_error("bmh_chain_batch: null argument"); return BMH_EINVAL; }
	memset(out, 0, sizeof(*out));
	{ const int rc = chain_check_args("bmh_chain_batch", w, opt, idx, n_reads, seeds); if (rc != BMH_OK) return rc; }
	hipStream_t st = (hipStream_t)stream_;
	w->n_regs = w->n_jobs = 0;
	w->last_reads = d_reads; w->last_pac = idx->dev.pac; w->last_l_pac = idx->dev.l_pac;
	if (n_reads == 0) return BMH_OK;
	chain_args_t A;
	chain_fill_args(w, A, opt, idx, d_reads, d_offs, d_lens, n_reads, seeds);
	{ const int rc = chain_launch(w, A, st, true); if (rc != BMH_OK) return rc; }
	size_t tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, w->regs_per_read, w->reg_off, 0u, (size_t)n_reads + 1, rocprim::plus<uint32_t>(), st));
	tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, w->jobs_per_read, w->job_off, 0u, (size_t)n_reads + 1, rocprim::plus<uint32_t>(), st));
	HIPCK(hipMemcpyAsync(w->h_pin + 0, w->reg_off + n_reads, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(w->h_pin + 1, w->job_off + n_reads, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(w->h_pin + 2, w->counters, 4 * (CH_N_CLASSES + 1), hipMemcpyDeviceToHost, st));
	HIPCK(hipEventRecord(w->ev_t[5], st));
	HIPCK(hipStreamSynchronize(st));
	HIPCK(hipGetLastError());
	(void)hipEventElapsedTime(&w->ms[0], w->ev_t[0], w->ev_t[1]); (void)hipEventElapsedTime(&w->ms[1], w->ev_t[1], w->ev_t[2]);
	(void)hipEventElapsedTime(&w->ms[2], w->ev_t[3], w->ev_t[4]); (void)hipEventElapsedTime(&w->ms[3], w->ev_t[0], w->ev_t[5]);
	for (int c = 0; c < CH_N_CLASSES; ++c) w->heavy_per_class[c] = w->h_pin[2 + c];
	if (getenv("BMH_CHAIN_STATS")) chain_print_stats(w);
	if (w->h_pin[2 + CH_N_CLASSES] == 1) { bmh_set_error("bmh_chain_batch: a read is longer than %d bp (the extension kernels' classes end at 768 columns): use bmh_build_jobs for this batch", CH_MAX_READ_LEN); return BMH_EINVAL; }
	if (w->h_pin[2 + CH_N_CLASSES] != 0) { bmh_set_error("bmh_chain_batch: internal error %u in the chaining kernel", w->h_pin[2 + CH_N_CLASSES]); return BMH_ENODEV; }
#ifdef CH_PROFILE
	if (A.x.prof) {
		long long pf[10];
		HIPCK(hipMemcpy(pf, w->counters + 12, sizeof(pf), hipMemcpyDeviceToHost));
		fprintf(stderr, "chain phases of read %u (x10ns ticks): chains %lld  weights %lld  sort %lld  kept %lld  chain2aln %lld;  occurrence batches: %lld ticks in %lld lock-step attempts, %lld in %lld sequential ones\n", A.x.prof_read,
		        pf[1] - pf[0], pf[2] - pf[1], pf[3] - pf[2], pf[4] - pf[3], pf[5] - pf[4], pf[6], pf[9], pf[7], pf[8]);
	}
#endif
	const uint64_t n_regs = w->h_pin[0], n_jobs = w->h_pin[1];
	w->n_regs = n_regs; w->n_jobs = n_jobs;
	out->n_regs = n_regs; out->n_jobs = n_jobs; out->n_heavy_reads = 0;
	for (int c = 0; c < CH_N_CLASSES; ++c) out->n_heavy_reads += w->h_pin[2 + c];
	for (int c = 0; c < CH_N_CLASSES; ++c) w->heavy_per_class[c] = w->h_pin[2 + c];
	out->d_regs_per_read = w->regs_per_read; out->d_frac_rep = w->frac_rep;
	if (n_regs > w->cap_regs) {
		const uint64_t c = n_regs + n_regs / 4 + 1024;
		if (grow(w->outregs, c) != BMH_OK) return BMH_ENOMEM;
		w->cap_regs = c;
	}
	if (n_jobs + 1 > w->cap_jobs) {
		const uint64_t c = n_jobs + n_jobs / 4 + 1024;
		uint32_t **u32s[] = {&w->qlen, &w->tlen, &w->h0, &w->job_read, &w->job_reg, &w->job_side, &w->jq_src, &w->qoff, &w->toff};
		for (uint32_t **p : u32s) if (grow(*p, c) != BMH_OK) return BMH_ENOMEM;
		if (grow(w->jt0, c) != BMH_OK || grow(w->qoff64, c) != BMH_OK || grow(w->toff64, c) != BMH_OK) return BMH_ENOMEM;
		w->cap_jobs = c;
	}
	if (n_regs == 0) return BMH_OK;
	emit_args_t E;
	E.regs = w->regs; E.prefix = seeds->d_prefix; E.regs_per_read = w->regs_per_read; E.reg_off = w->reg_off; E.job_off = w->job_off; E.need = nullptr; E.thresh = 0; E.pass = 0;
	E.read_offs = d_offs; E.read_lens = d_lens; E.n_reads = n_reads; E.outregs = w->outregs; E.reg_base = E.job_base = 0;
	E.qlen = w->qlen; E.tlen = w->tlen; E.h0 = w->h0; E.job_read = w->job_read; E.job_reg = w->job_reg; E.job_side = w->job_side; E.jq_src = w->jq_src; E.jt0 = w->jt0;
	emit_kernel<<<nblk(n_reads, 256), 256, 0, st>>>(E);
	emit_wave_kernel<<<1024, 256, 0, st>>>(E, w->heavy_list, w->counters, 0, CH_N_CLASSES - 1);
	out->d_qlen = w->qlen; out->d_tlen = w->tlen; out->d_h0 = w->h0; out->d_job_read = w->job_read; out->d_job_reg = w->job_reg; out->d_job_side = w->job_side;
	out->d_qoff = w->qoff; out->d_toff = w->toff;
	if (n_jobs == 0 || !w->materialize) { out->d_qoff = out->d_toff = nullptr; HIPCK(hipGetLastError()); return BMH_OK; }
	// one extra (zero) element so the scans also yield the totals
	HIPCK(hipMemsetAsync(w->qlen + n_jobs, 0, 4, st)); HIPCK(hipMemsetAsync(w->tlen + n_jobs, 0, 4, st));
	tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, w->qlen, w->qoff64, (uint64_t)0, (size_t)n_jobs + 1, rocprim::plus<uint64_t>(), st));
	tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, w->tlen, w->toff64, (uint64_t)0, (size_t)n_jobs + 1, rocprim::plus<uint64_t>(), st));
	uint64_t *h64 = (uint64_t *)(w->h_pin + 16);
	HIPCK(hipMemcpyAsync(h64 + 0, w->qoff64 + n_jobs, 8, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(h64 + 1, w->toff64 + n_jobs, 8, hipMemcpyDeviceToHost, st));
	HIPCK(hipStreamSynchronize(st));
	const uint64_t qb = h64[0], tbytes = h64[1];
	if (qb >= (1ull << 32) || tbytes >= (1ull << 32)) { bmh_set_error("bmh_chain_batch: batch exceeds 4 GiB of bases; split the read set"); return BMH_ECAPACITY; }
	if (qb + 16 > w->cap_q) { const uint64_t c = qb + qb / 4 + 4096; if (grow(w->q, c) != BMH_OK) return BMH_ENOMEM; w->cap_q = c; }
	if (tbytes + 16 > w->cap_t) { const uint64_t c = tbytes + tbytes / 4 + 4096; if (grow(w->t, c) != BMH_OK) return BMH_ENOMEM; w->cap_t = c; }
	mat_args_t M;
	M.reads = d_reads; M.pac = idx->dev.pac; M.l_pac = (int64_t)idx->dev.l_pac;
	M.qlen = w->qlen; M.tlen = w->tlen; M.job_side = w->job_side; M.jq_src = w->jq_src; M.jt0 = w->jt0; M.qoff64 = w->qoff64; M.toff64 = w->toff64;
	M.n_jobs = (uint32_t)n_jobs; M.q = w->q; M.t = w->t; M.qoff = w->qoff; M.toff = w->toff;
	unsigned gb = nblk(n_jobs, 16); if (gb > 65536u) gb = 65536u;
	materialize_kernel<<<gb, 256, 0, st>>>(M);
	HIPCK(hipGetLastError());
	out->q_bytes = qb; out->t_bytes = tbytes; out->d_q = w->q; out->d_t = w->t;
	return BMH_OK;
}

extern "C" int bmh_chain_extend(bmh_chain_ws_t *w, const bmh_ext_params_t *p, int32_t *d_out3, int32_t *d_raw, void *stream_)
{
	if (!w || !p) { bmh_set_error("bmh_chain_extend: null argument"); return BMH_EINVAL; }
	if (w->n_jobs == 0) return BMH_OK;
	bmh_ext_desc_t d;
	d.reads = w->last_reads; d.pac = w->last_pac; d.l_pac = (long long)w->last_l_pac; d.jq_src = w->jq_src; d.job_side = w->job_side; d.jt0 = w->jt0; d.max_qlen = 0;
	return bmh_extend_batch_desc(&d, w->qlen, w->tlen, w->h0, (uint32_t)w->n_jobs, p, d_out3, d_raw, stream_);
}

extern "C" int bmh_chain_merge(bmh_chain_ws_t *w, const int32_t *d_out3, int32_t *d_regs_out, void *stream_)
{
	if (!w || !d_regs_out || (w->n_jobs && !d_out3)) { bmh_set_error("bmh_chain_merge: null argument"); return BMH_EINVAL; }
	if (w->n_regs == 0) return BMH_OK;
	merge_kernel<<<nblk(w->n_regs, 256), 256, 0, (hipStream_t)stream_>>>(w->outregs, (uint32_t)w->n_regs, d_out3, d_regs_out, w->last_opt);
	HIPCK(hipGetLastError());
	return BMH_OK;
}


// ------------------------------------------------------------------------------------------------ the stage in one call

// per-read counts of one pass as a scan input: the count where the read belongs to the pass (pass 1: need > thresh), else 0; one
// element behind the last read (0) so that the scan also yields the total
struct ch_pass_count {
	const uint32_t *need, *cnt; uint32_t thresh, n_reads; int pass;
	__device__ uint32_t operator()(uint32_t r) const { return (r < n_reads && (need[r] > thresh) == (pass == 1)) ? cnt[r] : 0u; }
};
static inline auto ch_pass_iter(const uint32_t *need, const uint32_t *cnt, uint32_t thresh, uint32_t n_reads, int pass)
{
	ch_pass_count f; f.need = need; f.cnt = cnt; f.thresh = thresh; f.n_reads = n_reads; f.pass = pass;
	return rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0u), f);
}

// extension results -> regions in READ order: the regions of pass A sit at [0, n_a) of the pass-ordered list, those of pass B
// behind them; a read has regions in one pass only, so its final offset is off_a[read] + off_b[read]
__global__ void __launch_bounds__(256) merge2_kernel(const ch_outreg_t *__restrict__ regs, uint32_t n_regs, uint32_t n_a, const uint32_t *__restrict__ off_a,
                                                     const uint32_t *__restrict__ off_b, const int32_t *__restrict__ out3, int32_t *__restrict__ regs_out, const bmh_chain_opt_t copt)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_regs) return;
	const ch_outreg_t a = regs[i];
	int score, qb, qe; int64_t rb, re;
	const int sides = (a.job0 >= 0) + (a.job1 >= 0);
	if (sides > 0) {
		int ls = 0, lq = 0, lt = 0, rs = 0, rq = 0, rt = 0;
		if (a.job0 >= 0) { ls = out3[3 * (size_t)a.job0]; lq = out3[3 * (size_t)a.job0 + 1]; lt = out3[3 * (size_t)a.job0 + 2]; }
		if (a.job1 >= 0) { rs = out3[3 * (size_t)a.job1]; rq = out3[3 * (size_t)a.job1 + 1]; rt = out3[3 * (size_t)a.job1 + 2]; }
		score = ls + rs - (sides == 2 ? a.seedlen0 : 0);
		qb = a.seed_qbeg - lq; qe = a.seed_qbeg + a.seedlen0 + rq;
		rb = a.seed_rbeg - lt; re = a.seed_rbeg + a.seedlen0 + rt;
	} else {
		score = ch_bare_seed_score(copt, a.seedlen0, a.l_query); qb = 0; qe = a.l_query; rb = a.seed_rbeg; re = a.seed_rbeg + a.seedlen0;
	}
	const uint32_t dest = i < n_a ? i + off_b[a.read] : off_a[a.read] + (i - n_a);
	int32_t *o = regs_out + 8 * (size_t)dest;
	o[0] = (int32_t)a.read; o[1] = score; o[2] = qb; o[3] = qe;
	o[4] = (int32_t)(uint32_t)rb; o[5] = (int32_t)(rb >> 32); o[6] = (int32_t)(uint32_t)re; o[7] = (int32_t)(re >> 32);
}

static int chain_grow_jobs(bmh_chain_ws *w, uint64_t n_regs, uint64_t n_jobs)
{
	if (n_regs > w->cap_regs) {
		const uint64_t c = n_regs + n_regs / 4 + 1024;
		if (grow(w->outregs, c) != BMH_OK) return BMH_ENOMEM;
		w->cap_regs = c;
	}
	if (n_jobs + 1 > w->cap_jobs) {
		const uint64_t c = n_jobs + n_jobs / 4 + 1024;
		uint32_t **u32s[] = {&w->qlen, &w->tlen, &w->h0, &w->job_read, &w->job_reg, &w->job_side, &w->jq_src, &w->qoff, &w->toff};
		for (uint32_t **p : u32s) if (grow(*p, c) != BMH_OK) return BMH_ENOMEM;
		if (grow(w->jt0, c) != BMH_OK || grow(w->qoff64, c) != BMH_OK || grow(w->toff64, c) != BMH_OK) return BMH_ENOMEM;
		w->cap_jobs = c;
	}
	if (n_jobs + 1 > w->cap_out3) {
		const uint64_t c = n_jobs + n_jobs / 4 + 1024;
		if (grow(w->out3, 3 * c) != BMH_OK) return BMH_ENOMEM;
		w->cap_out3 = c;
	}
	return BMH_OK;
}

// bmh_chain_batch + bmh_chain_extend + bmh_chain_merge as ONE call that hides the chaining of the seed-rich reads.  On an
// hg38-like genome 6 % of the reads sample more than 16 seed occurrences (up to ~1500) and their chaining -- one wave per read,
// a chain of dependent steps, LDS-bound occupancy -- takes as long as the extension of the whole batch, while the other 94 % are
// chained in 2 ms.  So the batch goes in two passes: the reads of the lane kernel are chained, turned into jobs and EXTENDED
// while the wave kernels are still chaining the rest on their side streams; then the rest is extended and everything merged.
// Regions come out in read order, identical to the three-call form (tests/test_gpu_parity.py); the job arrays in *out are in
// pass order (all jobs of pass A, then pass B).
extern "C" int bmh_chain_extend_merge(bmh_chain_ws_t *w, const bmh_chain_opt_t *opt, const bmh_index_t *idx, const uint8_t *d_reads,
                                      const uint32_t *d_offs, const uint32_t *d_lens, uint32_t n_reads, const bmh_seeds_t *seeds,
                                      const bmh_ext_params_t *ep, int32_t *d_regs_out, uint64_t cap_regs_out, void *stream_, bmh_dev_jobs_t *out)
{
	if (!w || !opt || !idx || !seeds || !out || !ep || !d_regs_out) { bmh_set_error("bmh_chain_extend_merge: null argument"); return BMH_EINVAL; }
	memset(out, 0, sizeof(*out));
	{ const int rc = chain_check_args("bmh_chain_extend_merge", w, opt, idx, n_reads, seeds); if (rc != BMH_OK) return rc; }
	hipStream_t st = (hipStream_t)stream_;
	w->n_regs = w->n_jobs = 0;
	w->last_reads = d_reads; w->last_pac = idx->dev.pac; w->last_l_pac = idx->dev.l_pac;
	if (n_reads == 0) return BMH_OK;
	chain_args_t A;
	chain_fill_args(w, A, opt, idx, d_reads, d_offs, d_lens, n_reads, seeds);
	// (the host waits for the caller's stream -- the seeds -- and the short kernels go to the priority stream: a barrier packet waiting in a high-priority queue
	// costs what the priority gains, see bmh_seed_batch)
	hipStream_t st_user = st;
	if (w->st_hi) { HIPCK(hipStreamSynchronize(st)); st = w->st_hi; }
	{ const int rc = chain_launch(w, A, st, false); if (rc != BMH_OK) return rc; }
	// ---- pass A: the reads of the lane kernel
	// (the counts of a pass are read through a transform iterator: no pass over the reads to mask them first)
	size_t tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, ch_pass_iter(w->need, w->regs_per_read, A.heavy_thresh, n_reads, 0), w->off2[0], 0u, (size_t)n_reads + 1, rocprim::plus<uint32_t>(), st));
	tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, ch_pass_iter(w->need, w->jobs_per_read, A.heavy_thresh, n_reads, 0), w->off2[1], 0u, (size_t)n_reads + 1, rocprim::plus<uint32_t>(), st));
	uint64_t *h64 = (uint64_t *)(w->h_pin + 16);
	HIPCK(hipMemcpyAsync(w->h_pin + 0, w->off2[0] + n_reads, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(w->h_pin + 1, w->off2[1] + n_reads, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(h64, w->need_sum, 8, hipMemcpyDeviceToHost, st));
	HIPCK(hipMemcpyAsync(w->h_pin + 40, w->counters + 32 + CH_N_BINS, 4, hipMemcpyDeviceToHost, st));
	HIPCK(hipEventRecord(w->ev_t[5], st));
	HIPCK(hipStreamSynchronize(st));                          // (the lane kernel; the wave kernels go on)
	st = st_user;
	const uint64_t n_regs_a = w->h_pin[0], n_jobs_a = w->h_pin[1], need_b = h64[0];
	// capacity for both passes now (a reallocation later would wait for everything and have to move pass A): pass B makes at
	// most one region per sampled occurrence and two jobs per region
	{ const int rc = chain_grow_jobs(w, n_regs_a + need_b, n_jobs_a + 2 * need_b); if (rc != BMH_OK) return rc; }
	{ const int rc = bmh_extend_reserve(stream_, n_jobs_a > 2 * need_b ? n_jobs_a : 2 * need_b); if (rc != BMH_OK) return rc; }   // (the extension's own scratch likewise)
	emit_args_t E;
	E.regs = w->regs; E.prefix = seeds->d_prefix; E.read_offs = d_offs; E.read_lens = d_lens; E.n_reads = n_reads; E.outregs = w->outregs;
	E.qlen = w->qlen; E.tlen = w->tlen; E.h0 = w->h0; E.job_read = w->job_read; E.job_reg = w->job_reg; E.job_side = w->job_side; E.jq_src = w->jq_src; E.jt0 = w->jt0;
	bmh_ext_desc_t d;
	d.reads = d_reads; d.pac = idx->dev.pac; d.l_pac = (long long)idx->dev.l_pac;
	d.max_qlen = w->h_pin[40];                                  // (the longest read: the extension skips the classes beyond it)
	HIPCK(hipEventRecord(w->ev_x[0], st));
	if (n_regs_a) {
		E.regs_per_read = w->regs_per_read; E.need = w->need; E.thresh = A.heavy_thresh; E.pass = 0; E.reg_off = w->off2[0]; E.job_off = w->off2[1]; E.reg_base = E.job_base = 0;
		emit_kernel<<<nblk(n_reads, 256), 256, 0, st>>>(E);
		if (n_jobs_a) {
			d.jq_src = w->jq_src; d.job_side = w->job_side; d.jt0 = w->jt0;
			const int rc = bmh_extend_batch_desc(&d, w->qlen, w->tlen, w->h0, (uint32_t)n_jobs_a, ep, w->out3, nullptr, stream_);
			if (rc != BMH_OK) return rc;
		}
	}
	HIPCK(hipEventRecord(w->ev_x[1], st));
	// ---- pass B: the reads of the wave kernels, counted on a second stream so that the host does not wait for pass A's extension
	HIPCK(hipStreamWaitEvent(w->side2, w->ev_join, 0));
	// (its own scan scratch: pass A's extension may still be using nothing of ours, but the scans above share scan_tmp with nothing in flight on st)
	tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, ch_pass_iter(w->need, w->regs_per_read, A.heavy_thresh, n_reads, 1), w->off2[2], 0u, (size_t)n_reads + 1, rocprim::plus<uint32_t>(), w->side2));
	tb = w->scan_tmp_bytes;
	HIPCK(rocprim::exclusive_scan(w->scan_tmp, tb, ch_pass_iter(w->need, w->jobs_per_read, A.heavy_thresh, n_reads, 1), w->off2[3], 0u, (size_t)n_reads + 1, rocprim::plus<uint32_t>(), w->side2));
	HIPCK(hipMemcpyAsync(w->h_pin + 32, w->off2[2] + n_reads, 4, hipMemcpyDeviceToHost, w->side2));
	HIPCK(hipMemcpyAsync(w->h_pin + 33, w->off2[3] + n_reads, 4, hipMemcpyDeviceToHost, w->side2));
	HIPCK(hipMemcpyAsync(w->h_pin + 2, w->counters, 4 * (CH_N_CLASSES + 1), hipMemcpyDeviceToHost, w->side2));
	HIPCK(hipEventRecord(w->ev_x[2], w->side2));
	HIPCK(hipStreamSynchronize(w->side2));
	(void)hipEventElapsedTime(&w->ms[0], w->ev_t[0], w->ev_t[1]); (void)hipEventElapsedTime(&w->ms[1], w->ev_t[1], w->ev_t[2]);
	(void)hipEventElapsedTime(&w->ms[2], w->ev_t[3], w->ev_t[4]); (void)hipEventElapsedTime(&w->ms[3], w->ev_t[0], w->ev_t[5]);
	for (int c = 0; c < CH_N_CLASSES; ++c) w->heavy_per_class[c] = w->h_pin[2 + c];
	if (getenv("BMH_CHAIN_STATS")) chain_print_stats(w);
	if (w->h_pin[2 + CH_N_CLASSES] == 1) { bmh_set_error("bmh_chain_extend_merge: a read is longer than %d bp (the extension kernels' classes end at 768 columns): use bmh_build_jobs for this batch", CH_MAX_READ_LEN); return BMH_EINVAL; }
	if (w->h_pin[2 + CH_N_CLASSES] != 0) { bmh_set_error("bmh_chain_extend_merge: internal error %u in the chaining kernel", w->h_pin[2 + CH_N_CLASSES]); return BMH_ENODEV; }
	const uint64_t n_regs_b = w->h_pin[32], n_jobs_b = w->h_pin[33];
	const uint64_t n_regs = n_regs_a + n_regs_b, n_jobs = n_jobs_a + n_jobs_b;
	if (n_regs_b > need_b || n_jobs_b > 2 * need_b) { bmh_set_error("bmh_chain_extend_merge: internal error: pass B outgrew its bound"); return BMH_ENODEV; }
	if (n_regs > cap_regs_out) { bmh_set_error("bmh_chain_extend_merge: %llu regions > capacity %llu of the output array", (unsigned long long)n_regs, (unsigned long long)cap_regs_out); return BMH_ECAPACITY; }
	// Pass B's jobs and extension go to the second stream (knob CHAIN_B_SIDE, 0 = behind pass A on the caller's stream as until round 4): its emit kernels, its
	// prefilter and its job sort -- latency-bound, ~1.5 ms -- then run beside pass A's DP kernels instead of behind them; the merge waits for both.
	const bool b_side = bmh_tune("CHAIN_B_SIDE", CH_B_SIDE) != 0;
	hipStream_t sb = b_side ? w->side2 : st;
	if (!b_side) HIPCK(hipStreamWaitEvent(st, w->ev_x[2], 0));
	HIPCK(hipEventRecord(w->ev_x[3], sb));
	if (n_regs_b) {
		E.regs_per_read = w->regs_per_read; E.need = w->need; E.thresh = A.heavy_thresh; E.pass = 1; E.reg_off = w->off2[2]; E.job_off = w->off2[3]; E.reg_base = (uint32_t)n_regs_a; E.job_base = (uint32_t)n_jobs_a;
		emit_kernel<<<nblk(n_reads, 256), 256, 0, sb>>>(E);
		emit_wave_kernel<<<1024, 256, 0, sb>>>(E, w->heavy_list, w->counters, 0, CH_N_CLASSES - 1);
		if (n_jobs_b) {
			d.jq_src = w->jq_src + n_jobs_a; d.job_side = w->job_side + n_jobs_a; d.jt0 = w->jt0 + n_jobs_a;
			if (b_side) { const int rc = bmh_extend_reserve((void *)sb, 2 * need_b); if (rc != BMH_OK) return rc; }
			const int rc = bmh_extend_batch_desc(&d, w->qlen + n_jobs_a, w->tlen + n_jobs_a, w->h0 + n_jobs_a, (uint32_t)n_jobs_b, ep, w->out3 + 3 * n_jobs_a, nullptr, b_side ? (void *)sb : stream_);
			if (rc != BMH_OK) return rc;
		}
	}
	HIPCK(hipEventRecord(w->ev_x[6], sb));                     // end of pass B on ITS stream (what bmh_chain_extend_merge_timing reports)
	if (b_side) HIPCK(hipStreamWaitEvent(st, w->ev_x[6], 0));
	HIPCK(hipEventRecord(w->ev_x[4], st));
	if (n_regs) merge2_kernel<<<nblk(n_regs, 256), 256, 0, st>>>(w->outregs, (uint32_t)n_regs, (uint32_t)n_regs_a, w->off2[0], w->off2[2], w->out3, d_regs_out, w->last_opt);
	HIPCK(hipEventRecord(w->ev_x[5], st));
	HIPCK(hipGetLastError());
	w->n_regs = n_regs; w->n_jobs = n_jobs; w->n_regs_a = n_regs_a; w->n_jobs_a = n_jobs_a;
	out->n_regs = n_regs; out->n_jobs = n_jobs;
	for (int c = 0; c < CH_N_CLASSES; ++c) out->n_heavy_reads += w->h_pin[2 + c];
	out->d_regs_per_read = w->regs_per_read; out->d_frac_rep = w->frac_rep;
	out->d_qlen = w->qlen; out->d_tlen = w->tlen; out->d_h0 = w->h0; out->d_job_read = w->job_read; out->d_job_reg = w->job_reg; out->d_job_side = w->job_side;
	return BMH_OK;
}

// ms[0] = extension of pass A (emit + extension on the caller's stream), ms[1] = pass B's emit + extension from their first to their last
// command ON THE STREAM THEY RAN ON -- with CHAIN_B_SIDE (the default) that is the second stream and the interval lies BESIDE pass A's, not
// behind it: ms[0] + ms[1] then counts the overlap twice --, ms[2] = from the start of the stage to the end of the merge (HIP events; waits
// for the stream), jobs[0..1] = jobs of the two passes -- of the last bmh_chain_extend_merge
extern "C" int bmh_chain_extend_merge_timing(const bmh_chain_ws_t *w, float ms[3], uint64_t jobs[2])
{
	if (!w) return BMH_EINVAL;
	if (hipEventSynchronize(w->ev_x[5]) != hipSuccess) return BMH_ENODEV;
	(void)hipEventElapsedTime(&ms[0], w->ev_x[0], w->ev_x[1]); (void)hipEventElapsedTime(&ms[1], w->ev_x[3], w->ev_x[6]);
	(void)hipEventElapsedTime(&ms[2], w->ev_t[0], w->ev_x[5]);
	jobs[0] = w->n_jobs_a; jobs[1] = w->n_jobs - w->n_jobs_a;
	return BMH_OK;
}

// wave residency trace (wtrace.h): this translation unit's copy of the trace symbols
WTRACE_DEFINE_SETTER(bmh_wtrace_set_chain)
