// Wave residency trace -- a measurement tool, off unless bmh_wtrace_start was called (then one scalar load per wave otherwise).
// Every wave of an instrumented kernel leaves one record: which kernel, WHERE it ran (HW_ID: wave slot, SIMD, CU, shader array,
// shader engine; XCC_ID: the XCD) and WHEN (s_memrealtime, 100 MHz) -- from which scripts/wave_residency.py rebuilds, per SIMD
// and moment, the waves of each kernel that were resident side by side: how the VALU-bound extension of one batch and the
// gather-bound seeding / chaining of the other actually share the chip.
// Each translation unit that includes this header has its own copy of the three device symbols (the library is built without
// relocatable device code); WTRACE_DEFINE_SETTER(name) gives it the host function that points them at the shared buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WT_SEGS 2048u
struct wtrace_rec_t { uint32_t kid, hw, xcc, aux; unsigned long long t0, t1; };      // 32 bytes

static __device__ wtrace_rec_t *g_wt_buf = nullptr;
static __device__ unsigned int *g_wt_cnt = nullptr;
static __device__ unsigned int g_wt_cap = 0;

// kernel ids (scripts/wave_residency.py names them)
enum { WT_FORWARD = 1, WT_BACKWARD = 2, WT_LOCATE = 3, WT_EXPAND = 4, WT_FILTER = 5, WT_PACK = 6, WT_SCATTER = 7,
       WT_EXT_CLOSED = 16, WT_EXT_PK = 17, WT_EXT_PERSIST = 18, WT_EXT_32 = 19,
       WT_CHAIN_CLASSIFY = 32, WT_CHAIN_LANE = 33, WT_CHAIN_LIST = 34, WT_CHAIN_WAVE = 35, WT_CHAIN_EMIT = 36, WT_CHAIN_MERGE = 37, WT_CHAIN_SUB = 38 };

struct wtrace_scope_t {
	unsigned long long t0;
	uint32_t kid, aux;
	bool on;
	__device__ __forceinline__ wtrace_scope_t(const uint32_t kid_, const uint32_t aux_ = 0) : t0(0), kid(kid_), aux(aux_)
	{
		on = g_wt_buf != nullptr;
		if (on) t0 = wall_clock64();
	}
	__device__ __forceinline__ ~wtrace_scope_t()
	{
		if (on && (threadIdx.x & 63) == 0) {
			// (WT_SEGS counters, a cache line each, the buffer cut into as many segments: two million waves bumping ONE counter were 24 ms of
			// same-address atomics in a 25 ms trace)
			const unsigned int seg = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (WT_SEGS - 1u), per = g_wt_cap / WT_SEGS;
			unsigned int k = atomicAdd(g_wt_cnt + seg * 16u, 1u);
			if (k < per) {
				k += seg * per;
				unsigned hw, xcc;
				asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
				wtrace_rec_t r;
				r.kid = kid; r.hw = hw; r.xcc = xcc; r.aux = aux; r.t0 = t0; r.t1 = wall_clock64();
				g_wt_buf[k] = r;
			}
		}
	}
};

// (start: counter and capacity first, the buffer pointer -- what a wave tests -- last; stop: the buffer pointer first)
#define WTRACE_DEFINE_SETTER(fn)                                                                                              \
	int fn(void *buf, unsigned int *cnt, unsigned int cap)                                                                    \
	{                                                                                                                         \
		wtrace_rec_t *b = (wtrace_rec_t *)buf;                                                                                \
		if (!b && hipMemcpyToSymbol(HIP_SYMBOL(g_wt_buf), &b, sizeof(b)) != hipSuccess) return -1;                            \
		if (hipMemcpyToSymbol(HIP_SYMBOL(g_wt_cnt), &cnt, sizeof(cnt)) != hipSuccess) return -1;                              \
		if (hipMemcpyToSymbol(HIP_SYMBOL(g_wt_cap), &cap, sizeof(cap)) != hipSuccess) return -1;                              \
		if (b && hipMemcpyToSymbol(HIP_SYMBOL(g_wt_buf), &b, sizeof(b)) != hipSuccess) return -1;                             \
		return 0;                                                                                                             \
	}
