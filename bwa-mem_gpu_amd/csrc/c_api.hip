// C-ABI glue of libbwamem_hip.so: error state, device selection, index upload.
// Reference interfaces replaced: gpu_cpy_wrapper / bwt_destroy_gpu
// (/root/reference/src/GPUSeed/seed_gen.cu:1524-1556, 1338-1346).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "bmh_internal.h"

static thread_local char g_err[512] = "";

// ---- measurement knobs: bmh_tune("EXT_PERSIST", d) = bmh_tune_set's value, else $BMH_EXT_PERSIST, else d
static std::mutex g_tune_mu;
static std::map<std::string, int> g_tune_set;
int bmh_tune(const char *name, int dflt)
{
	{
		std::lock_guard<std::mutex> lk(g_tune_mu);
		auto it = g_tune_set.find(name);
		if (it != g_tune_set.end()) return it->second;
	}
	const std::string env = std::string("BMH_") + name;
	const char *e = getenv(env.c_str());
	return e && *e ? atoi(e) : dflt;
}
// sets (or, with clear != 0, forgets) a knob for this process; returns 0
extern "C" int bmh_tune_set(const char *name, int value, int clear)
{
	if (!name) return BMH_EINVAL;
	std::lock_guard<std::mutex> lk(g_tune_mu);
	if (clear) g_tune_set.erase(name); else g_tune_set[name] = value;
	return BMH_OK;
}

// ---- wave residency trace (csrc/wtrace.h): records of up to `cap` waves of the instrumented kernels launched between start and stop
static void *g_wt_dev = nullptr; static unsigned int *g_wt_cnt_dev = nullptr; static unsigned int g_wt_cap_host = 0, g_wt_kept = 0;
#define WT_SEGS 2048u      /* = csrc/wtrace.h */
// Call it while the device is idle: the trace symbols of the three translation units are set one after the other (cnt and cap first, the
// buffer pointer -- what a kernel tests -- last), and they are per device: only the current device is traced.
static void wtrace_drop()
{
	(void)bmh_wtrace_set_seed(nullptr, nullptr, 0); (void)bmh_wtrace_set_chain(nullptr, nullptr, 0); (void)bmh_wtrace_set_extend(nullptr, nullptr, 0);
	if (g_wt_dev) (void)hipFree(g_wt_dev);
	if (g_wt_cnt_dev) (void)hipFree(g_wt_cnt_dev);
	g_wt_dev = nullptr; g_wt_cnt_dev = nullptr; g_wt_cap_host = 0;
}
extern "C" int bmh_wtrace_start(uint32_t cap)
{
	cap -= cap % WT_SEGS;
	if (g_wt_dev || cap == 0) { bmh_set_error("bmh_wtrace_start: a trace is running, or cap < %u", WT_SEGS); return BMH_EINVAL; }
	(void)hipDeviceSynchronize();
	if (hipMalloc(&g_wt_dev, (size_t)cap * 32) != hipSuccess || hipMalloc((void **)&g_wt_cnt_dev, WT_SEGS * 64) != hipSuccess) {
		wtrace_drop(); bmh_set_error("bmh_wtrace_start: no device memory"); return BMH_ENOMEM;
	}
	(void)hipMemset(g_wt_cnt_dev, 0, WT_SEGS * 64);
	g_wt_cap_host = cap;
	if (bmh_wtrace_set_seed(g_wt_dev, g_wt_cnt_dev, cap) || bmh_wtrace_set_chain(g_wt_dev, g_wt_cnt_dev, cap) || bmh_wtrace_set_extend(g_wt_dev, g_wt_cnt_dev, cap)) {
		wtrace_drop(); bmh_set_error("bmh_wtrace_start: hipMemcpyToSymbol failed"); return BMH_ENODEV;
	}
	return BMH_OK;
}
// stops the trace (waits for the device) and copies up to max_recs records of 32 bytes {u32 kernel, hw_id, xcc_id, aux; u64 t0, t1 (100 MHz)} to `out`
// (the segments' records back to back); returns the number of waves that reported (may exceed what was kept), or a negative error
extern "C" int64_t bmh_wtrace_stop(void *out, uint32_t max_recs)
{
	if (!g_wt_dev) { bmh_set_error("bmh_wtrace_stop: no trace running"); return BMH_EINVAL; }
	(void)hipDeviceSynchronize();
	(void)bmh_wtrace_set_seed(nullptr, nullptr, 0); (void)bmh_wtrace_set_chain(nullptr, nullptr, 0); (void)bmh_wtrace_set_extend(nullptr, nullptr, 0);
	std::vector<unsigned int> cnt(WT_SEGS * 16);
	(void)hipMemcpy(cnt.data(), g_wt_cnt_dev, WT_SEGS * 64, hipMemcpyDeviceToHost);
	const unsigned int per = g_wt_cap_host / WT_SEGS;
	int64_t reported = 0; uint32_t kept = 0;
	for (unsigned int sgm = 0; sgm < WT_SEGS; ++sgm) {
		const unsigned int c = cnt[sgm * 16];
		reported += c;
		unsigned int k = c < per ? c : per;
		if (kept + k > max_recs) k = max_recs - kept;
		if (out && k) (void)hipMemcpy((char *)out + (size_t)kept * 32, (char *)g_wt_dev + (size_t)sgm * per * 32, (size_t)k * 32, hipMemcpyDeviceToHost);
		kept += k;
	}
	(void)hipFree(g_wt_dev); (void)hipFree(g_wt_cnt_dev);
	g_wt_dev = nullptr; g_wt_cnt_dev = nullptr; g_wt_cap_host = 0;
	g_wt_kept = kept;
	return reported;
}
// records the last bmh_wtrace_stop copied out
extern "C" uint32_t bmh_wtrace_kept(void) { return g_wt_kept; }

extern "C" void bmh_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

extern "C" const char *bmh_last_error(void) { return g_err; }


extern "C" int bmh_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { bmh_set_error("hipGetDeviceCount failed: no HIP device"); return 0; }
	return n;
}

extern "C" int bmh_set_device(int dev)
{
	hipError_t e = hipSetDevice(dev);
	if (e != hipSuccess) { bmh_set_error("hipSetDevice(%d): %s", dev, hipGetErrorString(e)); return BMH_ENODEV; }
	return BMH_OK;
}

static int check_geometry(uint64_t seq_len, const uint64_t L2[5], uint64_t n_words, int sa_intv, uint64_t n_sa)
{
	if (seq_len == 0 || (seq_len >> 33)) { bmh_set_error("index: seq_len %llu outside (0, 2^33)", (unsigned long long)seq_len); return BMH_EINVAL; }
	if (L2[0] != 0 || L2[4] != seq_len) { bmh_set_error("index: L2 inconsistent with seq_len"); return BMH_EINVAL; }
	for (int c = 0; c < 4; ++c)
		if (L2[c + 1] < L2[c] || ((L2[c + 1] - L2[c]) >> 32)) { bmh_set_error("index: count of base %d does not fit 32 bits", c); return BMH_EINVAL; }
	uint64_t nblk = (seq_len + 63) / 64;
	uint64_t need = (nblk - 1) * 8 + 4 + ((seq_len + 15) / 16 - (nblk - 1) * 4) + 4;
	if (n_words < need) { bmh_set_error("index: %llu bwt words, need %llu", (unsigned long long)n_words, (unsigned long long)need); return BMH_EINVAL; }
	if (sa_intv < 1 || (sa_intv & (sa_intv - 1))) { bmh_set_error("index: sa_intv %d is not a power of two", sa_intv); return BMH_EINVAL; }
	if (n_sa != (seq_len + sa_intv) / sa_intv) { bmh_set_error("index: n_sa mismatch"); return BMH_EINVAL; }
	return BMH_OK;
}

// Reference blocks {occ[4]; bwt[4]} -> native blocks {occ[4]; lo; hi} (fmd_dev.h), one thread per block; src == dst converts in
// place (a thread reads its whole block before it writes it).
__global__ void __launch_bounds__(256) fmd_native_blocks_kernel(const uint4 *__restrict__ src, uint4 *dst, uint64_t n_blocks)
{
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
		const uint4 occ = src[2 * b], w = src[2 * b + 1];
		const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
		uint64_t lo = 0, hi = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const uint32_t r = __brev(wv[i]);                  // symbol t: high bit at 2t, low bit at 2t + 1
			hi |= (uint64_t)fmd_even_bits16(r) << (16 * i);
			lo |= (uint64_t)fmd_even_bits16(r >> 1) << (16 * i);
		}
		dst[2 * b] = occ;
		dst[2 * b + 1] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
	}
}

static bool native_blocks(const void *src, void *dst, uint64_t seq_len)
{
	const uint64_t n_blocks = (seq_len + 63) / 64 + 1, nb = (n_blocks + 255) / 256;
	fmd_native_blocks_kernel<<<(unsigned)(nb < (1u << 20) ? nb : (1u << 20)), 256>>>((const uint4 *)src, (uint4 *)dst, n_blocks);
	return hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
}

static void fill_dev(bmh_index *ix, uint64_t primary, const uint64_t L2[5], uint64_t seq_len, int sa_intv, uint64_t n_sa, uint64_t l_pac)
{
	ix->dev.primary = primary;
	memcpy(ix->dev.L2, L2, sizeof(uint64_t) * 5);
	ix->dev.seq_len = seq_len;
	ix->dev.n_sa = n_sa;
	int sh = 0;
	while ((1 << sh) < sa_intv) ++sh;
	ix->dev.sa_shift = sh;
	ix->dev.l_pac = l_pac;
}

extern "C" bmh_index_t *bmh_index_upload(uint64_t primary, const uint64_t L2[5], uint64_t seq_len, const uint32_t *bwt_words,
                                         uint64_t n_words, int sa_intv, const uint32_t *sa, uint64_t n_sa, const uint32_t *sa_bits,
                                         const uint8_t *pac, uint64_t l_pac)
{
	if (!L2 || !bwt_words || !sa || !sa_bits) { bmh_set_error("bmh_index_upload: null argument"); return nullptr; }
	if (check_geometry(seq_len, L2, n_words, sa_intv, n_sa) != BMH_OK) return nullptr;
	bmh_index *ix = (bmh_index *)calloc(1, sizeof(bmh_index));
	ix->owns = true; ix->n_words = n_words;
	fill_dev(ix, primary, L2, seq_len, sa_intv, n_sa, pac ? l_pac : 0);
	// blocks are read as 32-byte units: pad the allocation so the last (partial) block is readable
	size_t bwt_bytes = ((size_t)((seq_len + 63) / 64) + 1) * 32;
	size_t bits_words = (size_t)(n_sa / 32 + 1);
	uint32_t *d_bwt = nullptr, *d_sa = nullptr, *d_bits = nullptr;
	uint8_t *d_pac = nullptr;
	bool ok = hipMalloc((void **)&d_bwt, bwt_bytes) == hipSuccess && hipMalloc((void **)&d_sa, n_sa * 4) == hipSuccess &&
	          hipMalloc((void **)&d_bits, bits_words * 4) == hipSuccess;
	// + 16: fmd_text16 reads the text in aligned words and may touch up to 7 bytes past the last symbol's byte
	if (ok && pac) ok = hipMalloc((void **)&d_pac, (size_t)(l_pac / 4 + 1) + 16) == hipSuccess && hipMemset(d_pac, 0, (size_t)(l_pac / 4 + 1) + 16) == hipSuccess;
	if (ok) ok = hipMemset(d_bwt, 0, bwt_bytes) == hipSuccess;
	if (ok) ok = hipMemcpy(d_bwt, bwt_words, (size_t)n_words * 4 < bwt_bytes ? (size_t)n_words * 4 : bwt_bytes, hipMemcpyHostToDevice) == hipSuccess;
	if (ok) ok = hipMemcpy(d_sa, sa, n_sa * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (ok) ok = hipMemcpy(d_bits, sa_bits, bits_words * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (ok && pac) ok = hipMemcpy(d_pac, pac, (size_t)(l_pac / 4 + 1), hipMemcpyHostToDevice) == hipSuccess;
	if (ok) ok = native_blocks(d_bwt, d_bwt, seq_len);            // the file layout becomes the native one in place
	if (!ok) {
		bmh_set_error("bmh_index_upload: %s", hipGetErrorString(hipGetLastError()));
		if (d_bwt) (void)hipFree(d_bwt); if (d_sa) (void)hipFree(d_sa); if (d_bits) (void)hipFree(d_bits); if (d_pac) (void)hipFree(d_pac);
		free(ix);
		return nullptr;
	}
	ix->dev.blocks = (const uint4 *)d_bwt; ix->dev.sa = d_sa; ix->dev.sa_bits = d_bits; ix->dev.pac = d_pac;
	return ix;
}

extern "C" bmh_index_t *bmh_index_from_device(uint64_t primary, const uint64_t L2[5], uint64_t seq_len, const uint32_t *d_bwt_words,
                                              uint64_t n_words, int sa_intv, const uint32_t *d_sa, uint64_t n_sa,
                                              const uint32_t *d_sa_bits, const uint8_t *d_pac, uint64_t l_pac)
{
	if (!L2 || !d_bwt_words || !d_sa || !d_sa_bits) { bmh_set_error("bmh_index_from_device: null argument"); return nullptr; }
	if (check_geometry(seq_len, L2, n_words, sa_intv, n_sa) != BMH_OK) return nullptr;
	if (((uintptr_t)d_bwt_words & 31) != 0) { bmh_set_error("bmh_index_from_device: bwt words must be 32-byte aligned"); return nullptr; }
	// whole 32-byte blocks must be readable: the caller's buffer has to cover ceil(seq_len/64)+1 blocks
	if (n_words < (((seq_len + 63) / 64) + 1) * 8) { bmh_set_error("bmh_index_from_device: buffer must be padded to %llu words", (unsigned long long)((((seq_len + 63) / 64) + 1) * 8)); return nullptr; }
	// the caller's buffer keeps the reference layout (it may write it to a file or broadcast it); the handle searches its own native
	// re-encoding of the blocks (seq_len / 2 bytes: 3.1 GB for hg38, of 288)
	const size_t bwt_bytes = ((size_t)((seq_len + 63) / 64) + 1) * 32;
	void *d_native = nullptr;
	if (hipMalloc(&d_native, bwt_bytes) != hipSuccess || !native_blocks(d_bwt_words, d_native, seq_len)) {
		bmh_set_error("bmh_index_from_device: native blocks: %s", hipGetErrorString(hipGetLastError()));
		if (d_native) (void)hipFree(d_native);
		return nullptr;
	}
	bmh_index *ix = (bmh_index *)calloc(1, sizeof(bmh_index));
	ix->owns = false; ix->owns_blocks = true; ix->n_words = n_words;
	fill_dev(ix, primary, L2, seq_len, sa_intv, n_sa, d_pac ? l_pac : 0);
	ix->dev.blocks = (const uint4 *)d_native; ix->dev.sa = d_sa; ix->dev.sa_bits = d_sa_bits; ix->dev.pac = d_pac;
	return ix;
}

// ---- several GPUs from C: the index lives on every device, the reads shard, one host worker thread per
// device.  The copy goes device to device (hipMemcpyPeer: xGMI between the GPUs of a node); the Python launcher does the same step
// with one RCCL broadcast per array (bwamem_hip/parallel.py).
extern "C" int bmh_index_replicate(const bmh_index_t *src, int src_device, int dst_device, bmh_index_t **out)
{
	if (!src || !out) { bmh_set_error("bmh_index_replicate: null argument"); return BMH_EINVAL; }
	*out = nullptr;
	int prev = 0;
	if (hipGetDevice(&prev) != hipSuccess) { bmh_set_error("bmh_index_replicate: no HIP device"); return BMH_ENODEV; }
	const fmd_dev_t &f = src->dev;
	const size_t bwt_bytes = ((size_t)((f.seq_len + 63) / 64) + 1) * 32, sa_bytes = (size_t)f.n_sa * 4, bits_bytes = (size_t)(f.n_sa / 32 + 1) * 4;
	const size_t pac_bytes = f.pac ? (size_t)(f.l_pac / 4 + 1) + 16 : 0;
	bmh_index *ix = (bmh_index *)calloc(1, sizeof(bmh_index));
	*ix = *src; ix->owns = true; ix->owns_sa = false; ix->owns_blocks = false;
	void *d_bwt = nullptr, *d_sa = nullptr, *d_bits = nullptr, *d_pac = nullptr;
	bool ok = hipSetDevice(dst_device) == hipSuccess;
	ok = ok && hipMalloc(&d_bwt, bwt_bytes) == hipSuccess && hipMalloc(&d_sa, sa_bytes) == hipSuccess && hipMalloc(&d_bits, bits_bytes) == hipSuccess;
	if (ok && pac_bytes) ok = hipMalloc(&d_pac, pac_bytes) == hipSuccess && hipMemset(d_pac, 0, pac_bytes) == hipSuccess;
	auto cp = [&](void *dst, const void *s_, size_t n) {
		return src_device == dst_device ? hipMemcpy(dst, s_, n, hipMemcpyDeviceToDevice) == hipSuccess : hipMemcpyPeer(dst, dst_device, s_, src_device, n) == hipSuccess;
	};
	// (the source's bwt buffer may be the caller's: bmh_index_from_device promises whole blocks only, which is what is copied)
	ok = ok && cp(d_bwt, f.blocks, bwt_bytes) && cp(d_sa, f.sa, sa_bytes) && cp(d_bits, f.sa_bits, bits_bytes);
	if (ok && pac_bytes) ok = cp(d_pac, f.pac, (size_t)(f.l_pac / 4 + 1));
	ok = ok && hipDeviceSynchronize() == hipSuccess;
	if (!ok) {
		bmh_set_error("bmh_index_replicate (device %d -> %d): %s", src_device, dst_device, hipGetErrorString(hipGetLastError()));
		void *ps[] = {d_bwt, d_sa, d_bits, d_pac};
		for (void *p : ps) if (p) (void)hipFree(p);
		free(ix); (void)hipSetDevice(prev);
		return BMH_ENOMEM;
	}
	ix->dev.blocks = (const uint4 *)d_bwt; ix->dev.sa = (const uint32_t *)d_sa; ix->dev.sa_bits = (const uint32_t *)d_bits; ix->dev.pac = (const uint8_t *)d_pac;
	(void)hipSetDevice(prev);
	*out = ix;
	return BMH_OK;
}

// contiguous shard of n units for worker `rank` of `world`, cut at multiples of `multiple` (2: interleaved pairs stay together);
// the same arithmetic as bwamem_hip/parallel.py shard_range
extern "C" void bmh_shard_range(uint64_t n, int rank, int world, uint32_t multiple, uint64_t *lo, uint64_t *hi)
{
	if (multiple < 1) multiple = 1;
	if (world < 1) world = 1;
	const uint64_t units = n / multiple;
	uint64_t a = units * (uint64_t)rank / (uint64_t)world * multiple, b = units * (uint64_t)(rank + 1) / (uint64_t)world * multiple;
	if (rank == world - 1) b = n;
	if (lo) *lo = a;
	if (hi) *hi = b;
}

// Denser suffix-array samples, computed on the device from the sparser ones (each new sample walks LF to the next old
// one, src/bwt.c:105-115).  The reference's GPU index keeps every 16th row (src/bwtindex.c:324) because its cards
// hold 16-32 GB; with 288 GB the samples of every 4th row -- or all of them -- fit beside the index, and locating a
// seed costs 2.5 (or 1) index gathers instead of 8.5.  Values are unchanged, so are all results.
__global__ void __launch_bounds__(256) sa_densify_kernel(fmd_dev_t f, int new_shift, uint64_t n_new, uint32_t *__restrict__ sa, uint32_t *__restrict__ bits)
{
	// grid-stride: a launch cannot have 2^32 threads, and hg38 has 6.2e9 rows
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_new; j += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t k = j << new_shift;
		const uint64_t v = j == 0 ? 0xFFFFFFFFull : fmd_sa(f, k);
		sa[j] = (uint32_t)v;
		if (j && ((v >> 32) & 1)) atomicOr(&bits[j >> 5], 1u << (j & 31));
	}
}

extern "C" int bmh_index_densify_sa(bmh_index_t *ix, int new_intv)
{
	if (!ix || new_intv < 1 || (new_intv & (new_intv - 1))) { bmh_set_error("bmh_index_densify_sa: bad argument"); return BMH_EINVAL; }
	int new_shift = 0;
	while ((1 << new_shift) < new_intv) ++new_shift;
	if (new_shift >= ix->dev.sa_shift) return BMH_OK;                      // already that dense
	const uint64_t n_new = (ix->dev.seq_len + (uint64_t)new_intv) / (uint64_t)new_intv;
	uint32_t *d_sa = nullptr, *d_bits = nullptr;
	const size_t bits_words = (size_t)(n_new / 32 + 1);
	if (hipMalloc((void **)&d_sa, n_new * 4) != hipSuccess || hipMalloc((void **)&d_bits, bits_words * 4) != hipSuccess ||
	    hipMemset(d_bits, 0, bits_words * 4) != hipSuccess) {
		bmh_set_error("bmh_index_densify_sa: %s", hipGetErrorString(hipGetLastError()));
		if (d_sa) (void)hipFree(d_sa); if (d_bits) (void)hipFree(d_bits);
		return BMH_ENODEV;
	}
	const uint64_t nb = (n_new + 255) / 256;
	sa_densify_kernel<<<(unsigned)(nb < (1u << 22) ? nb : (1u << 22)), 256>>>(ix->dev, new_shift, n_new, d_sa, d_bits);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipDeviceSynchronize();
	if (e != hipSuccess) { bmh_set_error("bmh_index_densify_sa: %s", hipGetErrorString(e)); (void)hipFree(d_sa); (void)hipFree(d_bits); return BMH_ENODEV; }
	if (ix->owns || ix->owns_sa) { (void)hipFree((void *)ix->dev.sa); (void)hipFree((void *)ix->dev.sa_bits); }
	ix->dev.sa = d_sa; ix->dev.sa_bits = d_bits; ix->dev.n_sa = n_new; ix->dev.sa_shift = new_shift; ix->owns_sa = true;
	return BMH_OK;
}

extern "C" void bmh_index_free(bmh_index_t *ix)
{
	if (!ix) return;
	if (ix->owns || ix->owns_blocks) (void)hipFree((void *)ix->dev.blocks);
	if (ix->owns && ix->dev.pac) (void)hipFree((void *)ix->dev.pac);
	if (ix->owns || ix->owns_sa) { (void)hipFree((void *)ix->dev.sa); (void)hipFree((void *)ix->dev.sa_bits); }
	free(ix);
}

// Rank primitives at given rows, for direct pins against the reference's known-answer vectors (tests/golden/occ_kat.npz) and for
// callers that want single queries: what = 0: out[4 i .. 4 i + 3] = Occ(rows[i], A / C / G / T) (bwt_occ4, src/bwt.c:309-330);
// 1: out[i] = LF(rows[i]) (bwt_invPsi, src/bwt.c:64-70);  2: out[i] = SA[rows[i]] (bwt_sa, src/bwt.c:105-115).
__global__ void __launch_bounds__(256) index_probe_kernel(fmd_dev_t f, const uint64_t *__restrict__ rows, uint64_t n, int what, uint64_t *__restrict__ out)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const uint64_t k = rows[t];
	if (what == 0) {
		uint64_t c4[4];
		fmd_occ4(f, k, c4);
		// the one-symbol form must agree with the four-symbol one: report a mismatch as an impossible value
		for (int c = 0; c < 4; ++c) out[4 * t + c] = fmd_occ1(f, k, c) == c4[c] ? c4[c] : ~0ull;
	} else if (what == 1) out[t] = fmd_inv_psi(f, k);
	else out[t] = fmd_sa(f, k);
}

extern "C" int bmh_index_probe(const bmh_index_t *ix, const uint64_t *d_rows, uint64_t n, int what, uint64_t *d_out, void *stream)
{
	if (!ix || !d_rows || !d_out || what < 0 || what > 2) { bmh_set_error("bmh_index_probe: bad argument"); return BMH_EINVAL; }
	if (n == 0) return BMH_OK;
	if (n >> 31) { bmh_set_error("bmh_index_probe: too many rows"); return BMH_EINVAL; }
	index_probe_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(ix->dev, d_rows, n, what, d_out);
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) { bmh_set_error("bmh_index_probe: %s", hipGetErrorString(e)); return BMH_ENODEV; }
	return BMH_OK;
}
