// Reads in host memory -> SAM text, driven from C threads (the host side of the gase_aln run around the
// device path; what worker1 / worker2 + kt_pipeline do in the reference, /root/reference/src/bwamem.c:2042-2340, src/fastmap.c:59-120).
//
// The stages exist already as C-ABI entry points (seeding, chaining, extension, merge, the region tail, CIGARs, the formatter); until
// round 4 a Python loop called them batch after batch and was three to ten times slower than the device path it drove.  Here:
//   * `n_lanes` worker threads take the batches of the read set in turn; a worker owns a stream, its workspaces and pinned staging and
//     takes its batch through H2D -> seeding -> chaining -> extension -> merge -> region tail -> CIGARs -> D2H;
//   * one writer thread formats the finished batches IN ORDER (bmh_format_sam is itself multi-threaded) and hands the text to the
//     caller's sink while the workers are on the next batches.
// Single-end batches finish their region tail on the device (bmh_finalize_regs_device); a batch it refuses (BMH_ECAPACITY), an index
// with ALT contigs and interleaved pairs (bmh_finalize_pairs_dev) take the host forms -- the same choices bwamem_hip/aligner.py makes,
// the same text.  Reads longer than 700 bases (the device job builder's limit) are refused: the caller takes the slower path.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "bmh_internal.h"
#include "pair_kernels.h"

namespace {

struct aligner_t {
	const bmh_index_t *idx;
	const uint8_t *pac; int64_t l_pac;
	int n_contigs;
	std::vector<std::string> names; std::vector<const char *> name_ptr;
	std::vector<int64_t> off; std::vector<int32_t> len; std::vector<uint8_t> alt; bool has_alt;
	bmh_chain_opt_t co; bmh_ext_params_t ep; bmh_post_opt_t po; bmh_pe_opt_t pe;
	std::string rg_id;             // the aligner's own copy of popt->rg_id (po.rg_id points into it): the caller's string need not outlive the call that created the aligner
};

}   // namespace

// the CPUs this process may actually use: its affinity mask, capped by the cgroup's CPU quota (a container that shows 256 hardware threads
// may be granted 16: two hundred threads on sixteen CPUs spend their time switching)
extern "C" int bmh_effective_cpus(void)
{
	int n = (int)std::thread::hardware_concurrency();
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (n < 1 || c < n)) n = c; }
	FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (f) {
		char q[64]; long long per = 0;
		if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { const long long c = (atoll(q) + per - 1) / per; if (c > 0 && c < n) n = (int)c; }
		fclose(f);
	} else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) != nullptr) {
		long long quota = -1, per = 0;
		if (fscanf(f, "%lld", &quota) != 1) quota = -1;
		fclose(f);
		FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
		if (g) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
		if (quota > 0 && per > 0) { const long long c = (quota + per - 1) / per; if (c > 0 && c < n) n = (int)c; }
	}
	return n < 1 ? 1 : n;
}

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// device / pinned buffers that only ever grow
template <class T> struct dbuf_t {
	T *p = nullptr; size_t cap = 0;
	dbuf_t() = default;
	dbuf_t(const dbuf_t &) = delete;
	dbuf_t &operator=(const dbuf_t &) = delete;
	void swap(dbuf_t &o) { T *q = p; p = o.p; o.p = q; const size_t c = cap; cap = o.cap; o.cap = c; }
	int need(size_t n) {
		if (n <= cap) return BMH_OK;
		if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
		const size_t c = n + n / 4 + 1024;
		if (hipMalloc((void **)&p, c * sizeof(T)) != hipSuccess) { bmh_set_error("bmh_aligner_run: %zu bytes of device memory: %s", c * sizeof(T), hipGetErrorString(hipGetLastError())); return BMH_ENOMEM; }
		cap = c;
		return BMH_OK;
	}
	~dbuf_t() { if (p) (void)hipFree(p); }
};
template <class T> struct hbuf_t {
	T *p = nullptr; size_t cap = 0;
	hbuf_t() = default;
	hbuf_t(const hbuf_t &) = delete;
	hbuf_t &operator=(const hbuf_t &) = delete;
	int need(size_t n) {
		if (n <= cap) return BMH_OK;
		if (p) (void)hipHostFree(p);
		p = nullptr; cap = 0;
		const size_t c = n + n / 4 + 1024;
		if (hipHostMalloc((void **)&p, c * sizeof(T), hipHostMallocDefault) != hipSuccess) { bmh_set_error("bmh_aligner_run: %zu bytes of pinned memory: %s", c * sizeof(T), hipGetErrorString(hipGetLastError())); return BMH_ENOMEM; }
		cap = c;
		return BMH_OK;
	}
	~hbuf_t() { if (p) (void)hipHostFree(p); }
};

// What the writer needs of a finished batch.  The large arrays are PINNED buffers the device copies straight into; a result goes back
// to a pool when its text is written, so the buffers are allocated a few times per run, not per batch (a million reads leave 70 MB
// of records and 190 MB of CIGAR / MD arrays: every extra pass over them on one thread costs what a device stage costs).
struct result_t {
	uint32_t b0 = 0, n = 0; uint64_t m = 0, n_sel = 0;
	const bmh_read_set_t *rs = nullptr; int64_t id0 = 0; void *token = nullptr;      // the read set the batch [b0, b0 + n) lies in, the index of its first read in the run, the source's handle
	hbuf_t<int32_t> fin, aln, slot32; hbuf_t<uint32_t> opr, off, packed;
	hbuf_t<char> text; uint64_t text_len = 0; bool has_text = false;     // the text written on the device (no formatting on the host)
	std::vector<int64_t> slot;                   // (host selection only; empty: slot32)
	std::vector<int32_t> h_rec, unflag;          // pairs
	std::vector<uint32_t> dev_index;             // record -> its place in the lane's d_fin when that is not the record's own index (ALT indexes); empty: identity
};

struct lane_t {
	hipStream_t st = nullptr, st2 = nullptr;     // st2: copies to the host beside the kernels of st
	hipStream_t st_hi = nullptr;                 // highest priority: takes st's place for what follows the extension (region tail, selection, CIGARs, text) -- run_batch
	bmh_seed_ws_t *sws = nullptr; uint32_t sws_reads = 0; uint64_t sws_bases = 0;
	bmh_chain_ws_t *cws = nullptr; uint32_t cws_reads = 0; uint64_t cws_seeds = 0; uint64_t regs_guess = 0;
	dbuf_t<uint8_t> d_reads, d_work; dbuf_t<uint32_t> d_offs, d_lens, d_sel, d_opr, d_cigar, d_off, d_packed, d_over, d_sel2; dbuf_t<int32_t> d_out3, d_regs, d_fin, d_aln, d_slot, d_hrec, d_unflag; dbuf_t<char> d_md;
	dbuf_t<int32_t> d_dedup, d_fin2, d_hrec2, d_unflag2, d_rslot, d_hh, d_ufh, d_finh; dbuf_t<uint32_t> d_roff, d_roff2, d_opr2, d_todo_pairs, d_oprh, d_offh; dbuf_t<uint8_t> d_todo, d_scan;   // pairs on the device
	hbuf_t<uint8_t> h_todo; std::vector<uint32_t> todo_pairs, offh; void *pair_scratch = nullptr;
	dbuf_t<uint32_t> d_alt_ids, d_alt_off; dbuf_t<int32_t> d_alt_sub;         // ALT contigs: the reads the host redid, on their way into d_fin
	dbuf_t<uint32_t> d_cg2; dbuf_t<int32_t> d_aln2; dbuf_t<char> d_md2; uint32_t max_read_len = 0;      // the redo of the alignments that overflow the fixed slots
	dbuf_t<char> d_names, d_text, d_ctg_names; dbuf_t<uint64_t> d_name_off, d_text_off; dbuf_t<uint32_t> d_ctg_name_off; dbuf_t<int64_t> d_ctg_off; bool ctg_up = false;
	hbuf_t<uint32_t> h_offs, h_sel, h_rpr; hbuf_t<int32_t> h_regs; hbuf_t<float> h_fr; hbuf_t<uint8_t> h_need, h_reads; hbuf_t<char> h_names; hbuf_t<uint64_t> h_name_off;
	double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // h2d, seed, chain+extend+merge, tail, select, cigar + d2h, wait at the gate
	// the copies themselves, timed by events on the lane's stream (what t[0] and t[5] are NOT: those are host clocks around the whole stage -- the staging of the
	// letters, offsets and names on host threads / the CIGAR and text kernels the host waits for): [0] reads, offsets, names H2D  [1] SAM text D2H
	hipEvent_t ev_c[4] = {nullptr, nullptr, nullptr, nullptr}; double copy_ms[2] = {0, 0}; uint64_t copy_bytes[2] = {0, 0}; bool d2h_marked = false;
	~lane_t()
	{
		for (hipEvent_t e : ev_c) if (e) (void)hipEventDestroy(e);
		if (sws) bmh_seed_ws_free(sws);
		if (cws) bmh_chain_ws_free(cws);
		if (st) { bmh_extend_release(st); (void)hipStreamDestroy(st); }
		if (st2) (void)hipStreamDestroy(st2);
		if (st_hi) { bmh_extend_release(st_hi); (void)hipStreamDestroy(st_hi); }
		if (pair_scratch) bmh_pairs_scratch_free(pair_scratch);
	}
};

#define LCK(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) { bmh_set_error("bmh_aligner_run: %s: %s", #x, hipGetErrorString(e_)); return BMH_ENODEV; } } while (0)
#define RCK(x) do { const int rc_ = (x); if (rc_ != BMH_OK) return rc_; } while (0)

// pageable -> pinned on a few threads (one thread copies 15 GB/s: the letters of a million reads were 10 ms of a lane's batch)
void par_memcpy(void *dst, const void *src, size_t n, int n_threads)
{
	const int T = n < (8u << 20) ? 1 : (n_threads < 4 ? (n_threads < 1 ? 1 : n_threads) : 4);
	if (T == 1) { memcpy(dst, src, n); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t) {
		const size_t a = (n * (size_t)t / T) & ~(size_t)63, b = t == T - 1 ? n : (n * (size_t)(t + 1) / T) & ~(size_t)63;
		th.emplace_back([=] { memcpy((uint8_t *)dst + a, (const uint8_t *)src + a, b - a); });
	}
	for (auto &x : th) x.join();
}

// CIGAR / NM / MD of the n_sel records listed in Ln.d_sel, packed on the device (bmh_cigar_pack: an alignment's operations and MD string are a dozen
// bytes, its fixed slots 160); the few alignments that overflow the fixed slots (flag 1: more operations, flag 8: a longer MD) are redone with
// large ones and put in place there too.  to_host: the alignments, their offsets and the packed words go to R (the host formatter's inputs).
int cigars(const aligner_t &A, lane_t &Ln, const int32_t *d_fin, uint64_t n_sel, result_t &R, bool to_host, uint64_t *words_out)
{
	// fixed slots: 16 operations; an MD string of 96 bytes for reads up to 192 bases, half a read's length beyond (a 300 bp read with a dozen mismatches
	// writes 60-100 characters: with 96 every tenth alignment went the way of the overflowed ones)
	// The slots of the redone alignments cannot overflow: an alignment of a read of L bases has at most 2 L + 1 operations (every operation but a
	// deletion consumes a base of the read, a deletion stands between two others) and an MD string of at most two characters per reference base it
	// spans (L + the band at most) -- so one pathological read no longer ends a run of millions
	const int max_cigar = 16, md_cap = Ln.max_read_len <= 192 ? 96 : (int)((Ln.max_read_len / 2 + 31) & ~31u);
	const int MC = std::max(64, 2 * (int)Ln.max_read_len + 2), MD = std::max(1024, (int)((4 * Ln.max_read_len + 2 * (uint32_t)std::max(A.co.w, 0) + 64 + 31) & ~31u));
	R.n_sel = n_sel;
	*words_out = 0;
	if (to_host) { RCK(R.aln.need(8 * (n_sel + 1))); RCK(R.off.need(n_sel + 2)); }
	RCK(Ln.d_aln.need(8 * (n_sel + 1))); RCK(Ln.d_off.need(n_sel + 2)); RCK(Ln.d_packed.need(1));
	if (n_sel == 0) return BMH_OK;
	RCK(Ln.d_cigar.need((size_t)max_cigar * n_sel)); RCK(Ln.d_md.need((size_t)md_cap * n_sel)); RCK(Ln.d_over.need(n_sel + 8)); RCK(Ln.d_sel2.need(n_sel));
	RCK(bmh_cigar_batch(A.idx, Ln.d_reads.p, Ln.d_offs.p, Ln.d_lens.p, d_fin, 16, Ln.d_sel.p, (uint32_t)n_sel, &A.ep, A.co.w, max_cigar, Ln.d_cigar.p, Ln.d_aln.p, md_cap, Ln.d_md.p, Ln.st));
	const size_t wb = bmh_cigar_pack_work((uint32_t)n_sel);
	RCK(Ln.d_work.need(wb));
	// (the counter of the overflowed alignments: the last word of d_over's spare room)
	const int64_t n_over = bmh_cigar_overflowed(Ln.d_aln.p, (uint32_t)n_sel, Ln.d_sel.p, Ln.d_over.p, Ln.d_sel2.p, Ln.d_over.p + n_sel + 4, Ln.st);
	if (n_over < 0) return (int)n_over;
	int64_t words = bmh_cigar_pack_sizes(Ln.d_aln.p, (uint32_t)n_sel, 1, Ln.d_off.p, Ln.d_work.p, Ln.d_work.cap, Ln.st);
	if (words < 0) return (int)words;
	RCK(Ln.d_packed.need((size_t)words + (size_t)n_over * (MC + MD / 4) + 1));
	RCK(bmh_cigar_pack(Ln.d_aln.p, Ln.d_cigar.p, max_cigar, Ln.d_md.p, md_cap, (uint32_t)n_sel, Ln.d_off.p, Ln.d_packed.p, Ln.st));
	if (n_over) {
		const size_t no = (size_t)n_over;
		RCK(Ln.d_cg2.need((size_t)MC * no)); RCK(Ln.d_aln2.need(8 * no)); RCK(Ln.d_md2.need((size_t)MD * no)); RCK(Ln.d_work.need(bmh_cigar_patch_work((uint32_t)no)));
		RCK(bmh_cigar_batch(A.idx, Ln.d_reads.p, Ln.d_offs.p, Ln.d_lens.p, d_fin, 16, Ln.d_sel2.p, (uint32_t)no, &A.ep, A.co.w, MC, Ln.d_cg2.p, Ln.d_aln2.p, MD, Ln.d_md2.p, Ln.st));
		words = bmh_cigar_patch(Ln.d_aln.p, Ln.d_off.p, Ln.d_packed.p, (uint64_t)words, Ln.d_over.p, (uint32_t)no, Ln.d_aln2.p, Ln.d_cg2.p, MC, Ln.d_md2.p, MD, Ln.d_work.p, Ln.d_work.cap, Ln.st);
		if (words < 0) return (int)words;
		if (getenv("BMH_ALIGNER_TRACE")) fprintf(stderr, "[aligner] %zu of %llu alignments redone with large CIGAR / MD slots\n", no, (unsigned long long)n_sel);
	}
	*words_out = (uint64_t)words;
	if (!to_host) return BMH_OK;
	RCK(R.packed.need((size_t)words + 1));
	LCK(hipMemcpyAsync(R.aln.p, Ln.d_aln.p, 32 * n_sel, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipMemcpyAsync(R.off.p, Ln.d_off.p, 4 * (n_sel + 1), hipMemcpyDeviceToHost, Ln.st));
	if (words) LCK(hipMemcpyAsync(R.packed.p, Ln.d_packed.p, 4 * (size_t)words, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipStreamSynchronize(Ln.st));
	for (uint64_t k = 0; k < n_sel; ++k)
		if (R.aln.p[8 * k + 7] & ~2) { bmh_set_error("bmh_aligner_run: bmh_cigar_batch flagged an alignment (CIGAR or MD longer than the buffers)"); return BMH_ECAPACITY; }
	return BMH_OK;
}

// the batch's SAM text written on the device (bmh_sam_text_sizes / _write) and copied into R.text
int text_on_device(const aligner_t &A, lane_t &Ln, const bmh_post_opt_t &po, const int32_t *d_fin, uint32_t n, bool paired, result_t &R)
{
	if (!Ln.ctg_up) {                                              // the sequences' names and offsets, once per lane
		std::vector<char> blob; std::vector<uint32_t> noff;
		for (const std::string &s : A.names) { noff.push_back((uint32_t)blob.size()); blob.insert(blob.end(), s.begin(), s.end()); blob.push_back(0); }
		noff.push_back((uint32_t)blob.size());
		RCK(Ln.d_ctg_names.need(blob.size())); RCK(Ln.d_ctg_name_off.need(noff.size())); RCK(Ln.d_ctg_off.need(A.off.size()));
		LCK(hipMemcpy(Ln.d_ctg_names.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
		LCK(hipMemcpy(Ln.d_ctg_name_off.p, noff.data(), 4 * noff.size(), hipMemcpyHostToDevice));
		LCK(hipMemcpy(Ln.d_ctg_off.p, A.off.data(), 8 * A.off.size(), hipMemcpyHostToDevice));
		Ln.ctg_up = true;
	}
	bmh_sam_dev_t d;
	memset(&d, 0, sizeof(d));
	d.n_reads = n; d.d_names = Ln.d_names.p; d.d_name_off = Ln.d_name_off.p; d.d_reads = Ln.d_reads.p; d.d_offs = Ln.d_offs.p; d.d_lens = Ln.d_lens.p;
	d.n_contigs = A.n_contigs; d.d_contig_names = Ln.d_ctg_names.p; d.d_contig_name_off = Ln.d_ctg_name_off.p; d.d_contig_offset = Ln.d_ctg_off.p;
	d.d_fin = d_fin; d.d_fin_per_read = Ln.d_opr.p; d.d_slot = Ln.d_slot.p; d.d_aln = Ln.d_aln.p; d.d_cig_off = Ln.d_off.p; d.d_packed = Ln.d_packed.p;
	d.d_h_rec = paired ? Ln.d_hrec.p : nullptr; d.d_unflag = paired ? Ln.d_unflag.p : nullptr;
	const size_t wb = bmh_sam_text_work(n);
	RCK(Ln.d_work.need(wb)); RCK(Ln.d_text_off.need((size_t)n + 2));
	const int64_t total = bmh_sam_text_sizes(&po, &d, Ln.d_text_off.p, Ln.d_work.p, Ln.d_work.cap, Ln.st);
	if (total < 0) return (int)total;
	RCK(Ln.d_text.need((size_t)total + 1)); RCK(R.text.need((size_t)total + 1));
	RCK(bmh_sam_text_write(&po, &d, Ln.d_text_off.p, Ln.d_text.p, Ln.d_work.p, Ln.d_work.cap, Ln.st));
	if (total) {
		LCK(hipEventRecord(Ln.ev_c[2], Ln.st));
		LCK(hipMemcpyAsync(R.text.p, Ln.d_text.p, (size_t)total, hipMemcpyDeviceToHost, Ln.st));
		LCK(hipEventRecord(Ln.ev_c[3], Ln.st));
		Ln.copy_bytes[1] += (uint64_t)total; Ln.d2h_marked = true;
	}
	RCK(bmh_sam_text_check(Ln.d_work.p, n, Ln.st));
	R.text_len = (uint64_t)total; R.has_text = true;
	return BMH_OK;
}

// An index with ALT contigs, single-end: the device tail has run as if there were none.  A read WITHOUT a hit on an ALT contig comes out of
// mem_mark_primary_se exactly as without the table (n_pri == n: one marking round, secondary_all = secondary, src/bwamem.c:714-760), so only the
// reads that have one are redone on the host, from their regions, with the table -- same number of records (mem_sort_dedup_patch does not look at
// is_alt), written over the device's; their copies for the CIGAR stage are appended behind the m records of d_fin (dev_index).
int patch_alt_reads(const aligner_t &A, lane_t &Ln, const bmh_dev_jobs_t &dj, const bmh_post_opt_t &po, const uint8_t *codes, const uint64_t *offs64,
                    uint32_t n, uint64_t nr, uint64_t m, int n_threads, result_t &R)
{
	const bool tr = getenv("BMH_ALIGNER_TRACE") != nullptr;
	const double ta = now_s();
	RCK(Ln.h_regs.need(8 * (nr + 1))); RCK(Ln.h_rpr.need(n + 1)); RCK(Ln.h_fr.need(n + 1));
	if (nr) LCK(hipMemcpyAsync(Ln.h_regs.p, Ln.d_regs.p, 32 * (size_t)nr, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipMemcpyAsync(Ln.h_rpr.p, dj.d_regs_per_read, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipMemcpyAsync(Ln.h_fr.p, dj.d_frac_rep, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipStreamSynchronize(Ln.st));
	const double tb = now_s();
	std::vector<uint64_t> rec_off((size_t)n + 1, 0), reg_off((size_t)n + 1, 0);
	for (uint32_t r = 0; r < n; ++r) { rec_off[r + 1] = rec_off[r] + R.opr.p[r]; reg_off[r + 1] = reg_off[r] + Ln.h_rpr.p[r]; }
	if (rec_off[n] != m || reg_off[n] != nr) { bmh_set_error("bmh_aligner_run: internal error: record / region counts do not add up"); return BMH_EINVAL; }
	auto on_alt = [&](const int32_t *q) {
		const int64_t rb = (int64_t)(uint32_t)q[4] | (int64_t)q[5] << 32, re = (int64_t)(uint32_t)q[6] | (int64_t)q[7] << 32;
		const int64_t pos = rb < A.l_pac ? rb : (A.l_pac << 1) - 1 - (re - 1);
		int lo = 0, hi = A.n_contigs;
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.off[(size_t)mid] <= pos) lo = mid; else hi = mid; }
		return A.alt[(size_t)lo] != 0;
	};
	const int T = n >= 65536 ? (n_threads < 8 ? n_threads : 8) : 1;
	std::vector<std::vector<uint32_t>> part((size_t)T);
	auto scan = [&](int t) {
		const uint32_t r0 = (uint32_t)((uint64_t)n * t / T), r1 = (uint32_t)((uint64_t)n * (t + 1) / T);
		for (uint32_t r = r0; r < r1; ++r) {
			bool any = false;
			for (uint64_t k = rec_off[r]; k < rec_off[r + 1] && !any; ++k) any = on_alt(R.fin.p + 16 * k);
			if (any) part[(size_t)t].push_back(r);
			else for (uint64_t k = rec_off[r]; k < rec_off[r + 1]; ++k) R.fin.p[16 * k + 11] = R.fin.p[16 * k + 12];      // the ALT-mode record: [11] = secondary_all (= secondary here)
		}
	};
	if (T == 1) scan(0);
	else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(scan, t); for (auto &x : th) x.join(); }
	std::vector<uint32_t> ids;
	for (const auto &v : part) ids.insert(ids.end(), v.begin(), v.end());
	R.dev_index.clear();
	if (ids.empty()) return bmh_alt_records_device(Ln.d_fin.p, m, nullptr, nullptr, nullptr, nullptr, 0, Ln.st);
	const uint32_t ns = (uint32_t)ids.size();
	const double tc = now_s();
	if (tr) fprintf(stderr, "[aligner] ALT contigs: %u of %u reads have a hit on one and are redone on the host; regions to the host %.1f ms, records looked through %.1f ms\n", ns, n, (tb - ta) * 1e3, (tc - tb) * 1e3);
	std::vector<uint32_t> sub_rpr(ns); std::vector<float> sub_fr(ns); std::vector<uint64_t> sub_offs(ns);
	uint64_t n_sub_regs = 0, n_sub_recs = 0;
	for (uint32_t j = 0; j < ns; ++j) { const uint32_t r = ids[j]; sub_rpr[j] = Ln.h_rpr.p[r]; sub_fr[j] = Ln.h_fr.p[r]; sub_offs[j] = offs64[r]; n_sub_regs += sub_rpr[j]; n_sub_recs += R.opr.p[r]; }
	std::vector<int32_t> sub_regs(8 * (size_t)(n_sub_regs + 1)), sub_out(16 * (size_t)(n_sub_regs + 1));
	std::vector<uint32_t> sub_opr(ns);
	{
		uint64_t w = 0;
		for (uint32_t j = 0; j < ns; ++j) { const uint32_t r = ids[j]; memcpy(&sub_regs[8 * w], Ln.h_regs.p + 8 * reg_off[r], 32 * (size_t)sub_rpr[j]); w += sub_rpr[j]; }
	}
	const int64_t ms = bmh_finalize_regs_ids(&A.co, &A.ep, &po, A.l_pac, A.pac, ns, codes, sub_offs.data(), sub_regs.data(), sub_rpr.data(), sub_fr.data(), A.n_contigs,
	                                         A.n_contigs > 1 ? A.off.data() : nullptr, sub_out.data(), sub_opr.data(), n_threads, ids.data());
	if (ms < 0) return (int)ms;
	if ((uint64_t)ms != n_sub_recs) { bmh_set_error("bmh_aligner_run: internal error: the host tail left %lld records where the device tail left %llu", (long long)ms, (unsigned long long)n_sub_recs); return BMH_EINVAL; }
	std::vector<uint32_t> sub_off((size_t)ns + 1, 0);
	uint64_t w = 0;
	for (uint32_t j = 0; j < ns; ++j) {
		const uint32_t r = ids[j];
		if (sub_opr[j] != R.opr.p[r]) { bmh_set_error("bmh_aligner_run: internal error: read %u has %u records on the host, %u on the device", r, sub_opr[j], R.opr.p[r]); return BMH_EINVAL; }
		memcpy(R.fin.p + 16 * rec_off[r], &sub_out[16 * w], 64 * (size_t)sub_opr[j]);
		sub_off[j] = (uint32_t)w;
		w += sub_opr[j];
	}
	sub_off[ns] = (uint32_t)w;
	// the redone records take their places in the device's array too (a read keeps its number of records: mem_sort_dedup_patch does not look at the table), after every
	// record of the batch has become an ALT-mode record ([11] = secondary_all)
	RCK(Ln.d_alt_ids.need(ns)); RCK(Ln.d_alt_off.need((size_t)ns + 1)); RCK(Ln.d_alt_sub.need(16 * ((size_t)ms + 1)));
	LCK(hipMemcpyAsync(Ln.d_alt_ids.p, ids.data(), 4 * (size_t)ns, hipMemcpyHostToDevice, Ln.st));
	LCK(hipMemcpyAsync(Ln.d_alt_off.p, sub_off.data(), 4 * ((size_t)ns + 1), hipMemcpyHostToDevice, Ln.st));
	if (ms) LCK(hipMemcpyAsync(Ln.d_alt_sub.p, sub_out.data(), 64 * (size_t)ms, hipMemcpyHostToDevice, Ln.st));
	RCK(bmh_alt_records_device(Ln.d_fin.p, m, Ln.d_roff.p, Ln.d_alt_ids.p, Ln.d_alt_off.p, Ln.d_alt_sub.p, ns, Ln.st));
	LCK(hipStreamSynchronize(Ln.st));                              // (ids, sub_off, sub_out are locals)
	if (tr) fprintf(stderr, "[aligner] ALT contigs: host tail of those reads and their way back %.1f ms\n", (now_s() - tc) * 1e3);
	return BMH_OK;
}

// Interleaved pairs with mem_pair / mem_sam_pe's choices on the device (csrc/pair_dev.hip) for the pairs the mate rescue does not touch: the single-end tail
// for every read (mem_sort_dedup_patch, mem_mark_primary_se, mem_reg2sam's selection) with a copy of the regions behind mem_sort_dedup_patch for the host;
// the host computes the insert-size statistics and finds the pairs the rescue touches (bmh_finalize_pairs_split), the device pairs the others meanwhile;
// the host walks the pairs that are left -- the rescued ones, the ones the device hands back -- and their records take their places on the device.
// On return *d_fin_out / Ln.d_opr / Ln.d_hrec / Ln.d_unflag hold the batch's records as bmh_finalize_pairs would have left them; R.m their number.
struct pd_user_t { const aligner_t *A; lane_t *Ln; const bmh_dev_jobs_t *dj; const bmh_post_opt_t *po; bmh_fin_extra_t *ex; uint32_t n; };
int pd_after_pestat(void *u_, const double *pes)
{
	pd_user_t &u = *(pd_user_t *)u_;
	lane_t &Ln = *u.Ln; const aligner_t &A = *u.A;
	RCK(bmh_pair_device(&A.co, &A.ep, u.po, &A.pe, pes, A.l_pac, A.n_contigs, u.ex->d_ctg_off, u.ex->d_logtab, u.ex->n_log, Ln.d_fin.p, Ln.d_opr.p, Ln.d_roff.p,
	                    u.dj->d_frac_rep, u.n, Ln.d_hrec.p, Ln.d_unflag.p, Ln.d_todo.p, Ln.st));
	LCK(hipMemcpyAsync(Ln.h_todo.p, Ln.d_todo.p, u.n / 2, hipMemcpyDeviceToHost, Ln.st));
	return BMH_OK;
}
int pd_before_final(void *u_, const uint8_t **extra)
{
	pd_user_t &u = *(pd_user_t *)u_;
	LCK(hipStreamSynchronize(u.Ln->st));
	*extra = u.Ln->h_todo.p;
	return BMH_OK;
}
int pairs_on_device(const aligner_t &A, lane_t &Ln, const bmh_read_set_t &rs, const bmh_dev_jobs_t &dj, const bmh_post_opt_t &po, const uint8_t *codes, const uint64_t *offs64,
                    uint32_t b0, uint32_t n, uint64_t nr, int n_threads, result_t &R, const int32_t **d_fin_out)
{
	const bool prof = getenv("BMH_PAIR_PROFILE") != nullptr;
	const double t0 = now_s();
	RCK(Ln.d_fin.need(16 * (nr + 1))); RCK(Ln.d_opr.need(n + 1)); RCK(Ln.d_dedup.need(16 * (nr + 1))); RCK(Ln.d_roff.need(n + 1));
	RCK(Ln.d_hrec.need(n + 1)); RCK(Ln.d_unflag.need(n + 1)); RCK(Ln.d_todo.need(n / 2 + 1)); RCK(Ln.h_todo.need(n / 2 + 1));
	bmh_fin_extra_t ex; memset(&ex, 0, sizeof(ex));
	ex.d_dedup_out = Ln.d_dedup.p; ex.d_out_off = Ln.d_roff.p; ex.alt_keep_sub_n = 1;
	const int64_t m1 = bmh_finalize_regs_device_ex(A.idx, &A.co, &A.ep, &po, Ln.d_reads.p, Ln.d_offs.p, n, Ln.d_regs.p, nr, dj.d_regs_per_read, dj.d_frac_rep,
	                                               A.n_contigs, A.n_contigs > 1 ? A.off.data() : nullptr, Ln.d_fin.p, Ln.d_opr.p, Ln.st, &ex);
	if (m1 < 0) return (int)m1;
	RCK(Ln.h_regs.need(16 * ((size_t)m1 + 1))); RCK(Ln.h_rpr.need(n + 1)); RCK(Ln.h_fr.need(n + 1));
	if (m1) LCK(hipMemcpyAsync(Ln.h_regs.p, Ln.d_dedup.p, 64 * (size_t)m1, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipMemcpyAsync(Ln.h_rpr.p, Ln.d_opr.p, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipMemcpyAsync(Ln.h_fr.p, dj.d_frac_rep, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipStreamSynchronize(Ln.st));
	const double t1 = now_s();
	// the host: statistics, the rescue's windows (aligned on the device), the pairs that are the host's
	pd_user_t u = {&A, &Ln, &dj, &po, &ex, n};
	Ln.todo_pairs.resize(n / 2 + 1);
	bmh_pairs_split_t split; memset(&split, 0, sizeof(split));
	split.after_pestat = pd_after_pestat; split.before_final = pd_before_final; split.user = &u; split.todo_pairs = Ln.todo_pairs.data(); split.scratch_slot = &Ln.pair_scratch;
	// knob ALIGNER_RESCUE_DEV=1: the rescue's windows found by a kernel on the regions the device keeps (pair_kernels.hip: rescue_jobs_kernel) instead of the host's
	// first walk -- the same records (tests/test_gpu_parity.py).  Off by default: measured beside each other on one box (profiles/r06_rescue_ab.txt) the host's walk
	// is the faster one by 3 %: the device is the bound of paired reads -> SAM, the walk's 9 ms a batch overlap the other lanes' kernels, the two passes (1 ms each) do not
	const bmh_rescue_in_t rin = {Ln.d_dedup.p, Ln.d_opr.p, Ln.d_roff.p, Ln.d_lens.p, ex.d_ctg_off, A.n_contigs};
	if (bmh_tune("ALIGNER_RESCUE_DEV", 0) != 0) split.rescue_in = &rin;
	uint64_t cap = (uint64_t)m1 + 2ull * n + 4096;                  // (room for every pair: the pairs the device hands back are the ones with the most records)
	int64_t mh = BMH_ECAPACITY;
	RCK(R.opr.need(n + 1)); R.h_rec.resize(n); R.unflag.resize(n);
	for (int attempt = 0; attempt < 4 && mh == BMH_ECAPACITY; ++attempt, cap *= 4) {
		// (a second attempt repeats the device's pair kernel on records it has changed already: they are restored first)
		if (attempt) { bmh_fin_extra_t ex2 = ex; const int64_t m2 = bmh_finalize_regs_device_ex(A.idx, &A.co, &A.ep, &po, Ln.d_reads.p, Ln.d_offs.p, n, Ln.d_regs.p, nr, dj.d_regs_per_read, dj.d_frac_rep,
		                                               A.n_contigs, A.n_contigs > 1 ? A.off.data() : nullptr, Ln.d_fin.p, Ln.d_opr.p, Ln.st, &ex2); if (m2 != m1) return m2 < 0 ? (int)m2 : BMH_EINVAL; }
		RCK(R.fin.need(16 * (size_t)cap));
		mh = bmh_finalize_pairs_split(A.idx, Ln.d_reads.p, Ln.d_offs.p, Ln.st, &A.co, &A.ep, &po, &A.pe, A.l_pac, A.pac, n, codes, offs64, rs.lens + b0,
		                              Ln.h_regs.p, Ln.h_rpr.p, Ln.h_fr.p, A.n_contigs, A.n_contigs > 1 ? A.off.data() : nullptr, A.n_contigs > 1 ? A.len.data() : nullptr,
		                              R.fin.p, cap, R.opr.p, R.h_rec.data(), R.unflag.data(), n_threads, &split);
	}
	if (mh < 0) return (int)mh;
	const double t2 = now_s();
	// the host's records take their places
	const uint32_t nt = (uint32_t)split.n_todo;
	Ln.offh.resize(2 * (size_t)nt + 1);
	{ uint32_t o = 0; for (uint32_t k = 0; k < 2 * nt; ++k) { Ln.offh[k] = o; o += R.opr.p[k]; } if ((int64_t)o != mh) { bmh_set_error("bmh_aligner_run: internal error: the host's pairs left %lld records, their counts add up to %u", (long long)mh, o); return BMH_EINVAL; } }
	RCK(Ln.d_todo_pairs.need(nt + 1)); RCK(Ln.d_oprh.need(2 * (size_t)nt + 1)); RCK(Ln.d_offh.need(2 * (size_t)nt + 1)); RCK(Ln.d_hh.need(2 * (size_t)nt + 1)); RCK(Ln.d_ufh.need(2 * (size_t)nt + 1));
	RCK(Ln.d_finh.need(16 * ((size_t)mh + 1))); RCK(Ln.d_rslot.need(n + 1)); RCK(Ln.d_opr2.need(n + 1)); RCK(Ln.d_hrec2.need(n + 1)); RCK(Ln.d_unflag2.need(n + 1)); RCK(Ln.d_roff2.need(n + 1));
	if (nt) {
		LCK(hipMemcpyAsync(Ln.d_todo_pairs.p, Ln.todo_pairs.data(), 4 * (size_t)nt, hipMemcpyHostToDevice, Ln.st));
		LCK(hipMemcpyAsync(Ln.d_oprh.p, R.opr.p, 8 * (size_t)nt, hipMemcpyHostToDevice, Ln.st));
		LCK(hipMemcpyAsync(Ln.d_offh.p, Ln.offh.data(), 8 * (size_t)nt, hipMemcpyHostToDevice, Ln.st));
		LCK(hipMemcpyAsync(Ln.d_hh.p, R.h_rec.data(), 8 * (size_t)nt, hipMemcpyHostToDevice, Ln.st));
		LCK(hipMemcpyAsync(Ln.d_ufh.p, R.unflag.data(), 8 * (size_t)nt, hipMemcpyHostToDevice, Ln.st));
		if (mh) LCK(hipMemcpyAsync(Ln.d_finh.p, R.fin.p, 64 * (size_t)mh, hipMemcpyHostToDevice, Ln.st));
	}
	RCK(bmh_pair_merge_counts(n, Ln.d_todo_pairs.p, nt, Ln.d_rslot.p, Ln.d_opr.p, Ln.d_hrec.p, Ln.d_unflag.p, Ln.d_oprh.p, Ln.d_hh.p, Ln.d_ufh.p, Ln.d_opr2.p, Ln.d_hrec2.p, Ln.d_unflag2.p, Ln.st));
	const size_t sb = bmh_pair_scan_bytes(n);
	RCK(Ln.d_scan.need(sb));
	RCK(bmh_pair_scan(Ln.d_opr2.p, Ln.d_roff2.p, n, Ln.d_scan.p, Ln.d_scan.cap, Ln.st));
	uint32_t last[2] = {0, 0};
	LCK(hipMemcpyAsync(&last[0], Ln.d_roff2.p + (n - 1), 4, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipMemcpyAsync(&last[1], Ln.d_opr2.p + (n - 1), 4, hipMemcpyDeviceToHost, Ln.st));
	LCK(hipStreamSynchronize(Ln.st));
	const uint64_t m = (uint64_t)last[0] + last[1];
	RCK(Ln.d_fin2.need(16 * (m + 1)));
	RCK(bmh_pair_merge_records(n, Ln.d_rslot.p, Ln.d_fin.p, Ln.d_roff.p, Ln.d_finh.p, Ln.d_offh.p, Ln.d_opr2.p, Ln.d_roff2.p, Ln.d_fin2.p, Ln.st));
	Ln.d_opr.swap(Ln.d_opr2); Ln.d_hrec.swap(Ln.d_hrec2); Ln.d_unflag.swap(Ln.d_unflag2);
	*d_fin_out = Ln.d_fin2.p;
	R.m = m;
	if (prof) {
		uint32_t n1 = 0, n2 = 0, n3 = 0;
		for (uint32_t q = 0; q < n / 2; ++q) { n1 += Ln.h_todo.p[q] == 1; n2 += Ln.h_todo.p[q] == 2; n3 += Ln.h_todo.p[q] == 3; }
		fprintf(stderr, "[pairs] handed back by the device: %u pairs with a score too close to an integer, %u beyond %d hits, %u with a hit on an ALT contig\n", n1, n2, bmh_pair_limit(), n3);
	}
	if (prof) fprintf(stderr, "[pairs] on the device: single-end tail + regions to the host %.1f ms, host (statistics, rescue, %u of %u pairs walked) %.1f ms, merge %.1f ms\n",
	                  (t1 - t0) * 1e3, nt, n / 2, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
	return BMH_OK;
}

// one batch [b0, b1) of the read set on lane Ln -> R
// id0: index of the batch's first read in the run (the tie-break hash of mem_mark_primary_se takes it); src_pinned: rs.ascii is pinned host memory (no staging copy)
// How many lanes may be in the path's device stages (seeding .. regions) at once: knob ALIGNER_GPU_SLOTS, 0 = as many as there are.  Lanes that start together
// stay in step -- all of them seeding, then all of them in the host's walks with the device idle --; with fewer slots than lanes they fall out of step for good.
// (one gate per run -- run_core owns it --, so that two runs in one process with different lane counts do not share a count under different caps)
struct gate_t { std::mutex mu; std::condition_variable cv; int in = 0; int cap = 0; };
struct gate_hold_t {
	gate_t &g; bool held = false;
	explicit gate_hold_t(gate_t &g_) : g(g_)
	{
		if (g.cap <= 0) return;
		std::unique_lock<std::mutex> lk(g.mu);
		g.cv.wait(lk, [&] { return g.in < g.cap; });
		++g.in; held = true;
	}
	void release() { if (!held) return; { std::lock_guard<std::mutex> lk(g.mu); --g.in; } g.cv.notify_all(); held = false; }
	~gate_hold_t() { release(); }
};

int run_batch(const aligner_t &A, lane_t &Ln, const bmh_read_set_t &rs, uint32_t b0, uint32_t b1, int64_t id0, bool src_pinned, bool paired, int n_threads, gate_t &gpu_gate, result_t &R)
{
	const uint32_t n = b1 - b0;
	R.b0 = b0; R.n = n; R.rs = &rs; R.id0 = id0;
	R.dev_index.clear();
	const uint64_t a0 = rs.offs[b0], a1 = rs.offs[b1 - 1] + rs.lens[b1 - 1], nb = a1 - a0;
	if (nb >> 31) { bmh_set_error("bmh_aligner_run: a batch holds 2^31 bases or more (offsets inside a batch are 32-bit)"); return BMH_EINVAL; }
	double t0 = now_s();
	// ---- reads to the device
	RCK(Ln.d_reads.need(nb + 16)); RCK(Ln.d_offs.need(n + 1)); RCK(Ln.d_lens.need(n + 1)); RCK(Ln.h_offs.need(n + 1));
	Ln.max_read_len = 0;
	for (uint32_t r = 0; r < n; ++r) { Ln.h_offs.p[r] = (uint32_t)(rs.offs[b0 + r] - a0); if (rs.lens[b0 + r] > Ln.max_read_len) Ln.max_read_len = rs.lens[b0 + r]; }
	for (hipEvent_t &e : Ln.ev_c) if (!e) LCK(hipEventCreate(&e));
	Ln.d2h_marked = false;
	uint64_t h2d_bytes = nb + 8 * (uint64_t)n;
	if (!src_pinned) RCK(Ln.h_reads.need(nb + 16));
	if (!src_pinned) par_memcpy(Ln.h_reads.p, rs.ascii + a0, nb, n_threads);       // (pageable -> pinned by this lane's threads, then one DMA: the lanes stage side by side)
	LCK(hipEventRecord(Ln.ev_c[0], Ln.st));
	if (src_pinned) LCK(hipMemcpyAsync(Ln.d_reads.p, rs.ascii + a0, nb, hipMemcpyHostToDevice, Ln.st));       // (pinned or registered host memory -- a batch of a read file filled by the loader, a caller's registered buffer: no staging copy)
	else LCK(hipMemcpyAsync(Ln.d_reads.p, Ln.h_reads.p, nb, hipMemcpyHostToDevice, Ln.st));
	LCK(hipMemcpyAsync(Ln.d_offs.p, Ln.h_offs.p, 4 * (size_t)n, hipMemcpyHostToDevice, Ln.st));
	LCK(hipMemcpyAsync(Ln.d_lens.p, rs.lens + b0, 4 * (size_t)n, hipMemcpyHostToDevice, Ln.st));
	// the text is written on the device (bmh_sam_text_*): the names go along
	const bool host_format = getenv("BMH_ALIGNER_HOST_FORMAT") != nullptr;          // (A/B and cross-check: records to the host, text by bmh_format_sam)
	const bool host_select = getenv("BMH_ALIGNER_HOST_SELECT") != nullptr;          // (A/B and cross-check: the host's selection for every batch; implies the host's text)
	const bool text_dev = !host_format && !host_select;
	R.has_text = false; R.text_len = 0;
	if (text_dev) {
		const uint64_t n0 = rs.name_offs[b0], n1 = b1 < rs.n_reads ? rs.name_offs[b1] : rs.n_name_bytes;
		RCK(Ln.h_names.need(n1 - n0 + 1)); RCK(Ln.h_name_off.need(n + 2)); RCK(Ln.d_names.need(n1 - n0 + 1)); RCK(Ln.d_name_off.need(n + 2));
		memcpy(Ln.h_names.p, rs.names + n0, n1 - n0);
		for (uint32_t r = 0; r < n; ++r) Ln.h_name_off.p[r] = rs.name_offs[b0 + r] - n0;
		Ln.h_name_off.p[n] = n1 - n0;
		LCK(hipMemcpyAsync(Ln.d_names.p, Ln.h_names.p, n1 - n0, hipMemcpyHostToDevice, Ln.st));
		LCK(hipMemcpyAsync(Ln.d_name_off.p, Ln.h_name_off.p, 8 * ((size_t)n + 1), hipMemcpyHostToDevice, Ln.st));
		h2d_bytes += (n1 - n0) + 8 * ((uint64_t)n + 1);
	}
	LCK(hipEventRecord(Ln.ev_c[1], Ln.st));
	Ln.copy_bytes[0] += h2d_bytes;
	// ---- seeding
	if (!Ln.sws || n > Ln.sws_reads || nb > Ln.sws_bases) {
		if (Ln.sws) bmh_seed_ws_free(Ln.sws);
		Ln.sws_reads = n + n / 4; Ln.sws_bases = nb + nb / 4;
		const uint64_t occ = 64ull * Ln.sws_reads > (1ull << 16) ? 64ull * Ln.sws_reads : (1ull << 16);
		Ln.sws = bmh_seed_ws_create(Ln.sws_reads, Ln.sws_bases, Ln.sws_bases, occ);          // one candidate per base is the hard upper bound
		if (!Ln.sws) return BMH_ENOMEM;
	}
	const double tg = now_s(); Ln.t[0] += tg - t0;
	gate_hold_t gate(gpu_gate);
	double t1 = now_s(); Ln.t[6] += t1 - tg;                     // (the wait at the gate is its own entry, not part of the upload)
	bmh_seeds_t seeds;
	RCK(bmh_seed_batch(Ln.sws, A.idx, Ln.d_reads.p, Ln.d_offs.p, Ln.d_lens.p, n, A.co.min_seed_len, Ln.st, &seeds));
	double t2 = now_s(); Ln.t[1] += t2 - t1;
	// ---- chains, jobs, extension, regions
	const uint64_t ns = seeds.n_seeds ? seeds.n_seeds : 1;
	if (!Ln.cws || n > Ln.cws_reads || ns > Ln.cws_seeds) {
		if (Ln.cws) bmh_chain_ws_free(Ln.cws);
		Ln.cws_reads = n + n / 4; Ln.cws_seeds = ns + ns / 4 + 1024;
		Ln.cws = bmh_chain_ws_create(Ln.cws_reads, Ln.cws_seeds);
		if (!Ln.cws) return BMH_ENOMEM;
		RCK(bmh_chain_set_materialize(Ln.cws, 0));
		if (A.n_contigs > 1) {
			RCK(bmh_chain_set_contigs(Ln.cws, A.n_contigs, A.off.data(), A.len.data()));
			if (A.has_alt) RCK(bmh_chain_set_alt(Ln.cws, A.n_contigs, A.alt.data()));
		}
	}
	bmh_dev_jobs_t dj;
	// the reference's GPU extension takes the deletion penalties for both gap kinds (src/fastmap.c:417-424)
	bmh_ext_params_t xp = A.ep; xp.o_ins = A.ep.o_del; xp.e_ins = A.ep.e_del;
	// one call for chaining, extension and merge: the seed-rich reads are chained beside the extension of the others (bmh_chain_extend_merge); the
	// regions' array is sized by a guess (the last batch's count, 6 per read to begin with) and the call repeated if the batch has more
	static const bool three_calls = getenv("BMH_ALIGNER_THREE_CALLS") != nullptr;       // (A/B: bmh_chain_batch -> bmh_chain_extend -> bmh_chain_merge)
	if (!three_calls) {
		uint64_t cap = Ln.regs_guess > 6ull * n ? Ln.regs_guess : 6ull * n;
		int rc = BMH_ECAPACITY;
		for (int attempt = 0; attempt < 4 && rc == BMH_ECAPACITY; ++attempt, cap *= 2) {
			RCK(Ln.d_regs.need(8 * (cap + 1)));
			rc = bmh_chain_extend_merge(Ln.cws, &A.co, A.idx, Ln.d_reads.p, Ln.d_offs.p, Ln.d_lens.p, n, &seeds, &xp, Ln.d_regs.p, cap, Ln.st, &dj);
		}
		if (rc != BMH_OK) return rc;
		Ln.regs_guess = dj.n_regs + dj.n_regs / 8;
	} else {
		RCK(bmh_chain_batch(Ln.cws, &A.co, A.idx, Ln.d_reads.p, Ln.d_offs.p, Ln.d_lens.p, n, &seeds, Ln.st, &dj));
		RCK(Ln.d_out3.need(3 * (dj.n_jobs + 1))); RCK(Ln.d_regs.need(8 * (dj.n_regs + 1)));
		RCK(bmh_chain_extend(Ln.cws, &xp, Ln.d_out3.p, nullptr, Ln.st));
		RCK(bmh_chain_merge(Ln.cws, Ln.d_out3.p, Ln.d_regs.p, Ln.st));
	}
	const uint64_t nr = dj.n_regs;
	// What follows the extension -- short kernels that wait on memory: the region tail, the selection, the CIGARs' traceback, the text -- goes to a stream of the
	// highest priority: beside the other lanes' extension kernels, whose waves fill the register files, these are the kernels that should get the slots that
	// become free.  The host waits for the lane's stream first (see bmh_seed_batch: no barrier packet waits in a high-priority queue).
	struct swap_back_t { lane_t &L; bool on; ~swap_back_t() { if (on) std::swap(L.st, L.st_hi); } } sb{Ln, false};
	if (Ln.st_hi) { LCK(hipStreamSynchronize(Ln.st)); std::swap(Ln.st, Ln.st_hi); sb.on = true; }
	else if (gate.held) LCK(hipStreamSynchronize(Ln.st));
	gate.release();
	double t3 = now_s(); Ln.t[2] += t3 - t2;
	bmh_post_opt_t po = A.po; po.id0 = id0;
	const uint8_t *codes = rs.codes + a0;
	std::vector<uint64_t> offs64;                                   // offsets relative to the batch, for the host forms
	auto host_offs = [&]() { if (offs64.empty()) { offs64.resize(n); for (uint32_t r = 0; r < n; ++r) offs64[r] = rs.offs[b0 + r] - a0; } return offs64.data(); };
	const int32_t *d_fin = nullptr;
	bool dev_select = true;                                         // the records that need a CIGAR are chosen on the device (bmh_sam_select_device)
	// ---- the region tail
	if (!paired) {
		int64_t m = -1;
		R.dev_index.clear();
		{
			// (ALT contigs: the table goes along; BMH_ALIGNER_ALT_HOST_PATCH: the device tail without it and the reads that touch an ALT contig redone on the host,
			// the form before the device's second marking round: the cross-check)
			const bool alt_patch = A.has_alt && getenv("BMH_ALIGNER_ALT_HOST_PATCH") != nullptr;
			bmh_post_opt_t po_dev = po; if (alt_patch) po_dev.contig_is_alt = nullptr;
			RCK(Ln.d_fin.need(16 * (nr + 1))); RCK(Ln.d_opr.need(n + 1));
			bmh_fin_extra_t ex; memset(&ex, 0, sizeof(ex));
			if (alt_patch) { RCK(Ln.d_roff.need(n + 1)); ex.d_out_off = Ln.d_roff.p; }      // (the first record of every read: where the host's redone reads go)
			m = bmh_finalize_regs_device_ex(A.idx, &A.co, &A.ep, &po_dev, Ln.d_reads.p, Ln.d_offs.p, n, Ln.d_regs.p, nr, dj.d_regs_per_read, dj.d_frac_rep,
			                                A.n_contigs, A.n_contigs > 1 ? A.off.data() : nullptr, Ln.d_fin.p, Ln.d_opr.p, Ln.st, &ex);
			if (m < 0 && m != BMH_ECAPACITY) return (int)m;
			if (m >= 0) {
				// (the stream is idle: the records go home on the second stream while the selection and the CIGAR kernels run on the first)
				if (!text_dev || alt_patch) {                        // (the host looks the records through for hits on ALT contigs)
					RCK(R.fin.need(16 * (size_t)m + 16)); RCK(R.opr.need(n + 1));
					if (m) LCK(hipMemcpyAsync(R.fin.p, Ln.d_fin.p, 64 * (size_t)m, hipMemcpyDeviceToHost, Ln.st2));
					LCK(hipMemcpyAsync(R.opr.p, Ln.d_opr.p, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st2));
				}
				d_fin = Ln.d_fin.p;
				if (alt_patch) {
					LCK(hipStreamSynchronize(Ln.st2));
					RCK(patch_alt_reads(A, Ln, dj, po, codes, host_offs(), n, nr, (uint64_t)m, n_threads, R));
				}
			}
		}
		if (m < 0) {                                                // the host tail: a read beyond the device tail's fixed limits
			RCK(Ln.h_regs.need(8 * (nr + 1))); RCK(Ln.h_rpr.need(n + 1)); RCK(Ln.h_fr.need(n + 1));
			if (nr) LCK(hipMemcpyAsync(Ln.h_regs.p, Ln.d_regs.p, 32 * (size_t)nr, hipMemcpyDeviceToHost, Ln.st));
			LCK(hipMemcpyAsync(Ln.h_rpr.p, dj.d_regs_per_read, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
			LCK(hipMemcpyAsync(Ln.h_fr.p, dj.d_frac_rep, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
			LCK(hipStreamSynchronize(Ln.st));
			RCK(R.fin.need(16 * (size_t)(nr + 1))); RCK(R.opr.need(n + 1));
			m = bmh_finalize_regs(&A.co, &A.ep, &po, A.l_pac, A.pac, n, codes, host_offs(), Ln.h_regs.p, Ln.h_rpr.p, Ln.h_fr.p, A.n_contigs,
			                      A.n_contigs > 1 ? A.off.data() : nullptr, R.fin.p, R.opr.p, n_threads);
			if (m < 0) return (int)m;
			RCK(Ln.d_fin.need(16 * ((size_t)m + 1))); RCK(Ln.d_opr.need(n + 1));
			if (m) LCK(hipMemcpyAsync(Ln.d_fin.p, R.fin.p, 64 * (size_t)m, hipMemcpyHostToDevice, Ln.st));
			LCK(hipMemcpyAsync(Ln.d_opr.p, R.opr.p, 4 * (size_t)n, hipMemcpyHostToDevice, Ln.st));
			d_fin = Ln.d_fin.p;
		}
		R.m = (uint64_t)m;
	} else {
		bool pe_done = false;
		if (text_dev && !getenv("BMH_ALIGNER_PE_HOST") && !getenv("BMH_ALIGNER_PE_HOST_DEDUP")) {
			int rc_pd = pairs_on_device(A, Ln, rs, dj, po, codes, host_offs(), b0, n, nr, n_threads, R, &d_fin);
			if (rc_pd == BMH_OK) pe_done = true;
			else if (rc_pd != BMH_ECAPACITY) return rc_pd;              // (BMH_ECAPACITY: a read beyond the device tail's fixed limits: the host forms below)
		}
		if (!pe_done) {
		// mem_sort_dedup_patch of every read on the device (the first step of the single-end tail: bmh_dedup_regs_device), the rest of mem_sam_pe on host
		// threads from its records; a batch the device refuses (a read beyond its fixed limits), or BMH_ALIGNER_PE_HOST_DEDUP, takes the regions themselves
		int64_t md = BMH_ECAPACITY;
		if (!getenv("BMH_ALIGNER_PE_HOST_DEDUP")) {
			RCK(Ln.d_fin.need(16 * (nr + 1))); RCK(Ln.d_opr.need(n + 1));
			md = bmh_dedup_regs_device(A.idx, &A.co, &A.ep, &po, Ln.d_reads.p, Ln.d_offs.p, n, Ln.d_regs.p, nr, dj.d_regs_per_read, A.n_contigs, A.n_contigs > 1 ? A.off.data() : nullptr,
			                           Ln.d_fin.p, Ln.d_opr.p, Ln.st);
			if (md < 0 && md != BMH_ECAPACITY) return (int)md;
		}
		const bool deduped = md >= 0;
		const uint64_t n_in = deduped ? (uint64_t)md : nr;
		RCK(Ln.h_regs.need((deduped ? 16 : 8) * (n_in + 1))); RCK(Ln.h_rpr.need(n + 1)); RCK(Ln.h_fr.need(n + 1));
		if (n_in) LCK(hipMemcpyAsync(Ln.h_regs.p, deduped ? Ln.d_fin.p : Ln.d_regs.p, (deduped ? 64 : 32) * (size_t)n_in, hipMemcpyDeviceToHost, Ln.st));
		LCK(hipMemcpyAsync(Ln.h_rpr.p, deduped ? Ln.d_opr.p : dj.d_regs_per_read, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
		LCK(hipMemcpyAsync(Ln.h_fr.p, dj.d_frac_rep, 4 * (size_t)n, hipMemcpyDeviceToHost, Ln.st));
		LCK(hipStreamSynchronize(Ln.st));
		if (getenv("BMH_PAIR_PROFILE")) fprintf(stderr, "[pairs] regions on the host after %.1f ms of the tail\n", (now_s() - t3) * 1e3);
		uint64_t cap = n_in + 2ull * n + 1024;                      // mate rescue adds a few regions per pair
		int64_t m = BMH_ECAPACITY;
		RCK(R.opr.need(n + 1)); R.h_rec.resize(n); R.unflag.resize(n);
		for (int attempt = 0; attempt < 3 && m == BMH_ECAPACITY; ++attempt, cap *= 2) {
			RCK(R.fin.need(16 * (size_t)cap));
			m = (deduped ? bmh_finalize_pairs_deduped : bmh_finalize_pairs_dev)(A.idx, Ln.d_reads.p, Ln.d_offs.p, Ln.st, &A.co, &A.ep, &po, &A.pe, A.l_pac, A.pac, n, codes, host_offs(), rs.lens + b0,
			                           Ln.h_regs.p, Ln.h_rpr.p, Ln.h_fr.p, A.n_contigs, A.n_contigs > 1 ? A.off.data() : nullptr, A.n_contigs > 1 ? A.len.data() : nullptr,
			                           R.fin.p, cap, R.opr.p, R.h_rec.data(), R.unflag.data(), nullptr, n_threads);
		}
		if (m < 0) return (int)m;
		if (getenv("BMH_PAIR_PROFILE")) fprintf(stderr, "[pairs] bmh_finalize_pairs_dev back after %.1f ms of the tail\n", (now_s() - t3) * 1e3);
		R.m = (uint64_t)m;
		RCK(Ln.d_fin.need(16 * ((size_t)m + 1))); RCK(Ln.d_opr.need(n + 1)); RCK(Ln.d_hrec.need(n + 1));
		if (m) LCK(hipMemcpyAsync(Ln.d_fin.p, R.fin.p, 64 * (size_t)m, hipMemcpyHostToDevice, Ln.st));
		LCK(hipMemcpyAsync(Ln.d_opr.p, R.opr.p, 4 * (size_t)n, hipMemcpyHostToDevice, Ln.st));
		LCK(hipMemcpyAsync(Ln.d_hrec.p, R.h_rec.data(), 4 * (size_t)n, hipMemcpyHostToDevice, Ln.st));
		if (text_dev) { RCK(Ln.d_unflag.need(n + 1)); LCK(hipMemcpyAsync(Ln.d_unflag.p, R.unflag.data(), 4 * (size_t)n, hipMemcpyHostToDevice, Ln.st)); }
		d_fin = Ln.d_fin.p;
		}
	}
	double t4 = now_s(); Ln.t[3] += t4 - t3;
	// ---- which records need a CIGAR
	const uint64_t m = R.m;
	uint64_t ns_sel = 0;
	RCK(Ln.d_sel.need(m + 1)); RCK(Ln.h_sel.need(m + 1));
	if (dev_select && !host_select) {
		R.slot.clear();
		RCK(Ln.d_slot.need(m + 1)); RCK(R.slot32.need(m + 1));
		const size_t wb = bmh_sam_select_work(n, m);
		RCK(Ln.d_work.need(wb));
		const int64_t k = bmh_sam_select_device(&po, d_fin, Ln.d_opr.p, paired ? Ln.d_hrec.p : nullptr, n, m, Ln.d_sel.p, Ln.d_slot.p, Ln.d_work.p, Ln.d_work.cap, Ln.st);
		if (k < 0) return (int)k;
		ns_sel = (uint64_t)k;
		if (m && !text_dev) LCK(hipMemcpyAsync(R.slot32.p, Ln.d_slot.p, 4 * (size_t)m, hipMemcpyDeviceToHost, Ln.st2));
	} else {
		LCK(hipStreamSynchronize(Ln.st2));                          // (the records)
		RCK(Ln.h_need.need(m + 1));
		// (reads are independent: ranges of them on host threads -- 3 M records of a million reads were 11 ms on one)
		std::vector<uint64_t> base((size_t)n + 1, 0);
		for (uint32_t r = 0; r < n; ++r) base[r + 1] = base[r] + R.opr.p[r];
		if (base[n] != m) { bmh_set_error("bmh_aligner_run: internal error: %llu records, the per-read counts add up to %llu", (unsigned long long)m, (unsigned long long)base[n]); return BMH_EINVAL; }
		const int T = n >= 65536 ? (n_threads < 8 ? n_threads : 8) : 1;
		std::vector<int64_t> part_rc((size_t)T, 0);
		auto part = [&](int t) {
			const uint32_t r0 = (uint32_t)((uint64_t)n * t / T) & ~1u, r1 = t == T - 1 ? n : (uint32_t)((uint64_t)n * (t + 1) / T) & ~1u;      // (even cuts: pairs stay together)
			if (r1 <= r0) return;
			part_rc[(size_t)t] = paired ? bmh_sam_need_cigar_pe(&po, R.fin.p + 16 * base[r0], R.opr.p + r0, R.h_rec.data() + r0, r1 - r0, Ln.h_need.p + base[r0])
			                            : bmh_sam_need_cigar(&po, R.fin.p + 16 * base[r0], R.opr.p + r0, r1 - r0, Ln.h_need.p + base[r0]);
		};
		if (T == 1) part(0);
		else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(part, t); for (auto &x : th) x.join(); }
		for (int64_t v : part_rc) if (v < 0) return (int)v;
		R.slot.assign(m ? m : 1, -1);
		for (uint64_t k = 0; k < m; ++k) if (Ln.h_need.p[k]) { Ln.h_sel.p[ns_sel] = R.dev_index.empty() ? (uint32_t)k : R.dev_index[k]; R.slot[k] = (int64_t)ns_sel; ++ns_sel; }
		if (ns_sel) LCK(hipMemcpyAsync(Ln.d_sel.p, Ln.h_sel.p, 4 * ns_sel, hipMemcpyHostToDevice, Ln.st));
	}
	double t5 = now_s(); Ln.t[4] += t5 - t4;
	uint64_t words = 0;
	const bool dev_text_now = text_dev && dev_select;
	RCK(cigars(A, Ln, d_fin, ns_sel, R, !dev_text_now, &words));
	if (dev_text_now) RCK(text_on_device(A, Ln, po, d_fin, n, paired, R));
	LCK(hipStreamSynchronize(Ln.st2));
	if (sb.on) LCK(hipStreamSynchronize(Ln.st));                   // (the priority stream is idle before the lane's own takes its place again)
	{
		float ms = 0.f;
		if (hipEventSynchronize(Ln.ev_c[1]) == hipSuccess && hipEventElapsedTime(&ms, Ln.ev_c[0], Ln.ev_c[1]) == hipSuccess) Ln.copy_ms[0] += ms;
		if (Ln.d2h_marked && hipEventSynchronize(Ln.ev_c[3]) == hipSuccess && hipEventElapsedTime(&ms, Ln.ev_c[2], Ln.ev_c[3]) == hipSuccess) Ln.copy_ms[1] += ms;
	}
	Ln.t[5] += now_s() - t5;
	return BMH_OK;
}

}   // namespace

extern "C" {

// (lanes and result objects stay with the aligner between runs: their workspaces and pinned buffers -- gigabytes for million-read batches
// -- cost more to allocate than a batch costs to align)
namespace {
struct fbatch_t {
	hbuf_t<uint8_t> ascii, codes, names; hbuf_t<uint64_t> offs, name_offs; hbuf_t<uint32_t> lens;
	bmh_read_set_t rs; uint32_t index = 0; int64_t id0 = 0;
};
}

struct bmh_aligner {
	std::vector<std::unique_ptr<fbatch_t>> fbatches;      // batch buffers of bmh_aligner_run_fasta (pinned host memory), kept between runs
	aligner_t a;
	std::vector<std::unique_ptr<lane_t>> lanes;
	std::vector<std::unique_ptr<result_t>> pool; int n_results = 0;
	std::vector<std::string> parts;
	int dev = -1;
};

bmh_aligner_t *bmh_aligner_create(const bmh_index_t *idx, const uint8_t *pac, int64_t l_pac, int n_contigs, const char *const *contig_names,
                                  const int32_t *contig_len, const uint8_t *contig_is_alt, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep,
                                  const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe)
{
	if (!idx || !pac || !copt || !ep || !popt || n_contigs < 1 || !contig_names || !contig_len) { bmh_set_error("bmh_aligner_create: null argument"); return nullptr; }
	if (!idx->dev.pac || (int64_t)idx->dev.l_pac != l_pac) { bmh_set_error("bmh_aligner_create: the index must carry the 2-bit reference of l_pac bases"); return nullptr; }
	bmh_aligner *h = new bmh_aligner();
	aligner_t &A = h->a;
	A.idx = idx; A.pac = pac; A.l_pac = l_pac; A.n_contigs = n_contigs;
	int64_t o = 0;
	for (int c = 0; c < n_contigs; ++c) { A.names.emplace_back(contig_names[c]); A.off.push_back(o); A.len.push_back(contig_len[c]); o += contig_len[c]; A.alt.push_back(contig_is_alt ? contig_is_alt[c] : 0); }
	for (const std::string &s : A.names) A.name_ptr.push_back(s.c_str());
	if (o != l_pac) { bmh_set_error("bmh_aligner_create: the sequences' lengths add up to %lld, l_pac is %lld", (long long)o, (long long)l_pac); delete h; return nullptr; }
	A.has_alt = false;
	for (uint8_t v : A.alt) A.has_alt = A.has_alt || v != 0;
	A.co = *copt; A.ep = *ep; A.po = *popt;
	if (popt->rg_id) { A.rg_id = popt->rg_id; A.po.rg_id = A.rg_id.c_str(); }
	if (pe) A.pe = *pe; else bmh_pe_opt_default(&A.pe);
	A.co.contig_is_alt = A.has_alt ? A.alt.data() : nullptr; A.po.contig_is_alt = A.has_alt ? A.alt.data() : nullptr;
	return h;
}

void bmh_aligner_free(bmh_aligner_t *h)
{
	if (!h) return;
	int prev = 0;
	const bool sw = h->dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != h->dev && hipSetDevice(h->dev) == hipSuccess;
	h->lanes.clear(); h->pool.clear(); h->fbatches.clear();                 // (device and pinned buffers, streams: released on their device)
	if (sw) (void)hipSetDevice(prev);
	delete h;
}

// where the batches of a run come from: a read set in host memory cut at given places, or a read file cut and filled by a loader thread
struct batch_t { uint32_t index = 0; const bmh_read_set_t *rs = nullptr; uint32_t b0 = 0, b1 = 0; int64_t id0 = 0; bool pinned = false; void *token = nullptr; };
struct batch_src_t {
	std::function<int(batch_t &)> next;               // 1: a batch (in index order over the callers), 0: no more, < 0: an error (message set); thread-safe, may block
	std::function<void(void *)> release;              // the text of the batch with this token has been written
	std::function<void()> stop;                       // the run failed: stop producing, unblock next()
};

static int run_core(bmh_aligner_t *h, batch_src_t &src, const char *fn, int paired, int n_lanes, int n_threads, bmh_sam_sink_t sink, void *user, bmh_align_stats_t *stats)
{
	const aligner_t &A = h->a;
	if (n_lanes < 1) n_lanes = 1;
	if (n_threads < 1) n_threads = bmh_effective_cpus();
	// half of the lanes in the path's device stages, the others in their tails (gate_t): measured on 2 / 3 / 4 lanes, single-end and paired, never slower than all of them, +5-7 % with 2 and 4
	gate_t gpu_gate; gpu_gate.cap = bmh_tune("ALIGNER_GPU_SLOTS", n_lanes > 1 ? n_lanes / 2 : 0);
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { bmh_set_error("%s: no HIP device", fn); return BMH_ENODEV; }
	const double t_start = now_s();
	const bool trace = getenv("BMH_ALIGNER_TRACE") != nullptr;        // per-batch timeline on stderr
	std::mutex mu; std::condition_variable cv;
	std::map<uint32_t, std::unique_ptr<result_t>> done;             // finished batches waiting for their turn at the writer
	std::vector<std::unique_ptr<result_t>> &pool = h->pool;         // result objects (pinned buffers) not in use: at most n_lanes + 2 exist
	int &n_results = h->n_results;
	uint32_t next_write = 0; int workers_alive = n_lanes;           // (guarded by mu)
	uint64_t n_reads_total = 0;
	int first_rc = BMH_OK; std::string first_err;
	auto fail = [&](int rc, const char *msg) { { std::lock_guard<std::mutex> lk(mu); if (first_rc == BMH_OK) { first_rc = rc; first_err = msg ? msg : ""; } cv.notify_all(); } if (src.stop) src.stop(); };
	std::vector<double> lane_t_sum(8, 0.0);
	double copy_sum[2] = {0, 0}; uint64_t copy_bytes_sum[2] = {0, 0};
	if (h->dev >= 0 && h->dev != dev) { bmh_set_error("%s: the aligner's lanes live on device %d, the current device is %d", fn, h->dev, dev); return BMH_EINVAL; }
	h->dev = dev;
	while ((int)h->lanes.size() < n_lanes) {
		std::unique_ptr<lane_t> ln(new lane_t());
		if (hipStreamCreateWithFlags(&ln->st, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&ln->st2, hipStreamNonBlocking) != hipSuccess) {
			bmh_set_error("%s: hipStreamCreate: %s", fn, hipGetErrorString(hipGetLastError())); return BMH_ENODEV;
		}
		{
			const char *pe = getenv("BMH_ALIGNER_TAIL_PRIO");          // (BMH_ALIGNER_TAIL_PRIO=normal: A/B -- everything on the lane's own stream)
			int prio_lo = 0, prio_hi = 0;
			(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
			if (!(pe && pe[0] == 'n') && hipStreamCreateWithPriority(&ln->st_hi, hipStreamNonBlocking, prio_hi) != hipSuccess) {
				bmh_set_error("%s: hipStreamCreateWithPriority: %s", fn, hipGetErrorString(hipGetLastError())); return BMH_ENODEV;
			}
		}
		h->lanes.push_back(std::move(ln));
	}
	for (auto &ln : h->lanes) { for (double &v : ln->t) v = 0.0; ln->copy_ms[0] = ln->copy_ms[1] = 0.0; ln->copy_bytes[0] = ln->copy_bytes[1] = 0; }
	auto worker = [&](int lane_index) {
		if (hipSetDevice(dev) != hipSuccess) { fail(BMH_ENODEV, "hipSetDevice failed in a worker thread"); return; }
		lane_t &Ln = *h->lanes[(size_t)lane_index];
		for (;;) {
			batch_t bt;
			const int got = src.next(bt);
			if (got < 0) { fail(got, bmh_last_error()); break; }
			if (got == 0) break;
			const uint32_t b = bt.index;
			std::unique_ptr<result_t> R;
			{   // a result object from the pool: no more than n_lanes + 2 batches are ahead of the writer (their text is hundreds of megabytes each); the
				// batch the writer waits for always gets one
				std::unique_lock<std::mutex> lk(mu);
				n_reads_total += bt.b1 - bt.b0;
				cv.wait(lk, [&] { return first_rc != BMH_OK || !pool.empty() || n_results < n_lanes + 2 || b == next_write; });
				if (first_rc != BMH_OK) { if (src.release) src.release(bt.token); break; }
				if (!pool.empty()) { R = std::move(pool.back()); pool.pop_back(); }
				else { R.reset(new result_t()); ++n_results; }
			}
			R->token = bt.token;
			int rc = BMH_OK;
			const double tb0 = now_s();
			if (bt.b1 > bt.b0) rc = run_batch(A, Ln, *bt.rs, bt.b0, bt.b1, bt.id0, bt.pinned, paired != 0, n_threads, gpu_gate, *R);
			else { R->b0 = bt.b0; R->n = 0; R->rs = bt.rs; R->id0 = bt.id0; R->has_text = false; }
			if (trace) fprintf(stderr, "[aligner] lane %d batch %u: %.1f .. %.1f ms\n", lane_index, b, (tb0 - t_start) * 1e3, (now_s() - t_start) * 1e3);
			if (rc != BMH_OK) { if (src.release) src.release(bt.token); fail(rc, bmh_last_error()); break; }
			std::lock_guard<std::mutex> lk(mu);
			done[b] = std::move(R);
			cv.notify_all();
		}
		std::lock_guard<std::mutex> lk(mu);
		for (int k = 0; k < 8; ++k) lane_t_sum[(size_t)k] += Ln.t[k];
		for (int k = 0; k < 2; ++k) { copy_sum[k] += Ln.copy_ms[k]; copy_bytes_sum[k] += Ln.copy_bytes[k]; }
		--workers_alive;                                             // (the writer ends when every worker has: a batch that was taken is in `done` by then)
		cv.notify_all();
	};
	double t_format = 0.0; uint64_t n_bytes = 0; uint32_t n_written = 0;
	auto writer = [&]() {
		std::vector<std::string> &parts = h->parts;                  // (kept with the aligner: their capacity is the text of a batch)
		for (uint32_t b = 0;; ++b) {
			std::unique_ptr<result_t> R;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return first_rc != BMH_OK || done.count(b) || workers_alive == 0; });
				if (first_rc != BMH_OK) return;
				if (!done.count(b)) return;                             // (every batch written)
				R = std::move(done[b]); done.erase(b);
			}
			const bmh_read_set_t *rs = R->rs;
			if (R->n && R->has_text) {                               // written on the device: nothing to format
				const double ts0 = now_s();
				if (R->text_len) { if (sink(user, R->text.p, (size_t)R->text_len) != 0) { fail(BMH_EINVAL, "the sink refused the text"); return; } n_bytes += R->text_len; }
				if (trace) fprintf(stderr, "[aligner] writer batch %u: text of the device, sink %.1f .. %.1f ms\n", b, (ts0 - t_start) * 1e3, (now_s() - t_start) * 1e3);
			} else if (R->n) {
				const double t0 = now_s();
				bmh_post_opt_t po = A.po; po.id0 = R->id0;
				const uint64_t a0 = rs->offs[R->b0];
				std::vector<uint64_t> offs64(R->n), noff(R->n);
				const uint64_t n0 = rs->name_offs[R->b0];
				for (uint32_t r = 0; r < R->n; ++r) { offs64[r] = rs->offs[R->b0 + r] - a0; noff[r] = rs->name_offs[R->b0 + r] - n0; }
				bmh_cigar_src_t cs;
				if (R->slot.empty()) cs.slot32 = R->slot32.p; else cs.slot64 = R->slot.data();
				cs.aln = R->aln.p; cs.packed = R->packed.p; cs.off = R->off.p;
				const bool ok = bmh_format_sam_parts(&po, R->n, (const char *)rs->names + n0, noff.data(), rs->codes + a0, offs64.data(), rs->lens + R->b0, A.n_contigs,
				                                     A.name_ptr.data(), A.off.data(), R->fin.p, R->opr.p, cs,
				                                     paired ? R->h_rec.data() : nullptr, paired ? R->unflag.data() : nullptr, parts);
				if (!ok) { fail(BMH_EINVAL, bmh_last_error()); return; }
				t_format += now_s() - t0;
				const double ts0 = now_s();
				for (const std::string &part : parts) {
					if (part.empty()) continue;
					if (sink(user, part.data(), part.size()) != 0) { fail(BMH_EINVAL, "the sink refused the text"); return; }
					n_bytes += part.size();
				}
				if (trace) fprintf(stderr, "[aligner] writer batch %u: format %.1f .. %.1f ms, sink .. %.1f\n", b, (t0 - t_start) * 1e3, (ts0 - t_start) * 1e3, (now_s() - t_start) * 1e3);
			}
			if (src.release) src.release(R->token);
			R->token = nullptr; R->rs = nullptr;
			std::lock_guard<std::mutex> lk(mu);
			next_write = b + 1; ++n_written;
			pool.push_back(std::move(R));
			cv.notify_all();
		}
	};
	{
		std::vector<std::thread> th;
		for (int k = 0; k < n_lanes; ++k) th.emplace_back(worker, k);
		std::thread wt(writer);
		for (auto &t : th) t.join();
		wt.join();
	}
	if (src.release) for (auto &kv : done) if (kv.second) src.release(kv.second->token);       // (a failed run: what was waiting for the writer)
	n_results = (int)pool.size();                                   // (after a failed run the results that were in flight are gone)
	if (first_rc != BMH_OK) { bmh_set_error("%s", first_err.c_str()); return first_rc; }
	if (stats) {
		stats->n_reads = n_reads_total; stats->n_bytes = n_bytes; stats->n_batches = n_written; stats->n_lanes = n_lanes;
		stats->seconds = now_s() - t_start; stats->format_seconds = t_format;
		stats->h2d_seconds = lane_t_sum[0]; stats->seed_seconds = lane_t_sum[1]; stats->chain_extend_seconds = lane_t_sum[2]; stats->tail_seconds = lane_t_sum[3];
		stats->select_seconds = lane_t_sum[4]; stats->cigar_seconds = lane_t_sum[5]; stats->gate_wait_seconds = lane_t_sum[6];
		stats->h2d_copy_seconds = copy_sum[0] * 1e-3; stats->d2h_copy_seconds = copy_sum[1] * 1e-3; stats->h2d_bytes = copy_bytes_sum[0]; stats->d2h_bytes = copy_bytes_sum[1];
	}
	return BMH_OK;
}

// page-locks (registers) a caller's host buffer so that the device copies straight out of it / releases it; what bmh_aligner_run does with a read set whose
// letters lie in such memory: no staging copy
int bmh_host_pin(void *p, size_t bytes)
{
	if (!p || !bytes) { bmh_set_error("bmh_host_pin: null argument"); return BMH_EINVAL; }
	const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
	if (e != hipSuccess) { bmh_set_error("bmh_host_pin: hipHostRegister of %zu bytes: %s", bytes, hipGetErrorString(e)); (void)hipGetLastError(); return BMH_ENODEV; }
	return BMH_OK;
}
int bmh_host_unpin(void *p)
{
	if (!p) return BMH_OK;
	const hipError_t e = hipHostUnregister(p);
	if (e != hipSuccess) { bmh_set_error("bmh_host_unpin: %s", hipGetErrorString(e)); (void)hipGetLastError(); return BMH_ENODEV; }
	return BMH_OK;
}

int bmh_aligner_run(bmh_aligner_t *h, const bmh_read_set_t *rs, const uint64_t *cuts, uint32_t n_batches, int paired, int n_lanes, int n_threads,
                    bmh_sam_sink_t sink, void *user, bmh_align_stats_t *stats)
{
	if (!h || !rs || !cuts || !sink) { bmh_set_error("bmh_aligner_run: null argument"); return BMH_EINVAL; }
	if (stats) memset(stats, 0, sizeof(*stats));
	if (n_batches == 0 || rs->n_reads == 0) return BMH_OK;
	// (the device path reads ascii / offs / lens only, but the host forms -- the pairs' walks, the host tail after BMH_ECAPACITY, the host formatter -- read every array)
	if (!rs->ascii || !rs->codes || !rs->offs || !rs->lens || !rs->names || !rs->name_offs) { bmh_set_error("bmh_aligner_run: the read set lacks one of ascii / codes / offs / lens / names / name_offs"); return BMH_EINVAL; }
	if (cuts[0] != 0 || cuts[n_batches] != rs->n_reads) { bmh_set_error("bmh_aligner_run: cuts[0] = 0 and cuts[n_batches] = n_reads are required"); return BMH_EINVAL; }
	for (uint32_t b = 0; b < n_batches; ++b)
		if (cuts[b + 1] < cuts[b] || (paired && ((cuts[b + 1] - cuts[b]) & 1)) || cuts[b + 1] - cuts[b] > 0xFFFFFFF0ull) { bmh_set_error("bmh_aligner_run: bad batch cuts"); return BMH_EINVAL; }
	for (uint64_t r = 0; r < rs->n_reads; ++r)
		if (rs->lens[r] > 700) { bmh_set_error("bmh_aligner_run: read %llu has %u bases: reads beyond 700 go through the host job builder (bmh_build_jobs)", (unsigned long long)r, rs->lens[r]); return BMH_EINVAL; }
	if ((uint32_t)n_lanes > n_batches) n_lanes = (int)n_batches;
	// the letters in pinned or registered host memory (hipHostMalloc, hipHostRegister / bmh_host_pin): the batches go to the device straight from there --
	// no staging copy into the lane's pinned buffer on host threads
	bool ascii_pinned = false;
	{
		hipPointerAttribute_t pa; memset(&pa, 0, sizeof(pa));
		const uint8_t *last = rs->ascii + (rs->n_bases ? rs->n_bases - 1 : 0);
		if (hipPointerGetAttributes(&pa, rs->ascii) == hipSuccess && pa.type == hipMemoryTypeHost) {
			hipPointerAttribute_t pb; memset(&pb, 0, sizeof(pb));
			ascii_pinned = hipPointerGetAttributes(&pb, last) == hipSuccess && pb.type == hipMemoryTypeHost;
		}
		(void)hipGetLastError();                                   // (an ordinary pointer: the query fails, and that is the answer)
	}
	std::atomic<uint32_t> next_batch{0};
	batch_src_t src;
	src.next = [&](batch_t &bt) {
		const uint32_t b = next_batch.fetch_add(1);
		if (b >= n_batches) return 0;
		bt.index = b; bt.rs = rs; bt.b0 = (uint32_t)cuts[b]; bt.b1 = (uint32_t)cuts[b + 1]; bt.id0 = (int64_t)cuts[b]; bt.pinned = ascii_pinned; bt.token = nullptr;
		return 1;
	};
	return run_core(h, src, "bmh_aligner_run", paired, n_lanes, n_threads, sink, user, stats);
}

// ---- the same from a read FILE, batch by batch: a loader thread cuts the mapped file the way the reference's bseq_read cuts its stream (reads are added until the
// batch holds at least batch_bases bases -- or exactly batch_reads reads if that is not 0 -- and, paired, an even number of reads; src/bwa.c:48-66) and fills every
// batch into pinned host memory (csrc/reads_io.cpp: bmh_fasta_cut / bmh_fasta_fill, on host threads) while the lanes are on the batches before it: the letters go to
// the device straight from there, nothing of the file is held beyond the batches in flight (n_lanes + 3 of them).

int bmh_aligner_run_fasta(bmh_aligner_t *h, const char *path, uint64_t batch_bases, uint64_t batch_reads, int paired, int n_lanes, int n_threads,
                          bmh_sam_sink_t sink, void *user, bmh_align_stats_t *stats)
{
	if (!h || !path || !sink) { bmh_set_error("bmh_aligner_run_fasta: null argument"); return BMH_EINVAL; }
	if (stats) memset(stats, 0, sizeof(*stats));
	if (batch_bases == 0 && batch_reads == 0) { bmh_set_error("bmh_aligner_run_fasta: batch_bases or batch_reads must be given"); return BMH_EINVAL; }
	if (batch_bases >= (1ull << 31) - 4096) batch_bases = (1ull << 31) - 4096;      // offsets inside a batch are 32-bit
	if (paired && (batch_reads & 1)) --batch_reads;
	if (n_lanes < 1) n_lanes = 1;
	if (n_threads < 1) n_threads = bmh_effective_cpus();
	const int fd = open(path, O_RDONLY);
	if (fd < 0) { bmh_set_error("bmh_aligner_run_fasta: cannot open %s", path); return BMH_EINVAL; }
	struct stat sb;
	if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); bmh_set_error("bmh_aligner_run_fasta: %s is not a regular, seekable file", path); return BMH_EINVAL; }
	const size_t sz = (size_t)sb.st_size;
	if (sz == 0) { close(fd); return BMH_OK; }
	void *m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (m == MAP_FAILED) { bmh_set_error("bmh_aligner_run_fasta: cannot map %s (%zu bytes)", path, sz); return BMH_ENOMEM; }
	(void)madvise(m, sz, MADV_SEQUENTIAL);
	const uint8_t *buf = (const uint8_t *)m;
	const bool need_codes = true;        // nt4 codes: every host form reads them (pairs' walks, ALT reads, a batch the device tail refuses, the host formatter)
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { (void)munmap(m, sz); bmh_set_error("bmh_aligner_run_fasta: no HIP device"); return BMH_ENODEV; }
	// the loader: batches in file order into `ready`; at most n_lanes + 3 batch buffers exist
	std::mutex qm; std::condition_variable qcv;
	std::vector<std::unique_ptr<fbatch_t>> &all = h->fbatches; std::vector<fbatch_t *> free_list; std::map<uint32_t, fbatch_t *> ready;
	for (auto &fb : all) free_list.push_back(fb.get());
	uint32_t next_out = 0; bool eof = false, stopped = false; int load_rc = BMH_OK; std::string load_err;
	const int max_batches = n_lanes + 3;
	const int load_threads = n_threads < 8 ? n_threads : 8;
	auto loader = [&]() {
		if (hipSetDevice(dev) != hipSuccess) { std::lock_guard<std::mutex> lk(qm); load_rc = BMH_ENODEV; load_err = "hipSetDevice failed in the loader thread"; eof = true; qcv.notify_all(); return; }
		size_t p = 0, est = 0; uint32_t index = 0; int64_t id0 = 0;
		while (p < sz) {
			fbatch_t *fb = nullptr;
			{
				std::unique_lock<std::mutex> lk(qm);
				qcv.wait(lk, [&] { return stopped || !free_list.empty() || (int)all.size() < max_batches; });
				if (stopped) break;
				if (!free_list.empty()) { fb = free_list.back(); free_list.pop_back(); }
				else { all.emplace_back(new fbatch_t()); fb = all.back().get(); }
			}
			size_t end = sz; uint64_t nr = 0, nb = 0, nn = 0;
			int rc = bmh_fasta_cut(buf, sz, p, batch_bases, batch_reads, /* even counts, as bseq_read ends its batches */ batch_reads == 0, load_threads, est, &end, &nr, &nb, &nn);
			if (rc == BMH_OK && paired && (nr & 1)) { bmh_set_error("bmh_aligner_run_fasta: an odd number of reads in a paired file"); rc = BMH_EINVAL; }
			if (rc == BMH_OK && (nb >> 31)) { bmh_set_error("bmh_aligner_run_fasta: a batch holds 2^31 bases or more"); rc = BMH_EINVAL; }
			if (rc == BMH_OK && nr) {
				if (fb->ascii.need(nb + 16) != BMH_OK || (need_codes && fb->codes.need(nb + 16) != BMH_OK) || fb->names.need(nn + 16) != BMH_OK || fb->offs.need(nr + 2) != BMH_OK ||
				    fb->name_offs.need(nr + 2) != BMH_OK || fb->lens.need(nr + 2) != BMH_OK) rc = BMH_ENOMEM;
			}
			if (rc == BMH_OK && nr) {
				memset(&fb->rs, 0, sizeof(fb->rs));
				fb->rs.ascii = fb->ascii.p; fb->rs.codes = need_codes ? fb->codes.p : nullptr; fb->rs.names = fb->names.p; fb->rs.offs = fb->offs.p; fb->rs.name_offs = fb->name_offs.p; fb->rs.lens = fb->lens.p;
				rc = bmh_fasta_fill(buf, p, end, nr, nb, nn, load_threads, &fb->rs);
				if (rc == BMH_OK) for (uint64_t r = 0; r < nr; ++r) if (fb->lens.p[r] > 700) { bmh_set_error("bmh_aligner_run_fasta: read %lld has %u bases: reads beyond 700 go through the host job builder (bmh_build_jobs)", (long long)(id0 + (int64_t)r), fb->lens.p[r]); rc = BMH_EINVAL; break; }
			}
			std::lock_guard<std::mutex> lk(qm);
			if (rc != BMH_OK) { load_rc = rc; load_err = bmh_last_error(); free_list.push_back(fb); break; }
			if (nr == 0) { free_list.push_back(fb); p = end; continue; }
			fb->index = index++; fb->id0 = id0; id0 += (int64_t)nr;
			est = end - p; p = end;
			ready[fb->index] = fb;
			qcv.notify_all();
		}
		std::lock_guard<std::mutex> lk(qm);
		eof = true;
		qcv.notify_all();
	};
	batch_src_t src;
	src.next = [&](batch_t &bt) {
		std::unique_lock<std::mutex> lk(qm);
		qcv.wait(lk, [&] { return stopped || ready.count(next_out) || eof; });
		if (stopped) return 0;
		if (!ready.count(next_out)) { if (load_rc != BMH_OK) { bmh_set_error("%s", load_err.c_str()); return load_rc; } return 0; }
		fbatch_t *fb = ready[next_out]; ready.erase(next_out); ++next_out;
		bt.index = fb->index; bt.rs = &fb->rs; bt.b0 = 0; bt.b1 = (uint32_t)fb->rs.n_reads; bt.id0 = fb->id0; bt.pinned = true; bt.token = fb;
		return 1;
	};
	src.release = [&](void *tok) { if (!tok) return; std::lock_guard<std::mutex> lk(qm); free_list.push_back((fbatch_t *)tok); qcv.notify_all(); };
	src.stop = [&]() { std::lock_guard<std::mutex> lk(qm); stopped = true; qcv.notify_all(); };
	std::thread lt(loader);
	const int rc = run_core(h, src, "bmh_aligner_run_fasta", paired, n_lanes, n_threads, sink, user, stats);
	src.stop();
	lt.join();
	(void)munmap(m, sz);
	return rc;
}

}   // extern "C"
