// Read files of the host side: the layout the reference's seeding library parses (one '>' header line and one sequence line per
// read, /root/reference/src/GPUSeed/seed_gen.cu:1698-1728; the host takes the same file through kseq / bseq_read, src/bwa.c:48-66)
// into the flat arrays the device path takes: letters back to back (what goes to HBM), nt4 codes (nst_nt4_table: what the host
// tail and the SAM text use), offsets, lengths, names (the header up to the first blank, NUL-terminated, back to back).
// Two passes over the file in memory, both on host threads: count per chunk, then fill at the chunk's offsets.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <chrono>
#include <vector>
#include "bmh_internal.h"
#include "../../include/bwamem_hip.h"

namespace {

// a record's name: the header up to the first blank, without a trailing "/<digit>" (kseq's name, then trim_readno: src/bwa.c:27-31,57,62 -- "read/1" and "read/2"
// of an interleaved file are the same QNAME, and mem_sam_pe insists that the two names of a pair are equal, src/bwamem_pair.c:369)
inline size_t name_len(const uint8_t *buf, size_t s, size_t le)
{
	size_t q = s;
	while (q < le && buf[q] != ' ' && buf[q] != '\t') ++q;
	size_t n = q - s;
	if (n > 2 && buf[s + n - 2] == '/' && buf[s + n - 1] >= '0' && buf[s + n - 1] <= '9') n -= 2;
	return n;
}

struct counts_t { uint64_t reads = 0, bases = 0, name_bytes = 0, max_len = 0; int bad = 0; };

// nst_nt4_table (src/bntseq.c): A/a 0, C/c 1, G/g 2, T/t 3, everything else 4
struct nt4_table_t {
	uint8_t v[256];
	nt4_table_t() { memset(v, 4, sizeof(v)); v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; }
};
const nt4_table_t NT4;

// walks the lines of buf[b, e): headers and sequence lines must alternate (blank lines, also "\r" alone, are skipped); FILL writes
template <bool FILL, bool CODES = true>
void walk(const uint8_t *buf, size_t b, size_t e, counts_t &c, bmh_read_set_t *o, uint64_t r0, uint64_t b0, uint64_t n0)
{
	bool want_hdr = true;
	uint64_t r = r0, nb = b0, nn = n0;
	size_t p = b;
	while (p < e) {
		const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', e - p);
		size_t le = nl ? (size_t)(nl - buf) : e;
		const size_t next = nl ? le + 1 : e;
		if (le > p && buf[le - 1] == '\r') --le;
		if (le > p) {
			const bool hdr = buf[p] == '>';
			if (hdr != want_hdr) { c.bad = 1; return; }
			if (hdr) {
				const size_t nl_ = name_len(buf, p + 1, le);
				if (FILL) { memcpy(o->names + nn, buf + p + 1, nl_); o->names[nn + nl_] = 0; o->name_offs[r] = nn; }
				nn += nl_ + 1;
			} else {
				const size_t L = le - p;
				if (L >> 32) { c.bad = 2; return; }
				if (FILL) {
					memcpy(o->ascii + nb, buf + p, L);
					if (CODES) { const uint8_t *src = buf + p; uint8_t *dst = o->codes + nb; for (size_t i = 0; i < L; ++i) dst[i] = NT4.v[src[i]]; }
					o->offs[r] = nb; o->lens[r] = (uint32_t)L;
				}
				nb += L; ++r;
				if (!FILL && L > c.max_len) c.max_len = L;
			}
			want_hdr = !hdr;
		}
		p = next;
	}
	if (!want_hdr) { c.bad = 1; return; }                  // a header without its sequence line
	c.reads = r - r0; c.bases = nb - b0; c.name_bytes = nn - n0;
}

} // namespace

extern "C" int bmh_reads_load_fasta(const char *path, int n_threads, bmh_read_set_t *out)
{
	if (!path || !out) { bmh_set_error("bmh_reads_load_fasta: null argument"); return BMH_EINVAL; }
	memset(out, 0, sizeof(*out));
	const bool prof = getenv("BMH_IO_PROFILE") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto lap = [&](const char *what, std::chrono::steady_clock::time_point &t) { if (prof) { const auto n = now(); fprintf(stderr, "[reads_io] %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count()); t = n; } };
	auto tp = now();
	// the file is mapped, not copied (its pages come straight from the page cache; the two passes below read it on host threads); it has to be
	// a regular file (a FIFO or a process substitution cannot be mapped or cut at headers)
	const int fd = open(path, O_RDONLY);
	if (fd < 0) { bmh_set_error("bmh_reads_load_fasta: cannot open %s", path); return BMH_EINVAL; }
	struct stat sb;
	if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); bmh_set_error("bmh_reads_load_fasta: %s is not a regular, seekable file", path); return BMH_EINVAL; }
	const size_t sz = (size_t)sb.st_size;
	const uint8_t *buf = (const uint8_t *)"";
	if (sz) {
		void *m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
		if (m == MAP_FAILED) { close(fd); bmh_set_error("bmh_reads_load_fasta: cannot map %s (%zu bytes)", path, sz); return BMH_ENOMEM; }
		(void)madvise(m, sz, MADV_SEQUENTIAL);
		buf = (const uint8_t *)m;
	}
	close(fd);
	auto unmap = [&]() { if (sz) (void)munmap((void *)buf, sz); };
	lap("read", tp);
	// chunks that begin at a header: the first '>' that follows a newline at or behind the nominal cut
	unsigned T = n_threads > 0 ? (unsigned)n_threads : (unsigned)bmh_effective_cpus();      // (the CPUs the process is granted, not the ones the machine shows)
	if (T == 0) T = 1;
	if (T > 32) T = 32;                                    // (memory-bound beyond a few threads; the host may show hundreds of hardware threads)
	if (sz < (1u << 20)) T = 1;
	std::vector<size_t> cut(T + 1, sz);
	cut[0] = 0;
	for (unsigned t = 1; t < T; ++t) {
		size_t p = sz / T * t;
		if (p < cut[t - 1]) p = cut[t - 1];
		size_t c = sz;
		while (p < sz) {
			const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', sz - p);
			if (!nl) break;
			p = (size_t)(nl - buf) + 1;
			if (p < sz && buf[p] == '>') { c = p; break; }
		}
		cut[t] = c;
	}
	std::vector<counts_t> cnt(T);
	auto run = [&](auto fn) {
		if (T == 1) { fn(0u); return; }
		std::vector<std::thread> th;
		for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t);
		for (auto &x : th) x.join();
	};
	run([&](unsigned t) { walk<false>(buf, cut[t], cut[t + 1], cnt[t], nullptr, 0, 0, 0); });
	lap("count", tp);
	uint64_t nr = 0, nb = 0, nn = 0;
	std::vector<uint64_t> r0(T), b0(T), n0(T);
	for (unsigned t = 0; t < T; ++t) {
		if (cnt[t].bad) {
			unmap();
			bmh_set_error(cnt[t].bad == 2 ? "bmh_reads_load_fasta: a sequence line of 2^32 bases or more" : "reads file: expected alternating '>' header and sequence lines");
			return BMH_EINVAL;
		}
		r0[t] = nr; b0[t] = nb; n0[t] = nn;
		nr += cnt[t].reads; nb += cnt[t].bases; nn += cnt[t].name_bytes;
	}
	out->n_reads = nr; out->n_bases = nb; out->n_name_bytes = nn;
	// (no MADV_HUGEPAGE on the two large arrays: with the kernel's defrag = madvise the fill pass sometimes stalled for a second in
	// direct compaction -- 60 ms or 1.7 s from one call to the next)
	out->ascii = (uint8_t *)malloc(nb + 1); out->codes = (uint8_t *)malloc(nb + 1);
	out->offs = (uint64_t *)malloc(8 * (nr + 1)); out->lens = (uint32_t *)malloc(4 * (nr + 1));
	out->names = (uint8_t *)malloc(nn + 1); out->name_offs = (uint64_t *)malloc(8 * (nr + 1));
	if (!out->ascii || !out->codes || !out->offs || !out->lens || !out->names || !out->name_offs) {
		unmap(); bmh_reads_free(out);
		bmh_set_error("bmh_reads_load_fasta: out of memory"); return BMH_ENOMEM;
	}
	out->ascii[nb] = out->codes[nb] = 0; out->names[nn] = 0;
	lap("alloc", tp);
	run([&](unsigned t) { counts_t c; walk<true>(buf, cut[t], cut[t + 1], c, out, r0[t], b0[t], n0[t]); });
	lap("fill", tp);
	unmap();
	return BMH_OK;
}

extern "C" void bmh_reads_free(bmh_read_set_t *r)
{
	if (!r) return;
	free(r->ascii); free(r->codes); free(r->offs); free(r->lens); free(r->names); free(r->name_offs);
	memset(r, 0, sizeof(*r));
}

// ---- a read file taken batch by batch (csrc/align_pipeline.hip: bmh_aligner_run_fasta).  The reference reads its batches one after the other while the
// previous one is aligned (bseq_read, src/bwa.c:48-66, inside kt_pipeline): reads are added until the batch holds at least chunk bases and an even number
// of reads.  Here the mapped file is cut the same way -- counted on host threads over a window of the expected size, the exact end found by walking the
// chunk in which the count is reached -- and a batch is then filled into the caller's arrays (pinned memory: the letters go to the device from there).
namespace {

// first record start (a '>' at the beginning of a line) at or behind p
size_t next_record(const uint8_t *buf, size_t p, size_t sz)
{
	if (p == 0) return 0;
	if (p >= sz) return sz;
	if (buf[p - 1] == '\n' && buf[p] == '>') return p;
	while (p < sz) {
		const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', sz - p);
		if (!nl) return sz;
		p = (size_t)(nl - buf) + 1;
		if (p < sz && buf[p] == '>') return p;
	}
	return sz;
}

// walks records from b until the batch is complete (bases >= want_bases, or reads == want_reads when that is not 0; an even count when `even`) or e is reached;
// reads0 / bases0: what the batch holds before b.  Returns the offset behind the last record taken; bad: the lines do not alternate
size_t walk_until(const uint8_t *buf, size_t b, size_t e, uint64_t reads0, uint64_t bases0, uint64_t want_bases, uint64_t want_reads, bool even, counts_t &c, bool *complete)
{
	bool want_hdr = true;
	uint64_t r = reads0, nb = bases0, nn = 0;
	size_t p = b, last = b;
	*complete = false;
	while (p < e) {
		const uint8_t *nl = (const uint8_t *)memchr(buf + p, '\n', e - p);
		size_t le = nl ? (size_t)(nl - buf) : e;
		const size_t next = nl ? le + 1 : e;
		if (le > p && buf[le - 1] == '\r') --le;
		if (le > p) {
			const bool hdr = buf[p] == '>';
			if (hdr != want_hdr) { c.bad = 1; return last; }
			if (hdr) nn += name_len(buf, p + 1, le) + 1;
			else {
				nb += le - p; ++r; last = next;
				const bool full = want_reads ? r >= want_reads : nb >= want_bases;
				if (full && (!even || !(r & 1))) { *complete = true; c.reads = r - reads0; c.bases = nb - bases0; c.name_bytes = nn; return last; }
			}
			want_hdr = !hdr;
		}
		p = next;
	}
	if (!want_hdr) { c.bad = 1; return last; }
	c.reads = r - reads0; c.bases = nb - bases0; c.name_bytes = nn;
	return e;
}

}   // namespace

// The end of the batch that starts at offset p of the mapped file (p at a record start): *end, and what it holds.  est_bytes: the caller's guess of its size in
// the file (0: none).  BMH_OK, or BMH_EINVAL (lines that do not alternate).  A batch that ends with the file may be short (and odd).
int bmh_fasta_cut(const uint8_t *buf, size_t sz, size_t p, uint64_t want_bases, uint64_t want_reads, bool even, int n_threads, size_t est_bytes,
                  size_t *end, uint64_t *n_reads, uint64_t *n_bases, uint64_t *n_name_bytes)
{
	unsigned T = n_threads > 0 ? (unsigned)n_threads : 1u;
	if (T > 16) T = 16;
	size_t window = est_bytes ? est_bytes + est_bytes / 16 + (1u << 16) : (size_t)(want_reads ? want_reads * 200 : want_bases + want_bases / 4) + (1u << 16);
	for (;;) {
		const size_t q = p + window >= sz ? sz : next_record(buf, p + window, sz);
		const unsigned Tw = (q - p) < (1u << 20) ? 1u : T;
		std::vector<size_t> cut(Tw + 1, q);
		cut[0] = p;
		for (unsigned t = 1; t < Tw; ++t) { size_t c = next_record(buf, p + (q - p) / Tw * t, q); if (c < cut[t - 1]) c = cut[t - 1]; cut[t] = c > q ? q : c; }
		std::vector<counts_t> cnt(Tw);
		if (Tw == 1) walk<false>(buf, cut[0], cut[1], cnt[0], nullptr, 0, 0, 0);
		else { std::vector<std::thread> th; for (unsigned t = 0; t < Tw; ++t) th.emplace_back([&, t] { walk<false>(buf, cut[t], cut[t + 1], cnt[t], nullptr, 0, 0, 0); }); for (auto &x : th) x.join(); }
		uint64_t r = 0, b = 0, nn = 0;
		for (unsigned t = 0; t < Tw; ++t) {
			if (cnt[t].bad) { bmh_set_error(cnt[t].bad == 2 ? "reads file: a sequence line of 2^32 bases or more" : "reads file: expected alternating '>' header and sequence lines"); return BMH_EINVAL; }
			const bool reached = want_reads ? r + cnt[t].reads >= want_reads : b + cnt[t].bases >= want_bases;
			if (reached) {                                          // the batch ends inside this chunk (or, for an even count, a record into the next ones)
				counts_t c; bool complete = false;
				const size_t e = walk_until(buf, cut[t], q, r, b, want_bases, want_reads, even, c, &complete);
				if (c.bad) { bmh_set_error("reads file: expected alternating '>' header and sequence lines"); return BMH_EINVAL; }
				if (complete || q == sz) { *end = complete ? e : sz; *n_reads = r + c.reads; *n_bases = b + c.bases; *n_name_bytes = nn + c.name_bytes; return BMH_OK; }
				break;                                               // (ran out of window behind the threshold: a larger window)
			}
			r += cnt[t].reads; b += cnt[t].bases; nn += cnt[t].name_bytes;
		}
		if (q == sz) { *end = sz; *n_reads = r; *n_bases = b; *n_name_bytes = nn; return BMH_OK; }
		window *= 2;
	}
}

// fills the batch [p, end) (bmh_fasta_cut's numbers) into o's arrays -- the caller's, large enough: ascii / codes n_bases + 1, offs / lens / name_offs n_reads + 1,
// names n_name_bytes + 1 -- on n_threads host threads; o->codes may be NULL (no nt4 codes wanted)
int bmh_fasta_fill(const uint8_t *buf, size_t p, size_t end, uint64_t n_reads, uint64_t n_bases, uint64_t n_name_bytes, int n_threads, bmh_read_set_t *o)
{
	unsigned T = n_threads > 0 ? (unsigned)n_threads : 1u;
	if (T > 16) T = 16;
	if (end - p < (1u << 20)) T = 1;
	std::vector<size_t> cut(T + 1, end);
	cut[0] = p;
	for (unsigned t = 1; t < T; ++t) { size_t c = next_record(buf, p + (end - p) / T * t, end); if (c < cut[t - 1]) c = cut[t - 1]; cut[t] = c > end ? end : c; }
	std::vector<counts_t> cnt(T);
	auto run = [&](auto fn) { if (T == 1) { fn(0u); return; } std::vector<std::thread> th; for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t); for (auto &x : th) x.join(); };
	run([&](unsigned t) { walk<false>(buf, cut[t], cut[t + 1], cnt[t], nullptr, 0, 0, 0); });
	std::vector<uint64_t> r0(T), b0(T), n0(T);
	uint64_t nr = 0, nb = 0, nn = 0;
	for (unsigned t = 0; t < T; ++t) { if (cnt[t].bad) { bmh_set_error("reads file: expected alternating '>' header and sequence lines"); return BMH_EINVAL; } r0[t] = nr; b0[t] = nb; n0[t] = nn; nr += cnt[t].reads; nb += cnt[t].bases; nn += cnt[t].name_bytes; }
	if (nr != n_reads || nb != n_bases || nn != n_name_bytes) { bmh_set_error("reads file: internal error: a batch counted twice gave different sizes"); return BMH_EINVAL; }
	o->n_reads = nr; o->n_bases = nb; o->n_name_bytes = nn;
	uint8_t *codes = o->codes;
	bmh_read_set_t w = *o;
	run([&](unsigned t) { counts_t c; if (codes) walk<true, true>(buf, cut[t], cut[t + 1], c, &w, r0[t], b0[t], n0[t]); else walk<true, false>(buf, cut[t], cut[t + 1], c, &w, r0[t], b0[t], n0[t]); });
	return BMH_OK;
}

// reads, bases, name bytes and the longest read of a read file, without loading it (one counting pass of the mapped file on host threads): out[4]
extern "C" int bmh_fasta_scan(const char *path, int n_threads, uint64_t *out)
{
	if (!path || !out) { bmh_set_error("bmh_fasta_scan: null argument"); return BMH_EINVAL; }
	out[0] = out[1] = out[2] = out[3] = 0;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) { bmh_set_error("bmh_fasta_scan: cannot open %s", path); return BMH_EINVAL; }
	struct stat sb;
	if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); bmh_set_error("bmh_fasta_scan: %s is not a regular, seekable file", path); return BMH_EINVAL; }
	const size_t sz = (size_t)sb.st_size;
	if (sz == 0) { close(fd); return BMH_OK; }
	void *m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (m == MAP_FAILED) { bmh_set_error("bmh_fasta_scan: cannot map %s (%zu bytes)", path, sz); return BMH_ENOMEM; }
	(void)madvise(m, sz, MADV_SEQUENTIAL);
	const uint8_t *buf = (const uint8_t *)m;
	unsigned T = n_threads > 0 ? (unsigned)n_threads : (unsigned)bmh_effective_cpus();
	if (T == 0) T = 1;
	if (T > 32) T = 32;
	if (sz < (1u << 20)) T = 1;
	std::vector<size_t> cut(T + 1, sz);
	cut[0] = 0;
	for (unsigned t = 1; t < T; ++t) { size_t c = next_record(buf, sz / T * t, sz); if (c < cut[t - 1]) c = cut[t - 1]; cut[t] = c; }
	std::vector<counts_t> cnt(T);
	if (T == 1) walk<false>(buf, cut[0], cut[1], cnt[0], nullptr, 0, 0, 0);
	else { std::vector<std::thread> th; for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] { walk<false>(buf, cut[t], cut[t + 1], cnt[t], nullptr, 0, 0, 0); }); for (auto &x : th) x.join(); }
	int rc = BMH_OK;
	for (unsigned t = 0; t < T; ++t) {
		if (cnt[t].bad) { bmh_set_error(cnt[t].bad == 2 ? "reads file: a sequence line of 2^32 bases or more" : "reads file: expected alternating '>' header and sequence lines"); rc = BMH_EINVAL; break; }
		out[0] += cnt[t].reads; out[1] += cnt[t].bases; out[2] += cnt[t].name_bytes; if (cnt[t].max_len > out[3]) out[3] = cnt[t].max_len;
	}
	(void)munmap(m, sz);
	return rc;
}
