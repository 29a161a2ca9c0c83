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

struct counts_t { uint64_t reads = 0, bases = 0, name_bytes = 0; int bad = 0; };

// nst_nt4_table (src/bntseq.c): A/a 0, C/c 1, G/g 2, T/t 3, everything else 4
struct nt4_table_t {
	uint8_t v[256];
	nt4_table_t() { memset(v, 4, sizeof(v)); v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; }
};
const nt4_table_t NT4;

// walks the lines of buf[b, e): headers and sequence lines must alternate (blank lines, also "\r" alone, are skipped); FILL writes
template <bool FILL>
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
				size_t q = p + 1;
				while (q < le && buf[q] != ' ' && buf[q] != '\t') ++q;
				const size_t nl_ = q - (p + 1);
				if (FILL) { memcpy(o->names + nn, buf + p + 1, nl_); o->names[nn + nl_] = 0; o->name_offs[r] = nn; }
				nn += nl_ + 1;
			} else {
				const size_t L = le - p;
				if (L >> 32) { c.bad = 2; return; }
				if (FILL) {
					memcpy(o->ascii + nb, buf + p, L);
					{ const uint8_t *src = buf + p; uint8_t *dst = o->codes + nb; for (size_t i = 0; i < L; ++i) dst[i] = NT4.v[src[i]]; }
					o->offs[r] = nb; o->lens[r] = (uint32_t)L;
				}
				nb += L; ++r;
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
