// SAM records of single-end reads from the outputs of bmh_finalize_regs and bmh_cigar_batch:
//   mem_aln2sam   /root/reference/src/bwamem.c:1506-1683   field order, hard clips on every record of a read after its first,
//                                                          SEQ/QUAL of secondary records, NM MD AS XS SA XA tags
//   mem_gen_alt   src/bwamem_extra.c:97-150                the XA tag: secondary hits within XA_drop_ratio of their primary,
//                                                          listed when there are at most max_XA_hits of them
//   mem_reg2sam   src/bwamem.c:1721-1770                   the unmapped record when nothing is reported
// Host code, like the reference's.  Pairing (bwamem_pair.c) and ALT contigs are not modelled; reads come without qualities
// (the seeding library reads FASTA only, src/GPUSeed/seed_gen.cu:1698-1728).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "bmh_internal.h"

namespace {

inline void put_int(std::string &s, long long v) { char b[24]; snprintf(b, sizeof(b), "%lld", v); s += b; }

struct Rec { const int32_t *fin; const int32_t *aln; const uint32_t *cigar; const char *md; };

inline long long aln_pos(const int32_t *a) { return (long long)(uint32_t)a[0] | (long long)a[1] << 32; }

int rid_of(int n_contigs, const int64_t *off, int64_t pos)
{
	if (n_contigs <= 1) return 0;
	int lo = 0, hi = n_contigs;                    // last sequence starting at or before pos
	while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= pos) lo = mid; else hi = mid; }
	return lo;
}

void put_cigar(std::string &s, const Rec &r, bool hard)
{
	const int n = r.aln[3];
	for (int i = 0; i < n; ++i) {
		int c = (int)(r.cigar[i] & 0xf);
		if (hard && (c == 3 || c == 4)) c = 4;
		put_int(s, r.cigar[i] >> 4);
		s += "MIDSH"[c];
	}
}

} // namespace

extern "C" void bmh_free(void *p) { free(p); }

// need[i] = 1 for every record of bmh_finalize_regs that must go through bmh_cigar_batch before formatting: the reported
// ones and the XA candidates (mem_gen_alt's two passes).  Returns their number.
extern "C" int64_t bmh_sam_need_cigar(const bmh_post_opt_t *po, const int32_t *fin, const uint32_t *fin_per_read, uint32_t n_reads, uint8_t *need)
{
	if (!po || !fin_per_read || !need || (n_reads && !fin)) { bmh_set_error("bmh_sam_need_cigar: null argument"); return BMH_EINVAL; }
	int64_t total = 0; uint64_t base = 0;
	std::vector<int> cnt;
	for (uint32_t r = 0; r < n_reads; ++r) {
		const int n = (int)fin_per_read[r];
		const int32_t *a = fin + 16 * base;
		cnt.assign(n, 0);
		for (int i = 0; i < n; ++i) need[base + i] = a[16 * i + 15] ? 1 : 0;
		if (!po->flag_all) {
			auto pri = [&](int i) { const int k = a[16 * i + 12]; return (k >= 0 && a[16 * i + 1] >= a[16 * k + 1] * po->XA_drop_ratio) ? k : -1; };
			for (int i = 0; i < n; ++i) { const int k = pri(i); if (k >= 0) ++cnt[k]; }
			for (int i = 0; i < n; ++i) { const int k = pri(i); if (k >= 0 && cnt[k] <= po->max_XA_hits) need[base + i] = 1; }
		}
		for (int i = 0; i < n; ++i) total += need[base + i];
		base += n;
	}
	return total;
}

// slot[i] = index of record i in the bmh_cigar_batch outputs (aln [..][8], cigar [..][max_cigar], md [..][md_cap]) or -1.
// names / contig_names: arrays of C strings.  reads: nt4 codes.  Returns malloc'd text (bmh_free), *len_out its length.
extern "C" char *bmh_format_sam(const bmh_post_opt_t *po, uint32_t n_reads, const char *const *names, const uint8_t *reads,
                                const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                                const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const int64_t *slot,
                                const int32_t *aln, const uint32_t *cigar, int max_cigar, const char *md, int md_cap, size_t *len_out)
{
	if (!po || !names || !reads || !read_offs || !read_lens || !contig_names || !fin_per_read || !len_out || (n_contigs > 1 && !contig_offset)) {
		bmh_set_error("bmh_format_sam: null argument"); return nullptr;
	}
	std::string out;
	out.reserve((size_t)n_reads * 400);
	uint64_t base = 0;
	std::vector<int> cnt, list;
	std::vector<std::string> xa;
	for (uint32_t r = 0; r < n_reads; ++r) {
		const int n = (int)fin_per_read[r];
		const int32_t *a = fin + 16 * base;
		auto rec = [&](int i) {
			Rec x; x.fin = a + 16 * i;
			const int64_t s = slot[base + i];
			x.aln = s >= 0 ? aln + 8 * s : nullptr; x.cigar = s >= 0 ? cigar + (size_t)max_cigar * s : nullptr; x.md = (s >= 0 && md) ? md + (size_t)md_cap * s : "";
			return x;
		};
		// XA strings per primary (mem_gen_alt)
		xa.assign(n, std::string());
		if (!po->flag_all) {
			cnt.assign(n, 0);
			auto pri = [&](int i) { const int k = a[16 * i + 12]; return (k >= 0 && a[16 * i + 1] >= a[16 * k + 1] * po->XA_drop_ratio) ? k : -1; };
			for (int i = 0; i < n; ++i) { const int k = pri(i); if (k >= 0) ++cnt[k]; }
			for (int i = 0; i < n; ++i) {
				const int k = pri(i);
				if (k < 0 || cnt[k] > po->max_XA_hits) continue;
				const Rec x = rec(i);
				if (!x.aln) { bmh_set_error("bmh_format_sam: record %d of read %u has no CIGAR (see bmh_sam_need_cigar)", i, r); return nullptr; }
				const long long pos = aln_pos(x.aln);
				const int rid = rid_of(n_contigs, contig_offset, pos);
				std::string &s = xa[k];
				s += contig_names[rid]; s += ','; s += "+-"[x.aln[2] ? 1 : 0]; put_int(s, pos - (n_contigs > 1 ? contig_offset[rid] : 0) + 1); s += ',';
				put_cigar(s, x, false);
				s += ','; put_int(s, x.aln[4]); s += ';';
			}
		}
		list.clear();
		for (int i = 0; i < n; ++i) if (a[16 * i + 15]) list.push_back(i);
		const uint8_t *seq = reads + read_offs[r];
		const int l_seq = (int)read_lens[r];
		if (list.empty()) {                                   // unmapped record
			out += names[r]; out += "\t4\t*\t0\t0\t*\t*\t0\t0\t";
			for (int i = 0; i < l_seq; ++i) out += "ACGTN"[seq[i] > 4 ? 4 : seq[i]];
			out += "\t*\tAS:i:0\tXS:i:0\n";
			base += n;
			continue;
		}
		for (size_t which = 0; which < list.size(); ++which) {
			const int i = list[which];
			const Rec x = rec(i);
			if (!x.aln) { bmh_set_error("bmh_format_sam: record %d of read %u has no CIGAR (see bmh_sam_need_cigar)", i, r); return nullptr; }
			const int flag = (x.aln[2] ? 0x10 : 0) | x.fin[14];
			const long long pos = aln_pos(x.aln);
			const int rid = rid_of(n_contigs, contig_offset, pos);
			const bool hard = which > 0;
			out += names[r]; out += '\t'; put_int(out, flag); out += '\t';
			out += contig_names[rid]; out += '\t'; put_int(out, pos - (n_contigs > 1 ? contig_offset[rid] : 0) + 1); out += '\t';
			put_int(out, x.fin[13]); out += '\t';
			if (x.aln[3]) put_cigar(out, x, hard); else out += '*';
			out += "\t*\t0\t0\t";
			if (flag & 0x100) out += "*\t*";
			else {
				int qb = 0, qe = l_seq;
				const int nc = x.aln[3];
				if (nc && hard) {                                 // hard-clipped records print only the aligned part
					const int c0 = (int)(x.cigar[0] & 0xf), c1 = (int)(x.cigar[nc - 1] & 0xf);
					if (!x.aln[2]) { if (c0 == 3 || c0 == 4) qb += x.cigar[0] >> 4; if (c1 == 3 || c1 == 4) qe -= x.cigar[nc - 1] >> 4; }
					else { if (c0 == 3 || c0 == 4) qe -= x.cigar[0] >> 4; if (c1 == 3 || c1 == 4) qb += x.cigar[nc - 1] >> 4; }
				}
				if (!x.aln[2]) for (int k = qb; k < qe; ++k) out += "ACGTN"[seq[k] > 4 ? 4 : seq[k]];
				else for (int k = qe - 1; k >= qb; --k) out += "TGCAN"[seq[k] > 4 ? 4 : seq[k]];
				out += "\t*";
			}
			if (x.aln[3]) { out += "\tNM:i:"; put_int(out, x.aln[4]); out += "\tMD:Z:"; out += x.md; }
			if (x.fin[1] >= 0) { out += "\tAS:i:"; put_int(out, x.fin[1]); }
			if (!(flag & 0x100)) {                               // sub is not printed for secondary records (q->sub = -1)
				if (x.fin[10] >= 0) { out += "\tXS:i:"; put_int(out, x.fin[10]); }
				bool other = false;
				for (size_t j = 0; j < list.size(); ++j) if (j != which && !(a[16 * list[j] + 14] & 0x100)) other = true;
				if (other) {
					out += "\tSA:Z:";
					for (size_t j = 0; j < list.size(); ++j) {
						if (j == which || (a[16 * list[j] + 14] & 0x100)) continue;
						const Rec y = rec(list[j]);
						const long long p2 = aln_pos(y.aln);
						const int rid2 = rid_of(n_contigs, contig_offset, p2);
						out += contig_names[rid2]; out += ','; put_int(out, p2 - (n_contigs > 1 ? contig_offset[rid2] : 0) + 1); out += ',';
						out += "+-"[y.aln[2] ? 1 : 0]; out += ',';
						put_cigar(out, y, false);
						out += ','; put_int(out, y.fin[13]); out += ','; put_int(out, y.aln[4]); out += ';';
					}
				}
			}
			if (!xa[i].empty()) { out += "\tXA:Z:"; out += xa[i]; }
			out += '\n';
		}
		base += n;
	}
	char *res = (char *)malloc(out.size() + 1);
	if (!res) { bmh_set_error("bmh_format_sam: out of memory"); return nullptr; }
	memcpy(res, out.data(), out.size()); res[out.size()] = 0;
	*len_out = out.size();
	return res;
}
