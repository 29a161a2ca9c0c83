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
#include <thread>
#include <vector>
#include "bmh_internal.h"

namespace {

// decimal digits by hand: a record holds nine or more numbers, and snprintf was a third of the formatting time
inline void put_int(std::string &s, long long v)
{
	char b[24]; int n = 0;
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) b[n++] = '-';
	const size_t at = s.size();
	s.resize(at + (size_t)n);
	char *d = &s[at];
	for (int i = 0; i < n; ++i) d[i] = b[n - 1 - i];
}
// SEQ: bases [qb, qe) of the read as letters, reverse-complemented for the reverse strand (one resize, then a table walk)
inline void put_seq(std::string &s, const uint8_t *seq, int qb, int qe, bool rev)
{
	if (qe <= qb) return;
	const size_t at = s.size();
	s.resize(at + (size_t)(qe - qb));
	char *d = &s[at];
	if (!rev) for (int k = qb; k < qe; ++k) *d++ = "ACGTN"[seq[k] > 4 ? 4 : seq[k]];
	else for (int k = qe - 1; k >= qb; --k) *d++ = "TGCAN"[seq[k] > 4 ? 4 : seq[k]];
}

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

struct Mate { bool present; int rid; long long pos; int is_rev, n_cigar; const uint32_t *cigar; };

int ref_len(int n, const uint32_t *cg)               // get_rlen, src/bwamem.c:1496-1504
{
	int l = 0;
	for (int k = 0; k < n; ++k) { const int op = (int)(cg[k] & 0xf); if (op == 0 || op == 2) l += (int)(cg[k] >> 4); }
	return l;
}

} // namespace

extern "C" void bmh_free(void *p) { free(p); }

// (get_pri_idx takes XA_drop_ratio as a double: the float 0.8 widened, so a hit at exactly 80 % of its primary is out)
// need[i] = 1 for every record of bmh_finalize_regs that must go through bmh_cigar_batch before formatting: the reported
// ones and the XA candidates (mem_gen_alt's two passes).  Returns their number.
extern "C" int64_t bmh_sam_need_cigar(const bmh_post_opt_t *po, const int32_t *fin, const uint32_t *fin_per_read, uint32_t n_reads, uint8_t *need)
{
	if (!po || !fin_per_read || !need || (n_reads && !fin)) { bmh_set_error("bmh_sam_need_cigar: null argument"); return BMH_EINVAL; }
	int64_t total = 0; uint64_t base = 0;
	std::vector<int> cnt, has_alt;
	for (uint32_t r = 0; r < n_reads; ++r) {
		const int n = (int)fin_per_read[r];
		const int32_t *a = fin + 16 * base;
		cnt.assign(n, 0);
		for (int i = 0; i < n; ++i) need[base + i] = (a[16 * i + 15] & 1) ? 1 : 0;
		if (!po->flag_all) {
			// (with ALT contigs a record keeps secondary_all, the XA tag's key, in [11]: see bmh_post_opt_t)
			const int sa = po->contig_is_alt ? 11 : 12;
			auto pri = [&](int i) { const int k = a[16 * i + sa]; return (k >= 0 && a[16 * i + 1] >= a[16 * k + 1] * (double)po->XA_drop_ratio) ? k : -1; };
			has_alt.assign(n, 0);
			for (int i = 0; i < n; ++i) { const int k = pri(i); if (k >= 0) { ++cnt[k]; if (a[16 * i + 15] & 2) has_alt[k] = 1; } }
			for (int i = 0; i < n; ++i) { const int k = pri(i); if (k >= 0 && !(cnt[k] > po->max_XA_hits_alt || (!has_alt[k] && cnt[k] > po->max_XA_hits))) need[base + i] = 1; }
		}
		for (int i = 0; i < n; ++i) total += need[base + i];
		base += n;
	}
	return total;
}

// slot[i] = index of record i in the bmh_cigar_batch outputs (aln [..][8], cigar [..][max_cigar], md [..][md_cap]) or -1.
// names: the read names, NUL-terminated, back to back; name_off[r] = start of read r's name.  contig_names: array of C
// strings.  reads: nt4 codes.  h_rec / unflag: NULL for single-end reads; for interleaved
// pairs the outputs of bmh_finalize_pairs (own-alignment record per read, flags of the unmapped record).
// the text as the parts its formatting threads made, in order (`parts` is the caller's: a caller that formats batch after batch keeps
// the strings, whose capacity survives clear(), so that no quarter-gigabyte buffer is allocated, faulted in and unmapped per batch)
bool bmh_format_sam_parts(const bmh_post_opt_t *po, uint32_t n_reads, const char *names, const uint64_t *name_off, const uint8_t *reads,
                          const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                          const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const bmh_cigar_src_t &cs,
                          const int32_t *h_rec, const int32_t *unflag, std::vector<std::string> &parts)
{
	std::vector<uint64_t> bases((size_t)n_reads + 1, 0);
	for (uint32_t r = 0; r < n_reads; ++r) bases[r + 1] = bases[r] + fin_per_read[r];
	const bool pe = h_rec != nullptr;
	const char *rg = po->rg_id && po->rg_id[0] ? po->rg_id : nullptr;      // read group: RG:Z:<id> on every record
	auto rec_at = [&](uint64_t base, const int32_t *a, int i) {
		Rec x; x.fin = a + 16 * i;
		const int64_t s = cs.slot32 ? (int64_t)cs.slot32[base + i] : cs.slot64[base + i];
		x.aln = nullptr; x.cigar = nullptr; x.md = "";
		if (s >= 0) {
			x.aln = cs.aln + 8 * s;
			if (cs.packed) { x.cigar = cs.packed + cs.off[s]; if (cs.packed_md) x.md = (const char *)(x.cigar + x.aln[3]); }
			else { x.cigar = cs.cigar + (size_t)cs.max_cigar * s; if (cs.md) x.md = cs.md + (size_t)cs.md_cap * s; }
		}
		return x;
	};
	// reads are independent: format ranges of them on host threads (the reference formats inside its worker threads)
	// (threads: the hardware's, not bmh_effective_cpus(): a CPU quota limits the RATE of CPU time, and a batch's text is a burst -- on a box that
	// shows 256 threads and grants 16 CPUs' worth of time, 64 threads format a million reads in 14 ms, 16 threads in 45 ms)
	unsigned n_thr = n_reads >= 8192 ? std::thread::hardware_concurrency() : 1;
	if (n_thr < 1) n_thr = 1;
	if (n_thr > 64) n_thr = 64;
	if (parts.size() < n_thr) parts.resize(n_thr);
	for (std::string &p : parts) p.clear();
	std::vector<int> failed(n_thr, 0);
	auto work = [&](unsigned t) {
	std::string &out = parts[t];
	const uint32_t r_lo = (uint32_t)((uint64_t)n_reads * t / n_thr), r_hi = (uint32_t)((uint64_t)n_reads * (t + 1) / n_thr);
	out.reserve((size_t)(r_hi - r_lo) * 400);
	std::vector<int> cnt, list, has_alt;
	std::vector<std::string> xa;
	for (uint32_t r = r_lo; r < r_hi; ++r) {
		const uint64_t base = bases[r];
		const int n = (int)fin_per_read[r];
		const int32_t *a = fin + 16 * base;
		auto rec = [&](int i) { return rec_at(base, a, i); };
		// the mate's own alignment (mem_sam_pe's h[!i])
		Mate m; m.present = pe; m.rid = -1; m.pos = 0; m.is_rev = 0; m.n_cigar = 0; m.cigar = nullptr;
		if (pe) {
			const uint32_t mr = r ^ 1u;
			if (h_rec[mr] >= 0) {
				const Rec y = rec_at(bases[mr], fin + 16 * bases[mr], h_rec[mr]);
				if (!y.aln) { bmh_set_error("bmh_format_sam_pe: the alignment record of read %u has no CIGAR", mr); failed[t] = 1; return; }
				m.pos = aln_pos(y.aln); m.rid = rid_of(n_contigs, contig_offset, m.pos); m.is_rev = y.aln[2]; m.n_cigar = y.aln[3]; m.cigar = y.cigar;
			}
		}
		// XA strings per primary (mem_gen_alt)
		if ((int)xa.size() < n) xa.resize(n);
		for (int i = 0; i < n; ++i) xa[i].clear();                  // (capacity kept: no allocation per read)
		if (!po->flag_all) {
			cnt.assign(n, 0); has_alt.assign(n, 0);
			const int sa = po->contig_is_alt ? 11 : 12;
			auto pri = [&](int i) { const int k = a[16 * i + sa]; return (k >= 0 && a[16 * i + 1] >= a[16 * k + 1] * (double)po->XA_drop_ratio) ? k : -1; };
			for (int i = 0; i < n; ++i) { const int k = pri(i); if (k >= 0) { ++cnt[k]; if (a[16 * i + 15] & 2) has_alt[k] = 1; } }
			for (int i = 0; i < n; ++i) {
				const int k = pri(i);
				if (k < 0 || cnt[k] > po->max_XA_hits_alt || (!has_alt[k] && cnt[k] > po->max_XA_hits)) continue;       // src/bwamem_extra.c:125
				const Rec x = rec(i);
				if (!x.aln) { bmh_set_error("bmh_format_sam: record %d of read %u has no CIGAR (see bmh_sam_need_cigar)", i, r); failed[t] = 1; return; }
				const long long pos = aln_pos(x.aln);
				const int rid = rid_of(n_contigs, contig_offset, pos);
				std::string &s = xa[k];
				s += contig_names[rid]; s += ','; s += "+-"[x.aln[2] ? 1 : 0]; put_int(s, pos - (n_contigs > 1 ? contig_offset[rid] : 0) + 1); s += ',';
				put_cigar(s, x, false);
				s += ','; put_int(s, x.aln[4]); s += ';';
			}
		}
		list.clear();
		for (int i = 0; i < n; ++i) if (a[16 * i + 15] & 1) list.push_back(i);
		const uint8_t *seq = reads + read_offs[r];
		const int l_seq = (int)read_lens[r];
		auto mate_fields = [&](int p_rid, long long p_pos, int p_rev, int p_ncig, const uint32_t *p_cig, bool mate_mapped, int m_rid, long long m_pos, int m_rev,
		                       int m_ncig, const uint32_t *m_cig) {
			if (pe && mate_mapped) {
				if (p_rid == m_rid) out += '='; else out += contig_names[m_rid];
				out += '\t'; put_int(out, m_pos - (n_contigs > 1 ? contig_offset[m_rid] : 0) + 1); out += '\t';
				if (p_rid == m_rid) {
					const long long p0 = p_pos + (p_rev ? ref_len(p_ncig, p_cig) - 1 : 0), p1 = m_pos + (m_rev ? ref_len(m_ncig, m_cig) - 1 : 0);
					if (m_ncig == 0 || p_ncig == 0) out += '0';
					else put_int(out, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
				} else out += '0';
			} else out += "*\t0\t0";
			out += '\t';
		};
		if (list.empty()) {                                   // unmapped record (mem_reg2sam's aa.n == 0 branch)
			int flag = 4 | (unflag ? unflag[r] : 0);
			const bool mm = pe && m.rid >= 0;
			if (pe && m.rid < 0) flag |= 8;
			const int p_rev = mm ? m.is_rev : 0;                 // an unmapped read takes its mate's coordinate and strand
			if (p_rev) flag |= 0x10;
			if (mm && m.is_rev) flag |= 0x20;
			out += names + name_off[r]; out += '\t'; put_int(out, flag); out += '\t';
			if (mm) { out += contig_names[m.rid]; out += '\t'; put_int(out, m.pos - (n_contigs > 1 ? contig_offset[m.rid] : 0) + 1); out += "\t0\t*\t"; }
			else out += "*\t0\t0\t*\t";
			mate_fields(mm ? m.rid : -1, m.pos, p_rev, 0, nullptr, mm, m.rid, m.pos, m.is_rev, m.n_cigar, m.cigar);
			put_seq(out, seq, 0, l_seq, p_rev != 0);
			out += "\t*\tAS:i:0\tXS:i:0";
			if (rg) { out += "\tRG:Z:"; out += rg; }
			out += '\n';
			continue;
		}
		for (size_t which = 0; which < list.size(); ++which) {
			const int i = list[which];
			const Rec x = rec(i);
			if (!x.aln) { bmh_set_error("bmh_format_sam: record %d of read %u has no CIGAR (see bmh_sam_need_cigar)", i, r); failed[t] = 1; return; }
			const long long pos = aln_pos(x.aln);
			const int rid = rid_of(n_contigs, contig_offset, pos);
			// a mapped read whose mate is unmapped lends it its coordinate and strand (mem_aln2sam :1518-1521)
			const bool mate_mapped = pe && m.rid >= 0;
			const int m_rid = mate_mapped ? m.rid : rid; const long long m_pos = mate_mapped ? m.pos : pos; const int m_rev = mate_mapped ? m.is_rev : (x.aln[2] ? 1 : 0);
			int flag = (x.aln[2] ? 0x10 : 0) | x.fin[14];
			if (pe) { if (m.rid < 0) flag |= 8; if (m_rev) flag |= 0x20; }
			const bool hard = which > 0 && !po->softclip && !(x.fin[15] & 2);      // src/bwamem.c:1540,1578 (never on an ALT hit)
			out += names + name_off[r]; out += '\t'; put_int(out, (flag & 0xffff) | (flag & 0x10000 ? 0x100 : 0)); out += '\t';
			out += contig_names[rid]; out += '\t'; put_int(out, pos - (n_contigs > 1 ? contig_offset[rid] : 0) + 1); out += '\t';
			put_int(out, x.fin[13]); out += '\t';
			if (x.aln[3]) put_cigar(out, x, hard); else out += '*';
			out += '\t';
			mate_fields(rid, pos, x.aln[2] ? 1 : 0, x.aln[3], x.cigar, pe, m_rid, m_pos, m_rev, mate_mapped ? m.n_cigar : 0, mate_mapped ? m.cigar : nullptr);
			if (flag & 0x100) out += "*\t*";
			else {
				int qb = 0, qe = l_seq;
				const int nc = x.aln[3];
				if (nc && hard) {                                 // hard-clipped records print only the aligned part
					const int c0 = (int)(x.cigar[0] & 0xf), c1 = (int)(x.cigar[nc - 1] & 0xf);
					if (!x.aln[2]) { if (c0 == 3 || c0 == 4) qb += x.cigar[0] >> 4; if (c1 == 3 || c1 == 4) qe -= x.cigar[nc - 1] >> 4; }
					else { if (c0 == 3 || c0 == 4) qe -= x.cigar[0] >> 4; if (c1 == 3 || c1 == 4) qb += x.cigar[nc - 1] >> 4; }
				}
				put_seq(out, seq, qb, qe, x.aln[2] != 0);
				out += "\t*";
			}
			if (x.aln[3]) { out += "\tNM:i:"; put_int(out, x.aln[4]); out += "\tMD:Z:"; out += x.md; }
			if (x.fin[1] >= 0) { out += "\tAS:i:"; put_int(out, x.fin[1]); }
			if (!(flag & 0x100) && x.fin[10] >= 0) { out += "\tXS:i:"; put_int(out, x.fin[10]); }      // sub is not printed for secondary records (q->sub = -1)
			if (rg) { out += "\tRG:Z:"; out += rg; }               // src/bwamem.c:1631-1634
			if (!(flag & 0x100)) {
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
			if (!(flag & 0x100) && (x.fin[15] >> 2) > 0) {       // pa:f:<score / score of the ALT hit that shadows it> (src/bwamem.c:1663)
				char buf[48]; snprintf(buf, sizeof(buf), "\tpa:f:%.3f", (double)x.fin[1] / (double)(x.fin[15] >> 2)); out += buf;
			}
			if (!xa[i].empty()) { out += "\tXA:Z:"; out += xa[i]; }
			out += '\n';
		}
	}
	};
	if (n_thr == 1) work(0);
	else { std::vector<std::thread> th; for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(work, t); for (auto &x : th) x.join(); }
	for (unsigned t = 0; t < n_thr; ++t)
		if (failed[t]) { if (n_thr > 1) bmh_set_error("bmh_format_sam: a record the text needs has no CIGAR (see bmh_sam_need_cigar)"); return false; }
	return true;
}

static char *format_sam(const bmh_post_opt_t *po, uint32_t n_reads, const char *names, const uint64_t *name_off, const uint8_t *reads,
                        const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                        const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const int64_t *slot,
                        const int32_t *aln, const uint32_t *cigar, int max_cigar, const char *md, int md_cap,
                        const int32_t *h_rec, const int32_t *unflag, size_t *len_out)
{
	std::vector<std::string> parts;
	bmh_cigar_src_t cs;
	cs.slot64 = slot; cs.aln = aln; cs.cigar = cigar; cs.max_cigar = max_cigar; cs.md = md; cs.md_cap = md_cap;
	if (!bmh_format_sam_parts(po, n_reads, names, name_off, reads, read_offs, read_lens, n_contigs, contig_names, contig_offset, fin, fin_per_read, cs, h_rec, unflag, parts)) return nullptr;
	const unsigned n_thr = (unsigned)parts.size();
	size_t total = 0;
	for (const std::string &p : parts) total += p.size();
	char *res = (char *)malloc(total + 1);
	if (!res) { bmh_set_error("bmh_format_sam: out of memory"); return nullptr; }
	{   // the parts into their places, on as many threads (the text of a million reads is a quarter of a gigabyte)
		std::vector<size_t> at(n_thr ? n_thr : 1, 0);
		for (unsigned t = 1; t < n_thr; ++t) at[t] = at[t - 1] + parts[t - 1].size();
		auto put = [&](unsigned t) { memcpy(res + at[t], parts[t].data(), parts[t].size()); std::string().swap(parts[t]); };
		if (n_thr <= 1) { if (n_thr) put(0); }
		else { std::vector<std::thread> th; for (unsigned t = 0; t < n_thr; ++t) th.emplace_back(put, t); for (auto &x : th) x.join(); }
	}
	res[total] = 0;
	*len_out = total;
	return res;
}

extern "C" char *bmh_format_sam(const bmh_post_opt_t *po, uint32_t n_reads, const char *names, const uint64_t *name_off, const uint8_t *reads,
                                const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                                const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const int64_t *slot,
                                const int32_t *aln, const uint32_t *cigar, int max_cigar, const char *md, int md_cap, size_t *len_out)
{
	if (!po || !names || !name_off || !reads || !read_offs || !read_lens || !contig_names || !fin_per_read || !len_out || (n_contigs > 1 && !contig_offset)) {
		bmh_set_error("bmh_format_sam: null argument"); return nullptr;
	}
	return format_sam(po, n_reads, names, name_off, reads, read_offs, read_lens, n_contigs, contig_names, contig_offset, fin, fin_per_read, slot, aln, cigar, max_cigar,
	                  md, md_cap, nullptr, nullptr, len_out);
}

// interleaved pairs: fin / fin_per_read / h_rec / unflag from bmh_finalize_pairs (mem_aln2sam with the mate: flags 0x8 0x20,
// RNEXT, PNEXT, TLEN; an unmapped read takes its mate's coordinate and strand)
extern "C" char *bmh_format_sam_pe(const bmh_post_opt_t *po, uint32_t n_reads, const char *names, const uint64_t *name_off, const uint8_t *reads,
                                   const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                                   const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const int32_t *h_rec,
                                   const int32_t *unflag, const int64_t *slot, const int32_t *aln, const uint32_t *cigar, int max_cigar,
                                   const char *md, int md_cap, size_t *len_out)
{
	if (!po || !names || !name_off || !reads || !read_offs || !read_lens || !contig_names || !fin_per_read || !h_rec || !unflag || !len_out || (n_reads & 1) ||
	    (n_contigs > 1 && !contig_offset)) { bmh_set_error("bmh_format_sam_pe: bad argument"); return nullptr; }
	return format_sam(po, n_reads, names, name_off, reads, read_offs, read_lens, n_contigs, contig_names, contig_offset, fin, fin_per_read, slot, aln, cigar, max_cigar,
	                  md, md_cap, h_rec, unflag, len_out);
}

// bmh_sam_need_cigar for pairs: additionally the own-alignment record of every read (the mate fields come from it)
extern "C" int64_t bmh_sam_need_cigar_pe(const bmh_post_opt_t *po, const int32_t *fin, const uint32_t *fin_per_read, const int32_t *h_rec,
                                         uint32_t n_reads, uint8_t *need)
{
	if (!h_rec) { bmh_set_error("bmh_sam_need_cigar_pe: null argument"); return BMH_EINVAL; }
	int64_t total = bmh_sam_need_cigar(po, fin, fin_per_read, n_reads, need);
	if (total < 0) return total;
	uint64_t base = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		if (h_rec[r] >= 0 && !need[base + h_rec[r]]) { need[base + h_rec[r]] = 1; ++total; }
		base += fin_per_read[r];
	}
	return total;
}
