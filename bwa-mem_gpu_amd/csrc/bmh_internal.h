// Internal declarations shared by the translation units of libbwamem_hip.so.
#pragma once
#include "../../include/bwamem_hip.h"
#include "fmd_dev.h"

struct bmh_index {
	fmd_dev_t dev;
	bool owns;             // arrays were hipMalloc'd by bmh_index_upload
	bool owns_sa;          // sa / sa_bits were replaced by bmh_index_densify_sa (its own allocations)
	bool owns_blocks;      // dev.blocks is the handle's own native re-encoding of a caller's buffer (bmh_index_from_device)
	uint64_t n_words;
};

// extension jobs described by where their bases live instead of materialised base arrays (device job builder)
struct bmh_ext_desc_t {
	const uint8_t *reads;        // ASCII reads of the batch
	const uint8_t *pac; long long l_pac;     // 2-bit forward strand
	const uint32_t *jq_src;      // [n] offset of the query segment in reads
	const uint32_t *job_side;    // [n] 0 = LEFT (both sequences run backwards), 1 = RIGHT
	uint32_t max_qlen;           // 0, or an upper bound of every query length of the batch (the longest read): classes no job can reach are not launched
	const int64_t *jt0;          // [n] first text position of the target window; a window lies on ONE strand of fwd . revcomp(fwd)
	                             // (mem_chain2aln clips it at l_pac, src/bwamem.c:1261-1264): the kernels decode eight rows at a time on that premise
};
int bmh_extend_batch_desc(const bmh_ext_desc_t *desc, const uint32_t *d_qlen, const uint32_t *d_tlen, const uint32_t *d_h0, uint32_t n,
                          const bmh_ext_params_t *p, int32_t *d_out, int32_t *d_raw, void *stream);
// sizes the extension's per-(device, stream) scratch for batches of up to n jobs (a later growth frees device memory, which waits for the device)
int bmh_extend_reserve(void *stream, uint64_t n);

#ifdef __cplusplus
#include <string>
#include <vector>
// where the formatter finds a record's alignment: slot (64- or 32-bit, -1 = none) -> aln[8], and the operations / MD string either in the
// fixed slots bmh_cigar_batch writes or in the packed words of bmh_cigar_pack (off[s] = first word of alignment s: its operations, then
// its MD string)
struct bmh_cigar_src_t {
	const int64_t *slot64 = nullptr; const int32_t *slot32 = nullptr;
	const int32_t *aln = nullptr;
	const uint32_t *cigar = nullptr; int max_cigar = 0; const char *md = nullptr; int md_cap = 0;
	const uint32_t *packed = nullptr; const uint32_t *off = nullptr; bool packed_md = true;
};
// bmh_format_sam / bmh_format_sam_pe (h_rec, unflag: the pairs' arrays, NULL for single-end reads) as the parts the formatting threads
// made, in order; `parts` is reused by the caller (csrc/sam_format.cpp, csrc/align_pipeline.hip)
bool bmh_format_sam_parts(const bmh_post_opt_t *po, uint32_t n_reads, const char *names, const uint64_t *name_off, const uint8_t *reads,
                          const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                          const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const bmh_cigar_src_t &cs,
                          const int32_t *h_rec, const int32_t *unflag, std::vector<std::string> &parts);
// csrc/sam_kernels.hip, for csrc/align_pipeline.hip: the alignments whose fixed slots overflowed (flags 1, 8) found and, once redone with large
// slots, put in place on the device (see the definitions)
int64_t bmh_cigar_overflowed(const int32_t *d_aln, uint32_t n, const uint32_t *d_sel, uint32_t *d_over, uint32_t *d_sel2, uint32_t *d_counter, void *stream);
size_t bmh_cigar_patch_work(uint32_t n_over);
int64_t bmh_cigar_patch(int32_t *d_aln, uint32_t *d_off, uint32_t *d_packed, uint64_t words, const uint32_t *d_over, uint32_t n_over,
                        const int32_t *d_aln2, const uint32_t *d_cigar2, int mc2, const char *d_md2, int mdc2, void *d_work, size_t work_bytes, void *stream);
// csrc/regs_kernels.hip: bmh_finalize_regs_device with by-products (in: d_dedup_out [n_regs][16] or NULL, d_out_off [n_reads] or NULL; out: the device's
// logarithm table and contig offsets, valid until the stream's scratch is released; alt_keep_sub_n: with an ALT table [11] stays sub_n -- regs_core.h ctx_t)
struct bmh_fin_extra_t { int32_t *d_dedup_out; uint32_t *d_out_off; const double *d_logtab; int n_log; const int64_t *d_ctg_off; int alt_keep_sub_n; };
int64_t bmh_finalize_regs_device_ex(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                                    const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                                    const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read, const float *d_frac_rep,
                                    int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream, bmh_fin_extra_t *extra);
// csrc/sam_kernels.hip: the device tail's records as ALT-mode records ([11] = [12]), then the records of the ns reads the host redid with the ALT table (d_ids, their
// records d_sub at d_sub_off [ns + 1]) in their places (d_rec_off: first record of every read)
int bmh_alt_records_device(int32_t *d_fin, uint64_t m, const uint32_t *d_rec_off, const uint32_t *d_ids, const uint32_t *d_sub_off, const int32_t *d_sub, uint32_t ns, void *stream);
// csrc/reads_io.cpp: a mapped read file cut and filled batch by batch (bmh_aligner_run_fasta)
int bmh_fasta_cut(const uint8_t *buf, size_t sz, size_t p, uint64_t want_bases, uint64_t want_reads, bool even, int n_threads, size_t est_bytes,
                  size_t *end, uint64_t *n_reads, uint64_t *n_bases, uint64_t *n_name_bytes);
int bmh_fasta_fill(const uint8_t *buf, size_t p, size_t end, uint64_t n_reads, uint64_t n_bases, uint64_t n_name_bytes, int n_threads, bmh_read_set_t *o);
// ---- interleaved pairs with mem_pair / mem_sam_pe's choices on the device (csrc/pair_dev.hip) for the pairs the mate rescue does not touch
// The host call (csrc/pair_post.cpp: bmh_finalize_pairs_split = bmh_finalize_pairs_deduped on a subset) tells the caller the insert-size statistics as
// soon as it has them (after_pestat: the caller starts the device's pair kernel), asks before its own final walk which pairs the device handed back
// (before_final: extra[n_pairs], non-zero = the host's), and walks those and the pairs the rescue touches; todo_pairs (room for n_reads / 2) receives them.
struct bmh_pairs_split_t {
	int (*after_pestat)(void *user, const double *pes /* [4][5] low, high, failed, avg, std */);
	int (*before_final)(void *user, const uint8_t **extra);
	void *user;
	uint32_t *todo_pairs; uint64_t n_todo;
	void **scratch_slot;            // optional: where the call keeps its large host arrays between calls (NULL at first; bmh_pairs_scratch_free); else the thread's
	const struct bmh_rescue_in_t *rescue_in;     // optional (csrc/pair_kernels.h): the same regions on the device -- the rescue's windows are then found there, not by a host walk
};
void bmh_pairs_scratch_free(void *p);
int64_t bmh_finalize_pairs_split(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, void *stream,
                                 const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                                 int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                 const uint32_t *read_lens, const int32_t *dedup_recs, const uint32_t *dedup_per_read, const float *frac_rep,
                                 int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                                 int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, int n_threads, bmh_pairs_split_t *split);
int bmh_pair_device(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe, const double *pes, int64_t l_pac,
                    int n_contigs, const int64_t *d_ctg_off, const double *d_logtab, int n_log, int32_t *d_fin, const uint32_t *d_opr, const uint32_t *d_off,
                    const float *d_frac_rep, uint32_t n_reads, int32_t *d_h_rec, int32_t *d_unflag, uint8_t *d_todo, void *stream);
int bmh_pair_limit(void);
int bmh_pair_merge_counts(uint32_t n_reads, const uint32_t *d_todo_pairs, uint32_t n_todo, int32_t *d_slot, const uint32_t *d_opr_dev, const int32_t *d_h_dev, const int32_t *d_uf_dev,
                          const uint32_t *d_opr_host, const int32_t *d_h_host, const int32_t *d_uf_host, uint32_t *d_opr, int32_t *d_h, int32_t *d_uf, void *stream);
int bmh_pair_merge_records(uint32_t n_reads, const int32_t *d_slot, const int32_t *d_fin_dev, const uint32_t *d_off_dev, const int32_t *d_fin_host, const uint32_t *d_off_host,
                           const uint32_t *d_opr, const uint32_t *d_off, int32_t *d_fin, void *stream);
size_t bmh_pair_scan_bytes(uint32_t n);
int bmh_pair_scan(const uint32_t *d_in, uint32_t *d_out, uint32_t n, void *d_tmp, size_t tmp_bytes, void *stream);
// bmh_finalize_regs on a subset of a batch's reads: read_ids[r] = the read's index in its batch (hash seed, record field [0]); NULL = r
int64_t bmh_finalize_regs_ids(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, int64_t l_pac,
                              const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                              const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                              int n_contigs, const int64_t *contig_offset,
                              int32_t *out, uint32_t *out_per_read, int n_threads, const uint32_t *read_ids);
extern "C" {
#endif
void bmh_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
// Measurement knobs (csrc/c_api.hip): the value of knob `name` -- what bmh_tune_set gave it in this process, else the environment
// variable BMH_<NAME>, else dflt.  Looked up at every use (a map lookup), so that one process can sweep a knob (scripts/corun_probe.py).
int bmh_tune(const char *name, int dflt);
// wave residency trace (csrc/wtrace.h): one setter per translation unit with instrumented kernels
int bmh_wtrace_set_seed(void *buf, unsigned int *cnt, unsigned int cap);
int bmh_wtrace_set_chain(void *buf, unsigned int *cnt, unsigned int cap);
int bmh_wtrace_set_extend(void *buf, unsigned int *cnt, unsigned int cap);
#ifdef __cplusplus
}
#endif
