/*
 * include/bwamem_hip.h -- device-level C ABI of the MI355X seed-and-extend library
 * (libbwamem_hip.so).  Plain pointers and sizes only; every pointer named d_* is a
 * device (HBM) pointer, everything else is host memory.
 *
 * This is the layer the reference-compatible entry points are built on:
 *   include/seed_gen.h   (replaces /root/reference/src/GPUSeed/seed_gen.h:92-106)
 *   include/gasal_ext.h  (replaces the GASAL2 calls of /root/reference/src/bwamem.c:1102-1167,
 *                         2106-2211 and src/fastmap.c:417-534)
 * and the layer bench.py / the multi-GPU launcher use so that inputs are already
 * resident in HBM when timing starts.
 *
 * Error behaviour: functions returning int return 0 on success and a negative
 * BMH_E* code on failure; bmh_last_error() gives the message.  Nothing here
 * falls back to a CPU implementation: without a HIP device every call fails.
 */
#ifndef BWAMEM_HIP_H
#define BWAMEM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMH_OK          0
#define BMH_ENODEV     -1   /* no HIP device / HIP runtime error */
#define BMH_EINVAL     -2   /* bad argument */
#define BMH_ECAPACITY  -3   /* a workspace capacity was exceeded (reference: exit(), seed_gen.cu:2037-2042) */
#define BMH_ENOMEM     -4

const char *bmh_last_error(void);
int bmh_device_count(void);
int bmh_set_device(int dev);

/* ------------------------------------------------------------------ index */

/* FMD index resident in HBM.  The arrays that cross this interface have the reference's
 * GPU layout (seed_gen.cu:28-48; seed_gen.h:21-33 bwt_t_gpu): blocks {u32 occ[4]; u32 bwt[4]}.
 * A handle keeps the blocks re-encoded as {u32 occ[4]; u64 low bit plane; u64 high bit plane}
 * (same 32 bytes per 64 symbols; csrc/fmd_dev.h) -- bmh_index_upload converts its device copy
 * in place, bmh_index_from_device makes its own converted copy (seq_len / 2 bytes of HBM) and
 * leaves the caller's buffer as it is. */
typedef struct bmh_index bmh_index_t;

/* host arrays -> HBM (replaces gpu_cpy_wrapper, seed_gen.cu:1524-1556).
 * bwt_words: interleaved occ/bwt blocks, n_words u32; sa: n_sa u32 samples with
 * sa[0] ignored; sa_bits: n_sa/32+1 words; pac (optional, may be NULL): 2-bit
 * forward strand, l_pac bases. */
bmh_index_t *bmh_index_upload(uint64_t primary, const uint64_t L2[5], uint64_t seq_len,
                              const uint32_t *bwt_words, uint64_t n_words, int sa_intv,
                              const uint32_t *sa, uint64_t n_sa, const uint32_t *sa_bits,
                              const uint8_t *pac, uint64_t l_pac);
/* wrap arrays that already live in HBM (e.g. received by an RCCL broadcast);
 * the index does not own them.  d_pac (optional): 4-byte aligned, readable for
 * l_pac/4 + 9 bytes (the kernels fetch the text in aligned words).  With the
 * text resident the seeding kernels stop ranking once an interval holds one
 * suffix and compare the read with the text instead. */
bmh_index_t *bmh_index_from_device(uint64_t primary, const uint64_t L2[5], uint64_t seq_len,
                                   const uint32_t *d_bwt_words, uint64_t n_words, int sa_intv,
                                   const uint32_t *d_sa, uint64_t n_sa, const uint32_t *d_sa_bits,
                                   const uint8_t *d_pac, uint64_t l_pac);
void bmh_index_free(bmh_index_t *idx);
/* Rank primitives at n given rows (d_rows, d_out: device memory), asynchronous on `stream`:
 *   what = 0: d_out[4 i + c] = Occ(rows[i], c), c = A,C,G,T  (bwt_occ4, src/bwt.c:309-330; rows -1 and seq_len allowed)
 *   what = 1: d_out[i] = LF(rows[i])                         (bwt_invPsi, src/bwt.c:64-70)
 *   what = 2: d_out[i] = SA[rows[i]]                         (bwt_sa, src/bwt.c:105-115)
 * rows of 1 / 2 must lie in [0, seq_len]. */
int bmh_index_probe(const bmh_index_t *idx, const uint64_t *d_rows, uint64_t n, int what, uint64_t *d_out, void *stream);

/* ---- several GPUs of one node from C: the index on every device, the reads sharded, one host worker thread per device
 * (the reference has no multi-GPU mode at all: gasal_set_device is commented out at src/fastmap.c:143).
 * bmh_index_replicate copies an index that lives on src_device into fresh allocations on dst_device (device to device: xGMI
 * between the GPUs of a node; src_device == dst_device makes a second copy on the same GPU).  The copy owns its arrays (free it
 * with bmh_index_free while dst_device is current).  bmh_shard_range: the contiguous part [lo, hi) of n reads that worker `rank`
 * of `world` takes, cut at multiples of `multiple` (2 keeps interleaved pairs together).
 * The drop-in entry points use them when BMH_DEVICES=N is set: seed_gpu() sends the batches of the read file round-robin to N
 * worker threads, one per device, and concatenates their seeds in file order; the gasal_gpu_storage_t objects of
 * gasal_init_streams are spread over the N devices.  With fewer physical devices than N several workers share one. */
int bmh_index_replicate(const bmh_index_t *src, int src_device, int dst_device, bmh_index_t **out);

/* ---- the same with RCCL over xGMI (csrc/rccl_bcast.hip).  RCCL is resolved at run time -- the symbols already in the process (a
 * host linked with -lrccl), then $BMH_RCCL_LIB, then librccl.so.1 of the loader path -- because a communicator is only valid in
 * the library instance that made it; bmh_rccl_where() names the instance in use (NULL: none found, bmh_last_error says why).
 * One process per GPU: rank 0 calls bmh_rccl_unique_id and hands the 128 bytes to the others by its own means (a file, MPI, a
 * socket), every rank calls bmh_rccl_comm_init_rank on its device and then bmh_index_broadcast_rccl: the root passes its index
 * (out == NULL: send only; out != NULL: it receives a fresh copy like the others), the others receive theirs in *out (owned:
 * bmh_index_free).  One 128-byte header, then the blocks (native layout), the suffix array, its high bits and the 2-bit text in
 * ONE grouped broadcast, in pieces of at most 1 GiB, on `stream`; returns when the arrays have arrived.
 * One process, n devices: bmh_index_replicate_all makes out[k] the index on devices[k] (src itself on src_device; duplicates in
 * the list share a copy) with ncclCommInitAll and the same grouped broadcast; *used_rccl (optional) says whether RCCL carried it
 * or the hipMemcpyPeer fallback did (RCCL not found, or BMH_REPLICATE=peer).  BMH_DEVICES=N of the drop-in uses this. */
const char *bmh_rccl_where(void);
int bmh_rccl_unique_id(void *id128);
int bmh_rccl_comm_init_rank(void **comm, int nranks, const void *id128, int rank);
void bmh_rccl_comm_destroy(void *comm);
int bmh_index_broadcast_rccl(void *comm, int root, const bmh_index_t *src, bmh_index_t **out, void *stream);
int bmh_index_replicate_all(const bmh_index_t *src, int src_device, const int *devices, int n, bmh_index_t **out, int *used_rccl);
void bmh_shard_range(uint64_t n, int rank, int world, uint32_t multiple, uint64_t *lo, uint64_t *hi);

/* Replaces the suffix-array samples by denser ones (every new_intv-th row, a power of two; a no-op if the index is that
 * dense already), computed on the device from the existing ones: same values, fewer LF steps per located seed, more HBM
 * (4.125 bytes per sample).  The reference's files hold every 16th row (src/bwtindex.c:324). */
int bmh_index_densify_sa(bmh_index_t *idx, int new_intv);

/* Builds the FMD index of fwd . revcomp(fwd) ON THE DEVICE from the 2-bit forward strand (d_pac: l_pac bases, 4 per byte,
 * first base in the top bits -- the .pac body; l_pac < 2^32, i.e. texts up to 2^33 symbols: hg38 has 6.2e9).  Replaces the
 * reference's offline host passes (bwa_index/bwtindex.c:287-358 bwtsw construction, :174-197 GPU re-blocking,
 * bwa_index/bwt.c:63-148 sampled suffix array) with the same values: the outputs are exactly what bmh_index_from_device
 * takes and, copied to the host, the bodies of the reference's .bwt / .sa files.
 * Caller-allocated device outputs (seq_len = 2 l_pac):
 *   d_bwt_words  (ceil(seq_len/64) + 1) * 8 words, 32-byte aligned: blocks {u32 occ[4]; u32 bwt[4]}, then one block whose occ
 *                is the totals;  d_sa  n_sa = (seq_len + sa_intv) / sa_intv words (sa[0] = 0xFFFFFFFF);  d_sa_bits  n_sa/32 + 1 words.
 * sa_intv: power of two; 16 = what the reference's files hold, 1 = the whole suffix array (what the seeding kernels like best).
 * flags: BMH_BUILD_VERIFY checks the finished suffix array completely (every adjacent pair of rows compared symbol by symbol,
 * SA a permutation) before anything is written.  Uses about 24 bytes of HBM per symbol of seq_len while it runs (175 GB for
 * hg38) and synchronises the device.  stats (optional) reports what happened. */
#define BMH_BUILD_VERIFY 1
typedef struct {
	int round0_passes, doubling_rounds, verified;
	uint64_t unresolved_after_round0;
	double round0_seconds, sa_seconds, verify_seconds, total_seconds;
} bmh_build_stats_t;
int bmh_index_build(const uint8_t *d_pac, uint64_t l_pac, int sa_intv, uint32_t *d_bwt_words, uint32_t *d_sa, uint32_t *d_sa_bits,
                    uint64_t *primary_out, uint64_t L2_out[5], int flags, bmh_build_stats_t *stats);

/* ---------------------------------------------------------------- seeding */

/* Workspace for batches of up to max_reads reads / max_bases bases.
 * max_cands bounds SMEM candidates per batch (0 = default guess: 16 per read + 0.4 per base; the hard
 * upper bound, which never fails, is one per base -- bmh_seed_batch returns BMH_ECAPACITY beyond the
 * capacity); max_occ sizes the arrays of located occurrences (0 = 64 per read): a batch that needs
 * more gets larger arrays inside bmh_seed_batch. */
typedef struct bmh_seed_ws bmh_seed_ws_t;
bmh_seed_ws_t *bmh_seed_ws_create(uint32_t max_reads, uint64_t max_bases, uint64_t max_cands, uint64_t max_occ);
void bmh_seed_ws_free(bmh_seed_ws_t *ws);

/* Seeds of one batch, in HBM, in the reference's mem_seed_v_gpu layout
 * (seed_gen.h:68-75): per read SMEMs by end ascending, occurrences by SA row
 * ascending; score = #occurrences at the first slot of each SMEM group, 0 in
 * the others.  Pointers stay valid until the next call on the same workspace. */
typedef struct {
	uint64_t n_seeds;            /* located occurrences in the batch */
	uint64_t n_smems;            /* SMEM groups */
	uint64_t n_cands;            /* forward candidates examined */
	const uint64_t *d_rbeg;      /* [n_seeds] position in the fwd+revcomp text */
	const int32_t *d_qbeg;       /* [n_seeds][2] = {begin, end} in the read */
	const uint32_t *d_score;     /* [n_seeds] */
	const uint32_t *d_n_ref_pos; /* [n_reads] occurrences per read */
	const uint32_t *d_prefix;    /* [n_reads] exclusive scan of d_n_ref_pos */
} bmh_seeds_t;

/* d_reads: ASCII bases of all reads back to back (no separators), d_offs/d_lens
 * per read (what the reference copies to the GPU, seed_gen.cu:1841-1843).
 * stream: a hipStream_t (NULL = default stream).  Synchronises the stream: the call waits on the host for what the stream
 * holds, runs its kernels on a stream of the highest priority that the workspace owns (beside another batch's extension kernels
 * the seeding waves -- which wait on HBM most of the time -- then take the wave slots that become free; BMH_SEED_PRIO=normal in the
 * environment when the workspace is created: on `stream` itself), and leaves `stream` waiting for them: work queued on `stream`
 * afterwards sees the seeds. */
int bmh_seed_batch(bmh_seed_ws_t *ws, const bmh_index_t *idx, const uint8_t *d_reads,
                   const uint32_t *d_offs, const uint32_t *d_lens, uint32_t n_reads,
                   int min_seed_len, void *stream, bmh_seeds_t *out);

/* per-kernel time of the last bmh_seed_batch, in ms (HIP events on the launch stream):
 * [0]=pack [1]=forward [2]=scatter+backward [3]=filter+scans [4]=expand [5]=locate [6]=total
 * (with BMH_SEED_FUSED=1: [1]=fused forward+backward [2]=sort of the SMEMs [3]=gather+scans) */
void bmh_seed_last_timing(const bmh_seed_ws_t *ws, float ms[7]);

/* Calibration: n_lanes lanes each gather `iters` random 32-byte index blocks (dependent != 0:
 * each address depends on the previous block, like a rank walk).  *ms = kernel time.  Known
 * byte count = n_lanes * iters * 32; used to calibrate rocprofv3 FETCH_SIZE for this access
 * pattern and to measure the practical random-gather ceiling of the chip. */
int bmh_calib_gather(const bmh_index_t *idx, uint64_t n_lanes, int iters, int dependent, void *stream, float *ms);

/* Calibration of the integer-VALU roofline the extension kernels are held against: waves_per_simd resident waves on every
 * SIMD execute `iters` rounds of 128 instructions of one kind -- mode 0: independent v_max_i32 / v_add_u32 (the issue ceiling),
 * 1: one dependent chain, 2: dependent DPP row_shr max (the scans), 3: independent DPP, 4: packed 16-bit add / max,
 * 5: v_bfe_i32, 6: v_fma_f32, 7: v_pk_fma_f32 (two fp32 lanes per instruction: the form the chip's 157 TFLOP/s vector figure
 * needs), 8: the packed DP kernels' own mix (v_pk_mad / sub / max / min / add_u16, v_perm_b32, v_and_b32).
 * *ms = kernel time, *lane_ops = 64 x 128 x iters per wave, summed over the waves (instructions x 64: a packed instruction
 * counts once).  bmh_calib_valu_placed also returns, for every wave of the timed launch, (XCC_ID << 32 | HW_ID) in place[]
 * (host memory for 4 x 256-thread-blocks words; *n_place = how many): which SIMD of which CU of which XCD the wave ran on. */
int bmh_calib_valu(int mode, int waves_per_simd, int iters, void *stream, float *ms, double *lane_ops);
int bmh_calib_valu_placed(int mode, int waves_per_simd, int iters, void *stream, float *ms, double *lane_ops,
                          unsigned long long *place, unsigned *n_place);
/* The shader clock (MHz) and the shader cycles per wave64 instruction of one wave during the calling thread's last bmh_calib_valu[_placed]
 * launch, measured inside the kernel (s_memtime over s_memrealtime); BMH_EINVAL before any calibration. */
int bmh_calib_last_clock(double *mhz, double *cycles_per_instr);

/* -------------------------------------------------------------- extension */

/* Scoring of ksw_extend2 (src/ksw.c:864) as used by the GPU pipeline
 * (src/bwamem.c:1887-1890): mat[i][j] = a / -b / -1 against code 4
 * (src/bwa.c:99-108), affine gaps, end_bonus = pen_clip5, zdrop (0 = off). */
typedef struct {
	int a, b;
	int o_del, e_del, o_ins, e_ins;
	int zdrop, end_bonus;
} bmh_ext_params_t;

/* n extensions: query/target bases are codes 0..4, one byte per base, at
 * d_q + d_qoff[i] (d_qlen[i] bases) and d_t + d_toff[i] (d_tlen[i]);
 * d_h0[i] = seed score.  Results (src/bwamem.c:1893-1901 rule applied):
 * d_out[3*i+0..2] = {aln_score, query_end, target_end}; if d_raw != NULL also
 * d_raw[6*i..] = {score, qle, tle, gtle, gscore, max_off} of ksw_extend2.
 * Asynchronous on stream. */
int bmh_extend_batch(const uint8_t *d_q, const uint32_t *d_qoff, const uint32_t *d_qlen,
                     const uint8_t *d_t, const uint32_t *d_toff, const uint32_t *d_tlen,
                     const uint32_t *d_h0, uint32_t n, const bmh_ext_params_t *p,
                     int32_t *d_out, int32_t *d_raw, void *stream);

/* Queries longer than 768 bases are not supported by the DP kernels (the reference's GASAL2 build has a compile-time
 * MAX_SEQ_LEN too, README.md:38): such a job gets INT32_MIN in all three outputs and is counted; this returns the count for the
 * calling thread's last bmh_extend_batch (waits for it; -1 on a HIP error).  The reference-compatible layer (gasal_aln_async)
 * refuses such a batch up front; bmh_chain_batch only admits reads up to 700 bases, whose flanks always fit. */
int64_t bmh_extend_last_unsupported(void);

/* Jobs whose scores fit 16 bits (h0 + qlen*a < 4096; scoring with 1 <= b, a + b <= 255) and whose sides fit a packed class -- qlen <= 128
 * with tlen <= 384 (4 lanes per job, 16 jobs per wave; qlen 129..136 too while h0 + qlen*a < 2048: the 17-pair class), qlen <= 256
 * with tlen <= 512 (8 lanes, 8 jobs per wave), qlen <= 288 with tlen <= 640 (16 lanes, 4 jobs per wave) -- run on the packed 16-bit
 * kernels (two DP columns per register), everything else on the 32-bit kernels; results are identical.  on = 0 sends every job to the 32-bit kernels (tests, A/B timing).  Process-wide; returns the previous setting. */
int bmh_extend_set_packed(int on);

/* Measurement knobs of the library (each documented where it is read): knob NAME takes the value given here, else
 * the environment variable BMH_<NAME>, else its default; clear != 0 forgets a value set earlier.  Process-wide; for A/B sweeps inside
 * one process (scripts/corun_probe.py).  No reference counterpart. */
int bmh_tune_set(const char *name, int value, int clear);
/* Wave residency trace (measurement; csrc/wtrace.h): between start and stop every wave of the seeding, chaining and packed extension
 * kernels leaves a 32-byte record {u32 kernel id, HW_ID, XCC_ID, aux; u64 t0, t1 in 100 MHz ticks}; stop waits for the device, copies up to
 * max_recs records and returns how many waves reported.  scripts/wave_residency.py turns them into waves of each kernel per SIMD over time. */
int bmh_wtrace_start(uint32_t cap);
int64_t bmh_wtrace_stop(void *out, uint32_t max_recs);
uint32_t bmh_wtrace_kept(void);          /* records the last bmh_wtrace_stop copied out */

/* bmh_extend_batch keeps scratch (the sorted job list, four side streams, events) per (device, stream) and reuses it across calls.
 * Call this before destroying a stream that ran extensions (stream idle, its device current); without it the entry stays until the
 * process ends and a recycled stream handle would inherit it.  One stream must not run extensions from two host threads at once.
 * bmh_finalize_regs_device and bmh_finalize_pairs_dev / bmh_matesw_batch_device keep scratch by the same rule (device buffers, side
 * streams, events, pinned words per (device, stream); ONE host thread per stream at a time): bmh_finalize_release and
 * bmh_matesw_release free theirs, and bmh_extend_release calls both. */
void bmh_extend_release(void *stream);
void bmh_finalize_release(void *stream);
void bmh_matesw_release(void *stream);

/* device time in ms of the DP kernels launched by the calling thread's last
 * bmh_extend_batch (HIP events on that call's stream; waits for them). */
float bmh_extend_last_ms(void);

/* ------------------------------------------------- host job builder */

/* The options of mem_opt_t this stage reads; bmh_chain_opt_default() = mem_opt_init()
 * (src/bwamem.c:101-146). */
typedef struct {
	int a, b, o_del, e_del, o_ins, e_ins, w;
	int min_seed_len, max_occ, max_chain_gap, min_chain_weight, max_chain_extend;
	float mask_level, drop_ratio;
	/* ALT contigs (the reference reads <prefix>.alt, src/bntseq.c:179-200; bns->anns[rid].is_alt): contig_is_alt[n_contigs] in HOST
	 * memory, NULL = none.  Read by bmh_build_jobs (mem_chain_flt ignores an overlap whose kept chain is ALT while the current one is
	 * not, src/bwamem.c:518); the device job builder takes the flags through bmh_chain_set_alt instead. */
	const uint8_t *contig_is_alt;
} bmh_chain_opt_t;
void bmh_chain_opt_default(bmh_chain_opt_t *o);

/* seeds (host copies of the mem_seed_v_gpu arrays) -> chains -> filtered chains -> extension jobs,
 * restating mem_chain / mem_chain_flt / mem_chain2aln (src/bwamem.c:404-477, 487-559, 1170-1479).
 * reads: nt4 codes; pac: 2-bit forward strand (the .pac file body); contigs: n_contigs offsets/lens
 * (n_contigs <= 1: one sequence of l_pac bases).  Jobs come out per read, per region, LEFT then RIGHT.
 * Includes the seed filter mem_flt_chained_seeds + mem_seed_sw (src/bwamem.c:970-991, 774-807), which the reference runs when
 * (min_chain_weight ? 1.1f * min_chain_weight : 5.5 ln l_query) <= 0.05f * l_query -- reads beyond ~730 bp, or a small explicit
 * -W: every seed of the kept chains is re-scored by a local alignment of its neighbourhood (ksw_align2 semantics) and weak
 * seeds are dropped.  bmh_chain_batch runs the same filter on the device (round 3) for the reads it admits (up to 700 bases, i.e.
 * whenever a small -W switches it on); a batch with a longer read is refused there (BMH_EINVAL) and belongs here. */
typedef struct bmh_jobs bmh_jobs_t;
bmh_jobs_t *bmh_build_jobs(const bmh_chain_opt_t *opt, int64_t l_pac, const uint8_t *pac, int n_contigs,
                           const int64_t *contig_offset, const int32_t *contig_len, uint32_t n_reads,
                           const uint8_t *reads, const uint64_t *read_offs, const uint32_t *read_lens,
                           const uint64_t *rbeg, const int32_t *qbeg, const uint32_t *score,
                           const uint32_t *n_ref_pos, const uint32_t *prefix, int n_threads);
void bmh_jobs_free(bmh_jobs_t *j);
void bmh_jobs_sizes(const bmh_jobs_t *j, uint64_t *n_jobs, uint64_t *n_regs, uint64_t *q_bytes, uint64_t *t_bytes);
/* borrowed pointers into the job batch (valid until bmh_jobs_free): the bmh_extend_batch inputs, the
 * read / region / side (0 left, 1 right) of every job, and the number of regions of every read */
void bmh_jobs_arrays(const bmh_jobs_t *j, const uint8_t **q, const uint32_t **qoff, const uint32_t **qlen,
                     const uint8_t **t, const uint32_t **toff, const uint32_t **tlen, const uint32_t **h0,
                     const uint32_t **job_read, const uint32_t **job_reg, const uint32_t **job_side,
                     const uint32_t **regs_per_read);
/* extension results -> alignment regions (src/bwamem.c:2297-2303): regs_out[n_regs][8] =
 * {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi} */
int bmh_merge_regs(const bmh_jobs_t *j, const int32_t *out3, int32_t *regs_out);

/* per read: the frac_rep of its chains (mem_chain, src/bwamem.c:415-459), needed by the MAPQ estimate */
const float *bmh_jobs_frac_rep(const bmh_jobs_t *j);

/* ------------------------------------------------- after the extension: which regions are reported (SURVEY 8f rank 1 tail, rank 4) */

/* the options of mem_opt_t this stage reads beyond bmh_chain_opt_t / bmh_ext_params_t; bmh_post_opt_default() = mem_opt_init() */
typedef struct {
	int T;                      /* minimum score of a reported alignment (30) */
	float mask_level_redun;     /* 0.95 */
	float mapQ_coef_len;        /* 50 */
	int mapQ_coef_fac;          /* (int)log(50) = 3 */
	int flag_all;               /* MEM_F_ALL (-a): report secondary alignments too */
	int64_t id0;                /* index of the batch's first read in the run (n_processed): seeds the tie-break hash */
	float XA_drop_ratio;        /* 0.80: secondary hits scoring at least this share of their primary go to its XA tag */
	int max_XA_hits;            /* 5: ... when there are no more than this many */
	int no_multi;               /* MEM_F_NO_MULTI (-M): shorter split hits are flagged secondary (0x100) instead of supplementary */
	int softclip;               /* MEM_F_SOFTCLIP (-Y): soft clips on every record (no hard clips on the later ones) */
	int max_XA_hits_alt;        /* 200: ... or this many when one of them lies on an ALT contig (-h INT,INT) */
	/* ALT contigs: contig_is_alt[n_contigs] in HOST memory, NULL = none (src/bwamem.c:571-574,702,721-760,1540,1578,1663,1742,1755,
	 * 2323; src/bwamem_extra.c:106-150).  With a table that flags at least one sequence the records change in two places:
	 * [11] holds secondary_all (the parent of the FIRST marking round over all hits, which the XA tag goes by) instead of sub_n, and
	 * [15] = reported | is_alt << 1 | alt_sc << 2 ([12] secondary is INT_MAX for an ALT hit that has a parent, as in the reference).
	 * bmh_finalize_regs and bmh_finalize_regs_device both take the table (the device form uploads it; the second marking round,
	 * secondary_all and alt_sc are in csrc/regs_core.h and the wave classes of csrc/regs_kernels.hip): same records. */
	const uint8_t *contig_is_alt;
	/* read group (-R: bwa_set_rg, src/bwa.c:425-452): the ID of the @RG line, NUL-terminated, in HOST memory, at most 255 characters; NULL or empty = none.  Every
	 * record -- the unmapped ones too -- gets RG:Z:<id> behind AS / XS and in front of SA (mem_aln2sam, src/bwamem.c:1631-1634); the caller writes the @RG header line. */
	const char *rg_id;
} bmh_post_opt_t;
void bmh_post_opt_default(bmh_post_opt_t *o);

/* Per read: mem_sort_dedup_patch (redundant regions removed, colinear ones merged when a global alignment across both
 * scores well enough), mem_mark_primary_se (primary / secondary, sub-optimal score), mem_approx_mapq_se and the
 * selection of mem_reg2sam (src/bwamem.c:620-680, 685-760, 1690-1717, 1721-1770).  Host code, like the reference's.
 * regs_in[..][8] = bmh_merge_regs / bmh_chain_merge layout, grouped by read (regs_per_read); frac_rep per read.
 * out[..][16] = {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi, truesc, w, sub (XS), sub_n, secondary (index within
 * the read's output, -1 = primary line), MAPQ, flag (0x100 secondary, 0x800 supplementary), reported (0/1)} in the
 * reference's order, capacity = the number of input regions; out_per_read[n_reads].  Returns the number of output
 * regions or a negative BMH_E* code.  contig_offset: start of every sequence in the packed reference (bns->anns[i].offset;
 * n_contigs <= 1: one sequence).  ALT contigs: popt->contig_is_alt (see bmh_post_opt_t). */
int64_t bmh_finalize_regs(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, int64_t l_pac,
                          const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                          const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                          int n_contigs, const int64_t *contig_offset,
                          int32_t *out, uint32_t *out_per_read, int n_threads);

/* The same tail ON THE DEVICE (csrc/regs_kernels.hip; reads -> reportable alignments without leaving HBM): every pointer except
 * contig_offset is device memory.  d_regs[n_regs][8], d_regs_per_read, d_frac_rep: what bmh_chain_merge / bmh_chain_extend_merge
 * left (bmh_dev_jobs_t); d_reads / d_offs: the batch's ASCII reads; the index must carry the 2-bit reference.  d_out[n_regs][16]:
 * the records of bmh_finalize_regs, in the same order; d_out_per_read[n_reads].  Returns their number or a negative BMH_E* code;
 * waits for the stream.  Same results as bmh_finalize_regs, record for record (the logarithms of the MAPQ formula come from a
 * table computed by the host's libm).  Interleaved pairs still go through bmh_finalize_pairs. */
int64_t bmh_finalize_regs_device(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                                 const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                                 const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read, const float *d_frac_rep,
                                 int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream);
float bmh_finalize_regs_device_last_ms(void);       /* device time of the thread's last bmh_finalize_regs_device (HIP events) */

/* SAM records of single-end reads (mem_aln2sam, src/bwamem.c:1506-1683; XA tag: mem_gen_alt, src/bwamem_extra.c:97-150).
 * bmh_sam_need_cigar marks (need[i] = 1) the records of bmh_finalize_regs that must go through bmh_cigar_batch first --
 * the reported ones and the XA candidates -- and returns their number.  bmh_format_sam then takes, per record, its slot
 * in the bmh_cigar_batch outputs (slot[i], -1 = none) and returns the text (malloc'd; free with bmh_free), one line per
 * record in the reference's order, an unmapped record for reads without a reported alignment.  names: the read names,
 * NUL-terminated, back to back, name_off[r] the start of read r's.  Formats ranges of reads on host threads.  reads: nt4 codes (the
 * path reads FASTA: QUAL is '*').  ALT contigs: po->contig_is_alt (soft clips on ALT hits, the pa:f tag, the XA limits); pairs: bmh_format_sam_pe. */
int64_t bmh_sam_need_cigar(const bmh_post_opt_t *po, const int32_t *fin, const uint32_t *fin_per_read, uint32_t n_reads, uint8_t *need);
char *bmh_format_sam(const bmh_post_opt_t *po, uint32_t n_reads, const char *names, const uint64_t *name_off, const uint8_t *reads,
                     const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                     const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const int64_t *slot,
                     const int32_t *aln, const uint32_t *cigar, int max_cigar, const char *md, int md_cap, size_t *len_out);
void bmh_free(void *p);

/* ---- the read file: one '>' header line and one sequence line per read, the only layout the reference's seeding library parses
 * (src/GPUSeed/seed_gen.cu:1698-1728; the host takes the same file through kseq, bseq_read src/bwa.c:48-66).  Loads it into the flat
 * arrays this API takes: the letters back to back (what bmh_seed_batch / bmh_chain_* read in HBM), their nt4 codes (nst_nt4_table,
 * src/bntseq.c: what the host tail and bmh_format_sam read), offsets, lengths, and the names (the header up to the first blank,
 * NUL-terminated, back to back: bmh_format_sam's names / name_off).  Blank lines are skipped, CR LF line ends accepted; headers and
 * sequence lines that do not alternate: BMH_EINVAL.  n_threads <= 0: all host threads.  Arrays are malloc'd: bmh_reads_free. */
/* bmh_fasta_scan: reads, bases, name bytes and the longest read of such a file -- out[4] -- from one counting pass of the mapped file, without loading it
 * (what a driver looks at before it chooses between bmh_aligner_run_fasta and the paths that take longer reads). */
int bmh_fasta_scan(const char *path, int n_threads, uint64_t *out);
typedef struct {
	uint64_t n_reads, n_bases, n_name_bytes;
	uint8_t *ascii, *codes;        /* [n_bases] */
	uint64_t *offs;                /* [n_reads] start of read r in ascii / codes */
	uint32_t *lens;                /* [n_reads] */
	uint8_t *names;                /* [n_name_bytes] */
	uint64_t *name_offs;           /* [n_reads] */
} bmh_read_set_t;
int bmh_reads_load_fasta(const char *path, int n_threads, bmh_read_set_t *out);
void bmh_reads_free(bmh_read_set_t *r);

/* ---- interleaved pairs (read 2i, 2i+1): mem_pestat, mem_matesw (mate rescue, host local alignment), mem_pair, mem_sam_pe
 * (src/bwamem_pair.c).  Same inputs as bmh_finalize_regs plus read_lens and contig_len; out has room for `cap` records
 * (mate rescue adds regions: regions_in + 16 per read is ample).  out_h[r]: the record of read r's own alignment within
 * its list (-1 unmapped); out_unflag[r]: pair flags of the unmapped record of a read without reported alignment;
 * pes_out[4][5] (optional) = {low, high, failed, avg, std} per orientation FF, FR, RF, RR.  popt->id0 = index of the
 * batch's first READ in the run.  The insert-size statistics are those of the batch, as in the reference. */
typedef struct { int pen_unpaired, max_ins, max_matesw; int no_rescue, no_pairing; } bmh_pe_opt_t;       /* 17, 10000, 50; -S, -P */
void bmh_pe_opt_default(bmh_pe_opt_t *o);
int64_t bmh_finalize_pairs(const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                           int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                           const uint32_t *read_lens, const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                           int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                           int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, double *pes_out,
                           int n_threads);
/* bmh_finalize_pairs with the local alignments of the mate rescue (mem_matesw's ksw_align2 calls, 98 % of its time on a repeat-rich
 * batch) computed as ONE batch on the device (csrc/pair_kernels.hip: the striped kernel's lanes as GPU lanes).  idx: the index with
 * its 2-bit reference in HBM; d_reads / d_offs: the batch's ASCII reads in HBM (the same reads as `reads`); stream: waited for.
 * Same results as bmh_finalize_pairs, record for record. */
int64_t bmh_finalize_pairs_dev(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, void *stream,
                               const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                               int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                               const uint32_t *read_lens, const int32_t *regs_in, const uint32_t *regs_per_read, const float *frac_rep,
                               int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                               int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, double *pes_out,
                               int n_threads);
/* mem_sort_dedup_patch alone on the device (csrc/regs_kernels.hip: the first step of bmh_finalize_regs_device, the step the tail of interleaved
 * pairs shares with it): arguments as bmh_finalize_regs_device without d_frac_rep; d_out [n_regs][16] receives, grouped by read in the input's
 * order, the regions that are left -- [0] read, [1] score, [2] qb, [3] qe, [4..7] rb / re, [8] truesc, [9] w (a patched region carries a wider
 * band), [13] its sequence; the other fields are unspecified --, d_out_per_read their numbers.  Returns their total or a negative BMH_E* code
 * (BMH_ECAPACITY: a read beyond the kernels' fixed limits: the caller takes the host form); waits for the stream.  ALT tables are not looked at.
 * bmh_finalize_pairs_deduped: bmh_finalize_pairs_dev from those records (copied to the host) instead of the regions: same results. */
int64_t bmh_dedup_regs_device(const bmh_index_t *idx, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt,
                              const uint8_t *d_reads, const uint32_t *d_offs, uint32_t n_reads,
                              const int32_t *d_regs, uint64_t n_regs, const uint32_t *d_regs_per_read,
                              int n_contigs, const int64_t *contig_offset, int32_t *d_out, uint32_t *d_out_per_read, void *stream);
int64_t bmh_finalize_pairs_deduped(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, void *stream,
                                   const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep, const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe,
                                   int64_t l_pac, const uint8_t *pac, uint32_t n_reads, const uint8_t *reads, const uint64_t *read_offs,
                                   const uint32_t *read_lens, const int32_t *dedup_recs, const uint32_t *dedup_per_read, const float *frac_rep,
                                   int n_contigs, const int64_t *contig_offset, const int32_t *contig_len,
                                   int32_t *out, uint64_t cap, uint32_t *out_per_read, int32_t *out_h, int32_t *out_unflag, double *pes_out,
                                   int n_threads);
/* The windows of the mate rescue (mem_matesw up to its ksw_align2 call, src/bwamem_pair.c:119-150) are found by the host's walk of every pair; with the knob
 * ALIGNER_RESCUE_DEV=1 bmh_aligner_run finds them with a kernel on the regions the device keeps (csrc/pair_kernels.hip: rescue_jobs_kernel; same records,
 * measured slower on a busy device).  With the knob RESCUE_CHECK set as well the host's walk runs beside the kernel; out[5] = batches checked, pairs,
 * alignments the device asked for, pairs whose "a call reached a window" flag differs (0: the flag decides which pairs the host walks), pairs whose list of
 * calls differs (the host walk's mem_sort_dedup_patch calls can change a list; such a call is then computed by the host's second walk). */
void bmh_rescue_check_counts(uint64_t *out);
int64_t bmh_sam_need_cigar_pe(const bmh_post_opt_t *po, const int32_t *fin, const uint32_t *fin_per_read, const int32_t *h_rec,
                              uint32_t n_reads, uint8_t *need);
char *bmh_format_sam_pe(const bmh_post_opt_t *po, uint32_t n_reads, const char *names, const uint64_t *name_off, const uint8_t *reads,
                        const uint64_t *read_offs, const uint32_t *read_lens, int n_contigs, const char *const *contig_names,
                        const int64_t *contig_offset, const int32_t *fin, const uint32_t *fin_per_read, const int32_t *h_rec,
                        const int32_t *unflag, const int64_t *slot, const int32_t *aln, const uint32_t *cigar, int max_cigar,
                        const char *md, int md_cap, size_t *len_out);

/* ------------------------------------------------- device job builder (SURVEY 8f ranks 1-2 on the GPU) */

/* The same stage as bmh_build_jobs, on the device: seeds of bmh_seed_batch (still in HBM) -> chains -> filtered
 * chains -> regions and their LEFT/RIGHT extension jobs, with the query bases taken from the reads and the target
 * bases fetched from the index's 2-bit reference on the device (bns_fetch_seq, src/bntseq.c:531-580, and the
 * LEFT-side reversal of src/bwamem.c:1328-1334).  The batch is byte-identical to bmh_build_jobs' and feeds
 * bmh_extend_batch directly; bmh_chain_merge then turns the extension results into regions.
 * The index must have been uploaded with its pac. */
typedef struct bmh_chain_ws bmh_chain_ws_t;
bmh_chain_ws_t *bmh_chain_ws_create(uint32_t max_reads, uint64_t max_seeds);
void bmh_chain_ws_free(bmh_chain_ws_t *ws);
/* contig table of the reference (host arrays, as in bmh_build_jobs); default: one sequence of l_pac bases */
int bmh_chain_set_contigs(bmh_chain_ws_t *ws, int n_contigs, const int64_t *contig_offset, const int32_t *contig_len);
/* which of them are ALT contigs (is_alt[n_contigs], host memory; NULL or all zero: none) -- call after bmh_chain_set_contigs */
int bmh_chain_set_alt(bmh_chain_ws_t *ws, int n_contigs, const uint8_t *is_alt);

typedef struct {
	uint64_t n_jobs, n_regs, q_bytes, t_bytes;
	uint64_t n_heavy_reads;          /* reads chained by a whole wave (more than BMH_CHAIN_HEAVY=16 sampled seed occurrences) */
	const uint8_t *d_q; const uint32_t *d_qoff, *d_qlen;          /* the bmh_extend_batch inputs */
	const uint8_t *d_t; const uint32_t *d_toff, *d_tlen, *d_h0;
	const uint32_t *d_job_read, *d_job_reg, *d_job_side;           /* read / region / side (0 left, 1 right) per job */
	const uint32_t *d_regs_per_read;                               /* [n_reads] */
	const float *d_frac_rep;                                       /* [n_reads] frac_rep of the read's chains (for bmh_finalize_regs) */
} bmh_dev_jobs_t;

/* kernel times of the last bmh_chain_batch / bmh_chain_extend_merge in ms (HIP events): [0] classify [1] lane kernel (reads that
 * sample at most 16 seed occurrences) [2] wave kernels (the others, one wave per read, size classes by LDS scratch of 64 ... 1250
 * entries on side streams) [3] the stage up to the (first) counts; [6] reads whose scratch has at most 512 entries, [7] the larger ones */
void bmh_chain_last_timing(const bmh_chain_ws_t *ws, float ms[8]);

/* on (default): bmh_chain_batch also materialises the base arrays d_q/d_t/d_qoff/d_toff for bmh_extend_batch;
 * off: it stops at the job descriptors (those four pointers come back NULL, q_bytes = t_bytes = 0) and the batch is
 * extended with bmh_chain_extend, which reads the bases where they already are (reads, 2-bit reference). */
int bmh_chain_set_materialize(bmh_chain_ws_t *ws, int on);

/* d_reads/d_offs/d_lens: as for bmh_seed_batch; seeds: its output for the same reads.  Pointers in *out stay valid
 * until the next call on the workspace.  Synchronises the stream (once for the region/job counts, once more for the
 * base counts when materialising). */
int bmh_chain_batch(bmh_chain_ws_t *ws, const bmh_chain_opt_t *opt, const bmh_index_t *idx, const uint8_t *d_reads,
                    const uint32_t *d_offs, const uint32_t *d_lens, uint32_t n_reads, const bmh_seeds_t *seeds,
                    void *stream, bmh_dev_jobs_t *out);
/* bmh_extend_batch on the jobs of the last bmh_chain_batch, without materialised base arrays: same results (d_out3
 * [n_jobs][3], optional d_raw [n_jobs][6]).  d_reads and the index of that call must still be alive.  Asynchronous. */
int bmh_chain_extend(bmh_chain_ws_t *ws, const bmh_ext_params_t *p, int32_t *d_out3, int32_t *d_raw, void *stream);
/* d_out3 = extension results of the batch's jobs -> d_regs_out[n_regs][8] =
 * {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi} (src/bwamem.c:2297-2303).  Asynchronous on stream. */
int bmh_chain_merge(bmh_chain_ws_t *ws, const int32_t *d_out3, int32_t *d_regs_out, void *stream);

/* bmh_chain_batch + bmh_chain_extend + bmh_chain_merge in ONE call that hides the chaining of the seed-rich reads behind the
 * extension of the others (two passes: the reads that sample at most 16 seed occurrences are chained, turned into jobs and
 * extended while one wave per read is still chaining the rest on side streams; then the rest).  d_regs_out [cap_regs_out][8]
 * receives the regions in read order -- the same array the three-call form produces; out->n_regs / n_jobs / d_regs_per_read /
 * d_frac_rep as for bmh_chain_batch; the job arrays in *out are in pass order (all jobs of the first pass, then the second);
 * d_q / d_t are not materialised.  Synchronises the stream twice (counts); the second extension and the merge are still in
 * flight on `stream` when it returns.  The short kernels the first extension pass waits for (classification, the lane kernel, the
 * counts) run on a stream of the highest priority owned by the workspace, behind a host-side wait for `stream` (see bmh_seed_batch;
 * BMH_CHAIN_LIGHT_PRIO=normal when the workspace is created: on `stream`); the wave-per-read classes on streams of the lowest.  BMH_ECAPACITY if the batch has more than cap_regs_out regions. */
int bmh_chain_extend_merge(bmh_chain_ws_t *ws, const bmh_chain_opt_t *opt, const bmh_index_t *idx, const uint8_t *d_reads,
                           const uint32_t *d_offs, const uint32_t *d_lens, uint32_t n_reads, const bmh_seeds_t *seeds,
                           const bmh_ext_params_t *ep, int32_t *d_regs_out, uint64_t cap_regs_out, void *stream, bmh_dev_jobs_t *out);
/* of the last bmh_chain_extend_merge (waits for it): ms[0], ms[1] = extension of the two passes, ms[2] = the whole call's device
 * time; jobs[0], jobs[1] = extension jobs of the two passes */
int bmh_chain_extend_merge_timing(const bmh_chain_ws_t *ws, float ms[3], uint64_t jobs[2]);

/* ------------------------------------------------- region -> CIGAR / NM / MD */

/* mem_reg2aln (src/bwamem.c:2344-2440) for n regions: banded global alignment ksw_global2 (src/ksw.c:1120-1241) of
 * read[qb,qe) against the reference text [rb,re) with the band of bwa_gen_cigar2 (src/bwa.c:111-216), the retry with a
 * doubled band, NM, the MD string, the squeeze of a leading / trailing deletion, soft clips and the position.
 * d_regs: records of reg_stride int32: 8 = {read, score, qb, qe, rb_lo, rb_hi, re_lo, re_hi}, the layout bmh_chain_merge /
 * bmh_merge_regs write (truesc = score, region band = opt_w); 16 = the records of bmh_finalize_regs (truesc and the
 * region's band at [8], [9]: a patched region carries a wider band).  d_sel (optional): n indices into d_regs, NULL = the
 * first n regions.  opt_w = mem_opt_t.w (300).
 * Out, per job: d_cigar[max_cigar] (len << 4 | op, op 0 M, 1 I, 2 D, 3 S), d_aln[8] = {pos_lo, pos_hi (0-based,
 * forward strand of the concatenated reference), is_rev, n_cigar, NM, global score, MD length, flags (1 = more than
 * max_cigar-2 ops, 2 = interval rejected by bwa_gen_cigar2, 4 = too large, 8 = MD longer than md_cap)}, d_md[md_cap]
 * (NUL-terminated; d_md may be NULL).  The index must carry its pac.  Synchronises the stream once (sizes). */
int bmh_cigar_batch(const bmh_index_t *idx, const uint8_t *d_reads, const uint32_t *d_offs, const uint32_t *d_lens,
                    const int32_t *d_regs, int reg_stride, const uint32_t *d_sel, uint32_t n, const bmh_ext_params_t *p, int opt_w,
                    int max_cigar, uint32_t *d_cigar, int32_t *d_aln, int md_cap, char *d_md, void *stream);

/* frees the (device, stream) scratch of bmh_cigar_batch (device buffers it keeps between calls: the direction matrices of a batch are gigabytes);
 * the stream idle, its device current -- bmh_extend_release(stream) calls it too */
void bmh_cigar_release(void *stream);

/* ---- what the SAM writer needs of a batch, chosen and packed on the device (csrc/sam_kernels.hip)
 * bmh_sam_select_device: bmh_sam_need_cigar over records in HBM (d_fin [m][16], d_fin_per_read [n_reads]: what bmh_finalize_regs_device
 * left; d_h_rec: NULL, or for interleaved pairs the own-alignment record of every read as in bmh_sam_need_cigar_pe).  d_sel [m]
 * receives the indices of the selected records in ascending order -- the d_sel of bmh_cigar_batch --, d_slot [m] the place of every
 * record in that list or -1 (the slot[] of bmh_format_sam, 32-bit).  d_work: bmh_sam_select_work(n_reads, m) bytes of device memory.
 * Returns the number selected (waits for the stream) or a negative BMH_E* code.
 * bmh_cigar_pack_sizes / bmh_cigar_pack: the outputs of bmh_cigar_batch without their padding.  _sizes fills d_off [n + 1] with the
 * start (in 32-bit words) of every alignment in the packed array and returns the total number of words (waits for the stream):
 * n_cigar operations, then -- with_md -- the MD string with its NUL, padded to a word; an alignment flagged 1, 4 or 8 takes no words
 * (the caller redoes it with larger slots).  bmh_cigar_pack then copies (asynchronous; md_cap a multiple of 4; d_md NULL if _sizes
 * was called with with_md = 0).  d_work: bmh_cigar_pack_work(n) bytes. */
size_t bmh_sam_select_work(uint32_t n_reads, uint64_t m);
int64_t bmh_sam_select_device(const bmh_post_opt_t *popt, const int32_t *d_fin, const uint32_t *d_fin_per_read, const int32_t *d_h_rec,
                              uint32_t n_reads, uint64_t m, uint32_t *d_sel, int32_t *d_slot, void *d_work, size_t work_bytes, void *stream);
size_t bmh_cigar_pack_work(uint32_t n);
int64_t bmh_cigar_pack_sizes(const int32_t *d_aln, uint32_t n, int with_md, uint32_t *d_off, void *d_work, size_t work_bytes, void *stream);
int bmh_cigar_pack(const int32_t *d_aln, const uint32_t *d_cigar, int max_cigar, const char *d_md, int md_cap, uint32_t n,
                   const uint32_t *d_off, uint32_t *d_packed, void *stream);

/* ---- the SAM text itself written on the device (csrc/sam_kernels.hip): the text of bmh_format_sam / bmh_format_sam_pe, byte for byte,
 * from records, alignments and packed CIGARs that never leave HBM.  Every pointer of bmh_sam_dev_t is device memory.  ALT contigs: with
 * popt->contig_is_alt set the records are ALT-mode records (bmh_post_opt_t: [11] the XA tag's key, [15] carries is_alt and alt_sc) and the
 * text follows (soft clips on ALT hits, the larger XA limit, pa:f written as printf's %.3f).
 * bmh_sam_text_sizes: first pass, d_text_off [n_reads + 1] = start of every read's records in the text; returns the text's length
 * (waits for the stream).  bmh_sam_text_write: second pass into d_text (asynchronous; the same d_work, untouched in between);
 * bmh_sam_text_check afterwards waits for the stream and reports an inconsistency between the passes.  d_work: bmh_sam_text_work bytes. */
typedef struct {
	uint32_t n_reads;
	const char *d_names; const uint64_t *d_name_off;           /* names NUL-terminated back to back; [n_reads + 1]: start of read r's, [n_reads] = their total length */
	const uint8_t *d_reads; const uint32_t *d_offs, *d_lens;   /* the batch's ASCII reads (what bmh_seed_batch takes) */
	int n_contigs; const char *d_contig_names; const uint32_t *d_contig_name_off; const int64_t *d_contig_offset;   /* names as above, [n_contigs + 1] */
	const int32_t *d_fin; const uint32_t *d_fin_per_read;      /* records [m][16], records per read */
	const int32_t *d_slot;                                     /* [m] record -> alignment (bmh_sam_select_device) */
	const int32_t *d_aln; const uint32_t *d_cig_off, *d_packed; /* bmh_cigar_batch's d_aln; bmh_cigar_pack's d_off / d_packed (with MD) */
	const int32_t *d_h_rec, *d_unflag;                         /* interleaved pairs (bmh_finalize_pairs' out_h / out_unflag) or NULL */
} bmh_sam_dev_t;
size_t bmh_sam_text_work(uint32_t n_reads);
int64_t bmh_sam_text_sizes(const bmh_post_opt_t *popt, const bmh_sam_dev_t *d, uint64_t *d_text_off, void *d_work, size_t work_bytes, void *stream);
int bmh_sam_text_write(const bmh_post_opt_t *popt, const bmh_sam_dev_t *d, const uint64_t *d_text_off, char *d_text, void *d_work, size_t work_bytes, void *stream);
int bmh_sam_text_check(const void *d_work, uint32_t n_reads, void *stream);

/* ------------------------------------------------- reads in host memory -> SAM text, batches driven by C threads (csrc/align_pipeline.hip)
 *
 * What gase_aln's worker threads do around the device libraries (src/bwamem.c:2042-2340, src/fastmap.c:59-120), on top of the entry
 * points above: n_lanes worker threads take the batches [cuts[b], cuts[b+1]) of the read set in turn -- each with its own stream,
 * workspaces and pinned staging: H2D, seeding, chaining, extension, merge, the region tail, the selection of the records that need a
 * CIGAR, CIGAR / NM / MD, and the SAM text itself (bmh_sam_text_*), all on the device; the text goes to the host -- and ONE writer
 * thread hands every batch's text to `sink` in order (return 0 to go on) while the workers are on the next ones.  Interleaved pairs:
 * mem_sort_dedup_patch, mem_mark_primary_se and, for the pairs the mate rescue does not touch, mem_pair and mem_sam_pe's choices on
 * the device too; the insert-size statistics, the rescue's bookkeeping and the pairs it touches on n_threads host threads in the
 * middle of the batch.  An index with ALT contigs: the same (the device tail and the pairing kernel know the ALT rules).  A batch the device tail refuses (BMH_ECAPACITY)
 * takes the host tail.  The text is what
 * bmh_format_sam / bmh_format_sam_pe write, byte for byte (records only: the caller writes the @SQ header).
 * cuts: n_batches + 1 read indices, cuts[0] = 0, cuts[n_batches] = n_reads, even batch sizes when paired (the reference cuts its
 * batches by bases, bseq_read src/bwa.c:48-66, and the insert-size statistics are those of a batch).  popt->id0 is ignored (a batch's
 * id0 is its first read).  Reads longer than 700 bases: BMH_EINVAL (the device job builder's limit; such a set goes through
 * bmh_build_jobs).  n_threads: host threads of the host forms (<= 0: all).  The aligner borrows idx and pac: both outlive it; popt->rg_id is copied. */
int bmh_effective_cpus(void);      /* CPUs this process may use: affinity mask capped by the cgroup quota (what n_threads <= 0 resolves to) */
typedef struct bmh_aligner bmh_aligner_t;
typedef int (*bmh_sam_sink_t)(void *user, const char *text, size_t len);
typedef struct {
	uint64_t n_reads, n_bytes; uint32_t n_batches; int n_lanes;
	double seconds;                 /* wall clock of the run */
	double format_seconds;          /* in the writer thread (overlaps the workers) */
	double h2d_seconds, seed_seconds, chain_extend_seconds, tail_seconds, select_seconds, cigar_seconds;    /* summed over the lanes' host clocks */
	double gate_wait_seconds;       /* lanes waiting for a slot in the path's device stages (ALIGNER_GPU_SLOTS), summed likewise */
	/* the copies themselves, timed by HIP events on the lanes' streams (h2d_seconds / cigar_seconds above are HOST clocks around whole stages: the staging of
	 * letters, offsets and names on host threads; the CIGAR and text kernels the host waits for): reads + offsets + names in, SAM text out */
	double h2d_copy_seconds, d2h_copy_seconds; uint64_t h2d_bytes, d2h_bytes;
} bmh_align_stats_t;
/* Page-locks a caller's host buffer (hipHostRegister) / releases it.  bmh_aligner_run sends the batches of a read set whose letters lie in pinned or registered
 * memory to the device straight from there (no staging copy on host threads). */
int bmh_host_pin(void *p, size_t bytes);
int bmh_host_unpin(void *p);
bmh_aligner_t *bmh_aligner_create(const bmh_index_t *idx, const uint8_t *pac, int64_t l_pac, int n_contigs, const char *const *contig_names,
                                  const int32_t *contig_len, const uint8_t *contig_is_alt, const bmh_chain_opt_t *copt, const bmh_ext_params_t *ep,
                                  const bmh_post_opt_t *popt, const bmh_pe_opt_t *pe);
void bmh_aligner_free(bmh_aligner_t *a);
int bmh_aligner_run(bmh_aligner_t *a, const bmh_read_set_t *reads, const uint64_t *cuts, uint32_t n_batches, int paired, int n_lanes, int n_threads,
                    bmh_sam_sink_t sink, void *user, bmh_align_stats_t *stats);
/* The same from a read FILE (one '>' header line and one sequence line per read, as bmh_reads_load_fasta takes), batch by batch: a loader thread cuts the
 * mapped file the way bseq_read cuts its stream -- reads are added until the batch holds at least batch_bases bases (or exactly batch_reads reads, if that is
 * not 0) and, paired, an even number of reads (src/bwa.c:48-66; the reference's batch_bases is 10 000 000 x its thread count) -- and fills every batch into
 * pinned host memory while the workers are on the batches before it; nothing of the file is held beyond the n_lanes + 3 batches in flight.  Same text as
 * bmh_reads_load_fasta + bmh_aligner_run with the same cuts. */
int bmh_aligner_run_fasta(bmh_aligner_t *a, const char *reads_fa, uint64_t batch_bases, uint64_t batch_reads, int paired, int n_lanes, int n_threads,
                          bmh_sam_sink_t sink, void *user, bmh_align_stats_t *stats);

#ifdef __cplusplus
}
#endif
#endif
