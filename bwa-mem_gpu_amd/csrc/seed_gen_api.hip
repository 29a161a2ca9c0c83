// Reference-compatible seeding entry points (include/seed_gen.h) on top of the
// device-level API.  Replaces the host side of
// /root/reference/src/GPUSeed/seed_gen.cu: loaders :1386-1468, gpu_cpy_wrapper
// :1524-1556, seed_gpu :1625-2188, free_gpuseed_data :1567-1573.
//
// seed_gpu keeps the reference's contract -- it parses the read file itself
// (single-line FASTA, every non-'>' line is one read, seed_gen.cu:1698-1728),
// seeds it batch by batch and returns malloc()'d flat arrays the caller frees
// with free() (src/fastmap.c:537-542) -- but batches are sized for 288 GB of
// HBM (SEED_BATCH_READS reads per launch set instead of ~1 Mbase), reads are
// staged through pinned host memory, and the per-batch H2D/D2H copies are
// asynchronous on one stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <memory>
#include <mutex>
#include <algorithm>
#include <vector>
#include "bmh_internal.h"
#include "../../include/seed_gen.h"

#define FATAL(...) do { fprintf(stderr, "[bwamem_hip] " __VA_ARGS__); fprintf(stderr, "\n"); exit(EXIT_FAILURE); } while (0)
#define HIPX(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FATAL("%s: %s", #x, hipGetErrorString(e_)); } while (0)

static uint64_t g_last_n_reads = 0;
extern "C" uint64_t seed_gpu_last_n_reads(void) { return g_last_n_reads; }

extern "C" bwt_t_gpu *bwt_restore_bwt_gpu(const char *fn)
{
	FILE *fp = fopen(fn, "rb");
	if (!fp) FATAL("Unable to open .bwt file %s", fn);
	bwt_t_gpu *bwt = (bwt_t_gpu *)calloc(1, sizeof(bwt_t_gpu));
	fseek(fp, 0, SEEK_END);
	long sz = ftell(fp);
	if (sz < 40) FATAL("%s: truncated", fn);
	bwt->bwt_size = (uint64_t)(sz - 40) >> 2;
	// pinned host staging, like the reference (cudaMallocHost, seed_gen.cu:1454-1455); padded to whole blocks
	size_t words = (size_t)bwt->bwt_size + 16;
	HIPX(hipHostMalloc((void **)&bwt->bwt, words * 4, hipHostMallocDefault));
	memset(bwt->bwt, 0, words * 4);
	bwt->L2 = (bwtint_t_gpu *)calloc(5, sizeof(bwtint_t_gpu));
	fseek(fp, 0, SEEK_SET);
	if (fread(&bwt->primary, 8, 1, fp) != 1 || fread(bwt->L2 + 1, 8, 4, fp) != 4 ||
	    fread(bwt->bwt, 4, bwt->bwt_size, fp) != bwt->bwt_size) FATAL("%s: short read", fn);
	bwt->seq_len = bwt->L2[4];
	fclose(fp);
	return bwt;
}

extern "C" void bwt_restore_sa_gpu(const char *fn, bwt_t_gpu *bwt)
{
	FILE *fp = fopen(fn, "rb");
	if (!fp) FATAL("Unable to open .sa file %s", fn);
	uint64_t primary, skipped[4], intv, seq_len;
	if (fread(&primary, 8, 1, fp) != 1 || fread(skipped, 8, 4, fp) != 4 || fread(&intv, 8, 1, fp) != 1 ||
	    fread(&seq_len, 8, 1, fp) != 1) FATAL("%s: short header", fn);
	if (primary != bwt->primary) FATAL("SA-BWT inconsistency: primary is not the same.");
	if (seq_len != bwt->seq_len) FATAL("SA-BWT inconsistency: seq_len is not the same.");
	bwt->sa_intv = (int)intv;
	bwt->n_sa = (bwt->seq_len + bwt->sa_intv) / bwt->sa_intv;
	HIPX(hipHostMalloc((void **)&bwt->sa, bwt->n_sa * 4, hipHostMallocDefault));
	bwt->sa[0] = (uint32_t)-1;
	if (fread(bwt->sa + 1, 4, bwt->n_sa - 1, fp) != bwt->n_sa - 1) FATAL("%s: short SA", fn);
	if (fread(&bwt->pack_size, 1, 1, fp) != 1) FATAL("%s: no pack_size", fn);
	if (bwt->pack_size != 1) FATAL("%s: pack_size %d unsupported (seq_len >= 2^33)", fn, bwt->pack_size);
	size_t nb = (size_t)bwt->pack_size * bwt->n_sa / 32 + 1;
	HIPX(hipHostMalloc((void **)&bwt->sa_upper_bits, nb * 4, hipHostMallocDefault));
	if (fread(bwt->sa_upper_bits, 4, nb, fp) != nb) FATAL("%s: short bit array", fn);
	bwt->sa_upper_bits[0] |= 0x1;
	fclose(fp);
}

extern "C" void bwt_destroy_gpu(bwt_t_gpu *bwt)
{
	if (!bwt) return;
	if (bwt->sa) (void)hipHostFree(bwt->sa);
	if (bwt->sa_upper_bits) (void)hipHostFree(bwt->sa_upper_bits);
	if (bwt->bwt) (void)hipHostFree(bwt->bwt);
	free(bwt->L2);
	free(bwt);
}

extern "C" bwt_t_gpu gpu_cpy_wrapper(bwt_t_gpu *bwt)
{
	bwt_t_gpu g;
	memset(&g, 0, sizeof(g));
	size_t bwt_bytes = ((size_t)((bwt->seq_len + 63) / 64) + 1) * 32;
	size_t nb = (size_t)bwt->pack_size * bwt->n_sa / 32 + 1;
	HIPX(hipMalloc((void **)&g.bwt, bwt_bytes));
	HIPX(hipMalloc((void **)&g.sa, bwt->n_sa * 4));
	HIPX(hipMalloc((void **)&g.sa_upper_bits, nb * 4));
	HIPX(hipMemset(g.bwt, 0, bwt_bytes));
	size_t cp = (size_t)bwt->bwt_size * 4 < bwt_bytes ? (size_t)bwt->bwt_size * 4 : bwt_bytes;
	HIPX(hipMemcpy(g.bwt, bwt->bwt, cp, hipMemcpyHostToDevice));
	HIPX(hipMemcpy(g.sa, bwt->sa, bwt->n_sa * 4, hipMemcpyHostToDevice));
	HIPX(hipMemcpy(g.sa_upper_bits, bwt->sa_upper_bits, nb * 4, hipMemcpyHostToDevice));
	g.pack_size = bwt->pack_size; g.primary = bwt->primary; g.seq_len = bwt->seq_len;
	g.sa_intv = bwt->sa_intv; g.n_sa = bwt->n_sa;
	g.bwt_size = bwt_bytes / 4;
	g.L2 = (bwtint_t_gpu *)malloc(5 * sizeof(bwtint_t_gpu));   // host copy (see include/seed_gen.h)
	memcpy(g.L2, bwt->L2, 5 * sizeof(bwtint_t_gpu));
	bwt_destroy_gpu(bwt);      // the reference frees the host copy here too (seed_gen.cu:1553)
	return g;
}

extern "C" void pre_calc_seed_intervals_wrapper(uint2 *, int, bwt_t_gpu)
{
	// exported by the reference but never called (src/fastmap.c:455 sets the flag to 0)
	FATAL("pre_calc_seed_intervals_wrapper: not used by the pipeline (reference fastmap.c:455)");
}

extern "C" void free_gpuseed_data(gpuseed_storage_vector *d)
{
	if (!d) return;
	if (d->bwt_gpu.bwt) (void)hipFree(d->bwt_gpu.bwt);
	if (d->bwt_gpu.sa) (void)hipFree(d->bwt_gpu.sa);
	if (d->bwt_gpu.sa_upper_bits) (void)hipFree(d->bwt_gpu.sa_upper_bits);
	free(d->bwt_gpu.L2);
	memset(&d->bwt_gpu, 0, sizeof(d->bwt_gpu));
}

#ifndef SEED_BATCH_READS
#define SEED_BATCH_READS (1u << 20)
#endif
#define SEED_BATCH_BASES ((uint64_t)SEED_BATCH_READS * 320)

extern "C" mem_seed_v_gpu *seed_gpu(gpuseed_storage_vector *d)
{
	if (!d->is_smem) FATAL("seed_gpu: MEM mode (-g) is not implemented; only SMEM seeding (the default, fastmap.c:442)");
	const bwt_t_gpu &g = d->bwt_gpu;
	if (!g.bwt || !g.L2) FATAL("seed_gpu: index not on the device (call gpu_cpy_wrapper first)");
	bmh_index_t *idx = bmh_index_from_device(g.primary, g.L2, g.seq_len, g.bwt, g.bwt_size, g.sa_intv, g.sa, g.n_sa,
	                                         g.sa_upper_bits, nullptr, 0);
	if (!idx) FATAL("seed_gpu: %s", bmh_last_error());
	{   // denser suffix-array samples than the files hold (every 4th row; BMH_SA_INTV=16 keeps the file's): milliseconds
		// on the device, 3.4x fewer index gathers per located seed
		static const int want = [] { const char *e = getenv("BMH_SA_INTV"); const int v = e ? atoi(e) : 4; return v >= 1 ? v : 4; }();
		if (bmh_index_densify_sa(idx, want) != BMH_OK) FATAL("seed_gpu: %s", bmh_last_error());
	}
	FILE *fp = fopen(d->read_file, "r");
	if (!fp) FATAL("seed_gpu: cannot open %s", d->read_file);

	// size the batch by the file: at most one read per two bytes, one base per byte
	fseek(fp, 0, SEEK_END);
	uint64_t fsz = (uint64_t)ftell(fp);
	fseek(fp, (long)d->file_bytes_skip, SEEK_SET);
	uint64_t left = fsz > d->file_bytes_skip ? fsz - d->file_bytes_skip : 0;
	// (BMH_SEED_BATCH_READS: smaller batches, so that a small test file still has several of them for the BMH_DEVICES workers)
	static const uint32_t batch_reads_cfg = [] { const char *e = getenv("BMH_SEED_BATCH_READS"); const long v = e ? atol(e) : 0; return v > 0 ? (uint32_t)v : (uint32_t)SEED_BATCH_READS; }();
	const uint32_t BATCH_READS = (uint32_t)(left / 2 + 1 < batch_reads_cfg ? left / 2 + 1 : batch_reads_cfg);
	const uint64_t BATCH_BASES = left + 1 < SEED_BATCH_BASES ? left + 1 : SEED_BATCH_BASES;

	// The file is read batch by batch, by whichever worker is free next, straight into that worker's pinned staging buffers (so
	// the H2D copy is one asynchronous DMA and the host never holds more reads than the workers are seeding); what stays per
	// batch is its seeds, in file order.
	struct batch_t { std::vector<uint64_t> rbeg; std::vector<int2> qbeg; std::vector<uint32_t> score, n_ref; };
	std::vector<std::unique_ptr<batch_t>> batches;
	std::mutex rd_mu;
	uint64_t file_bytes = 0;
	bool rd_eof = false;
	char *rd_line = nullptr; size_t rd_cap = 0;
	// next batch of the file into (bases, offs, lens); returns its index, or -1 at the end of the file
	auto read_batch = [&](uint8_t *bases, uint32_t *offs, uint32_t *lens, uint32_t *n_reads, uint64_t *n_bases) -> long {
		std::lock_guard<std::mutex> lk(rd_mu);
		uint64_t nb = 0; uint32_t nr = 0;
		while (!rd_eof && nr < BATCH_READS) {
			ssize_t n = getline(&rd_line, &rd_cap, fp);
			if (n < 0) { rd_eof = true; break; }
			const size_t raw = (size_t)n;
			file_bytes += raw;
			if (rd_line[0] == '>') continue;
			while (n > 0 && (rd_line[n - 1] == '\n' || rd_line[n - 1] == '\r')) --n;
			if (n == 0) continue;                     // blank line (kseq skips it too)
			if (nb + (uint64_t)n > BATCH_BASES) {  // batch full: push the line back
				fseek(fp, -(long)raw, SEEK_CUR); file_bytes -= raw;
				break;
			}
			memcpy(bases + nb, rd_line, (size_t)n);
			offs[nr] = (uint32_t)nb; lens[nr] = (uint32_t)n; ++nr;
			nb += (uint64_t)n;
		}
		*n_reads = nr; *n_bases = nb;
		if (nr == 0) return -1;
		batches.emplace_back(new batch_t());
		return (long)batches.size() - 1;
	};

	// BMH_DEVICES=N: N worker threads, worker k on device k mod (devices present), each with its own stream, workspace and --
	// on another device than the one the index was uploaded to -- its own replica of the index (bmh_index_replicate)
	static const int n_workers_env = [] { const char *e = getenv("BMH_DEVICES"); const int v = e ? atoi(e) : 1; return v >= 1 ? v : 1; }();
	int n_dev = 1, home = 0;
	HIPX(hipGetDeviceCount(&n_dev)); HIPX(hipGetDevice(&home));
	// (no more workers than the file can have batches)
	const uint64_t max_batches = std::max<uint64_t>(1, std::max<uint64_t>((left / 2 + BATCH_READS) / BATCH_READS, (left + BATCH_BASES) / BATCH_BASES));
	const int n_workers = (int)std::min<uint64_t>((uint64_t)n_workers_env, max_batches);
	std::vector<bmh_index_t *> widx(n_workers, idx);
	std::vector<int> wdev(n_workers, home);
	for (int k = 1; k < n_workers; ++k) wdev[k] = (home + k) % n_dev;
	// the index on every worker's device: one grouped RCCL broadcast per array over xGMI (hipMemcpyPeer copies where RCCL is absent)
	if (n_workers > 1 && bmh_index_replicate_all(idx, home, wdev.data(), n_workers, widx.data(), nullptr) != BMH_OK) FATAL("seed_gpu: %s", bmh_last_error());
	const int min_seed = d->min_seed_size;
	auto work = [&](int k) {
		HIPX(hipSetDevice(wdev[k]));
		hipStream_t st;
		HIPX(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
		// candidate capacity = the hard bound (one per base): hard read sets need more than the library's default guess
		bmh_seed_ws_t *ws = bmh_seed_ws_create(BATCH_READS, BATCH_BASES, BATCH_BASES, 0);
		if (!ws) FATAL("seed_gpu: %s", bmh_last_error());
		uint8_t *d_bases; uint32_t *d_offs, *d_lens;
		HIPX(hipMalloc((void **)&d_bases, BATCH_BASES));
		HIPX(hipMalloc((void **)&d_offs, (size_t)BATCH_READS * 4));
		HIPX(hipMalloc((void **)&d_lens, (size_t)BATCH_READS * 4));
		// pinned staging: the batch's reads on the way in, its seeds on the way out (grown by half beyond the need when a batch has more)
		uint8_t *h_bases; uint32_t *h_offs, *h_lens;
		HIPX(hipHostMalloc((void **)&h_bases, BATCH_BASES, hipHostMallocDefault));
		HIPX(hipHostMalloc((void **)&h_offs, (size_t)BATCH_READS * 4, hipHostMallocDefault));
		HIPX(hipHostMalloc((void **)&h_lens, (size_t)BATCH_READS * 4, hipHostMallocDefault));
		uint8_t *h_out = nullptr; size_t h_out_cap = 0;
		for (;;) {
			uint32_t nr = 0; uint64_t nbases = 0;
			const long bi = read_batch(h_bases, h_offs, h_lens, &nr, &nbases);
			if (bi < 0) break;
			HIPX(hipMemcpyAsync(d_bases, h_bases, nbases, hipMemcpyHostToDevice, st));
			HIPX(hipMemcpyAsync(d_offs, h_offs, (size_t)nr * 4, hipMemcpyHostToDevice, st));
			HIPX(hipMemcpyAsync(d_lens, h_lens, (size_t)nr * 4, hipMemcpyHostToDevice, st));
			bmh_seeds_t s;
			if (bmh_seed_batch(ws, widx[k], d_bases, d_offs, d_lens, nr, min_seed, st, &s) != BMH_OK)
				FATAL("seed_gpu: %s", bmh_last_error());
			const size_t need = (size_t)s.n_seeds * 20 + (size_t)nr * 4;
			if (need > h_out_cap) {
				if (h_out) HIPX(hipHostFree(h_out));
				h_out_cap = need + need / 2;
				HIPX(hipHostMalloc((void **)&h_out, h_out_cap, hipHostMallocDefault));
			}
			uint8_t *p_rbeg = h_out, *p_qbeg = p_rbeg + (size_t)s.n_seeds * 8, *p_score = p_qbeg + (size_t)s.n_seeds * 8, *p_nref = p_score + (size_t)s.n_seeds * 4;
			if (s.n_seeds) {
				HIPX(hipMemcpyAsync(p_rbeg, s.d_rbeg, s.n_seeds * 8, hipMemcpyDeviceToHost, st));
				HIPX(hipMemcpyAsync(p_qbeg, s.d_qbeg, s.n_seeds * 8, hipMemcpyDeviceToHost, st));
				HIPX(hipMemcpyAsync(p_score, s.d_score, s.n_seeds * 4, hipMemcpyDeviceToHost, st));
			}
			HIPX(hipMemcpyAsync(p_nref, s.d_n_ref_pos, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
			HIPX(hipStreamSynchronize(st));
			batch_t *b;
			{ std::lock_guard<std::mutex> lk(rd_mu); b = batches[(size_t)bi].get(); }
			b->rbeg.assign((const uint64_t *)p_rbeg, (const uint64_t *)p_rbeg + s.n_seeds);
			b->qbeg.assign((const int2 *)p_qbeg, (const int2 *)p_qbeg + s.n_seeds);
			b->score.assign((const uint32_t *)p_score, (const uint32_t *)p_score + s.n_seeds);
			b->n_ref.assign((const uint32_t *)p_nref, (const uint32_t *)p_nref + nr);
		}
		if (h_out) HIPX(hipHostFree(h_out));
		HIPX(hipHostFree(h_bases)); HIPX(hipHostFree(h_offs)); HIPX(hipHostFree(h_lens));
		(void)hipFree(d_bases); (void)hipFree(d_offs); (void)hipFree(d_lens);
		bmh_seed_ws_free(ws);
		(void)hipStreamDestroy(st);
	};
	{
		std::vector<std::thread> th;
		for (int k = 1; k < n_workers; ++k) th.emplace_back(work, k);
		work(0);
		for (auto &t : th) t.join();
		HIPX(hipSetDevice(home));
	}
	free(rd_line);
	fclose(fp);
	for (int k = 1; k < n_workers; ++k) {
		if (widx[k] == idx) continue;
		bool first = true;
		for (int q = 1; q < k; ++q) if (widx[q] == widx[k]) first = false;
		if (first) { HIPX(hipSetDevice(wdev[k])); bmh_index_free(widx[k]); HIPX(hipSetDevice(home)); }
	}
	// the run's arrays (caller frees with free(), src/fastmap.c:537-542): the batches' seeds in file order, each copied once
	size_t ns = 0, nr = 0;
	for (const auto &b : batches) { ns += b->rbeg.size(); nr += b->n_ref.size(); }
	if ((uint64_t)ns >> 32) FATAL("seed_gpu: more than 2^32 seeds in one run (32-bit prefix sums, seed_gen.h:73)");
	mem_seed_v_gpu *out = (mem_seed_v_gpu *)malloc(sizeof(mem_seed_v_gpu));
	out->rbeg = (bwtint_t_gpu *)malloc((ns + 1) * 8);
	out->qbeg = (int2 *)malloc((ns + 1) * sizeof(int2));
	out->score = (uint32_t *)malloc((ns + 1) * 4);
	out->n_ref_pos_fow_rev_results = (uint32_t *)malloc((nr + 1) * 4);
	out->n_ref_pos_fow_rev_prefix_sums = (uint32_t *)malloc((nr + 1) * 4);
	{
		size_t so = 0, ro = 0;
		for (auto &b : batches) {
			const size_t bs = b->rbeg.size(), br = b->n_ref.size();
			if (bs) { memcpy(out->rbeg + so, b->rbeg.data(), bs * 8); memcpy(out->qbeg + so, b->qbeg.data(), bs * sizeof(int2)); memcpy(out->score + so, b->score.data(), bs * 4); }
			if (br) memcpy(out->n_ref_pos_fow_rev_results + ro, b->n_ref.data(), br * 4);
			so += bs; ro += br;
			b.reset();
		}
	}
	const uint32_t *n_ref = out->n_ref_pos_fow_rev_results;
	uint32_t acc = 0;
	for (size_t i = 0; i < nr; ++i) { out->n_ref_pos_fow_rev_prefix_sums[i] = acc; acc += n_ref[i]; }
	out->file_bytes_skip = file_bytes;
	g_last_n_reads = nr;

	bmh_index_free(idx);
	return out;
}
