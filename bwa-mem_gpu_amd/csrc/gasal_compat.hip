// GASAL2-source-compatible layer (include/gasal2_root/GASAL2/include/*.h) over bmh_extend_batch.
// Replaces what the reference links from the un-vendored GASAL2 submodule: the batch containers and
// stream objects of src/fastmap.c:473-534 and the fill / launch / poll calls of
// src/bwamem.c:1102-1167, 2106-2211.  One storage = one HIP stream + pinned host staging + device
// buffers; gasal_aln_async enqueues H2D, the extension kernels and D2H on that stream and returns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include "bmh_internal.h"
#include "../../include/gasal2_root/GASAL2/include/gasal.h"
#include "../../include/gasal2_root/GASAL2/include/args_parser.h"
#include "../../include/gasal2_root/GASAL2/include/host_batch.h"
#include "../../include/gasal2_root/GASAL2/include/gasal_align.h"
#include "../../include/gasal2_root/GASAL2/include/ctors.h"
#include "../../include/gasal2_root/GASAL2/include/interfaces.h"

#define FATAL(...) do { fprintf(stderr, "[bwamem_hip/gasal] " __VA_ARGS__); fprintf(stderr, "\n"); exit(EXIT_FAILURE); } while (0)
#define HIPX(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FATAL("%s: %s", #x, hipGetErrorString(e_)); } while (0)

static bmh_ext_params_t g_params = {1, 4, 6, 1, 6, 1, 0, 5};

void gasal_copy_subst_scores(gasal_subst_scores *s)
{
	g_params.a = s->match; g_params.b = s->mismatch;
	g_params.o_del = g_params.o_ins = s->gap_open;
	g_params.e_del = g_params.e_ins = s->gap_extend;
}
void gasal_set_ksw_extras(int end_bonus, int zdrop) { g_params.end_bonus = end_bonus; g_params.zdrop = zdrop; }

Parameters::Parameters(int argc_, char **argv_)
{
	sa = 1; sb = 4; gapo = 6; gape = 1; start_pos = WITHOUT_START; print_out = 0; n_threads = 1; k_band = 0;
	secondBest = false; isPacked = false; isReverseComplement = false;
	semiglobal_skipping_head = NONE; semiglobal_skipping_tail = NONE; algo = KSW; argc = argc_; argv = argv_;
}
Parameters::~Parameters() {}
void Parameters::print() {}
void Parameters::failure(int) {}
void Parameters::help() {}
void Parameters::parse() {}
void Parameters::fileopen() {}

void gasal_set_device(int gpu_select, bool) { HIPX(hipSetDevice(gpu_select)); }

struct storage_impl {
	int dev;                      // the device this storage lives on (BMH_DEVICES=N spreads the storages over the GPUs)
	hipStream_t stream;
	uint8_t *d_q, *d_t;
	uint32_t *d_qoff, *d_toff, *d_qlen, *d_tlen, *d_h0;
	int32_t *d_out, *h_out;       // interleaved {score, qend, tend}
	uint32_t d_q_cap, d_t_cap, d_n_cap;
	uint32_t n_launched;
	bool running;
};

host_batch_t *gasal_host_batch_new(uint32_t batch_bytes, uint32_t offset)
{
	host_batch_t *p = (host_batch_t *)calloc(1, sizeof(host_batch_t));
	HIPX(hipHostMalloc((void **)&p->data, batch_bytes ? batch_bytes : 8, hipHostMallocDefault));
	p->page_size = batch_bytes; p->data_size = 0; p->offset = offset; p->is_locked = 0; p->next = NULL;
	return p;
}
void gasal_host_batch_destroy(host_batch_t *p)
{
	while (p) { host_batch_t *n = p->next; (void)hipHostFree(p->data); free(p); p = n; }
}

// the unpacked host batch is ONE growable page (offset 0, next NULL): decoy_cpu_align-style readers that
// walk the page list still work, and the whole batch goes to the GPU in a single copy
static void page_reserve(host_batch_t *p, uint32_t need, uint32_t *host_max)
{
	if (need <= p->page_size) return;
	uint32_t ns = p->page_size ? p->page_size : 4096;
	while (ns < need) ns = ns < (1u << 30) ? ns * 2 : ns + (1u << 30);
	uint8_t *nd;
	HIPX(hipHostMalloc((void **)&nd, ns, hipHostMallocDefault));
	memcpy(nd, p->data, p->data_size);
	(void)hipHostFree(p->data);
	p->data = nd; p->page_size = ns;
	if (host_max && *host_max < ns) *host_max = ns;
}

uint32_t gasal_host_batch_fill(gasal_gpu_storage_t *s, uint32_t idx, const uint8_t *data, uint32_t size, data_source SRC)
{
	host_batch_t *p = SRC == QUERY ? s->extensible_host_unpacked_query_batch : s->extensible_host_unpacked_target_batch;
	uint32_t *hm = SRC == QUERY ? &s->host_max_query_batch_bytes : &s->host_max_target_batch_bytes;
	uint32_t padded = (size + 7u) & ~7u;
	page_reserve(p, idx + padded, hm);
	memcpy(p->data + idx, data, size);
	memset(p->data + idx + size, N_CODE, padded - size);
	if (p->data_size < idx + padded) p->data_size = idx + padded;
	return idx + padded;
}
uint32_t gasal_host_batch_fill(gasal_gpu_storage_t *s, uint32_t idx, const char *data, uint32_t size, data_source SRC)
{
	return gasal_host_batch_fill(s, idx, (const uint8_t *)data, size, SRC);
}
uint32_t gasal_host_batch_addbase(gasal_gpu_storage_t *s, uint32_t idx, const char base, data_source SRC)
{
	host_batch_t *p = SRC == QUERY ? s->extensible_host_unpacked_query_batch : s->extensible_host_unpacked_target_batch;
	uint32_t *hm = SRC == QUERY ? &s->host_max_query_batch_bytes : &s->host_max_target_batch_bytes;
	page_reserve(p, idx + 1, hm);
	p->data[idx] = (uint8_t)base;
	if (p->data_size < idx + 1) p->data_size = idx + 1;
	return idx + 1;
}

gasal_res_t *gasal_res_new_host(uint32_t n, Parameters *)
{
	gasal_res_t *r = (gasal_res_t *)calloc(1, sizeof(gasal_res_t));
	r->aln_score = (int32_t *)calloc(n ? n : 1, 4); r->query_batch_end = (int32_t *)calloc(n ? n : 1, 4);
	r->target_batch_end = (int32_t *)calloc(n ? n : 1, 4); r->query_batch_start = (int32_t *)calloc(n ? n : 1, 4);
	r->target_batch_start = (int32_t *)calloc(n ? n : 1, 4);
	return r;
}
void gasal_res_destroy_host(gasal_res_t *r)
{
	if (!r) return;
	free(r->aln_score); free(r->query_batch_end); free(r->target_batch_end); free(r->query_batch_start); free(r->target_batch_start);
	free(r);
}

static void host_alns_alloc(gasal_gpu_storage_t *s, uint32_t n_new, uint32_t n_old)
{
	auto grow = [&](uint32_t *&p) {
		uint32_t *q;
		HIPX(hipHostMalloc((void **)&q, (size_t)n_new * 4, hipHostMallocDefault));
		if (p) { memcpy(q, p, (size_t)n_old * 4); (void)hipHostFree(p); }
		p = q;
	};
	grow(s->host_query_batch_offsets); grow(s->host_target_batch_offsets);
	grow(s->host_query_batch_lens); grow(s->host_target_batch_lens); grow(s->host_seed_scores);
	gasal_res_t *nr = gasal_res_new_host(n_new, NULL);
	if (s->host_res) {
		memcpy(nr->aln_score, s->host_res->aln_score, (size_t)n_old * 4);
		memcpy(nr->query_batch_end, s->host_res->query_batch_end, (size_t)n_old * 4);
		memcpy(nr->target_batch_end, s->host_res->target_batch_end, (size_t)n_old * 4);
		gasal_res_destroy_host(s->host_res);
	}
	s->host_res = nr;
	s->host_max_n_alns = n_new;
}

void gasal_host_alns_resize(gasal_gpu_storage_t *s, int new_max_alns, Parameters *)
{
	if ((uint32_t)new_max_alns <= s->host_max_n_alns) return;
	host_alns_alloc(s, (uint32_t)new_max_alns, s->host_max_n_alns);
}

gasal_gpu_storage_v gasal_init_gpu_storage_v(int n_streams)
{
	gasal_gpu_storage_v v;
	v.n = n_streams;
	v.a = (gasal_gpu_storage_t *)calloc(n_streams, sizeof(gasal_gpu_storage_t));
	return v;
}

void gasal_init_streams(gasal_gpu_storage_v *v, int host_max_q, int gpu_max_q, int host_max_t, int gpu_max_t, int host_max_n,
                        int gpu_max_n, Parameters *params)
{
	if (params && params->algo != KSW) FATAL("gasal_init_streams: only algo == KSW is implemented (the reference uses nothing else)");
	// BMH_DEVICES=N: the storages (the reference creates one vector of them per host thread, src/fastmap.c:487-507) are dealt out
	// to the N devices in turn, starting from the calling thread's current device; every later call on a storage switches to its
	// device.  The extension needs nothing but the batch itself on the device, so this is all the multi-GPU support it takes.
	static const int n_devices_env = [] { const char *e = getenv("BMH_DEVICES"); const int q = e ? atoi(e) : 1; return q >= 1 ? q : 1; }();
	static std::atomic<unsigned> next_storage{0};
	int n_dev = 1, home = 0;
	HIPX(hipGetDeviceCount(&n_dev)); HIPX(hipGetDevice(&home));
	for (int i = 0; i < v->n; ++i) {
		gasal_gpu_storage_t *s = &v->a[i];
		storage_impl *m = (storage_impl *)calloc(1, sizeof(storage_impl));
		s->impl = m;
		m->dev = n_devices_env > 1 ? (home + (int)(next_storage++ % (unsigned)n_devices_env)) % n_dev : home;
		HIPX(hipSetDevice(m->dev));
		HIPX(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
		s->host_max_query_batch_bytes = host_max_q; s->host_max_target_batch_bytes = host_max_t;
		s->gpu_max_query_batch_bytes = gpu_max_q; s->gpu_max_target_batch_bytes = gpu_max_t; s->gpu_max_n_alns = gpu_max_n;
		// host pages start small and grow on demand (the reference's sizes are upper-bound guesses, fastmap.c:487-507)
		uint32_t q0 = (uint32_t)host_max_q < (1u << 20) ? (uint32_t)host_max_q : (1u << 20);
		uint32_t t0 = (uint32_t)host_max_t < (1u << 21) ? (uint32_t)host_max_t : (1u << 21);
		s->extensible_host_unpacked_query_batch = gasal_host_batch_new(q0, 0);
		s->extensible_host_unpacked_target_batch = gasal_host_batch_new(t0, 0);
		host_alns_alloc(s, (uint32_t)host_max_n, 0);
		s->current_n_alns = 0;
		s->is_free = 1;
	}
	HIPX(hipSetDevice(home));
}

static void dev_reserve(storage_impl *m, uint32_t qb, uint32_t tb, uint32_t n)
{
	if (qb > m->d_q_cap) { if (m->d_q) (void)hipFree(m->d_q); m->d_q_cap = qb + qb / 4 + 64; HIPX(hipMalloc((void **)&m->d_q, m->d_q_cap)); }
	if (tb > m->d_t_cap) { if (m->d_t) (void)hipFree(m->d_t); m->d_t_cap = tb + tb / 4 + 64; HIPX(hipMalloc((void **)&m->d_t, m->d_t_cap)); }
	if (n > m->d_n_cap) {
		void *ps[] = {m->d_qoff, m->d_toff, m->d_qlen, m->d_tlen, m->d_h0, m->d_out};
		for (void *p : ps) if (p) (void)hipFree(p);
		if (m->h_out) (void)hipHostFree(m->h_out);
		m->d_n_cap = n + n / 4 + 64;
		size_t b = (size_t)m->d_n_cap * 4;
		HIPX(hipMalloc((void **)&m->d_qoff, b)); HIPX(hipMalloc((void **)&m->d_toff, b)); HIPX(hipMalloc((void **)&m->d_qlen, b));
		HIPX(hipMalloc((void **)&m->d_tlen, b)); HIPX(hipMalloc((void **)&m->d_h0, b)); HIPX(hipMalloc((void **)&m->d_out, b * 3));
		HIPX(hipHostMalloc((void **)&m->h_out, b * 3, hipHostMallocDefault));
	}
}

void gasal_aln_async(gasal_gpu_storage_t *s, const uint32_t qb, const uint32_t tb, const uint32_t n, Parameters *params)
{
	if (params && params->algo != KSW) FATAL("gasal_aln_async: only algo == KSW is implemented");
	storage_impl *m = (storage_impl *)s->impl;
	if (!m) FATAL("gasal_aln_async: storage not initialised (gasal_init_streams)");
	if (n == 0) return;
	if (n > s->host_max_n_alns) FATAL("gasal_aln_async: %u alignments > host_max_n_alns %u", n, s->host_max_n_alns);
	if (qb > s->extensible_host_unpacked_query_batch->data_size || tb > s->extensible_host_unpacked_target_batch->data_size)
		FATAL("gasal_aln_async: batch bytes beyond what was filled");
	// the DP kernels take queries of up to 768 bases (GASAL2 has a compile-time MAX_SEQ_LEN as well, README.md:38): refuse loudly
	for (uint32_t i = 0; i < n; ++i)
		if (s->host_query_batch_lens[i] > 768u) FATAL("gasal_aln_async: alignment %u has a query of %u bases; this library supports up to 768", i, s->host_query_batch_lens[i]);
	// BMH_GASAL_DUMP=<file>: append every submitted job (qlen, tlen, h0, bases) -- lets tests compare the job
	// stream of the reference's own host code with bmh_build_jobs
	static const char *dump = getenv("BMH_GASAL_DUMP");
	if (dump) {
		static std::mutex mu;
		std::lock_guard<std::mutex> lk(mu);
		FILE *f = fopen(dump, "ab");
		if (f) {
			for (uint32_t i = 0; i < n; ++i) {
				uint32_t h[3] = {s->host_query_batch_lens[i], s->host_target_batch_lens[i], s->host_seed_scores[i]};
				fwrite(h, 4, 3, f);
				fwrite(s->extensible_host_unpacked_query_batch->data + s->host_query_batch_offsets[i], 1, h[0], f);
				fwrite(s->extensible_host_unpacked_target_batch->data + s->host_target_batch_offsets[i], 1, h[1], f);
			}
			fclose(f);
		}
	}
	// BMH_GASAL_SYNC=1 (debug aid): one submission at a time, complete before returning
	static const bool serial = getenv("BMH_GASAL_SYNC") != nullptr;
	static std::mutex serial_mu;
	std::unique_lock<std::mutex> serial_lk(serial_mu, std::defer_lock);
	if (serial) serial_lk.lock();
	HIPX(hipSetDevice(m->dev));
	dev_reserve(m, qb, tb, n);
	hipStream_t st = m->stream;
	HIPX(hipMemcpyAsync(m->d_q, s->extensible_host_unpacked_query_batch->data, qb, hipMemcpyHostToDevice, st));
	HIPX(hipMemcpyAsync(m->d_t, s->extensible_host_unpacked_target_batch->data, tb, hipMemcpyHostToDevice, st));
	HIPX(hipMemcpyAsync(m->d_qoff, s->host_query_batch_offsets, (size_t)n * 4, hipMemcpyHostToDevice, st));
	HIPX(hipMemcpyAsync(m->d_toff, s->host_target_batch_offsets, (size_t)n * 4, hipMemcpyHostToDevice, st));
	HIPX(hipMemcpyAsync(m->d_qlen, s->host_query_batch_lens, (size_t)n * 4, hipMemcpyHostToDevice, st));
	HIPX(hipMemcpyAsync(m->d_tlen, s->host_target_batch_lens, (size_t)n * 4, hipMemcpyHostToDevice, st));
	HIPX(hipMemcpyAsync(m->d_h0, s->host_seed_scores, (size_t)n * 4, hipMemcpyHostToDevice, st));
	if (bmh_extend_batch(m->d_q, m->d_qoff, m->d_qlen, m->d_t, m->d_toff, m->d_tlen, m->d_h0, n, &g_params, m->d_out, nullptr, st) != BMH_OK)
		FATAL("gasal_aln_async: %s", bmh_last_error());
	HIPX(hipMemcpyAsync(m->h_out, m->d_out, (size_t)n * 12, hipMemcpyDeviceToHost, st));
	if (serial) HIPX(hipStreamSynchronize(st));
	m->n_launched = n; m->running = true;
	s->is_free = 0;
}

int gasal_is_aln_async_done(gasal_gpu_storage_t *s)
{
	storage_impl *m = (storage_impl *)s->impl;
	if (!m || !m->running) return -2;
	HIPX(hipSetDevice(m->dev));
	hipError_t e = hipStreamQuery(m->stream);
	if (e == hipErrorNotReady) return -1;
	if (e != hipSuccess) FATAL("gasal_is_aln_async_done: %s", hipGetErrorString(e));
	gasal_res_t *r = s->host_res;
	for (uint32_t i = 0; i < m->n_launched; ++i) {
		r->aln_score[i] = m->h_out[3 * i]; r->query_batch_end[i] = m->h_out[3 * i + 1]; r->target_batch_end[i] = m->h_out[3 * i + 2];
	}
	// BMH_GASAL_CHECK=1: bounds every result must satisfy (score <= h0 + qlen*a, ends inside the sequences)
	static const bool check = getenv("BMH_GASAL_CHECK") != nullptr;
	if (check) {
		unsigned long bad = 0;
		for (uint32_t i = 0; i < m->n_launched; ++i) {
			int32_t sc = r->aln_score[i], qe = r->query_batch_end[i], te = r->target_batch_end[i];
			int64_t ub = (int64_t)s->host_seed_scores[i] + (int64_t)s->host_query_batch_lens[i] * g_params.a + g_params.end_bonus;
			if (sc < 0 || sc > ub || qe < 0 || te < 0 || (uint32_t)qe > s->host_query_batch_lens[i] || (uint32_t)te > s->host_target_batch_lens[i]) {
				if (bad++ < 4) fprintf(stderr, "[gasal check] job %u of %u: score %d qend %d tend %d (h0 %u qlen %u tlen %u)\n", i, m->n_launched, sc, qe, te,
				                       s->host_seed_scores[i], s->host_query_batch_lens[i], s->host_target_batch_lens[i]);
			}
		}
		if (bad) fprintf(stderr, "[gasal check] %lu results out of bounds\n", bad);
	}
	m->running = false;
	s->is_free = 1;
	s->current_n_alns = 0;
	return 0;
}

void gasal_destroy_streams(gasal_gpu_storage_v *v, Parameters *)
{
	for (int i = 0; i < v->n; ++i) {
		gasal_gpu_storage_t *s = &v->a[i];
		storage_impl *m = (storage_impl *)s->impl;
		if (!m) continue;
		(void)hipSetDevice(m->dev);
		(void)hipStreamSynchronize(m->stream);
		bmh_extend_release(m->stream);                         // the extension's per-stream scratch goes with the stream
		void *ps[] = {m->d_q, m->d_t, m->d_qoff, m->d_toff, m->d_qlen, m->d_tlen, m->d_h0, m->d_out};
		for (void *p : ps) if (p) (void)hipFree(p);
		if (m->h_out) (void)hipHostFree(m->h_out);
		(void)hipStreamDestroy(m->stream);
		free(m); s->impl = NULL;
		gasal_host_batch_destroy(s->extensible_host_unpacked_query_batch);
		gasal_host_batch_destroy(s->extensible_host_unpacked_target_batch);
		uint32_t *hs[] = {s->host_query_batch_offsets, s->host_target_batch_offsets, s->host_query_batch_lens, s->host_target_batch_lens, s->host_seed_scores};
		for (uint32_t *p : hs) if (p) (void)hipHostFree(p);
		gasal_res_destroy_host(s->host_res);
		memset(s, 0, sizeof(*s));
	}
}

void gasal_destroy_gpu_storage_v(gasal_gpu_storage_v *v) { free(v->a); v->a = NULL; v->n = 0; }
